#!/bin/bash
# VGPR / SGPR / spill / LDS figures of every kernel of libnbls.so, read from the code-object metadata the compiler emits (hipcc -S of each .hip file with the
# flags of csrc/Makefile) -- no GPU needed.  Usage: tools/kernel_resources.sh | grep -v rocprim > profiles/round3_kernel_resources.txt   (the hipCUB sort kernels of the MSM are left out)
cd "$(dirname "$0")/../noble-bls12-381_amd/csrc"
# aot_kernel.hip is compiled once per part (aot.h NBLS_AOT_PARTS; Makefile -DNBLS_AOT_PART=i)
for f in vm_kernel.hip aot_kernel.hip:0 aot_kernel.hip:1 aot_kernel.hip:2 aot_kernel.hip:3 aot_kernel.hip:4 aot_kernel.hip:5 aot_kernel.hip:6 aot_kernel.hip:7 pow_kernels.hip xmd_kernel.hip msm_kernels.hip; do
  part=${f#*:}; [ "$part" = "$f" ] && part=0; f=${f%%:*}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -align-all-nofallthru-blocks=6 -I../../include -DNBLS_AOT_PART=$part -S --cuda-device-only -o /tmp/kres_$$.s $f 2>/dev/null
  python3 - /tmp/kres_$$.s $f <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for blk in txt.split('  - .agpr_count:')[1:]:
    g = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    print('%-16s %-34s vgpr %-4s agpr %-3s sgpr %-4s vgpr_spill %-3s sgpr_spill %-3s static_lds %-6s scratch %s' % (sys.argv[2], g('name'), g('vgpr_count'), blk.split()[0], g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('group_segment_fixed_size'), g('private_segment_fixed_size')))
PY
  rm -f /tmp/kres_$$.s
done
