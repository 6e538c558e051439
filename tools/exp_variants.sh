#!/bin/bash
# Build an experimental variant of libnbls.so with extra compiler flags into noble-bls12-381_amd/variants/ (git-ignored);
# run it with NBLS_LIBRARY=<path> tools/exp_time.py.   Usage: tools/exp_variants.sh <name> "<extra hipcc flags>"
set -e
cd "$(dirname "$0")/../noble-bls12-381_amd/csrc"
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -Wno-unused-value -Wno-unused-result -shared -mllvm -align-all-nofallthru-blocks=6 $2 -I../../include -o ../variants/libnbls_$1.so runtime.cpp tuning.cpp pipelines_pairing.cpp pipelines_codec.cpp pipelines_verify.cpp nbls_multi.cpp vm_wide_kernel.hip vm_kernel.hip pow_kernels.hip xmd_kernel.hip msm_kernels.hip trace.cpp programs.cpp
ls -la ../variants
