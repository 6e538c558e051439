#!/bin/bash
# shader clock and package power under a saturated stream of the engine's multiply-add (tools/ubench/mad_power.hip): the power-limited peak of v_mad_u64_u32 on this box.
# Usage (GPU box): bash tools/mad_power.sh
cd "$(dirname "$0")/ubench"; [ -x mad_power -a mad_power -nt mad_power.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o mad_power mad_power.hip || exit 1
smp() { for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|Socket Graphics Package Power" | sed 's/.*: //' | tr '\n' ' '; echo; sleep 0.6; done; }
for m in ${MODES:-0 1 3}; do      # mode 2 (operands through LDS) reads with dword instructions and is LDS-bound: not the engine's 16-byte reads, kept for reference only for w in ${WPS:-8 3 1}; do (./mad_power $m 4 $w > /tmp/mp.txt 2>&1 &); sleep 1.5; smp | tail -2; wait; sleep 2.6; cat /tmp/mp.txt; done; done
