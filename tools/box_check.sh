# is this box one of the slow ones?  interpreter against ahead-of-time kernels at one 4096-pairing call, host probe, clocks under load
python tools/host_probe.py 2>/dev/null
NBLS_AOT=1 python tools/exp_time.py 4096 10 2>&1 | tail -1 | cut -c1-260
NBLS_AOT=0 NBLS_LS_MAX=0 python tools/exp_time.py 4096 10 2>&1 | tail -1 | cut -c1-260
(python tools/exp_time.py 4096 300 > /dev/null 2>&1 &) ; sleep 4; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
env | grep -i "^HSA\|^HIP\|^ROC\|^GPU\|^AMD" | head
