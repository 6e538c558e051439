#!/bin/bash
# shader clock and package power while the engine runs saturated (65,536-pairing calls back to back) and while it runs one 4096-pairing call at a time: is the saturated
# figure power-limited on this box?  Usage (GPU box): bash tools/clocks_under_load.sh
smp() { for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -iE "sclk|Power \(W\)|Socket Power" | tr -s ' ' | tr '\n' ';'; echo; sleep 0.7; done; }
echo "== idle"; smp | tail -2
(python tools/exp_time.py 65536 400 > /tmp/load1.txt 2>&1 &) ; sleep 6; echo "== saturated (65,536-pairing calls)"; smp; wait; sleep 1; tail -1 /tmp/load1.txt | cut -c1-80
(python tools/exp_time.py 4096 3000 > /tmp/load2.txt 2>&1 &) ; sleep 4; echo "== one 4096-pairing call at a time"; smp; sleep 3; tail -1 /tmp/load2.txt | cut -c1-80
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -iE "max|perf" | head -4
