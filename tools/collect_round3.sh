#!/bin/bash
# copies the summaries of tools/profile_round3.sh (gpurun_out/prof_r3/) that are kept into profiles/round3_*
s=gpurun_out/prof_r3; d=profiles
for f in kernel_stats_b4096.csv kernel_stats_b4096_inflight12.csv kernel_stats_b65536.csv pmc_b4096.csv pmc_b65536.csv inflight12_span.json bench_default.json bench_driver_args.json verify_breakdown_n65536.txt verify_breakdown_n1.txt; do cp $s/$f $d/round3_$f; done
cp $s/hbm_traffic.json $d/hbm_traffic.json
ls -la $d/round3_*
