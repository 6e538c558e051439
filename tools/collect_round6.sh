#!/bin/bash
# copies the summaries of tools/profile_round6.sh (gpurun_out/prof_r6/) that are kept into profiles/round6_*
s=gpurun_out/prof_r6; d=profiles
for f in kernel_stats_b4096.csv kernel_stats_b4096_inflight12.csv kernel_stats_b65536.csv pmc_b4096.csv pmc_b65536.csv pmc_b4096.json pmc_b65536.json inflight12_span.json bench_default.json bench_driver_args.json \
         verify_breakdown_n65536.txt verify_breakdown_n1.txt verify_timeline_n65536.txt verify_timeline_n1.txt ab_latency.txt wide_time.txt ab_verify.txt ab_split_min.txt sign_sizes.txt init_time.txt pair_ab.txt clocks_under_load.txt stress_parity.txt; do
  [ -f $s/$f ] && cp $s/$f $d/round6_$f
done
cp $s/hbm_traffic.json $d/hbm_traffic.json
ls -la $d/round6_*
