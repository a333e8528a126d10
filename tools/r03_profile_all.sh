#!/bin/bash
# tools/profile_round.sh for the workloads named on the command line (default: all four), each into
# gpurun_out/r03_<workload>/, plus the counters file bench.py reads for that workload.
set -u
WLS=${*:-vht_mcs8_1500B_snr30 vht_mcs4_1500B_snr30 ht_mcs15_2x2_1500B_snr30 mixed_l_ht_vht_snr30}
declare -A FR=([vht_mcs8_1500B_snr30]=100000 [vht_mcs4_1500B_snr30]=10000 [ht_mcs15_2x2_1500B_snr30]=50000 [mixed_l_ht_vht_snr30]=125000 [l_mcs0_1000B_snr30]=10000)
for wl in $WLS; do
  echo "=== $wl"
  bash tools/profile_round.sh r03_$wl --workload $wl --cpu-frames 4000 --latency 0 || exit 1
  python3 tools/make_counters_json.py gpurun_out/r03_$wl $wl ${FR[$wl]} r03_$wl > gpurun_out/r03_$wl/counters.json || exit 2
done
