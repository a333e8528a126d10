#!/bin/bash
# HBM traffic of one step with and without the axis-value soft bits (GR80211_AXIS_LLR), one-wave decoder form in both:
# FETCH_SIZE and WRITE_SIZE in a counter pass of their own per workload and switch position.
set -u
export GR80211_GROUP_WAVES=1
for wl in vht_mcs8_1500B_snr30 ht_mcs15_2x2_1500B_snr30; do
  for ax in 0 1; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
      echo "== $wl axis=$ax $ctr"
      GR80211_AXIS_LLR=$ax BENCH_ARGS="--workload $wl --in-flight 1 --pipeline 1 --latency 0" bash tools/pmc_pass.sh axis_${wl}_${ax}_$ctr $ctr || exit 1
    done
  done
done
