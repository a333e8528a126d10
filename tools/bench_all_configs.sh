#!/bin/bash
# One bench line per configuration of the baseline list (single GPU), default batch sizes.
for wl in l_mcs0_1000B_snr30 vht_mcs4_1500B_snr30 vht_mcs8_1500B_snr30 ht_mcs15_2x2_1500B_snr30 mixed_l_ht_vht_snr30; do
  timeout -k 10 300 python3 bench.py --workload $wl --cpu-frames 4000 2>/dev/null | grep '^{' || exit 1
done
