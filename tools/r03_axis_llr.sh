#!/bin/bash
# The axis-value soft-bit experiment (GR80211_AXIS_LLR=1) beside the default path: parity through the GPU suite's
# single-modulation cases, then step time one batch at a time and in rotation for the two configurations whose
# soft-bit traffic is largest.  Prints one line per run.
set -u
export PYTHONUNBUFFERED=1
echo "== parity under the switch (cases with one modulation per group)"
timeout -k 10 600 python3 -u tools/axis_parity.py 2>&1 | tail -50 || exit 1
for wl in vht_mcs8_1500B_snr30 ht_mcs15_2x2_1500B_snr30; do
  for ax in 0 1; do
    for pipe in 1 0; do
      echo "== $wl axis=$ax pipeline=$pipe"
      GR80211_AXIS_LLR=$ax GR80211_GROUP_WAVES=1 timeout -k 10 300 python3 bench.py --workload $wl --cpu-frames 0 --latency 0 --pipeline $pipe --steps 20 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.read())
print({k:r[k] for k in ('value','ms_per_step')}, r.get('kernel_ms'), r.get('psdu'))" || exit 1
    done
  done
done
