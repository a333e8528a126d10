#!/bin/bash
# The timed rotation by the number of hardware queues the process asks for (GR80211_BENCH_HW_QUEUES -> GPU_MAX_HW_QUEUES),
# plain and with the one-rank collective, alternating on one box.
for i in 1 2; do
for q in 8 16; do
for args in "" "--force-collective"; do
echo "== queues=$q $args run $i"
GR80211_BENCH_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --cpu-frames 0 --steps 40 --latency 0 --profile-steps 0 --in-flight 1 $args 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],3), (r.get('without_gather') or {}).get('ms_per_step'))" || exit 1
done; done; done
