#!/bin/bash
# The timed rotation by the number of contexts it rotates over (16 hardware queues, pooled work buffers as bench.py
# takes them), alternating on one box.
for i in 1 2; do
for p in 6 8 10; do
echo "== contexts=$p run $i"
timeout -k 10 300 python3 bench.py --cpu-frames 0 --steps 48 --latency 0 --profile-steps 0 --in-flight 1 --pipeline $p 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['device_memory']['in_use_GB_after_timed_steps'],1))" || exit 1
done; done
