#!/bin/bash
# The timed rotation by the number of pooled buffers: survivor buffers (--pool) and the soft-bit buffers beyond them
# (GR80211_POOL_SOFT_EXTRA), against a set of their own per context (--pool 0); alternating on one box.
for i in 1 2; do
for cfg in "0 1" "3 1" "3 0" "2 1" "2 2" "4 0"; do
set -- $cfg
echo "== survivor buffers=$1 soft-bit buffers=+$2 run $i"
GR80211_POOL_SOFT_EXTRA=$2 timeout -k 10 300 python3 bench.py --steps 40 --pool $1 --cpu-frames 0 --latency 0 --in-flight 1 --profile-steps 0 2>&1 | grep '^{\|Error' | python3 -c "
import json,sys
t=sys.stdin.read()
try:
    r=json.loads(t); print(round(r['value'],1), round(r['ms_per_step'],3), round(r['device_memory']['in_use_GB_after_timed_steps'],1), round(r['device_memory']['library_work_buffers_GB'],1), r['psdu']['bit_exact_vs_tx'])
except Exception: print('failed', t[-300:])"
done; done
