#!/bin/bash
# The timed rotation with the contexts' own work buffers (--pool 0) and on a pool of three work sets (--pool 3),
# alternating on one box; ARGS adds bench.py options.
for i in 1 2; do
for pool in 0 3; do
echo "== pool=$pool run $i ${ARGS:-}"
timeout -k 10 300 python3 bench.py --steps 40 --pool $pool ${ARGS:-} 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['one_batch_at_a_time']['ms_per_step'], r['device_memory']['in_use_GB_after_timed_steps'])" || exit 1
done; done
