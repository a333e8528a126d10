for args in "" "--pool 2" "--pool 0"; do
echo "== force-collective $args"
timeout -k 10 300 python3 bench.py --force-collective --cpu-frames 0 --steps 20 --latency 0 --profile-steps 0 $args 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['without_gather']['ms_per_step'])" || exit 1
done
echo "== 16 queues default"
GR80211_BENCH_HW_QUEUES=16 timeout -k 10 300 python3 bench.py --force-collective --cpu-frames 0 --steps 20 --latency 0 --profile-steps 0 2>/dev/null | grep '^{' | python3 -c "
import json,sys
r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['without_gather']['ms_per_step'])"
