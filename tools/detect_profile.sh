#!/bin/bash
# Kernel-trace statistics of the frame detection / capture path: bash tools/detect_profile.sh <tag>
set -u
TAG=${1:-det}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --profile-steps 0 --in-flight 1 --detect > "$OUT/kt.log" 2>&1 || { tail -5 "$OUT/kt.log"; exit 2; }
python3 tools/rocpd_summary.py "$OUT"/kt/kt_results.db > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/kt"
grep '^{' "$OUT/kt.log" | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['detect'])"
head -30 "$OUT/kernel_stats.txt" | cut -c1-150
