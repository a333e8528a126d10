#!/bin/bash
# One rocprofv3 counter pass over a single bench step; prints the per-kernel averages.
#   bash tools/pmc_pass.sh <tag> <counter> [<counter> ...]      (at most 8 SQ_ / 4 TCC_ / 4 TCP_ counters per pass)
# BENCH_ARGS adds bench.py options (e.g. --detect), KERNELS is the pattern of the kernels to print.
set -u
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/p" -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --profile-steps 0 ${BENCH_ARGS:-} > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 3; }
python3 tools/rocpd_summary.py "$OUT"/p/p_results.db > "$OUT/summary.txt" 2>&1
rm -rf "$OUT/p"
grep -E "${KERNELS:-k_demod|k_preamble|k_decode}" "$OUT/summary.txt"
