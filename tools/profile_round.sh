#!/bin/bash
# Collects the evidence kept under profiles/: the bench line, the rocprofv3 kernel-trace statistics of the same
# command, and the counter passes (one rocprofv3 --pmc run each: HBM traffic, vector-instruction issue, clock).  The
# profiled runs take `--in-flight 1`: the batches_in_flight leg of the default command runs kernels of two contexts
# side by side, as do the timed steps of the default command (two contexts in rotation), which stretches every launch and would blur the per-launch averages the roofline figure is checked
# against.  Run on the GPU box from the repository root:
#   bash tools/profile_round.sh <tag> [bench.py options]
set -u
TAG=${1:-r02}
shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
echo "[profile_round] bench"; timeout -k 10 400 python3 bench.py "$@" > "$OUT/bench.log" 2>"$OUT/bench.err" || exit 1
echo "[profile_round] kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 --profile-steps 0 --in-flight 1 --pipeline 1 "$@" > "$OUT/kt.log" 2>&1 || exit 2
python3 tools/rocpd_summary.py "$OUT"/kt/kt_results.db > "$OUT/kernel_stats.txt" 2>&1
rm -rf "$OUT/kt"
pass() {  # <name> <counters...>
    local name=$1; shift
    echo "[profile_round] pmc pass $name: $*"
    timeout -s KILL 300 rocprofv3 --kernel-trace --pmc "$@" -d "$OUT/p_$name" -o p -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --profile-steps 0 --in-flight 1 --pipeline 1 $BENCH_EXTRA > "$OUT/p_$name.log" 2>&1 || return 1
    python3 tools/rocpd_summary.py "$OUT/p_$name"/p_results.db > "$OUT/pmc_$name.txt" 2>&1
    rm -rf "$OUT/p_$name"
}
BENCH_EXTRA="$*"
pass fetch FETCH_SIZE || exit 3
pass write WRITE_SIZE || exit 4
pass sq_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES || exit 5
pass sq_cycles SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 6
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 7
echo "[profile_round] done"
