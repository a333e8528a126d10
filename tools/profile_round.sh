#!/bin/bash
# Collects the evidence kept under profiles/: the bench line, the rocprofv3 kernel-trace statistics of the same
# command, and the HBM traffic counters (one PMC pass each).  The profiled runs take `--in-flight 1`: the
# batches_in_flight leg of the default command runs kernels of two contexts side by side, which stretches every
# launch and would blur the per-launch averages the roofline figure is checked against.  Run on the GPU box from
# the repository root:
#   bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 400 python3 bench.py --steps 5 --warmup 2 > "$OUT/bench.log" 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 bench.py --steps 5 --warmup 2 --cpu-frames 0 --profile-steps 0 --in-flight 1 > "$OUT/kt.log" 2>&1 || exit 2
python3 tools/rocpd_summary.py "$OUT"/kt/kt_results.db > "$OUT/kernel_stats.txt" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pf" -o pf -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --profile-steps 0 --in-flight 1 > "$OUT/pf.log" 2>&1 || exit 3
python3 tools/rocpd_summary.py "$OUT"/pf/pf_results.db > "$OUT/pmc_fetch.txt" 2>&1
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pw" -o pw -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --profile-steps 0 --in-flight 1 > "$OUT/pw.log" 2>&1 || exit 4
python3 tools/rocpd_summary.py "$OUT"/pw/pw_results.db > "$OUT/pmc_write.txt" 2>&1
rm -rf "$OUT/kt" "$OUT/pf" "$OUT/pw"
echo done
