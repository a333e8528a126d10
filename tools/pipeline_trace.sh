#!/bin/bash
# Kernel trace of the benchmark's timed steps (two contexts in rotation): when each step's kernels start and end.
#   bash tools/pipeline_trace.sh [bench.py options]
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ptrace
cd "$R"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptrace -- python3 bench.py --steps 8 --warmup 2 --cpu-frames 0 --in-flight 1 --profile-steps 0 "$@" > /tmp/ptrace.log 2>&1 || { tail -5 /tmp/ptrace.log; exit 1; }
F=$(find /tmp/ptrace -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, re, sys
def short(name):
    m = re.search(r"k_\w+", name)
    return m.group(0) if m else None
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", "?")) for r in rows if short(r["Kernel_Name"]))
dec = [e for e in ev if e[2].startswith("k_decode")]
# the timed steps follow the warm-up steps (one per context at least); the one-batch-at-a-time leg comes last
t0 = dec[2][0] if len(dec) > 12 else dec[0][0]
t1 = dec[10][1] if len(dec) > 12 else dec[-1][1]
print("the timed steps (ms since the start of the third decoder launch):")
for e in ev:
    if e[0] < t0 - 3_000_000 or e[0] > t1 or e[2] in ("k_zero_u32", "k_sched_hist", "k_sched_scan", "k_sched_assign", "k_sched_groups", "k_clear_outputs", "k_ndp_emit"):
        continue
    print("  stream %s %-12s %8.2f .. %8.2f  (%6.2f ms)" % (e[3], e[2], (e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6))
PY
