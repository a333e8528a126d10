#!/bin/bash
# Kernel trace of tools/overlap_probe.py: do kernels of the two streams run at the same time?
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/ovl && timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/ovl -- python3 $R/tools/overlap_probe.py --steps 4 --frames ${FRAMES:-50000} $PROBE_ARGS > /tmp/ovl.log 2>&1 || { tail -5 /tmp/ovl.log; exit 1; }
F=$(find /tmp/ovl -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
ev.sort()
t0 = ev[0][0]
big = [e for e in ev if e[1] - e[0] > 200000]
print("kernels > 0.2 ms:", len(big))
for e in big[-40:]:
    print("%10.3f %10.3f  %8.3f ms  q=%s s=%s %s" % ((e[0] - t0) / 1e6, (e[1] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[3], e[4], e[2]))
PY
