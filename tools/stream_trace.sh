#!/bin/bash
# Kernel trace of the rx block driven by gr80211_pump: which queue every kernel ran on and whether the uploads of one
# round run beside the decoder of the round before.   bash tools/stream_trace.sh <frames> <window> <batch>
R=$(pwd)
python3 - "$@" <<'PY' || exit 1
import importlib, os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
pkg = importlib.import_module("fork-gr-ieee80211_amd")
n = int(sys.argv[1])
ctx = pkg.RxContext(0)
arenas, sync, navail, exp, idx, L, _ = bench.build_corpus_gpu(torch, ctx, pkg.txgen, "vht_mcs8_1500B_snr30", n, 0, torch.device("cuda:0"))
os.makedirs("/tmp/sp", exist_ok=True)
arenas[0][: n * L + 64].cpu().numpy().tofile("/tmp/sp/iq0.bin")
with open("/tmp/sp/trig.txt", "w") as fh:
    for p in sync.tolist():
        fh.write("%d 0.0 30.0 -40.0\n" % p)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/strace
GR80211_PUMP_QUIET=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/strace -- $R/fork-gr-ieee80211_amd/gr80211_pump rx 1 $2 $3 /tmp/sp/trig.txt /tmp/sp/iq0.bin > /tmp/strace.log 2>&1 || { tail -5 /tmp/strace.log; exit 1; }
F=$(find /tmp/strace -name '*kernel_trace.csv' | head -1)
python3 - "$F" <<'PY'
import csv, re, sys, collections, statistics
def short(name):
    m = re.search(r"k_\w+", name)
    return m.group(0) if m else name[:24]
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?"), r.get("Stream_Id", "?")) for r in rows)
t0 = ev[0][0]
print("queues used:", collections.Counter((e[3], e[4]) for e in ev))
# for every decoder kernel: pull kernels that STARTED while it ran
dec = [e for e in ev if "viterbi" in e[2] or "k_decode" in e[2]]
pull = [e for e in ev if "k_pull" in e[2]]
print("kernels:", collections.Counter(e[2] for e in ev).most_common(12))
for d in dec[2:10]:
    inside = [p for p in pull if d[0] <= p[0] <= d[1]]
    print("%-22s q=%s s=%s %9.3f .. %9.3f ms: %d pull kernels started meanwhile (queues %s)" %
          (d[2], d[3], d[4], (d[0] - t0) / 1e6, (d[1] - t0) / 1e6, len(inside), sorted(set((p[3], p[4]) for p in inside))))
# one decoder interval in detail: everything that ran from 0.3 ms before it to 0.3 ms after it
if len(dec) > 6:
    d = dec[6]
    print("around the decoder kernel at %.3f ms (%.3f ms long, stream %s):" % ((d[0] - t0) / 1e6, (d[1] - d[0]) / 1e6, d[4]))
    last = None
    for e in ev:
        if e[1] < d[0] - 300000 or e[0] > d[1] + 300000: continue
        line = "  %+9.1f us  %8.1f us  s=%s %s" % ((e[0] - d[0]) / 1e3, (e[1] - e[0]) / 1e3, e[4], e[2])
        if e[2] == "k_pull" and last == "k_pull" and (e[1] - e[0]) < 30000:
            skipped += 1
            continue
        if last == "k_pull" and skipped: print("      ... %d more short k_pull" % skipped)
        skipped = 0
        last = e[2]
        print(line)
# pull kernels beside a decoder's forward pass and on their own: how long they take, and how much of the time one runs
def busy(lo, hi):
    return sum(max(0, min(p[1], hi) - max(p[0], lo)) for p in pull)
t_dec = sum(d[1] - d[0] for d in dec)
b_in = sum(busy(d[0], d[1]) for d in dec)
span = ev[-1][1] - dec[0][0] if dec else 1
b_all = busy(dec[0][0], ev[-1][1]) if dec else 0
ins = [p for p in pull if any(d[0] <= p[0] <= d[1] for d in dec)]
outs = [p for p in pull if dec and p[0] >= dec[0][0] and not any(d[0] <= p[0] <= d[1] for d in dec)]
md = lambda v: statistics.median([(p[1] - p[0]) / 1e3 for p in v]) if v else 0.0
import statistics
print("beside a forward pass: %d pull kernels, median %.1f us, one running %.0f %% of the time; otherwise: %d, median %.1f us, %.0f %%" %
      (len(ins), md(ins), 100.0 * b_in / max(1, t_dec), len(outs), md(outs), 100.0 * (b_all - b_in) / max(1, span - t_dec)))
gaps = [(pull[i + 1][0] - pull[i][1]) / 1e3 for i in range(len(pull) - 1)]
dur = [(p[1] - p[0]) / 1e3 for p in pull]
import statistics
print("pull kernels: %d, median duration %.1f us, median gap to the next %.1f us, gaps > 1 ms: %d" %
      (len(pull), statistics.median(dur), statistics.median(gaps), sum(1 for g in gaps if g > 1000)))
PY
