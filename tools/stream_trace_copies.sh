#!/bin/bash
# Kernel and copy trace of the rx block with every upload going by the runtime's copy engines: how many bytes arrive
# beside a decoder's forward pass and how many otherwise.   bash tools/stream_trace_copies.sh <frames> <window> <batch>
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
export GR80211_PUMP_QUIET=1 GR80211_UPLOAD=dma
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/strace -- $R/fork-gr-ieee80211_amd/gr80211_pump rx 1 $2 $3 /tmp/sp/trig.txt /tmp/sp/iq0.bin > /tmp/strace.log 2>&1 || { tail -5 /tmp/strace.log; exit 1; }
K=$(find /tmp/strace -name '*kernel_trace.csv' | head -1)
M=$(find /tmp/strace -name '*memory_copy_trace.csv' | head -1)
python3 - "$K" "$M" <<'PY'
import csv, re, sys
def short(name):
    m = re.search(r"k_\w+", name)
    return m.group(0) if m else name[:24]
ker = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1])))
rows = list(csv.DictReader(open(sys.argv[2])))
print("copy trace columns:", list(rows[0].keys()))
cp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "?"), int(r.get("Bytes", r.get("Size", 0)) or 0)) for r in rows)
h2d = [c for c in cp if "HOST_TO_DEVICE" in c[2].upper() or c[2].upper().startswith("H")]
dec = [k for k in ker if k[2] == "k_viterbi_split"]
print("%d kernels, %d copies of which %d host-to-device, %d forward passes" % (len(ker), len(cp), len(h2d), len(dec)))
if not dec or not h2d: sys.exit(0)
lo, hi = dec[1][0], dec[-1][1]
def overlap(a0, a1, b0, b1): return max(0, min(a1, b1) - max(a0, b0))
inside_t = sum(d[1] - d[0] for d in dec[1:])
bytes_in = bytes_out = busy_in = busy_out = 0.0
for c in h2d:
    if c[1] <= lo or c[0] >= hi: continue
    dur = max(1, c[1] - c[0])
    ov = sum(overlap(c[0], c[1], d[0], d[1]) for d in dec[1:])
    bytes_in += c[3] * ov / dur
    bytes_out += c[3] * (dur - ov) / dur
    busy_in += ov
    busy_out += dur - ov
print("beside a forward pass: %.1f GB/s over %.1f ms (a copy running %.0f %% of it); otherwise: %.1f GB/s over %.1f ms (%.0f %%)" %
      (bytes_in / inside_t, inside_t / 1e6, 100 * busy_in / inside_t, bytes_out / (hi - lo - inside_t), (hi - lo - inside_t) / 1e6,
       100 * busy_out / (hi - lo - inside_t)))
d = dec[len(dec) // 2]
print("copies around the forward pass at %.3f ms (%.3f ms long):" % ((d[0] - lo) / 1e6, (d[1] - d[0]) / 1e6))
for c in h2d:
    if c[1] < d[0] - 500000 or c[0] > d[1] + 500000: continue
    print("  %+9.1f us  %8.1f us  %9d bytes  %.1f GB/s" % ((c[0] - d[0]) / 1e3, (c[1] - c[0]) / 1e3, c[3], c[3] / max(1, c[1] - c[0])))
PY
