#!/bin/bash
# Where the rx block's host thread spends its time, and the trip of every round:
#   bash tools/stream_probe.sh <frames> <window> <batch> [pin: 1|0]
python3 - "$@" <<'PY'
import importlib, os, subprocess, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
pkg = importlib.import_module("fork-gr-ieee80211_amd")
n, chunk, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pins = sys.argv[4:] or ["1", "0"]
ctx = pkg.RxContext(0)
dev = torch.device("cuda:0")
arenas, sync, navail, exp, idx, L, _ = bench.build_corpus_gpu(torch, ctx, pkg.txgen, "vht_mcs8_1500B_snr30", n, 0, dev)
os.makedirs("/tmp/sp", exist_ok=True)
arenas[0][: n * L + 64].cpu().numpy().tofile("/tmp/sp/iq0.bin")
with open("/tmp/sp/trig.txt", "w") as fh:
    for p in sync.tolist():
        fh.write("%d 0.0 30.0 -40.0\n" % p)
ctx.close(); del arenas; torch.cuda.empty_cache()
env = dict(os.environ, GR80211_PUMP_QUIET="1", GR80211_BLOCK_PROFILE="1")
for pin in pins:
    env["GR80211_PUMP_PIN"] = pin
    p = subprocess.run(["fork-gr-ieee80211_amd/gr80211_pump", "rx", "1", str(chunk), str(batch), "/tmp/sp/trig.txt", "/tmp/sp/iq0.bin"],
                       capture_output=True, text=True, env=env)
    print("pin", pin, p.stdout.strip().replace("\n", " | "))
    print(p.stderr.strip())
PY
