"""gr80211_rx_batch on the first n frames of a VHT MCS8 corpus, one batch at a time: call + wait, and the stages' times from
the library's own events.  python3 tools/rx_latency.py [n ...]"""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import torch, bench
pkg = importlib.import_module("fork-gr-ieee80211_amd")
dev = torch.device("cuda:0")
ctx = pkg.RxContext(0)
arenas, sync, navail, exp, idx, L, _ = bench.build_corpus_gpu(torch, ctx, pkg.txgen, "vht_mcs8_1500B_snr30", 4096, 0, dev)
desc = torch.from_numpy(ctx.make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
out = torch.zeros((4096, 1536), dtype=torch.uint8, device=dev); ol = torch.zeros(4096, dtype=torch.int32, device=dev); npd = torch.zeros(4096, dtype=torch.int32, device=dev)
c = pkg.RxContext(0)
for a in (sys.argv[1:] or "1 2 8 32 64 128 512 1000".split()):
    n, at = (int(x) for x in (a.split("@") + ["0"])[:2])   # "n@first": n frames from frame `first` of the corpus on
    ts = []
    for k in range(9):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        c.rx_batch_into(arenas[0], None, desc[at * 32: (at + n) * 32], n, out[:n], ol[:n], npd[:n]); c.sync()
        ts.append((time.perf_counter() - t0) * 1e3)
    c.set_profiling(True)
    c.rx_batch_into(arenas[0], None, desc[at * 32: (at + n) * 32], n, out[:n], ol[:n], npd[:n]); c.sync()
    tm = c.last_timing(); c.set_profiling(False)
    print(a, "median %.3f ms" % np.median(ts[2:]), {k: round(v, 3) for k, v in tm.items()}, int(npd[:n].sum()))
