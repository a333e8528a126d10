#!/usr/bin/env python3
"""Experiment: one batch split over S contexts (one HIP stream each) issued back to back.
usage: stream_exp.py S [S ...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

pkg = importlib.import_module("fork-gr-ieee80211_amd")
dev = torch.device("cuda", 0)
n = 100000
arenas, sync, navail, exp, idx, L = bench.build_corpus(torch, pkg.txgen, "vht_mcs8_1500B_snr30", n, 64, 0, dev)
arena = arenas[0]
ctx0 = pkg.RxContext(0)
desc_np = ctx0.make_desc(sync, navail)
blob_len = exp.shape[1]
stride = ((blob_len + 4 + 63) // 64) * 64
out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
out_len = torch.zeros(n, dtype=torch.int32, device=dev)
n_pdu = torch.zeros(n, dtype=torch.int32, device=dev)
for S in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
    ctxs = [pkg.RxContext(0) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]  # the binding launches on torch's current stream
    bounds = [(n * i // S // 64 * 64 if i < S else n) for i in range(S + 1)]
    bounds[S] = n
    descs = [torch.from_numpy(desc_np[bounds[i]:bounds[i + 1]].view(np.uint8).reshape(-1).copy()).to(dev) for i in range(S)]

    def step():
        for i in range(S):
            s, e = bounds[i], bounds[i + 1]
            with torch.cuda.stream(streams[i]):
                ctxs[i].rx_batch_into(arena, None, descs[i], e - s, out[s:e], out_len[s:e], n_pdu[s:e])

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    ok = int(n_pdu.sum().item())
    print("S=%d  %.3f ms/step  %.1f Msamp/s  fcs_ok=%d" % (S, dt * 1e3, n * L / dt / 1e6, ok), flush=True)
    for c in ctxs:
        c.close()
