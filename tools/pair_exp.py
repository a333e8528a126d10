#!/usr/bin/env python3
"""Experiment: 1024 long groups + 1024 short groups in one decode launch (one of each per SIMD if the
dispatcher places blocks in index order).  usage: pair_exp.py [short_fraction ...]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("fork-gr-ieee80211_amd")
ctx = pkg.RxContext(0)
T, NL, CR = 12168, 16224, 2
n = 131072
g = torch.Generator(device="cuda")
g.manual_seed(1)
llr = torch.randn(n * NL, generator=g, device="cuda", dtype=torch.float32)
for frac in [float(a) for a in sys.argv[1:]] or [1.0, 0.565, 0.5]:
    desc = np.zeros(n, dtype=pkg.rx.DEC_DESC_DTYPE)
    desc["llr_off"] = np.arange(n, dtype=np.int64) * NL
    desc["format"] = 2
    desc["len"] = 1500
    desc["cr"] = CR
    desc["mcs"] = 8
    desc["ampdu"] = 1
    ts = int(T * frac) // 6 * 6
    tr = np.full(n, T, dtype=np.int32)
    tr[n // 2:] = ts
    desc["trellis"] = tr
    desc["n_llr"] = np.where(tr == T, NL, (ts * 4 + 2) // 3)
    dd = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
    ctx.set_profiling(True)
    acc = 0.0
    for it in range(4):
        ctx.decode_batch(llr, dd, n, out_stride=1536)
        if it:
            acc += ctx.last_timing()["viterbi"] / 3
    print("short fraction %.3f (%d steps): decode %.3f ms" % (frac, ts, acc), flush=True)
