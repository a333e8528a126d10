#!/usr/bin/env python3
"""Micro-benchmark of the decode path (k_viterbi / k_traceback) on random soft bits.
usage: vit_bench.py [frames ...]   (GR80211_RX_LIB selects an alternative build for A/B runs)"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("fork-gr-ieee80211_amd")
ctx = pkg.RxContext(0)
T, NL, CR = 12168, 16224, 2
for n in [int(a) for a in sys.argv[1:]] or [100000]:
    g = torch.Generator(device="cuda")
    g.manual_seed(1)
    llr = torch.randn(n * NL, generator=g, device="cuda", dtype=torch.float32)
    desc = np.zeros(n, dtype=pkg.rx.DEC_DESC_DTYPE)
    desc["llr_off"] = np.arange(n, dtype=np.int64) * NL
    desc["n_llr"] = NL
    desc["format"] = 2
    desc["len"] = 1500
    desc["cr"] = CR
    desc["mcs"] = 8
    desc["ampdu"] = 1
    desc["trellis"] = T
    dd = torch.from_numpy(desc.view(np.uint8).reshape(-1)).cuda()
    ctx.set_profiling(True)
    acc = {}
    for it in range(4):
        res = ctx.decode_batch(llr, dd, n, out_stride=1536)
        if it:
            for k, v in ctx.last_timing().items():
                acc[k] = acc.get(k, 0.0) + v / 3
    chk = int(res["out"][:, :64].to(torch.int64).sum().item()) ^ int(res["out_len"].sum().item())
    print("lib=%s frames=%d viterbi=%.3f ms traceback=%.3f ms schedule=%.3f ms total=%.3f ms  (%.2f Mframes/s viterbi) chk=%d"
          % (os.path.basename(pkg.rx.LIB_PATH), n, acc["viterbi"], acc["traceback"], acc["schedule"], acc["total"],
             n / acc["viterbi"] / 1e3, chk), flush=True)
    del llr
