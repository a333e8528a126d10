#!/usr/bin/env python3
"""Packet delivery ratio against SNR, closed loop on one GPU: gr80211_tx_batch -> AWGN -> gr80211_rx_batch.
The kind of sweep the reference runs through GNU Radio with tools/performance/perf_siso.py, here one batch per
point.  usage: pdr_sweep.py [frames per point] [mpdu bytes]"""
import importlib
import os
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

pkg = importlib.import_module("fork-gr-ieee80211_amd")
tx = pkg.txgen
dev = torch.device("cuda", 0)
ctx = pkg.RxContext(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
nbytes = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rng = np.random.default_rng(1)
SNRS = list(range(2, 34, 2))
FMT = {0: "L", 1: "HT", 2: "VHT"}
print("frames per point %d, MPDU %d bytes, AWGN, one antenna; cells = %% of frames whose FCS passed" % (n, nbytes))
print("%-9s" % "snr dB" + "".join("%6d" % s for s in SNRS))
t_start = time.perf_counter()
total = 0
for fmt, mcss in ((0, range(8)), (1, range(8)), (2, range(9))):
    for mcs in mcss:
        body = rng.integers(0, 256, (n, nbytes - 4), dtype=np.uint8)
        mpdu = np.empty((n, nbytes), np.uint8)
        mpdu[:, :-4] = body
        mpdu[:, -4:] = np.fromiter((zlib.crc32(r) for r in body), np.uint32, n).astype("<u4").view(np.uint8).reshape(n, 4)
        if fmt == 2:
            psdu = np.tile(np.frombuffer(tx.ampdu_vht([bytes(nbytes)]), np.uint8), (n, 1))
            psdu[:, 4:4 + nbytes] = mpdu
        else:
            psdu = mpdu
        plen = psdu.shape[1]
        L = ctx.tx_samples(fmt, mcs, 1, plen)
        stride = L + 160
        d = ctx.make_tx_desc(n)
        d["out_off"] = np.arange(n, dtype=np.uint64) * stride + 80
        d["psdu_off"] = np.arange(n, dtype=np.uint64) * plen
        d["psdu_len"], d["format"], d["mcs"], d["scale"] = plen, fmt, mcs, 12.0
        clean = torch.zeros(n * stride + 64, dtype=torch.complex64, device=dev)
        ctx.tx_batch(torch.from_numpy(psdu.reshape(-1)).to(dev), d, n, clean)
        desc = ctx.make_desc(np.arange(n, dtype=np.uint64) * stride + 80 + 176, np.full(n, L - 176 + 80, np.uint32))
        exp = torch.from_numpy(mpdu).to(dev)
        row = []
        for snr in SNRS:
            sigma = 12.0 / 64.0 * 10 ** (-snr / 20.0) / np.sqrt(2.0)
            noisy = clean + sigma * torch.view_as_complex(torch.randn((clean.numel(), 2), device=dev))
            res = ctx.rx_batch(noisy, None, desc, n, out_stride=((nbytes + 8 + 63) // 64) * 64, want_info=False)
            ok = (res["n_pdu"] == 1) & (res["out"][:, 3:3 + nbytes] == exp).all(dim=1)
            assert int(((res["n_pdu"] == 1) & ~ok).sum().item()) == 0  # a passed FCS with wrong bytes would be a bug
            row.append(100.0 * int(ok.sum().item()) / n)
            total += n
        print("%-9s" % ("%s mcs%d" % (FMT[fmt], mcs)) + "".join("%6.1f" % v for v in row), flush=True)
dt = time.perf_counter() - t_start
print("%d frames transmitted, impaired, received and checked in %.1f s" % (total, dt))
