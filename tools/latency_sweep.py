#!/usr/bin/env python3
"""Latency of small batches: gr80211_decode_batch (soft bits resident, host descriptors, nothing waited for inside the
call) and gr80211_rx_batch for n frames of the headline kind (VHT MCS8, 1500-byte A-MPDU, 12 168 trellis steps), by
decoder form (GR80211_GROUP_WAVES: 64 = one wavefront per frame, 4 / 2 / 1 = wavefronts per group of 64 frames).
Prints one line per (n, form): median and minimum of the call + synchronisation in ms, and checks every PDU."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1,64,256,1000,2000,4000,8000,16000,32000")
    ap.add_argument("--forms", default="64,4,2,1")
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--rx", action="store_true", help="time gr80211_rx_batch (IQ in) as well")
    args = ap.parse_args()
    import torch
    from test_gpu_decode import coded_frame
    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx = pkg.txgen
    rng = np.random.default_rng(3)
    base = coded_frame(tx, rng, 2, 2, 1496, mcs=8, snr_scale=3.0, noise=0.0)
    L = len(base["llr"])
    want = bytes([2, 1496 % 256, 1496 // 256]) + base["mpdu"] + bytes([8])
    sizes = [int(x) for x in args.sizes.split(",")]
    nmax = max(sizes)
    gen = torch.Generator(device="cuda").manual_seed(1)
    llr = torch.from_numpy(base["llr"]).cuda()[None, :] + 0.8 * torch.randn((nmax, L), generator=gen, device="cuda")
    llr = llr.contiguous().view(-1)
    wt = torch.frombuffer(bytearray(want), dtype=torch.uint8).cuda()
    ctxs = {}
    for f in args.forms.split(","):
        os.environ["GR80211_GROUP_WAVES"] = f
        ctxs[f] = pkg.RxContext(0)
    os.environ.pop("GR80211_GROUP_WAVES", None)
    print("# decode_batch: n frames x %d soft bits, T = %d; ms = call + stream synchronisation" % (L, base["trellis"]))
    print("%8s %6s %10s %10s %8s" % ("frames", "form", "median_ms", "min_ms", "fcs_ok"))
    for n in sizes:
        desc = np.zeros(n, dtype=pkg.rx.DEC_DESC_DTYPE)
        desc["llr_off"] = np.arange(n, dtype=np.int64) * L
        desc["n_llr"], desc["format"], desc["len"], desc["cr"], desc["mcs"], desc["ampdu"], desc["trellis"] = \
            L, 2, base["len"], 2, 8, 1, base["trellis"]
        out = torch.zeros((n, 1536), dtype=torch.uint8, device="cuda")
        out_len = torch.zeros(n, dtype=torch.int32, device="cuda")
        n_pdu = torch.zeros(n, dtype=torch.int32, device="cuda")
        for f, c in ctxs.items():
            ts = []
            for r in range(args.reps + 2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rc = c.lib.gr80211_decode_batch(c.h, llr.data_ptr(), n * L, desc.ctypes.data, n, 0, out.data_ptr(), 1536,
                                                out_len.data_ptr(), n_pdu.data_ptr(), None, 0, None)
                assert rc == 0, c.lib.gr80211_last_error(c.h)
                c.lib.gr80211_sync(c.h, None)
                ts.append((time.perf_counter() - t0) * 1e3)
            ts = ts[2:]
            ok = int(n_pdu.sum().item())
            good = bool((out[:, :len(want)][n_pdu == 1] == wt).all().item())
            assert os.environ.get('LS_PROBE') or (good and ok >= 0.98 * n), (n, f, ok)
            c.set_profiling(True)
            rc = c.lib.gr80211_decode_batch(c.h, llr.data_ptr(), n * L, desc.ctypes.data, n, 0, out.data_ptr(), 1536,
                                            out_len.data_ptr(), n_pdu.data_ptr(), None, 0, None)
            tm = c.last_timing()
            c.set_profiling(False)
            print("%8d %6s %10.3f %10.3f %8d   kernels: records %.3f sched %.3f decode %.3f traceback %.3f" %
                  (n, f, float(np.median(ts)), min(ts), ok, tm["preamble"], tm["schedule"], tm["viterbi"], tm["traceback"]), flush=True)
    for c in ctxs.values():
        c.close()


if __name__ == "__main__":
    main()
