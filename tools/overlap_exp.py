#!/usr/bin/env python3
"""Experiment: does the front end (preamble + schedule + demod) of the next batch fit under the decode kernel of
the current one?  Context A runs whole batches (rx_batch) on one stream, context B runs front ends only
(demod_batch) on another; compares back-to-back pairs with both running serially."""
import ctypes as C
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

pkg = importlib.import_module("fork-gr-ieee80211_amd")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
arenas, sync, navail, exp, idx, L = bench.build_corpus(torch, pkg.txgen, "vht_mcs8_1500B_snr30", n, 64, 0, dev)
arena = arenas[0]
A, B = pkg.RxContext(0), pkg.RxContext(0)
desc = A.to_device(A.make_desc(sync, navail))
stride = ((exp.shape[1] + 4 + 63) // 64) * 64
out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
out_len = torch.zeros(n, dtype=torch.int32, device=dev)
n_pdu = torch.zeros(n, dtype=torch.int32, device=dev)
OWN = os.environ.get("OWN_STREAMS", "0") == "1"   # use the contexts' own streams instead of torch's
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
if OWN:
    A._stream = lambda: None
    B._stream = lambda: None
torch.cuda.synchronize()


def whole():
    with torch.cuda.stream(sA):
        A.rx_batch_into(arena, None, desc, n, out, out_len, n_pdu)


def front():
    with torch.cuda.stream(sB):
        p, nfl = C.c_void_p(), C.c_uint64(0)
        rc = B.lib.gr80211_demod_batch(B.h, C.c_void_p(arena.data_ptr()), None, arena.numel(), C.c_void_p(desc.data_ptr()),
                                       n, 1, None, C.byref(p), C.byref(nfl), B._stream())
        assert rc == 0


def timed(fn, k=4):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


def both():
    whole()
    front()


def both_serial():
    whole()
    torch.cuda.synchronize()
    front()
    torch.cuda.synchronize()


print("whole batch      %.3f ms" % timed(whole), flush=True)
print("front end only   %.3f ms" % timed(front), flush=True)
print("serial pair      %.3f ms" % timed(both_serial), flush=True)
print("overlapped pair  %.3f ms" % timed(both), flush=True)
print("fcs_ok", int(n_pdu.sum().item()), flush=True)
