"""Two batches in flight: the benchmark workload decoded by two contexts on two streams (one host thread each),
against the same number of steps on one context.  While one batch's decoder drains (SIMDs that held a single
wave fall idle first) the other batch's preamble, demodulator and first decoder waves take their place.

    python tools/overlap_probe.py [--frames N] [--steps K] [--workload W]

Prints one JSON line: Msamples/s and ms per step for the two forms, and the bit-exact counts of the overlapped run.
"""
import argparse
import importlib
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--workload", default="vht_mcs8_1500B_snr30")
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--own-streams", action="store_true", help="every context on the stream it created itself")
    ap.add_argument("--single-thread", action="store_true",
                    help="one host thread keeps the contexts busy with gr80211_rx_submit / gr80211_rx_finish")
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench

    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctxs = [pkg.RxContext(0) for _ in range(args.depth)]
    n = args.frames if args.frames > 0 else bench.DEFAULT_FRAMES[args.workload]
    arenas, sync, navail, exp, idx, L, _ = bench.build_corpus_gpu(torch, ctxs[0], pkg.txgen, args.workload, n, 0, dev)
    arena, arena1 = arenas[0], (arenas[1] if len(arenas) > 1 else None)
    desc = torch.from_numpy(ctxs[0].make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
    blob_len = exp.shape[1]
    stride = ((blob_len + 4 + 63) // 64) * 64
    outs = [(torch.zeros((n, stride), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(args.depth)]
    streams = [torch.cuda.Stream() for _ in range(args.depth)]

    def run(k, steps):
        if args.own_streams:  # torch's current stream is the null stream here: the library takes the context's own
            for _ in range(steps):
                ctxs[k].rx_batch_into(arena, arena1, desc, n, *outs[k])
            ctxs[k].sync()
            return
        with torch.cuda.stream(streams[k]):
            for _ in range(steps):
                ctxs[k].rx_batch_into(arena, arena1, desc, n, *outs[k])

    for k in range(args.depth):  # warm-up: buffers of every context reach their size
        run(k, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(0, args.steps)
    torch.cuda.synchronize()
    seq = time.perf_counter() - t0

    per = args.steps // args.depth
    for o in outs:
        for t in o:
            t.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.single_thread:
        # submit every context's batch, then finish them in the same order; the next round's submits queue behind the
        # decoders on each context's own stream
        for _ in range(per):
            keep = [ctxs[k].rx_submit(arena, arena1, desc, n) for k in range(args.depth)]
            for k in range(args.depth):
                ctxs[k].rx_finish(*outs[k])
            del keep
        for c in ctxs:
            c.sync()
    else:
        threads = [threading.Thread(target=run, args=(k, per)) for k in range(args.depth)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    torch.cuda.synchronize()
    ovl = time.perf_counter() - t0
    ok = []
    for o, ol, npd in outs:
        rows = npd == 1
        match = (o[:, :blob_len] == exp[idx]).all(dim=1)
        ok.append({"fcs_ok": int(rows.sum().item()), "bit_exact_vs_tx": int((rows & match).sum().item()),
                   "fcs_ok_but_wrong": int((rows & ~match).sum().item())})
    print(json.dumps({
        "workload": args.workload, "frames": n, "depth": args.depth,
        "one_context": {"steps": args.steps, "ms_per_step": seq / args.steps * 1e3,
                        "Msamples_per_s": n * L * args.steps / seq / 1e6},
        "batches_in_flight": {"steps": per * args.depth, "ms_per_step": ovl / (per * args.depth) * 1e3,
                              "Msamples_per_s": n * L * per * args.depth / ovl / 1e6},
        "psdu": ok}))


if __name__ == "__main__":
    main()
