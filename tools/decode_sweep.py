"""Randomised sweep of the decode path (gr80211_decode_batch) against the CPU oracle: frames of random code rate,
format and length (20 to 4095 bytes), clean, noisy, with small-integer soft bits (ties), erasures or pure noise, decoded
by both forms of the decoder (one wave and four waves per group); decoded bits, packets and FCS verdicts bit for bit.

    python tools/decode_sweep.py [--frames N] [--seed S]

Prints one JSON line; exit code 1 on a difference (the comparison raises).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    from test_gpu_decode import coded_frame, run_and_compare
    from oracle import oracle as O

    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx = pkg.txgen
    rng = np.random.default_rng(args.seed)
    frames = []
    kinds = {"clean": 0, "noisy": 0, "ties": 0, "erasures": 0, "noise_only": 0}
    for k in range(args.frames):
        fmt, cr = int(rng.integers(3)), int(rng.integers(4))
        nbytes = int(rng.integers(20, 600)) if rng.random() < 0.9 else int(rng.integers(600, 4000 if fmt == 2 else 4095))
        kind = ("clean", "noisy", "ties", "erasures", "noise_only")[int(rng.integers(5))]
        kinds[kind] += 1
        if kind == "clean":
            f = coded_frame(tx, rng, fmt, cr, nbytes, mcs=k % 8, noise=0.3)
        elif kind == "noisy":
            f = coded_frame(tx, rng, fmt, cr, nbytes, mcs=k % 8, snr_scale=2.0, noise=1.3)
        elif kind == "ties":
            f = coded_frame(tx, rng, fmt, cr, nbytes, snr_scale=1.0, noise=1.5, integer=True)
        elif kind == "erasures":
            f = coded_frame(tx, rng, fmt, cr, nbytes)
            f["llr"][rng.random(len(f["llr"])) < 0.25] = 0.0
        else:
            f = coded_frame(tx, rng, fmt, cr, nbytes)
            f["llr"] = rng.standard_normal(len(f["llr"])).astype(np.float32)
        frames.append(f)
    t0 = time.time()
    ok = {}
    for waves in (1, 2, 4, 10, 64):  # 10: one wave per group in the form of batches beyond 1024 groups; 64: one wave per frame
        os.environ["GR80211_GROUP_WAVES"] = str(waves)
        ctx = pkg.RxContext(0)
        ok[str(waves)] = run_and_compare(pkg, ctx, O, frames)
        ctx.close()
    print(json.dumps({"frames": args.frames, "kinds": kinds, "fcs_ok_by_group_waves": ok, "differences": 0,
                      "seconds": round(time.time() - t0, 1), "seed": args.seed}))


if __name__ == "__main__":
    main()
