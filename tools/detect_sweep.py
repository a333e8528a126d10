"""Randomised sweep of frame detection, GPU (gr80211_detect) against the CPU oracle of presiso -> trigger -> sync:
captures of frames of random format, length, SNR and carrier offset with random gaps between them.  The GPU sums the
autocorrelation windows from prefix sums, the reference (and the oracle) keeps running sums, so the two round
differently and a plateau that grazes the 0.3 threshold can come out differently; this counts how often.

    python tools/detect_sweep.py [--captures N] [--frames-per-capture M] [--seed S]

Prints one JSON line.
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
    ap.add_argument("--captures", type=int, default=8)
    ap.add_argument("--frames-per-capture", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from oracle import oracle as O

    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx = pkg.txgen
    ctx = pkg.RxContext(0)
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    tot = {"frames_sent": 0, "oracle_sync_bytes": 0, "gpu_sync_bytes": 0, "same_position": 0, "oracle_only": 0,
           "gpu_only": 0, "rad_beyond_2e-6": 0, "rssi_different": 0}
    for cap in range(args.captures):
        parts = [np.zeros(int(rng.integers(200, 900)), np.complex128)]
        for _ in range(args.frames_per_capture):
            fmt, mcs = [(0, int(rng.integers(8))), (1, int(rng.integers(8))), (2, int(rng.integers(9)))][int(rng.integers(3))]
            sig, m = tx.gen_frame(fmt, mcs, 1, tx.make_mpdu(int(rng.integers(20, 400)), rng))
            sig = sig[0] * 12.0
            cfo = float(rng.uniform(-150e3, 150e3))
            sig = sig * np.exp(1j * 2 * np.pi * cfo / 20e6 * np.arange(len(sig)))
            # per-frame SNR: noise is added per frame so that the level differs from frame to frame
            snr = float(rng.uniform(4.0, 35.0))
            gap = np.zeros(int(rng.integers(250, 1500)), np.complex128)
            seg = np.concatenate([sig, gap])
            p = np.mean(np.abs(sig) ** 2) / 10 ** (snr / 10)
            seg = seg + np.sqrt(p / 2) * (rng.standard_normal(len(seg)) + 1j * rng.standard_normal(len(seg)))
            parts.append(seg)
            tot["frames_sent"] += 1
        x = np.concatenate(parts).astype(np.complex64)
        ref = O.detect(x)
        d, n = ctx.detect(torch.from_numpy(x).cuda())
        ctx.sync()
        got = d.cpu().numpy()[: n * 32].view(pkg.rx.FRAME_DESC_DTYPE)
        rp = {int(p): k for k, p in enumerate(ref["pos"])}
        gp = {int(p): k for k, p in enumerate(got["sync_off"])}
        both = sorted(set(rp) & set(gp))
        tot["oracle_sync_bytes"] += len(rp)
        tot["gpu_sync_bytes"] += len(gp)
        tot["same_position"] += len(both)
        tot["oracle_only"] += len(set(rp) - set(gp))
        tot["gpu_only"] += len(set(gp) - set(rp))
        for p in both:
            if abs(float(got["rad"][gp[p]]) - float(ref["rad"][rp[p]])) > 2e-6:
                tot["rad_beyond_2e-6"] += 1
            if float(got["rssi"][gp[p]]) != float(ref["rssi"][rp[p]]):
                tot["rssi_different"] += 1
        print("capture %d: %d samples, oracle %d, gpu %d" % (cap, len(x), len(rp), len(gp)), file=sys.stderr, flush=True)
    tot["seconds"] = round(time.time() - t0, 1)
    tot["seed"] = args.seed
    print(json.dumps(tot))


if __name__ == "__main__":
    main()
