#!/usr/bin/env python3
"""Rate of the reference's three-block chain (signal -> demod -> decode as three blocks, one thread each) on a corpus of
VHT MCS8 frames: python3 tools/chain_rate.py [frames] [window items] [batch frames].  GR80211_BLOCK_PROFILE=1 prints
where each block's thread spent its time."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PUMP = os.path.join(ROOT, "fork-gr-ieee80211_amd", "gr80211_pump")


def main():
    import torch
    import bench
    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    windows = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "65536").split(",")]
    batches = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "4096").split(",")]
    work = "/tmp/chain_rate"
    os.makedirs(work, exist_ok=True)
    dev = torch.device("cuda:0")
    ctx = pkg.RxContext(0)
    arenas, sync, navail, exp, idx, L, _ = bench.build_corpus_gpu(torch, ctx, pkg.txgen, "vht_mcs8_1500B_snr30", n, 0, dev)
    iq = arenas[0][: n * L + 64]
    iq[n * L:] = 0
    f_iq, f_tr = os.path.join(work, "iq0.bin"), os.path.join(work, "trig.txt")
    iq.cpu().numpy().tofile(f_iq)
    with open(f_tr, "w") as fh:
        for p in sync.tolist():
            fh.write("%d 0.0 30.0 -40.0\n" % p)
    ctx.close()
    del arenas, iq
    torch.cuda.empty_cache()
    for pin in ("1", "0"):
        for w in windows:
            for b in batches:
                env = dict(os.environ, GR80211_PUMP_QUIET="1", GR80211_PUMP_PIN=pin)
                p = subprocess.run([PUMP, "chaint", "1", str(w), str(b), f_tr, f_iq], capture_output=True, text=True, env=env, timeout=600)
                tm = [l for l in p.stdout.splitlines() if l.startswith("TIME")]
                st = [l for l in p.stdout.splitlines() if l.startswith("STAT")]
                if p.returncode != 0 or not tm:
                    print("failed", p.returncode, p.stderr[-800:], flush=True)
                    continue
                sec = float(tm[0].split("seconds=")[1])
                print("chain, 3 threads  window %7d batch %5d  %s  %s  %.3f s  %.1f Msamples/s" %
                      (w, b, "page-locked" if pin == "1" else "ordinary   ", st[0], sec, n * L / sec / 1e6), flush=True)
                if os.environ.get("GR80211_BLOCK_PROFILE"):
                    print(p.stderr.strip(), flush=True)
    os.remove(f_iq)
    os.remove(f_tr)


if __name__ == "__main__":
    main()
