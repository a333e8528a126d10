"""Rate of the streaming host blocks on host-resident streams: a corpus of VHT MCS8 frames is made with the GPU
transmitter and written to files, then the command-line scheduler (gr80211_pump) plays it through the `rx` block
(sync bytes + tags, as `sync` hands them to `signal`), through `capture_rx` (antenna stream only) and, as soft
bits, through `decode`, in scheduler-sized windows.  The figures include everything a flowgraph pays: window
copies, host-to-device transfers, the trigger search, messages; reading the files is outside the timed part.
    python3 tools/stream_rate.py [frames] [workdir]"""
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PUMP = os.path.join(ROOT, "fork-gr-ieee80211_amd", "gr80211_pump")


def run(args, n_samples, label, pin=True):
    env = dict(os.environ, GR80211_PUMP_QUIET="1", GR80211_PUMP_PIN="1" if pin else "0")
    label += "  page-locked buffers" if pin else "  ordinary buffers   "
    p = subprocess.run([PUMP] + [str(a) for a in args], capture_output=True, text=True, env=env, timeout=300)
    stat = [l for l in p.stdout.splitlines() if l.startswith("STAT")]
    tm = [l for l in p.stdout.splitlines() if l.startswith("TIME")]
    if p.returncode != 0 or not tm:
        print(label, "failed", p.returncode, p.stderr[-500:], flush=True)
        return
    sec = float(tm[0].split("seconds=")[1])
    if os.environ.get("GR80211_BLOCK_PROFILE"):  # where the block's host thread spent its time
        print(p.stderr.strip(), flush=True)
    print("%-34s %s  %.3f s  %.1f Msamples/s" % (label, stat[0], sec, n_samples / sec / 1e6), flush=True)


def main():
    import torch
    import bench
    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/stream_rate"
    os.makedirs(work, exist_ok=True)
    dev = torch.device("cuda:0")
    ctx = pkg.RxContext(0)
    arenas, sync, navail, exp, idx, L, _ = bench.build_corpus_gpu(torch, ctx, pkg.txgen, "vht_mcs8_1500B_snr30", n, 0, dev)
    iq_t = arenas[0][: n * L + 64]
    iq_t[n * L:] = 0
    iq = iq_t.cpu().numpy()
    f_iq = os.path.join(work, "iq0.bin")
    iq.tofile(f_iq)
    f_tr = os.path.join(work, "trig.txt")
    with open(f_tr, "w") as fh:
        for p in sync.tolist():
            fh.write("%d 0.0 30.0 -40.0\n" % p)
    # the soft bits `demod` would hand to `decode`, with the tags of each packet's first item (of the first 60000
    # frames at most: a soft bit takes twice the bytes of the sample it came from)
    n_all, n = n, min(n, 60000)
    info, llr = ctx.demod_batch(iq_t[: n * L + 64], None, ctx.make_desc(sync[:n], navail[:n]), n)
    f_llr = os.path.join(work, "llr.bin")
    np.asarray(llr).astype(np.float32).tofile(f_llr)
    f_tg = os.path.join(work, "tags.txt")
    with open(f_tg, "w") as fh:
        for i in range(n):
            fh.write("%d %d %d %d %d %d %d %d\n" % (int(info["llr_off"][i]), int(info["format"][i]), int(info["len"][i]),
                                                  int(info["total"][i]), int(info["cr"][i]), int(info["mcs"][i]),
                                                  int(info["ampdu"][i]), int(info["trellis"][i])))
    ctx.close()
    del arenas, iq_t
    torch.cuda.empty_cache()
    ns_dec = n * L
    n = n_all
    ns = n * L
    print("%d VHT MCS8 frames, %d samples per antenna, %.0f MB of IQ; %d of them as soft bits, %.0f MB" %
          (n, ns, ns * 8 / 1e6, ns_dec // L, os.path.getsize(f_llr) / 1e6), flush=True)
    only = os.environ.get("GR80211_STREAM_ONLY", "")  # "chain": the three-block rows alone
    for pin in (True, False):
        for chunk, batch in ((8192, 4096), (65536, 4096), (65536, 16384), (1 << 18, 16384), (1 << 20, 16384)):
            if not only or only == "rx":
                batch = int(os.environ.get("GR80211_RX_BATCH", batch))  # frames per round (experiments; 0: made without a batch size)
                run(["rx", 1, chunk, batch, f_tr, f_iq], ns, "rx      window %7d batch %5d" % (chunk, batch), pin)
    if only == "capture":
        run(["capture", 1, 65536, f_iq], ns, "capture window   65536")
        run(["capture", 1, 8192, f_iq], ns, "capture window    8192")
    if only in ("rx", "capture"):
        for f in (f_iq, f_tr, f_llr, f_tg):
            os.remove(f)
        return
    # the reference's three blocks as three blocks: one thread each (a flowgraph), and all on one thread
    n_chain = min(n, int(os.environ.get("GR80211_CHAIN_FRAMES", "200000")))  # frames the three-block rows play
    f_iqc = f_iq
    if n_chain < n:
        f_iqc = os.path.join(work, "iq0_chain.bin")
        iq[: n_chain * L + 64].tofile(f_iqc)
    f_trc = os.path.join(work, "trig_chain.txt")
    with open(f_trc, "w") as fh:
        for p in sync[:n_chain].tolist():
            fh.write("%d 0.0 30.0 -40.0\n" % p)
    for pin in (True, False):
        for chunk, batch in ((65536, 4096), (1 << 18, 4096), (1 << 20, 8192)):
            batch = int(os.environ.get("GR80211_CHAIN_BATCH", batch))  # frames per round (experiments)
            run(["chaint", 1, chunk, batch, f_trc, f_iqc], n_chain * L, "chain, 3 threads window %7d batch %5d" % (chunk, batch), pin)
    for f in (f_trc,) + ((f_iqc,) if f_iqc != f_iq else ()):
        os.remove(f)
    # the same blocks called in turn by one thread, on the first 60 000 frames (it keeps whole streams between the blocks)
    n_one = min(n, 60000)
    f_iq1, f_tr1 = os.path.join(work, "iq0_one.bin"), os.path.join(work, "trig_one.txt")
    iq[: n_one * L + 64].tofile(f_iq1)
    with open(f_tr1, "w") as fh:
        for p in sync[:n_one].tolist():
            fh.write("%d 0.0 30.0 -40.0\n" % p)
    run(["chain", 1, 65536, 4096, f_tr1, f_iq1], n_one * L, "chain, 1 thread  window   65536 batch  4096")
    for f in (f_iq1, f_tr1):
        os.remove(f)
    if not only:
        run(["capture", 1, 65536, f_iq], ns, "capture window   65536")
        for pin in (True, False):
            for chunk, batch in ((65536, 4096), (65536, 16384), (1 << 18, 16384), (1 << 20, 16384)):
                run(["decode", chunk, batch, f_tg, f_llr], ns_dec, "decode  window %7d batch %5d" % (chunk, batch), pin)
    for f in (f_iq, f_tr, f_llr, f_tg):
        os.remove(f)


if __name__ == "__main__":
    main()
