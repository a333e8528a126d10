#!/usr/bin/env python3
"""What a live receiver waits for: demod and decode blocks made without a batch size (they finish a frame in the call
that completes it, as the reference's blocks do), one 1500-byte VHT MCS8 frame per general_work call.  Prints the
median and the extremes of the call's duration -- soft bits (messages) are there when it returns."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import gc
    gc.disable()  # a collection in the middle of a call is not the block's time
    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx, stream = pkg.txgen, importlib.import_module("fork-gr-ieee80211_amd.stream")
    rng = np.random.default_rng(5)
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    parts, sync_at, pos = [], [], 0
    for _ in range(n_frames):
        sig, _ = tx.gen_frame(tx.F_VHT, 8, 1, tx.make_mpdu(1496, rng))
        sig = sig[0] * 12.0
        p = np.mean(np.abs(sig) ** 2)
        sig = sig + np.sqrt(p / 1000.0 / 2) * (rng.standard_normal(len(sig)) + 1j * rng.standard_normal(len(sig)))
        parts.append(np.concatenate([np.zeros(400), sig, np.zeros(400)]).astype(np.complex64))
        sync_at.append(pos + 400 + 176)
        pos += len(parts[-1])
    iq = np.concatenate(parts)
    sync = np.zeros(len(iq), np.uint8)
    tags = []
    for p in sync_at:
        sync[p] = 1
        tags += [(p, 0, "rad", 0.0), (p, 0, "snr", 30.0), (p, 0, "rssi", -40.0)]
    with stream.StreamBlock(stream.SIGNAL, 8) as sg:
        sg.run([sync, iq], tags, 65536, 1 << 20)
        items, sig_tags = sg.output(0), [(t[0], 0, t[1], t[2], t[3]) for t in sg.out_tags]
    nsamp = [int(t[3]) for t in sig_tags if t[2] == "nsamp"]
    assert len(nsamp) == n_frames, (len(nsamp), n_frames)

    def play(b, data, tg, bounds, nout):
        at, ts = 0, []
        for end in bounds:
            t0 = time.perf_counter()
            got = b.work([data[at:end]], [t for t in tg if at <= t[0] < end], nout)
            ts.append((time.perf_counter() - t0) * 1e3)
            assert got == end - at, (got, end - at)
            at = end
        return np.array(ts[2:])   # the first calls make buffers

    ends = list(np.cumsum(np.array(nsamp) + 320) - 320)
    ends[-1] = len(items)
    with stream.StreamBlock(stream.DEMOD) as dm:
        ts = play(dm, items, sig_tags, ends, 32768)
        llr, llr_tags = dm.output(0), [(t[0], 0, t[1], t[2], t[3]) for t in dm.out_tags if t[1] in stream.TAG_KEYS[:11]]
    print("demod  block, one frame (%d samples) per call: median %.3f ms, min %.3f, max %.3f (call %d); %d soft bits a frame" %
          (nsamp[0], np.median(ts), ts.min(), ts.max(), 2 + int(ts.argmax()), len(llr) // n_frames))
    firsts = [t[0] for t in llr_tags if t[2] == "total"]
    totals = [int(t[3]) for t in llr_tags if t[2] == "total"]
    with stream.StreamBlock(stream.DECODE) as dc:
        ts = play(dc, llr, llr_tags, [f + t for f, t in zip(firsts, totals)], 0)
        n_msgs = len(dc.msgs)
    print("decode block, one packet (%d soft bits) per call: median %.3f ms, min %.3f, max %.3f; %d of %d packets published by "
          "the calls themselves" % (totals[0], np.median(ts), ts.min(), ts.max(), n_msgs, n_frames))


if __name__ == "__main__":
    main()
