#!/usr/bin/env python3
"""A live receiver's view: a 20-Msamples/s antenna stream with a 1500-byte VHT MCS8 frame every 3 ms, handed over in
windows of 8192 samples at the pace of the air (one every 0.41 ms), and for every frame the time from the arrival of the
window that holds its last sample to its message.  Blocks: the fused rx block made without a batch size (rounds by the traffic) and with rounds of 4096
frames (the latter shows what a batch size costs a quiet channel), capture_rx with the shortest rounds it takes (438 016 samples), and signal
-> demod -> decode as three blocks called in turn, demod and decode made without a batch size."""
import gc
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
RATE, WIN = 20e6, 8192


def main():
    gc.disable()
    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx, stream = pkg.txgen, importlib.import_module("fork-gr-ieee80211_amd.stream")
    rng = np.random.default_rng(11)
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    gap = 60000
    parts, sync_at, last, pos = [], [], [], 0
    for _ in range(n_frames):
        sig, _ = tx.gen_frame(tx.F_VHT, 8, 1, tx.make_mpdu(1496, rng))
        sig = (sig[0] * 12.0).astype(np.complex64)
        parts += [np.zeros(gap, np.complex64), sig]
        sync_at.append(pos + gap + 176)
        pos += gap + len(sig)
        last.append(pos - 1)
    parts.append(np.zeros(4 * gap, np.complex64))
    iq = np.concatenate(parts)
    iq = (iq + 0.003 * (rng.standard_normal(len(iq)) + 1j * rng.standard_normal(len(iq)))).astype(np.complex64)
    sync = np.zeros(len(iq), np.uint8)
    tags = []
    for p in sync_at:
        sync[p] = 1
        tags += [(p, 0, "rad", 0.0), (p, 0, "snr", 30.0), (p, 0, "rssi", -40.0)]

    def paced(step):
        """Calls step(at, end) for every window when the air would have delivered it; returns each window's arrival time."""
        t0 = time.perf_counter()
        arrive = []
        for at in range(0, len(iq), WIN):
            end = min(at + WIN, len(iq))
            due = t0 + end / RATE
            while time.perf_counter() < due:
                pass
            arrive.append(time.perf_counter())
            step(at, end)
        return arrive

    def report(name, arrive, stamps):
        lat = []
        for k, t in enumerate(stamps[:n_frames]):
            lat.append((t - arrive[last[k] // WIN]) * 1e3)
        if len(stamps) < n_frames:
            print("%-58s %d of %d messages while the stream ran" % (name, len(stamps), n_frames))
        if lat:
            lat = np.array(lat)
            print("%-58s message after the last sample's window: median %.2f ms, min %.2f, max %.2f (%d frames)" %
                  (name, np.median(lat), lat.min(), lat.max(), len(lat)))

    def fused(kind, batch, name, with_sync=True, hold=None, min_round=None):
        with stream.StreamBlock(kind, batch) as b:
            stamps = []
            b._cb = stream.MSG_FN(lambda user, p, n: stamps.append(time.perf_counter()))
            b.lib.gr80211_block_set_msg_handler(b.h, b._cb, None)
            if hold:
                b.set_option(stream.OPT_MAX_HOLD, hold)
            if min_round:
                b.set_option(stream.OPT_MIN_ROUND, min_round)
            held = [0]

            def step(at, end):
                lo = held[0]
                ins = ([sync[lo:end]] if with_sync else []) + [iq[lo:end]]
                got = b.work(ins, [t for t in tags if lo <= t[0] < end] if with_sync else [])
                held[0] = lo + got
            arrive = paced(step)
            n_live = len(stamps)
            b.flush()
        if len(stamps) != n_frames:
            print("%-58s %d messages in all, %d frames sent" % (name, len(stamps), n_frames))
        report(name, arrive, stamps[:n_live])

    fused(stream.RX, 0, "rx block made without a batch size (rounds by the traffic)")
    fused(stream.RX, 4096, "rx block, rounds of 4096 frames")
    fused(stream.CAPTURE, 0, "capture_rx block, rounds of 438016 samples (22 ms)", with_sync=False, hold=438016)
    fused(stream.CAPTURE, 0, "capture_rx block, rounds from 16384 new samples on", with_sync=False, min_round=16384)

    # three blocks in turn on one thread: what signal gives goes to demod in the same pass, demod's soft bits to decode
    with stream.StreamBlock(stream.SIGNAL) as sg, stream.StreamBlock(stream.DEMOD) as dm, stream.StreamBlock(stream.DECODE) as dc:
        stamps = []
        dc._cb = stream.MSG_FN(lambda user, p, n: stamps.append(time.perf_counter()))
        dc.lib.gr80211_block_set_msg_handler(dc.h, dc._cb, None)
        state = dict(sg_at=0, items=np.zeros(0, np.complex64), it_tags=[], it_base=0, llr=np.zeros(0, np.float32), llr_tags=[], llr_base=0)

        def step(at, end):
            lo = state["sg_at"]
            n_before = sum(len(x) for x in sg.out[0])
            nt = len(sg.out_tags)
            got = sg.work([sync[lo:end], iq[lo:end]], [t for t in tags if lo <= t[0] < end], 1 << 16)
            state["sg_at"] = lo + got
            new = sg.out[0][-1] if sum(len(x) for x in sg.out[0]) > n_before else np.zeros(0, np.complex64)
            state["items"] = np.concatenate([state["items"], new])
            state["it_tags"] += [(t[0], 0, t[1], t[2], t[3]) for t in sg.out_tags[nt:]]
            if len(state["items"]):
                base = state["it_base"]
                nl, ntl = sum(len(x) for x in dm.out[0]), len(dm.out_tags)
                got = dm.work([state["items"]], state["it_tags"], 1 << 17)   # tag offsets count from the stream's start
                state["items"] = state["items"][got:]
                state["it_base"] = base + got
                state["it_tags"] = [t for t in state["it_tags"] if t[0] >= base + got]
                if sum(len(x) for x in dm.out[0]) > nl:
                    state["llr"] = np.concatenate([state["llr"], dm.out[0][-1]])
                state["llr_tags"] += [(t[0], 0, t[1], t[2], t[3]) for t in dm.out_tags[ntl:] if t[1] in stream.TAG_KEYS[:11]]
            if len(state["llr"]):
                base = state["llr_base"]
                got = dc.work([state["llr"]], state["llr_tags"], 0)
                state["llr"] = state["llr"][got:]
                state["llr_base"] = base + got
                state["llr_tags"] = [t for t in state["llr_tags"] if t[0] >= base + got]
        arrive = paced(step)
        n_live = len(stamps)
        report("signal -> demod -> decode made without batch sizes, in turn", arrive, stamps[:n_live])


if __name__ == "__main__":
    main()
