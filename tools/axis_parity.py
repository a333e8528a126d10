#!/usr/bin/env python3
"""The axis-value soft-bit path (GR80211_AXIS_LLR=1) against the default path, modulation by modulation: batches of
frames of one modulation and many lengths, made by the GPU transmitter with noise added, received by a context of each
kind; PDU bytes, lengths, counts and tags must be the same bytes.  Both contexts use the one-wave decoder form."""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
H22 = ((1.0, 0.5j), (0.4, -1.0))


def batch(torch, ctx, tx, fmt, mcs, nss, n, rng, snr_db):
    lens = rng.integers(30, 1500, n)
    lens[: n // 4] = 1496  # a run of equal frames, so that whole tiles lie inside every frame of a group
    psdus = []
    for nb in lens:
        body = rng.integers(0, 256, int(nb) - 4, dtype=np.uint8).tobytes()
        body += zlib.crc32(body).to_bytes(4, "little")
        psdus.append(np.frombuffer(tx.ampdu_vht([body]) if fmt == 2 else body, dtype=np.uint8))
    plen = np.array([len(p) for p in psdus], dtype=np.int64)
    psdu_off = np.concatenate([[0], np.cumsum(plen)[:-1]])
    nsamp = np.array([ctx.tx_samples(fmt, mcs, nss, int(p)) for p in plen], dtype=np.int64)
    gap = 64
    out_off = np.concatenate([[0], np.cumsum(nsamp + gap)[:-1]]) + gap
    total = int(out_off[-1] + nsamp[-1] + gap)
    d = ctx.make_tx_desc(n)
    d["out_off"], d["psdu_off"], d["psdu_len"] = out_off, psdu_off, plen
    d["format"], d["mcs"], d["nss"], d["scale"] = fmt, mcs, nss, 12.0 * np.sqrt(nss)
    arenas = [torch.zeros(total + 64, dtype=torch.complex64, device="cuda") for _ in range(nss)]
    ns = ctx.tx_batch(torch.from_numpy(np.concatenate(psdus)).cuda(), torch.from_numpy(d.view(np.uint8).reshape(-1)).cuda(),
                      n, arenas[0], arenas[1] if nss == 2 else None)
    assert int((ns > 0).sum().item()) == n
    if nss == 2:
        a, b = arenas
        arenas = [H22[0][0] * a + H22[0][1] * b, H22[1][0] * a + H22[1][1] * b]
    sigma = float(12.0 / 64.0 * 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0))
    gen = torch.Generator(device="cuda")
    gen.manual_seed(int(rng.integers(1 << 30)))
    for x in arenas:
        v = torch.view_as_real(x)
        v += torch.randn(v.shape, generator=gen, device="cuda", dtype=torch.float32) * sigma
    return arenas, (out_off + 176).astype(np.uint64), (nsamp - 176 + gap).astype(np.uint32)


def compare(torch, pkg, mods, snr0s=(12.0, 5.0), n=300, seed=5):
    """Number of (noise level, modulation) batches on which the two paths differ."""
    tx = pkg.txgen
    os.environ["GR80211_GROUP_WAVES"] = "1"
    os.environ["GR80211_AXIS_LLR"] = "0"
    plain = pkg.rx.RxContext(0)
    os.environ["GR80211_AXIS_LLR"] = "1"
    axis = pkg.rx.RxContext(0)
    rng = np.random.default_rng(seed)
    bad = 0
    for snr0, (fmt, mcs, nss) in [(s, m) for s in snr0s for m in mods]:
        # noise for which some frames of the denser modulations fail their check: wrong bytes must be the same wrong bytes
        arenas, sync, navail = batch(torch, plain, tx, fmt, mcs, nss, n, rng, snr_db=snr0 + 2.2 * (mcs % 8 if fmt == 1 else mcs))
        desc = plain.make_desc(sync, navail)
        res = []
        for c in (plain, axis):
            r = c.rx_batch(arenas[0], arenas[1] if nss == 2 else None, desc, n)
            c.sync()
            res.append({k: r[k].cpu().numpy().copy() for k in ("out", "out_len", "n_pdu", "info")})
        same = all(np.array_equal(res[0][k], res[1][k]) for k in ("out_len", "n_pdu"))
        info0 = res[0]["info"].view(pkg.rx.FRAME_INFO_DTYPE)
        ok_frames = int((res[0]["n_pdu"] > 0).sum())
        cols = np.arange(res[0]["out"].shape[1])[None, :] < res[0]["out_len"][:, None]
        rows = np.array_equal(res[0]["out"] * cols, res[1]["out"] * cols)
        wrong = int(((info0["status"] == 0) & (res[0]["n_pdu"] == 0)).sum())
        okf = info0["status"] == 0
        foreign = int((okf & ((info0["format"] != fmt) | (info0["mcs"] != mcs) | (info0["nSS"] != nss))).sum())
        if foreign:
            # header errors turned frames into other modulations: their groups are mixed, which the experiment leaves out
            print(f"snr0 {snr0:4.1f} format {fmt} mcs {mcs:2d} nss {nss}: {foreign} frames taken for another modulation, "
                  f"batch not comparable", flush=True)
            continue
        print(f"snr0 {snr0:4.1f} format {fmt} mcs {mcs:2d} nss {nss}: frames {n}, with PDUs {ok_frames}, failed their check {wrong}, "
              f"lengths/counts equal {same}, bytes equal {rows}", flush=True)
        bad += (not same) or (not rows)
    return bad


def main():
    import importlib
    import torch
    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    mods = [(0, m, 1) for m in range(8)] + [(1, m, 1) for m in range(8)] + [(1, m, 2) for m in range(8, 16)] + \
           [(2, m, 1) for m in range(9)] + [(2, m, 2) for m in range(9)]
    bad = compare(torch, pkg, mods)
    print("AXIS PARITY", "OK" if bad == 0 else f"FAILED for {bad} modulations")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
