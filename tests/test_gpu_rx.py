"""Whole RX path on the GPU (gr80211_rx_batch / gr80211_demod_batch) against the CPU oracle and the
golden fixtures: PSDU bytes and FCS verdicts bit-exact, tags equal, soft bits within 1e-4 relative."""
import numpy as np
import pytest

from util import H22, LLR_ATOL, LLR_RTOL, MALFORMED, awgn, build_arena, golden_frames, ref_fixture_mpdu, ref_fixtures

pytestmark = pytest.mark.gpu

TAGS = ("status", "lsig_mcs", "lsig_len", "nsamp", "format", "mcs", "len", "cr", "ampdu", "trellis", "total",
        "nSym", "nSS")


def oracle_tags(f):
    return dict(status=f.status, lsig_mcs=f.lsig_mcs, lsig_len=f.lsig_len, nsamp=f.nsamp, format=f.m.format,
                mcs=f.m.mcs, len=f.m.len, cr=f.m.cr, ampdu=f.m.ampdu, trellis=f.trellis, total=f.total,
                nSym=f.m.nSym, nSS=f.m.nSS)


def run_rx(pkg, rxctx, oracle, frames, rads=None, check_llr=True, two_ant=None, mu_pos=None, strict=False):
    """frames: list of ([antenna arrays], sync_idx).  Compares GPU against oracle frame by frame."""
    import torch
    iq0, iq1, sync, navail = build_arena(frames)
    if two_ant is False:
        iq1 = None
    n = len(frames)
    rads = np.zeros(n, np.float32) if rads is None else np.asarray(rads, np.float32)
    mu_pos = np.zeros(n, np.uint32) if mu_pos is None else np.asarray(mu_pos, np.uint32)
    desc = rxctx.make_desc(sync, navail, rad=rads, mu_pos=mu_pos)
    d0 = torch.from_numpy(iq0).cuda()
    d1 = torch.from_numpy(iq1).cuda() if iq1 is not None else None
    res = rxctx.rx_batch(d0, d1, desc, n)
    rxctx.sync()
    info = res["info"].cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)
    out_len = res["out_len"].cpu().numpy()
    out = res["out"].cpu().numpy()
    n_pdu = res["n_pdu"].cpu().numpy()
    if check_llr:
        info2, llr = rxctx.demod_batch(d0, d1, desc, n)
    results = []
    for i in range(n):
        # the oracle would see the rest of the arena too; restrict it like the descriptor does
        ref = oracle.rx_frame(iq0[: int(sync[i]) + int(navail[i])], None if iq1 is None else iq1[: int(sync[i]) + int(navail[i])],
                              int(sync[i]), float(rads[i]), mu_pos=int(mu_pos[i]), strict=strict)
        st = ref["status"]
        assert int(info["status"][i]) == st, (i, int(info["status"][i]), st)
        if st == 0:
            want = oracle_tags(ref["frame"])
            got = {k: int(info[k][i]) for k in TAGS}
            assert got == want, (i, got, want)
            if ref["frame"].m.format == 2:
                # per-stream SNR from the re-encoded VHT-SIG-B.  On noise-free input the "noise" is float
                # rounding (> 100 dB), where only the order of magnitude is meaningful.
                a, b = float(info["sssnr0"][i]), ref["frame"].sssnr0
                assert abs(a - b) <= 1e-3 * max(1.0, abs(b)) if b < 80.0 else a > 80.0, (i, a, b)
        assert int(n_pdu[i]) == ref["n_pdu"], (i, int(n_pdu[i]), ref["n_pdu"])
        assert bytes(out[i, : out_len[i]]) == ref["blob"], i
        if check_llr and st == 0:
            o = int(info2["llr_off"][i])
            g = llr[o: o + ref["frame"].total]
            r = ref["llr"]
            err = np.abs(g - r)
            tol = LLR_RTOL * np.abs(r) + LLR_ATOL
            assert np.all(err <= tol), (i, float(err.max()), int(np.argmax(err - tol)))
        results.append(ref)
    return results


def test_reference_fixtures(pkg, rxctx, oracle):
    exp = ref_fixture_mpdu()
    fx = ref_fixtures()
    r = run_rx(pkg, rxctx, oracle, [(fx[0][1], fx[0][2])])
    assert r[0]["pdus"][0]["mpdu"] == exp
    r = run_rx(pkg, rxctx, oracle, [(fx[1][1], fx[1][2])])
    assert r[0]["pdus"][0]["mpdu"] == exp


def test_golden_frames_siso(pkg, rxctx, oracle):
    gf = [f for f in golden_frames() if f["nss"] == 1]
    frames = [([f["iq"][0]], f["sync_idx"]) for f in gf]
    r = run_rx(pkg, rxctx, oracle, frames, rads=[f["rad"] for f in gf])
    for f, ref in zip(gf, r):
        assert [p["mpdu"] for p in ref["pdus"]] == [f["mpdu"]], f["name"]


def test_golden_frames_2x2(pkg, rxctx, oracle):
    gf = [f for f in golden_frames() if f["nss"] == 2]
    frames = [([f["iq"][0], f["iq"][1]], f["sync_idx"]) for f in gf]
    r = run_rx(pkg, rxctx, oracle, frames)
    for f, ref in zip(gf, r):
        assert [p["mpdu"] for p in ref["pdus"]] == [f["mpdu"]], f["name"]


def test_soft_bits_against_reference_generator(pkg, rxctx):
    """The GPU demodulator's soft bits on the noise-free golden frames against the stream worked out from the
    constellation points the reference's own transmitter held for them (tests/golden/soft_*.npz, max-log formula and
    interleaver written out in tests/util.py): values to 1e-4 relative + the floor, signs = the generator's coded
    bits.  No oracle in the loop."""
    import torch
    from util import expected_llr_from_generator, golden_soft
    for two in (False, True):
        gf = [f for f in golden_frames() if (f["nss"] == 2) == two]
        frames = [([f["iq"][a] for a in range(f["iq"].shape[0])], f["sync_idx"]) for f in gf]
        iq0, iq1, sync, navail = build_arena(frames)
        desc = rxctx.make_desc(sync, navail, rad=np.array([f["rad"] for f in gf], np.float32))
        d0 = torch.from_numpy(iq0).cuda()
        d1 = torch.from_numpy(iq1).cuda() if two else None
        info, llr = rxctx.demod_batch(d0, d1, desc, len(gf))
        for i, f in enumerate(gf):
            soft = golden_soft(f["name"])
            want = expected_llr_from_generator(soft, legacy=f["fmt"] == 0)
            assert int(info["status"][i]) == 0 and int(info["total"][i]) == len(want), f["name"]
            got = llr[int(info["llr_off"][i]): int(info["llr_off"][i]) + len(want)]
            assert np.array_equal(got > 0, soft["coded"] == 1), f["name"]
            err = np.abs(got - want)
            assert np.all(err <= LLR_RTOL * np.abs(want) + LLR_ATOL), (f["name"], float(err.max()))


def make_noisy(tx, rng, fmt, mcs, nss, nbytes, snr_db, cfo_hz=0.0):
    mpdu = tx.make_mpdu(nbytes, rng)
    sig, m = tx.gen_frame(fmt, mcs, nss, mpdu)
    sig = sig * 12.0 * np.sqrt(m.nss)
    if cfo_hz:
        sig = sig * np.exp(1j * 2 * np.pi * cfo_hz / 20e6 * np.arange(sig.shape[1]))
    rx = H22 @ sig if m.nss == 2 else sig
    rx = awgn(rx, snr_db, rng)
    pad = np.zeros((rx.shape[0], 30))
    rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
    return ([rx[a] for a in range(rx.shape[0])], 30 + 176), mpdu


def test_every_mcs_noisy_siso(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(21)
    frames, mpdus = [], []
    for fmt, nm in ((0, 8), (1, 8), (2, 9)):
        for mcs in range(nm):
            for rep in range(2):
                fr, mp = make_noisy(tx, rng, fmt, mcs, 1, int(rng.integers(30, 500)), 33.0)
                frames.append(fr)
                mpdus.append(mp)
    r = run_rx(pkg, rxctx, oracle, frames)
    ok = sum(1 for ref, mp in zip(r, mpdus) if [p["mpdu"] for p in ref["pdus"]] == [mp])
    assert ok == len(frames)


def test_every_mcs_noisy_2x2(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(22)
    frames, mpdus = [], []
    for fmt, mcss in ((1, range(8, 16)), (2, range(0, 9))):
        for mcs in mcss:
            fr, mp = make_noisy(tx, rng, fmt, mcs, 2, int(rng.integers(40, 600)), 36.0)
            frames.append(fr)
            mpdus.append(mp)
    # single-stream frames seen by the two-antenna path use antenna 1 only
    for fmt, mcs in ((0, 3), (1, 4), (2, 6)):
        fr, mp = make_noisy(tx, rng, fmt, mcs, 1, 200, 33.0)
        fr = ([fr[0][0], np.zeros_like(fr[0][0])], fr[1])
        frames.append(fr)
        mpdus.append(mp)
    r = run_rx(pkg, rxctx, oracle, frames)
    ok = sum(1 for ref, mp in zip(r, mpdus) if [p["mpdu"] for p in ref["pdus"]] == [mp])
    assert ok == len(frames)


def test_cfo_derotation(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(23)
    frames, rads, mpdus = [], [], []
    for fmt, mcs, cfo in ((0, 2, 120e3), (1, 5, -90e3), (2, 7, 232e3), (2, 8, -40e3)):
        fr, mp = make_noisy(tx, rng, fmt, mcs, 1, 300, 35.0, cfo_hz=cfo)
        frames.append(fr)
        rads.append(-2 * np.pi * cfo / 20e6)
        mpdus.append(mp)
    r = run_rx(pkg, rxctx, oracle, frames, rads=rads)
    for ref, mp in zip(r, mpdus):
        assert [p["mpdu"] for p in ref["pdus"]] == [mp]


def test_low_snr_same_verdicts(pkg, rxctx, oracle):
    """At marginal SNR some frames fail the FCS: the GPU must fail exactly the same ones."""
    tx = pkg.txgen
    rng = np.random.default_rng(24)
    frames = []
    for k in range(48):
        fmt, mcs = ((0, 6), (1, 7), (2, 8))[k % 3]
        fr, _ = make_noisy(tx, rng, fmt, mcs, 1, 250, 21.0 + (k % 4))
        frames.append(fr)
    r = run_rx(pkg, rxctx, oracle, frames, check_llr=False)
    n_ok = sum(ref["n_pdu"] for ref in r)
    assert 0 < n_ok < len(frames)


def test_garbage_and_truncated_frames_dropped(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(25)
    noise = (0.1 * (rng.standard_normal(3000) + 1j * rng.standard_normal(3000))).astype(np.complex64)
    fr, _ = make_noisy(tx, rng, 2, 4, 1, 400, 35.0)
    cut = ([fr[0][0][:1200]], fr[1])          # data symbols missing: zeros are read, FCS fails
    short = ([fr[0][0][:300]], fr[1])         # fewer than 224 samples after the trigger
    good, mp = make_noisy(tx, rng, 1, 3, 1, 90, 35.0)
    r = run_rx(pkg, rxctx, oracle, [([noise], 400), cut, short, good], check_llr=False)
    assert r[3]["pdus"][0]["mpdu"] == mp
    assert r[0]["n_pdu"] == 0 and r[1]["n_pdu"] == 0 and r[2]["status"] == 1


def test_symbols_without_samples_are_erasures(pkg, rxctx, oracle):
    """Data symbols whose samples are exactly zero -- a capture that ends inside the frame, silence written into the
    middle of one -- make the reference's arithmetic produce 0 * 0 / 0 for every soft bit of the symbol.  Both the
    GPU demodulator and the oracle hand such soft bits on as 0.0 (erasures), so everything stays comparable: soft
    bits, decoded bits, PDUs."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(26)
    frames = []
    for fmt, mcs, nss, nbytes in ((2, 4, 1, 400), (0, 5, 1, 300), (1, 12, 2, 500), (2, 8, 2, 900), (1, 7, 1, 1200)):
        fr, _ = make_noisy(tx, rng, fmt, mcs, nss, nbytes, 35.0)
        ants, sidx = fr
        n = len(ants[0])
        hole = [a.copy() for a in ants]
        for a in hole:
            a[n // 2: n // 2 + 400] = 0      # five whole symbols of silence in the middle
        frames.append((hole, sidx))
        frames.append(([a[: n - 30 - 200] for a in ants], sidx))   # the capture ends 200 samples before the frame does
        frames.append(([a[: n - 30 - 60] for a in ants], sidx))    # only part of the last symbol is missing
    for two in (False, True):
        sel = [f for f in frames if (len(f[0]) == 2) == two]
        r = run_rx(pkg, rxctx, oracle, sel, check_llr=True)
        assert all(ref["status"] == 0 for ref in r)
        assert all(np.all(np.isfinite(ref["llr"])) for ref in r)
        assert any(np.count_nonzero(ref["llr"] == 0.0) >= ref["frame"].m.nCBPS for ref in r)
        # decoded bits as well: the oracle's decoder on its soft bits against the GPU decoder on the same stream
        for ref in r:
            f = ref["frame"]
            desc = np.zeros(1, dtype=pkg.rx.DEC_DESC_DTYPE)
            desc[0] = (0, f.total, f.m.format, f.m.len, f.m.cr, f.m.mcs, f.m.ampdu, f.trellis, 0)
            res = rxctx.decode_batch(torch.from_numpy(ref["llr"]).cuda(), desc, 1, want_bits=True)
            rxctx.sync()
            want = oracle.decode(ref["llr"], f.m.format, f.m.len, f.m.cr, f.m.mcs, f.m.ampdu, f.trellis, want_bits=True)
            got = np.unpackbits(res["bits"][0].cpu().numpy(), bitorder="little")[: f.trellis]
            assert np.array_equal(got, want["bits"])


def test_decoder_given_non_finite_soft_bits(pkg, rxctx, oracle):
    """Soft bits that are NaN or infinite can only come from a caller of gr80211_decode_batch (the demodulator never
    emits them).  The reference's compare `a > b` and this decoder's max / sign-of-difference treat a NaN metric
    differently, so the bits of such a frame are not defined; what is: nothing faults, the frame publishes nothing
    it should not (a PDU only ever leaves with a good FCS), and the other frames of the batch -- the same decoder
    group, lane = frame -- are decoded exactly as without it."""
    import zlib
    import torch
    from test_gpu_decode import coded_frame
    tx = pkg.txgen
    rng = np.random.default_rng(27)
    good = [coded_frame(tx, rng, fmt, cr, int(rng.integers(60, 700))) for fmt in (0, 2) for cr in range(4) for _ in range(4)]
    bad = []
    for k, f in enumerate(good[:12]):
        g = dict(f)
        llr = f["llr"].copy()
        llr[rng.integers(0, len(llr), 40)] = (np.nan, np.inf, -np.inf)[k % 3]
        if k % 4 == 0:
            llr[:] = np.nan
        g["llr"] = llr
        bad.append(g)
    frames = [x for pair in zip(good, bad + good[12:]) for x in pair]
    is_bad = [False, True] * 12 + [False] * (len(frames) - 24)
    desc = np.zeros(len(frames), dtype=pkg.rx.DEC_DESC_DTYPE)
    off, parts = 0, []
    for i, f in enumerate(frames):
        desc[i] = (off, len(f["llr"]), f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], 0)
        parts.append(f["llr"])
        off += len(f["llr"])
    res = rxctx.decode_batch(torch.from_numpy(np.concatenate(parts)).cuda(), desc, len(frames), want_bits=True)
    rxctx.sync()
    out, out_len, n_pdu = res["out"].cpu().numpy(), res["out_len"].cpu().numpy(), res["n_pdu"].cpu().numpy()
    bits = res["bits"].cpu().numpy()
    for i, f in enumerate(frames):
        if is_bad[i]:
            for p in pkg.rx.split_pdus(bytes(out[i, : out_len[i]])):
                body = p["mpdu"]
                assert zlib.crc32(body[:-4]) == int.from_bytes(body[-4:], "little")
            continue
        want = oracle.decode(f["llr"], f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], want_bits=True)
        assert np.array_equal(np.unpackbits(bits[i], bitorder="little")[: f["trellis"]], want["bits"]), i
        assert int(n_pdu[i]) == want["n_pdu"] == 1 and bytes(out[i, : out_len[i]]) == want["blob"], i


def test_short_guard_interval_is_a_stride_only(pkg, rxctx, oracle):
    """Short-GI frames: the reference steps 72 samples per data symbol but keeps its 8-sample window offset and
    the long-GI channel estimate (lib/cloud80211phy.cc:898-902, 1155-1159), so it does not recover the payload.
    The GPU follows it sample for sample: same tags, same soft bits, same (failed) FCS verdicts."""
    tx = pkg.txgen
    rng = np.random.default_rng(31)
    frames = []
    for fmt, mcs, nss in ((1, 3, 1), (2, 5, 1), (1, 11, 2), (2, 2, 2)):
        sig, m = tx.gen_frame(fmt, mcs, nss, tx.make_mpdu(220, rng), short_gi=True)
        sig = sig * 12.0 * np.sqrt(m.nss)
        rx = awgn(H22 @ sig if m.nss == 2 else sig, 35.0, rng)
        pad = np.zeros((rx.shape[0], 30))
        rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
        frames.append(([rx[a] for a in range(rx.shape[0])], 206))
    r = run_rx(pkg, rxctx, oracle, [f for f in frames if len(f[0]) == 1])
    r += run_rx(pkg, rxctx, oracle, [f for f in frames if len(f[0]) == 2])
    assert all(ref["status"] == 0 and ref["frame"].m.nSymSamp == 72 for ref in r)


def test_vht_mcs9_with_truncated_ndbps(pkg, rxctx, oracle):
    """VHT MCS9 at 20 MHz: not a combination of the standard for one or two streams, but the reference receiver
    takes it with N_DBPS = 346 (693 for two streams) from an integer division (lib/cloud80211phy.cc:1287-1303)
    and decodes the coded bits as one continuous stream.  Same here: the payload comes back."""
    tx = pkg.txgen
    rng = np.random.default_rng(32)
    for nss in (1, 2):
        frames, mpdus = [], []
        for nb in (60, 300, 1500, 2999):
            mp = tx.make_mpdu(nb, rng)
            sig, m = tx.gen_frame(tx.F_VHT, 9, nss, mp)
            sig = sig * 12.0 * np.sqrt(m.nss)
            rx = awgn(H22 @ sig if m.nss == 2 else sig, 38.0, rng)
            pad = np.zeros((rx.shape[0], 30))
            rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
            frames.append(([rx[a] for a in range(rx.shape[0])], 206))
            mpdus.append(mp)
        r = run_rx(pkg, rxctx, oracle, frames)
        for ref, mp in zip(r, mpdus):
            assert ref["frame"].m.nDBPS == 346 * nss + (nss - 1) and [p["mpdu"] for p in ref["pdus"]] == [mp]


def test_malformed_sig_fields(pkg, rxctx, oracle):
    """Frames whose SIG fields the reference rejects, misreads or accepts although the standard would not: the
    same verdict, tags, soft bits and output for each (run_rx compares them all)."""
    tx = pkg.txgen
    rng = np.random.default_rng(33)
    frames = []
    for fmt, mcs, edit in MALFORMED:
        sig, m = tx.gen_frame(fmt, mcs, 1, tx.make_mpdu(94, rng), sig_edit=edit)
        rx = np.concatenate([np.zeros(30), awgn(sig * 12.0, 35.0, rng)[0], np.zeros(400)]).astype(np.complex64)
        frames.append(([rx], 206))
    r = run_rx(pkg, rxctx, oracle, frames, check_llr=False)
    statuses = [ref["status"] for ref in r]
    assert sum(1 for s in statuses if s != 0) >= 8, statuses
    assert sum(ref["n_pdu"] for ref in r) >= 3, statuses     # the fields nobody looks at do no harm
    # the same frames on the two-antenna path
    run_rx(pkg, rxctx, oracle, [([f[0][0], 0.5 * f[0][0]], f[1]) for f in frames], check_llr=False)


def test_vht_multi_subframe_ampdu(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(26)
    ms = [tx.make_mpdu(96, rng) for _ in range(3)]
    sig, m = tx.gen_frame(tx.F_VHT, 6, 1, None, n_mpdu_list=ms)
    rx = np.concatenate([np.zeros(30), sig[0] * 12.0, np.zeros(30)]).astype(np.complex64)
    r = run_rx(pkg, rxctx, oracle, [([rx], 206)])
    assert [p["mpdu"] for p in r[0]["pdus"]] == ms


def test_full_size_batch_roundtrip(pkg, rxctx):
    """Benchmark frame shape (VHT MCS8, 1500-byte A-MPDU, 30 dB): a few hundred frames in one batch,
    every PSDU must come back bit-exact (encode -> channel -> decode round trip)."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(27)
    uniq = []
    for k in range(6):
        mp = tx.make_mpdu(1496, rng)
        sig, m = tx.gen_frame(tx.F_VHT, 8, 1, mp)
        uniq.append((sig[0] * 12.0, mp))
    n = 200
    L = 3920
    arena = np.zeros(n * L + 64, dtype=np.complex64)
    for i in range(n):
        arena[i * L:(i + 1) * L] = awgn(uniq[i % 6][0], 30.0, rng)
    sync = (np.arange(n) * L + 176).astype(np.uint64)
    navail = np.full(n, L - 176, dtype=np.uint32)
    res = rxctx.rx_batch(torch.from_numpy(arena).cuda(), None, rxctx.make_desc(sync, navail), n)
    rxctx.sync()
    n_pdu = res["n_pdu"].cpu().numpy()
    assert int(n_pdu.sum()) == n
    for i in (0, 1, 63, 64, 127, 199):
        got = rxctx.pdus_of(res, i)
        assert got[0]["mpdu"] == uniq[i % 6][1]


def test_vht_sounding_ndp(pkg, rxctx, oracle):
    """VHT NDP with two space-time streams on one antenna: no data field, the path publishes the raw samples of
    the two VHT-LTF windows as a channel message (lib/demod_impl.cc:393-409,238-271, lib/decode_impl.cc:100-121).
    Seen by the two-antenna path the same frame fails demod2's length check and is dropped."""
    import torch
    from util import golden_ndp
    g = golden_ndp()
    rng = np.random.default_rng(31)
    vht = [f for f in golden_frames() if f["name"] == "vht_mcs4_200"][0]
    noisy = awgn(g["iq"], 30.0, rng).astype(np.complex64)
    frames = [([g["iq"][0]], g["sync_idx"]), ([vht["iq"][0]], vht["sync_idx"]), ([noisy[0]], g["sync_idx"])]
    r = run_rx(pkg, rxctx, oracle, frames, check_llr=False)
    s = g["sync_idx"]
    for k, src in ((0, g["iq"][0]), (2, noisy[0])):
        assert r[k]["status"] == pkg.rx.ST_NAMES_INV["ndp"] and r[k]["n_pdu"] == 1
        want = bytes([20, 0, 4]) + src[s + 472:s + 536].tobytes() + src[s + 552:s + 616].tobytes()
        assert r[k]["blob"] == want
    assert [p["mpdu"] for p in r[1]["pdus"]] == [vht["mpdu"]]
    # two antennas: dropped, like the reference's demod2
    r2 = run_rx(pkg, rxctx, oracle, [([g["iq"][0], g["iq"][1]], s)], check_llr=False)
    assert r2[0]["status"] == 2 and r2[0]["n_pdu"] == 0

    # with a frequency offset the message holds derotated samples: float work, compared to tolerance
    cfo = 150e3
    rot = (g["iq"][0] * np.exp(1j * 2 * np.pi * cfo / 20e6 * np.arange(g["iq"].shape[1]))).astype(np.complex64)
    rad = np.float32(-2 * np.pi * cfo / 20e6)
    iq0, _, sync, navail = build_arena([([rot], s)])
    res = rxctx.rx_batch(torch.from_numpy(iq0).cuda(), None, rxctx.make_desc(sync, navail, rad=np.array([rad], np.float32)), 1)
    rxctx.sync()
    ref = oracle.rx_frame(iq0[: int(sync[0]) + int(navail[0])], None, int(sync[0]), float(rad))
    got = bytes(res["out"][0, : int(res["out_len"][0].item())].cpu().numpy())
    assert ref["status"] == 5 and len(got) == 1027 and got[:3] == ref["blob"][:3]
    a = np.frombuffer(got[3:], np.float32)
    b = np.frombuffer(ref["blob"][3:], np.float32)
    assert np.all(np.abs(a - b) <= 1e-5 * np.abs(b).max())


def test_vht_mu_mimo_two_users(pkg, rxctx, oracle):
    """Two-user VHT MU-MIMO frame from the reference's Python generator (group id 2, zero-forcing precoder for the
    fixed 2x2 channel): each station's antenna, with its group position, yields its own A-MPDU
    (lib/demod_impl.cc:347-379, lib/cloud80211phy.cc:1170-1192).  Two antennas: demod2 has no MU branch -> dropped."""
    from util import golden_mu
    g = golden_mu()
    rng = np.random.default_rng(41)
    noisy = awgn(g["iq"], 32.0, rng).astype(np.complex64)
    s = g["sync_idx"]
    frames = [([g["iq"][0]], s), ([g["iq"][1]], s), ([noisy[0]], s), ([noisy[1]], s)]
    r = run_rx(pkg, rxctx, oracle, frames, mu_pos=[0, 1, 0, 1])
    for k, ref in enumerate(r):
        assert ref["status"] == 0 and ref["frame"].sumu == 1 and ref["frame"].m.mcs == g["mcs"][k % 2]
        assert [p["mpdu"] for p in ref["pdus"]] == [g["mpdus"][k % 2]]
    r2 = run_rx(pkg, rxctx, oracle, [([g["iq"][0], g["iq"][1]], s)], check_llr=False)
    assert r2[0]["status"] == 3 and r2[0]["n_pdu"] == 0


def test_descriptors_beyond_the_arena_read_nothing(pkg, rxctx):
    """Descriptors on the device are not validated by the host: one that points past the arena, or claims more
    samples than the arena holds, must be cut to the arena by every kernel (no out-of-bounds read), and the good
    frame beside them still decodes."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(91)
    fr, mp = make_noisy(tx, rng, 2, 5, 1, 300, 35.0)
    sig = fr[0][0]
    arena = torch.from_numpy(sig).cuda()
    n = len(sig)
    sync = np.array([fr[1], fr[1], n + 5000, n - 100, 2 ** 40], np.uint64)
    navail = np.array([n - fr[1], 0xFFFFFFFF, 4000, 0xFFFFFFFF, 4000], np.uint32)
    desc = rxctx.to_device(rxctx.make_desc(sync, navail))
    res = rxctx.rx_batch(arena, None, desc, 5)
    rxctx.sync()
    info = res["info"].cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)
    n_pdu = res["n_pdu"].cpu().numpy()
    assert list(n_pdu[:2]) == [1, 1] and rxctx.pdus_of(res, 0)[0]["mpdu"] == mp and rxctx.pdus_of(res, 1)[0]["mpdu"] == mp
    assert list(info["status"][2:]) == [1, 1, 1] and not n_pdu[2:].any()
    # decode-only entry: soft-bit runs outside the LLR arena are refused per frame
    llr = torch.zeros(1000, dtype=torch.float32, device="cuda")
    dd = np.zeros(2, dtype=pkg.rx.DEC_DESC_DTYPE)
    dd["llr_off"], dd["n_llr"], dd["trellis"], dd["len"] = [900, 0], [200, 96], [100, 46], [10, 1]
    r = rxctx.decode_batch(llr, rxctx.to_device(dd), 2)
    rxctx.sync()
    assert not r["n_pdu"].cpu().numpy().any()


def test_ampdu_strict_deaggregation(pkg, rxctx, oracle):
    """gr80211_set_ampdu_strict: the standard's de-aggregation instead of the reference's.  (1) A VHT A-MPDU of three
    different lengths from the reference's Python generator: the reference rule keeps only the first MPDU (its
    length field is OR-accumulated), the strict rule returns all three.  (2) A corrupted delimiter loses its own
    MPDU only; the search resumes at the next valid delimiter.  (3) Aggregated HT PSDUs, which the reference does
    not decode at all."""
    import torch
    from util import golden_ampdu
    tx = pkg.txgen
    rng = np.random.default_rng(51)
    g = golden_ampdu()
    strict_ctx = pkg.RxContext(0)
    strict_ctx.set_ampdu_strict(True)
    try:
        frames = [([g["iq"][0]], g["sync_idx"])]
        # (2) four MPDUs, the second delimiter damaged, through the GPU transmitter
        ms = [tx.make_mpdu(n, rng) for n in (90, 150, 41, 260)]
        ampdu = bytearray(tx.ampdu_vht(ms))
        second = 4 + 92  # delimiter of the second subframe (first: 4 + 90 bytes padded to 92)
        assert ampdu[second + 3] == 0x4E
        ampdu[second + 1] ^= 0x10
        d = rxctx.make_tx_desc(1)
        ns = rxctx.tx_samples(2, 4, 1, len(ampdu))
        d[0]["out_off"], d[0]["psdu_len"], d[0]["format"], d[0]["mcs"], d[0]["scale"] = 50, len(ampdu), 2, 4, 12.0
        arena = torch.zeros(ns + 100, dtype=torch.complex64, device="cuda")
        rxctx.tx_batch(torch.from_numpy(np.frombuffer(bytes(ampdu), np.uint8).copy()).cuda(), d, 1, arena)
        frames.append(([awgn(arena.cpu().numpy(), 33.0, rng).astype(np.complex64)], 50 + 176))
        # (3) HT A-MPDU of three MPDUs
        hs = [tx.make_mpdu(n, rng) for n in (77, 200, 133)]
        sig, m = tx.gen_frame(1, 5, 1, None, n_mpdu_list=hs, ht_aggregate=True)
        ht = np.concatenate([np.zeros(40), awgn(sig[0] * 12.0, 33.0, rng), np.zeros(40)]).astype(np.complex64)
        frames.append(([ht], 40 + 176))
        loose = run_rx(pkg, rxctx, oracle, frames, check_llr=False)
        tight = run_rx(pkg, strict_ctx, oracle, frames, check_llr=False, strict=True)
        assert [p["mpdu"] for p in loose[0]["pdus"]] == g["mpdus"][:1]
        assert [p["mpdu"] for p in tight[0]["pdus"]] == g["mpdus"]
        assert [p["mpdu"] for p in tight[1]["pdus"]] == [ms[0], ms[2], ms[3]]
        assert loose[2]["n_pdu"] == 0 and loose[2]["frame"].m.ampdu == 1
        assert [p["mpdu"] for p in tight[2]["pdus"]] == hs
    finally:
        strict_ctx.close()


def test_signal_and_demod_as_separate_stages(pkg, rxctx, oracle):
    """`signal` alone writes the items and tags the reference block writes (lib/signal_impl.cc:135-202): nsamp
    samples after L-SIG with the CFO taken out, 320 items of padding, L-SIG rate / length and the L-LTF channel;
    `demod` alone, fed with exactly those, returns the soft bits and tags of the fused path bit for bit, and they
    agree with the oracle's."""
    import torch
    from util import H22, build_arena, golden_frames
    for two in (False, True):
        frames = [f for f in golden_frames() if (f["nss"] == 2) == two]
        if two:
            srcs = [([r[0], r[1]], f["sync_idx"]) for f in frames for r in [(H22 @ f["iq"]).astype(np.complex64)]]
        else:
            srcs = [([f["iq"][0]], f["sync_idx"]) for f in frames]
        iq0, iq1, sync, navail = build_arena(srcs)
        rads = np.array([f["rad"] for f in frames], np.float32)
        n = len(frames)
        d0 = torch.from_numpy(iq0).cuda()
        d1 = torch.from_numpy(iq1).cuda() if two else None
        desc = rxctx.make_desc(sync, navail, rad=rads)
        probe = rxctx.signal_batch(d0, d1, desc, n)
        assert (probe["status"] != 1).all()
        out_off = np.concatenate([[0], np.cumsum(probe["nsamp"].astype(np.int64) + 320)])
        info, chan, o0, o1 = rxctx.signal_emit(d0, d1, desc, n, out_off[:-1], int(out_off[-1]))
        h0 = o0.cpu().numpy()
        for i, f in enumerate(frames):
            ns = int(info["nsamp"][i])
            assert ns == int(probe["nsamp"][i]) and int(info["lsig_len"][i]) == int(probe["lsig_len"][i])
            k = np.arange(ns)
            src = iq0[int(sync[i]) + 224:int(sync[i]) + 224 + ns]
            want = src * np.exp(1j * (k + 224) * np.float64(rads[i]))
            got = h0[out_off[i]:out_off[i] + ns]
            assert np.abs(got - want).max() < 2e-5 * max(1.0, np.abs(src).max())
            assert not h0[out_off[i] + ns:out_off[i] + ns + 320].any()
            ref = oracle.rx_frame(iq0[: int(sync[i]) + int(navail[i])], iq1[: int(sync[i]) + int(navail[i])] if two else None,
                                  int(sync[i]), float(rads[i]), want_llr=True)
            assert int(info["lsig_mcs"][i]) == ref["frame"].lsig_mcs and int(info["lsig_len"][i]) == ref["frame"].lsig_len
        # demod alone from those items and tags
        sd = np.zeros(n, dtype=pkg.rx.SIG_DESC_DTYPE)
        sd["off"], sd["n_avail"] = out_off[:-1], info["nsamp"]
        sd["mcs"], sd["len"], sd["nsamp"] = info["lsig_mcs"], info["lsig_len"], info["nsamp"]
        sd["cfo"], sd["snr"], sd["rssi"] = info["cfo_hz"], info["snr"], info["rssi"]
        info2, llr2 = rxctx.demod_tagged(o0, o1, sd, n, chan)
        info1, llr1 = rxctx.demod_batch(d0, d1, desc, n)
        for key in ("status", "format", "mcs", "len", "cr", "ampdu", "trellis", "total", "nSym", "nSS", "cfo_hz"):
            assert np.array_equal(info1[key], info2[key]), key
        for i in range(n):
            a = llr1[int(info1["llr_off"][i]):int(info1["llr_off"][i]) + int(info1["total"][i])]
            b = llr2[int(info2["llr_off"][i]):int(info2["llr_off"][i]) + int(info2["total"][i])]
            if rads[i] == 0.0:
                assert np.array_equal(a, b), i          # same samples, same arithmetic
            else:
                assert np.allclose(a, b, rtol=1e-3, atol=2e-3), i   # derotated once more: rounded differently
    # tags a correct `signal` cannot have sent are refused
    bad = sd.copy()
    bad["nsamp"][0] += 80
    bad["mcs"][1] = 9
    info3, _ = rxctx.demod_tagged(o0, o1, bad, n, chan)
    assert int(info3["status"][0]) == 1 and int(info3["status"][1]) == 1 and int(info3["status"][2]) == int(info2["status"][2])


def test_two_contexts_on_two_threads(pkg, oracle):
    """Distinct contexts share nothing: two threads, each with a context and a stream of its own, decode different
    batches at the same time and get what they get alone (GNU Radio runs every block on a thread of its own)."""
    import threading
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(41)
    jobs = []
    for k in range(2):
        frames = [make_noisy(tx, rng, fmt, mcs, 1, int(rng.integers(100, 900)), 33.0)[0]
                  for fmt, mcs in ((0, 5), (1, 7), (2, 8), (2, 3)) * (6 + 4 * k)]
        iq0, _, sync, navail = build_arena(frames)
        jobs.append((iq0, sync, navail, len(frames)))

    def run(job, reps):
        iq0, sync, navail, n = job
        ctx = pkg.RxContext(0)
        outs = []
        with torch.cuda.stream(torch.cuda.Stream()):
            d0 = torch.from_numpy(iq0).cuda()
            desc = ctx.make_desc(sync, navail)
            for _ in range(reps):
                res = ctx.rx_batch(d0, None, desc, n)
                ctx.sync()
                ln = res["out_len"].cpu().numpy()
                out = res["out"].cpu().numpy()
                outs.append((res["n_pdu"].cpu().numpy().copy(), [bytes(out[i, : ln[i]]) for i in range(n)]))
        ctx.close()
        return outs

    alone = [run(job, 1)[0] for job in jobs]
    assert all(int(a[0].sum()) == job[3] for a, job in zip(alone, jobs))
    got, errors = [None, None], []

    def worker(k):
        try:
            got[k] = run(jobs[k], 12)
        except Exception as e:  # noqa: BLE001 - reported below, from the main thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(150)
    assert not errors and not any(t.is_alive() for t in threads), errors
    for k in range(2):
        for rep in got[k]:
            assert np.array_equal(rep[0], alone[k][0]) and rep[1] == alone[k][1]


def test_submit_and_finish_on_two_contexts_from_one_thread(pkg, oracle):
    """gr80211_rx_submit / gr80211_rx_finish: two contexts kept busy by one thread (submit A, submit B, finish A,
    finish B, several rounds) give what gr80211_rx_batch gives; calls out of order are refused and leave the context
    usable."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(43)
    jobs = []
    for k in range(2):
        frames = [make_noisy(tx, rng, fmt, mcs, 1, int(rng.integers(100, 900)), 33.0)[0]
                  for fmt, mcs in ((0, 4), (1, 6), (2, 8), (2, 2)) * (5 + 3 * k)]
        iq0, _, sync, navail = build_arena(frames)
        jobs.append((torch.from_numpy(iq0).cuda(), pkg.RxContext.make_desc(sync, navail), len(frames)))
    ctxs = [pkg.RxContext(0), pkg.RxContext(0)]
    want = []
    for (d0, desc, n), ctx in zip(jobs, ctxs):
        res = ctx.rx_batch(d0, None, desc, n)
        ctx.sync()
        want.append((res["out_len"].cpu().numpy().copy(), res["n_pdu"].cpu().numpy().copy(), res["out"].cpu().numpy().copy(),
                     res["info"].cpu().numpy().copy()))
        assert int(want[-1][1].sum()) == n
    outs = [(torch.zeros((n, 4104), dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
             torch.zeros(n, dtype=torch.int32, device="cuda"),
             torch.zeros(n * pkg.rx.FRAME_INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")) for _, _, n in jobs]
    with pytest.raises(RuntimeError):
        ctxs[0].rx_finish(*outs[0][:3])                     # nothing submitted
    for rnd in range(4):
        keep = [ctx.rx_submit(d0, None, desc, n) for (d0, desc, n), ctx in zip(jobs, ctxs)]
        if rnd == 1:
            with pytest.raises(RuntimeError):
                ctxs[1].rx_submit(jobs[1][0], None, jobs[1][1], jobs[1][2])   # the batch before is not finished
            with pytest.raises(RuntimeError):
                ctxs[1].rx_batch(jobs[1][0], None, jobs[1][1], jobs[1][2])    # nor may another batch call cut in
        if rnd == 2:
            # every other batch entry point is refused before it touches the workspaces the submitted batch lives
            # in (frame records, descriptors): larger batches than the pending one, so that a call that got past the
            # check would reallocate them
            big = 4 * jobs[1][2] + 1000
            llr = torch.zeros(big * 64, dtype=torch.float32, device="cuda")
            dd = np.zeros(big, dtype=pkg.rx.DEC_DESC_DTYPE)
            dd["llr_off"], dd["n_llr"], dd["trellis"], dd["len"] = np.arange(big) * 64, 64, 30, 1
            sd = np.zeros(big, dtype=pkg.rx.SIG_DESC_DTYPE)
            sd["n_avail"], sd["len"], sd["nsamp"] = 400, 20, 640
            fdesc = pkg.RxContext.make_desc(np.zeros(big, np.uint64), np.full(big, 400, np.uint32))
            intruders = [
                lambda: ctxs[1].decode_batch(llr, dd, big),
                lambda: ctxs[1].demod_tagged(jobs[1][0], None, sd, big, np.zeros((big, 64), np.complex64)),
                lambda: ctxs[1].demod_batch(jobs[1][0], None, fdesc, big),
                lambda: ctxs[1].signal_batch(jobs[1][0], None, fdesc, big),
                lambda: ctxs[1].signal_emit(jobs[1][0], None, fdesc, big, np.arange(big, dtype=np.uint64) * 1000, big * 1000),
                lambda: ctxs[1].rx_capture(jobs[1][0]),
                lambda: ctxs[1].detect(jobs[1][0]),
            ]
            for call in intruders:
                with pytest.raises(RuntimeError, match="waiting for gr80211_rx_finish"):
                    call()
            # a finish call with a bad argument is refused and leaves the batch where it is
            with pytest.raises(RuntimeError, match="bad argument"):
                ctxs[1]._check(ctxs[1].lib.gr80211_rx_finish(ctxs[1].h, None, 4104, None, None, None), "gr80211_rx_finish")
        for ctx, o in zip(ctxs, outs):
            ctx.rx_finish(o[0], o[1], o[2], o[3])
        for ctx in ctxs:
            ctx.sync()
        del keep
        for (ln, npd, out, info), o in zip(want, outs):
            assert np.array_equal(o[1].cpu().numpy(), ln) and np.array_equal(o[2].cpu().numpy(), npd)
            got = o[0].cpu().numpy()
            assert all(bytes(got[i, : ln[i]]) == bytes(out[i, : ln[i]]) for i in range(len(ln)))
            assert np.array_equal(o[3].cpu().numpy(), info)
        for o in outs:
            for t in o:
                t.zero_()
    for ctx in ctxs:
        ctx.close()



def test_resume_after_signal_batch_equals_rx_batch(pkg, oracle):
    """gr80211_signal_batch, then gr80211_rx_resume for a subset of its frames (what a streaming caller does around
    `signal`'s trigger search): the same PDUs and records as gr80211_rx_batch over those frames, with nothing waited
    for in between; out of order use is refused; a caller whose copy of the records understates the frames gets a
    batch that is not run (no PDUs, flag set) instead of one that overruns its buffers."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(44)
    frames = []
    for k in range(150):
        fmt = int(rng.integers(0, 3))
        mcs = int(rng.integers(0, 8 if fmt < 2 else 9))
        frames.append(make_noisy(tx, rng, fmt, mcs, 1, int(rng.integers(30, 1400)), 33.0)[0])
        if k % 9 == 0:
            noise = (0.1 * (rng.standard_normal(900) + 1j * rng.standard_normal(900))).astype(np.complex64)
            frames.append(([noise], 300))      # a sync byte on noise: L-SIG fails
    iq0, _, sync, navail = build_arena(frames)
    n = len(frames)
    d0 = torch.from_numpy(iq0).cuda()
    desc = pkg.RxContext.make_desc(sync, navail)
    ctx, ref = pkg.RxContext(0), pkg.RxContext(0)
    try:
        info_all = ctx.signal_batch(d0, None, desc, n)
        sel = np.array([i for i in range(n) if info_all["status"][i] != 1 and i % 3 != 1], dtype=np.int32)
        dsel = desc[sel].copy()
        dsel["n_avail"] = np.minimum(dsel["n_avail"], 224 + np.maximum(info_all["nsamp"][sel], 0))  # cut to the frame's span
        m = len(sel)
        out = torch.zeros((m, 4104), dtype=torch.uint8, device="cuda")
        out_len = torch.full((m,), -1, dtype=torch.int32, device="cuda")
        n_pdu = torch.full((m,), -1, dtype=torch.int32, device="cuda")
        info = torch.zeros(m * pkg.rx.FRAME_INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        ctx.rx_resume(d0, None, dsel, sel, info_all, out, out_len, n_pdu, info)
        with pytest.raises(RuntimeError, match="holds no results"):
            ctx.rx_resume(d0, None, dsel, sel, info_all, out, out_len, n_pdu)       # the results are spent
        ctx.sync()
        assert ctx.lib.gr80211_last_batch_refused(ctx.h) == 0
        want = ref.rx_batch(d0, None, dsel, m)
        ref.sync()
        assert torch.equal(out_len, want["out_len"]) and torch.equal(n_pdu, want["n_pdu"])
        assert int(n_pdu.sum().item()) >= m - 3
        wl = want["out_len"].cpu().numpy()
        go, wo = out.cpu().numpy(), want["out"].cpu().numpy()
        assert all(bytes(go[i, : wl[i]]) == bytes(wo[i, : wl[i]]) for i in range(m))
        gi = info.cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)
        wi = want["info"].cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)
        for k in TAGS + ("nCBPS", "nDBPS", "llr_off"):
            assert np.array_equal(gi[k], wi[k]), k
        # another batch call between the two steps spends the results as well
        ctx.signal_batch(d0, None, desc, n)
        ctx.rx_batch(d0, None, dsel, m)
        with pytest.raises(RuntimeError, match="holds no results"):
            ctx.rx_resume(d0, None, dsel, sel, info_all, out, out_len, n_pdu)
        # a fresh context (small buffers) and records that understate every frame: refused on the device, nothing runs
        small = pkg.RxContext(0)
        small.signal_batch(d0, None, desc, n)
        lie = info_all.copy()
        lie["total"] = 8
        lie["trellis"] = 64
        out_len.fill_(-1)
        small.rx_resume(d0, None, dsel, sel, lie, out, out_len, n_pdu)
        small.sync()
        assert small.lib.gr80211_last_batch_refused(small.h) == 1
        assert int(out_len.abs().sum().item()) == 0 and int(n_pdu.abs().sum().item()) == 0
        # and the same context works again with honest records
        small.signal_batch(d0, None, desc, n)
        small.rx_resume(d0, None, dsel, sel, info_all, out, out_len, n_pdu)
        small.sync()
        assert small.lib.gr80211_last_batch_refused(small.h) == 0 and torch.equal(out_len, want["out_len"])
        small.close()
        # a context whose buffers are large already (`ctx` has run the whole batch above) and records that understate
        # the NUMBER of frames the decoder admits (most marked as dropped): every buffer has room, so only the group
        # count the decoder was launched for can tell -- the scheduler arrives at more groups and refuses the batch
        ctx.signal_batch(d0, None, desc, n)
        few = info_all.copy()
        few["status"][info_all["cr"] != info_all["cr"][sel[0]]] = 1   # frames of one code rate only: one group, not four
        out_len.fill_(-1)
        ctx.rx_resume(d0, None, dsel, sel, few, out, out_len, n_pdu)
        ctx.sync()
        assert ctx.lib.gr80211_last_batch_refused(ctx.h) == 1
        assert int(out_len.abs().sum().item()) == 0 and int(n_pdu.abs().sum().item()) == 0
        # ... and none at all: the host skips the decoder, the device still says what it found
        none = info_all.copy()
        none["status"][:] = 1
        ctx.signal_batch(d0, None, desc, n)
        ctx.rx_resume(d0, None, dsel, sel, none, out, out_len, n_pdu)
        ctx.sync()
        assert ctx.lib.gr80211_last_batch_refused(ctx.h) == 1
        ctx.signal_batch(d0, None, desc, n)
        ctx.rx_resume(d0, None, dsel, sel, info_all, out, out_len, n_pdu)
        ctx.sync()
        assert ctx.lib.gr80211_last_batch_refused(ctx.h) == 0 and torch.equal(out_len, want["out_len"])
    finally:
        ctx.close()
        ref.close()


def test_rx_batch_without_the_wait(pkg):
    """gr80211_rx_batch with GR80211_FLAG_NO_WAIT: the whole batch is queued without the host reading the scheduler's
    totals; PDUs and records equal those of the call that waits, for a batch of mixed formats with sync bytes on
    noise among them and for part of it (fewer groups than the grid was launched for).  Before any batch has sized
    the work buffers the flag is refused; a batch that needs more room than the batches before left is not run (flag
    set, nothing written) and runs when repeated without the flag, after which the larger batch passes with it."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(45)
    frames = []
    for k in range(200):
        fmt = int(rng.integers(0, 3))
        mcs = int(rng.integers(0, 8 if fmt < 2 else 9))
        frames.append(make_noisy(tx, rng, fmt, mcs, 1, int(rng.integers(30, 1400)), 33.0)[0])
        if k % 11 == 0:
            noise = (0.1 * (rng.standard_normal(900) + 1j * rng.standard_normal(900))).astype(np.complex64)
            frames.append(([noise], 300))
    iq0, _, sync, navail = build_arena(frames)
    n = len(frames)
    d0 = torch.from_numpy(iq0).cuda()
    desc = pkg.RxContext.make_desc(sync, navail)
    ctx, ref = pkg.RxContext(0), pkg.RxContext(0)

    def run(c, m, no_wait):
        out = torch.zeros((m, 4104), dtype=torch.uint8, device="cuda")
        out_len = torch.full((m,), -1, dtype=torch.int32, device="cuda")
        n_pdu = torch.full((m,), -1, dtype=torch.int32, device="cuda")
        info = torch.zeros(m * pkg.rx.FRAME_INFO_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
        c.rx_batch_into(d0, None, desc[:m], m, out, out_len, n_pdu, info, no_wait=no_wait)
        c.sync()
        return out.cpu().numpy(), out_len.cpu().numpy(), n_pdu.cpu().numpy(), info.cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)

    def same(a, b):
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert all(bytes(a[0][i, : a[1][i]]) == bytes(b[0][i, : a[1][i]]) for i in range(len(a[1])))
        for k in TAGS + ("nCBPS", "nDBPS", "llr_off", "status"):
            assert np.array_equal(a[3][k], b[3][k]), k

    try:
        with pytest.raises(RuntimeError, match="before a batch without it"):
            run(ctx, 8, True)
        small = 70
        want_small, want_all = run(ref, small, False), run(ref, n, False)
        assert int(want_all[2].sum()) >= 195
        run(ctx, small, False)                         # sizes the buffers for 70 frames
        got = run(ctx, small, True)
        assert not ctx.last_batch_refused()
        same(got, want_small)
        same(run(ctx, 40, True), run(ref, 40, False))  # a part of it: fewer groups than were launched for
        got = run(ctx, n, True)                        # three times the frames: no room
        assert ctx.last_batch_refused()
        assert not got[1].any() and not got[2].any()
        same(run(ctx, n, False), want_all)             # the call that waits makes room
        assert not ctx.last_batch_refused()
        got = run(ctx, n, True)
        assert not ctx.last_batch_refused()
        same(got, want_all)
    finally:
        ctx.close()
        ref.close()


def test_copies_by_kernel_over_the_link(pkg, rxctx):
    """gr80211_pull_h2d / gr80211_push_d2h: copies done by a kernel between page-locked host memory and the device, at
    every alignment of source and destination a stream of items can have (16, 8, 4 bytes and none), sizes from one
    byte to a few megabytes; gr80211_pull_done turns 1 in the order of the numbers handed out; memory that is not
    page-locked is refused without leaving an error behind for the next call."""
    import ctypes as C
    lib, c = rxctx.lib, rxctx.h
    n = (3 << 20) + 77
    h_src = lib.gr80211_host_alloc(c, n + 64)
    h_dst = lib.gr80211_host_alloc(c, n + 64)
    d = lib.gr80211_dev_alloc(c, n + 64)
    assert h_src and h_dst and d
    try:
        src = np.ctypeslib.as_array(C.cast(h_src, C.POINTER(C.c_uint8)), (n + 64,))
        dst = np.ctypeslib.as_array(C.cast(h_dst, C.POINTER(C.c_uint8)), (n + 64,))
        rng = np.random.default_rng(11)
        src[:] = rng.integers(0, 256, n + 64, dtype=np.uint8)
        last = 0
        for so, do, size in ((0, 0, n), (8, 16, 1 << 20), (8, 8, 65536 + 8), (4, 12, 4097), (1, 3, 1000), (0, 0, 1), (16, 0, 255),
                             (3, 16, (1 << 20) + 5)):
            dst[:] = 0
            num = lib.gr80211_pull_h2d(c, d + do, h_src + so, size, None)
            assert num == last + 1, num
            last = num
            assert lib.gr80211_pull_done(c, num, 1) == 1 and lib.gr80211_pull_done(c, num, 0) == 1
            assert lib.gr80211_push_d2h(c, h_dst + so, d + do, size, None) == 0
            rxctx.sync()
            assert np.array_equal(dst[so:so + size], src[so:so + size]), (so, do, size)
            assert not dst[:so].any() and not dst[so + size:].any()
        assert lib.gr80211_pull_done(c, last + 1, 0) < 0          # a number never handed out
        plain = np.zeros(4096, np.uint8)
        assert lib.gr80211_pull_h2d(c, d, plain.ctypes.data, 4096, None) < 0
        assert lib.gr80211_push_d2h(c, plain.ctypes.data, d, 4096, None) < 0
        # and the context is as good as before
        num = lib.gr80211_pull_h2d(c, d, h_src, 4096, None)
        assert num == last + 1 and lib.gr80211_pull_done(c, num, 1) == 1
    finally:
        lib.gr80211_dev_free(c, d)
        lib.gr80211_host_free(c, h_src)
        lib.gr80211_host_free(c, h_dst)


def test_six_contexts_on_a_pool_of_three_work_sets(pkg):
    """Six contexts take turns over batches of different sizes (as the benchmark's rotation does) while sharing
    pooled work buffers (three for survivor words, five for soft bits): every batch gives what a context with buffers
    of its own gives, with and without the wait inside the call, the pool never holds more than it may, and leaving it
    frees everything."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(61)
    batches = []
    for b in range(4):
        frames = [make_noisy(tx, rng, int(rng.integers(0, 3)), int(rng.integers(0, 8)), 1, int(rng.integers(40, 1500)), 30.0)[0]
                  for _ in range(int(rng.integers(90, 400)))]
        iq0, _, sync, navail = build_arena(frames)
        batches.append((torch.from_numpy(iq0).cuda(), torch.from_numpy(pkg.RxContext.make_desc(sync, navail).view(np.uint8).reshape(-1)).cuda(), len(frames)))
    ref = pkg.RxContext(0)
    want = []
    for iq, desc, n in batches:
        r = ref.rx_batch(iq, None, desc, n)
        ref.sync()
        want.append((r["out_len"].clone(), r["n_pdu"].clone(), r["out"].clone()))
    ctxs = [pkg.RxContext(0) for _ in range(6)]
    try:
        for c in ctxs:
            c.use_work_pool(3)
        outs = [[(torch.zeros((n, 4104), dtype=torch.uint8, device="cuda"), torch.zeros(n, dtype=torch.int32, device="cuda"),
                  torch.zeros(n, dtype=torch.int32, device="cuda")) for _, _, n in batches] for _ in ctxs]
        for no_wait in (False, True, True):
            for step in range(24):
                k, b = step % 6, (step // 2) % 4
                iq, desc, n = batches[b]
                with torch.cuda.stream(ctxs[k].own_stream()):
                    ctxs[k].rx_batch_into(iq, None, desc, n, *outs[k][b], no_wait=no_wait)
            torch.cuda.synchronize()
            for k, c in enumerate(ctxs):
                assert not c.last_batch_refused()
            for step in range(24):
                k, b = step % 6, (step // 2) % 4
                o, ol, npd = outs[k][b]
                assert torch.equal(ol, want[b][0]) and torch.equal(npd, want[b][1])
                keep = torch.arange(4104, device="cuda")[None, :] < ol[:, None]
                assert torch.equal(o[keep], want[b][2][keep])
                o.zero_(), ol.zero_(), npd.zero_()
        own, pool, sets = ctxs[0].work_memory()
        assert sets <= 3 and pool > 0
        assert all(c.work_memory()[0] == 0 for c in ctxs)   # nobody grew buffers of its own
        one = ref.work_memory()[0]
        assert pool <= 5 * one * 1.05 + (1 << 20)   # three survivor buffers, five soft-bit buffers
        for c in ctxs:
            c.use_work_pool(0)
        assert ctxs[0].work_memory()[1:] == (0, 0)
        # on its own again, a context grows its own buffers
        iq, desc, n = batches[0]
        ctxs[0].rx_batch_into(iq, None, desc, n, *outs[0][0])
        ctxs[0].sync()
        assert torch.equal(outs[0][0][1], want[0][0]) and ctxs[0].work_memory()[0] > 0
    finally:
        for c in ctxs + [ref]:
            c.close()


def test_axis_value_soft_bits_give_the_same_bytes(pkg):
    """The experimental soft-bit layout (GR80211_AXIS_LLR=1: one float per constellation axis from the demodulator, folded
    into soft bits by the decoder's tile loader) against the default path on batches of one modulation and many lengths:
    the same PDU bytes, lengths and counts for every modulation family and both stream counts."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import axis_parity
    saved = {k: os.environ.get(k) for k in ("GR80211_GROUP_WAVES", "GR80211_AXIS_LLR")}
    try:
        bad = axis_parity.compare(torch, pkg, [(0, 0, 1), (0, 5, 1), (1, 7, 1), (1, 11, 2), (2, 8, 1), (2, 4, 2), (2, 8, 2)],
                                  snr0s=(12.0,), n=200)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert bad == 0
