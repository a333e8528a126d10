"""decode path on the GPU (gr80211_decode_batch) against the CPU oracle: bit-exact decoded bits,
PDUs and FCS verdicts, for every code rate, ragged lengths, ties, erasures and edge cases."""
import os

import numpy as np
import pytest

from util import unpack_bits

pytestmark = pytest.mark.gpu

CR_KEEP = {0: [1, 1], 1: [1, 1, 1, 0], 2: [1, 1, 1, 0, 0, 1], 3: [1, 1, 1, 0, 0, 1, 1, 0, 0, 1]}


def n_llr_for(trellis, cr):
    pat = CR_KEEP[cr]
    reps = -(-2 * trellis // len(pat))
    return int(np.sum(np.tile(pat, reps)[: 2 * trellis]))


def coded_frame(tx, rng, fmt, cr, nbytes, mcs=0, snr_scale=4.0, noise=1.0, integer=False):
    """LLR stream of one frame whose PSDU is a valid MPDU (L/HT) or single-subframe A-MPDU (VHT)."""
    mpdu = tx.make_mpdu(nbytes, rng)
    if fmt == 2:
        payload = tx.ampdu_vht([mpdu])
        trellis = 16 + 8 * len(payload) + 6
        pad = (-trellis) % {0: 1, 1: 2, 2: 3, 3: 5}[cr]
        trellis += pad
        bits = np.concatenate([np.zeros(16, np.uint8), np.unpackbits(np.frombuffer(payload, np.uint8), bitorder="little"),
                               np.zeros(pad, np.uint8)])
        scr = bits ^ tx.scramble_seq(len(bits), 93)
        scr = np.concatenate([scr, np.zeros(6, np.uint8)])
        length = len(payload)
    else:
        trellis = 16 + 8 * nbytes + 6
        bits = np.concatenate([np.zeros(16, np.uint8), np.unpackbits(np.frombuffer(mpdu, np.uint8), bitorder="little"),
                               np.zeros(6, np.uint8)])
        scr = bits ^ tx.scramble_seq(len(bits), 93)
        scr[-6:] = 0
        length = nbytes
    coded = tx.puncture(tx.bcc_encode(scr), cr)[: n_llr_for(trellis, cr)]
    llr = (2.0 * coded - 1.0) * snr_scale + noise * rng.standard_normal(len(coded))
    if integer:
        llr = np.round(llr)
    return dict(llr=llr.astype(np.float32), fmt=fmt, len=length, cr=cr, mcs=mcs, ampdu=1 if fmt == 2 else 0,
                trellis=trellis, mpdu=mpdu)


def run_and_compare(pkg, rxctx, oracle, frames, expect_all_ok=None):
    import torch
    rx = pkg.rx
    n = len(frames)
    desc = np.zeros(n, dtype=rx.DEC_DESC_DTYPE)
    off = 0
    parts = []
    for i, f in enumerate(frames):
        desc[i] = (off, len(f["llr"]), f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], 0)
        parts.append(f["llr"])
        off += len(f["llr"]) + 3  # unaligned, non-contiguous runs
        parts.append(np.full(3, np.nan, np.float32))
    llr = torch.from_numpy(np.concatenate(parts)).cuda()
    res = rxctx.decode_batch(llr, desc, n, want_bits=True)
    rxctx.sync()
    out_len = res["out_len"].cpu().numpy()
    n_pdu = res["n_pdu"].cpu().numpy()
    out = res["out"].cpu().numpy()
    bits = res["bits"].cpu().numpy()
    n_ok = 0
    for i, f in enumerate(frames):
        ref = oracle.decode(f["llr"], f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], want_bits=True)
        got_bits = unpack_bits(bits[i], f["trellis"])
        assert np.array_equal(got_bits, ref["bits"]), "decoded bits differ, frame %d (cr %d T %d)" % (i, f["cr"], f["trellis"])
        assert int(n_pdu[i]) == ref["n_pdu"], i
        assert bytes(out[i, : out_len[i]]) == ref["blob"], i
        n_ok += ref["n_pdu"]
    if expect_all_ok:
        assert n_ok == n
    return n_ok


def test_all_rates_mixed_lengths(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(11)
    frames = []
    for k in range(150):
        cr = k % 4
        fmt = (0, 1, 2)[k % 3]
        frames.append(coded_frame(tx, rng, fmt, cr, int(rng.integers(20, 420)), mcs=k % 8))
    run_and_compare(pkg, rxctx, oracle, frames, expect_all_ok=True)


def test_low_snr_and_ties_bit_exact(pkg, rxctx, oracle):
    """Small-integer LLRs make equal path metrics common: survivor choice on ties must match
    the reference rule (strict '>' scanning predecessors in ascending order)."""
    tx = pkg.txgen
    rng = np.random.default_rng(12)
    frames = []
    for k in range(96):
        frames.append(coded_frame(tx, rng, (0, 2)[k & 1], k % 4, int(rng.integers(30, 200)), snr_scale=1.0,
                                  noise=1.5, integer=True))
    run_and_compare(pkg, rxctx, oracle, frames)


def test_erasures_and_pure_noise(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(13)
    frames = []
    for k in range(40):
        f = coded_frame(tx, rng, 0, k % 4, int(rng.integers(40, 300)))
        if k % 3 == 0:
            f["llr"][rng.random(len(f["llr"])) < 0.2] = 0.0       # erasures
        elif k % 3 == 1:
            f["llr"] = rng.standard_normal(len(f["llr"])).astype(np.float32)  # garbage: FCS must fail on both
        frames.append(f)
    run_and_compare(pkg, rxctx, oracle, frames)


def test_edge_lengths(pkg, rxctx, oracle):
    tx = pkg.txgen
    rng = np.random.default_rng(14)
    frames = [coded_frame(tx, rng, 0, 0, 14), coded_frame(tx, rng, 0, 2, 15), coded_frame(tx, rng, 1, 3, 17),
              coded_frame(tx, rng, 0, 1, 4095), coded_frame(tx, rng, 0, 0, 4095), coded_frame(tx, rng, 2, 2, 4084),
              coded_frame(tx, rng, 1, 3, 2301)]
    # HT A-MPDU flag: the reference publishes nothing
    f = coded_frame(tx, rng, 1, 0, 100)
    f["ampdu"] = 1
    frames.append(f)
    # over the decoder's limits: dropped without output (lib/decode_impl.cc:93)
    g = coded_frame(tx, rng, 0, 0, 100)
    g["len"] = 4096
    frames.append(g)
    run_and_compare(pkg, rxctx, oracle, frames)


def test_single_and_empty_batch(pkg, rxctx, oracle):
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(15)
    run_and_compare(pkg, rxctx, oracle, [coded_frame(tx, rng, 2, 2, 333)], expect_all_ok=True)
    llr = torch.zeros(16, dtype=torch.float32, device="cuda")
    res = rxctx.decode_batch(llr, np.zeros(0, dtype=pkg.rx.DEC_DESC_DTYPE), 0)
    assert res["out"].shape[0] == 0


def test_bad_descriptor_rejected(pkg, rxctx):
    import torch
    llr = torch.zeros(100, dtype=torch.float32, device="cuda")
    desc = np.zeros(1, dtype=pkg.rx.DEC_DESC_DTYPE)
    desc[0] = (50, 100, 0, 20, 0, 0, 0, 182, 0)   # runs past the arena
    with pytest.raises(RuntimeError):
        rxctx.decode_batch(llr, desc, 1)


def test_linearity_property_full_size(pkg, rxctx, oracle):
    """Full-size VHT MCS8-like frames (12168 trellis steps, rate 3/4): decode(encode(x)) = x for many
    frames at once, and a spot check of a few of them against the oracle."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(16)
    frames = [coded_frame(tx, rng, 2, 2, 1496, mcs=8, snr_scale=3.0) for _ in range(130)]
    rx = pkg.rx
    n = len(frames)
    desc = np.zeros(n, dtype=rx.DEC_DESC_DTYPE)
    off = 0
    for i, f in enumerate(frames):
        desc[i] = (off, len(f["llr"]), f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], 0)
        off += len(f["llr"])
    llr = torch.from_numpy(np.concatenate([f["llr"] for f in frames])).cuda()
    res = rxctx.decode_batch(llr, desc, n)
    rxctx.sync()
    for i, f in enumerate(frames):
        got = rxctx.pdus_of(res, i)
        assert len(got) == 1 and got[0]["mpdu"] == f["mpdu"] and got[0]["mcs"] == 8 and got[0]["format"] == 2
    for i in (0, 64, 129):
        f = frames[i]
        ref = oracle.decode(f["llr"], f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"])
        ln = int(res["out_len"][i].item())
        assert bytes(res["out"][i, :ln].cpu().numpy()) == ref["blob"]


@pytest.fixture(scope="module")
def ctx_by_waves(pkg):
    """Contexts pinned to 1, 2 and 4 wavefronts per decoder group, to the one-wave form of batches beyond 1024
    groups (10) and to one wavefront per frame with a trellis state per lane (64) (by default the library chooses by
    batch size, so the small batches of these tests would all take the last one)."""
    import os
    made = {}
    old = os.environ.get("GR80211_GROUP_WAVES")
    try:
        for w in (1, 2, 4, 10, 64):
            os.environ["GR80211_GROUP_WAVES"] = str(w)
            made[w] = pkg.RxContext(0)
    finally:
        if old is None:
            os.environ.pop("GR80211_GROUP_WAVES", None)
        else:
            os.environ["GR80211_GROUP_WAVES"] = old
    yield made
    for c in made.values():
        c.close()


@pytest.mark.parametrize("waves", [1, 2, 4, 10, 64])
def test_each_group_width_against_oracle(pkg, ctx_by_waves, oracle, waves):
    """Ragged lengths, all four code rates, integer LLRs with many ties: every way of spreading a group over
    wavefronts gives the oracle's bits and PDUs."""
    tx = pkg.txgen
    rng = np.random.default_rng(17)
    frames = []
    for k in range(200):
        cr = k % 4
        fmt = (0, 1, 2)[k % 3]
        frames.append(coded_frame(tx, rng, fmt, cr, int(rng.integers(20, 520)), mcs=k % 8, snr_scale=1.5, noise=1.2,
                                  integer=(k % 2 == 0)))
    run_and_compare(pkg, ctx_by_waves[waves], oracle, frames)


def test_group_widths_agree_bit_for_bit(pkg, ctx_by_waves):
    """A few thousand noisy frames (many groups, partly filled last groups per rate): survivor words written by 1
    2 or 4 waves per group lead to identical decoded bits and PDU slots."""
    import torch
    tx = pkg.txgen
    rx = pkg.rx
    rng = np.random.default_rng(18)
    base = [coded_frame(tx, rng, (0, 1, 2)[k % 3], k % 4, int(rng.integers(30, 700)), mcs=k % 8, snr_scale=1.6, noise=0.8)
            for k in range(48)]
    n = 3000
    desc = np.zeros(n, dtype=rx.DEC_DESC_DTYPE)
    parts, off = [], 0
    for i in range(n):
        f = base[i % len(base)]
        llr = f["llr"] + (0.7 * rng.standard_normal(len(f["llr"]))).astype(np.float32)
        desc[i] = (off, len(llr), f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], 0)
        parts.append(llr)
        off += len(llr)
    llr = torch.from_numpy(np.concatenate(parts)).cuda()
    got = {}
    for w, c in ctx_by_waves.items():
        r = c.decode_batch(llr, desc, n, want_bits=True, out_stride=768, bits_stride=768)
        c.sync()
        got[w] = (r["out"].cpu().numpy(), r["out_len"].cpu().numpy(), r["n_pdu"].cpu().numpy(), r["bits"].cpu().numpy())
    assert 0 < int(got[1][2].sum()) < n   # some frames fail the FCS, some pass
    valid = np.arange(768)[None, :] < got[1][1][:, None]   # the slots are uninitialised beyond out_len
    for w in (2, 4, 10, 64):
        assert np.array_equal(got[1][1], got[w][1]) and np.array_equal(got[1][2], got[w][2]), w
        # decoded bits up to each frame's own trellis length (what lies beyond depends on the other frames that
        # happen to share its group)
        tb = np.arange(768 * 8)[None, :] < desc["trellis"][:, None]
        assert np.array_equal(np.unpackbits(got[1][3], axis=1, bitorder="little")[tb],
                              np.unpackbits(got[w][3], axis=1, bitorder="little")[tb]), w
        assert np.array_equal(got[1][0][valid], got[w][0][valid]), w


def test_soft_bits_through_lds_experiment_is_bit_exact(pkg, ctx_by_waves, oracle):
    """GR80211_LS_RING=1: the one-wave-per-frame decoder with its soft bits fetched by vector loads and read from a
    ring in LDS instead of scalar loads (an experiment the design notes quote; the library does not use it by itself).
    Same frames as test_each_group_width_against_oracle, same answer: the oracle's."""
    import os
    tx = pkg.txgen
    rng = np.random.default_rng(17)
    frames = []
    for k in range(200):
        frames.append(coded_frame(tx, rng, (0, 1, 2)[k % 3], k % 4, int(rng.integers(20, 520)), mcs=k % 8, snr_scale=1.5, noise=1.2,
                                  integer=(k % 2 == 0)))
    os.environ["GR80211_LS_RING"] = "1"
    try:
        run_and_compare(pkg, ctx_by_waves[64], oracle, frames)
    finally:
        os.environ.pop("GR80211_LS_RING", None)


def test_small_batch_form_is_the_callers_choice(pkg):
    """gr80211_set_lane_state_max moves the batch size up to which the one-frame-per-wave decoder is taken: the same
    small batch with the default, with the form switched off and with the default again gives the same bytes."""
    import torch
    tx = pkg.txgen
    rx = pkg.rx
    rng = np.random.default_rng(3)
    frames = [coded_frame(tx, rng, (0, 1, 2)[k % 3], k % 4, int(rng.integers(30, 900)), mcs=k % 8, snr_scale=1.6, noise=0.9)
              for k in range(150)]
    desc = np.zeros(len(frames), dtype=rx.DEC_DESC_DTYPE)
    off = 0
    for i, f in enumerate(frames):
        desc[i] = (off, len(f["llr"]), f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], 0)
        off += len(f["llr"])
    llr = torch.from_numpy(np.concatenate([f["llr"] for f in frames])).cuda()
    c = pkg.RxContext(0)
    try:
        got = []
        for limit in (-1, 0, 100, -1):   # 100 < 150 frames: the limit counts frames
            c.set_lane_state_max(limit)
            r = c.decode_batch(llr, desc, len(frames), out_stride=1024)
            c.sync()
            ol = r["out_len"].cpu().numpy()
            valid = np.arange(1024)[None, :] < ol[:, None]
            got.append((ol, r["n_pdu"].cpu().numpy(), r["out"].cpu().numpy() * valid))
        assert 0 < int(got[0][1].sum())
        for g in got[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(got[0], g))
    finally:
        c.close()


def test_reference_bit_vector_decodes_to_reference_packet(pkg, ctx_by_waves):
    """The interleaved coded bits the reference keeps for its example packet (tools/txbits.py, HT MCS0), put
    through its own deinterleaver map as hard soft bits: both decoder forms must return the packet its IQ
    fixtures carry.  No oracle in the loop: reference data in, reference data out."""
    import torch
    from util import GOLDEN, ref_fixture_mpdu
    z = np.load(os.path.join(GOLDEN, "ref_txbits.npz"))
    inter = z["htMcs0interBits"]
    m = np.asarray(pkg.txgen.interleave_map(52, 1, 13))
    llr = np.zeros(len(inter), np.float32)
    for s in range(len(inter) // 52):
        # the transmitter puts coded bit k at place m[k]; the receiver reads it back from there
        llr[52 * s:52 * s + 52] = inter[52 * s + m].astype(np.float32) * 2 - 1
    assert np.array_equal((llr > 0).astype(np.uint8), z["htMcs0CodedBits"])
    n = 130   # more than a wave of copies: two groups
    desc = np.zeros(n, dtype=pkg.rx.DEC_DESC_DTYPE)
    for i in range(n):
        desc[i] = (i * len(llr), len(llr), 1, 94, 0, 0, 0, 8 * 94 + 22, 0)
    d_llr = torch.from_numpy(np.tile(llr, n)).cuda()
    want = bytes([1, 94, 0]) + ref_fixture_mpdu() + bytes([0])
    for waves in (1, 4, 64):
        c = ctx_by_waves[waves]
        r = c.decode_batch(d_llr, desc, n)
        c.sync()
        lens = r["out_len"].cpu().numpy()
        out = r["out"].cpu().numpy()
        assert (r["n_pdu"].cpu().numpy() == 1).all()
        for i in (0, 63, 64, 129):
            assert bytes(out[i, : lens[i]]) == want, (waves, i)
