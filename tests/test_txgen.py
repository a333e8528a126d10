"""The corpus generator reproduces the reference generator's waveform and round-trips through the oracle."""
import numpy as np
import pytest

from util import H22, MALFORMED, awgn, golden_frames


@pytest.mark.parametrize("fr", [f for f in golden_frames() if f["rad"] == 0.0], ids=lambda f: f["name"])
def test_waveform_equals_python_reference(pkg, fr):
    tx = pkg.txgen
    sig, m = tx.gen_frame(fr["fmt"], fr["mcs"], fr["nss"], fr["mpdu"])
    sig = sig * 12.0 * np.sqrt(fr["nss"])
    if fr["nss"] == 2:
        sig = H22 @ sig
    ref = fr["iq"][:, 400:400 + sig.shape[1]]
    assert sig.shape[1] == m.n_samples() == fr["iq"].shape[1] - 800
    assert np.abs(sig - ref).max() < 1e-6


CASES = [(0, m, 1, 40 + 37 * m) for m in range(8)] + [(1, m, 1, 50 + 41 * m) for m in range(8)] + \
        [(1, m, 2, 60 + 53 * (m - 8)) for m in range(8, 16)] + [(2, m, 1, 30 + 29 * m) for m in range(9)] + \
        [(2, m, 2, 45 + 31 * m) for m in range(9)]


@pytest.mark.parametrize("fmt,mcs,nss,nbytes", CASES)
def test_roundtrip_through_oracle(pkg, oracle, fmt, mcs, nss, nbytes):
    tx = pkg.txgen
    rng = np.random.default_rng(1000 * fmt + 10 * mcs + nss)
    mpdu = tx.make_mpdu(nbytes, rng)
    sig, m = tx.gen_frame(fmt, mcs, nss, mpdu)
    sig = sig * 12.0 * np.sqrt(m.nss)
    rx = H22 @ sig if m.nss == 2 else sig
    rx = awgn(rx, 35.0, rng)
    pad = np.zeros((rx.shape[0], 50))
    rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
    r = oracle.rx_frame(rx[0], rx[1] if m.nss == 2 else None, 50 + 176, 0.0)
    assert r["status"] == 0
    f = r["frame"]
    assert (f.m.format, f.m.mcs, f.m.nSS, f.m.nSym) == (fmt, mcs, m.nss, m.nsym)
    assert f.lsig_len == m.lsig_len
    assert [p["mpdu"] for p in r["pdus"]] == [mpdu]


def test_vht_multi_mpdu_ampdu(pkg, oracle):
    """Two equal-length subframes: both are published (the reference's length accumulation is
    only well defined for equal lengths, lib/decode_impl.cc:336-351)."""
    tx = pkg.txgen
    rng = np.random.default_rng(5)
    ms = [tx.make_mpdu(120, rng), tx.make_mpdu(120, rng)]
    sig, m = tx.gen_frame(tx.F_VHT, 5, 1, None, n_mpdu_list=ms)
    rx = np.concatenate([np.zeros(30), sig[0] * 12.0, np.zeros(30)]).astype(np.complex64)
    r = oracle.rx_frame(rx, None, 30 + 176, 0.0)
    assert [p["mpdu"] for p in r["pdus"]] == ms


@pytest.mark.parametrize("nss", [1, 2])
def test_vht_mcs9_roundtrip_with_truncated_ndbps(pkg, oracle, nss):
    """VHT MCS9 at 20 MHz: the receiver's N_DBPS comes out of an integer division (346 per stream and a third
    lost, lib/cloud80211phy.cc:1287-1303); a transmitter that pads at the end of the coded stream is decoded."""
    tx = pkg.txgen
    rng = np.random.default_rng(900 + nss)
    mpdu = tx.make_mpdu(700, rng)
    sig, m = tx.gen_frame(tx.F_VHT, 9, nss, mpdu)
    sig = sig * 12.0 * np.sqrt(m.nss)
    rx = awgn(H22 @ sig if m.nss == 2 else sig, 38.0, rng)
    pad = np.zeros((rx.shape[0], 50))
    rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
    r = oracle.rx_frame(rx[0], rx[1] if m.nss == 2 else None, 50 + 176, 0.0)
    assert r["status"] == 0 and r["frame"].m.nDBPS == (346, 693)[nss - 1]
    assert [p["mpdu"] for p in r["pdus"]] == [mpdu]


@pytest.mark.parametrize("fmt,mcs,nss", [(1, 3, 1), (2, 5, 1), (1, 11, 2)])
def test_short_gi_changes_the_stride_only(pkg, oracle, fmt, mcs, nss):
    """The short-GI bit of HT-SIG / VHT-SIG-A makes the receiver step 72 samples per data symbol
    (lib/cloud80211phy.cc:898-902, 1155-1159) and nothing else, so the payload of a real short-GI frame is lost."""
    tx = pkg.txgen
    rng = np.random.default_rng(950 + mcs)
    sig, m = tx.gen_frame(fmt, mcs, nss, tx.make_mpdu(150, rng), short_gi=True)
    assert sig.shape[1] == m.n_samples() - 8 * m.nsym
    sig = sig * 12.0 * np.sqrt(m.nss)
    rx = H22 @ sig if m.nss == 2 else sig
    pad = np.zeros((rx.shape[0], 50))
    rx = np.concatenate([pad, rx, pad], axis=1).astype(np.complex64)
    r = oracle.rx_frame(rx[0], rx[1] if m.nss == 2 else None, 50 + 176, 0.0)
    assert r["status"] == 0 and r["frame"].m.nSymSamp == 72 and r["frame"].m.nSym == m.nsym
    assert r["n_pdu"] == 0


def test_malformed_sig_fields_through_oracle(pkg, oracle):
    """What the receiver makes of SIG fields that break a rule: dropped at L-SIG (status 1), decoded as a legacy
    frame of the L-SIG length when HT-SIG / VHT-SIG-A fail their checks (format 0, no packet), dropped on an
    unsupported stream count or a bad VHT-SIG-B, and decoded normally when only an unchecked bit differs."""
    tx = pkg.txgen
    rng = np.random.default_rng(33)
    seen = []
    for fmt, mcs, edit in MALFORMED:
        sig, m = tx.gen_frame(fmt, mcs, 1, tx.make_mpdu(94, rng), sig_edit=edit)
        rx = np.concatenate([np.zeros(30), sig[0] * 12.0, np.zeros(400)]).astype(np.complex64)
        r = oracle.rx_frame(rx, None, 206, 0.0)
        seen.append((r["status"], r["frame"].m.format if r["status"] == 0 else -1, r["n_pdu"]))
    assert seen[:4] == [(1, -1, 0)] * 4
    assert seen[4:11] == [(0, 0, 0)] * 7 and seen[11] == (0, 1, 1)
    assert seen[12:17] == [(0, 0, 0)] * 5 and seen[17:19] == [(0, 2, 1)] * 2
    assert all(s[0] != 0 and s[2] == 0 for s in seen[19:26]) and seen[26] == (0, 1, 0)


def test_reference_bit_vectors(pkg):
    """The stages of the transmitter against the bit vectors in the reference's tools/txbits.py: scrambler (seed
    93) with the tail zeroed afterwards, convolutional encoder, HT BPSK interleaver, and the coded and interleaved
    VHT-SIG-B field."""
    import os
    tx = pkg.txgen
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_txbits.npz"))
    data = z["htMcs0DataBits"]
    assert np.array_equal(data, z["htMcs1DataBits"]) and np.array_equal(data, z["htMcs7DataBits"])
    scr = data ^ np.asarray(tx.scramble_seq(len(data), 93), dtype=np.uint8)
    scr[16 + 8 * 94:16 + 8 * 94 + 6] = 0
    coded = np.asarray(tx.bcc_encode(scr), dtype=np.uint8)
    assert np.array_equal(coded, z["htMcs0CodedBits"])
    m = np.asarray(tx.interleave_map(52, 1, 13))
    inter = np.zeros_like(coded)
    for s in range(len(coded) // 52):
        inter[52 * s + m] = coded[52 * s:52 * s + 52]
    assert np.array_equal(inter, z["htMcs0interBits"])
    sv = z["vhtDataBitsMcs0"] ^ np.asarray(tx.scramble_seq(len(z["vhtDataBitsMcs0"]), 93), dtype=np.uint8)
    assert np.array_equal(sv, z["vhtScrameBitsMcs0"])
    sb = np.asarray(tx.bcc_encode(z["vhtSigBmcs0"]), dtype=np.uint8)
    assert np.array_equal(sb, z["vhtSigBmcs0Bcc"])
    si = np.zeros(52, np.uint8)
    si[m] = sb
    assert np.array_equal(si, z["vhtSigBmcs0BccInter"])
