"""Shared helpers of the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H22 = np.array([[1.0, 0.5j], [0.4, -1.0]], dtype=np.complex128)
LLR_RTOL = 1e-4   # relative tolerance on soft bits stated by the north star
LLR_ATOL = 1e-4   # absolute floor (LLR units: constellation spacing = 2) for values near zero


def golden_frames():
    out = []
    for fn in sorted(glob.glob(os.path.join(GOLDEN, "frame_*.npz"))):
        z = np.load(fn)
        out.append(dict(name=os.path.basename(fn)[6:-4], iq=z["iq"], sync_idx=int(z["sync_idx"]),
                        rad=float(z["rad"]), mpdu=z["mpdu"].tobytes(), fmt=int(z["fmt"]), mcs=int(z["mcs"]),
                        nss=int(z["nss"])))
    return out


def golden_soft(name):
    """What the reference's Python generator held after building golden frame `name` (tests/golden/make_golden.py
    make_soft): coded bits of the one encoder after puncturing [nSym * nCBPS], interleaved bits [nSS, nSym * nCBPSS],
    constellation points [nSym, nSS, nSD] (unit average power, tools/phy80211.py:796-805)."""
    z = np.load(os.path.join(GOLDEN, "soft_%s.npz" % name))
    nsym, nsd, nb = int(z["nsym"]), int(z["nsd"]), int(z["nbpscs"])
    nss = z["symbols"].shape[0]
    return dict(coded=np.unpackbits(z["coded"])[: int(z["n_coded"])],
                inted=np.unpackbits(z["inted"], axis=1)[:, : nsym * nsd * nb],
                points=z["symbols"].reshape(nss, nsym, nsd).transpose(1, 0, 2), nsym=nsym, nsd=nsd, nbpscs=nb, nss=nss)


QAM_SCALE = {1: 1.0, 2: 1.4142135623730951, 4: 3.1622776601683795, 6: 6.48074069840786, 8: 13.038404810405298}


def maxlog_llr(points, nbpscs):
    """Max-log soft bits of constellation points as the reference's procSymQamToLlr forms them
    (lib/cloud80211phy.cc:2090-2148), in float32: per point the I-axis bits, then the Q-axis bits; per axis
    x, E - |x|, E/2 - |E - |x||, ... with E = 2, 4, 8 for 2, 3, 4 bits.  Returns [..., nbpscs]."""
    q = np.asarray(points, np.complex64)
    if nbpscs == 1:
        return q.real.astype(np.float32)[..., None]
    q = q * np.float32(QAM_SCALE[nbpscs])
    nax = nbpscs // 2
    out = np.zeros(q.shape + (nbpscs,), np.float32)
    for a, v in enumerate((q.real.astype(np.float32), q.imag.astype(np.float32))):
        cur = v
        out[..., a * nax] = cur
        edge = np.float32(2 ** (nax - 1))
        for b in range(1, nax):
            cur = edge - np.abs(cur)
            out[..., a * nax + b] = cur
            edge = edge / np.float32(2)
    return out


def interleaver_position(k, ncbpss, nbpscs, legacy, iss=0):
    """Place of stream bit k of one OFDM symbol after the interleaver, from the standard's formulas (IEEE 802.11-2016
    17.3.5.7 for the 16-column legacy interleaver, 19.3.11.8.3 for HT/VHT at 20 MHz: 13 columns, 4 N_BPSCS rows,
    frequency rotation 11 N_BPSCS places per stream index step of the second stream)."""
    s = max(nbpscs // 2, 1)
    if legacy:
        i = (ncbpss // 16) * (k % 16) + k // 16
        return s * (i // s) + (i + ncbpss - (16 * i) // ncbpss) % s
    i = 4 * nbpscs * (k % 13) + k // 13
    j = s * (i // s) + (i + ncbpss - (13 * i) // ncbpss) % s
    if iss:
        j = (j - ((iss * 2) % 3 + 3 * (iss // 3)) * 11 * nbpscs) % ncbpss
    return j


def expected_llr_from_generator(soft, legacy):
    """The soft-bit stream `demod` hands to `decode` for a noise-free frame, worked out from the generator's
    constellation points alone: max-log soft bits per tone, taken out of the interleaver (and, for two streams, out
    of the stream parser: blocks of max(N_BPSCS / 2, 1) coded bits go to the streams in turn) with the standard's
    formulas.  Element c of symbol n is the soft bit of coded bit n * nCBPS + c."""
    nsym, nsd, nb, nss = soft["nsym"], soft["nsd"], soft["nbpscs"], soft["nss"]
    ncbpss = nsd * nb
    per_tone = maxlog_llr(soft["points"], nb).reshape(nsym, nss, ncbpss)   # interleaved order within a stream
    out = np.zeros((nsym, nss * ncbpss), np.float32)
    s = max(nb // 2, 1)
    k = np.arange(ncbpss)
    for iss in range(nss):
        r = np.array([interleaver_position(int(x), ncbpss, nb, legacy, iss) for x in k])
        stream = per_tone[:, iss, :][:, r]                                  # stream bit k sits at interleaved place r(k)
        c = iss * s + nss * s * (k // s) + k % s if nss > 1 else k          # coded bit the stream's k-th bit is
        out[:, c] = stream
    return out.reshape(-1)


def golden_ndp():
    z = np.load(os.path.join(GOLDEN, "ndp_vht_2sts.npz"))
    return dict(iq=z["iq"], sync_idx=int(z["sync_idx"]), rad=float(z["rad"]))


def golden_mu():
    z = np.load(os.path.join(GOLDEN, "mu_vht_2users.npz"))
    return dict(iq=z["iq"], sync_idx=int(z["sync_idx"]), mpdus=[z["mpdu0"].tobytes(), z["mpdu1"].tobytes()],
                mcs=[int(v) for v in z["mcs"]])


def golden_ampdu():
    z = np.load(os.path.join(GOLDEN, "ampdu_vht_3mpdu.npz"))
    raw, out, at = z["mpdus"].tobytes(), [], 0
    for n in z["lens"]:
        out.append(raw[at:at + int(n)])
        at += int(n)
    return dict(iq=z["iq"], sync_idx=int(z["sync_idx"]), mpdus=out)


def ref_fixture_mpdu():
    with open(os.path.join(GOLDEN, "ref_fixture_mpdu.hex")) as fh:
        return bytes.fromhex(fh.read().strip())


def ref_fixtures():
    """The reference's own IQ files: (name, [antenna arrays], sync_idx)."""
    s = np.fromfile(os.path.join(GOLDEN, "sig80211GenMultipleSiso_1x1_0.bin"), dtype=np.complex64)
    a = np.fromfile(os.path.join(GOLDEN, "sig80211GenMultipleMimo_2x2_0.bin"), dtype=np.complex64)
    b = np.fromfile(os.path.join(GOLDEN, "sig80211GenMultipleMimo_2x2_1.bin"), dtype=np.complex64)
    return [("ref_siso_l_mcs0", [s], 1200 + 176), ("ref_mimo_ht_mcs8", [a, b], 1200 + 176)]


def build_arena(frames, gap=64):
    """Lay frames (each: list of per-antenna complex arrays + sync index) into one arena per antenna.
    Returns iq0, iq1 (or None), sync_off[], n_avail[]."""
    nant = max(len(f[0]) for f in frames)
    total = sum(len(f[0][0]) + gap for f in frames) + gap
    iq = np.zeros((nant, total), dtype=np.complex64)
    sync, navail = [], []
    pos = gap
    for ants, sidx in frames:
        n = len(ants[0])
        for a in range(nant):
            if a < len(ants):
                iq[a, pos:pos + n] = ants[a]
        sync.append(pos + sidx)
        navail.append(n - sidx)
        pos += n + gap
    return iq[0], (iq[1] if nant > 1 else None), np.array(sync, dtype=np.uint64), np.array(navail, dtype=np.uint32)


def awgn(sig, snr_db, rng):
    p = np.mean(np.abs(sig) ** 2)
    sigma = np.sqrt(p / (10 ** (snr_db / 10.0)) / 2.0)
    return sig + sigma * (rng.standard_normal(sig.shape) + 1j * rng.standard_normal(sig.shape))


def unpack_bits(bytes_row, nbits):
    return np.unpackbits(np.asarray(bytes_row, dtype=np.uint8), bitorder="little")[:nbits]


ST_DROP_LSIG = 1


def blob_of(p):
    n = len(p["mpdu"])
    tail = b"" if p["mcs"] is None else bytes([p["mcs"]])
    return bytes([p["format"], n % 256, n // 256]) + p["mpdu"] + tail


def model_chain(oracle, iq0, iq1, triggers):
    """What signal -> demod -> decode publish for a stream with tagged sync bytes at `triggers`."""
    out, cursor, n_frames = [], 0, 0
    model_chain.dealt_with = 0
    for pos, rad in sorted(triggers):
        if pos < cursor:
            continue  # inside the previous frame: the trigger search has not resumed yet
        if len(iq0) - pos < 224:
            break
        r = oracle.rx_frame(iq0, iq1, pos, rad, want_llr=False)
        model_chain.dealt_with += 1
        if r["status"] == ST_DROP_LSIG:
            cursor = pos + 80
            continue
        span = 224 + max(int(r["frame"].nsamp), 0)
        end = min(len(iq0), pos + span)
        r = oracle.rx_frame(iq0[:end], None if iq1 is None else iq1[:end], pos, rad, want_llr=False)
        out += [blob_of(p) for p in r["pdus"]]
        n_frames += 1
        cursor = pos + span
    return out, n_frames


def flip(name, *idx):
    def edit(which, bits):
        if which == name:
            bits = list(bits)
            for i in idx:
                bits[i] ^= 1
        return bits
    return edit


# (format, mcs, SIG edit): what signalCheckLegacy / signalCheckHt / signalCheckVhtA and the parsers reject or
# let through (lib/cloud80211phy.cc:650-771, 852-1223)
MALFORMED = [
    (0, 3, flip("lsig", 4)),            # L-SIG reserved bit
    (0, 3, flip("lsig_crc", 17)),       # L-SIG parity
    (0, 3, flip("lsig", 3)),            # rate field without its top bit
    (0, 0, flip("lsig", 5, 6, 9, 11)),  # length 94 -> 13, below the shortest frame (four flips: parity unchanged)
    (1, 4, flip("htsig", 26)),          # HT-SIG reserved one
    (1, 4, flip("htsig_crc", 36)),      # HT-SIG CRC
    (1, 4, flip("htsig", 7)),           # 40 MHz
    (1, 4, flip("htsig", 28)),          # STBC
    (1, 4, flip("htsig", 30)),          # LDPC
    (1, 4, flip("htsig", 32)),          # extension spatial streams
    (1, 4, flip("htsig", 5)),           # MCS 36
    (1, 4, flip("htsig", 24)),          # smoothing: not looked at
    (2, 4, flip("vhta", 2)),            # VHT-SIG-A reserved ones
    (2, 4, flip("vhta", 23)),
    (2, 4, flip("vhta", 33)),
    (2, 4, flip("vhta", 0)),            # 40 MHz
    (2, 4, flip("vhta_crc", 40)),       # VHT-SIG-A CRC
    (2, 4, flip("vhta", 3)),            # STBC: not looked at
    (2, 4, flip("vhta", 26)),           # LDPC: not looked at
    (2, 4, flip("vhta", 29, 31)),       # MCS 4 -> 14
    (2, 4, flip("vhta", 11)),           # three space-time streams
    (2, 4, flip("vhta", 5)),            # group id 2: a multi-user frame
    (2, 4, flip("vhtb", 18)),           # VHT-SIG-B reserved ones broken, not the NDP pattern either
    (1, 4, flip("htsig", 23)),          # HT length 94 + 32768: beyond 4095
    (1, 4, flip("htsig", 16)),          # HT length 94 + 256: more symbols than the L-SIG duration covers
    (2, 4, flip("vhtb", 8)),            # VHT-SIG-B length + 1024 bytes: the same
    (1, 4, flip("htsig", 9)),           # HT length 92: decoded two bytes short, the FCS fails
]
