"""Synthetic 802.11a/n/ac 20 MHz frame generator (numpy) used to build test and benchmark corpora.

Written from IEEE 802.11-2016 (clauses 17, 19, 21).  It produces the same waveform as the
reference's Python generator (tools/phy80211.py genFromMpdu/genFromAmpdu + genFinalSig): same
field order, tone scaling, cyclic shifts and the reference's symbol-boundary convention (the two
samples either side of every symbol boundary are halved, tools/phy80211header.py:910-915);
tests/test_txgen.py checks this sample by sample on the golden frames.
"""
import struct
import zlib

import numpy as np

F_L, F_HT, F_VHT = 0, 1, 2

_LTF_L = [1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1]
_LTF_R = [1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1]
LTF_L_53 = np.array(_LTF_L + [0] + _LTF_R, dtype=np.float64)              # tones -26..26
LTF_NL_57 = np.array([1, 1] + _LTF_L + [0] + _LTF_R + [-1, -1], dtype=np.float64)  # tones -28..28
_s = np.sqrt(0.5) * (1 + 1j)
STF_53 = np.zeros(53, dtype=np.complex128)                                   # tones -26..26
for _k, _v in zip(range(-24, 25, 4), [1, -1, 1, -1, -1, 1, 0, -1, -1, 1, 1, 1, 1]):
    STF_53[_k + 26] = _v * _s

CSD_L = {1: [0], 2: [0, -200]}     # ns, legacy part (L-STF .. HT-SIG / VHT-SIG-A)
CSD_NL = {1: [0], 2: [0, -400]}    # ns, HT/VHT part
P_LTF = [[1, -1, 1, 1], [1, 1, -1, 1]]
R_LTF = [1, -1, 1, 1]
PILOT_HT = {1: [[1, 1, 1, -1]], 2: [[1, 1, -1, -1], [1, -1, -1, 1]]}
PILOT_BASE = [1, 1, 1, -1]


def pilot_polarity():
    st, out = 0x7F, []
    for _ in range(127):
        fb = ((st >> 6) ^ (st >> 3)) & 1
        out.append(-1.0 if fb else 1.0)
        st = ((st << 1) | fb) & 0x7F
    return np.array(out)


PILOT_PS = pilot_polarity()

# ----------------------------------------------------------------------------- rate tables
# (modulation bits per subcarrier, coding rate index 0:1/2 1:2/3 2:3/4 3:5/6)
L_RATES = [(1, 0), (1, 2), (2, 0), (2, 2), (4, 0), (4, 2), (6, 1), (6, 2)]
L_RATE_BITS = [[1, 1, 0, 1], [1, 1, 1, 1], [0, 1, 0, 1], [0, 1, 1, 1], [1, 0, 0, 1], [1, 0, 1, 1], [0, 0, 0, 1],
               [0, 0, 1, 1]]
NL_RATES = [(1, 0), (2, 0), (2, 2), (4, 0), (4, 2), (6, 1), (6, 2), (6, 3), (8, 2), (8, 3)]
CR_NUM = [(1, 2), (2, 3), (3, 4), (5, 6)]


class Mod:
    """Modulation / length bookkeeping for one frame."""

    def __init__(self, fmt, mcs, nss=1):
        self.fmt, self.mcs, self.nss = fmt, mcs, nss
        if fmt == F_L:
            self.nbpscs, self.cr = L_RATES[mcs]
            self.nsd = 48
            self.nss = 1
        elif fmt == F_HT:
            self.nbpscs, self.cr = NL_RATES[mcs % 8]
            self.nss = mcs // 8 + 1
            self.nsd = 52
        else:
            self.nbpscs, self.cr = NL_RATES[mcs]
            self.nsd = 52
        self.ncbpss = self.nbpscs * self.nsd
        self.ncbps = self.ncbpss * self.nss
        num, den = CR_NUM[self.cr]
        self.ndbps = self.ncbps * num // den
        self.nltf = 0 if fmt == F_L else (1 if self.nss == 1 else 2)

    def set_len(self, nbytes):
        """nbytes = MPDU length (L/HT) or A-MPDU length (VHT)."""
        if self.fmt == F_VHT:
            self.apep = nbytes
            self.nsym = -(-(8 * nbytes + 16 + 6) // self.ndbps)
            self.psdu = (self.nsym * self.ndbps - 16 - 6) // 8
            self.npad_eof = (self.psdu - nbytes) // 4
            self.npad_oct = self.psdu - nbytes - 4 * self.npad_eof
            self.npad_bits = self.nsym * self.ndbps - 8 * self.psdu - 16 - 6
            tx = 20 + 8 + 4 + 4 * self.nltf + 4 + 4 * self.nsym
            self.lsig_len = (tx - 20) // 4 * 3 - 3
        else:
            self.psdu = nbytes
            self.nsym = -(-(8 * nbytes + 16 + 6) // self.ndbps)
            self.npad_bits = self.nsym * self.ndbps - 8 * nbytes - 16 - 6
            if self.fmt == F_L:
                self.lsig_len = nbytes
            else:
                tx = 20 + 8 + 4 + 4 * self.nltf + 4 * self.nsym
                self.lsig_len = (tx - 20) // 4 * 3 - 3
        return self

    def n_samples(self):
        pre = 400 if self.fmt == F_L else 400 + 160 + 80 + 80 * self.nltf + (80 if self.fmt == F_VHT else 0)
        return pre + 80 * self.nsym


# ----------------------------------------------------------------------------- bit-level pieces
def crc8_bits(bits):
    """CRC-8 of the SIG fields and A-MPDU delimiters: x^8 + x^2 + x + 1, register preset to ones, bits fed most
    significant first, the ones' complement sent (IEEE 802.11-2016 19.3.9.4.4)."""
    c = 0xFF
    for b in bits:
        c = ((c << 1) & 0xFF) ^ (0x07 if ((c >> 7) ^ (1 if b else 0)) & 1 else 0)
    c = 0xFF - c
    return [(c >> (7 - i)) & 1 for i in range(8)]


def scramble_seq(n, seed):
    """Scrambler x^7+x^4+1 output sequence for the given 7-bit initial state."""
    out = np.zeros(n, dtype=np.uint8)
    st = seed
    for i in range(min(n, 127)):
        fb = ((st >> 6) ^ (st >> 3)) & 1
        out[i] = fb
        st = ((st << 1) & 0x7E) | fb
    if n > 127:
        reps = -(-n // 127)
        out = np.tile(out[:127], reps)[:n]
    return out


_G0 = np.array([(0o155 >> j) & 1 for j in range(7)], dtype=np.uint8)  # taps on (newest .. oldest)
_G1 = np.array([(0o117 >> j) & 1 for j in range(7)], dtype=np.uint8)


def bcc_encode(bits):
    """Rate-1/2 K=7 convolutional code (133,171), zero initial state; returns [A0,B0,A1,B1,...]."""
    b = np.concatenate([np.zeros(6, dtype=np.uint8), np.asarray(bits, dtype=np.uint8)])
    n = len(bits)
    a = np.zeros(n, dtype=np.uint8)
    c = np.zeros(n, dtype=np.uint8)
    for j in range(7):
        seg = b[6 - j: 6 - j + n]
        if _G0[j]:
            a ^= seg
        if _G1[j]:
            c ^= seg
    out = np.empty(2 * n, dtype=np.uint8)
    out[0::2] = a
    out[1::2] = c
    return out


PUNCT = {0: [1, 1], 1: [1, 1, 1, 0], 2: [1, 1, 1, 0, 0, 1], 3: [1, 1, 1, 0, 0, 1, 1, 0, 0, 1]}


def puncture(coded, cr):
    pat = np.array(PUNCT[cr], dtype=bool)
    reps = -(-len(coded) // len(pat))
    keep = np.tile(pat, reps)[: len(coded)]
    return coded[keep]


def interleave_map(ncbpss, nbpscs, ncol, iss=0, nss=1):
    """position r of coded bit k within one symbol (first, second and third permutation)."""
    nrow = ncbpss // ncol
    s = max(nbpscs // 2, 1)
    k = np.arange(ncbpss)
    i = nrow * (k % ncol) + k // ncol
    j = s * (i // s) + (i + ncbpss - (ncol * i) // ncbpss) % s
    if nss >= 2:
        j = (j - ((iss * 2) % 3 + 3 * (iss // 3)) * 11 * nbpscs) % ncbpss
    return j


def _gray_axis(nb):
    """axis level for each LSB-first chip value (IEEE 802.11-2016 tables 17-10..17-16, 21-xx)."""
    lv = np.zeros(1 << nb)
    for v in range(1 << nb):
        bits = [(v >> i) & 1 for i in range(nb)]
        # b0 selects the sign, the remaining bits walk a reflected Gray code towards the edge
        g = 0
        acc = 0
        for bb in bits:
            acc ^= bb
            g = (g << 1) | acc
        # g counts 0 .. 2^nb-1 from the most negative level
        lv[v] = 2 * g - ((1 << nb) - 1)
    return lv


def qam_map(chips, nbpscs):
    """chips: int array of LSB-first groups of nbpscs interleaved bits -> constellation points."""
    if nbpscs == 1:
        return (2.0 * chips - 1.0).astype(np.complex128)
    nb = nbpscs // 2
    ax = _gray_axis(nb)
    norm = {1: np.sqrt(2.0), 2: np.sqrt(10.0), 3: np.sqrt(42.0), 4: np.sqrt(170.0)}[nb]
    i = ax[chips & ((1 << nb) - 1)]
    q = ax[chips >> nb]
    return (i + 1j * q) / norm


# ----------------------------------------------------------------------------- OFDM helpers
def _to_time(tones, kmin, ntf, nss, cs_ns):
    """tones on subcarriers kmin.. -> 64 time samples, tone scaling and cyclic shift applied."""
    k = np.arange(kmin, kmin + len(tones))
    x = np.zeros(64, dtype=np.complex128)
    ph = np.exp(-1j * 2 * np.pi * cs_ns * 20 * 0.001 * k / 64.0)
    x[k % 64] = np.asarray(tones, dtype=np.complex128) * ph
    return np.fft.ifft(x) / np.sqrt(ntf * nss)


def _gi(sym):
    return np.concatenate([sym[48:], sym])


def _insert_pilots_dc(data, pil, nsd):
    """data in ascending tone order + 4 pilots (tones -21,-7,7,21) + DC null."""
    if nsd == 48:
        parts = [data[0:5], [pil[0]], data[5:18], [pil[1]], data[18:24], [0], data[24:30], [pil[2]], data[30:43],
                 [pil[3]], data[43:48]]
    else:
        parts = [data[0:7], [pil[0]], data[7:20], [pil[1]], data[20:26], [0], data[26:32], [pil[2]], data[32:45],
                 [pil[3]], data[45:52]]
    return np.concatenate([np.asarray(p, dtype=np.complex128) for p in parts])


def _sig_symbol(coded48, qbpsk, nss, iss):
    """One BPSK / QBPSK SIG symbol (48 coded bits, legacy interleaver, legacy tone plan)."""
    il = np.zeros(48, dtype=np.uint8)
    il[interleave_map(48, 1, 16)] = coded48
    q = (2.0 * il - 1.0).astype(np.complex128)
    if qbpsk:
        q = q * 1j
    tones = _insert_pilots_dc(q, PILOT_BASE, 48)
    return _gi(_to_time(tones, -26, 52, nss, CSD_L[nss][iss]))


# ----------------------------------------------------------------------------- frame generator
def make_mpdu(nbytes, rng):
    body = rng.integers(0, 256, nbytes - 4, dtype=np.uint8).tobytes()
    return body + struct.pack("<L", zlib.crc32(body) & 0xFFFFFFFF)


def ampdu_vht(mpdus, eof=None):
    """A-MPDU subframes: 4-byte delimiter (EOF, rsvd, len[13:12], len[11:0], CRC-8, 0x4E) + MPDU + pad to 4.
    eof: the EOF bit of every delimiter (default: set for a single-MPDU A-MPDU as VHT requires; HT delimiters
    have it reserved = 0)."""
    out = b""
    for m in mpdus:
        ln = len(m)
        lb = [(ln >> i) & 1 for i in range(14)]
        bits = [(1 if len(mpdus) == 1 else 0) if eof is None else eof, 0] + lb[12:14] + lb[0:12]
        bits = bits + crc8_bits(bits) + [(0x4E >> i) & 1 for i in range(8)]
        by = bytes(sum(bits[i * 8 + j] << j for j in range(8)) for i in range(4))
        pkt = by + m
        pkt += b"\x00" * (-len(pkt) % 4)
        out += pkt
    return out


def gen_frame(fmt, mcs, nss, mpdu, scrambler=93, n_mpdu_list=None, ht_aggregate=False, short_gi=False, sig_edit=None):
    """Baseband frame(s): returns (array [nss][N] complex128, Mod).  For VHT `mpdu` is wrapped into a
    single-subframe A-MPDU (pass n_mpdu_list=[m0, m1, ...] for several subframes).  ht_aggregate: HT PSDU =
    A-MPDU of n_mpdu_list (or [mpdu]), HT-SIG aggregation bit set.  short_gi (HT / VHT): the SIG field's
    short-GI bit is set and the data symbols carry an 8-sample guard interval (72 samples each); the L-SIG
    length stays the long-GI one, which covers the shorter frame.  sig_edit(name, bits) -> bits, for malformed
    frames: called with the bit list of every SIG field before its parity / CRC is computed ("lsig", "htsig",
    "vhta", "vhtb") and again with the protected field (name + "_crc"); whatever it returns is sent."""
    edit = sig_edit if sig_edit is not None else (lambda name, bits: bits)
    m = Mod(fmt, mcs, nss)
    nss = m.nss
    if fmt == F_VHT:
        payload = ampdu_vht(n_mpdu_list if n_mpdu_list is not None else [mpdu])
    elif fmt == F_HT and ht_aggregate:
        payload = ampdu_vht(n_mpdu_list if n_mpdu_list is not None else [mpdu], eof=0)
    else:
        payload = mpdu
    m.set_len(len(payload))

    segs = [[] for _ in range(nss)]  # per stream: list of symbol-level segments
    # ---- legacy preamble
    for iss in range(nss):
        cs = CSD_L[nss][iss]
        stf = _to_time(STF_53, -26, 12, nss, cs)
        ltf = _to_time(LTF_L_53, -26, 52, nss, cs)
        segs[iss].append(np.concatenate([stf[32:], stf, stf]))
        segs[iss].append(np.concatenate([ltf[32:], ltf, ltf]))
    # ---- L-SIG
    rate = L_RATE_BITS[mcs] if fmt == F_L else L_RATE_BITS[0]
    hb = edit("lsig", list(rate) + [0] + [(m.lsig_len >> i) & 1 for i in range(12)])
    hb = edit("lsig_crc", hb + [sum(hb) % 2] + [0] * 6)
    lsig = bcc_encode(hb)
    for iss in range(nss):
        segs[iss].append(_sig_symbol(lsig, False, nss, iss))
    sigb_crc = [0] * 8
    if fmt == F_HT:
        b = [(mcs >> i) & 1 for i in range(7)] + [0] + [(m.psdu >> i) & 1 for i in range(16)]
        b += [1, 1, 1, 1 if ht_aggregate else 0, 0, 0, 0, 1 if short_gi else 0, 0, 0]   # smoothing, not sounding, reserved, aggregation, stbc x2, bcc, long GI, ess x2
        b = edit("htsig", b)
        b = edit("htsig_crc", b + crc8_bits(b) + [0] * 6)
        c = bcc_encode(b)
        for iss in range(nss):
            segs[iss].append(_sig_symbol(c[:48], True, nss, iss))
            segs[iss].append(_sig_symbol(c[48:], True, nss, iss))
    elif fmt == F_VHT:
        b = [0, 0, 1, 0] + [0] * 6 + [((nss - 1) >> i) & 1 for i in range(3)] + [0] * 9 + [0, 1]
        b += [1 if short_gi else 0, 0, 0, 0] + [(mcs >> i) & 1 for i in range(4)] + [0, 1]
        b = edit("vhta", b)
        b = edit("vhta_crc", b + crc8_bits(b) + [0] * 6)
        c = bcc_encode(b)
        for iss in range(nss):
            segs[iss].append(_sig_symbol(c[:48], False, nss, iss))
            segs[iss].append(_sig_symbol(c[48:], True, nss, iss))
    # ---- HT/VHT training
    if fmt != F_L:
        stf57 = np.concatenate([[0, 0], STF_53, [0, 0]])
        for iss in range(nss):
            cs = CSD_NL[nss][iss]
            segs[iss].append(_gi(_to_time(stf57, -28, 12, nss, cs)))
            for il in range(m.nltf):
                t = LTF_NL_57 * P_LTF[iss][il]
                if fmt == F_VHT:
                    for k in (-21, -7, 7, 21):
                        t[k + 28] = LTF_NL_57[k + 28] * R_LTF[il]
                segs[iss].append(_gi(_to_time(t, -28, 56, nss, cs)))
    # ---- VHT-SIG-B
    if fmt == F_VHT:
        ln = -(-m.apep // 4)
        b = edit("vhtb", [(ln >> i) & 1 for i in range(17)] + [1, 1, 1])
        sigb_crc = crc8_bits(b)
        c = bcc_encode(b + [0] * 6)
        il = np.zeros(52, dtype=np.uint8)
        il[interleave_map(52, 1, 13)] = c
        q = (2.0 * il - 1.0).astype(np.complex128)
        tones = _insert_pilots_dc(q, PILOT_BASE, 52)
        for iss in range(nss):
            segs[iss].append(_gi(_to_time(tones, -28, 56, nss, CSD_NL[nss][iss])))
    # ---- data bits
    pbits = np.unpackbits(np.frombuffer(payload, dtype=np.uint8), bitorder="little")
    if fmt == F_VHT:
        eof = [1] + [0] * 15 + crc8_bits([1] + [0] * 15) + [(0x4E >> i) & 1 for i in range(8)]
        psdu_bits = np.concatenate([pbits, np.tile(np.array(eof, dtype=np.uint8), m.npad_eof),
                                    np.zeros(8 * m.npad_oct, dtype=np.uint8)])
        data = np.concatenate([np.zeros(8, dtype=np.uint8), np.array(sigb_crc, dtype=np.uint8), psdu_bits,
                               np.zeros(m.npad_bits, dtype=np.uint8)])
        scr = data ^ scramble_seq(len(data), scrambler)
        scr = np.concatenate([scr, np.zeros(6, dtype=np.uint8)])
    else:
        data = np.concatenate([np.zeros(16, dtype=np.uint8), pbits, np.zeros(6 + m.npad_bits, dtype=np.uint8)])
        scr = data ^ scramble_seq(len(data), scrambler)
        scr[16 + 8 * m.psdu: 16 + 8 * m.psdu + 6] = 0
    assert len(scr) == m.nsym * m.ndbps
    coded = puncture(bcc_encode(scr), m.cr)
    if fmt == F_VHT and mcs == 9 and m.ndbps * 6 != m.ncbps * 5:
        # VHT MCS9 at 20 MHz with one or two streams is not a combination of the standard (N_DBPS is not an
        # integer); the reference receiver takes it with N_DBPS rounded down (lib/cloud80211phy.cc:1287-1303)
        # and reads the coded bits as one continuous stream, so the symbols' spare bits all come at the end
        coded = np.concatenate([coded, np.zeros(m.nsym * m.ncbps - len(coded), dtype=np.uint8)])
    assert len(coded) == m.nsym * m.ncbps
    # ---- stream parse, interleave, map, OFDM
    s = max(m.nbpscs // 2, 1)
    ncol = 16 if fmt == F_L else 13
    ntf = 52 if fmt == F_L else 56
    coded = coded.reshape(m.nsym, m.ncbps)
    for iss in range(nss):
        if nss == 1:
            st_bits = coded
        else:
            blk = coded.reshape(m.nsym, m.ncbpss // s, nss, s)
            st_bits = blk[:, :, iss, :].reshape(m.nsym, m.ncbpss)
        imap = interleave_map(m.ncbpss, m.nbpscs, ncol, iss, nss)
        il = np.zeros_like(st_bits)
        il[:, imap] = st_bits
        chips = (il.reshape(m.nsym, m.nsd, m.nbpscs) << np.arange(m.nbpscs)).sum(axis=2)
        pts = qam_map(chips, m.nbpscs)
        if fmt == F_L:
            pil0, pidx, cs, rot = PILOT_BASE, 1, 0, False
        elif fmt == F_HT:
            pil0, pidx, cs, rot = PILOT_HT[nss][iss], 3, CSD_NL[nss][iss], True
        else:
            pil0, pidx, cs, rot = PILOT_BASE, 4, CSD_NL[nss][iss], True
        for n in range(m.nsym):
            pil = [pil0[(k + n) % 4] if rot else pil0[k] for k in range(4)]
            pil = [p * PILOT_PS[(pidx + n) % 127] for p in pil]
            tones = _insert_pilots_dc(pts[n], pil, m.nsd)
            body = _to_time(tones, -26 if fmt == F_L else -28, ntf, nss, cs)
            segs[iss].append(np.concatenate([body[56:], body]) if short_gi and fmt != F_L else _gi(body))
    # ---- join: halve the two samples either side of every boundary
    out = []
    for iss in range(nss):
        ss = [np.array(x, dtype=np.complex128) for x in segs[iss]]
        for i in range(len(ss) - 1):
            ss[i][-1] *= 0.5
            ss[i + 1][0] *= 0.5
        out.append(np.concatenate(ss))
    return np.array(out), m
