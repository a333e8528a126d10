"""ctypes front end of the CPU oracle (oracle/rx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

DROP_NAMES = {0: "ok", 1: "lsig", 2: "len_check", 3: "unsupported", 4: "decode_limit"}


class Mod(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "format", "ampdu", "nSym", "nSymSamp", "nSD", "nSS", "nLTF",
        "mcs", "len", "mod", "cr", "nBPSCS", "nDBPS", "nCBPS", "nCBPSS")]


class Frame(C.Structure):
    _fields_ = [
        ("status", C.c_int),
        ("lsig_mcs", C.c_int), ("lsig_len", C.c_int), ("nsamp", C.c_int),
        ("h_l", C.c_float * 128),
        ("lsig_bits", C.c_uint8 * 24),
        ("m", Mod),
        ("trellis", C.c_int), ("total", C.c_int),
        ("sssnr0", C.c_float), ("sssnr1", C.c_float),
        ("sig_bits", C.c_uint8 * 48),
        ("sigb_bits", C.c_uint8 * 26),
        ("mu2x1chan", C.c_float * 256),
        ("sumu", C.c_int),
        ("mu_pos", C.c_int),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("rx_oracle.c", "detect_oracle.c", "rx_oracle.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        fp = C.POINTER(C.c_float)
        u8 = C.POINTER(C.c_uint8)
        ip = C.POINTER(C.c_int)
        L.orc_init.restype = None
        L.orc_fft64.argtypes = [fp, fp]
        L.orc_signal.argtypes = [fp, C.c_int, C.c_float, C.POINTER(Frame)]
        L.orc_demod.argtypes = [fp, fp, C.c_int, C.c_float, C.POINTER(Frame), fp, C.c_int]
        L.orc_decode.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 u8, C.c_int, ip, u8]
        L.orc_rx_frame.argtypes = [fp, fp, C.c_int, C.c_float, C.POINTER(Frame), fp, C.c_int,
                                   u8, C.c_int, ip, ip]
        L.orc_rx_frame_mu.argtypes = [fp, fp, C.c_int, C.c_float, C.c_int, C.POINTER(Frame), fp, C.c_int,
                                      u8, C.c_int, ip, ip]
        L.orc_rx_frame_ex.argtypes = [fp, fp, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(Frame), fp, C.c_int,
                                      u8, C.c_int, ip, ip]
        L.orc_decode_ex.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    u8, C.c_int, ip, u8]
        L.orc_capture_points.argtypes = [fp, C.c_int]
        L.orc_capture_points.restype = None
        L.orc_sig_viterbi.argtypes = [fp, u8, C.c_int]
        L.orc_viterbi_bits.argtypes = [fp, C.c_int, C.c_int, C.c_int, u8]
        L.orc_get_table.argtypes = [C.c_char_p, ip, C.c_int, ip]
        L.orc_crc32.argtypes = [u8, C.c_int]
        L.orc_crc32.restype = C.c_uint32
        L.orc_bcc_encode.argtypes = [u8, u8, C.c_int]
        L.orc_crc8_bits.argtypes = [u8, C.c_int, u8]
        L.orc_init()
        _LIB = L
    return _LIB


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def table(name):
    out = np.zeros(1024, dtype=np.int32)
    n = C.c_int(0)
    lib().orc_get_table(name.encode(), out.ctypes.data_as(C.POINTER(C.c_int)), 1024, C.byref(n))
    return out[: n.value].copy()


def fft64(x):
    x = np.ascontiguousarray(x, dtype=np.complex64)
    y = np.zeros(64, dtype=np.complex64)
    lib().orc_fft64(_fp(x.view(np.float32)), _fp(y.view(np.float32)))
    return y


def sig_viterbi(llr, trellis):
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    bits = np.zeros(48, dtype=np.uint8)
    lib().orc_sig_viterbi(_fp(llr), _u8(bits), trellis)
    return bits[:trellis].copy()


def viterbi_bits(llr, cr, trellis):
    """Scrambled decoded bits (v_scramBits) for one frame's LLR stream."""
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    bits = np.zeros(max(trellis, 1), dtype=np.uint8)
    used = lib().orc_viterbi_bits(_fp(llr), len(llr), cr, trellis, _u8(bits))
    return bits[:trellis].copy(), used


def split_pdus(blob):
    """Split concatenated [format][len16 LE][MPDU][mcs] records."""
    out = []
    p = 0
    while p + 4 <= len(blob):
        ln = blob[p + 1] | (blob[p + 2] << 8)
        if blob[p] == 20:  # sounding channel message: no trailing mcs byte (decode_impl.cc:105-120)
            out.append(dict(format=20, mcs=None, mpdu=bytes(blob[p + 3:p + 3 + ln])))
            p += ln + 3
            continue
        out.append(dict(format=int(blob[p]), mcs=int(blob[p + 3 + ln]), mpdu=bytes(blob[p + 3:p + 3 + ln])))
        p += ln + 4
    return out


def decode(llr, format, length, cr, mcs, ampdu, trellis, want_bits=False, strict=False):
    llr = np.ascontiguousarray(llr, dtype=np.float32)
    out = np.zeros(8192, dtype=np.uint8)
    olen = C.c_int(0)
    bits = np.zeros(max(trellis, 1) + 8, dtype=np.uint8)
    n = lib().orc_decode_ex(_fp(llr), len(llr), format, length, cr, mcs, ampdu, trellis, 1 if strict else 0,
                            _u8(out), len(out), C.byref(olen), _u8(bits))
    res = dict(n_pdu=n, blob=bytes(out[: olen.value]), pdus=split_pdus(bytes(out[: olen.value])))
    if want_bits:
        res["bits"] = bits[:trellis].copy()
    return res


def rx_frame(iq0, iq1, sync_idx, rad=0.0, want_llr=True, mu_pos=0, strict=False, want_points=False):
    """Run signal -> demod/demod2 -> decode on one frame.

    iq0/iq1: complex64 capture(s) of each antenna (iq1 None for the SISO demod block);
    sync_idx: index of the sync byte (L-LTF start + 16).
    want_points: also return the equalised constellation points, [symbol, stream, nSD] complex64.
    """
    iq0 = np.ascontiguousarray(iq0, dtype=np.complex64)
    navail = len(iq0) - sync_idx
    p0 = _fp(iq0[sync_idx:].view(np.float32))
    p1 = None
    if iq1 is not None:
        iq1 = np.ascontiguousarray(iq1, dtype=np.complex64)
        navail = min(navail, len(iq1) - sync_idx)
        p1 = _fp(iq1[sync_idx:].view(np.float32))
    f = Frame()
    cap = 140000
    llr = np.zeros(cap, dtype=np.float32)
    out = np.zeros(8192, dtype=np.uint8)
    olen = C.c_int(0)
    npdu = C.c_int(0)
    pts = None
    if want_points:
        pts = np.zeros(cap // 2, dtype=np.complex64)
        lib().orc_capture_points(_fp(pts.view(np.float32)), len(pts))
    try:
        st = lib().orc_rx_frame_ex(p0, p1, navail, np.float32(rad), int(mu_pos), 1 if strict else 0, C.byref(f), _fp(llr),
                                   cap, _u8(out), len(out), C.byref(olen), C.byref(npdu))
        npts = lib().orc_captured_points() if want_points else 0
    finally:
        if want_points:
            lib().orc_capture_points(None, 0)
    blob = bytes(out[: olen.value])
    res = dict(status=st, frame=f, n_pdu=npdu.value, blob=blob, pdus=split_pdus(blob))
    if want_points and st == 0:
        res["points"] = pts[:npts].reshape(f.m.nSym, f.m.nSS, f.m.nSD).copy()
    if want_llr and st == 0:
        res["llr"] = llr[: f.total].copy()
    return res


def detect(iq, cap=None, want_stages=False):
    """presiso -> trigger -> sync over one capture (oracle/detect_oracle.c).
    Returns dict(pos, rad, snr, rssi[, ac, conj, trig, trig_idx, conj_idx])."""
    L = lib()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    n = len(iq)
    cap = cap or max(16, n // 400 + 16)
    ac = np.zeros(n, np.float32)
    conj = np.zeros(n, np.complex64)
    trig = np.zeros(n, np.uint8)
    pos = np.zeros(cap, np.int64)
    tidx = np.zeros(cap, np.int64)
    cidx = np.zeros(cap, np.int64)
    rad = np.zeros(cap, np.float32)
    snr = np.zeros(cap, np.float32)
    rssi = np.zeros(cap, np.float32)
    lp = C.POINTER(C.c_long)
    L.orc_presiso.argtypes = [C.POINTER(C.c_float), C.c_long, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_presiso.restype = None
    L.orc_trigger.argtypes = [C.POINTER(C.c_float), C.c_long, C.POINTER(C.c_uint8)]
    L.orc_trigger.restype = None
    L.orc_sync.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_long, lp,
                           C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), lp, lp, C.c_long]
    L.orc_sync.restype = C.c_long
    L.orc_presiso(_fp(iq.view(np.float32)), n, _fp(ac), _fp(conj.view(np.float32)))
    L.orc_trigger(_fp(ac), n, _u8(trig))
    cnt = L.orc_sync(_u8(trig), _fp(conj.view(np.float32)), _fp(iq.view(np.float32)), n,
                     pos.ctypes.data_as(lp), _fp(rad), _fp(snr), _fp(rssi), tidx.ctypes.data_as(lp),
                     cidx.ctypes.data_as(lp), cap)
    assert cnt <= cap, "raise cap"
    res = dict(pos=pos[:cnt].copy(), rad=rad[:cnt].copy(), snr=snr[:cnt].copy(), rssi=rssi[:cnt].copy())
    if want_stages:
        res.update(ac=ac, conj=conj, trig=trig, trig_idx=tidx[:cnt].copy(), conj_idx=cidx[:cnt].copy())
    return res
