#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ with the reference's Python tool box.

Runs only in the build container (needs /root/reference/tools).  The reference is used as
TX truth (tools/phy80211.py waveform generator) and as Viterbi truth
(tools/phy80211header.py:998-1063 procViterbiDecoder on integer LLRs, where float64 and
float32 arithmetic coincide, so tie-breaking is pinned exactly).  Outputs are data only:
IQ samples, transmitted MPDUs, LLR/bit vectors.

Also copies the reference's own IQ fixtures (tools/sig80211GenMultiple*.bin, data files)
and records the expected MPDU from tools/debug.txt:7.
"""
import contextlib
import io
import os
import shutil
import struct
import sys
import zlib

import numpy as np

REF_TOOLS = "/root/reference/tools"
HERE = os.path.dirname(os.path.abspath(__file__))

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
sys.path.insert(0, REF_TOOLS)
_cwd = os.getcwd()
os.chdir(REF_TOOLS)
with contextlib.redirect_stdout(io.StringIO()):
    import mac80211  # noqa: E402
    import phy80211  # noqa: E402
    import phy80211header as p8h  # noqa: E402
os.chdir(_cwd)

GAP = 400


def mpdu_of(n, seed):
    rng = np.random.RandomState(seed)
    body = bytes(rng.randint(0, 256, n - 4, dtype=np.uint8))
    return body + struct.pack("<L", zlib.crc32(body) & 0xFFFFFFFF)


def gen(fmt, mcs, nsts, mpdu, cfo_hz=0.0, keep=None):
    """keep: a dict that receives what the generator holds after building the frame -- the coded (punctured) bits of
    its one encoder, the interleaved bits and the constellation points of every spatial stream
    (phy80211.esCodedBits / ssIntedBits / ssSymbols, tools/phy80211.py:51-54,716-805) -- as arrays."""
    phy = phy80211.phy80211(ifDebug=False)
    with contextlib.redirect_stdout(io.StringIO()):
        if fmt == "L":
            phy.genFromMpdu(mpdu, p8h.modulation(p8h.F.L, mcs, p8h.BW.BW20, 1, False))
        elif fmt == "HT":
            phy.genFromMpdu(mpdu, p8h.modulation(p8h.F.HT, mcs, p8h.BW.BW20, nsts, False))
        else:
            ampdu = mac80211.genAmpduVHT([mpdu])
            phy.genFromAmpdu(ampdu, p8h.modulation(p8h.F.VHT, mcs, p8h.BW.BW20, nsts, False),
                             vhtPartialAid=0, vhtGroupId=0)
        ss = phy.genFinalSig(multiplier=12.0 * np.sqrt(nsts), cfoHz=cfo_hz, num=1, gap=True, gapLen=GAP)
    if keep is not None:
        assert len(phy.esCodedBits) == 1
        keep["coded"] = np.array(phy.esCodedBits[0], dtype=np.uint8)
        keep["inted"] = np.array(phy.ssIntedBits, dtype=np.uint8)
        keep["symbols"] = np.array(phy.ssSymbols, dtype=np.complex128)
        keep["nsym"], keep["nsd"], keep["nbpscs"] = int(phy.m.nSym), int(phy.m.nSD), int(phy.m.nBPSCS)
    return np.array(ss, dtype=np.complex128)


# Fixed 2x2 channel used to turn two TX streams into two RX antennas (well conditioned).
H22 = np.array([[1.0, 0.5j], [0.4, -1.0]], dtype=np.complex128)

CASES = [
    # name, fmt, mcs, nsts, mpdu bytes, cfo Hz
    ("l_mcs0_100", "L", 0, 1, 100, 0.0),
    ("l_mcs4_120", "L", 4, 1, 120, 0.0),
    ("l_mcs7_300", "L", 7, 1, 300, 0.0),
    ("l_mcs3_64_cfo", "L", 3, 1, 64, 100e3),
    ("ht_mcs0_100", "HT", 0, 1, 100, 0.0),
    ("ht_mcs5_200", "HT", 5, 1, 200, 0.0),
    ("ht_mcs7_300", "HT", 7, 1, 300, 0.0),
    ("ht_mcs8_100_2x2", "HT", 8, 2, 100, 0.0),
    ("ht_mcs15_300_2x2", "HT", 15, 2, 300, 0.0),
    ("vht_mcs0_100", "VHT", 0, 1, 100, 0.0),
    ("vht_mcs4_200", "VHT", 4, 1, 200, 0.0),
    ("vht_mcs7_250", "VHT", 7, 1, 250, -50e3),
    ("vht_mcs8_300", "VHT", 8, 1, 300, 0.0),
    ("vht_mcs8_1496", "VHT", 8, 1, 1496, 0.0),
    ("vht_mcs1_150_2x2", "VHT", 1, 2, 150, 0.0),
    ("vht_mcs8_300_2x2", "VHT", 8, 2, 300, 0.0),
]


def make_frames():
    for idx, (name, fmt, mcs, nsts, nbytes, cfo) in enumerate(CASES):
        mpdu = mpdu_of(nbytes, 1000 + idx)
        tx = gen(fmt, mcs, nsts, mpdu, cfo)            # [nsts][N]
        rx = (H22 @ tx) if nsts == 2 else tx
        np.savez_compressed(
            os.path.join(HERE, "frame_%s.npz" % name),
            iq=rx.astype(np.complex64),
            sync_idx=np.int32(GAP + 160 + 16),
            rad=np.float32(-2.0 * np.pi * cfo / 20e6),
            mpdu=np.frombuffer(mpdu, dtype=np.uint8),
            fmt=np.int32({"L": 0, "HT": 1, "VHT": 2}[fmt]), mcs=np.int32(mcs), nss=np.int32(nsts))
        print("frame", name, rx.shape)


def make_soft():
    """Companions of the frame_*.npz fixtures (same frames, same seeds): the generator's own coded bits, interleaved
    bits and constellation points, which pin the receiver's equalised points and soft bits -- values, order and
    scale -- to data the reference holds.  Bits are stored packed (np.packbits, big-endian bit order)."""
    for idx, (name, fmt, mcs, nsts, nbytes, cfo) in enumerate(CASES):
        keep = {}
        tx = gen(fmt, mcs, nsts, mpdu_of(nbytes, 1000 + idx), cfo, keep=keep)
        old = np.load(os.path.join(HERE, "frame_%s.npz" % name))
        rx = (H22 @ tx) if nsts == 2 else tx
        assert np.array_equal(old["iq"], rx.astype(np.complex64)), name   # the very frame the IQ fixture holds
        np.savez_compressed(
            os.path.join(HERE, "soft_%s.npz" % name),
            coded=np.packbits(keep["coded"]), n_coded=np.int32(len(keep["coded"])),
            inted=np.packbits(keep["inted"], axis=1), symbols=keep["symbols"].astype(np.complex64),
            nsym=np.int32(keep["nsym"]), nsd=np.int32(keep["nsd"]), nbpscs=np.int32(keep["nbpscs"]))
        print("soft", name, keep["symbols"].shape, len(keep["coded"]))


def make_ndp():
    """VHT sounding NDP (two space-time streams, no data field) as one RX antenna sees it through H22's first
    row.  The expected message is derived in the tests from the IQ itself (the channel message carries raw
    VHT-LTF window samples), so the fixture is the waveform only."""
    phy = phy80211.phy80211(ifDebug=False)
    with contextlib.redirect_stdout(io.StringIO()):
        phy.genFromAmpdu(b"", p8h.modulation(p8h.F.VHT, 0, p8h.BW.BW20, 2, False), vhtPartialAid=0, vhtGroupId=0)
        ss = phy.genFinalSig(multiplier=12.0 * np.sqrt(2), cfoHz=0.0, num=1, gap=True, gapLen=GAP)
    tx = np.array(ss, dtype=np.complex128)
    np.savez_compressed(
        os.path.join(HERE, "ndp_vht_2sts.npz"),
        iq=(H22 @ tx).astype(np.complex64),
        sync_idx=np.int32(GAP + 160 + 16), rad=np.float32(0.0))
    print("ndp", tx.shape)


def make_mu():
    """Two-user VHT MU-MIMO frame (group id 2, one space-time stream per user, different MCS), precoded with
    the zero-forcing Q of the fixed 2x2 channel (rows = users) as tools/cmu_ap.py:138-151 builds it; row u of the
    fixture is what user u's single antenna receives."""
    mp = [mpdu_of(150, 2000), mpdu_of(220, 2001)]
    mcs = [3, 5]
    q = np.linalg.inv(H22)
    q = q / np.linalg.norm(q) * np.sqrt(2.0)
    phy = phy80211.phy80211(ifDebug=False)
    with contextlib.redirect_stdout(io.StringIO()):
        phy.genAmpduMu(nUser=2, bfQ=[q] * 64, groupId=2,
                       ampdu0=mac80211.genAmpduVHT([mp[0]]), mod0=p8h.modulation(p8h.F.VHT, mcs[0], p8h.BW.BW20, 1, False),
                       ampdu1=mac80211.genAmpduVHT([mp[1]]), mod1=p8h.modulation(p8h.F.VHT, mcs[1], p8h.BW.BW20, 1, False))
        ss = phy.genFinalSig(multiplier=18.0, cfoHz=0.0, num=1, gap=True, gapLen=GAP)
    tx = np.array(ss, dtype=np.complex128)
    np.savez_compressed(
        os.path.join(HERE, "mu_vht_2users.npz"), iq=(H22 @ tx).astype(np.complex64), sync_idx=np.int32(GAP + 160 + 16),
        mpdu0=np.frombuffer(mp[0], dtype=np.uint8), mpdu1=np.frombuffer(mp[1], dtype=np.uint8),
        mcs=np.array(mcs, dtype=np.int32))
    print("mu", tx.shape)


def make_ampdu():
    """VHT A-MPDU with three MPDUs of different lengths (mac80211.genAmpduVHT): what a standard-conformant
    de-aggregation must return; the reference's own receiver OR-accumulates the delimiter lengths and loses them."""
    mps = [mpdu_of(n, 3000 + i) for i, n in enumerate((120, 333, 64))]
    phy = phy80211.phy80211(ifDebug=False)
    with contextlib.redirect_stdout(io.StringIO()):
        phy.genFromAmpdu(mac80211.genAmpduVHT(mps), p8h.modulation(p8h.F.VHT, 5, p8h.BW.BW20, 1, False),
                         vhtPartialAid=0, vhtGroupId=0)
        ss = phy.genFinalSig(multiplier=12.0, cfoHz=0.0, num=1, gap=True, gapLen=GAP)
    np.savez_compressed(
        os.path.join(HERE, "ampdu_vht_3mpdu.npz"), iq=np.array(ss, dtype=np.complex128).astype(np.complex64),
        sync_idx=np.int32(GAP + 160 + 16), lens=np.array([len(m) for m in mps], dtype=np.int32),
        mpdus=np.frombuffer(b"".join(mps), dtype=np.uint8))
    print("ampdu", np.array(ss).shape)


def make_viterbi():
    """Integer-LLR Viterbi vectors from the reference's Python decoder (ties included)."""
    rng = np.random.RandomState(7)
    out = {}
    crs = [p8h.CR.CR12, p8h.CR.CR23, p8h.CR.CR34, p8h.CR.CR56]
    per = [(2, 1), (4, 2), (6, 3), (10, 5)]  # puncture period: coded-slots, trellis steps
    kept = [2, 3, 4, 6]
    for ci, cr in enumerate(crs):
        for tag, T, amp in (("a", 120, 2), ("b", 330, 6)):
            T = (T // per[ci][1]) * per[ci][1]
            n = T // per[ci][1] * kept[ci]
            # encode a random message so that a real path exists, then perturb heavily
            msg = list(rng.randint(0, 2, T - 6)) + [0] * 6
            coded = p8h.procBcc([int(b) for b in msg], cr)
            llr = (2 * np.array(coded[:n]) - 1) * amp + rng.randint(-amp - 1, amp + 2, n)
            llr = llr.astype(np.int32)
            with contextlib.redirect_stdout(io.StringIO()):
                bits = p8h.procViterbiDecoder([float(x) for x in llr], T, cr)
            out["llr_%d%s" % (ci, tag)] = llr.astype(np.float32)
            out["bits_%d%s" % (ci, tag)] = np.array(bits, dtype=np.uint8)
            out["T_%d%s" % (ci, tag)] = np.int32(T)
            print("viterbi cr", ci, tag, "T", T, "n", n, "errs vs msg", int(np.sum(np.array(bits) != np.array(msg))))
    np.savez_compressed(os.path.join(HERE, "viterbi_pyref.npz"), **out)


def copy_ref_fixtures():
    for fn in ("sig80211GenMultipleSiso_1x1_0.bin", "sig80211GenMultipleMimo_2x2_0.bin",
               "sig80211GenMultipleMimo_2x2_1.bin"):
        shutil.copyfile(os.path.join(REF_TOOLS, fn), os.path.join(HERE, fn))
        os.chmod(os.path.join(HERE, fn), 0o644)
    with open(os.path.join(REF_TOOLS, "debug.txt")) as fh:
        for line in fh:
            if line.startswith("mac packet"):
                hexs = line.split()[3]
                with open(os.path.join(HERE, "ref_fixture_mpdu.hex"), "w") as o:
                    o.write(hexs + "\n")


def make_txbits():
    """The bit vectors the reference keeps in tools/txbits.py (data, scrambled, coded and interleaved bits of its
    94-byte example packet at HT MCS0; a VHT-SIG-B field before and after coding and interleaving; VHT data bits
    before and after scrambling), read as literals and stored as arrays."""
    import ast
    import re
    src = open(os.path.join(REF_TOOLS, "txbits.py")).read()
    out = {}
    for m in re.finditer(r"^(\w+)\s*=\s*(\[.*?\])", src, re.S | re.M):
        out[m.group(1)] = np.array(ast.literal_eval(m.group(2)), dtype=np.uint8)
    assert len(out) == 11
    np.savez_compressed(os.path.join(HERE, "ref_txbits.npz"), **out)


if __name__ == "__main__":
    copy_ref_fixtures()
    make_txbits()
    make_viterbi()
    make_frames()
    make_soft()
    make_ndp()
    make_mu()
    make_ampdu()
