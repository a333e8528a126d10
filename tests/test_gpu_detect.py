"""Frame detection (presiso -> trigger -> sync) on the GPU against the CPU oracle of the same blocks.
Sync positions and RSSI are exact given the same trigger byte; the autocorrelation stream itself is computed
with a different (more accurate) summation than the reference's running sums, so CFO and SNR are compared
to tolerance and the inputs keep clear of the 0.3 plateau threshold's rounding margin."""
import numpy as np
import pytest

from util import H22, blob_of, golden_frames, model_chain, ref_fixtures

pytestmark = pytest.mark.gpu

RAD_ATOL = 2e-6   # rad/sample: atan2f / sinf / cosf of the device against glibc's
SNR_RTOL = 1e-4


def run_detect(pkg, rxctx, iq):
    import torch
    d, n = rxctx.detect(torch.from_numpy(np.ascontiguousarray(iq, dtype=np.complex64)).cuda())
    rxctx.sync()
    return d.cpu().numpy()[: n * 32].view(pkg.rx.FRAME_DESC_DTYPE)


def check(pkg, rxctx, oracle, iq):
    ref = oracle.detect(iq)
    got = run_detect(pkg, rxctx, iq)
    assert list(got["sync_off"].astype(np.int64)) == list(ref["pos"])
    assert np.array_equal(got["n_avail"].astype(np.int64), len(iq) - ref["pos"])
    assert np.allclose(got["rad"], ref["rad"], rtol=0, atol=RAD_ATOL)
    fin = np.isfinite(ref["snr"])
    assert np.array_equal(np.isfinite(got["snr"]), fin)
    assert np.allclose(got["snr"][fin], ref["snr"][fin], rtol=SNR_RTOL, atol=1e-3)
    assert np.array_equal(got["rssi"], ref["rssi"])
    return ref, got


def noisy(x, sigma, seed):
    rng = np.random.default_rng(seed)
    return (x + sigma * (rng.standard_normal(len(x)) + 1j * rng.standard_normal(len(x)))).astype(np.complex64)


def test_reference_fixtures(pkg, rxctx, oracle):
    for name, ants, sidx in ref_fixtures():
        ref, got = check(pkg, rxctx, oracle, ants[0])
        assert len(ref["pos"]) == 1
    # the reference's own capture: the sync byte lands on L-LTF start + 16
    ref, got = check(pkg, rxctx, oracle, ref_fixtures()[0][1][0])
    assert int(got["sync_off"][0]) == 1376


def test_stream_of_frames(pkg, rxctx, oracle):
    parts = []
    for f in golden_frames():
        iq = f["iq"] if f["nss"] == 1 else (H22 @ f["iq"])
        parts.append(iq[0].astype(np.complex64))
    x = noisy(np.concatenate(parts + [np.zeros(700, np.complex64)]), 0.004, 3)
    ref, got = check(pkg, rxctx, oracle, x)
    assert len(ref["pos"]) == len(parts)
    # CFO frames: the estimate is the transmitted offset
    rads = [f["rad"] for f in golden_frames()]
    assert np.allclose(got["rad"], rads, atol=2e-4)


def test_back_to_back_frames_and_tile_borders(pkg, rxctx, oracle):
    """Frames without a gap between them, at offsets that put preambles across the kernel's 2048-sample tiles."""
    fr = [f for f in golden_frames() if f["nss"] == 1]
    for lead in (0, 1507, 1890, 2047):
        parts = [np.zeros(lead, np.complex64)]
        for f in fr[:6]:
            parts.append(f["iq"][0][400:].astype(np.complex64))   # drop the leading silence
        x = noisy(np.concatenate(parts + [np.zeros(300, np.complex64)]), 0.003, lead)
        ref, got = check(pkg, rxctx, oracle, x)
        assert len(ref["pos"]) >= 5


def test_capture_not_on_a_16_byte_boundary(pkg, rxctx, oracle):
    """The capture pointer may sit on any sample (8-byte) boundary: the scan then loads sample by sample and finds
    the same sync bytes as on an aligned copy."""
    import torch
    parts = [f["iq"][0].astype(np.complex64) for f in golden_frames() if f["nss"] == 1][:5]
    x = noisy(np.concatenate(parts + [np.zeros(333, np.complex64)]), 0.004, 11)
    ref, want = check(pkg, rxctx, oracle, x)
    buf = torch.zeros(len(x) + 1, dtype=torch.complex64, device="cuda")
    view = buf[1:]
    view.copy_(torch.from_numpy(x))
    assert view.data_ptr() % 16 == 8
    d, n = rxctx.detect(view)
    rxctx.sync()
    got = d.cpu().numpy()[: n * 32].view(pkg.rx.FRAME_DESC_DTYPE)
    assert got.tobytes() == want.tobytes()


def test_nothing_to_find(pkg, rxctx, oracle):
    rng = np.random.default_rng(1)
    for x in (np.zeros(5000, np.complex64),
              (0.1 * (rng.standard_normal(20000) + 1j * rng.standard_normal(20000))).astype(np.complex64),
              np.zeros(17, np.complex64)):
        got = run_detect(pkg, rxctx, x)
        ref = oracle.detect(x)
        assert len(got) == len(ref["pos"])


def test_periodic_signal_long_plateau(pkg, rxctx, oracle):
    """A 16-periodic tone keeps the autocorrelation at 1 for thousands of samples: one countdown per plateau,
    sync rejects what is not an L-LTF or accepts it exactly as the oracle does."""
    n = 6000
    t = np.arange(n)
    x = np.zeros(n, np.complex64)
    x[500:4500] = (0.3 * np.exp(2j * np.pi * t[500:4500] / 16.0)).astype(np.complex64)
    x = noisy(x, 0.002, 7)
    check(pkg, rxctx, oracle, x)


def test_truncated_capture(pkg, rxctx, oracle):
    f = golden_frames()[5]
    iq = f["iq"][0].astype(np.complex64)
    for cut in (560, 700, 760, 815, 900):   # before / inside the 240 items sync needs after the trigger byte
        check(pkg, rxctx, oracle, noisy(iq[:cut], 0.002, cut))


def test_detect_then_decode(pkg, rxctx, oracle):
    """Raw capture in, MPDUs out: the descriptors go straight into the RX batch on the device."""
    import torch
    frames = [f for f in golden_frames() if f["nss"] == 1]
    x = noisy(np.concatenate([f["iq"][0].astype(np.complex64) for f in frames] + [np.zeros(500, np.complex64)]), 0.003, 11)
    d_iq = torch.from_numpy(x).cuda()
    desc, n = rxctx.detect(d_iq)
    assert n == len(frames)
    res = rxctx.rx_batch(d_iq, None, desc[: n * 32], n, out_stride=1600)
    rxctx.sync()
    out_len = res["out_len"].cpu().numpy()
    out = res["out"].cpu().numpy()
    for i, f in enumerate(frames):
        assert bytes(out[i, 3: out_len[i] - 1]) == f["mpdu"], f["name"]


def capture_pdus(rxctx, x0, x1=None):
    import torch
    d0 = torch.from_numpy(np.ascontiguousarray(x0, np.complex64)).cuda()
    d1 = torch.from_numpy(np.ascontiguousarray(x1, np.complex64)).cuda() if x1 is not None else None
    r = rxctx.rx_capture(d0, d1, out_stride=1600)
    rxctx.sync()
    n = r["n"]
    out_len = r["out_len"].cpu().numpy()[:n]
    out = r["out"].cpu().numpy()[:n]
    pdus = []
    for i in range(n):
        blob, at = bytes(out[i, : out_len[i]]), 0
        while at + 4 <= len(blob):
            ln = blob[at + 1] | (blob[at + 2] << 8)
            pdus.append(blob[at: at + ln + 4])
            at += ln + 4
    return pdus, n


def test_capture_to_pdus_with_false_sync_bytes(pkg, rxctx, oracle):
    """gr80211_rx_capture against the chain presiso -> trigger -> sync -> signal -> demod -> decode of the oracle.
    A 16-periodic burst inside a frame's payload and another one on its own make sync bytes that `signal` must
    ignore (inside a frame) or drop after a failed L-SIG check."""
    frames = [f for f in golden_frames() if f["nss"] == 1]
    parts = [f["iq"][0].astype(np.complex64) for f in frames]
    x = np.concatenate(parts + [np.zeros(900, np.complex64)])
    t = np.arange(len(x))
    tone = (0.25 * np.exp(2j * np.pi * t / 16.0)).astype(np.complex64)
    x[1500:1900] += tone[1500:1900]                    # inside frame 0's payload
    tail = len(x) - 800
    x[tail: tail + 500] += tone[tail: tail + 500]        # after the last frame
    x = noisy(x, 0.003, 21)
    det = oracle.detect(x)
    assert len(det["pos"]) > len(frames)
    want, n_ok = model_chain(oracle, x, None, list(zip(det["pos"].tolist(), det["rad"].tolist())))
    got, n = capture_pdus(rxctx, x)
    assert n == model_chain.dealt_with
    assert got == want
    # the burst destroys frame 0's payload (no FCS pass); every other frame comes back
    assert [p[3:-1] for p in got] == [f["mpdu"] for f in frames[1:]]


def test_capture_two_antennas(pkg, rxctx, oracle):
    parts = []
    sent = []
    for f in golden_frames():
        if f["nss"] == 2:
            parts.append((H22 @ f["iq"]).astype(np.complex64))
            sent.append(f["mpdu"])
    name, ants, sidx = ref_fixtures()[1]
    parts.append(np.stack(ants))
    x = np.concatenate(parts + [np.zeros((2, 400), np.complex64)], axis=1)
    rng = np.random.default_rng(4)
    x = (x + 0.002 * (rng.standard_normal(x.shape) + 1j * rng.standard_normal(x.shape))).astype(np.complex64)
    det = oracle.detect(x[0])
    want, n_ok = model_chain(oracle, x[0], x[1], list(zip(det["pos"].tolist(), det["rad"].tolist())))
    got, n = capture_pdus(rxctx, x[0], x[1])
    assert got == want and n == model_chain.dealt_with
    assert [p[3:-1] for p in got][: len(sent)] == sent
