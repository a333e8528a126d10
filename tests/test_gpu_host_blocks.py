"""The C++ host blocks (fork-gr-ieee80211_amd/host) driven like GNU Radio drives the reference's blocks:
streams cut into scheduler-sized windows, tags on the sync byte / first soft bit, messages out.
Expected messages come from the CPU oracle run frame by frame under a Python model of `signal`'s
trigger/copy state machine (lib/signal_impl.cc:75-106,153-202)."""
import os
import subprocess

import numpy as np
import pytest

from util import H22, blob_of, golden_frames, model_chain, ref_fixtures

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# GR80211_PUMP: another build of the same program (e.g. one compiled with host sanitizers)
PUMP = os.environ.get("GR80211_PUMP") or os.path.join(ROOT, "fork-gr-ieee80211_amd", "gr80211_pump")


ABI_CHAIN = os.environ.get("GR80211_ABI_CHAIN") or os.path.join(ROOT, "fork-gr-ieee80211_amd", "gr80211_abi_chain")


def run_pump(args, prog=PUMP, env=None):
    assert os.path.exists(prog), "host blocks not built: run __graft_entry__.build()"
    p = subprocess.run([prog] + [str(a) for a in args], capture_output=True, text=True, timeout=170,
                       env=dict(os.environ, **env) if env else None)
    assert p.returncode == 0, (p.returncode, p.stderr[-2000:])
    pdus = [bytes.fromhex(l.split()[1]) for l in p.stdout.splitlines() if l.startswith("PDU ")]
    stat = [l for l in p.stdout.splitlines() if l.startswith("STAT")]
    return pdus, stat[0] if stat else ""


def write_stream(tmp_path, iq0, iq1, triggers):
    f0 = tmp_path / "iq0.bin"
    iq0.astype(np.complex64).tofile(f0)
    files = [f0]
    if iq1 is not None:
        f1 = tmp_path / "iq1.bin"
        iq1.astype(np.complex64).tofile(f1)
        files.append(f1)
    tf = tmp_path / "trig.txt"
    with open(tf, "w") as fh:
        for pos, rad in triggers:
            fh.write("%d %.9g 30.0 -40.0\n" % (pos, rad))
    return tf, files


def siso_stream(noise_sigma, seed):
    rng = np.random.default_rng(seed)
    frames = [f for f in golden_frames() if f["nss"] == 1]
    parts, triggers, pos = [], [], 0
    for k, f in enumerate(frames):
        iq = f["iq"][0]
        parts.append(iq)
        triggers.append((pos + f["sync_idx"], f["rad"]))
        if k == 0:
            triggers.append((pos + f["sync_idx"] + 900, 0.0))   # inside frame 0: must be ignored
        if k == 2:
            triggers.append((pos + 100, 0.0))                    # on the silence before frame 2: L-SIG fails
        pos += len(iq)
    iq0 = np.concatenate(parts + [np.zeros(500, np.complex64)]).astype(np.complex64)
    if noise_sigma > 0:
        iq0 = (iq0 + noise_sigma * (rng.standard_normal(len(iq0)) + 1j * rng.standard_normal(len(iq0)))).astype(np.complex64)
    return iq0, triggers, frames


def mimo_stream(seed):
    """The two-stream golden frames through the fixed 2x2 channel, back to back on two antennas."""
    rng = np.random.default_rng(seed)
    frames = [f for f in golden_frames() if f["nss"] == 2]
    parts, triggers, pos = [], [], 0
    for f in frames:
        r = (H22 @ f["iq"]).astype(np.complex64)
        parts.append(r)
        triggers.append((pos + f["sync_idx"], f["rad"]))
        pos += r.shape[1]
    iq = np.concatenate(parts, axis=1)
    iq = (iq + 0.001 * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
    return iq[0], iq[1], triggers, frames


@pytest.mark.parametrize("chunk,batch", [(4096, 1000), (777, 2), (8192, 1)])
def test_rx_block_siso_stream(tmp_path, oracle, chunk, batch):
    iq0, triggers, frames = siso_stream(0.002, 6)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    assert n_frames == len(frames)
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    got, stat = run_pump(["rx", 1, chunk, batch, tf] + files)
    assert "frames=%d " % n_frames in stat + " ", stat
    assert got == want
    # every transmitted MPDU came back, in stream order
    mpdus = [p[3:-1] for p in got]
    assert mpdus == [f["mpdu"] for f in frames]


@pytest.mark.parametrize("env", [{"GR80211_UPLOAD": "dma", "GR80211_UPLOAD_STAGE": "0"},
                                 {"GR80211_UPLOAD": "pull", "GR80211_UPLOAD_STAGE": "0"},
                                 {"GR80211_UPLOAD_STREAMS": "2", "GR80211_UPLOAD_SPLIT": "3", "GR80211_UPLOAD_STAGE": "0"},
                                 {"GR80211_PUMP_PIN": "0", "GR80211_UPLOAD_STAGE": "0"}, {"GR80211_PUMP_PIN": "0"}, {},
                                 {"GR80211_STAGE_MAX_PINNED": "1000"}, {"GR80211_STAGE_BYTES": "40000"},
                                 {"GR80211_RX_MAX_HOLD": "20000", "GR80211_DECODE_MAX_HOLD": "5000"}])
def test_rx_and_decode_blocks_whatever_way_the_windows_travel(tmp_path, pkg, rxctx, oracle, env):
    """The streaming blocks give the same messages whether their windows go up by the runtime's copy engines, by the
    copy kernel, by two upload streams in three pieces, synchronously from ordinary memory, copied by hand into the
    uploader's page-locked ring (the default for windows this small, from page-locked and from ordinary memory), partly
    one way and partly the other, through a ring so small that it wraps with every window, or in rounds cut by the sample limit instead of the frame count."""
    iq0, triggers, frames = siso_stream(0.002, 6)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    got, stat = run_pump(["rx", 1, 3000, 2, tf] + files, env=env)
    assert got == want
    tgf, lf, want_d, _, _ = decode_stream_files(tmp_path, pkg, rxctx, oracle)
    got, stat = run_pump(["decode", 5000, 3, tgf, lf], env=env)
    assert got == want_d


class UdpListener:
    """Collects the datagrams that arrive on a loopback port of its own."""

    def __init__(self):
        import socket
        import threading
        self.sock = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
        self.sock.setsockopt(socket.SOL_SOCKET, socket.SO_RCVBUF, 1 << 22)
        self.sock.bind(("127.0.0.1", 0))
        self.sock.settimeout(0.2)
        self.port = self.sock.getsockname()[1]
        self.got = []
        self._stop = False
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()

    def _run(self):
        import socket
        quiet = 0
        while quiet < 5 or not self._stop:
            try:
                self.got.append(self.sock.recvfrom(1 << 17)[0])
                quiet = 0
            except socket.timeout:
                quiet += 1 if self._stop else 0

    def finish(self):
        self._stop = True
        self._th.join(10)
        self.sock.close()
        return self.got


def test_pdus_leave_as_udp_datagrams(tmp_path, pkg, oracle):
    """The hand-off of the reference flowgraphs (socket_pdu UDP_CLIENT behind decode's "out" port,
    examples/cmu_v3/trxSta0.grc:301-313,1054): with --udp the command-line scheduler, and with send_udp a block of
    the streaming C-ABI, send every published blob as one datagram; what arrives equals the messages, in order."""
    iq0, triggers, frames = siso_stream(0.002, 6)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    ear = UdpListener()
    got, stat = run_pump(["--udp", "127.0.0.1:%d" % ear.port, "rx", 1, 4096, 3, tf] + files)
    heard = ear.finish()
    assert got == want and heard == want
    # the same through the C-ABI block
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    sync, tags = sync_stream(len(iq0), triggers)
    ear = UdpListener()
    with stream.StreamBlock(stream.RX, 3) as b:
        b.send_udp("127.0.0.1", ear.port)
        msgs = b.run([sync, iq0], tags, 4096)
    heard = ear.finish()
    assert msgs == want and heard == want


def test_rx_block_false_lsig_swallows_stream(tmp_path, oracle):
    """With this noise seed the tagged sync byte on silence passes the L-SIG check by chance and announces a
    frame longer than the rest of the stream: like `signal`, the block must then ignore every later trigger and
    decode the (garbage) frame with the samples it has when the stream ends."""
    iq0, triggers, frames = siso_stream(0.002, 5)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    assert n_frames == 3 and len(want) == 2
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    got, stat = run_pump(["rx", 1, 2048, 4, tf] + files)
    assert "frames=3 " in stat + " ", stat
    assert got == want


def test_rx_block_two_antennas(tmp_path, oracle):
    rng = np.random.default_rng(9)
    parts, triggers, pos, sent = [], [], 0, []
    for f in [f for f in golden_frames() if f["nss"] == 2]:
        rx = (H22 @ f["iq"]).astype(np.complex64)
        parts.append(rx)
        triggers.append((pos + f["sync_idx"], f["rad"]))
        sent.append(f["mpdu"])
        pos += rx.shape[1]
    name, ants, sidx = ref_fixtures()[1]
    parts.append(np.stack(ants))
    triggers.append((pos + sidx, 0.0))
    iq = np.concatenate(parts, axis=1)
    iq = (iq + 0.001 * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
    want, n_frames = model_chain(oracle, iq[0], iq[1], triggers)
    assert n_frames == len(triggers)
    tf, files = write_stream(tmp_path, iq[0], iq[1], triggers)
    got, stat = run_pump(["rx", 2, 3000, 3, tf] + files)
    assert got == want
    assert [p[3:-1] for p in got][: len(sent)] == sent


def test_rx_block_reference_fixture(tmp_path, oracle):
    from util import ref_fixture_mpdu
    name, ants, sidx = ref_fixtures()[0]
    tf, files = write_stream(tmp_path, ants[0], None, [(sidx, 0.0)])
    got, stat = run_pump(["rx", 1, 1024, 16, tf] + files)
    assert len(got) == 1 and got[0][3:-1] == ref_fixture_mpdu()
    assert got[0][0] == 0 and got[0][-1] == 0  # legacy, 6 Mb/s


def decode_stream_files(tmp_path, pkg, rxctx, oracle):
    """The golden one-stream frames as the soft-bit stream `demod` hands to `decode` (tags on a packet's first item,
    untagged items between packets, an oversize packet at the end): (tag file, soft-bit file, expected messages)."""
    import torch
    from util import build_arena
    frames = [f for f in golden_frames() if f["nss"] == 1]
    iq0, iq1, sync, navail = build_arena([([f["iq"][0]], f["sync_idx"]) for f in frames])
    rads = np.array([f["rad"] for f in frames], np.float32)
    desc = rxctx.make_desc(sync, navail, rad=rads)
    info, llr = rxctx.demod_batch(torch.from_numpy(iq0).cuda(), None, desc, len(frames))
    stream, tags, want = [], [], []
    junk = np.full(37, 0.25, np.float32)  # untagged items between packets are passed over
    at = 0
    for i, f in enumerate(frames):
        assert int(info["status"][i]) == 0
        tot = int(info["total"][i])
        o = int(info["llr_off"][i])
        stream.append(junk)
        at += len(junk)
        tags.append((at, int(info["format"][i]), int(info["len"][i]), tot, int(info["cr"][i]), int(info["mcs"][i]),
                     int(info["ampdu"][i]), int(info["trellis"][i])))
        stream.append(llr[o:o + tot])
        at += tot
        ref = oracle.rx_frame(iq0[: int(sync[i]) + int(navail[i])], None, int(sync[i]), float(rads[i]), want_llr=False)
        want += [blob_of(p) for p in ref["pdus"]]
    # an oversize packet is skipped whole (lib/decode_impl.cc:93-97)
    tags.append((at, 0, 5000, 64, 0, 0, 0, 40022))
    stream.append(np.zeros(64, np.float32))
    lf = tmp_path / "llr.bin"
    np.concatenate(stream).astype(np.float32).tofile(lf)
    tf = tmp_path / "tags.txt"
    with open(tf, "w") as fh:
        for t in tags:
            fh.write(" ".join(str(v) for v in t) + "\n")
    return tf, lf, want, len(tags), frames


@pytest.mark.parametrize("chunk,batch", [(4096, 1000), (1000, 2)])
def test_decode_block_stream(tmp_path, pkg, rxctx, oracle, chunk, batch):
    """decode drop-in: soft-bit stream with the tags demod attaches to a packet's first item."""
    tf, lf, want, n_tags, frames = decode_stream_files(tmp_path, pkg, rxctx, oracle)
    got, stat = run_pump(["decode", chunk, batch, tf, lf])
    assert got == want
    assert "packets=%d " % n_tags in stat + " ", stat
    assert [p[3:-1] for p in got] == [f["mpdu"] for f in frames]


def test_capture_block_raw_stream(tmp_path, oracle):
    """capture_rx: antenna stream only; detection, `signal`'s trigger search and the RX path all run on the GPU."""
    iq0, triggers, frames = siso_stream(0.003, 6)
    det = oracle.detect(iq0)
    want, n_ok = model_chain(oracle, iq0, None, list(zip(det["pos"].tolist(), det["rad"].tolist())))
    f0 = tmp_path / "iq0.bin"
    iq0.astype(np.complex64).tofile(f0)
    got, stat = run_pump(["capture", 1, 5000, f0])
    assert got == want
    assert [p[3:-1] for p in got] == [f["mpdu"] for f in frames]
    assert "frames=%d " % model_chain.dealt_with in stat + " ", stat


@pytest.mark.parametrize("hold,min_round", [(438016, 0), (700001, 0), (438016, 8192), (700001, 30000)])
def test_capture_block_in_rounds_equals_the_whole_capture(tmp_path, pkg, rxctx, hold, min_round):
    """capture_rx cuts a long stream into rounds and finds each round's sync bytes when it cuts it; the messages are
    those of gr80211_rx_capture over the whole stream at once, whatever the round length and the windows (frames of
    mixed formats and lengths with gaps of noise between them, so that round ends fall inside preambles, payloads
    and gaps alike).  min_round: the rounds of live reception, which go as soon as they hold that many new samples and
    fewer than two are on their way -- where they end then depends on the GPU's pace, the messages must not."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(71)
    parts = []
    for k in range(420):
        fmt = int(rng.integers(0, 3))
        mcs = int(rng.integers(0, 8 if fmt < 2 else 9))
        iq, _ = tx.gen_frame(fmt, mcs, 1, tx.make_mpdu(int(rng.integers(40, 1500)), rng))
        parts.append(np.zeros(int(rng.integers(200, 1200)), np.complex64))
        parts.append(np.asarray(iq[0] * 12.0, np.complex64))
    parts.append(np.zeros(600, np.complex64))
    iq0 = np.concatenate(parts)
    sigma = np.sqrt(np.mean(np.abs(parts[1]) ** 2) / 2) * 10 ** (-35.0 / 20)  # 35 dB below the frames
    iq0 = (iq0 + sigma * (rng.standard_normal(len(iq0)) + 1j * rng.standard_normal(len(iq0)))).astype(np.complex64)
    assert len(iq0) > 3 * hold
    whole = rxctx.rx_capture(torch.from_numpy(iq0).cuda(), None, max_frames=2000)
    rxctx.sync()
    nf = whole["n"]
    lens = whole["out_len"][:nf].cpu().numpy()
    rows = whole["out"][:nf].cpu().numpy()
    want = []
    for i in range(nf):
        want += [blob_of(p) for p in pkg.rx.split_pdus(bytes(rows[i, : lens[i]]))]
    assert len(want) >= 400, (nf, len(want))
    f0 = tmp_path / "iq0.bin"
    iq0.tofile(f0)
    for chunk in (65536, 7001):
        got, stat = run_pump(["capture", 1, chunk, f0], env={"GR80211_RX_MAX_HOLD": str(hold), "GR80211_CAPTURE_MIN_ROUND": str(min_round)})
        bad = [i for i in range(min(len(got), len(want))) if got[i] != want[i]]
        assert got == want, (chunk, len(got), len(want), bad[:10], [len(got[i]) for i in bad[:5]], [len(want[i]) for i in bad[:5]])
        assert "frames=%d " % nf in stat + " ", stat


def test_rx_block_publishes_ndp_channel(tmp_path, oracle):
    """A sounding NDP in the stream: the block publishes the 1027-byte channel message (no mcs byte) between the
    data frames' PDUs."""
    from util import golden_ndp
    g = golden_ndp()
    frames = [f for f in golden_frames() if f["nss"] == 1][:2]
    parts = [frames[0]["iq"][0], g["iq"][0], frames[1]["iq"][0], np.zeros(300, np.complex64)]
    at = np.cumsum([0] + [len(p) for p in parts])
    triggers = [(int(at[0]) + frames[0]["sync_idx"], frames[0]["rad"]), (int(at[1]) + g["sync_idx"], 0.0),
                (int(at[2]) + frames[1]["sync_idx"], frames[1]["rad"])]
    iq0 = np.concatenate(parts).astype(np.complex64)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    assert n_frames == 3 and [len(w) for w in want][1] == 1027 and want[1][0] == 20
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    got, stat = run_pump(["rx", 1, 1500, 8, tf] + files)
    assert got == want


# ---- the same blocks through the streaming C-ABI (include/gr80211_host.h), driven from ctypes


def sync_stream(n, triggers):
    sync = np.zeros(n, np.uint8)
    tags = []
    for pos, rad in triggers:
        sync[pos] = 1
        tags += [(pos, 0, "rad", rad), (pos, 0, "snr", 30.0), (pos, 0, "rssi", -40.0)]
    return sync, tags


@pytest.mark.parametrize("chunk,batch", [(4096, 1000), (777, 2)])
def test_streaming_abi_rx_block(pkg, oracle, chunk, batch):
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    iq0, triggers, frames = siso_stream(0.002, 6)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    sync, tags = sync_stream(len(iq0), triggers)
    with stream.StreamBlock(stream.RX, batch) as b:
        assert b.n_in == 2 and len(b.forecast(512)) == 2
        got = b.run([sync, iq0], tags, chunk)
    assert got == want
    assert [p[3:-1] for p in got] == [f["mpdu"] for f in frames]


def test_streaming_abi_rx2_and_capture_blocks(pkg, oracle):
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    rng = np.random.default_rng(9)
    parts, triggers, pos = [], [], 0
    for f in [f for f in golden_frames() if f["nss"] == 2]:
        r = (H22 @ f["iq"]).astype(np.complex64)
        parts.append(r)
        triggers.append((pos + f["sync_idx"], f["rad"]))
        pos += r.shape[1]
    iq = np.concatenate(parts, axis=1)
    iq = (iq + 0.001 * (rng.standard_normal(iq.shape) + 1j * rng.standard_normal(iq.shape))).astype(np.complex64)
    want, n_frames = model_chain(oracle, iq[0], iq[1], triggers)
    sync, tags = sync_stream(iq.shape[1], triggers)
    with stream.StreamBlock(stream.RX2, 3) as b:
        assert b.run([sync, iq[0], iq[1]], tags, 3000) == want
    iq0, triggers, frames = siso_stream(0.003, 6)
    det = oracle.detect(iq0)
    want, _ = model_chain(oracle, iq0, None, list(zip(det["pos"].tolist(), det["rad"].tolist())))
    with stream.StreamBlock(stream.CAPTURE) as b:
        assert b.run([iq0], (), 5000) == want


def test_streaming_abi_mu_option_and_errors(pkg, oracle):
    """GR80211_OPT_MU_POS reaches the demodulator: the second user's PSDU of the two-user fixture comes out; bad
    arguments are refused without touching the device."""
    from importlib import import_module
    from util import golden_mu
    stream = import_module("fork-gr-ieee80211_amd.stream")
    g = golden_mu()
    for pos in (0, 1):
        iq0 = np.concatenate([np.asarray(g["iq"][pos], np.complex64), np.zeros(300, np.complex64)])
        ref = oracle.rx_frame(iq0, None, int(g["sync_idx"]), 0.0, want_llr=False, mu_pos=pos)
        sync, tags = sync_stream(len(iq0), [(int(g["sync_idx"]), 0.0)])
        with stream.StreamBlock(stream.RX, 4) as b:
            b.set_option(stream.OPT_MU_POS, pos)
            got = b.run([sync, iq0], tags, 2048)
        assert got == [blob_of(p) for p in ref["pdus"]] and [p[3:-1] for p in got] == [g["mpdus"][pos]]
    with stream.StreamBlock(stream.DECODE, 4) as b:
        with pytest.raises(RuntimeError):
            b.set_option(stream.OPT_MU_POS, 1)
        with pytest.raises(RuntimeError):
            b.work([np.zeros(8, np.float32)], [(0, 3, "format", 0)])
        b.set_option(stream.OPT_MAX_HOLD, 100000)
        b.set_option(stream.OPT_FINISH_IN_CALL, 1)
    with stream.StreamBlock(stream.RX, 4) as b:
        b.set_option(stream.OPT_MAX_HOLD, 500000)
        with pytest.raises(RuntimeError):
            b.set_option(stream.OPT_FINISH_IN_CALL, 1)   # a block that reads the antenna stream never waits in a call
        with pytest.raises(RuntimeError):
            b.set_option(stream.OPT_MAX_HOLD, 0)
        with pytest.raises(RuntimeError):
            b.set_option(stream.OPT_MIN_ROUND, 16384)    # rounds by samples are the capture block's
    with stream.StreamBlock(stream.CAPTURE) as b:
        b.set_option(stream.OPT_MIN_ROUND, 16384)
        b.set_option(stream.OPT_MIN_ROUND, 0)
    with stream.StreamBlock(stream.DEMOD, 4) as b:
        b.set_option(stream.OPT_FINISH_IN_CALL, 0)
        with pytest.raises(RuntimeError):
            b.set_option(stream.OPT_MAX_HOLD, 100000)      # demod's rounds are cut by frames alone
    with pytest.raises(RuntimeError):
        stream.StreamBlock(17)


def test_streaming_abi_decode_block(pkg, rxctx, oracle):
    """decode through the streaming C-ABI: the tags of a packet's first soft bit as records, a sounding packet
    (trellis 0) whose channel vector travels in the tag, and the oracle's PSDUs for the rest."""
    import torch
    from importlib import import_module
    from util import build_arena
    stream = import_module("fork-gr-ieee80211_amd.stream")
    frames = [f for f in golden_frames() if f["nss"] == 1]
    iq0, iq1, sync, navail = build_arena([([f["iq"][0]], f["sync_idx"]) for f in frames])
    rads = np.array([f["rad"] for f in frames], np.float32)
    info, llr = rxctx.demod_batch(torch.from_numpy(iq0).cuda(), None, rxctx.make_desc(sync, navail, rad=rads), len(frames))
    chan = (np.arange(128) * (0.5 - 0.25j)).astype(np.complex64)
    parts, tags, want, at = [], [], [], 0
    for i, f in enumerate(frames):
        tot, o = int(info["total"][i]), int(info["llr_off"][i])
        for key in ("format", "len", "total", "cr", "mcs", "ampdu", "trellis"):
            tags.append((at, 0, key, int(info[key][i])))
        parts.append(llr[o:o + tot])
        at += tot
        ref = oracle.rx_frame(iq0[: int(sync[i]) + int(navail[i])], None, int(sync[i]), float(rads[i]), want_llr=False)
        want += [blob_of(p) for p in ref["pdus"]]
        if i == 1:
            for key, v in (("format", 2), ("len", 0), ("total", 16), ("cr", 0), ("mcs", 0), ("ampdu", 0), ("trellis", 0)):
                tags.append((at, 0, key, v))
            tags.append((at, 0, "mu2x1chan", 0, chan))
            parts.append(np.zeros(16, np.float32))
            at += 16
            want.append(bytes([20, 0, 4]) + chan.tobytes())
    parts.append(np.zeros(200, np.float32))
    with stream.StreamBlock(stream.DECODE, 2) as b:
        assert b.forecast(1000) == [1160]
        got = b.run([np.concatenate(parts).astype(np.float32)], tags, 1500)
    assert got == want


@pytest.mark.parametrize("two,chunk,nout,batch", [(False, 4096, 8192, 1000), (False, 777, 500, 2), (True, 3000, 4096, 3)])
def test_streaming_abi_signal_demod_decode_chain(pkg, oracle, two, chunk, nout, batch):
    """The three blocks of the reference's flowgraph as three blocks here, each a drop-in for its own: signal's
    items and tags go into demod, demod's soft bits and tags into decode; the messages equal the fused block's and
    the oracle's.  Windows and output buffers are small, so frames straddle calls in every block."""
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    if two:
        a, b, triggers, frames = mimo_stream(9)
        ants = [a, b]
    else:
        iq0, triggers, frames = siso_stream(0.002, 6)
        ants = [iq0]
    want, n_frames = model_chain(oracle, ants[0], ants[1] if two else None, triggers)
    sync, tags = sync_stream(len(ants[0]), triggers)
    with stream.StreamBlock(stream.SIGNAL2 if two else stream.SIGNAL, batch) as sg:
        sg.run([sync] + ants, tags, chunk, nout)
        sig_items = [sg.output(p) for p in range(len(ants))]
        sig_tags = sg.out_tags
    # signal: eight tags per frame on its first item, nsamp + 320 items per frame
    firsts = sorted(set(t[0] for t in sig_tags))
    assert len(firsts) == n_frames and len(sig_tags) == 8 * n_frames
    nsamp = [int(t[2]) for t in sig_tags if t[1] == "nsamp"]
    assert firsts == [int(v) for v in np.concatenate([[0], np.cumsum(np.array(nsamp) + 320)])[:-1]]
    assert len(sig_items[0]) == sum(nsamp) + 320 * n_frames
    assert [int(t[2]) for t in sig_tags if t[1] == "seq"] == list(range(1, n_frames + 1))
    assert all(t[3] is not None and len(t[3]) == 64 for t in sig_tags if t[1] == "chan")
    with stream.StreamBlock(stream.DEMOD2 if two else stream.DEMOD, batch) as dm:
        dm.run(sig_items, [(t[0], 0, t[1], t[2], t[3]) for t in sig_tags], chunk, nout)
        llr = dm.output(0)
        llr_tags = dm.out_tags
    totals = [int(t[2]) for t in llr_tags if t[1] == "total"]
    assert len(llr) == sum(totals)
    with stream.StreamBlock(stream.DECODE, batch) as dc:
        got = dc.run([llr], [(t[0], 0, t[1], t[2], t[3]) for t in llr_tags if t[1] in stream.TAG_KEYS[:11]], chunk)
    assert got == want


@pytest.mark.parametrize("whole_frames", [True, False])
def test_blocks_made_without_a_batch_size_finish_within_the_call(pkg, oracle, whole_frames):
    """demod and decode made the way the reference makes its own (no batch size) keep its timing: a frame's soft bits
    leave demod, and a packet's messages leave decode, in the call that brings the last item (lib/demod_impl.cc:279-314,
    lib/decode_impl.cc:128-160), so that nothing waits for a later call while the stream is silent.  Nothing is flushed
    here: what the calls themselves returned is all there is."""
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    iq0, triggers, frames = siso_stream(0.002, 6)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    sync, tags = sync_stream(len(iq0), triggers)
    with stream.StreamBlock(stream.SIGNAL, 8) as sg:
        sg.run([sync, iq0], tags, 4096, 8192)
        items, sig_tags = sg.output(0), [(t[0], 0, t[1], t[2], t[3]) for t in sg.out_tags]
    nsamp = [int(t[3]) for t in sig_tags if t[2] == "nsamp"]
    assert len(nsamp) == n_frames

    def play(b, data, tg, bounds, nout):
        """Windows end at `bounds`; after every call: (items consumed so far, what the block has given so far)."""
        at, seen = 0, []
        for end in bounds:
            idle = 0
            while at < end:
                # (a call may take nothing while pieces it sent on their way have not arrived: it is asked again)
                got = b.work([data[at:end]], [t for t in tg if at <= t[0] < end], nout)
                idle = 0 if got else idle + 1
                assert idle < 2000, "a window of new items was never taken"
                at += got
            seen.append((at, sum(len(x) for x in b.out[0]) if b.n_out else len(b.msgs)))
        return seen

    if whole_frames:
        ends = list(np.cumsum(np.array(nsamp) + 320) - 320)       # a window ends with a frame's last sample ...
        ends[-1] = len(items)
    else:
        ends = list(range(1500, len(items), 1500)) + [len(items)]   # ... or anywhere
    with stream.StreamBlock(stream.DEMOD) as dm:                    # batch size 0: the default
        seen = play(dm, items, sig_tags, ends, 1 << 20)
        llr, llr_tags = dm.output(0), [(t[0], 0, t[1], t[2], t[3]) for t in dm.out_tags if t[1] in stream.TAG_KEYS[:11]]
        totals = [int(t[3]) for t in llr_tags if t[2] == "total"]
        assert len(totals) == n_frames and len(llr) == sum(totals)
        if whole_frames:
            assert [g for _, g in seen] == list(np.cumsum(totals)), "soft bits of a complete frame waited for a later call"
    firsts = [t[0] for t in llr_tags if t[2] == "total"]
    if whole_frames:
        ends = [f + t for f, t in zip(firsts, totals)]
    else:
        ends = list(range(5000, len(llr), 5000)) + [len(llr)]
    with stream.StreamBlock(stream.DECODE) as dc:
        seen = play(dc, llr, llr_tags, ends, 0)
        assert dc.msgs == want, "messages of complete packets waited for a later call"
        counts = [g for _, g in seen]
        assert counts == sorted(counts) and counts[-1] == len(want)
        if whole_frames:
            assert counts[0] >= 1 and all(b > a for a, b in zip(counts, counts[1:])) or len(want) < n_frames
    # the fused block reads the antenna stream, which never pauses: made without a batch size it works in rounds of one
    # frame and publishes on the calls that follow, none of which needs to bring a frame
    import time
    with stream.StreamBlock(stream.RX) as rx:
        at, idle = 0, 0
        while at < len(iq0):
            got = rx.work([sync[at:at + 4096], iq0[at:at + 4096]], [t for t in tags if at <= t[0] < at + 4096])
            idle = 0 if got else idle + 1
            assert idle < 2000
            at += got
        for _ in range(400):
            if len(rx.msgs) >= len(want):
                break
            time.sleep(0.005)
            rx.work([sync[:0], iq0[:0]], [])
        assert rx.msgs == want
    # the option turns the timing off and on whatever the batch size
    with stream.StreamBlock(stream.DECODE, 1000) as dc:
        dc.set_option(stream.OPT_FINISH_IN_CALL, 1)
        play(dc, llr, llr_tags, [len(llr)], 0)
        assert dc.msgs == want
    with stream.StreamBlock(stream.DECODE) as dc:
        dc.set_option(stream.OPT_FINISH_IN_CALL, 0)
        play(dc, llr, llr_tags, [len(llr)], 0)
        assert dc.msgs == []          # a round of 4096 packets is not full: they wait for it, or for the end of the stream
        dc.flush()
        assert dc.msgs == want


# GR80211_STREAM_SEEDS=N: a longer sweep over seeds 1..N
@pytest.mark.parametrize("seed", list(range(1, 1 + int(os.environ.get("GR80211_STREAM_SEEDS", "6")))))
def test_streaming_chain_with_windows_that_change_every_call(pkg, oracle, seed):
    """signal -> demod -> decode and the fused block, every general_work call with another window size (1 to 6000
    items) and another amount of room on the outputs (64 to 9000 items), batches of 1 to 40 frames: frames, tags and
    packets straddle the calls in ever new ways, the messages stay those of the model of the reference's chain."""
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    rng = np.random.default_rng(seed)
    two = seed % 2 == 0
    if two:
        a, b, triggers, frames = mimo_stream(seed)
        ants = [a, b]
    else:
        iq0, triggers, frames = siso_stream(0.002, seed)
        ants = [iq0]
    want, n_frames = model_chain(oracle, ants[0], ants[1] if two else None, triggers)
    sync, tags = sync_stream(len(ants[0]), triggers)

    def chunk():
        return int(rng.integers(1, 6000))

    def room():
        return int(rng.integers(64, 9000))

    with stream.StreamBlock(stream.SIGNAL2 if two else stream.SIGNAL, int(rng.integers(1, 40))) as sg:
        sg.run([sync] + ants, tags, chunk, room)
        sig_items = [sg.output(p) for p in range(len(ants))]
        sig_tags = sg.out_tags
    with stream.StreamBlock(stream.DEMOD2 if two else stream.DEMOD, int(rng.integers(1, 40))) as dm:
        dm.run(sig_items, [(t[0], 0, t[1], t[2], t[3]) for t in sig_tags], chunk, room)
        llr = dm.output(0)
        llr_tags = dm.out_tags
    with stream.StreamBlock(stream.DECODE, int(rng.integers(1, 40))) as dc:
        got = dc.run([llr], [(t[0], 0, t[1], t[2], t[3]) for t in llr_tags if t[1] in stream.TAG_KEYS[:11]], chunk)
    assert got == want
    with stream.StreamBlock(stream.RX2 if two else stream.RX, int(rng.integers(1, 40))) as rx:
        assert rx.run([sync] + ants, tags, chunk) == want


def test_streaming_abi_chain_with_sounding_frame_and_mu(pkg, oracle):
    """A sounding NDP between data frames through signal -> demod -> decode: demod passes its two VHT-LTF windows on
    in the mu2x1chan tag with 1024 empty items and decode publishes the channel message; a two-user frame with the
    demod block set to the second user's position gives that user's PSDU."""
    from importlib import import_module
    from util import golden_mu, golden_ndp
    stream = import_module("fork-gr-ieee80211_amd.stream")
    g = golden_ndp()
    mu = golden_mu()
    frames = [f for f in golden_frames() if f["nss"] == 1][:2]
    for mu_pos in (0, 1):
        parts = [frames[0]["iq"][0], g["iq"][0], np.asarray(mu["iq"][mu_pos], np.complex64), frames[1]["iq"][0],
                 np.zeros(300, np.complex64)]
        at = np.cumsum([0] + [len(p) for p in parts])
        triggers = [(int(at[0]) + frames[0]["sync_idx"], frames[0]["rad"]), (int(at[1]) + g["sync_idx"], 0.0),
                    (int(at[2]) + int(mu["sync_idx"]), 0.0), (int(at[3]) + frames[1]["sync_idx"], frames[1]["rad"])]
        iq0 = np.concatenate(parts).astype(np.complex64)
        sync, tags = sync_stream(len(iq0), triggers)
        with stream.StreamBlock(stream.RX, 8) as b:
            b.set_option(stream.OPT_MU_POS, mu_pos)
            want = list(b.run([sync, iq0], tags, 1500))
        assert len(want) == 4 and len(want[1]) == 1027 and want[1][0] == 20 and want[2][3:-1] == mu["mpdus"][mu_pos]
        with stream.StreamBlock(stream.SIGNAL, 8) as sg:
            sg.run([sync, iq0], tags, 1500, 2000)
            items, stags = sg.output(0), sg.out_tags
        with stream.StreamBlock(stream.DEMOD, 8) as dm:
            dm.set_option(stream.OPT_MU_POS, mu_pos)
            dm.run([items], [(t[0], 0, t[1], t[2], t[3]) for t in stags], 1500, 2000)
            llr, ltags = dm.output(0), dm.out_tags
        assert sum(1 for t in ltags if t[1] == "mu2x1chan" and t[3] is not None and len(t[3]) == 128) == 1
        with stream.StreamBlock(stream.DECODE, 8) as dc:
            got = dc.run([llr], [(t[0], 0, t[1], t[2], t[3]) for t in ltags if t[1] in stream.TAG_KEYS[:11]], 1500)
        assert got == want


def test_instance_and_empty_run_of_every_block_kind(pkg):
    """What the reference's own qa_signal.py / qa_demod.py / qa_decode.py (python/ieee80211/qa_*.py:28-38) check for
    each block: it can be made, and a flowgraph that feeds it nothing runs and stops.  Here for every kind of the
    streaming C-ABI, with the port counts of the reference's io signatures."""
    from importlib import import_module
    stream = import_module("fork-gr-ieee80211_amd.stream")
    ports = {stream.SIGNAL: (2, 1), stream.SIGNAL2: (3, 2), stream.DEMOD: (1, 1), stream.DEMOD2: (2, 1),
             stream.DECODE: (1, 0), stream.RX: (2, 0), stream.RX2: (3, 0), stream.CAPTURE: (1, 0), stream.CAPTURE2: (2, 0)}
    for kind, (n_in, n_out) in ports.items():
        with stream.StreamBlock(kind, 16) as b:
            assert (b.n_in, b.n_out) == (n_in, n_out), kind
            # what gr::block::forecast returns for the reference's blocks: decode asks for 160 items more than it is to
            # produce (lib/decode_impl.cc:54-58), the others for as many (lib/signal_impl.cc:56-60, lib/demod_impl.cc:53-57)
            assert b.forecast(512) == [512 + 160 if kind == stream.DECODE else 512] * n_in
            dt = [np.uint8] + [np.complex64] * (n_in - 1) if kind in (stream.SIGNAL, stream.SIGNAL2, stream.RX, stream.RX2) \
                else [np.float32] if kind == stream.DECODE else [np.complex64] * n_in
            assert b.work([np.zeros(0, d) for d in dt]) == 0
            assert b.work([np.zeros(100, d) for d in dt]) == 100   # untagged items are passed over
            b.flush()
            assert b.msgs == [] and all(len(b.output(p)) == 0 for p in range(n_out))


@pytest.mark.parametrize("chunk,batch", [(4096, 1000), (777, 2), (2000, 0)])   # 0: no batch size, the reference's timing
def test_pump_chain_of_three_blocks(tmp_path, oracle, chunk, batch):
    """The command-line scheduler joins signal -> demod -> decode by buffers of `chunk` items (C++ only, no Python
    between the blocks): same messages as the fused block."""
    iq0, triggers, frames = siso_stream(0.002, 6)
    want, n_frames = model_chain(oracle, iq0, None, triggers)
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    got, stat = run_pump(["chain", 1, chunk, batch, tf] + files)
    assert got == want
    assert "frames=%d " % n_frames in stat + " ", stat


def test_decode_block_debug_counters_in_the_references_wording(tmp_path, oracle):
    """decode made with ifdebug prints, per packet, the running per-MCS counts of packets whose FCS held, worded as the
    reference's lines are (lib/decode_impl.cc:375-411,453-509), so that its performance scripts can scrape them
    (tools/performance/perf_siso.py:105-118: last line with "crc32" and the format's name, fields 3 ... 3 + nMcs).  The
    last line of every format must count exactly the messages of that format and MCS that were published."""
    iq0, triggers, frames = siso_stream(0.002, 6)
    tf, files = write_stream(tmp_path, iq0, None, triggers)
    p = subprocess.run([PUMP, "chain", "1", "4096", "1000", str(tf)] + [str(f) for f in files], capture_output=True, text=True,
                       timeout=170, env=dict(os.environ, GR80211_PUMP_DEBUG="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    pdus = [bytes.fromhex(l.split()[1]) for l in p.stdout.splitlines() if l.startswith("PDU ")]
    lines = [l for l in p.stdout.splitlines() if l.startswith("ieee80211 decode, ")]
    assert lines and all("crc32" in l for l in lines)
    total = 0
    for fmt, name, n_mcs in ((0, "legacy", 8), (1, "ht", 8), (2, "vht", 10)):
        want = [0] * n_mcs
        for m in pdus:
            if m[0] == fmt:
                want[m[-1] % 8 if fmt == 1 else m[-1]] += 1
        mine = [l for l in lines if ", %s crc32" % name in l]
        if not any(want):
            continue
        items = mine[-1].split(",")          # as the reference's scraper reads it
        got = [int(items[i].split(":")[1]) for i in range(3, 3 + n_mcs)]
        assert got == want, (name, got, want)
        assert items[2 + n_mcs + 1].startswith("cfo:") and items[2 + n_mcs + 3].startswith("rssi:")
        total += sum(want)
    assert total == len([m for m in pdus if m[0] in (0, 1, 2)]) > 0
    # the shared total in the very last line counts every packet of every format
    assert int(lines[-1].split("total:")[1].split(",")[0]) == total


@pytest.mark.parametrize("n_rx,chunk,batch", [(1, 4096, 1000), (1, 777, 2), (2, 1500, 3), (1, 3000, 0), (2, 900, 0)])
def test_streaming_abi_from_plain_c(tmp_path, oracle, n_rx, chunk, batch):
    """signal -> demod -> decode through gr80211_block_work_out / gr80211_block_work from a C program (tags as
    records, vector values copied between blocks): same messages as the model of the reference's chain."""
    if n_rx == 1:
        iq0, triggers, frames = siso_stream(0.002, 6)
        iq1 = None
    else:
        iq0, iq1, triggers, frames = mimo_stream(7)
    want, n_frames = model_chain(oracle, iq0, iq1, triggers)
    tf, files = write_stream(tmp_path, iq0, iq1, sorted(triggers))
    got, stat = run_pump([n_rx, chunk, batch, tf] + files, prog=ABI_CHAIN)
    assert got == want
    assert stat == "STAT pdus=%d" % len(want)


@pytest.mark.parametrize("n_rx,chunk,batch", [(1, 3000, 7), (1, 777, 2), (1, 65536, 1000), (2, 1500, 3), (1, 2500, 0), (2, 4096, 0)])
def test_pump_chain_on_three_threads_equals_one_thread(pkg, tmp_path, oracle, n_rx, chunk, batch):
    """gr80211_pump chaint (signal, demod and decode on a thread each, ring buffers between them, as GNU Radio's
    scheduler runs a flowgraph) publishes the messages of gr80211_pump chain (the same blocks called in turn) and of
    the model of the reference's chain, in the same order, with page-locked buffers (demod then takes its frames'
    items straight from the window) and with ordinary ones (it collects them first)."""
    import subprocess
    if n_rx == 1:
        iq0, triggers, frames = siso_stream(0.002, 40)
        iq1 = None
    else:
        iq0, iq1, triggers, frames = mimo_stream(41)
    want, n_frames = model_chain(oracle, iq0, iq1, triggers)
    tf, files = write_stream(tmp_path, iq0, iq1, sorted(triggers))
    outs = {}
    # "direct": items also leave signal and demod straight from the device into the page-locked buffer of the block
    # downstream (what the blocks do by themselves for windows of a megabyte and more)
    for mode, pin in (("chain", "1"), ("chaint", "1"), ("chaint", "0"), ("chaint", "direct")):
        env = dict(os.environ, GR80211_PUMP_PIN="0" if pin == "0" else "1")
        if pin == "direct":
            env.update(GR80211_DEMOD_DIRECT_OUT="1", GR80211_SIGNAL_DIRECT_OUT="1")
        r = subprocess.run([PUMP, mode, str(n_rx), str(chunk), str(batch), str(tf)] + [str(f) for f in files],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[(mode, pin)] = [bytes.fromhex(l.split()[1]) for l in r.stdout.splitlines() if l.startswith("PDU ")]
    assert len(want) > 3
    assert outs[("chain", "1")] == want
    assert outs[("chaint", "1")] == want and outs[("chaint", "0")] == want and outs[("chaint", "direct")] == want
