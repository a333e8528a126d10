"""The configured batch sizes, checked through properties that do not need the CPU oracle on every frame:
transmit -> noisy channel -> receive round trip (every frame whose FCS passes must carry exactly the
transmitted MPDU, and at 30 dB nearly all must pass), idempotence of a second call, and the oracle on a
sample of frames."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,n_oracle", [("l_mcs0_1000B_snr30", 12), ("vht_mcs4_1500B_snr30", 24),
                                               ("ht_mcs15_2x2_1500B_snr30", 24), ("vht_mcs8_1500B_snr30", 24)])
def test_configured_batch_round_trip(pkg, rxctx, oracle, workload, n_oracle):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    n = bench.DEFAULT_FRAMES[workload]
    arenas, sync, navail, exp, idx, L = bench.build_corpus(torch, pkg.txgen, workload, n, 64, 0, dev)
    a0, a1 = arenas[0], (arenas[1] if len(arenas) > 1 else None)
    desc = torch.from_numpy(rxctx.make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
    blob_len = exp.shape[1]
    stride = ((blob_len + 4 + 63) // 64) * 64
    outs = []
    for rep in range(2):
        out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
        out_len = torch.zeros(n, dtype=torch.int32, device=dev)
        n_pdu = torch.zeros(n, dtype=torch.int32, device=dev)
        rxctx.rx_batch_into(a0, a1, desc, n, out, out_len, n_pdu)
        rxctx.sync()
        outs.append((out, out_len, n_pdu))
    out, out_len, n_pdu = outs[0]
    ok = n_pdu == 1
    match = (out[:, :blob_len] == exp[idx]).all(dim=1)
    assert int((ok & ~match).sum().item()) == 0          # FCS pass => the transmitted MPDU, bit for bit
    assert int((n_pdu > 1).sum().item()) == 0
    assert int(ok.sum().item()) >= int(0.99 * n)         # 30 dB: nearly every frame decodes
    assert bool((out_len[ok] == blob_len).all().item())
    assert bool((out_len[~ok] == 0).all().item())
    # same input, same output
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][0][ok], outs[1][0][ok])
    # the oracle on a sample spread over the batch (includes frames that fail, if any are in the sample)
    # ... and the soft bits of the same frames out of one gr80211_demod_batch over the whole batch: the tolerance the
    # path states for them (1e-4 relative, tests/util.py), at the configured batch size
    from util import LLR_ATOL, LLR_RTOL
    pick = np.unique(np.linspace(0, n - 1, n_oracle).astype(np.int64))
    lens = out_len.cpu().numpy()
    info, read_llr = rxctx.demod_batch_device(a0, a1, desc, n)
    for i in pick:
        s, e = int(sync[i]) - 176, int(sync[i]) + int(navail[i])
        h0 = a0[s:e].cpu().numpy()
        h1 = a1[s:e].cpu().numpy() if a1 is not None else None
        ref = oracle.rx_frame(h0, h1, 176, 0.0, want_llr=True)
        got = bytes(out[i, : lens[i]].cpu().numpy())
        assert got == ref["blob"], (workload, int(i))
        assert int(info["status"][i]) == ref["status"] == 0 and int(info["total"][i]) == len(ref["llr"]), (workload, int(i))
        g = read_llr(int(info["llr_off"][i]), int(info["total"][i]))
        assert len(g) > 10000
        assert np.all(np.abs(g - ref["llr"]) <= LLR_ATOL + LLR_RTOL * np.abs(ref["llr"])), (workload, int(i))


@pytest.mark.parametrize("two_ant,n_frames", [(False, 125000), (False, 52000), (True, 20000)])
def test_mixed_capture_corpus_replicated(pkg, rxctx, oracle, two_ant, n_frames):
    """Mixed L/HT/VHT corpus (the reference's IQ fixtures + the Python-reference frames) replicated round-robin
    to a full shard: every replica must give exactly what the oracle gives for its source frame.  Exercises the
    group scheduler with many (code rate, length) classes and partly filled groups, in the three regimes of the
    decoder (more groups than SIMDs, at most one group per SIMD, four waves per group)."""
    import torch
    from util import H22, golden_frames, ref_fixtures
    dev = torch.device("cuda", 0)
    srcs = []  # (antenna arrays, sync index, rad)
    if two_ant:
        for f in golden_frames():
            if f["nss"] == 2:
                rx = (H22 @ f["iq"]).astype(np.complex64)
                srcs.append(([rx[0], rx[1]], f["sync_idx"], f["rad"]))
        name, ants, sidx = ref_fixtures()[1]
        srcs.append((ants, sidx, 0.0))
    else:
        for f in golden_frames():
            if f["nss"] == 1:
                srcs.append(([f["iq"][0].astype(np.complex64)], f["sync_idx"], f["rad"]))
        name, ants, sidx = ref_fixtures()[0]
        srcs.append((ants, sidx, 0.0))
    nant = 2 if two_ant else 1
    blk = [np.concatenate([s[0][a] for s in srcs]) for a in range(nant)]
    blk_len = len(blk[0])
    offs, pos = [], 0
    for s in srcs:
        offs.append(pos + s[1])
        pos += len(s[0][0])
    navail_u = [len(s[0][0]) - s[1] for s in srcs]
    # what the oracle says for each source frame, restricted like the descriptor restricts the GPU
    want = []
    for k, s in enumerate(srcs):
        end = offs[k] + navail_u[k]
        r = oracle.rx_frame(blk[0][:end], blk[1][:end] if two_ant else None, offs[k], s[2], want_llr=False)
        want.append(r["blob"])
    assert sum(1 for w in want if len(w) > 0) == len(srcs)
    u = len(srcs)
    reps = (n_frames + u - 1) // u
    arenas = [torch.cat([torch.from_numpy(b).to(dev).repeat(reps), torch.zeros(64, dtype=torch.complex64, device=dev)])
              for b in blk]
    k = np.arange(n_frames)
    sync = (k // u).astype(np.uint64) * np.uint64(blk_len) + np.array(offs, dtype=np.uint64)[k % u]
    navail = np.array(navail_u, dtype=np.uint32)[k % u]
    rads = np.array([s[2] for s in srcs], dtype=np.float32)[k % u]
    desc = torch.from_numpy(rxctx.make_desc(sync, navail, rad=rads).view(np.uint8).reshape(-1)).to(dev)
    stride = 1536
    out = torch.zeros((n_frames, stride), dtype=torch.uint8, device=dev)
    out_len = torch.zeros(n_frames, dtype=torch.int32, device=dev)
    n_pdu = torch.zeros(n_frames, dtype=torch.int32, device=dev)
    rxctx.rx_batch_into(arenas[0], arenas[1] if two_ant else None, desc, n_frames, out, out_len, n_pdu)
    rxctx.sync()
    exp = torch.zeros((u, stride), dtype=torch.uint8)
    exp_len = torch.zeros(u, dtype=torch.int32)
    for i, w in enumerate(want):
        exp[i, : len(w)] = torch.frombuffer(bytearray(w), dtype=torch.uint8)
        exp_len[i] = len(w)
    idx = torch.from_numpy(k % u).to(dev)
    assert torch.equal(out_len, exp_len.to(dev)[idx])
    assert torch.equal(out, exp.to(dev)[idx])
    assert int(n_pdu.sum().item()) == n_frames


def test_mixed_workload_of_the_benchmark(pkg, rxctx, oracle):
    """The benchmark's own mixed corpus at its configured size (125 000 legacy / HT / VHT frames of nine rates and
    lengths in rotation, made by the GPU transmitter, a different payload in every frame, 30 dB): every frame whose
    FCS passes carries exactly its MPDU, nearly all pass, and the oracle agrees on a sample of every kind."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    n = bench.DEFAULT_FRAMES[bench.MIXED]
    arenas, sync, navail, exp, idx, L, tx_info, extra = bench.build_corpus_mixed(torch, rxctx, pkg.txgen, n, 0, dev)
    desc = torch.from_numpy(rxctx.make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
    blob_len = exp.shape[1]
    stride = ((blob_len + 4 + 63) // 64) * 64
    out = torch.zeros((n, stride), dtype=torch.uint8, device=dev)
    out_len = torch.zeros(n, dtype=torch.int32, device=dev)
    n_pdu = torch.zeros(n, dtype=torch.int32, device=dev)
    rxctx.rx_batch_into(arenas[0], None, desc, n, out, out_len, n_pdu)
    rxctx.sync()
    ok = n_pdu == 1
    match = (out[:, :blob_len] == exp).all(dim=1)
    assert int((ok & ~match).sum().item()) == 0 and int((n_pdu > 1).sum().item()) == 0
    assert int(ok.sum().item()) >= int(0.999 * n)
    kinds = len(bench.MIXED_KINDS)
    want_len = torch.tensor([k[2] + 4 for k in bench.MIXED_KINDS], dtype=torch.int32, device=dev)[torch.arange(n, device=dev) % kinds]
    assert bool((out_len[ok] == want_len[ok]).all().item()) and bool((out_len[~ok] == 0).all().item())
    lens = out_len.cpu().numpy()
    a0 = arenas[0]
    for i in list(range(kinds)) + list(range(n - 2 * kinds, n, 2)) + [n // 2, n // 3]:
        s, e = int(sync[i]) - 176, int(sync[i]) + int(navail[i])
        ref = oracle.rx_frame(a0[s:e].cpu().numpy(), None, 176, 0.0, want_llr=False)
        assert bytes(out[i, : lens[i]].cpu().numpy()) == ref["blob"], int(i)


def test_raw_capture_at_the_headline_size(pkg, rxctx, oracle):
    """gr80211_rx_capture (detection -> trigger search -> RX) over the headline corpus as one raw capture: 100 000 VHT
    MCS8 frames back to back, 392 M samples.  Every frame the trigger search keeps is decoded as the oracle chain
    decodes it from the detected position (a sample of them is run), FCS passes carry the transmitted MPDU, and
    the count of frames is the corpus's."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    wl = "vht_mcs8_1500B_snr30"
    n = bench.DEFAULT_FRAMES[wl]
    arenas, sync, navail, exp, idx, L, tx_info = bench.build_corpus_gpu(torch, rxctx, pkg.txgen, wl, n, 0, dev)
    blob_len = exp.shape[1]
    stride = ((blob_len + 4 + 63) // 64) * 64
    r = rxctx.rx_capture(arenas[0], None, max_frames=2 * n, out_stride=stride)
    rxctx.sync()
    nf = r["n"]
    assert abs(nf - n) <= n // 500
    d = r["desc"][: nf * 32].cpu().numpy().view(pkg.rx.FRAME_DESC_DTYPE)
    so = d["sync_off"].astype(np.int64)
    fidx = np.clip(so // L, 0, n - 1)
    okd = (r["n_pdu"][:nf] == 1)
    md = (r["out"][:nf, :blob_len] == exp[torch.from_numpy(fidx).to(dev)]).all(dim=1)
    assert int((okd & ~md).sum().item()) == 0
    assert int(okd.sum().item()) >= int(0.998 * n)
    # the positions found are the nominal ones within the cyclic-prefix margin
    assert np.abs(so - (fidx * L + 176)).max() <= 8
    lens = r["out_len"][:nf].cpu().numpy()
    host = arenas[0]
    for j in np.unique(np.linspace(0, nf - 1, 40).astype(np.int64)):
        s = int(so[j]) - 176 - 8
        e = int(so[j]) + int(d["n_avail"][j])
        e = min(e, int(so[j]) + 224 + 3920)   # what `signal` forwards of this frame: no later frame's samples
        seg = host[max(s, 0):e].cpu().numpy()
        ref = oracle.rx_frame(seg, None, int(so[j]) - max(s, 0), float(d["rad"][j]), want_llr=False)
        assert bytes(r["out"][j, : lens[j]].cpu().numpy()) == ref["blob"], int(j)
