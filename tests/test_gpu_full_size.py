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


@pytest.mark.parametrize("workload,n_oracle", [("vht_mcs4_1500B_snr30", 24), ("ht_mcs15_2x2_1500B_snr30", 24),
                                               ("vht_mcs8_1500B_snr30", 24)])
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
    pick = np.unique(np.linspace(0, n - 1, n_oracle).astype(np.int64))
    lens = out_len.cpu().numpy()
    for i in pick:
        s, e = int(sync[i]) - 176, int(sync[i]) + int(navail[i])
        h0 = a0[s:e].cpu().numpy()
        h1 = a1[s:e].cpu().numpy() if a1 is not None else None
        ref = oracle.rx_frame(h0, h1, 176, 0.0, want_llr=False)
        got = bytes(out[i, : lens[i]].cpu().numpy())
        assert got == ref["blob"], (workload, int(i))
