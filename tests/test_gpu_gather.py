"""The exchange step of the multi-GPU receiver on one GPU: the packing kernel against its host statement, and the
gathers themselves in a process group of one rank over RCCL (device tensors, stream ordering, rotation of slots)."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,stride", [(1, 16), (50, 64), (1024, 1536), (5000, 256), (70000, 64)])
def test_pack_kernel_against_host_statement(pkg, rxctx, n, stride):
    import torch
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    dev = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(n + stride)
    out = torch.randint(0, 256, (n, stride), dtype=torch.uint8, generator=gen)
    lens = torch.randint(0, stride + 1, (n,), dtype=torch.int32, generator=gen)
    lens[::5] = 0
    lens[-1] = stride
    cap = n * stride
    want = torch.zeros(cap, dtype=torch.uint8)
    want_total = torch.zeros(1, dtype=torch.int64)
    shard.pack_pdus_host(out, lens, want, want_total)
    packed = torch.full((cap,), 0xEE, dtype=torch.uint8, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    off = torch.zeros(n, dtype=torch.int64, device=dev)
    rxctx.pack_pdus(out.to(dev), lens.to(dev), packed, total, off)
    rxctx.sync()
    t = int(total.item())
    assert t == int(want_total.item())
    assert torch.equal(packed[:t].cpu(), want[:t])
    assert bool((packed[t:] == 0xEE).all().item())          # nothing written past the run
    sizes = shard.pad16(lens.to(torch.int64))
    assert torch.equal(off.cpu(), torch.cumsum(sizes, 0) - sizes)
    rows = shard.unpack_pdus(packed[:t], lens.to(dev), stride).cpu()
    keep = torch.arange(stride)[None, :] < lens[:, None]
    assert torch.equal(rows, torch.where(keep, out, torch.zeros((), dtype=torch.uint8)))


def test_pack_refuses_a_run_that_is_too_small(pkg, rxctx):
    import torch
    dev = torch.device("cuda", 0)
    out = torch.full((8, 64), 7, dtype=torch.uint8, device=dev)
    lens = torch.full((8,), 40, dtype=torch.int32, device=dev)
    packed = torch.zeros(3 * 48, dtype=torch.uint8, device=dev)
    total = torch.zeros(1, dtype=torch.int64, device=dev)
    rxctx.pack_pdus(out, lens, packed, total)
    rxctx.sync()
    assert int(total.item()) == 8 * 48          # what it would have taken
    assert bool((packed.view(3, 48)[:, :40] == 7).all().item()) and bool((packed.view(3, 48)[:, 40:] == 0).all().item())
    with pytest.raises(RuntimeError):
        rxctx.pack_pdus(out[:, :40], lens, packed, total)   # stride not a multiple of 16


def test_gathers_in_a_group_of_one_rank_over_rccl(pkg):
    """Both forms of the MPDU gather with the nccl backend at world size 1, in a process of its own (the process
    group lives and dies with it): slots rotate, every batch arrives as it was sent."""
    import subprocess
    code = r'''
import importlib, os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pkg = importlib.import_module("fork-gr-ieee80211_amd"); shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
ctx = pkg.RxContext(0)
n, stride, depth = 3000, 128, 3
gen = torch.Generator().manual_seed(5)
def batch(step):
    g = torch.Generator().manual_seed(step)
    o = torch.randint(0, 256, (n, stride), dtype=torch.uint8, generator=g)
    l = torch.randint(0, stride + 1, (n,), dtype=torch.int32, generator=g); l[::3] = 0
    return o, l
with torch.cuda.stream(ctx.own_stream()):
    for form in ("fixed", "compact"):
        g = (shard.PduGather(n, stride, dev, depth=depth) if form == "fixed"
             else shard.PduGatherCompact(n, stride, dev, ctx.pack_pdus, depth=depth))
        outs = [(torch.zeros((n, stride), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(depth)]
        seen = 0
        def check(step, res):
            o, l = batch(step)
            keep = torch.arange(stride)[None, :] < l[:, None]
            want = torch.where(keep, o, torch.zeros((), dtype=torch.uint8))
            lens = res[1].cpu()
            assert torch.equal(lens, l), (form, step)
            rows = (shard.unpack_pdus(res[0][0], res[1], stride) if form == "compact" else torch.where(keep.to(dev), res[0], torch.zeros((), dtype=torch.uint8, device=dev))).cpu()
            assert torch.equal(rows, want), (form, step)
        for step in range(7):
            slot = step %% depth
            res = g.wait(slot)
            if res[0] is not None:
                check(step - depth, res); seen += 1
            o, l = batch(step)
            outs[slot][0].copy_(o, non_blocking=True); outs[slot][1].copy_(l, non_blocking=True)
            g.start(slot, outs[slot][0], outs[slot][1])
        for step in range(7, 7 + depth):
            res = g.wait(step %% depth)
            if res[0] is not None:
                check(step - depth, res); seen += 1
        assert seen == 7, (form, seen)
torch.cuda.synchronize()
dist.barrier(); dist.destroy_process_group(); ctx.close()
print("gathers ok")
''' % ROOT
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "gathers ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
