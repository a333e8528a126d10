"""N > 1 path on CPU: frame sharding and the MPDU gather with world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    import importlib
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.shard_range(n_total, rank, world)
    n = hi - lo
    stride = 16
    out = torch.zeros((n, stride), dtype=torch.uint8)
    out_len = torch.zeros(n, dtype=torch.int32)
    for i in range(n):
        g = lo + i
        out[i, 0] = g % 251
        out[i, 1] = rank
        out_len[i] = 2 + (g % 5)
    allo, alll = shard.gather_pdus(out, out_len, dst=0)
    if rank == 0:
        ok = allo.shape[0] == n_total and all(int(allo[g, 0]) == g % 251 for g in range(n_total)) and \
            all(int(alll[g]) == 2 + (g % 5) for g in range(n_total))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def _worker_pipeline(rank, world, port, n_local, q):
    """Three batches through the double-buffered asynchronous gather."""
    sys.path.insert(0, ROOT)
    import importlib
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stride, depth = 8, 2
    g = shard.PduGather(n_local, stride, torch.device("cpu"), dst=0, depth=depth)
    outs = [(torch.zeros((n_local, stride), dtype=torch.uint8), torch.zeros(n_local, dtype=torch.int32))
            for _ in range(depth)]
    ok = True
    seen = []
    for step in range(3):
        slot = step % depth
        res = g.wait(slot)
        if res[0] is not None:
            seen.append((step - depth, res[0].clone(), res[1].clone()))
        o, ol = outs[slot]
        o[:, 0] = rank
        o[:, 1] = step
        o[:, 2] = torch.arange(n_local, dtype=torch.uint8)
        ol[:] = 3 + step
        g.start(slot, o, ol)
    for slot in ((3 % depth), ((3 + 1) % depth)):
        res = g.wait(slot)
        if res[0] is not None:
            seen.append((None, res[0].clone(), res[1].clone()))
    if rank == 0:
        ok = len(seen) == 3
        steps_seen = sorted(int(a[0, 1]) for _, a, _ in seen)
        ok = ok and steps_seen == [0, 1, 2]
        for _, a, l in seen:
            st = int(a[0, 1])
            for r in range(world):
                blk = a[r * n_local:(r + 1) * n_local]
                ok = ok and bool((blk[:, 0] == r).all()) and bool((blk[:, 1] == st).all())
                ok = ok and bool((blk[:, 2] == torch.arange(n_local, dtype=torch.uint8)).all())
            ok = ok and bool((l == 3 + st).all())
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_pipeline, args=(r, 2, port, 40, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ranges_cover_everything(pkg):
    import importlib
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    for n, w in ((10, 2), (1000000, 8), (7, 8), (64, 4), (100001, 8)):
        seen = []
        for r in range(w):
            lo, hi = shard.shard_range(n, r, w)
            assert 0 <= lo <= hi <= n
            seen += list(range(lo, hi)) if n < 2000 else [lo, hi]
        if n < 2000:
            assert seen == list(range(n))
        else:
            assert seen[0] == 0 and seen[-1] == n and all(seen[2 * i + 1] == seen[2 * i + 2] for i in range(w - 1))


def test_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    n_total = 64   # equal shards (weak scaling)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_compact(rank, world, port, n_local, q):
    """Three batches of blobs of unequal lengths through the length-compacted gather: rank 0 must end up with
    every rank's blobs, byte for byte, and with no more bytes than the padded lengths add up to."""
    sys.path.insert(0, ROOT)
    import importlib
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    stride, depth = 64, 2
    g = shard.PduGatherCompact(n_local, stride, torch.device("cpu"), shard.pack_pdus_host, dst=0, depth=depth)

    def batch(r, step):
        gen = torch.Generator().manual_seed(1000 * r + step)
        out = torch.randint(0, 256, (n_local, stride), dtype=torch.uint8, generator=gen)
        lens = torch.randint(0, stride + 1, (n_local,), dtype=torch.int32, generator=gen)
        lens[::7] = 0  # frames without a PDU
        if r == 1 and step == 1:
            lens[:] = 0  # a rank with nothing to send
        return out, lens

    ok, seen = True, []
    for step in range(3):
        slot = step % depth
        res = g.wait(slot)
        if res[0] is not None:
            seen.append((step - depth, [v.clone() for v in res[0]], res[1].clone()))
        o, ol = batch(rank, step)
        g.start(slot, o, ol)
    for step in (3, 4):
        res = g.wait(step % depth)
        if res[0] is not None:
            seen.append((step - depth, [v.clone() for v in res[0]], res[1].clone()))
    if rank == 0:
        ok = sorted(s for s, _, _ in seen) == [0, 1, 2]
        for step, runs, lens_all in seen:
            for r in range(world):
                o, ol = batch(r, step)
                lens = lens_all[r * n_local:(r + 1) * n_local]
                ok = ok and torch.equal(lens, ol)
                ok = ok and runs[r].numel() == int(shard.pad16(ol.to(torch.int64)).sum())
                rows = shard.unpack_pdus(runs[r], lens, stride)
                keep = torch.arange(stride)[None, :] < ol[:, None]
                ok = ok and torch.equal(rows, torch.where(keep, o, torch.zeros((), dtype=torch.uint8)))
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_compact_gather_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_compact, args=(r, 2, port, 50, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher and without WORLD_SIZE starts torch.distributed.run itself,
    before anything touches a GPU.  No GPU here: the ranks get as far as selecting a device and fail there, and
    the parent leaves with the launcher's (non-zero) exit code instead of the old refusal."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--frames", "64"], env=env, capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available():
        assert r.returncode == 0, r.stderr[-2000:]
        return
    assert r.returncode != 0
    assert "must be launched with" not in (r.stdout + r.stderr)
    assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "rank" in r.stderr.lower()
