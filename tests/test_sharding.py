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
