"""Multi-GPU sharding of a frame corpus: contiguous frame ranges per rank, one gather of the decoded
MPDUs to rank 0 (the only exchange step of the path)."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Frames [lo, hi) of rank `rank`: frame i belongs to rank floor(i*world/n_total), so the global
    order is rank 0's frames, then rank 1's, ..."""
    lo = (n_total * rank + world - 1) // world
    hi = (n_total * (rank + 1) + world - 1) // world
    return lo, hi


def gather_pdus(out, out_len, dst=0, group=None):
    """Collect every rank's PDU slots on rank `dst`.

    out: uint8 [n_local, stride], out_len: int32 [n_local] (equal n_local on every rank, weak scaling).
    With the nccl backend (RCCL over xGMI) the tensors stay on the GPU; with gloo they are staged
    through host memory with synchronous copies.  Returns (out_all [world*n_local, stride],
    out_len_all) on dst, (None, None) elsewhere.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    backend = dist.get_backend(group)
    if backend == "gloo" and out.is_cuda:
        out_s, len_s = out.cpu(), out_len.cpu()
    else:
        out_s, len_s = out, out_len
    if rank == dst:
        bufs = [torch.empty_like(out_s) for _ in range(world)]
        lens = [torch.empty_like(len_s) for _ in range(world)]
    else:
        bufs = lens = None
    dist.gather(out_s, bufs, dst=dst, group=group)
    dist.gather(len_s, lens, dst=dst, group=group)
    if rank != dst:
        return None, None
    return torch.cat(bufs, dim=0), torch.cat(lens, dim=0)
