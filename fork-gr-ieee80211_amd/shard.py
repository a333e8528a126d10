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


class PduGather:
    """The gather of decoded MPDUs as a pipeline stage: persistent destination buffers on `dst`, the
    collective started asynchronously so that it overlaps the next batch's kernels, `depth` batches in
    flight.  The caller rotates its output tensors with the same depth and calls wait(slot) before it
    lets a batch overwrite the tensors of slot `slot`.

    With nccl (RCCL over xGMI) everything stays on the GPU and the collective is ordered after the
    kernels already queued on the current stream.  gloo gets CPU tensors only (synchronous copies).
    """

    def __init__(self, n_local, stride, device, dst=0, group=None, depth=2):
        self.group, self.dst, self.depth = group, dst, depth
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.n_local, self.stride = n_local, stride
        dev = torch.device("cpu") if self.host_staged else device
        self.out_all, self.len_all = [], []
        if self.rank == dst:
            for _ in range(depth):
                self.out_all.append(torch.empty((self.world * n_local, stride), dtype=torch.uint8, device=dev))
                self.len_all.append(torch.empty(self.world * n_local, dtype=torch.int32, device=dev))
        self.pending = [None] * depth

    def start(self, slot, out, out_len):
        """Begin gathering this rank's batch (out [n_local, stride] u8, out_len [n_local] i32) into slot."""
        assert self.pending[slot] is None, "wait(slot) before reusing a slot"
        assert out.shape == (self.n_local, self.stride) and out_len.shape == (self.n_local,)
        if self.host_staged:
            out, out_len = out.cpu(), out_len.cpu()
        bufs = lens = None
        if self.rank == self.dst:
            n = self.n_local
            bufs = [self.out_all[slot][r * n:(r + 1) * n] for r in range(self.world)]
            lens = [self.len_all[slot][r * n:(r + 1) * n] for r in range(self.world)]
        h0 = dist.gather(out, bufs, dst=self.dst, group=self.group, async_op=True)
        h1 = dist.gather(out_len, lens, dst=self.dst, group=self.group, async_op=True)
        self.pending[slot] = (h0, h1, out, out_len)  # the inputs stay referenced until the collective is done

    def wait(self, slot):
        """Make the current stream (nccl) or the host (gloo) wait for slot's gather.  Returns
        (out_all, out_len_all) on dst, (None, None) elsewhere or if nothing was pending."""
        p = self.pending[slot]
        if p is None:
            return None, None
        p[0].wait()
        p[1].wait()
        self.pending[slot] = None
        if self.rank != self.dst:
            return None, None
        return self.out_all[slot], self.len_all[slot]

    def finish(self):
        last = (None, None)
        for s in range(self.depth):
            r = self.wait(s)
            if r[0] is not None:
                last = r
        return last
