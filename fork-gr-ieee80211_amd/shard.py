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


def pad16(lens):
    """Bytes a blob of `lens` bytes takes in a packed run (gr80211_pack_pdus rounds every blob up to 16 bytes)."""
    return (lens + 15) & ~15


def pack_pdus_host(out, out_len, packed, total):
    """gr80211_pack_pdus for CPU tensors (the host-staged rehearsal of the exchange and its tests; on a GPU the
    library's kernel packs): blob i at the exclusive prefix sum of the padded lengths, zeros up to the boundary."""
    n, stride = out.shape
    lens = out_len.clamp(min=0, max=stride).to(torch.int64)
    sizes = pad16(lens)
    off = torch.cumsum(sizes, 0) - sizes
    tot = int(sizes.sum().item())
    total[0] = tot
    packed[:tot] = 0
    cols = torch.arange(stride, dtype=torch.int64)
    keep = cols[None, :] < lens[:, None]
    place = (off[:, None] + cols[None, :])[keep]
    packed[place] = out[keep]


def unpack_pdus(packed, lens, stride):
    """Inverse of the packing on the receiving side: slots [n, stride] rebuilt from a rank's packed run and its
    gathered lengths (bytes past a blob's end are zero)."""
    n = lens.numel()
    lens64 = lens.clamp(min=0, max=stride).to(torch.int64)
    sizes = pad16(lens64)
    off = torch.cumsum(sizes, 0) - sizes
    rows = torch.zeros((n, stride), dtype=torch.uint8, device=packed.device)
    if packed.numel() == 0:
        return rows  # a rank that published nothing
    cols = torch.arange(stride, dtype=torch.int64, device=packed.device)
    for lo in range(0, n, 16384):  # bounded index tensors
        hi = min(n, lo + 16384)
        keep = cols[None, :] < lens64[lo:hi, None]
        idx = (off[lo:hi, None] + cols[None, :]).clamp(max=max(packed.numel() - 1, 0))
        rows[lo:hi] = torch.where(keep, packed[idx], torch.zeros((), dtype=torch.uint8, device=packed.device))
    return rows


class PduGatherCompact:
    """The gather of decoded MPDUs with the padding left at home, for corpora of mixed lengths: every rank packs its
    blobs into one run (gr80211_pack_pdus), the lengths travel first (a gather of equal-sized int32 arrays, started
    asynchronously like PduGather's), and the bytes follow as point-to-point transfers of each rank's own size once
    the lengths are known -- which is when the slot comes up for reuse, `depth` batches later, so nothing waits on
    the batch that has just been queued.  Same start / wait / finish protocol as PduGather.

    pack(out, out_len, packed, total) queues the packing on the current stream (RxContext.pack_pdus on a GPU,
    pack_pdus_host for CPU tensors).  On dst, wait() returns ([packed run of rank r, cut to its size], lengths of
    all ranks [world * n_local]); elsewhere (None, None).
    """

    def __init__(self, n_local, stride, device, pack, dst=0, group=None, depth=2):
        assert stride % 16 == 0, "slots of a multiple of 16 bytes"
        self.group, self.dst, self.depth, self.pack = group, dst, depth, pack
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.host_staged = dist.get_backend(group) == "gloo"
        self.n_local, self.stride = n_local, stride
        self.dev = device
        cdev = torch.device("cpu") if self.host_staged else device
        cap = n_local * stride
        self.packed = [torch.zeros(cap, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.total = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(depth)]
        self.packed_all, self.len_all = [], []
        if self.rank == dst:
            for _ in range(depth):
                self.packed_all.append([torch.zeros(cap, dtype=torch.uint8, device=cdev) for _ in range(self.world)])
                self.len_all.append(torch.zeros(self.world * n_local, dtype=torch.int32, device=cdev))
        self.pending = [None] * depth
        self.bytes_moved = 0  # payload bytes this rank sent (or, on dst, received) so far

    def start(self, slot, out, out_len):
        assert self.pending[slot] is None, "wait(slot) before reusing a slot"
        assert out.shape == (self.n_local, self.stride) and out_len.shape == (self.n_local,)
        self.pack(out, out_len, self.packed[slot], self.total[slot])
        src = out_len.cpu() if (self.host_staged and out_len.is_cuda) else out_len
        lens = None
        if self.rank == self.dst:
            n = self.n_local
            lens = [self.len_all[slot][r * n:(r + 1) * n] for r in range(self.world)]
        h = dist.gather(src, lens, dst=self.dst, group=self.group, async_op=True)
        self.pending[slot] = (h, src)

    def wait(self, slot):
        p = self.pending[slot]
        if p is None:
            return None, None
        p[0].wait()
        self.pending[slot] = None
        mine = int(self.total[slot].item())  # the batch that filled the slot has run by now
        run = self.packed[slot][:mine]
        if self.host_staged and run.is_cuda:
            run = run.cpu()
        if self.rank != self.dst:
            if mine:
                dist.send(run, self.dst, group=self.group)
                self.bytes_moved += mine
            return None, None
        sizes = pad16(self.len_all[slot].clamp(min=0, max=self.stride).to(torch.int64)).view(self.world, -1).sum(1).tolist()
        ops, views = [], []
        for r in range(self.world):
            v = self.packed_all[slot][r][:int(sizes[r])]
            views.append(v)
            if r == self.rank:
                assert int(sizes[r]) == mine
                v.copy_(run)
            elif sizes[r]:
                ops.append(dist.P2POp(dist.irecv, v, r, self.group))
                self.bytes_moved += int(sizes[r])
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return views, self.len_all[slot]

    def finish(self):
        last = (None, None)
        for s in range(self.depth):
            r = self.wait(s)
            if r[0] is not None:
                last = r
        return last
