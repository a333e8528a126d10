"""gr80211_decode_batch_noalloc, the entry point that allocates nothing (its work buffers lie in a fixed array of the
library's code object), against gr80211_decode_batch on the same soft bits: decoded bits, PDUs, lengths and counts
equal byte for byte, on two calls in a row, the second smaller than the first (the array keeps the first call's
contents), then a third call on a second context of the same GPU, which shares the array.  Batches beyond the stated
limits are refused before anything runs."""
import numpy as np
import pytest

from test_gpu_decode import coded_frame

pytestmark = pytest.mark.gpu


def _batch(tx, rng, kinds):
    return [coded_frame(tx, rng, fmt, cr, nbytes, mcs=mcs, noise=noise) for fmt, cr, nbytes, mcs, noise in kinds]


def _pack(rx, frames):
    desc = np.zeros(len(frames), dtype=rx.DEC_DESC_DTYPE)
    parts, off = [], 0
    for i, f in enumerate(frames):
        desc[i] = (off, len(f["llr"]), f["fmt"], f["len"], f["cr"], f["mcs"], f["ampdu"], f["trellis"], 0)
        parts.append(f["llr"])
        off += len(f["llr"])
    return desc, np.concatenate(parts)


def _compare(pkg, ctx_ref, ctx_na, frames, bufs):
    import torch
    desc, llr_h = _pack(pkg.rx, frames)
    n = len(frames)
    llr = torch.from_numpy(llr_h).cuda()
    want = ctx_ref.decode_batch(llr, desc, n, want_bits=True)
    ctx_ref.sync()
    out, out_len, n_pdu, bits = bufs
    for t in bufs:
        t.fill_(0x5A if t.dtype == torch.uint8 else -7)  # whatever an earlier call left must not show through
    bits.zero_()  # decode_batch hands out zeroed bit slots too: bits past a frame's trellis are never written
    ctx_na.decode_batch_noalloc(llr, desc, n, out, out_len, n_pdu, bits)
    ctx_na.sync()
    wl, wn = want["out_len"].cpu().numpy(), want["n_pdu"].cpu().numpy()
    gl, gn = out_len[:n].cpu().numpy(), n_pdu[:n].cpu().numpy()
    assert np.array_equal(gl, wl) and np.array_equal(gn, wn)
    wo, go = want["out"].cpu().numpy(), out[:n].cpu().numpy()
    for i in range(n):
        assert bytes(go[i, : gl[i]]) == bytes(wo[i, : wl[i]]), i
    wb, gb = want["bits"].cpu().numpy(), bits[:n].cpu().numpy()
    for i, f in enumerate(frames):
        nb = (f["trellis"] + 7) // 8
        assert np.array_equal(gb[i, :nb], wb[i, :nb]), i
    return int(gn.sum())


def test_no_alloc_entry_equals_decode_batch(pkg):
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(77)
    # first call: every code rate, three formats, ragged lengths, some frames too noisy to pass their FCS
    first = _batch(tx, rng, [(fmt, cr, int(rng.integers(40, 900)), int(rng.integers(0, 8)), noise)
                             for fmt in (0, 1, 2) for cr in range(4) for noise in (1.0, 1.0, 1.0, 4.5) for _ in range(4)])
    # second call: fewer and shorter frames, so that pieces of the first call's records, lanes and survivor words lie
    # behind and between what this call writes
    second = _batch(tx, rng, [(fmt, cr, int(rng.integers(20, 300)), 3, 1.0) for fmt in (2, 0) for cr in (2, 0, 3)
                              for _ in range(5)])
    third = _batch(tx, rng, [(1, 1, 1500, 7, 1.0)] * 70 + [(2, 2, 64, 2, 1.0)] * 3)
    cap = max(len(first), len(second), len(third))
    bufs = (torch.zeros((cap, 4104), dtype=torch.uint8, device="cuda"), torch.zeros(cap, dtype=torch.int32, device="cuda"),
            torch.zeros(cap, dtype=torch.int32, device="cuda"), torch.zeros((cap, 4104), dtype=torch.uint8, device="cuda"))
    ref, na, na2 = pkg.RxContext(0), pkg.RxContext(0), pkg.RxContext(0)
    try:
        ok1 = _compare(pkg, ref, na, first, bufs)
        ok2 = _compare(pkg, ref, na, second, bufs)
        assert ok1 >= len(first) * 3 // 4 - 8 and ok2 == len(second)
        assert _compare(pkg, ref, na2, third, bufs) == len(third)   # another context, the same array
        assert _compare(pkg, ref, na, second, bufs) == len(second)  # and the first context again
        # the ordinary entry point of the same context still works, with its own buffers
        desc, llr_h = _pack(pkg.rx, second)
        res = na.decode_batch(torch.from_numpy(llr_h).cuda(), desc, len(second))
        na.sync()
        assert int(res["n_pdu"].sum().item()) == len(second)
    finally:
        for c in (ref, na, na2):
            c.close()


def test_no_alloc_entry_refuses_what_does_not_fit(pkg):
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(78)
    ctx = pkg.RxContext(0)
    try:
        one = coded_frame(tx, rng, 0, 0, 60)
        bufs = (torch.zeros((1100, 256), dtype=torch.uint8, device="cuda"), torch.zeros(1100, dtype=torch.int32, device="cuda"),
                torch.zeros(1100, dtype=torch.int32, device="cuda"))
        # more than 1024 frames
        desc, llr_h = _pack(pkg.rx, [one] * 1025)
        llr = torch.from_numpy(llr_h).cuda()
        with pytest.raises(RuntimeError, match="at most 1024 frames"):
            ctx.decode_batch_noalloc(llr, desc, 1025, *bufs)
        # 1024 frames are fine
        ctx.decode_batch_noalloc(llr, desc[:1024], 1024, *bufs)
        ctx.sync()
        assert int(bufs[2][:1024].sum().item()) == 1024
        # survivor words beyond the array: 8 groups of 1500-byte frames at one rate (descriptors only: the check
        # comes before anything looks at the soft bits, which are all there)
        big = coded_frame(tx, rng, 1, 2, 1500, mcs=4)
        desc, llr_h = _pack(pkg.rx, [big])
        desc = np.repeat(desc, 8 * 64)
        llr = torch.from_numpy(llr_h).cuda()
        with pytest.raises(RuntimeError, match="bytes of survivor words"):
            ctx.decode_batch_noalloc(llr, desc, len(desc), *bufs)
        # seven groups fit (every frame reads the same soft bits)
        ctx.decode_batch_noalloc(llr, desc[:448], 448, torch.zeros((448, 1600), dtype=torch.uint8, device="cuda"), bufs[1], bufs[2])
        ctx.sync()
        assert int(bufs[2][:448].sum().item()) == 448
        # a descriptor outside the soft-bit arena, and device descriptors, are refused as well
        bad = desc[:4].copy()
        bad["llr_off"][2] = llr.numel()
        with pytest.raises(RuntimeError, match="outside the LLR arena"):
            ctx.decode_batch_noalloc(llr, bad, 4, *bufs)
    finally:
        ctx.close()


def test_no_alloc_entry_leaves_device_memory_as_it_is(pkg):
    """The property the entry point is named for: repeated calls on a warmed context, of growing and shrinking
    batches, leave the free device memory where it was (hipMemGetInfo through torch; gr80211_decode_batch on the
    same inputs, which grows its work buffers, is the contrast that shows the probe sees allocations)."""
    import torch
    tx = pkg.txgen
    rng = np.random.default_rng(79)
    batches = [_batch(tx, rng, [(2, 2, 1400, 8, 1.0)] * k) for k in (16, 200, 60, 440)]
    packed = [(_pack(pkg.rx, b), len(b)) for b in batches]
    llrs = [torch.from_numpy(p[0][1]).cuda() for p in packed]
    cap = 440
    bufs = (torch.zeros((cap, 1600), dtype=torch.uint8, device="cuda"), torch.zeros(cap, dtype=torch.int32, device="cuda"),
            torch.zeros(cap, dtype=torch.int32, device="cuda"))
    ctx = pkg.RxContext(0)
    try:
        (desc, _), n = packed[0]
        ctx.decode_batch_noalloc(llrs[0], desc, n, *bufs)      # first call: page-locked staging and events exist now
        ctx.sync()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        for _ in range(3):
            for ((desc, _), n), llr in zip(packed, llrs):
                ctx.decode_batch_noalloc(llr, desc, n, *bufs)
                ctx.sync()
                assert int(bufs[2][:n].sum().item()) == n
                assert torch.cuda.mem_get_info()[0] == free0
        # the allocating entry point on the largest batch does move the figure (its survivor arena alone is 43 MB)
        (desc, _), n = packed[3]
        res = ctx.decode_batch(llrs[3], desc, n, out_stride=1600)
        ctx.sync()
        assert torch.cuda.mem_get_info()[0] < free0
        del res
    finally:
        ctx.close()
