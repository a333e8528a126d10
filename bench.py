#!/usr/bin/env python3
"""Benchmark of the MI355X RX hot path: RX Msamples/s (demod + Viterbi) on synthetic 20 MHz frames.

    python bench.py --gpus N --steps K --warmup W

One rank per GPU (torch.distributed / RCCL when N > 1; RANK, LOCAL_RANK, WORLD_SIZE, MASTER_* from the
environment).  A step = one pass of the whole RX path (signal -> demod -> decode) over the rank's
batch of frames, IQ already resident in HBM; for N > 1 the step ends with the gather of the decoded
MPDUs on rank 0.  Rank 0 prints one JSON line.

Workload (default): BASELINE.json configs[2], the configuration the target is quoted on -- VHT MCS8
(256-QAM r=3/4) SISO 20 MHz, 1500-byte A-MPDU (1496-byte MPDU), AWGN SNR 30 dB, 100 000 frames per GPU.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (format, mcs, nss, mpdu bytes, snr dB, algorithmic bytes per frame = 8*nRx*N_samp + L_psdu + 64)
    "vht_mcs8_1500B_snr30": (2, 8, 1, 1496, 30.0, 32924),
    "vht_mcs4_1500B_snr30": (2, 4, 1, 1496, 30.0, 57884),
    "l_mcs0_1000B_snr30": (0, 0, 1, 1000, 30.0, 218664),
    "ht_mcs15_2x2_1500B_snr30": (1, 15, 2, 1500, 30.0, 45084),
}
# frames per GPU the configuration list quotes for each workload, and trellis steps per frame
DEFAULT_FRAMES = {"vht_mcs8_1500B_snr30": 100000, "vht_mcs4_1500B_snr30": 10000, "l_mcs0_1000B_snr30": 10000,
                  "ht_mcs15_2x2_1500B_snr30": 50000}
TRELLIS_STEPS = {"vht_mcs8_1500B_snr30": 12168, "vht_mcs4_1500B_snr30": 12168, "l_mcs0_1000B_snr30": 8022,
                 "ht_mcs15_2x2_1500B_snr30": 12022}
H22 = ((1.0, 0.5j), (0.4, -1.0))  # fixed well-conditioned 2x2 channel of the two-antenna workload
HBM_PEAK_GBS = 8000.0


def build_corpus(torch, tx, wl, n_frames, n_unique, seed, dev):
    """Returns ([arena per antenna], sync, navail, expected blobs, payload index per frame, samples per frame)."""
    fmt, mcs, nss, nbytes, snr_db, _ = WORKLOADS[wl]
    rng = np.random.default_rng(80211 + seed)
    sigs, blobs = [], []
    for _ in range(n_unique):
        mpdu = tx.make_mpdu(nbytes, rng)
        sig, m = tx.gen_frame(fmt, mcs, nss, mpdu)
        sig = np.asarray(sig) * (12.0 * np.sqrt(nss))
        if nss == 2:
            sig = np.array(H22, dtype=np.complex128) @ sig
        sigs.append(sig.astype(np.complex64))
        blobs.append(bytes([fmt, nbytes % 256, nbytes // 256]) + mpdu + bytes([mcs]))
    L = sigs[0].shape[1]
    stack = np.stack(sigs)  # [unique][antenna][L]
    power = float(np.mean(np.abs(stack) ** 2))
    sigma = float(np.sqrt(power / (10.0 ** (snr_db / 10.0)) / 2.0))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + seed)
    idx = torch.arange(n_frames, device=dev) % n_unique
    arenas = []
    for a in range(nss):
        uniq = torch.from_numpy(np.ascontiguousarray(stack[:, a, :])).to(dev)
        arena = torch.empty(n_frames * L + 64, dtype=torch.complex64, device=dev)
        view = torch.view_as_real(arena[: n_frames * L].view(n_frames, L))
        chunk = 8192
        for s in range(0, n_frames, chunk):
            e = min(n_frames, s + chunk)
            noise = torch.randn((e - s, L, 2), generator=gen, device=dev, dtype=torch.float32) * sigma
            view[s:e] = torch.view_as_real(uniq[idx[s:e]]) + noise
        arena[n_frames * L:] = 0
        arenas.append(arena)
    sync = (np.arange(n_frames, dtype=np.uint64) * L + 176)
    navail = np.full(n_frames, L - 176, dtype=np.uint32)
    exp = torch.from_numpy(np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(n_unique, -1).copy()).to(dev)
    return arenas, sync, navail, exp, idx, L


def cpu_baseline(arena_host, arena1_host, sync, navail, n_sample, gpu_blobs, budget_s=12.0):
    """Times the CPU oracle (scalar port of the reference path) on the first frames of the same corpus,
    first on one thread and then on every core of this process's CPU share (frames sharded over threads;
    the oracle call releases the GIL), and checks the GPU output against it frame by frame."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    O.build()
    O.lib()

    def run(lo, hi, deadline):
        n_done, samples, mism, n_ok = 0, 0, 0, 0
        for i in range(lo, hi):
            end = int(sync[i]) + int(navail[i])
            r = O.rx_frame(arena_host[:end], None if arena1_host is None else arena1_host[:end], int(sync[i]), 0.0,
                           want_llr=False)
            n_done += 1
            samples += int(navail[i]) + 176
            n_ok += r["n_pdu"]
            if r["blob"] != gpu_blobs[i]:
                mism += 1
            if time.perf_counter() > deadline:
                break
        return n_done, samples, mism, n_ok

    n1 = max(1, n_sample // 16)
    t0 = time.perf_counter()
    d1 = run(0, n1, t0 + budget_s)
    dt1 = time.perf_counter() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("OMP_NUM_THREADS", "16")), 64))
    per = (n_sample + cores - 1) // cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(lambda k: run(min(k * per, n_sample), min((k + 1) * per, n_sample), t0 + budget_s),
                            range(cores)))
    dtn = time.perf_counter() - t0
    n_done = sum(p[0] for p in parts)
    samples = sum(p[1] for p in parts)
    return dict(value=samples / dtn / 1e6, unit="Msamples/s", cores=cores, kind="port",
                sample="first %d frames of the benchmark corpus (%d samples) sharded over %d threads of the CPU "
                       "oracle, %.1f s; one thread alone: %d frames in %.1f s" % (n_done, samples, cores, dtn, d1[0], dt1),
                single_thread_value=d1[1] / dt1 / 1e6, frames=n_done + d1[0],
                fcs_ok=sum(p[3] for p in parts), gpu_mismatches=sum(p[2] for p in parts) + d1[2])


def measured_traffic(workload, n):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*_traffic.json), if they were taken on this workload and batch size."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("workload") == workload and t.get("frames") == n:
            return t.get("k_decode_hbm_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (0 = the workload's configured size)")
    ap.add_argument("--workload", default="vht_mcs8_1500B_snr30", choices=sorted(WORKLOADS))
    ap.add_argument("--unique", type=int, default=64, help="distinct payloads in the corpus")
    ap.add_argument("--cpu-frames", type=int, default=48000, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--detect", action="store_true", help="also time frame detection over the raw capture")
    ap.add_argument("--host-buffers", action="store_true", help="also time one step with PCIe transfers included")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra untimed steps with per-kernel HIP events")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (WORLD_SIZE=%d)" % (args.gpus, world))
    # GR80211_BENCH_BACKEND=gloo rehearses the N > 1 path where RCCL cannot run (several ranks on one GPU):
    # ranks then share devices round-robin and every collective gets host tensors
    backend = os.environ.get("GR80211_BENCH_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    tx = pkg.txgen
    ctx = pkg.RxContext(local)
    n = args.frames if args.frames > 0 else DEFAULT_FRAMES[args.workload]
    arenas, sync, navail, exp, idx, L = build_corpus(torch, tx, args.workload, n, args.unique, rank, dev)
    arena, arena1 = arenas[0], (arenas[1] if len(arenas) > 1 else None)
    fmt, mcs, nss, nbytes, snr_db, b_alg = WORKLOADS[args.workload]
    desc = torch.from_numpy(ctx.make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
    blob_len = exp.shape[1]
    out_stride = ((blob_len + 4 + 63) // 64) * 64
    # two sets of outputs: at N > 1 the gather of one batch overlaps the kernels of the next
    depth = 2 if world > 1 else 1
    outs = [(torch.zeros((n, out_stride), dtype=torch.uint8, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev),
             torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(depth)]
    out, out_len, n_pdu = outs[0]
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    gatherer = shard.PduGather(n, out_stride, dev, dst=0, depth=depth) if world > 1 else None
    counter = [0]

    def step():
        slot = counter[0] % depth
        counter[0] += 1
        o, ol, npd = outs[slot]
        if gatherer is not None:
            gatherer.wait(slot)  # the batch that used these tensors two steps ago has left
        ctx.rx_batch_into(arena, arena1, desc, n, o, ol, npd)
        if gatherer is not None:
            # the one exchange step of the path: decoded MPDUs travel to rank 0 (RCCL gather over xGMI)
            gatherer.start(slot, o, ol)

    def fence():
        if gatherer is not None:
            gatherer.finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- correctness of what was just timed
    n_ok = int(n_pdu.sum().item())
    ok_rows = (n_pdu == 1)
    match = (out[:, :blob_len] == exp[idx]).all(dim=1)
    bitexact = int((match & ok_rows).sum().item())
    wrong = int((ok_rows & ~match).sum().item())

    # ---- per-kernel durations from HIP events on the launch stream (untimed extra steps)
    timing = {}
    if args.profile_steps > 0:
        ctx.set_profiling(True)
        acc = {}
        for _ in range(args.profile_steps):
            ctx.rx_batch_into(arena, arena1, desc, n, out, out_len, n_pdu)
            for k, v in ctx.last_timing().items():
                acc[k] = acc.get(k, 0.0) + v
        ctx.set_profiling(False)
        timing = {k: v / args.profile_steps for k, v in acc.items()}

    samples_per_step = n * L * world
    value = samples_per_step * args.steps / dt / 1e6
    res = {
        "metric": "RX Msamples/s (demod+Viterbi) per GPU + %HBM-roofline; PSDU bit-exact",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": args.workload, "frames_per_gpu": n, "samples_per_frame": L,
                   "format": "VHT" if fmt == 2 else "HT" if fmt == 1 else "L", "mcs": mcs, "n_rx": nss,
                   "psdu_bytes": nbytes + 4 if fmt == 2 else nbytes, "snr_db": snr_db, "unique_payloads": args.unique,
                   "parallelism": "frames sharded per GPU, RCCL gather of MPDUs" if world > 1 else "single GPU"},
        "psdu": {"frames": n, "fcs_ok": n_ok, "bit_exact_vs_tx": bitexact, "fcs_ok_but_wrong": wrong},
        "kernel_ms": timing,
    }
    # forward pass and traceback run as one kernel (k_decode); when the library is switched to the
    # two-kernel form the two slots add up to the same work
    dec_ms = timing.get("viterbi", 0.0) + timing.get("traceback", 0.0)
    if dec_ms > 0:
        timing["decode"] = dec_ms
        achieved = b_alg * n / (dec_ms * 1e-3) / 1e9
        steps_per_frame = TRELLIS_STEPS[args.workload]
        res["roofline"] = {"bound": "hbm", "kernel": "k_decode", "achieved": achieved, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                           "traffic": measured_traffic(args.workload, n),
                           "algorithmic_bytes_per_frame": b_alg,
                           "note": "k_decode is bound by VALU issue of the 64-state add-compare-select recursion, "
                                   "not by HBM; it also reads the materialised soft bits and writes/reads the "
                                   "survivor words, which the per-frame figure (IQ in + PSDU out) does not count",
                           "acs_state_updates_per_s": 64.0 * steps_per_frame * n / (dec_ms * 1e-3)}
        # VALU issue, the resource that does bound k_decode: one wave instruction updates one state of 64
        # frames; the add/add/sub/max/alignbit mix costs 14.6 SIMD-cycles per wave-wide state update at the
        # instruction rates measured on this chip (tools/ubench), on 1024 SIMDs at 2.4 GHz
        wave_updates = 64.0 * steps_per_frame * ((n + 63) // 64)
        cyc = 1024 * 2.4e9 * dec_ms * 1e-3 / wave_updates
        res["roofline"]["valu_issue"] = {"simd_cycles_per_wave_state_update": cyc, "floor": 14.6, "frac": 14.6 / cyc}
    if rank == 0 and world == 1 and args.detect:
        # raw capture in: frame detection (presiso -> trigger -> sync) finds the frames, `signal`'s trigger search
        # picks the ones it deals with, then the same RX kernels run; reported beside the headline, which starts
        # from given descriptors
        cap_desc = 4 * n + 1024
        ddesc, nd = ctx.detect(arena, max_desc=cap_desc)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ddesc, nd = ctx.detect(arena, max_desc=cap_desc)
        torch.cuda.synchronize()
        dtd = (time.perf_counter() - t0) / 3
        del ddesc
        r = ctx.rx_capture(arena, arena1, max_frames=cap_desc, out_stride=out_stride)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = ctx.rx_capture(arena, arena1, max_frames=cap_desc, out_stride=out_stride)
        torch.cuda.synchronize()
        dte = time.perf_counter() - t0
        nf = r["n"]
        so = torch.from_numpy(r["desc"][: nf * 32].cpu().numpy().view(np.uint64).reshape(-1, 4)[:, 0].astype(np.int64)).to(dev)
        fidx = torch.clamp(so // L, max=n - 1)
        okd = r["n_pdu"][:nf] == 1
        md = (r["out"][:nf, :blob_len] == exp[idx[fidx]]).all(dim=1)
        res["detect"] = {"value": arena.numel() / dtd / 1e6, "unit": "Msamples/s", "ms": dtd * 1e3, "sync_bytes": nd,
                         "achieved_GBps": 8.0 * arena.numel() / dtd / 1e9,
                         "hbm_frac": 8.0 * arena.numel() / dtd / 1e9 / HBM_PEAK_GBS,
                         "capture_to_pdu_value": n * L / dte / 1e6, "capture_to_pdu_ms": dte * 1e3,
                         "frames_dealt_with": nf, "fcs_ok": int(okd.sum().item()),
                         "bit_exact_vs_tx": int((okd & md).sum().item()),
                         "fcs_ok_but_wrong": int((okd & ~md).sum().item())}
        del r
    if rank == 0 and world == 1 and args.host_buffers:
        # the same step when the boundary hands over host buffers: IQ travels H2D, PDUs travel D2H (pinned memory)
        h_in = [a.cpu().pin_memory() for a in arenas]
        h_out = torch.empty(out.shape, dtype=out.dtype).pin_memory()
        h_len = torch.empty(out_len.shape, dtype=out_len.dtype).pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, h in zip(arenas, h_in):
            a.copy_(h, non_blocking=True)
        ctx.rx_batch_into(arena, arena1, desc, n, out, out_len, n_pdu)
        h_out.copy_(out, non_blocking=True)
        h_len.copy_(out_len, non_blocking=True)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t0
        res["host_buffers"] = {"value": n * L / dth / 1e6, "unit": "Msamples/s", "ms_per_step": dth * 1e3,
                               "note": "IQ arena H2D + step + PDU slots D2H over PCIe, pinned host memory; not the headline"}
        del h_in, h_out, h_len
    if rank == 0 and args.cpu_frames > 0 and world == 1:
        ns = min(args.cpu_frames, n)
        host = arena[: ns * L + 64].cpu().numpy()
        host1 = arena1[: ns * L + 64].cpu().numpy() if arena1 is not None else None
        lens = out_len[:ns].cpu().numpy()
        rows = out[:ns].cpu().numpy()
        gpu_blobs = [bytes(rows[i, : lens[i]]) for i in range(ns)]
        res["cpu_baseline"] = cpu_baseline(host, host1, sync, navail, ns, gpu_blobs)
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
