#!/usr/bin/env python3
"""Benchmark of the MI355X RX hot path: RX Msamples/s (demod + Viterbi) on synthetic 20 MHz frames.

    python bench.py --gpus N --steps K --warmup W

One rank per GPU (torch.distributed / RCCL when N > 1; RANK, LOCAL_RANK, WORLD_SIZE, MASTER_* from the
environment).  A step = one pass of the whole RX path (signal -> demod -> decode) over the rank's
batch of frames, IQ already resident in HBM; for N > 1 the step ends with the gather of the decoded
MPDUs on rank 0.  The timed steps rotate over six contexts per rank (--pipeline), queued by one host
thread without a wait between them, so that one batch's decoder tail runs beside the next batches' front
ends; `one_batch_at_a_time` in the output is the same steps on one context.  Rank 0 prints one JSON line.

Workload (default): BASELINE.json configs[2], the configuration the target is quoted on -- VHT MCS8
(256-QAM r=3/4) SISO 20 MHz, 1500-byte A-MPDU (1496-byte MPDU), AWGN SNR 30 dB, 100 000 frames per GPU.

`python bench.py --gpus N` without a launcher starts itself under torch.distributed.run (one rank per GPU) before
anything touches a GPU and relays rank 0's line.  At N = 8 the line also carries `capture_corpus`: BASELINE.json
configs[4], the reference's own IQ files and the frames of its Python generator replicated to 125 000 frames per GPU,
with the length-compacted gather (`--capture-corpus 1` adds it at any N; `--workload capture_corpus_replicated` makes it
the workload of the headline fields).
"""
import argparse
import importlib
import json
import os
import sys
import time

# Streams that share a hardware queue run one after the other, and a process gets four queues unless it asks for
# more: six contexts' streams, the collective's (RCCL brings its own) and the default stream want one each, and with
# eight the collective's stream was seen to share one with a context's stream (12.8 instead of 10.5 ms per step once
# the contexts wait for each other's pooled buffers).  Read by the HIP runtime when it starts, hence set before torch
# is imported.  GR80211_BENCH_HW_QUEUES=<n> asks for another number, 0 leaves the environment as it is.
_q = int(os.environ.get("GR80211_BENCH_HW_QUEUES", "16") or 0)
if _q > 0:
    os.environ["GPU_MAX_HW_QUEUES"] = str(min(_q, 32))

import numpy as np  # noqa: E402

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
                  "ht_mcs15_2x2_1500B_snr30": 50000, "mixed_l_ht_vht_snr30": 125000, "capture_corpus_replicated": 125000}
TRELLIS_STEPS = {"vht_mcs8_1500B_snr30": 12168, "vht_mcs4_1500B_snr30": 12168, "l_mcs0_1000B_snr30": 8022,
                 "ht_mcs15_2x2_1500B_snr30": 12022}
# k_demod's own algorithmic bytes per frame: the data symbols' samples in (80 per symbol and antenna, 8 B each) and the
# soft bits out (4 B each): (symbols, antennas, soft bits per frame)
DEMOD_SHAPE = {"vht_mcs8_1500B_snr30": (39, 1, 16224), "vht_mcs4_1500B_snr30": (78, 1, 16224),
               "l_mcs0_1000B_snr30": (335, 1, 16080), "ht_mcs15_2x2_1500B_snr30": (24, 2, 14976)}
H22 = ((1.0, 0.5j), (0.4, -1.0))  # fixed well-conditioned 2x2 channel of the two-antenna workload
HBM_PEAK_GBS = 8000.0
COPY_KERNEL_GBS = 4950.0  # measured: bytes read + bytes written of a float4 copy kernel over 3 GiB


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


def payloads_uniform(wl, n_frames, seed):
    """The MPDUs of corpus `seed` (one per frame, random bodies with their FCS) and the PDU blob each one comes back
    as: [format][len lo][len hi][MPDU][mcs].  A function of (workload, frames, seed) alone, so that rank 0 can
    rebuild what every other rank should have decoded."""
    import zlib
    fmt, mcs, nss, nbytes, snr_db, _ = WORKLOADS[wl]
    rng = np.random.default_rng(80211 + seed)
    body = rng.integers(0, 256, (n_frames, nbytes - 4), dtype=np.uint8)
    mpdu = np.empty((n_frames, nbytes), dtype=np.uint8)
    mpdu[:, :nbytes - 4] = body
    fcs = np.fromiter((zlib.crc32(row) for row in body), dtype=np.uint32, count=n_frames)
    mpdu[:, nbytes - 4:] = fcs.astype("<u4").view(np.uint8).reshape(n_frames, 4)
    blob = np.empty((n_frames, nbytes + 4), dtype=np.uint8)
    blob[:, 0], blob[:, 1], blob[:, 2], blob[:, 3:3 + nbytes], blob[:, 3 + nbytes] = fmt, nbytes % 256, nbytes // 256, mpdu, mcs
    return mpdu, blob


def build_corpus_gpu(torch, ctx, tx, wl, n_frames, seed, dev):
    """Same corpus shape with a different random payload in every frame: the frames are generated by the GPU
    transmitter (gr80211_tx_batch), mixed through the 2x2 channel if they have two streams, and noise is added.
    Returns the tuple of build_corpus plus the transmitter's rate."""
    fmt, mcs, nss, nbytes, snr_db, _ = WORKLOADS[wl]
    mpdu, blob = payloads_uniform(wl, n_frames, seed)
    if fmt == 2:
        # one-subframe A-MPDU: the delimiter and the padding are the same for every MPDU of this length
        shell = np.frombuffer(tx.ampdu_vht([bytes(nbytes)]), dtype=np.uint8)
        psdu = np.tile(shell, (n_frames, 1))
        psdu[:, 4:4 + nbytes] = mpdu
    else:
        psdu = mpdu
    plen = psdu.shape[1]
    L = ctx.tx_samples(fmt, mcs, nss, plen)
    d = ctx.make_tx_desc(n_frames)
    d["out_off"] = np.arange(n_frames, dtype=np.uint64) * L
    d["psdu_off"] = np.arange(n_frames, dtype=np.uint64) * plen
    d["psdu_len"], d["format"], d["mcs"], d["nss"], d["scale"] = plen, fmt, mcs, nss, 12.0 * np.sqrt(nss)
    d_psdu = torch.from_numpy(psdu.reshape(-1)).to(dev)
    d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(dev)
    arenas = [torch.zeros(n_frames * L + 64, dtype=torch.complex64, device=dev) for _ in range(nss)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ns = ctx.tx_batch(d_psdu, d_desc, n_frames, arenas[0], arenas[1] if nss == 2 else None)
    torch.cuda.synchronize()
    tx_ms = (time.perf_counter() - t0) * 1e3
    assert int((ns == L).sum().item()) == n_frames, "transmitter rejected frames"
    if nss == 2:
        a, b = arenas
        arenas = [H22[0][0] * a + H22[0][1] * b, H22[1][0] * a + H22[1][1] * b]
        del a, b
    power = float(sum((torch.view_as_real(x[: n_frames * L]) ** 2).sum().item() for x in arenas) / (nss * n_frames * L))
    sigma = float(np.sqrt(power / (10.0 ** (snr_db / 10.0)) / 2.0))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + seed)
    chunk = 8192 * L
    for x in arenas:
        v = torch.view_as_real(x)
        for s in range(0, n_frames * L, chunk):
            e = min(n_frames * L, s + chunk)
            v[s:e] += torch.randn((e - s, 2), generator=gen, device=dev, dtype=torch.float32) * sigma
    sync = (np.arange(n_frames, dtype=np.uint64) * L + 176)
    navail = np.full(n_frames, L - 176, dtype=np.uint32)
    exp = torch.from_numpy(blob).to(dev)
    idx = torch.arange(n_frames, device=dev)
    tx_info = {"ms": tx_ms, "Msamples_per_s": nss * n_frames * L / tx_ms / 1e3, "frames": n_frames,
               "note": "gr80211_tx_batch, one call incl. launch and synchronisation; PSDU bytes resident"}
    return arenas, sync, navail, exp, idx, L, tx_info


# The mixed capture corpus: legacy, HT and VHT single-stream frames of different rates and lengths, interleaved
MIXED = "mixed_l_ht_vht_snr30"
MIXED_KINDS = [(0, 0, 94), (0, 4, 300), (0, 7, 1000), (1, 0, 200), (1, 5, 800), (1, 7, 1500), (2, 2, 400), (2, 6, 1200),
               (2, 8, 1496)]  # (format, mcs, MPDU bytes); the first is the shape of the reference's own IQ fixture


def payloads_mixed(tx, n_frames, seed):
    """PSDU bytes and expected PDU blobs of the mixed corpus `seed`: a function of (frames, seed) alone (see
    payloads_uniform).  Returns (psdu bytes, psdu_off, plen, blob, kind_of, per_kind shells)."""
    import zlib
    rng = np.random.default_rng(80211 + seed)
    kinds = MIXED_KINDS
    maxb = max(k[2] for k in kinds)
    per_kind = []
    for fmt, mcs, nb in kinds:
        shell = np.frombuffer(tx.ampdu_vht([bytes(nb)]), dtype=np.uint8).copy() if fmt == 2 else np.zeros(nb, np.uint8)
        per_kind.append((shell, len(shell)))
    kind_of = np.arange(n_frames) % len(kinds)
    plen = np.array([per_kind[k][1] for k in kind_of], dtype=np.int64)
    psdu_off = np.concatenate([[0], np.cumsum(plen)[:-1]])
    psdu = np.zeros(int(plen.sum()), dtype=np.uint8)
    blob = np.zeros((n_frames, maxb + 4), dtype=np.uint8)
    for k, (fmt, mcs, nb) in enumerate(kinds):
        rows = np.nonzero(kind_of == k)[0]
        body = rng.integers(0, 256, (len(rows), nb - 4), dtype=np.uint8)
        mpdu = np.empty((len(rows), nb), dtype=np.uint8)
        mpdu[:, :nb - 4] = body
        mpdu[:, nb - 4:] = np.fromiter((zlib.crc32(r) for r in body), np.uint32, len(rows)).astype("<u4").view(np.uint8).reshape(-1, 4)
        ps = np.tile(per_kind[k][0], (len(rows), 1))
        ps[:, (4 if fmt == 2 else 0):(4 if fmt == 2 else 0) + nb] = mpdu
        idx = psdu_off[rows][:, None] + np.arange(per_kind[k][1])[None, :]
        psdu[idx] = ps
        blob[rows, 0], blob[rows, 1], blob[rows, 2] = fmt, nb % 256, nb // 256
        blob[rows, 3:3 + nb] = mpdu
        blob[rows, 3 + nb] = mcs
    return psdu, psdu_off, plen, blob, kind_of


def build_corpus_mixed(torch, ctx, tx, n_frames, seed, dev, snr_db=30.0):
    """Frames of MIXED_KINDS in rotation, each with its own payload, back to back with 64-sample gaps, made by
    the GPU transmitter.  Returns the tuple of build_corpus_gpu plus totals for the rate figures."""
    kinds = MIXED_KINDS
    psdu, psdu_off, plen, blob, kind_of = payloads_mixed(tx, n_frames, seed)
    per_kind = []
    for fmt, mcs, nb in kinds:
        pl = len(tx.ampdu_vht([bytes(nb)])) if fmt == 2 else nb
        m = tx.Mod(fmt, mcs, 1).set_len(pl)
        per_kind.append((None, pl, ctx.tx_samples(fmt, mcs, 1, pl), m.nsym * m.ndbps))
    nsamp = np.array([per_kind[k][2] for k in kind_of], dtype=np.int64)
    steps = np.array([per_kind[k][3] for k in kind_of], dtype=np.int64)
    gap = 64
    out_off = np.concatenate([[0], np.cumsum(nsamp + gap)[:-1]]) + gap
    total = int(out_off[-1] + nsamp[-1] + gap)
    d = ctx.make_tx_desc(n_frames)
    d["out_off"], d["psdu_off"], d["psdu_len"] = out_off, psdu_off, plen
    d["format"] = np.array([kinds[k][0] for k in kind_of])
    d["mcs"] = np.array([kinds[k][1] for k in kind_of])
    d["scale"] = 12.0
    arena = torch.zeros(total + 64, dtype=torch.complex64, device=dev)
    d_psdu = torch.from_numpy(psdu).to(dev)
    d_desc = torch.from_numpy(d.view(np.uint8).reshape(-1)).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ns = ctx.tx_batch(d_psdu, d_desc, n_frames, arena, None)
    torch.cuda.synchronize()
    tx_ms = (time.perf_counter() - t0) * 1e3
    assert int((ns > 0).sum().item()) == n_frames, "transmitter rejected frames"
    sigma = float(12.0 / 64.0 * 10.0 ** (-snr_db / 20.0) / np.sqrt(2.0))
    gen = torch.Generator(device=dev)
    gen.manual_seed(1 + seed)
    v = torch.view_as_real(arena)
    chunk = 1 << 25
    for s0 in range(0, arena.numel(), chunk):
        e = min(arena.numel(), s0 + chunk)
        v[s0:e] += torch.randn((e - s0, 2), generator=gen, device=dev, dtype=torch.float32) * sigma
    sync = (out_off + 176).astype(np.uint64)
    navail = (nsamp - 176 + gap).astype(np.uint32)
    exp = torch.from_numpy(blob).to(dev)
    tx_info = {"ms": tx_ms, "Msamples_per_s": float(nsamp.sum()) / tx_ms / 1e3, "frames": n_frames,
               "note": "gr80211_tx_batch, one call incl. launch and synchronisation; PSDU bytes resident"}
    extra = {"total_samples": int(nsamp.sum()), "total_steps": int(steps.sum()),
             "b_alg": float(8 * nsamp.sum() + sum(kinds[k][2] for k in kind_of) + 64 * n_frames) / n_frames}
    return [arena], sync, navail, exp, torch.arange(n_frames, device=dev), float(nsamp.mean()), tx_info, extra


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
    # one thread per physical core this process may run on (its affinity mask, /proc/cpuinfo's topology); the box's
    # share of the machine (OMP_NUM_THREADS, set by the harness) caps it where the mask shows the whole machine
    model, phys, logical = host_cpu()
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(phys, share) if share > 0 else phys)
    per = (n_sample + cores - 1) // cores
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(lambda k: run(min(k * per, n_sample), min((k + 1) * per, n_sample), t0 + budget_s),
                            range(cores)))
    dtn = time.perf_counter() - t0
    n_done = sum(p[0] for p in parts)
    samples = sum(p[1] for p in parts)
    # `frames`, `fcs_ok` and `gpu_mismatches` all describe the threaded leg, the one `value` is measured on; the
    # single-thread leg (a subset of the same frames, run first) has its own three figures
    return dict(value=samples / dtn / 1e6, unit="Msamples/s", cores=cores, kind="port", cpu_model=model,
                physical_cores_visible=phys, logical_cpus_visible=logical, cpu_share_of_the_box=share or None,
                sample="first %d frames of the benchmark corpus (%d samples) sharded over %d threads of the CPU "
                       "oracle, %.1f s; one thread alone: %d frames in %.1f s" % (n_done, samples, cores, dtn, d1[0], dt1),
                frames=n_done, fcs_ok=sum(p[3] for p in parts), gpu_mismatches=sum(p[2] for p in parts),
                single_thread_value=d1[1] / dt1 / 1e6,
                single_thread={"frames": d1[0], "fcs_ok": d1[3], "gpu_mismatches": d1[2]})


def measured_counters(workload, n):
    """Per-launch hardware counters of the dominant kernels from the committed rocprofv3 passes
    (profiles/counters_<workload>.json, or profiles/counters.json for the headline workload; written by
    tools/make_counters_json.py from the pass summaries beside them), if they were taken on this workload and batch size;
    None otherwise."""
    for name in ("counters_%s.json" % workload, "counters.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if t.get("workload") == workload and t.get("frames") == n:
                return t
        except (OSError, ValueError):
            pass
    return None


VECTOR_FP32_PEAK_TFLOPS = 157.3   # MI355X vector fp32 peak (256 CUs x 4 SIMDs x 32 lanes x 2 flop (FMA) x 2.4 GHz)
FLOP_PER_STATE_UPDATE = 3.0       # two float adds and one compare-select per trellis state and step (lib/decode_impl.cc:242-258)
# SIMD-cycles per wave-wide state update of the decoder's instruction mix at the issue rates measured on this chip
# (tools/ubench/valu2.hip: v_add/v_sub 2.1, v_max/v_alignbit 4.1 per instruction at saturation; 1.5 adds + sub + max +
# alignbit per update): the floor the kernel is priced against
MIX_FLOOR_CYCLES = 14.6


def launch_ranks(n_ranks):
    """Starts this script again under torch.distributed.run, one rank per GPU, as a child process; relays what the
    ranks print and leaves with the launcher's exit code.  Called before torch is imported: this process never
    touches a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL shares memory between ranks by dmabuf handles
    env.setdefault("OMP_NUM_THREADS", "16")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    sys.exit(child.wait())


def host_cpu():
    """(CPU model string, physical cores this process may run on, logical CPUs it may run on) from /proc/cpuinfo and
    the process's affinity mask: a physical core is a distinct (physical id, core id) pair among the allowed CPUs."""
    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    model, cores, cur = "unknown", set(), {}
    try:
        with open("/proc/cpuinfo") as f:
            for line in f.read().split("\n") + [""]:
                if not line.strip():
                    if cur and int(cur.get("processor", -1)) in allowed:
                        cores.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                    cur = {}
                    continue
                k, _, v = line.partition(":")
                cur[k.strip()] = v.strip()
                if k.strip() == "model name":
                    model = v.strip()
    except OSError:
        pass
    return model, max(1, len(cores)) if cores else len(allowed), len(allowed)


CAPTURE = "capture_corpus_replicated"


def build_corpus_capture(torch, ctx, n_frames, dev):
    """BASELINE.json configs[4]: the reference's own IQ files (one legacy frame on one antenna, one HT MCS8 frame on
    two) and the frames made by its Python generator (tests/golden/frame_*.npz: legacy, HT and VHT, one and two
    streams), replicated round-robin to n_frames.  One- and two-antenna sources form the two parts of a step (the
    `demod` and the `demod2` chain).  Expected PDUs come from the MPDUs the fixtures carry, not from the oracle.
    Returns a dict: parts [(arena0, arena1, desc, n, row_lo)], exp [u, blob_len], idx [n], totals."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import util as TU
    h22 = np.array(H22, dtype=np.complex128)
    srcs = []  # (antenna arrays, sync index, rad, format, mcs, mpdu)
    for f in TU.golden_frames():
        if f["nss"] == 1:
            srcs.append(([f["iq"][0].astype(np.complex64)], f["sync_idx"], f["rad"], f["fmt"], f["mcs"], f["mpdu"]))
        else:
            rx = (h22 @ f["iq"]).astype(np.complex64)
            srcs.append(([rx[0], rx[1]], f["sync_idx"], f["rad"], f["fmt"], f["mcs"], f["mpdu"]))
    ref_mpdu = TU.ref_fixture_mpdu()
    for (name, ants, sidx), (fmt, mcs) in zip(TU.ref_fixtures(), ((0, 0), (1, 8))):
        srcs.append((ants, sidx, 0.0, fmt, mcs, ref_mpdu))
    u = len(srcs)
    blob_len = max(len(s[5]) for s in srcs) + 4
    exp = np.zeros((u, blob_len), np.uint8)
    exp_len = np.zeros(u, np.int32)
    for k, s_ in enumerate(srcs):
        m = s_[5]
        b = bytes([s_[3], len(m) % 256, len(m) // 256]) + m + bytes([s_[4]])
        exp[k, :len(b)] = np.frombuffer(b, np.uint8)
        exp_len[k] = len(b)
    kind_of = np.arange(n_frames) % u
    parts, row_lo, samples, psdu_bytes, on_air = [], 0, 0.0, 0.0, np.zeros(u)
    order = []
    for nant in (1, 2):
        ks = [k for k in range(u) if len(srcs[k][0]) == nant]
        if not ks:
            continue
        frames = np.nonzero(np.isin(kind_of, ks))[0]          # this part's frames, in corpus order
        blk = [np.concatenate([srcs[k][0][a] for k in ks]) for a in range(nant)]
        blk_len = len(blk[0])
        offs, navail_u, pos = {}, {}, 0
        for k in ks:
            offs[k] = pos + srcs[k][1]
            navail_u[k] = len(srcs[k][0][0]) - srcs[k][1]
            pos += len(srcs[k][0][0])
        reps = (len(frames) + len(ks) - 1) // len(ks) + 1
        arenas = [torch.cat([torch.from_numpy(b).to(dev).repeat(reps), torch.zeros(64, dtype=torch.complex64, device=dev)])
                  for b in blk]
        # the j-th frame of kind k sits in replica j of the block
        rep_of = np.zeros(len(frames), np.int64)
        for k in ks:
            sel = kind_of[frames] == k
            rep_of[sel] = np.arange(int(sel.sum()))
        kk = kind_of[frames]
        sync = rep_of.astype(np.uint64) * np.uint64(blk_len) + np.array([offs[k] for k in kk], dtype=np.uint64)
        navail = np.array([navail_u[k] for k in kk], dtype=np.uint32)
        rads = np.array([srcs[k][2] for k in kk], dtype=np.float32)
        desc = torch.from_numpy(ctx.make_desc(sync, navail, rad=rads).view(np.uint8).reshape(-1)).to(dev)
        # on-air samples of every source frame: preamble (400) + what the L-SIG announces
        first = [int(np.nonzero(kk == k)[0][0]) for k in ks]
        info = ctx.signal_batch(arenas[0], arenas[1] if nant == 2 else None,
                                ctx.make_desc(sync[first], navail[first], rad=rads[first]), len(first))
        for k, fi in zip(ks, info):
            on_air[k] = 400 + int(fi["nsamp"])
        parts.append((arenas[0], arenas[1] if nant == 2 else None, desc, len(frames), row_lo, nant))
        order.append(frames)
        row_lo += len(frames)
    order = np.concatenate(order)                            # output row -> corpus frame
    idx = torch.from_numpy(kind_of[order]).to(dev)
    nrx = np.array([len(s_[0]) for s_ in srcs])
    per_kind_bytes = 8.0 * nrx * on_air + np.array([len(s_[5]) for s_ in srcs]) + 64
    return {"parts": parts, "n": n_frames, "exp": torch.from_numpy(exp).to(dev), "exp_len": torch.from_numpy(exp_len).to(dev),
            "idx": idx, "blob_len": blob_len, "total_samples": float(on_air[kind_of].sum()),
            "b_alg": float(per_kind_bytes[kind_of].sum()) / n_frames, "sources": u,
            "source_kinds": [[int(s_[3]), int(s_[4]), len(s_[0]), len(s_[5])] for s_ in srcs]}


class Rotation:
    """The timed steps of one corpus: contexts, their streams, rotating output tensors and the MPDU gather."""

    def __init__(self, torch, dist, pkg, shard, local, dev, parts, n, out_stride, pipe, collective, compact, first_ctx,
                 side_copy=False, pool=0):
        self.torch, self.dist, self.parts, self.n, self.pipe = torch, dist, parts, n, pipe
        self.depth = max(pipe, 2 if collective else 1)
        self.ctxs = [first_ctx] + [pkg.RxContext(local) for _ in range(pipe - 1)]
        for c in self.ctxs:
            c.use_work_pool(pool)
            # several batches in flight: the decoder forms with the fewest instructions per frame, whatever the batch size
            c.set_lane_state_max(0 if pipe > 1 else -1)
        self.outs = [(torch.zeros((n, out_stride), dtype=torch.uint8, device=dev),
                      torch.zeros(n, dtype=torch.int32, device=dev),
                      torch.zeros(n, dtype=torch.int32, device=dev)) for _ in range(self.depth)]
        # Kernels and collective share a stream explicitly: every step runs under the torch view of its context's own
        # stream (streams a context creates get hardware queues of their own; two from torch's pool were seen to share
        # one), the library is handed that stream, the gather is ordered behind whatever the stream holds when it is
        # started, and wait() makes the same stream wait for the collective before a slot's tensors are reused.
        self.streams = [c.own_stream() for c in self.ctxs]
        self.gatherer = None
        self.compact = compact
        if collective:
            if compact:
                # the packing kernels go to the stream of the context whose batch they follow
                self.gatherer = shard.PduGatherCompact(
                    n, out_stride, dev, lambda o, ol, pk, tot: self.ctxs[self._slot % pipe].pack_pdus(o, ol, pk, tot),
                    dst=0, depth=self.depth)
            else:
                self.gatherer = shard.PduGather(n, out_stride, dev, dst=0, depth=self.depth)
        self.counter, self.no_wait, self._slot = 0, False, 0
        self.gathered = None
        self.side = torch.cuda.Stream() if side_copy else None
        self.side_dst = [torch.empty_like(o[0]) for o in self.outs] if side_copy else None
        self.side_done = [None] * self.depth

    def step(self, gather=True):
        torch = self.torch
        slot = self.counter % self.depth
        self.counter += 1
        self._slot = slot
        o, ol, npd = self.outs[slot]
        c = self.ctxs[slot % self.pipe]
        g = self.gatherer if gather else None
        with torch.cuda.stream(self.streams[slot % self.pipe]):
            assert c._stream().value == c.lib.gr80211_get_stream(c.h), "the library must be handed the context's stream"
            if g is not None:
                r = g.wait(slot)  # the batch that used these tensors `depth` steps ago has left
                if r[0] is not None:
                    self.gathered = r
            if self.side is not None and self.side_done[slot] is not None:
                torch.cuda.current_stream().wait_event(self.side_done[slot])
            for a0, a1, desc, npart, lo, _ in self.parts:
                c.rx_batch_into(a0, a1, desc, npart, o[lo:lo + npart], ol[lo:lo + npart], npd[lo:lo + npart], no_wait=self.no_wait)
            if self.side is not None:
                self.side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self.side):
                    self.side_dst[slot].copy_(o, non_blocking=True)
                    self.side_done[slot] = self.side.record_event()
            if g is not None:
                # the one exchange step of the path: decoded MPDUs travel to rank 0
                g.start(slot, o, ol)

    def fence(self, barrier):
        torch = self.torch
        if self.gatherer is not None:
            for slot in range(self.depth):
                self._slot = slot
                with torch.cuda.stream(self.streams[slot % self.pipe]):
                    r = self.gatherer.wait(slot)
                if r[0] is not None:
                    self.gathered = r
        if barrier:
            self.dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, warmup, barrier, gather=True):
        """warm-up (every context's work buffers reach their size), fence, `steps` steps, fence: seconds."""
        for _ in range(max(warmup, self.pipe)):
            self.step(gather)
        self.fence(barrier)
        self.no_wait = True
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(gather)
        self.fence(barrier)
        dt = time.perf_counter() - t0
        refused = [c.last_batch_refused() for c in self.ctxs]
        if any(refused):
            raise RuntimeError("a batch queued without the wait found too little room: %r" % (refused,))
        return dt

    def close(self):
        for c in self.ctxs[1:]:  # the first context is the caller's
            c.close()


def check_rows(torch, rows, lens, exp, idx):
    """(FCS passes, of them bit-exact against the expected blob, of them wrong) of PDU slots against exp[idx]."""
    got = lens > 0
    m = (rows[:, :exp.shape[1]] == exp[idx]).all(dim=1)
    return int(got.sum().item()), int((got & m).sum().item()), int((got & ~m).sum().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU (0 = the workload's configured size)")
    ap.add_argument("--workload", default="vht_mcs8_1500B_snr30", choices=sorted(WORKLOADS) + [MIXED, CAPTURE])
    ap.add_argument("--unique", type=int, default=0,
                    help="distinct payloads in the corpus; 0 = every frame its own, frames made by the GPU transmitter")
    ap.add_argument("--cpu-frames", type=int, default=48000, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--detect", action="store_true", help="also time frame detection over the raw capture")
    ap.add_argument("--host-buffers", action="store_true", help="also time one step with PCIe transfers included")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra untimed steps with per-kernel HIP events")
    ap.add_argument("--pipeline", type=int, default=0,
                    help="contexts the timed steps rotate over (one host thread, nothing waited for between steps); "
                         "1 = one batch at a time; 0 = six, or two when the process has only the four hardware "
                         "queues of the runtime's default (GPU_MAX_HW_QUEUES) and the collective of an N > 1 run "
                         "needs one of them")
    ap.add_argument("--side-copy", action="store_true",
                    help="N = 1 only: after every step copy the PDU slots on one more stream, ordered as the gather of "
                         "an N > 1 run is (stands in for the collective's stream: what one more stream costs)")
    ap.add_argument("--in-flight", type=int, default=2,
                    help="also time the same steps with this many batches in flight (one context, stream and host "
                         "thread each; N = 1 only; 1 = skip)")
    ap.add_argument("--pool", type=int, default=-1,
                    help="work sets (soft bits, survivor words, decoded bits) the rotating contexts share "
                         "(gr80211_use_work_pool: that many survivor buffers and two soft-bit buffers more); 0 = every context its "
                         "own; -1 = two when more than three contexts rotate "
                         "and their own buffers would take more than 64 GB together")
    ap.add_argument("--latency", type=int, default=1, help="also time small batches one at a time (0 = skip)")
    ap.add_argument("--force-collective", action="store_true",
                    help="N = 1: run the exchange step all the same (process group of one rank, the gather of the MPDUs on "
                         "device tensors, its stream ordering and the check of what arrived): the N > 1 code path on one GPU")
    ap.add_argument("--gather", default="auto", choices=["auto", "fixed", "compact"],
                    help="exchange of the MPDUs: fixed-stride slots, or blobs packed to their lengths (lengths first, then "
                         "the bytes); auto = compact for corpora of mixed lengths, fixed for uniform ones")
    ap.add_argument("--capture-corpus", type=int, default=-1,
                    help="also time BASELINE.json configs[4] (the reference's IQ files and generator frames replicated, "
                         "125 000 frames per GPU) and report it as `capture_corpus`: 1 = yes, 0 = no, -1 = only at 8 GPUs")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args.gpus)  # does not return

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (args.gpus, world))
    # RCCL needs a GPU per rank.  Where there are fewer (a rehearsal of the N > 1 path on one GPU), or with
    # GR80211_BENCH_BACKEND=gloo, the ranks share devices round-robin and every collective gets host tensors.
    backend = os.environ.get("GR80211_BENCH_BACKEND") or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    local = local % max(1, torch.cuda.device_count()) if backend == "gloo" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cdev = dev if backend == "nccl" else "cpu"

    def max_over_ranks(x):
        if not collective:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    pkg = importlib.import_module("fork-gr-ieee80211_amd")
    shard = importlib.import_module("fork-gr-ieee80211_amd.shard")
    tx = pkg.txgen
    ctx = pkg.RxContext(local)
    n = args.frames if args.frames > 0 else DEFAULT_FRAMES[args.workload]
    tx_info = None
    extra = None
    cap = None
    sync = navail = None
    if args.workload == CAPTURE:
        cap = build_corpus_capture(torch, ctx, n, dev)
        parts, exp, idx, L = cap["parts"], cap["exp"], cap["idx"], cap["total_samples"] / n
        extra = {"total_samples": cap["total_samples"], "b_alg": cap["b_alg"], "total_steps": None}
        arena, arena1 = parts[0][0], parts[0][1]
    else:
        if args.workload == MIXED:
            arenas, sync, navail, exp, idx, L, tx_info, extra = build_corpus_mixed(torch, ctx, tx, n, rank, dev)
        elif args.unique > 0:
            arenas, sync, navail, exp, idx, L = build_corpus(torch, tx, args.workload, n, args.unique, rank, dev)
        else:
            arenas, sync, navail, exp, idx, L, tx_info = build_corpus_gpu(torch, ctx, tx, args.workload, n, rank, dev)
        arena, arena1 = arenas[0], (arenas[1] if len(arenas) > 1 else None)
        desc = torch.from_numpy(ctx.make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
        parts = [(arena, arena1, desc, n, 0, 2 if arena1 is not None else 1)]
    fmt, mcs, nss, nbytes, snr_db, b_alg = WORKLOADS[args.workload] if extra is None else (-1, -1, 1, 0, 30.0, extra["b_alg"])
    blob_len = exp.shape[1]
    out_stride = ((blob_len + 4 + 63) // 64) * 64
    # The timed steps rotate over `pipe` contexts, each with its own stream, work buffers and output tensors: step k is
    # queued on context k mod pipe without the host waiting for anything (GR80211_FLAG_NO_WAIT once the warm-up has
    # sized the buffers), so the tail of one batch's decoder -- SIMDs that held one trellis fall idle before those that
    # held two -- runs beside the next batch's preamble, demodulator and first decoder waves.  Every step is a whole
    # pass over the whole batch; the steps differ only in where their results land.
    queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
    pipe = args.pipeline if args.pipeline > 0 else (6 if queues >= 8 else 3 if not collective else 2)
    compact = args.gather == "compact" or (args.gather == "auto" and args.workload in (MIXED, CAPTURE))
    if args.pool >= 0:
        pool = args.pool
    elif pipe <= 3:
        pool = 0
    else:
        # what one context's work buffers take for this batch decides: six sets of a small batch are cheap and let six
        # batches overlap (up to 64 GB of them are left alone), six of the headline batch would hold 100 GB
        probe = pkg.RxContext(local)
        for a0, a1, d_, npart, _, _ in parts:
            probe.rx_batch(a0, a1, d_, npart, out_stride=out_stride, want_info=False)
        probe.sync()
        own_bytes = probe.work_memory()[0]
        probe.close()
        del probe
        torch.cuda.empty_cache()
        pool = 2 if own_bytes * pipe > 64e9 else 0
    rot = Rotation(torch, dist, pkg, shard, local, dev, parts, n, out_stride, pipe, collective, compact, ctx,
                   side_copy=args.side_copy and not collective, pool=pool)
    out, out_len, n_pdu = rot.outs[0]
    dt = max_over_ranks(rot.timed(args.steps, args.warmup, collective))
    free_b, total_b = torch.cuda.mem_get_info(dev)
    mem_after_timed = (total_b - free_b) / 1e9
    wm = [c.work_memory() for c in rot.ctxs]
    work_gb = (sum(w[0] for w in wm) + wm[0][1]) / 1e9

    # ---- with the exchange: the same steps without it, to show what the gather costs (not the headline)
    value_local = None
    if collective:
        rot.fence(True)
        t1 = time.perf_counter()
        for _ in range(args.steps):
            rot.step(gather=False)
        rot.fence(True)
        value_local = max_over_ranks(time.perf_counter() - t1)

    # ---- what arrived on rank 0 is checked rank by rank against the MPDUs that rank's corpus was made from (the
    # corpora are functions of the rank: rank 0 rebuilds every rank's expected PDUs)
    gather_check = None
    if collective and rank == 0 and rot.gathered is not None:
        got_all, len_all = rot.gathered
        per_rank = []
        for r in range(world):
            if args.workload == MIXED:
                exp_r, idx_r = torch.from_numpy(payloads_mixed(tx, n, r)[3]), torch.arange(n)
            elif args.workload == CAPTURE:
                exp_r, idx_r = exp, idx  # every rank replicates the same sources
            elif args.unique > 0:
                exp_r = None  # the replicated corpus keeps its payloads on the rank that drew them
            else:
                exp_r, idx_r = torch.from_numpy(payloads_uniform(args.workload, n, r)[1]), torch.arange(n)
            lens = len_all[r * n:(r + 1) * n]
            rows = shard.unpack_pdus(got_all[r], lens, out_stride) if compact else got_all[r * n:(r + 1) * n]
            rec = {"rank": r, "fcs_ok": int((lens > 0).sum().item())}
            if compact:
                rec["bytes"] = int(got_all[r].numel())
            if exp_r is not None:
                c3 = check_rows(torch, rows, lens, exp_r.to(rows.device), idx_r.to(rows.device))
                rec["bit_exact_vs_tx"], rec["fcs_ok_but_wrong"] = c3[1], c3[2]
            per_rank.append(rec)
        gather_check = {"ranks": per_rank, "slots_per_rank": n, "slot_bytes": out_stride,
                        "form": "blobs packed to their lengths (lengths gathered first, bytes point to point)" if compact
                                else "fixed-stride slots, one gather",
                        "backend": backend,
                        "note": "the PDUs rank 0 holds after the last gather of the timed region, "
                                "compared with the MPDUs each rank's corpus was generated from"}

    # ---- correctness of what was just timed: the results of the last step of every context (the first set is the one
    # reported; the others must not differ from it)
    n_ok = int(n_pdu.sum().item())
    ok_rows = (n_pdu == 1)
    match = (out[:, :blob_len] == exp[idx]).all(dim=1)
    bitexact = int((match & ok_rows).sum().item())
    wrong = int((ok_rows & ~match).sum().item())
    for o2, ol2, np2 in rot.outs[1:min(rot.depth, max(args.warmup, pipe) + args.steps)]:
        if not (torch.equal(np2, n_pdu) and torch.equal(ol2, out_len) and torch.equal(o2[:, :blob_len], out[:, :blob_len])):
            raise RuntimeError("the contexts of the pipeline disagree about the same batch")

    def one_step(c, no_wait=False):
        for a0, a1, d_, npart, lo, _ in parts:
            c.rx_batch_into(a0, a1, d_, npart, out[lo:lo + npart], out_len[lo:lo + npart], n_pdu[lo:lo + npart], no_wait=no_wait)

    # ---- the same steps one batch at a time (one context, each step queued behind the one before): the figure the
    # pipelined rate is to be compared with
    serial_ms = None
    if pipe > 1:
        rot.fence(collective)
        rot.ctxs[0].set_lane_state_max(-1)  # a batch alone: small ones take the form that gets them done soonest
        t1 = time.perf_counter()
        with torch.cuda.stream(rot.streams[0]):
            for _ in range(args.steps):
                one_step(rot.ctxs[0], no_wait=True)
        rot.fence(collective)
        serial_ms = (time.perf_counter() - t1) / args.steps * 1e3
        if rot.ctxs[0].last_batch_refused():
            raise RuntimeError("a batch queued without the wait found too little room")

    # ---- the same steps with several batches in flight: one context per batch, each on the stream it created (these
    # land on hardware queues of their own; streams from torch's pool were seen to share one), one host thread each.
    # While one batch's decoder drains, the next batch's kernels take the SIMDs that fall idle.  Not the headline.
    in_flight = None
    if not collective and args.in_flight > 1 and len(parts) == 1:
        try:
            import threading
            depth_f = args.in_flight
            fctxs = rot.ctxs[:depth_f] + [pkg.RxContext(local) for _ in range(depth_f - len(rot.ctxs[:depth_f]))]
            for c in fctxs:
                c.set_lane_state_max(0)
            fouts = [(out, out_len, n_pdu)] + [tuple(torch.zeros_like(t) for t in (out, out_len, n_pdu))
                                                for _ in range(depth_f - 1)]
            errors = []

            stagger = dt / args.steps / depth_f  # the batches start a fraction of a step apart, as a stream delivers them

            def flight(k, steps, delay=0.0):
                try:
                    if delay > 0:
                        time.sleep(delay)
                    for _ in range(steps):
                        fctxs[k].rx_batch_into(arena, arena1, parts[0][2], n, *fouts[k])
                    fctxs[k].sync()
                except Exception as e:  # noqa: BLE001 - reported from the main thread
                    errors.append(repr(e))

            for k in range(depth_f):
                flight(k, 2)  # the contexts' work buffers reach their size
            per = max(4, (args.steps + depth_f - 1) // depth_f)
            torch.cuda.synchronize()
            threads = [threading.Thread(target=flight, args=(k, per, k * stagger)) for k in range(depth_f)]
            t1 = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            torch.cuda.synchronize()
            dtf = time.perf_counter() - t1
            if errors:
                raise RuntimeError("; ".join(errors))
            exact = []
            for o, ol, npd in fouts:
                rows = (npd == 1)
                m = (o[:, :blob_len] == exp[idx]).all(dim=1)
                exact.append([int(rows.sum().item()), int((rows & m).sum().item()), int((rows & ~m).sum().item())])
            in_flight = {"contexts": depth_f, "steps": per * depth_f, "ms_per_step": dtf / (per * depth_f) * 1e3,
                         "fcs_ok__bit_exact__wrong_per_context": exact}
            for c in fctxs[len(rot.ctxs):]:
                c.close()
            del fouts, fctxs
        except Exception as e:  # noqa: BLE001 - this leg is an extra: the headline line is printed whatever happens here
            in_flight = {"error": repr(e)}

    # ---- per-kernel durations from HIP events on the launch stream (untimed extra steps)
    timing = {}
    ctx.set_lane_state_max(-1)  # from here on one batch at a time
    if args.profile_steps > 0 and len(parts) == 1:
        ctx.set_profiling(True)
        acc = {}
        for _ in range(args.profile_steps):
            one_step(ctx)
            for k, v in ctx.last_timing().items():
                acc[k] = acc.get(k, 0.0) + v
        ctx.set_profiling(False)
        timing = {k: v / args.profile_steps for k, v in acc.items()}

    samples_per_step = (n * L if extra is None else extra["total_samples"]) * world
    value = samples_per_step * args.steps / dt / 1e6
    res = {
        "metric": "RX Msamples/s (demod+Viterbi) per GPU + %HBM-roofline; PSDU bit-exact",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if cap is None else "the reference's IQ fixtures and generator frames, replicated",
        "config": {"workload": args.workload, "frames_per_gpu": n, "samples_per_frame": L,
                   "format": "VHT" if fmt == 2 else "HT" if fmt == 1 else "L" if fmt == 0 else "L/HT/VHT in rotation",
                   "mcs": mcs if extra is None else ([list(k) for k in MIXED_KINDS] if cap is None else cap["source_kinds"]),
                   "n_rx": nss if cap is None else "1 and 2",
                   "psdu_bytes": nbytes + 4 if fmt == 2 else nbytes, "snr_db": snr_db, "unique_payloads": args.unique if args.unique > 0 else n,
                   "parallelism": ("frames sharded per rank, gather of MPDUs to rank 0 over %s" %
                                   ("RCCL" if backend == "nccl" else backend + " (host-staged rehearsal, ranks share GPUs)"))
                   if collective else "single GPU",
                   "batches_in_flight": pipe, "work_sets_shared": pool, "hw_queues": queues},
        "psdu": {"frames": n, "fcs_ok": n_ok, "bit_exact_vs_tx": bitexact, "fcs_ok_but_wrong": wrong},
        "kernel_ms": timing,
    }
    if timing and pipe > 1:
        res["kernels_overlap_across_steps"] = True
        res["kernel_ms_note"] = ("kernel_ms: separate steps on one context, nothing beside them (HIP events); ms_per_step: "
                                 "steps rotating over %d contexts whose kernels overlap, so a kernel's time alone can exceed "
                                 "a step's share of the rotation" % pipe)
    res["device_memory"] = {"allocated_GB_peak_torch": torch.cuda.max_memory_allocated(dev) / 1e9,
                            "in_use_GB_after_timed_steps": mem_after_timed,
                            "library_work_buffers_GB": work_gb, "work_sets_in_pool": wm[0][2],
                            "note": "in_use: total minus free device memory (hipMemGetInfo) right after the timed steps: corpus, "
                                    "output slots, what torch's allocator keeps cached and the library's work buffers; "
                                    "library_work_buffers: soft bits, survivor words and decoded bits of the rotating "
                                    "contexts and of the pool they share (gr80211_work_memory)"}
    if serial_ms is not None:
        res["one_batch_at_a_time"] = {"ms_per_step": serial_ms, "value": samples_per_step / serial_ms / 1e3,
                                      "unit": "Msamples/s",
                                      "note": "the same steps on one context per rank, each queued behind the one before, "
                                              "no exchange; `value` above rotates its steps over %d contexts" % pipe}
    if gather_check is not None:
        res["gathered_on_rank0"] = gather_check
    if value_local is not None:
        res["without_gather"] = {"value": samples_per_step * args.steps / value_local / 1e6, "unit": "Msamples/s",
                                 "ms_per_step": value_local / args.steps * 1e3,
                                 "note": "same steps with the MPDU gather to rank 0 left out; max over ranks"}
    if in_flight is not None and "error" not in in_flight:
        in_flight["value"] = samples_per_step / (in_flight["ms_per_step"] * 1e-3) / 1e6
        in_flight["unit"] = "Msamples/s"
        in_flight["note"] = ("the same batch decoded by %d contexts at once, each on its own stream and host thread; "
                             "`value` above is one batch at a time" % in_flight["contexts"])
    if in_flight is not None:
        res["batches_in_flight"] = in_flight
    if tx_info is not None:
        res["tx"] = tx_info
    # forward pass and traceback run as one kernel (k_decode); when the library is switched to the
    # two-kernel form the two slots add up to the same work
    dec_ms = timing.get("viterbi", 0.0) + timing.get("traceback", 0.0)
    if dec_ms > 0 and (extra is None or extra.get("total_steps")):
        timing["decode"] = dec_ms
        dur = dec_ms * 1e-3
        steps_per_frame = TRELLIS_STEPS[args.workload] if extra is None else extra["total_steps"] / n
        updates = 64.0 * steps_per_frame * n                       # trellis state updates of the launch, as the path defines them
        ctr = measured_counters(args.workload, n)
        kd = (ctr or {}).get("k_decode", {})
        traffic = None
        if "FETCH_SIZE_KiB" in kd and "WRITE_SIZE_KiB" in kd:
            # gfx950 tallies the 128-byte read requests of wide streaming loads at 64 bytes: FETCH_SIZE is doubled
            # (pinned by tools/ubench/hbm.hip under --pmc: a read of known size counts half, profiles/r03_fetch_size_check.txt)
            traffic = 2.0 * kd["FETCH_SIZE_KiB"] * 1024.0 + kd["WRITE_SIZE_KiB"] * 1024.0
        hbm = b_alg * n / dur / 1e9
        tflops = FLOP_PER_STATE_UPDATE * updates / dur / 1e12
        # The kernel is bound by vector-instruction issue of the add-compare-select recursion, not by memory: the
        # record is priced against the vector fp32 peak (adds, a subtract and a maximum per update; nothing in it can
        # be an FMA, so half of that peak is the most any schedule of it could reach), the HBM figures stay beside it.
        res["roofline"] = {"bound": "valu", "kernel": "k_decode", "achieved": tflops, "peak": VECTOR_FP32_PEAK_TFLOPS,
                           "unit": "TFLOP/s", "frac": tflops / VECTOR_FP32_PEAK_TFLOPS, "traffic": traffic,
                           "flop_per_state_update": FLOP_PER_STATE_UPDATE,
                           "acs_state_updates_per_s": updates / dur,
                           "kernel_ms": dec_ms,
                           "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS,
                                   "algorithmic_bytes_per_frame": b_alg, "traffic_bytes_per_launch": traffic,
                                   "traffic_over_algorithmic": (traffic / (b_alg * n)) if traffic else None,
                                   "counters_raw": {k: kd[k] for k in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB") if k in kd},
                                   "note": "IQ in + PSDU out per frame; the launch also reads the soft bits the "
                                           "demodulator materialised and writes and reads the survivor words; traffic = "
                                           "2 x FETCH_SIZE + WRITE_SIZE (the raw counters are beside it)"}}
        # vector-instruction issue, the resource that binds: every figure but the last two comes from counters of
        # the profiled launch (rocprofv3 passes under profiles/, same workload and batch size) and from this run's
        # own kernel duration
        wave_updates = 64.0 * steps_per_frame * ((n + 63) // 64) if extra is None else float(extra["total_steps"])
        vi = {"wave_state_updates": wave_updates}
        if "SQ_INSTS_VALU" in kd and "GRBM_GUI_ACTIVE" in kd and kd.get("duration_ns", 0) > 0:
            clk = kd["GRBM_GUI_ACTIVE"] / 8.0 / kd["duration_ns"]          # GHz: the counter sums the 8 XCDs
            simd_cycles = 1024.0 * clk * 1e9 * dur
            vi.update({"source": ctr.get("source"), "valu_insts_per_launch": kd["SQ_INSTS_VALU"],
                       "valu_insts_per_wave_state_update": kd["SQ_INSTS_VALU"] / wave_updates,
                       "clock_ghz_under_load": clk,
                       "issue_slots_used_frac": 2.0 * kd["SQ_INSTS_VALU"] / simd_cycles,
                       "simd_cycles_per_wave_state_update": simd_cycles / wave_updates})
            if "SQ_WAVE_CYCLES" in kd and "SQ_WAIT_ANY" in kd:
                vi["wave_cycles_waiting_frac"] = kd["SQ_WAIT_ANY"] / kd["SQ_WAVE_CYCLES"]
        else:
            vi["simd_cycles_per_wave_state_update"] = 1024 * 2.4e9 * dur / wave_updates
            vi["note"] = "no counter pass for this workload and batch size: cycles at the nominal 2.4 GHz"
        vi["mix_floor_cycles"] = MIX_FLOOR_CYCLES
        vi["frac_of_mix_floor"] = MIX_FLOOR_CYCLES / vi["simd_cycles_per_wave_state_update"]
        res["roofline"]["valu_issue"] = vi
        if pipe > 1 and "clock_ghz_under_load" in vi and all("SQ_INSTS_VALU" in (ctr or {}).get(k, {}) for k in ("k_demod", "k_preamble")):
            # the timed steps, with the kernels of several batches side by side: what all of a step's kernels issue
            # against the SIMD cycles one step takes
            insts = kd["SQ_INSTS_VALU"] + ctr["k_demod"]["SQ_INSTS_VALU"] + ctr["k_preamble"]["SQ_INSTS_VALU"]
            cyc = 1024.0 * vi["clock_ghz_under_load"] * 1e9 * res["ms_per_step"] * 1e-3
            res["roofline"]["rotation"] = {"valu_insts_per_step": insts, "simd_cycles_per_inst": cyc / insts,
                                           "issue_slots_used_frac": 2.0 * insts / cyc,
                                           "note": "decoder, demodulator and preamble kernels of one step (counters of the "
                                                   "profiled launches) over the SIMD cycles of ms_per_step at the clock "
                                                   "measured under the decoder"}
        if extra is None and timing.get("demod", 0.0) > 0:
            # the kernel of the path that HBM does bound: samples in, soft bits out
            nsym, nrx, nllr = DEMOD_SHAPE[args.workload]
            b_demod = nsym * 80 * 8 * nrx + 4 * nllr
            ach = b_demod * n / (timing["demod"] * 1e-3) / 1e9
            km = (ctr or {}).get("k_demod", {})
            tr = (2.0 * km["FETCH_SIZE_KiB"] + km["WRITE_SIZE_KiB"]) * 1024.0 if "FETCH_SIZE_KiB" in km and "WRITE_SIZE_KiB" in km else None
            res["roofline"]["k_demod"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": ach / HBM_PEAK_GBS, "algorithmic_bytes_per_frame": b_demod, "traffic": tr,
                                          "counters_raw": {k: km[k] for k in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB") if k in km},
                                          "copy_kernel_gbs": COPY_KERNEL_GBS,
                                          "note": "copy_kernel_gbs: what a plain device-to-device copy kernel sustains "
                                                  "on this chip (tools/ubench/hbm.hip, profiles/r02_hbm_ubench.txt)"}
    if rank == 0 and not collective and args.detect and extra is None:
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
    if rank == 0 and not collective and args.host_buffers and len(parts) == 1:
        # the same step when the boundary hands over host buffers: IQ travels H2D, PDUs travel D2H (pinned memory)
        arenas_ = [a for a in (arena, arena1) if a is not None]
        h_in = [a.cpu().pin_memory() for a in arenas_]
        h_out = torch.empty(out.shape, dtype=out.dtype).pin_memory()
        h_len = torch.empty(out_len.shape, dtype=out_len.dtype).pin_memory()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for a, h in zip(arenas_, h_in):
            a.copy_(h, non_blocking=True)
        one_step(ctx)
        h_out.copy_(out, non_blocking=True)
        h_len.copy_(out_len, non_blocking=True)
        torch.cuda.synchronize()
        dth = time.perf_counter() - t0
        res["host_buffers"] = {"value": samples_per_step / world / dth / 1e6, "unit": "Msamples/s", "ms_per_step": dth * 1e3,
                               "note": "IQ arena H2D + step + PDU slots D2H over PCIe, pinned host memory; not the headline"}
        del h_in, h_out, h_len

    # ---- small batches one at a time: what a receiver that gets n frames and wants them back pays (call + wait)
    if rank == 0 and not collective and len(parts) == 1 and args.latency:
        import ctypes as C
        lat = []
        lctx = pkg.RxContext(local)
        for nl in (1, 64, 1000, 4000, 16000):
            if nl > n:
                continue
            d_n = parts[0][2][: nl * 32]
            o_l, ol_l, np_l = out[:nl], out_len[:nl], n_pdu[:nl]
            ts = []
            for _ in range(9):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                lctx.rx_batch_into(arena, arena1, d_n, nl, o_l, ol_l, np_l)
                lctx.sync()
                ts.append((time.perf_counter() - t1) * 1e3)
            rec = {"frames": nl, "rx_batch_ms": float(np.median(ts[2:])), "fcs_ok": int(np_l.sum().item())}
            # the decoder alone on the same frames' soft bits (gr80211_demod_batch leaves them in the context)
            info_t = torch.empty(nl * pkg.rx.FRAME_INFO_DTYPE.itemsize, dtype=torch.uint8, device=dev)
            llr_ptr, nfl = C.c_void_p(), C.c_uint64(0)
            rcq = lctx.lib.gr80211_demod_batch(lctx.h, arena.data_ptr(), arena1.data_ptr() if arena1 is not None else None,
                                               arena.numel(), d_n.data_ptr(), nl, 1, info_t.data_ptr(), C.byref(llr_ptr),
                                               C.byref(nfl), None)
            lctx.sync()
            if rcq == 0 and nfl.value:
                fi = info_t.cpu().numpy().view(pkg.rx.FRAME_INFO_DTYPE)
                dd = np.zeros(nl, dtype=pkg.rx.DEC_DESC_DTYPE)
                for k_ in ("llr_off", "format", "len", "cr", "mcs", "ampdu", "trellis"):
                    dd[k_] = fi[k_]
                dd["n_llr"] = fi["total"]
                dctx = ctx  # another context: the soft bits belong to lctx
                ts = []
                for _ in range(9):
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    rcq = dctx.lib.gr80211_decode_batch(dctx.h, llr_ptr, nfl.value, dd.ctypes.data, nl, 0, o_l.data_ptr(),
                                                        o_l.shape[1], ol_l.data_ptr(), np_l.data_ptr(), None, 0, None)
                    dctx.lib.gr80211_sync(dctx.h, None)
                    ts.append((time.perf_counter() - t1) * 1e3)
                rec["decode_batch_ms"] = float(np.median(ts[2:]))
                rec["decode_fcs_ok"] = int(np_l.sum().item())
            lat.append(rec)
        lctx.close()
        res["latency"] = {"batches": lat,
                          "note": "first n frames of the corpus, one batch at a time on an idle GPU, median of 7: call + "
                                  "wait for the stream; rx_batch from IQ (descriptors on the device: one wait for the "
                                  "scheduler inside the call), decode_batch from resident soft bits (host descriptors)"}

    # ---- BASELINE.json configs[4] beside the headline: the capture corpus, sharded like everything else (every rank
    # its 125 000 replicas), MPDUs gathered on rank 0 packed to their lengths
    want_cap = args.capture_corpus == 1 or (args.capture_corpus < 0 and world == 8)
    if want_cap and args.workload != CAPTURE:
        rot.close()
        del rot, out, out_len, n_pdu
        torch.cuda.empty_cache()
        ncap = DEFAULT_FRAMES[CAPTURE]
        cc = build_corpus_capture(torch, ctx, ncap, dev)
        stride_c = ((cc["blob_len"] + 4 + 63) // 64) * 64
        rc = Rotation(torch, dist, pkg, shard, local, dev, cc["parts"], ncap, stride_c, pipe, collective, args.gather != "fixed", ctx,
                      pool=pool)
        steps_c = max(4, args.steps // 2)
        dtc = max_over_ranks(rc.timed(steps_c, args.warmup, collective))
        o_c, ol_c, np_c = rc.outs[0]
        c3 = check_rows(torch, o_c, ol_c, cc["exp"], cc["idx"])
        rec = {"workload": CAPTURE, "value": cc["total_samples"] * world * steps_c / dtc / 1e6, "unit": "Msamples/s",
               "ms_per_step": dtc / steps_c * 1e3, "steps": steps_c, "frames_per_gpu": ncap, "sources": cc["sources"],
               "source_kinds__format_mcs_antennas_mpdu_bytes": cc["source_kinds"],
               "algorithmic_bytes_per_frame": cc["b_alg"],
               "psdu": {"frames": ncap, "fcs_ok": c3[0], "bit_exact_vs_fixture_mpdu": c3[1], "fcs_ok_but_wrong": c3[2]},
               "note": "the reference's tools/sig80211GenMultiple*.bin and the frames of its Python generator "
                       "(tests/golden), replicated round-robin; one- and two-antenna frames are the two calls of a step"}
        if collective and rank == 0 and rc.gathered is not None:
            got_all, len_all = rc.gathered
            per_rank = []
            for r in range(world):
                lens = len_all[r * ncap:(r + 1) * ncap]
                rows = (shard.unpack_pdus(got_all[r], lens, stride_c) if rc.compact else got_all[r * ncap:(r + 1) * ncap])
                k3 = check_rows(torch, rows, lens, cc["exp"].to(rows.device), cc["idx"].to(rows.device))
                per_rank.append({"rank": r, "fcs_ok": k3[0], "bit_exact_vs_fixture_mpdu": k3[1], "fcs_ok_but_wrong": k3[2],
                                 "bytes": int(got_all[r].numel()) if rc.compact else ncap * stride_c})
            rec["gathered_on_rank0"] = {"ranks": per_rank, "fixed_stride_bytes_per_rank": ncap * stride_c,
                                        "form": "packed to lengths" if rc.compact else "fixed stride"}
        res["capture_corpus"] = rec
        rc.close()
        del rc, cc

    if rank == 0 and args.cpu_frames > 0 and not collective and sync is not None:
        ns = min(args.cpu_frames, n)
        upto = int(sync[ns - 1]) + int(navail[ns - 1]) + 64  # the first ns frames end here
        host = arena[:upto].cpu().numpy()
        host1 = arena1[:upto].cpu().numpy() if arena1 is not None else None
        o_, ol_, np_ = (torch.zeros((n, out_stride), dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.int32, device=dev),
                        torch.zeros(n, dtype=torch.int32, device=dev))
        ctx.rx_batch_into(arena, arena1, parts[0][2], n, o_, ol_, np_)
        ctx.sync()
        lens = ol_[:ns].cpu().numpy()
        rows = o_[:ns].cpu().numpy()
        gpu_blobs = [bytes(rows[i, : lens[i]]) for i in range(ns)]
        res["cpu_baseline"] = cpu_baseline(host, host1, sync, navail, ns, gpu_blobs)
    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    if collective:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
