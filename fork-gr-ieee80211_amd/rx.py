"""ctypes binding of libgr80211_rx.so (the MI355X RX hot path) for tests, bench and tools.

Device memory is owned by torch tensors (torch is plumbing here: allocation, streams,
torch.distributed); every compute call goes through the C-ABI declared in
include/gr80211_rx.h.  There is no CPU fallback: if the HIP library is missing or no GPU is
present, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GR80211_RX_LIB", os.path.join(_HERE, "libgr80211_rx.so"))  # override: A/B builds

ST_NAMES = {0: "ok", 1: "drop_lsig", 2: "drop_len_check", 3: "drop_unsupported", 4: "drop_decode_limit", 5: "ndp"}
ST_NAMES_INV = {v: k for k, v in ST_NAMES.items()}
F_VHT_CHAN = 20  # format byte of a sounding channel message (C8P_F_VHT_CHAN, lib/cloud80211phy.h)
FLAG_DESC_ON_DEVICE = 1
FLAG_NO_WAIT = 2


class FrameDesc(C.Structure):
    _fields_ = [("sync_off", C.c_uint64), ("n_avail", C.c_uint32), ("rad", C.c_float),
                ("snr", C.c_float), ("rssi", C.c_float), ("mu_pos", C.c_uint32), ("reserved", C.c_uint32)]


class FrameInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "status", "lsig_mcs", "lsig_len", "nsamp", "format", "mcs", "len", "cr", "ampdu", "trellis",
        "total", "nSym", "nSS", "nLTF", "nCBPS", "nDBPS", "nBPSCS", "mod")] + [
        ("sssnr0", C.c_float), ("sssnr1", C.c_float), ("cfo_hz", C.c_float), ("snr", C.c_float),
        ("rssi", C.c_float), ("slot", C.c_int32), ("llr_off", C.c_int64), ("data_off", C.c_int32),
        ("nSymSamp", C.c_int32)]


class DecDesc(C.Structure):
    _fields_ = [("llr_off", C.c_int64), ("n_llr", C.c_int32), ("format", C.c_int32), ("len", C.c_int32),
                ("cr", C.c_int32), ("mcs", C.c_int32), ("ampdu", C.c_int32), ("trellis", C.c_int32),
                ("reserved", C.c_int32)]


FRAME_DESC_DTYPE = np.dtype([("sync_off", "<u8"), ("n_avail", "<u4"), ("rad", "<f4"), ("snr", "<f4"),
                             ("rssi", "<f4"), ("mu_pos", "<u4"), ("reserved", "<u4")])
FRAME_INFO_DTYPE = np.dtype(
    [(n, "<i4") for n in ("status", "lsig_mcs", "lsig_len", "nsamp", "format", "mcs", "len", "cr", "ampdu",
                          "trellis", "total", "nSym", "nSS", "nLTF", "nCBPS", "nDBPS", "nBPSCS", "mod")] +
    [("sssnr0", "<f4"), ("sssnr1", "<f4"), ("cfo_hz", "<f4"), ("snr", "<f4"), ("rssi", "<f4"),
     ("slot", "<i4"), ("llr_off", "<i8"), ("data_off", "<i4"), ("nSymSamp", "<i4")])
SIG_DESC_DTYPE = np.dtype([("off", "<u8"), ("n_avail", "<u4"), ("mcs", "<i4"), ("len", "<i4"), ("nsamp", "<i4"),
                           ("cfo", "<f4"), ("snr", "<f4"), ("rssi", "<f4"), ("mu_pos", "<u4"), ("reserved", "<u4"),
                           ("pad", "<u4")])
assert SIG_DESC_DTYPE.itemsize == 48
DEC_DESC_DTYPE = np.dtype([("llr_off", "<i8"), ("n_llr", "<i4"), ("format", "<i4"), ("len", "<i4"),
                           ("cr", "<i4"), ("mcs", "<i4"), ("ampdu", "<i4"), ("trellis", "<i4"),
                           ("reserved", "<i4")])

TX_DESC_DTYPE = np.dtype([("out_off", "<u8"), ("psdu_off", "<u4"), ("psdu_len", "<i4"), ("format", "<i4"),
                          ("mcs", "<i4"), ("nss", "<i4"), ("scrambler", "<i4"), ("scale", "<f4"), ("cfo_rad", "<f4")])
assert TX_DESC_DTYPE.itemsize == 40
TX_MU_DESC_DTYPE = np.dtype([("out_off", "<u8"), ("psdu_off", "<u4", (2,)), ("psdu_len", "<i4", (2,)), ("mcs", "<i4", (2,)),
                             ("group_id", "<i4"), ("q_index", "<i4"), ("scrambler", "<i4"), ("scale", "<f4"),
                             ("cfo_rad", "<f4"), ("reserved", "<u4")])
assert TX_MU_DESC_DTYPE.itemsize == 56
assert FRAME_DESC_DTYPE.itemsize == C.sizeof(FrameDesc) == 32
assert FRAME_INFO_DTYPE.itemsize == C.sizeof(FrameInfo) == 112
assert DEC_DESC_DTYPE.itemsize == C.sizeof(DecDesc) == 40

EXPORTS = [
    "gr80211_create", "gr80211_destroy", "gr80211_last_error", "gr80211_sync", "gr80211_rx_batch", "gr80211_signal_batch", "gr80211_detect", "gr80211_rx_capture",
    "gr80211_demod_batch", "gr80211_decode_batch", "gr80211_dev_alloc", "gr80211_dev_free",
    "gr80211_copy_h2d", "gr80211_copy_d2h", "gr80211_set_profiling", "gr80211_last_timing",
    "gr80211_version", "gr80211_table_checksum", "gr80211_tx_samples", "gr80211_tx_batch", "gr80211_set_mu_pos", "gr80211_set_ampdu_strict", "gr80211_set_lane_state_max", "gr80211_tx_mu_batch",
    "gr80211_host_alloc", "gr80211_host_free", "gr80211_copy_d2d", "gr80211_signal_emit_batch", "gr80211_demod_tagged_batch",
    "gr80211_rx_submit", "gr80211_rx_finish", "gr80211_decode_batch_noalloc", "gr80211_rx_resume",
    "gr80211_last_batch_refused", "gr80211_copy_h2d_async", "gr80211_copy_d2h_async", "gr80211_copy_d2d_async", "gr80211_pull_h2d", "gr80211_pull_done", "gr80211_push_d2h", "gr80211_get_stream",
    "gr80211_stream_idle", "gr80211_stream_wait", "gr80211_marker_record", "gr80211_marker_done",
    "gr80211_host_register", "gr80211_host_unregister", "gr80211_pack_pdus", "gr80211_use_work_pool", "gr80211_work_memory",
]

_lib = None


def load_library():
    """dlopen the HIP library and declare the C-ABI signatures.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("HIP library %s not built; run __graft_entry__.build()" % LIB_PATH)
    # torch ships a HIP runtime of its own: it has to be the one already loaded when the library's libamdhip64
    # dependency is resolved, or a process that uses both ends up with two runtimes and torch sees no device
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.gr80211_create.argtypes = [C.POINTER(vp), C.c_int]
    L.gr80211_destroy.argtypes = [vp]
    L.gr80211_destroy.restype = None
    L.gr80211_last_error.argtypes = [vp]
    L.gr80211_last_error.restype = C.c_char_p
    L.gr80211_sync.argtypes = [vp, vp]
    L.gr80211_rx_batch.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, C.c_int, vp, vp, vp, vp]
    L.gr80211_rx_submit.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp]
    L.gr80211_rx_finish.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    L.gr80211_detect.argtypes = [vp, vp, C.c_uint64, vp, C.c_int, C.POINTER(C.c_int), vp]
    L.gr80211_rx_capture.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, vp, vp, C.c_int, vp, C.POINTER(C.c_int), vp]
    L.gr80211_signal_batch.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, vp]
    L.gr80211_demod_batch.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, C.POINTER(vp),
                                      C.POINTER(C.c_uint64), vp]
    L.gr80211_decode_batch.argtypes = [vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, C.c_int, vp, vp, vp,
                                       C.c_int, vp]
    L.gr80211_rx_resume.argtypes = [vp, vp, vp, C.c_uint64, vp, vp, vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    L.gr80211_last_batch_refused.argtypes = [vp]
    for name in ("gr80211_copy_h2d_async", "gr80211_copy_d2h_async", "gr80211_copy_d2d_async", "gr80211_pull_h2d"):
        getattr(L, name).argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.gr80211_pull_h2d.restype = C.c_longlong
    L.gr80211_pull_done.argtypes = [vp, C.c_longlong, C.c_int]
    L.gr80211_push_d2h.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.gr80211_get_stream.argtypes = [vp]
    L.gr80211_get_stream.restype = vp
    L.gr80211_stream_idle.argtypes = [vp, vp]
    L.gr80211_stream_wait.argtypes = [vp, vp, vp, vp]
    L.gr80211_marker_record.argtypes = [vp, vp]
    L.gr80211_marker_record.restype = C.c_longlong
    L.gr80211_marker_done.argtypes = [vp, C.c_longlong, C.c_int]
    L.gr80211_host_register.argtypes = [vp, vp, C.c_size_t]
    L.gr80211_host_unregister.argtypes = [vp, vp]
    L.gr80211_decode_batch_noalloc.argtypes = [vp, vp, C.c_uint64, vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, vp]
    L.gr80211_pack_pdus.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_uint64, vp, vp, vp]
    L.gr80211_dev_alloc.argtypes = [vp, C.c_size_t]
    L.gr80211_dev_alloc.restype = vp
    L.gr80211_host_alloc.argtypes = [vp, C.c_size_t]
    L.gr80211_host_alloc.restype = vp
    L.gr80211_copy_d2d.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.gr80211_signal_emit_batch.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, vp, vp, C.c_uint64, vp, vp, vp]
    L.gr80211_demod_tagged_batch.argtypes = [vp, vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, vp, C.POINTER(vp),
                                             C.POINTER(C.c_uint64), vp]
    L.gr80211_host_free.argtypes = [vp, vp]
    L.gr80211_host_free.restype = None
    L.gr80211_dev_free.argtypes = [vp, vp]
    L.gr80211_dev_free.restype = None
    L.gr80211_copy_h2d.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.gr80211_copy_d2h.argtypes = [vp, vp, vp, C.c_size_t, vp]
    L.gr80211_set_profiling.argtypes = [vp, C.c_int]
    L.gr80211_set_mu_pos.argtypes = [vp, C.c_int]
    L.gr80211_use_work_pool.argtypes = [vp, C.c_int]
    L.gr80211_work_memory.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.gr80211_set_ampdu_strict.argtypes = [vp, C.c_int]
    L.gr80211_set_lane_state_max.argtypes = [vp, C.c_int]
    L.gr80211_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    L.gr80211_tx_samples.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.gr80211_tx_batch.argtypes = [vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, vp, C.c_uint64, vp, vp]
    L.gr80211_tx_mu_batch.argtypes = [vp, vp, C.c_uint64, vp, C.c_int, C.c_uint, vp, C.c_int, vp, vp, C.c_uint64, vp, vp]
    L.gr80211_version.restype = C.c_char_p
    L.gr80211_table_checksum.restype = C.c_uint32
    _lib = L
    return L


def split_pdus(blob):
    """Split concatenated [format][len16 LE][MPDU][mcs] messages (lib/decode_impl.cc:512-516)."""
    out = []
    p = 0
    while p + 4 <= len(blob):
        ln = blob[p + 1] | (blob[p + 2] << 8)
        if blob[p] == F_VHT_CHAN:  # 128 complex64 samples, no trailing mcs byte (lib/decode_impl.cc:105-120)
            out.append(dict(format=F_VHT_CHAN, mcs=None, mpdu=bytes(blob[p + 3:p + 3 + ln])))
            p += ln + 3
            continue
        out.append(dict(format=int(blob[p]), mcs=int(blob[p + 3 + ln]), mpdu=bytes(blob[p + 3:p + 3 + ln])))
        p += ln + 4
    return out


class RxContext:
    """One GPU's RX pipeline.  Inputs/outputs are torch CUDA tensors or numpy host arrays."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU: the RX hot path has no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        self.device = int(device)
        self.dev = torch.device("cuda", self.device)
        h = C.c_void_p()
        rc = self.lib.gr80211_create(C.byref(h), self.device)
        if rc != 0:
            raise RuntimeError("gr80211_create failed: %d" % rc)
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.gr80211_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.gr80211_last_error(self.h).decode()))

    def own_stream(self):
        """The context's own HIP stream as a torch stream (torch.cuda.ExternalStream): work queued under
        `with torch.cuda.stream(ctx.own_stream())` -- this library's calls, torch kernels, collectives -- shares it."""
        if getattr(self, "_own", None) is None:
            self._own = self.torch.cuda.ExternalStream(self.lib.gr80211_get_stream(self.h), device=self.dev)
        return self._own

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.dev).cuda_stream)

    def to_device(self, a):
        t = self.torch
        if isinstance(a, t.Tensor):
            return a.to(self.dev).contiguous()
        a = np.ascontiguousarray(a)
        if a.dtype.fields is not None:
            return t.from_numpy(a.view(np.uint8).reshape(-1)).to(self.dev)
        return t.from_numpy(a).to(self.dev)

    def set_profiling(self, on):
        self._check(self.lib.gr80211_set_profiling(self.h, 1 if on else 0), "set_profiling")

    def use_work_pool(self, max_sets):
        """Take the large work buffers from the GPU's shared pool of at most max_sets sets (0: own buffers again)."""
        self._check(self.lib.gr80211_use_work_pool(self.h, int(max_sets)), "use_work_pool")

    def work_memory(self):
        """(bytes of the large work buffers this context holds, bytes of its GPU's pool, sets in the pool)."""
        own, pool, sets = C.c_uint64(0), C.c_uint64(0), C.c_int(0)
        self._check(self.lib.gr80211_work_memory(self.h, C.byref(own), C.byref(pool), C.byref(sets)), "work_memory")
        return int(own.value), int(pool.value), int(sets.value)

    def set_mu_pos(self, pos):
        self._check(self.lib.gr80211_set_mu_pos(self.h, int(pos)), "set_mu_pos")

    def set_ampdu_strict(self, on):
        self._check(self.lib.gr80211_set_ampdu_strict(self.h, 1 if on else 0), "set_ampdu_strict")

    def set_lane_state_max(self, max_frames):
        """Batches of at most max_frames frames take the one-frame-per-wave decoder (lowest latency alone); 0 = never,
        for callers that keep several batches in flight; < 0 = the default."""
        self._check(self.lib.gr80211_set_lane_state_max(self.h, int(max_frames)), "set_lane_state_max")

    def last_timing(self):
        ms = (C.c_float * 6)()
        self._check(self.lib.gr80211_last_timing(self.h, ms, 6), "last_timing")
        return dict(zip(("preamble", "schedule", "demod", "viterbi", "traceback", "total"), list(ms)))

    def sync(self):
        self._check(self.lib.gr80211_sync(self.h, self._stream()), "sync")

    @staticmethod
    def make_desc(sync_off, n_avail, rad=None, snr=None, rssi=None, mu_pos=None):
        n = len(sync_off)
        d = np.zeros(n, dtype=FRAME_DESC_DTYPE)
        d["sync_off"] = sync_off
        d["n_avail"] = n_avail
        if rad is not None:
            d["rad"] = rad
        if snr is not None:
            d["snr"] = snr
        if rssi is not None:
            d["rssi"] = rssi
        if mu_pos is not None:
            d["mu_pos"] = mu_pos
        return d

    def _desc_arg(self, desc):
        """Returns (pointer, flags, keepalive)."""
        t = self.torch
        if isinstance(desc, t.Tensor):
            return C.c_void_p(desc.data_ptr()), FLAG_DESC_ON_DEVICE, desc
        desc = np.ascontiguousarray(desc)
        return C.c_void_p(desc.ctypes.data), 0, desc

    # -- entry points -------------------------------------------------------------------
    def rx_batch(self, iq0, iq1, desc, n_frames, out_stride=4104, want_info=True):
        """IQ -> PDUs.  iq0/iq1: complex64 CUDA tensors (iq1 None = one antenna).
        Returns dict of CUDA tensors out[n,out_stride] u8, out_len[n], n_pdu[n], info (bytes)."""
        t = self.torch
        assert iq0.is_cuda and iq0.dtype == t.complex64 and iq0.is_contiguous()
        if iq1 is not None:
            assert iq1.is_cuda and iq1.dtype == t.complex64 and iq1.numel() == iq0.numel()
        out = t.empty((n_frames, out_stride), dtype=t.uint8, device=self.dev)
        out_len = t.empty(n_frames, dtype=t.int32, device=self.dev)
        n_pdu = t.empty(n_frames, dtype=t.int32, device=self.dev)
        info = t.empty(n_frames * FRAME_INFO_DTYPE.itemsize, dtype=t.uint8, device=self.dev) if want_info else None
        dp, flags, keep = self._desc_arg(desc)
        rc = self.lib.gr80211_rx_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None,
            iq0.numel(), dp, n_frames, flags, C.c_void_p(out.data_ptr()), out_stride,
            C.c_void_p(out_len.data_ptr()), C.c_void_p(n_pdu.data_ptr()),
            C.c_void_p(info.data_ptr()) if info is not None else None, self._stream())
        self._check(rc, "gr80211_rx_batch")
        return dict(out=out, out_len=out_len, n_pdu=n_pdu, info=info, _keep=keep)

    def rx_batch_into(self, iq0, iq1, desc, n_frames, out, out_len, n_pdu, info=None, no_wait=False):
        """Same as rx_batch with caller-owned output tensors (no allocation in the timed region).  no_wait: the call
        queues the whole batch without waiting for the scheduler (GR80211_FLAG_NO_WAIT); last_batch_refused() tells,
        once the stream has passed the batch, whether the work buffers were too small for it."""
        dp, flags, keep = self._desc_arg(desc)
        if no_wait:
            flags |= FLAG_NO_WAIT
        rc = self.lib.gr80211_rx_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None,
            iq0.numel(), dp, n_frames, flags, C.c_void_p(out.data_ptr()), out.shape[1],
            C.c_void_p(out_len.data_ptr()), C.c_void_p(n_pdu.data_ptr()),
            C.c_void_p(info.data_ptr()) if info is not None else None, self._stream())
        self._check(rc, "gr80211_rx_batch")
        return keep

    def last_batch_refused(self):
        """True if the last batch whose sizes the host supplied (rx_resume, decode_batch with host descriptors,
        rx_batch_into(no_wait=True)) found too little room on the device and was not run.  Call after sync()."""
        return self.lib.gr80211_last_batch_refused(self.h) != 0

    def rx_submit(self, iq0, iq1, desc, n_frames):
        """First half of rx_batch_into: preamble and scheduling are launched, nothing is waited for.  Returns what has
        to stay alive until rx_finish."""
        dp, flags, keep = self._desc_arg(desc)
        rc = self.lib.gr80211_rx_submit(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None,
            iq0.numel(), dp, n_frames, flags, self._stream())
        self._check(rc, "gr80211_rx_submit")
        return keep

    def rx_finish(self, out, out_len, n_pdu, info=None):
        """Second half: waits for the scheduler's totals, launches demodulator and decoder, returns without waiting."""
        rc = self.lib.gr80211_rx_finish(
            self.h, C.c_void_p(out.data_ptr()), out.shape[1], C.c_void_p(out_len.data_ptr()),
            C.c_void_p(n_pdu.data_ptr()), C.c_void_p(info.data_ptr()) if info is not None else None)
        self._check(rc, "gr80211_rx_finish")

    def rx_resume(self, iq0, iq1, desc_sel, index_sel, info_all, out, out_len, n_pdu, info=None):
        """demod + decode for some of the frames the last signal_batch call on this context ran over.  desc_sel:
        FRAME_DESC_DTYPE ndarray of the kept frames; index_sel: their positions in that call (int32); info_all: the
        records that call returned (FRAME_INFO_DTYPE ndarray).  Nothing is waited for."""
        d = np.ascontiguousarray(desc_sel)
        ix = np.ascontiguousarray(index_sel, dtype=np.int32)
        ia = np.ascontiguousarray(info_all)
        rc = self.lib.gr80211_rx_resume(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None, iq0.numel(),
            C.c_void_p(d.ctypes.data), C.c_void_p(ix.ctypes.data), C.c_void_p(ia.ctypes.data), len(ix),
            C.c_void_p(out.data_ptr()), out.shape[1], C.c_void_p(out_len.data_ptr()), C.c_void_p(n_pdu.data_ptr()),
            C.c_void_p(info.data_ptr()) if info is not None else None, self._stream())
        self._check(rc, "gr80211_rx_resume")

    def detect(self, iq, max_desc=None):
        """Frame detection over a capture (complex64 CUDA tensor).  Returns (desc tensor on the device as
        bytes, n) -- desc[: n * 32] feeds rx_batch / signal_batch directly."""
        t = self.torch
        assert iq.is_cuda and iq.dtype == t.complex64 and iq.is_contiguous()
        if max_desc is None:
            max_desc = iq.numel() // 320 + 16
        desc = t.zeros(max(max_desc, 1) * FRAME_DESC_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        n = C.c_int(0)
        rc = self.lib.gr80211_detect(self.h, C.c_void_p(iq.data_ptr()), iq.numel(), C.c_void_p(desc.data_ptr()),
                                     max_desc, C.byref(n), self._stream())
        self._check(rc, "gr80211_detect")
        return desc, n.value

    def rx_capture(self, iq0, iq1=None, max_frames=None, out_stride=4104):
        """Raw capture(s) -> PDUs.  Returns dict(out, out_len, n_pdu, desc (bytes), n) with n frames in stream order."""
        t = self.torch
        assert iq0.is_cuda and iq0.dtype == t.complex64 and iq0.is_contiguous()
        if max_frames is None:
            max_frames = iq0.numel() // 400 + 16
        m = max(max_frames, 1)
        out = t.zeros((m, out_stride), dtype=t.uint8, device=self.dev)
        out_len = t.zeros(m, dtype=t.int32, device=self.dev)
        n_pdu = t.zeros(m, dtype=t.int32, device=self.dev)
        desc = t.zeros(m * FRAME_DESC_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        n = C.c_int(0)
        rc = self.lib.gr80211_rx_capture(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None, iq0.numel(),
            C.c_void_p(out.data_ptr()), out_stride, C.c_void_p(out_len.data_ptr()), C.c_void_p(n_pdu.data_ptr()),
            max_frames, C.c_void_p(desc.data_ptr()), C.byref(n), self._stream())
        self._check(rc, "gr80211_rx_capture")
        return dict(out=out, out_len=out_len, n_pdu=n_pdu, desc=desc, n=n.value)

    def signal_batch(self, iq0, iq1, desc, n_frames):
        """IQ -> per-frame tag records only (L-SIG, format detection, SIG fields).  Returns the info ndarray."""
        t = self.torch
        info = t.empty(max(n_frames, 1) * FRAME_INFO_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        dp, flags, keep = self._desc_arg(desc)
        rc = self.lib.gr80211_signal_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None,
            iq0.numel(), dp, n_frames, flags, C.c_void_p(info.data_ptr()), self._stream())
        self._check(rc, "gr80211_signal_batch")
        self.sync()
        return info.cpu().numpy().view(FRAME_INFO_DTYPE)[:n_frames]

    def demod_batch(self, iq0, iq1, desc, n_frames):
        """IQ -> LLRs.  Returns (info ndarray, llr ndarray on host)."""
        t = self.torch
        info = t.empty(n_frames * FRAME_INFO_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        dp, flags, keep = self._desc_arg(desc)
        llr_ptr = C.c_void_p()
        nfl = C.c_uint64(0)
        rc = self.lib.gr80211_demod_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None,
            iq0.numel(), dp, n_frames, flags, C.c_void_p(info.data_ptr()), C.byref(llr_ptr), C.byref(nfl),
            self._stream())
        self._check(rc, "gr80211_demod_batch")
        self.sync()
        llr = np.zeros(max(int(nfl.value), 1), dtype=np.float32)
        if nfl.value:
            self._check(self.lib.gr80211_copy_d2h(self.h, C.c_void_p(llr.ctypes.data), llr_ptr,
                                                  int(nfl.value) * 4, self._stream()), "copy_d2h")
        info_h = info.cpu().numpy().view(FRAME_INFO_DTYPE)
        return info_h, llr[: int(nfl.value)]

    def demod_batch_device(self, iq0, iq1, desc, n_frames):
        """demod_batch without the copy of the whole soft-bit arena: returns (info ndarray, read) where
        read(off, n) fetches n soft bits from float offset `off` of the arena the context holds until its next call."""
        t = self.torch
        info = t.empty(n_frames * FRAME_INFO_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        dp, flags, keep = self._desc_arg(desc)
        llr_ptr = C.c_void_p()
        nfl = C.c_uint64(0)
        rc = self.lib.gr80211_demod_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None,
            iq0.numel(), dp, n_frames, flags, C.c_void_p(info.data_ptr()), C.byref(llr_ptr), C.byref(nfl),
            self._stream())
        self._check(rc, "gr80211_demod_batch")
        self.sync()
        total = int(nfl.value)

        def read(off, n):
            assert 0 <= off and off + n <= total
            a = np.zeros(max(n, 1), dtype=np.float32)
            if n:
                self._check(self.lib.gr80211_copy_d2h(self.h, C.c_void_p(a.ctypes.data), C.c_void_p(llr_ptr.value + 4 * off),
                                                      4 * n, self._stream()), "copy_d2h")
            return a[:n]
        return info.cpu().numpy().view(FRAME_INFO_DTYPE), read

    def signal_emit(self, iq0, iq1, desc, n_frames, out_off, out_items):
        """`signal` alone: (info ndarray, chan [n, 64] complex64 ndarray, out0, out1 CUDA tensors).  out_off: item
        offset of every frame in the output (host array of uint64)."""
        t = self.torch
        info = t.empty(max(n_frames, 1) * FRAME_INFO_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        chan = t.zeros((max(n_frames, 1), 64), dtype=t.complex64, device=self.dev)
        out0 = t.zeros(max(out_items, 1), dtype=t.complex64, device=self.dev)
        out1 = t.zeros(max(out_items, 1), dtype=t.complex64, device=self.dev) if iq1 is not None else None
        dp, flags, keep = self._desc_arg(desc)
        assert flags == 0, "host descriptors"
        off = np.ascontiguousarray(out_off, dtype=np.uint64)
        rc = self.lib.gr80211_signal_emit_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None, iq0.numel(), dp,
            n_frames, 0, C.c_void_p(off.ctypes.data), C.c_void_p(out0.data_ptr()),
            C.c_void_p(out1.data_ptr()) if out1 is not None else None, out_items, C.c_void_p(info.data_ptr()),
            C.c_void_p(chan.data_ptr()), self._stream())
        self._check(rc, "gr80211_signal_emit_batch")
        self.sync()
        return info.cpu().numpy().view(FRAME_INFO_DTYPE)[:n_frames], chan.cpu().numpy()[:n_frames], out0, out1

    def demod_tagged(self, iq0, iq1, sig_desc, n_frames, chan):
        """`demod` alone, from `signal`'s items and tags: (info ndarray, llr ndarray on host).  sig_desc: SIG_DESC_DTYPE
        ndarray; chan: [n, 64] complex64 (ndarray or CUDA tensor)."""
        t = self.torch
        info = t.empty(max(n_frames, 1) * FRAME_INFO_DTYPE.itemsize, dtype=t.uint8, device=self.dev)
        d_chan = chan if isinstance(chan, t.Tensor) else t.from_numpy(np.ascontiguousarray(chan, np.complex64)).to(self.dev)
        sd = np.ascontiguousarray(sig_desc)
        llr_ptr = C.c_void_p()
        nfl = C.c_uint64(0)
        rc = self.lib.gr80211_demod_tagged_batch(
            self.h, C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()) if iq1 is not None else None, iq0.numel(),
            C.c_void_p(sd.ctypes.data), n_frames, 0, C.c_void_p(d_chan.data_ptr()), C.c_void_p(info.data_ptr()),
            C.byref(llr_ptr), C.byref(nfl), self._stream())
        self._check(rc, "gr80211_demod_tagged_batch")
        self.sync()
        llr = np.zeros(max(int(nfl.value), 1), dtype=np.float32)
        if nfl.value:
            self._check(self.lib.gr80211_copy_d2h(self.h, C.c_void_p(llr.ctypes.data), llr_ptr,
                                                  int(nfl.value) * 4, self._stream()), "copy_d2h")
        return info.cpu().numpy().view(FRAME_INFO_DTYPE)[:n_frames], llr[: int(nfl.value)]

    def decode_batch(self, llr, desc, n_frames, out_stride=4104, want_bits=False, bits_stride=4104):
        """LLRs -> PDUs.  llr: float32 CUDA tensor; desc: DEC_DESC_DTYPE ndarray or device bytes."""
        t = self.torch
        assert llr.is_cuda and llr.dtype == t.float32 and llr.is_contiguous()
        out = t.empty((n_frames, out_stride), dtype=t.uint8, device=self.dev)
        out_len = t.empty(n_frames, dtype=t.int32, device=self.dev)
        n_pdu = t.empty(n_frames, dtype=t.int32, device=self.dev)
        bits = t.zeros((n_frames, bits_stride), dtype=t.uint8, device=self.dev) if want_bits else None
        dp, flags, keep = self._desc_arg(desc)
        rc = self.lib.gr80211_decode_batch(
            self.h, C.c_void_p(llr.data_ptr()), llr.numel(), dp, n_frames, flags,
            C.c_void_p(out.data_ptr()), out_stride, C.c_void_p(out_len.data_ptr()),
            C.c_void_p(n_pdu.data_ptr()), C.c_void_p(bits.data_ptr()) if bits is not None else None,
            bits_stride if bits is not None else 0, self._stream())
        self._check(rc, "gr80211_decode_batch")
        return dict(out=out, out_len=out_len, n_pdu=n_pdu, bits=bits, _keep=keep)

    def decode_batch_noalloc(self, llr, desc, n_frames, out, out_len, n_pdu, bits=None):
        """gr80211_decode_batch_noalloc: LLRs -> PDUs inside the library's fixed scratch array.  desc: DEC_DESC_DTYPE
        ndarray on the host; out [n, stride] u8, out_len [n], n_pdu [n] i32 and bits [n, stride] u8 (optional) are CUDA
        tensors of the caller: nothing is allocated here either."""
        desc = np.ascontiguousarray(desc)
        rc = self.lib.gr80211_decode_batch_noalloc(
            self.h, C.c_void_p(llr.data_ptr()), llr.numel(), C.c_void_p(desc.ctypes.data), n_frames,
            C.c_void_p(out.data_ptr()), out.shape[1], C.c_void_p(out_len.data_ptr()), C.c_void_p(n_pdu.data_ptr()),
            C.c_void_p(bits.data_ptr()) if bits is not None else None, bits.shape[1] if bits is not None else 0,
            self._stream())
        self._check(rc, "gr80211_decode_batch_noalloc")

    def pack_pdus(self, out, out_len, packed, total, off=None):
        """gr80211_pack_pdus: the PDU slots out [n, stride] / out_len [n] as one run of bytes in `packed` (uint8 CUDA
        tensor), every blob rounded up to 16 bytes; total: uint64-sized CUDA tensor of one element (int64) that
        receives the run's length; off (optional, int64 [n]): where each blob starts.  Queued on the current stream."""
        rc = self.lib.gr80211_pack_pdus(
            self.h, C.c_void_p(out.data_ptr()), out.shape[1], C.c_void_p(out_len.data_ptr()), out.shape[0],
            C.c_void_p(packed.data_ptr()), packed.numel(), C.c_void_p(total.data_ptr()),
            C.c_void_p(off.data_ptr()) if off is not None else None, self._stream())
        self._check(rc, "gr80211_pack_pdus")

    # -- transmitter ---------------------------------------------------------------------
    @staticmethod
    def make_tx_desc(n):
        d = np.zeros(n, dtype=TX_DESC_DTYPE)
        d["nss"] = 1
        d["scrambler"] = 93
        d["scale"] = 1.0
        return d

    def tx_samples(self, fmt, mcs, nss, psdu_len):
        return int(self.lib.gr80211_tx_samples(int(fmt), int(mcs), int(nss), int(psdu_len)))

    def tx_batch(self, psdu, desc, n_frames, iq0, iq1=None):
        """PSDU bytes (uint8 CUDA tensor) -> frames written into the complex64 CUDA arena(s) iq0 / iq1 at each
        descriptor's out_off.  Returns the int32 CUDA tensor of per-frame sample counts (-1 = rejected)."""
        t = self.torch
        assert psdu.is_cuda and psdu.dtype == t.uint8 and psdu.is_contiguous()
        assert iq0.is_cuda and iq0.dtype == t.complex64 and iq0.is_contiguous()
        if iq1 is not None:
            assert iq1.is_cuda and iq1.dtype == t.complex64 and iq1.numel() == iq0.numel()
        ns = t.empty(max(n_frames, 1), dtype=t.int32, device=self.dev)
        dp, flags, keep = self._desc_arg(desc)
        rc = self.lib.gr80211_tx_batch(
            self.h, C.c_void_p(psdu.data_ptr()), psdu.numel(), dp, n_frames, flags, C.c_void_p(iq0.data_ptr()),
            C.c_void_p(iq1.data_ptr()) if iq1 is not None else None, iq0.numel(), C.c_void_p(ns.data_ptr()),
            self._stream())
        self._check(rc, "gr80211_tx_batch")
        return ns[:n_frames]

    def tx_mu_batch(self, psdu, desc, n_frames, q, iq0, iq1):
        """Two-user MU-MIMO frames.  q: complex64 CUDA tensor [n_q, 64, 2, 2] (FFT bin, chain, stream)."""
        t = self.torch
        assert psdu.is_cuda and psdu.dtype == t.uint8 and psdu.is_contiguous()
        assert q.is_cuda and q.dtype == t.complex64 and q.is_contiguous() and tuple(q.shape[1:]) == (64, 2, 2)
        for a in (iq0, iq1):
            assert a.is_cuda and a.dtype == t.complex64 and a.is_contiguous() and a.numel() == iq0.numel()
        ns = t.empty(max(n_frames, 1), dtype=t.int32, device=self.dev)
        dp, flags, keep = self._desc_arg(desc)
        rc = self.lib.gr80211_tx_mu_batch(
            self.h, C.c_void_p(psdu.data_ptr()), psdu.numel(), dp, n_frames, flags, C.c_void_p(q.data_ptr()), q.shape[0],
            C.c_void_p(iq0.data_ptr()), C.c_void_p(iq1.data_ptr()), iq0.numel(), C.c_void_p(ns.data_ptr()), self._stream())
        self._check(rc, "gr80211_tx_mu_batch")
        return ns[:n_frames]

    @staticmethod
    def pdus_of(res, i):
        """Host-side view of frame i's published PDUs from an rx_batch/decode_batch result."""
        ln = int(res["out_len"][i].item())
        blob = bytes(res["out"][i, :ln].cpu().numpy())
        return split_pdus(blob)
