"""ctypes binding of the streaming C-ABI (include/gr80211_host.h, libgr80211_host.so): the host blocks driven
through their general_work-shaped entry, with tags as plain records and messages through a callback.  It is what a
GNU Radio wrapper block would do per work() call (lib/decode_impl.cc:60-162, lib/signal_impl.cc:60-210)."""
import ctypes
import os

import numpy as np

from . import rx as _rx

DECODE, RX, RX2, CAPTURE, CAPTURE2, SIGNAL, SIGNAL2, DEMOD, DEMOD2 = range(9)
TAG_KEYS = ("rad", "snr", "rssi", "format", "len", "total", "cr", "mcs", "ampdu", "trellis", "mu2x1chan",
            "cfo", "seq", "nsamp", "chan", "sssnr0", "sssnr1")
OUT_DTYPES = {SIGNAL: np.complex64, SIGNAL2: np.complex64, DEMOD: np.float32, DEMOD2: np.float32}
OPT_MU_POS, OPT_AMPDU_STRICT, OPT_DEBUG, OPT_FINISH_IN_CALL, OPT_MAX_HOLD, OPT_MIN_ROUND = 0, 1, 2, 3, 4, 5
EXPORTS = ("gr80211_block_create", "gr80211_block_destroy", "gr80211_block_set_msg_handler",
           "gr80211_block_set_option", "gr80211_block_n_inputs", "gr80211_block_forecast", "gr80211_block_work",
           "gr80211_block_flush", "gr80211_block_last_error", "gr80211_block_n_outputs", "gr80211_block_work_out",
           "gr80211_block_pending_output", "gr80211_block_alloc_input", "gr80211_block_free_input",
           "gr80211_block_pin_input", "gr80211_block_declare_pinned", "gr80211_block_send_udp")
UDP_EXPORTS = ("gr80211_udp_open", "gr80211_udp_send", "gr80211_udp_close")


class Tag(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("port", ctypes.c_uint32), ("key", ctypes.c_uint32),
                ("value", ctypes.c_double), ("cvec", ctypes.POINTER(ctypes.c_float)), ("cvec_len", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


MSG_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_int)
_lib = None


def load_host_library():
    """libgr80211_host.so with prototypes set; loads the C-ABI library first (it links against it)."""
    global _lib
    if _lib is not None:
        return _lib
    _rx.load_library()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgr80211_host.so")
    if not os.path.exists(path):
        raise RuntimeError("libgr80211_host.so is not built: run __graft_entry__.build()")
    L = ctypes.CDLL(path)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.gr80211_block_create.restype = vp
    L.gr80211_block_create.argtypes = [ci, ci, ci]
    L.gr80211_block_destroy.restype = None
    L.gr80211_block_destroy.argtypes = [vp]
    L.gr80211_block_set_msg_handler.restype = None
    L.gr80211_block_set_msg_handler.argtypes = [vp, MSG_FN, vp]
    L.gr80211_block_set_option.argtypes = [vp, ci, ci]
    L.gr80211_block_n_inputs.argtypes = [vp]
    L.gr80211_block_forecast.argtypes = [vp, ci, ctypes.POINTER(ci), ci]
    L.gr80211_block_work.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(vp), ci, ctypes.POINTER(Tag), ci,
                                     ctypes.POINTER(ci)]
    L.gr80211_block_n_outputs.argtypes = [vp]
    L.gr80211_block_work_out.argtypes = [vp, ci, ctypes.POINTER(ci), ctypes.POINTER(vp), ci, ctypes.POINTER(Tag), ci,
                                         ctypes.POINTER(vp), ci, ctypes.POINTER(Tag), ci, ctypes.POINTER(ci),
                                         ctypes.POINTER(ci)]
    L.gr80211_block_pending_output.restype = ctypes.c_longlong
    L.gr80211_block_pending_output.argtypes = [vp]
    L.gr80211_block_flush.argtypes = [vp]
    L.gr80211_block_last_error.restype = ctypes.c_char_p
    L.gr80211_block_last_error.argtypes = [vp]
    L.gr80211_block_alloc_input.restype = vp
    L.gr80211_block_alloc_input.argtypes = [vp, ctypes.c_uint64]
    L.gr80211_block_free_input.restype = None
    L.gr80211_block_free_input.argtypes = [vp, vp]
    L.gr80211_block_pin_input.argtypes = [vp, vp, ctypes.c_uint64]
    L.gr80211_block_declare_pinned.argtypes = [vp, vp, ctypes.c_uint64]
    L.gr80211_block_send_udp.argtypes = [vp, ctypes.c_char_p, ci]
    L.gr80211_udp_open.restype = vp
    L.gr80211_udp_open.argtypes = [ctypes.c_char_p, ci]
    L.gr80211_udp_send.argtypes = [vp, ctypes.c_char_p, ci]
    L.gr80211_udp_close.restype = None
    L.gr80211_udp_close.argtypes = [vp]
    _lib = L
    return L


class StreamBlock:
    """One host block.  `run(streams, tags, chunk)` plays whole input streams through it the way a scheduler
    would (windows of at most `chunk` items, unconsumed items offered again) and returns the published blobs."""

    def __init__(self, kind, batch_frames=0, device=0):
        self.lib = load_host_library()
        self.h = self.lib.gr80211_block_create(kind, batch_frames, device)
        if not self.h:
            raise RuntimeError(self.lib.gr80211_block_last_error(None).decode())
        self.kind = kind
        self.n_in = self.lib.gr80211_block_n_inputs(self.h)
        self.n_out = self.lib.gr80211_block_n_outputs(self.h)
        self.out = [[] for _ in range(self.n_out)]  # arrays produced on every output port
        self.out_tags = []                          # (offset, key name, value, complex vector or None)
        self.msgs = []
        self._cb = MSG_FN(lambda user, p, n: self.msgs.append(ctypes.string_at(p, n)))
        self.lib.gr80211_block_set_msg_handler(self.h, self._cb, None)

    def close(self):
        if self.h:
            self.lib.gr80211_block_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc < 0:
            raise RuntimeError("%s: %s" % (what, self.lib.gr80211_block_last_error(self.h).decode()))
        return rc

    def send_udp(self, host, port):
        """Every message the block publishes also leaves as one UDP datagram to host:port (host None: stop)."""
        self._check(self.lib.gr80211_block_send_udp(self.h, host.encode() if host else None, int(port)), "send_udp")

    def set_option(self, option, value):
        self._check(self.lib.gr80211_block_set_option(self.h, option, int(value)), "set_option")

    def forecast(self, noutput_items):
        req = (ctypes.c_int * self.n_in)()
        self._check(self.lib.gr80211_block_forecast(self.h, noutput_items, req, self.n_in), "forecast")
        return list(req)

    def work(self, windows, tags=(), noutput_items=8192):
        """One general_work call.  windows: one contiguous numpy array per input port; tags: (offset, port, key
        name, value[, complex vector]) inside the windows.  Returns the items consumed on every port; what a block
        with outputs produced is appended to self.out / self.out_tags."""
        assert len(windows) == self.n_in
        nin = (ctypes.c_int * self.n_in)(*[len(w) for w in windows])
        ptrs = (ctypes.c_void_p * self.n_in)(*[w.ctypes.data for w in windows])
        recs = (Tag * max(len(tags), 1))()
        keep = []
        for i, t in enumerate(tags):
            recs[i].offset, recs[i].port, recs[i].key, recs[i].value = int(t[0]), int(t[1]), TAG_KEYS.index(t[2]), float(t[3])
            if len(t) > 4 and t[4] is not None:
                v = np.ascontiguousarray(t[4], np.complex64).view(np.float32)
                keep.append(v)
                recs[i].cvec = v.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
                recs[i].cvec_len = len(v) // 2
        consumed = ctypes.c_int(0)
        if self.n_out == 0:
            self._check(self.lib.gr80211_block_work(self.h, max(len(w) for w in windows), nin, ptrs, self.n_in, recs,
                                                    len(tags), ctypes.byref(consumed)), "work")
            return consumed.value
        bufs = [np.zeros(noutput_items, OUT_DTYPES[self.kind]) for _ in range(self.n_out)]
        optrs = (ctypes.c_void_p * self.n_out)(*[b.ctypes.data for b in bufs])
        cap = 16 * (noutput_items // 48 + 1)   # no frame has more than 16 tags or fewer than 48 items
        trecs = (Tag * cap)()
        nt = ctypes.c_int(0)
        produced = self._check(self.lib.gr80211_block_work_out(
            self.h, noutput_items, nin, ptrs, self.n_in, recs, len(tags), optrs, self.n_out, trecs, cap, ctypes.byref(nt),
            ctypes.byref(consumed)), "work_out")
        for p in range(self.n_out):
            self.out[p].append(bufs[p][:produced].copy())
        for i in range(nt.value):
            r = trecs[i]
            vec = None
            if r.cvec and r.cvec_len:
                vec = np.ctypeslib.as_array(r.cvec, shape=(2 * r.cvec_len,)).view(np.complex64).copy()
            self.out_tags.append((int(r.offset), TAG_KEYS[r.key], float(r.value), vec))
        return consumed.value

    def drain(self, noutput_items=8192):
        """Fetch what a block with outputs still holds (calls without input)."""
        empty = [np.zeros(0, np.uint8 if (self.kind in (SIGNAL, SIGNAL2) and p == 0) else np.complex64)
                 for p in range(self.n_in)]
        guard = 0
        while self.n_out and self.lib.gr80211_block_pending_output(self.h) > 0 and guard < 1 << 20:
            self.work(empty, (), noutput_items)
            guard += 1

    def output(self, port=0):
        return np.concatenate(self.out[port]) if self.out[port] else np.zeros(0, OUT_DTYPES[self.kind])

    def flush(self):
        self._check(self.lib.gr80211_block_flush(self.h), "flush")

    def run(self, streams, tags=(), chunk=4096, noutput_items=8192):
        """Feeds whole streams through work() in windows of `chunk` items with room for `noutput_items` on every output;
        either may be a callable that names the next call's size (a scheduler's windows vary from call to call)."""
        streams = [np.ascontiguousarray(s) for s in streams]
        n = min(len(s) for s in streams)
        tags = sorted(tags, key=lambda t: (t[1], t[0]))
        at, idle, grow = 0, 0, 1
        nout = noutput_items if callable(noutput_items) else (lambda: noutput_items)
        while at < n:
            win = min((chunk() if callable(chunk) else chunk) * grow, n - at)
            got = self.work([s[at:at + win] for s in streams], [t for t in tags if at <= t[0] < at + win], nout())
            # tags of items not consumed are handed over again with the next window, like a scheduler's tag store
            at += got
            if got == 0:
                idle += 1
                if win == n - at or idle > 4:
                    break
                grow *= 2
            else:
                idle, grow = 0, 1
        self.flush()
        self.drain(nout())
        return self.msgs
