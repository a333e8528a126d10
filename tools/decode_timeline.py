#!/usr/bin/env python3
"""Per-wave timeline of k_decode (debug build with -DGR80211_TIMELINE, selected through GR80211_RX_LIB):
when each group's forward pass and traceback end, and how many waves shared its SIMD.
usage: GR80211_RX_LIB=.../libgr80211_rx_tl.so decode_timeline.py [frames]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

pkg = importlib.import_module("fork-gr-ieee80211_amd")
dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
arenas, sync, navail, exp, idx, L = bench.build_corpus(torch, pkg.txgen, "vht_mcs8_1500B_snr30", n, 64, 0, dev)
ctx = pkg.RxContext(0)
desc = torch.from_numpy(ctx.make_desc(sync, navail).view(np.uint8).reshape(-1)).to(dev)
out = torch.zeros((n, 1536), dtype=torch.uint8, device=dev)
out_len = torch.zeros(n, dtype=torch.int32, device=dev)
n_pdu = torch.zeros(n, dtype=torch.int32, device=dev)
for _ in range(2):
    ctx.rx_batch_into(arenas[0], None, desc, n, out, out_len, n_pdu)
    ctx.sync()
g = (n + 63) // 64
buf = np.zeros(4 * g, dtype=np.uint64)
rc = ctx.lib.gr80211_debug_timeline(C.c_void_p(buf.ctypes.data), g)
assert rc == 0, rc
t = buf.reshape(g, 4)
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0   # us
fwd = (t[:, 1] - t0) / 100.0
end = (t[:, 2] - t0) / 100.0
hw = t[:, 3].astype(np.int64)
# HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; the XCC is not in it, so
# waves are keyed by (hw id without wave slot, start order) only for the per-SIMD census inside one XCC
simd_key = (hw >> 4) & 0xfff
print("groups %d  kernel span %.0f us" % (g, end.max()))
for name, v in (("start", start), ("forward end", fwd), ("end", end), ("traceback", end - fwd)):
    q = np.percentile(v, [0, 10, 50, 90, 100])
    print("%-12s min %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f us" % ((name,) + tuple(q)))
late = fwd > np.percentile(fwd, 40)
print("forward end histogram (ms):", np.histogram(fwd / 1000.0, bins=12)[0].tolist(), np.histogram(fwd / 1000.0, bins=12)[1].round(1).tolist())
print("end histogram (ms):", np.histogram(end / 1000.0, bins=12)[0].tolist(), np.histogram(end / 1000.0, bins=12)[1].round(1).tolist())
print("traceback of early finishers: %.0f us median; of late finishers: %.0f us median" %
      (np.median((end - fwd)[~late]), np.median((end - fwd)[late])))
