#!/usr/bin/env python3
"""profiles/counters.json from the rocprofv3 pass summaries of one profiling round (tools/profile_round.sh):

    python3 tools/make_counters_json.py <dir with kernel_stats.txt and pmc_*.txt> <workload> <frames> <tag>

Per kernel of interest: the average launch duration of the kernel trace and the per-dispatch average of every
counter found in the pmc_*.txt summaries (FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB).  bench.py
reads the file to fill `roofline.traffic` and `roofline.valu_issue`."""
import glob
import json
import os
import re
import sys

# (the counter summaries cut kernel names at 50 characters: k_decode_lane_stat, or k_decode_lane behind "void ")
KERNELS = {"k_decode": "k_decode(?:_lean|_split2|_lane(?:_stat[e]?)?)?", "k_demod": "k_demod", "k_preamble": "k_preamble", "k_viterbi_split": "k_viterbi_split",
           "k_traceback": "k_traceback", "k_det_ac": "k_det_ac"}


def kernel_key(name):
    for k, pat in KERNELS.items():
        if re.search(r"\b%s\b" % pat, name):
            return k
    return None


def main(d, workload, frames, tag):
    out = {"workload": workload, "frames": int(frames),
           "source": "profiles/%s_kernel_stats.txt, profiles/%s_pmc_*.txt (rocprofv3 --kernel-trace --stats; one --pmc pass per "
                     "counter group)" % (tag, tag)}
    ks = os.path.join(d, "kernel_stats.txt")
    if os.path.exists(ks):
        for line in open(ks):
            m = re.match(r"(.{70})\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)\s+(\d+)", line)
            if m:
                k = kernel_key(m.group(1))
                if k and k not in out:
                    out[k] = {"duration_ns": float(m.group(4)), "launches": int(m.group(2))}
                    if k == "k_decode":  # which of the decoder's forms the batch size selected
                        out[k]["kernel"] = re.search(r"k_decode\w*(?:<[^>]*>)?", m.group(1)).group(0)
    for fn in sorted(glob.glob(os.path.join(d, "pmc_*.txt"))):
        seen = False
        for line in open(fn):
            if "avg value per dispatch" in line:
                seen = True
                continue
            if not seen:
                continue
            m = re.match(r"(.{50})\s+(\S+)\s+(\d+)\s+([0-9.eE+-]+)", line)
            if m:
                k = kernel_key(m.group(1))
                if k:
                    name = m.group(2)
                    if name in ("FETCH_SIZE", "WRITE_SIZE"):
                        name += "_KiB"
                    out.setdefault(k, {})[name] = float(m.group(4))
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(*sys.argv[1:5])
