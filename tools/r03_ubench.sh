#!/bin/bash
# Micro-benchmark evidence kept under profiles/: per-instruction issue rates (tools/ubench/valu2, valu3), streaming
# rates (hbm), and the check of how FETCH_SIZE / WRITE_SIZE count on this chip: kernels of known traffic (a read of
# 3 GiB, a fill of 9 GiB, a copy of 3 GiB in and out) under rocprofv3 --pmc, one counter per pass.
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r03_ubench
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 120 tools/ubench/valu2 > "$OUT/valu2.txt" 2>&1 || exit 1
timeout -k 10 120 tools/ubench/valu3 > "$OUT/valu3.txt" 2>&1 || exit 1
timeout -k 10 120 tools/ubench/hbm > "$OUT/hbm.txt" 2>&1 || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $ctr -d "$OUT/p_$ctr" -o p -- tools/ubench/hbm > "$OUT/p_$ctr.log" 2>&1 || exit 2
    python3 tools/rocpd_summary.py "$OUT/p_$ctr"/p_results.db > "$OUT/hbm_pmc_$ctr.txt" 2>&1
    rm -rf "$OUT/p_$ctr"
done
cat "$OUT/valu2.txt"; cat "$OUT"/hbm_pmc_*.txt
