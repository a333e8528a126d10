#!/bin/bash
# Counter passes over tools/rx_latency.py for given batch sizes (lane-state decoder): cycles, waits, instruction mix.
#   bash tools/rx_latency_pmc.sh 1 2
set -u
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
cd "$ROOT"
for n in "$@"; do
  i=0
  for pass in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
              "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU"; do
    i=$((i + 1))
    OUT=$ROOT/gpurun_out/rxlat_pmc_${n}_$i
    mkdir -p "$OUT"
    timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $pass -d "$OUT/p" -o p -- python3 tools/rx_latency.py "$n" > "$OUT/run.log" 2>&1 || { tail -5 "$OUT/run.log"; exit 3; }
    python3 tools/rocpd_summary.py "$OUT"/p/p_results.db > "$OUT/summary.txt" 2>&1
    rm -rf "$OUT/p"
    echo "== n=$n pass $i"; grep -E "k_decode_lane" "$OUT/summary.txt"
  done
done
