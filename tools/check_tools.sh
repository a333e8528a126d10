#!/bin/bash
# Runs every stand-alone script under tools/ once against the library as built (smallest arguments each takes) and
# prints one line per script: ok, or the exit status and the last line it printed.  The index in tools/README.md is
# kept from this output.
set -u
run() {  # <seconds> <command...>
  local t=$1; shift
  local log; log=$(mktemp)
  timeout -k 10 "$t" "$@" > "$log" 2>&1
  local rc=$?
  if [ $rc -eq 0 ]; then echo "ok      $*"; else echo "rc=$rc  $*  :: $(grep -v amdgpu.ids "$log" | tail -n 1 | cut -c1-160)"; fi
  rm -f "$log"
  # a time limit or a crash ends the run: nothing more is started on the GPU after it
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 134 ] || [ $rc -eq 139 ]; then exit 1; fi
}
run 200 python3 tools/overlap_exp.py
run 200 python3 tools/pair_exp.py
run 200 python3 tools/stream_exp.py
run 200 python3 tools/vit_bench.py
run 200 python3 tools/overlap_probe.py
run 300 python3 tools/pdr_sweep.py
run 200 python3 tools/chain_rate.py 4000
run 200 python3 tools/latency_sweep.py
run 200 python3 tools/rx_latency.py 1 64
run 200 python3 tools/block_latency.py
run 200 python3 tools/live_latency.py
run 200 python3 tools/decode_sweep.py
run 200 python3 tools/parity_sweep.py
run 200 python3 tools/detect_sweep.py
run 200 python3 tools/axis_parity.py
run 300 python3 tools/stream_rate.py 4000
