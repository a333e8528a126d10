#!/bin/bash
# The multi-GPU leg on one GPU: the exchange tests, bench with a one-rank RCCL group, the capture corpus, and
# `bench.py --gpus 2` launching itself (gloo rehearsal: two ranks share the GPU).  Logs under gpurun_out/.
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 600 python -u -m pytest tests/test_gpu_gather.py -x -v --timeout 300 --timeout-method thread > gpurun_out/r03_gather_tests.log 2>&1 || { tail -40 gpurun_out/r03_gather_tests.log; exit 1; }
tail -3 gpurun_out/r03_gather_tests.log
echo "== nccl N=1"; timeout -k 10 400 python bench.py --force-collective --cpu-frames 0 --steps 20 > gpurun_out/r03_bench_nccl_N1.log 2> gpurun_out/r03_bench_nccl_N1.err || { tail -30 gpurun_out/r03_bench_nccl_N1.err; exit 1; }
echo "== nccl N=1 capture corpus"; timeout -k 10 400 python bench.py --force-collective --cpu-frames 0 --steps 20 --workload capture_corpus_replicated > gpurun_out/r03_bench_nccl_N1_capture.log 2> gpurun_out/r03_bench_nccl_N1_capture.err || { tail -30 gpurun_out/r03_bench_nccl_N1_capture.err; exit 1; }
echo "== self-launch N=2 (gloo rehearsal)"; timeout -k 10 500 python bench.py --gpus 2 --steps 10 --frames 40000 --capture-corpus 1 > gpurun_out/r03_bench_selflaunch_N2.log 2> gpurun_out/r03_bench_selflaunch_N2.err || { tail -30 gpurun_out/r03_bench_selflaunch_N2.err; exit 1; }
echo "== plain N=1"; timeout -k 10 500 python bench.py > gpurun_out/r03_bench_first.log 2> gpurun_out/r03_bench_first.err || { tail -30 gpurun_out/r03_bench_first.err; exit 1; }
for f in r03_bench_nccl_N1 r03_bench_nccl_N1_capture r03_bench_selflaunch_N2 r03_bench_first; do echo "-- $f"; python - "$f" <<'PY'
import json,sys
for line in open("gpurun_out/%s.log" % sys.argv[1]):
    if line.startswith("{"):
        r=json.loads(line)
        print({k:r[k] for k in ("value","ms_per_step","n_gpus")}, r["psdu"], r.get("gathered_on_rank0",{}).get("ranks"), (r.get("capture_corpus") or {}).get("value"), (r.get("capture_corpus") or {}).get("psdu"))
PY
done
