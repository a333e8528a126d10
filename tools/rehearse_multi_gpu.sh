#!/bin/bash
# Rehearsal of bench.py's N > 1 path on a box with ONE GPU: 2 and 4 ranks share the card, the collectives run over
# gloo on host tensors (GR80211_BENCH_BACKEND=gloo), everything else -- sharded corpora seeded by rank, the gather of
# the PDU slots to rank 0 in every step, rank 0's check of every rank's slice -- is the code the 8-GPU run executes.
# The rates it prints say nothing about scaling (the ranks share one GPU); the JSON lines are kept as evidence that
# the path runs end to end (two contexts in rotation per rank instead of the default six: here the ranks share one
# GPU's memory).   bash tools/rehearse_multi_gpu.sh <out dir> [workload]
set -u
OUT=${1:-gpurun_out}
WL=${2:-mixed_l_ht_vht_snr30}
mkdir -p "$OUT"
export GR80211_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
for N in 2 4; do
  echo "[rehearse] N=$N workload=$WL"
  timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29510 + N)) \
      bench.py --gpus $N --steps 4 --warmup 1 --pipeline 2 --workload $WL --in-flight 1 --profile-steps 0 > "$OUT/bench_gloo_N$N.log" 2> "$OUT/bench_gloo_N$N.err" || { tail -5 "$OUT/bench_gloo_N$N.err"; exit 1; }
  grep '^{' "$OUT/bench_gloo_N$N.log" | python3 -c "import sys, json; d = json.loads(sys.stdin.readline()); print(d['n_gpus'], d['value'], d['gathered_on_rank0'])"
done
