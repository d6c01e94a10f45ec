#!/bin/bash
# The multi-GPU scaling run of bench.py, one rank per GPU over RCCL/xGMI, N = 1 2 4 8 back to back (or the counts given).
#   tools/launch_scale.sh [N ...]            e.g. tools/launch_scale.sh 1 2 4 8
# Each line of output is rank 0's JSON line for that N.  `python bench.py --gpus N` alone does the same for one N (it
# starts its own ranks when no launcher set WORLD_SIZE); this script is the explicit form of that recipe.
# FMMBEM_BENCH_BACKEND=gloo rehearses the N > 1 path on a ONE-GPU box (ranks share device 0, collectives through the host:
# a correctness rehearsal, the numbers mean nothing).
set -euo pipefail
cd "$(dirname "$0")/.."
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-3}; PORT=${PORT:-29541}
for n in "${@:-1 2 4 8}"; do
  for g in $n; do
    if [ "$g" = 1 ]; then
      python3 bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" ${BENCH_ARGS:-}
    else
      python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$g" --master-addr 127.0.0.1 --master-port "$PORT" \
        bench.py --gpus "$g" --steps "$STEPS" --warmup "$WARMUP" ${BENCH_ARGS:---no-cpu-baseline}
    fi
  done
done
