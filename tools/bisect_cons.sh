#!/bin/bash
# The shared-device anomaly ("open issue"): bench.py --gpus 2 with both ranks on ONE GPU, N times per configuration,
# counting failed batch-consistency checks.  SMX_SHARED_DEVICE=0 re-enables the LDS-DMA kernels the harness mode switches off.
# usage (on the GPU box): bash tools/bisect_cons.sh [runs per configuration]
cd ${GRAFT_REPO_ROOT:-$PWD}
N=${1:-10}
export SMX_BENCH_ONE_DEVICE=1 SMX_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
two() { env $1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 2 --steps 2 --warmup 1 --batch 6 --no-cpu-baseline --no-roofline --dtype bf16 2>&1 | grep -c "consistency FAILED"; }
port=30100
for e in "SMX_FORCE_LDSDMA=0" "SMX_FORCE_LDSDMA=1" "SMX_FORCE_LDSDMA=1 SMX_GEMM16_RP=0" "SMX_FORCE_LDSDMA=1 SMX_HEADS_X3=0 SMX_ATTNBLOCK_FUSED16=0"; do
  f=0; for i in $(seq $N); do port=$((port+1)); r=$(two "$e" $port); f=$((f+r)); done; echo "== $e: failed $f of $N"
done
