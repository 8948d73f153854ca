cd $GRAFT_REPO_ROOT
export SMX_BENCH_ONE_DEVICE=1 SMX_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
two() { env $1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 2 --steps 2 --warmup 1 --batch 6 --no-cpu-baseline --no-roofline --dtype bf16 2>&1 | grep -c "consistency FAILED"; }
port=30100
for e in "SMX_WARP_NT=0" "SMX_WARP_NT=1"; do
  f=0; for i in 1 2 3 4 5 6 7 8 9 10; do port=$((port+1)); r=$(two "$e" $port); f=$((f+r)); done; echo "== $e: failed $f of 10"
done
