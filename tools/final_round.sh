#!/bin/bash
# The round-closing GPU call: smoke, full GPU suite, bench (+rocprof, PMC) in fp32, the bf16 PMC pass, the training-step forms, the VQ table.
# usage: tools/final_round.sh <tag>
TAG=${1:-r03k}; OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/gpu_round.sh $TAG tests pmc
bash tools/gpu_round.sh ${TAG}_bf16 notests pmc --dtype bf16 --no-cpu-baseline
for a in "--perceptual --graph" "--perceptual --graph --gan" "--perceptual --graph --dtype bf16" "--perceptual --graph --gan --dtype bf16" "--perceptual" "--perceptual --gan"; do
  python tools/train_bench.py $a 2>/dev/null | tail -1
done | tee $OUT/train_bench_$TAG.txt
python tools/vq_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/vq_bench_$TAG.txt
