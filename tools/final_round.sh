#!/bin/bash
# The round-closing GPU call: smoke, full GPU suite, rocprof + PMC passes, THEN the bench (fp32; it quotes those counters), the bf16 PMC pass + bench, the
# training-step forms, the VQ table, the split-bf16 A/B tables (Winograd, 1x1 GEMM, attention), the file-to-file figure.
# usage: tools/final_round.sh <tag>        (tag = <round>[suffix], e.g. r06a: the counter summaries land in profiles/<round>_*_pmc*.json ON THE BOX and in gpurun_out/)
TAG=${1:-r06a}; OUT=$PWD/gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
RT=$(echo $TAG | sed -E 's/^(r[0-9]+).*/\1/')
bash tools/gpu_round.sh ${RT}_$TAG tests pmc
bash tools/gpu_round.sh ${RT}_${TAG}_bf16 notests pmc --dtype bf16 --no-cpu-baseline
for a in "--perceptual --graph" "--perceptual --graph --gan" "--perceptual --graph --dtype bf16" "--perceptual --graph --gan --dtype bf16" "--perceptual" "--perceptual --gan"; do
  python tools/train_bench.py $a 2>/dev/null | tail -1
done | tee $OUT/train_bench_$TAG.txt
python tools/vq_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/vq_bench_$TAG.txt
python tools/wino_bf3_bench.py 300 2>&1 | grep -v amdgpu.ids | tee $OUT/wino_split_$TAG.txt
python tools/wino_bf3_bench.py 300 plain 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wino_split_$TAG.txt
python tools/gemm_bf3_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/gemm_split_$TAG.txt
python tools/attn_bf3_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/attn_split_$TAG.txt
python -m pytest tests/test_gpu_attn_bf3.py tests/test_gpu_gemm_bf3.py -q -s -k "not_less_accurate" 2>&1 | grep -E "max\|err\|" | tee -a $OUT/attn_split_$TAG.txt
for d in f32 bf16; do python tools/pipe_bench.py $d --from-png --to-png 1200 60 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $OUT/file_to_file_$TAG.txt
