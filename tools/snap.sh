#!/bin/bash
# A mid-round snapshot: the default bench line (fp32 + bf16 + training legs), rocprofv3 kernel stats of the fp32 / bf16 inference steps and of the training step.
# usage: tools/snap.sh <tag>
TAG=${1:-r04a}; OUT=$PWD/gpurun_out; mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$PWD
timeout 1200 python bench.py --dump-shapes $OUT/shapes_$TAG.txt > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline --dump-shapes $OUT/shapes_${TAG}_bf16.txt > $OUT/bench_${TAG}_bf16.json 2> $OUT/bench_${TAG}_bf16.err; echo "bench16 rc=$?"
cd /tmp && export TMPDIR=/tmp
for d in f32 bf16; do
  rm -rf $OUT/prof_${TAG}_$d; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_$d -o run --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --profile-only --dtype $d > $OUT/prof_${TAG}_$d.log 2>&1; echo "rocprof $d rc=$?"
  rm -rf $OUT/prof_${TAG}_train_$d; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_train_$d -o run --output-format csv -- python $R/tools/train_bench.py --perceptual --dtype $d --steps 5 --warmup 2 > $OUT/prof_${TAG}_train_$d.log 2>&1; echo "rocprof train $d rc=$?"
done
find $OUT -name "*kernel_trace.csv" -size +20M -delete
cd $R
for a in "--perceptual --graph" "--perceptual --graph --dtype bf16"; do python tools/train_bench.py $a 2>/dev/null | tail -1; done | tee $OUT/train_bench_$TAG.txt
du -sh $OUT
