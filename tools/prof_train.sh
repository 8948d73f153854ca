R=$PWD; OUT=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
for d in f32 bf16; do
  rm -rf $OUT/prof_r04j_train_$d; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_r04j_train_$d -o run --output-format csv -- python $R/tools/train_bench.py --perceptual --dtype $d --steps 5 --warmup 2 > $OUT/prof_r04j_train_$d.log 2>&1
  find $OUT/prof_r04j_train_$d -name "*kernel_trace.csv" -delete
done
