#!/bin/bash
# gpurun_out/* of a tools/final_round.sh <tag> call -> the tracked, per-round names under profiles/.   usage: tools/collect_profiles.sh <tag>   (e.g. r06a)
TAG=${1:-r06a}; RT=$(echo $TAG | sed -E 's/^(r[0-9]+).*/\1/'); G=gpurun_out; P=profiles
last_json() { grep '^{' "$1" | tail -1; }
last_json $G/bench_${RT}_$TAG.json > $P/${RT}_bench.json
last_json $G/bench_${RT}_${TAG}_bf16.json > $P/${RT}_bench_bf16.json
cp "$(find $G/prof_${RT}_$TAG -name '*kernel_stats.csv' | head -1)" $P/${RT}_kernel_stats.csv
cp "$(find $G/prof_${RT}_${TAG}_bf16 -name '*kernel_stats.csv' | head -1)" $P/${RT}_kernel_stats_bf16.csv
cp $G/traffic_pmc_${RT}_$TAG.json $P/${RT}_traffic_pmc.json; cp $G/mfma_pmc_${RT}_$TAG.json $P/${RT}_mfma_pmc.json
cp $G/traffic_pmc_${RT}_${TAG}_bf16.json $P/${RT}_traffic_pmc_bf16.json; cp $G/mfma_pmc_${RT}_${TAG}_bf16.json $P/${RT}_mfma_pmc_bf16.json
cp $G/shapes_${RT}_$TAG.txt $P/${RT}_gemm_shapes.txt; cp $G/shapes_${RT}_${TAG}_bf16.txt.bf16 $P/${RT}_gemm_shapes_bf16.txt 2>/dev/null
cp $G/train_bench_$TAG.txt $P/${RT}_train_bench.txt; cp $G/vq_bench_$TAG.txt $P/${RT}_vq_bench.txt
cp $G/wino_split_$TAG.txt $P/${RT}_wino_split.txt; cp $G/gemm_split_$TAG.txt $P/${RT}_gemm_split.txt; cp $G/attn_split_$TAG.txt $P/${RT}_attn_split.txt; cp $G/file_to_file_$TAG.txt $P/${RT}_file_to_file.txt
tail -3 $G/pytest_${RT}_$TAG.log > $P/${RT}_pytest_gpu.txt
ls -la $P | grep " ${RT}_" | awk '{print $5, $9}'
