#!/bin/bash
# One gpurun call: GPU tests, rocprofv3 kernel stats, PMC passes (MFMA + traffic) and their summaries, THEN the bench (+shape table) -- so that the bench record
# quotes counters of its own build and box (round 5's record carried `traffic: null` because the order was the other way round).
# usage: tools/gpu_round.sh <tag> [tests|notests] [pmc|nopmc] [extra bench args...]
TAG=${1:-r02a}; TESTS=${2:-tests}; PMC=${3:-pmc}; shift 3 || true
OUT=$PWD/gpurun_out; mkdir -p $OUT; R0=$PWD
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$TESTS" = "tests" ]; then
  timeout 3000 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_$TAG.log
  tail -5 $OUT/pytest_$TAG.log
fi
BENCH="python $PWD/bench.py --steps 2 --warmup 1 --profile-only $@"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG; timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o run --output-format csv -- $BENCH > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
if [ "$PMC" = "pmc" ]; then
  rm -rf $OUT/pmc_mfma_$TAG; timeout 900 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc_mfma_$TAG -o run --output-format csv -- $BENCH > $OUT/pmc_mfma_$TAG.log 2>&1; echo "pmc mfma rc=$?"
  rm -rf $OUT/pmc_fetch_$TAG; timeout 900 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch_$TAG -o run --output-format csv -- $BENCH > $OUT/pmc_fetch_$TAG.log 2>&1; echo "pmc fetch rc=$?"
  rm -rf $OUT/pmc_write_$TAG; timeout 900 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write_$TAG -o run --output-format csv -- $BENCH > $OUT/pmc_write_$TAG.log 2>&1; echo "pmc write rc=$?"
  # drop the multi-hundred-MB raw traces, keep the counter csv
  find $OUT/pmc_mfma_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG -name "*kernel_trace.csv" -delete 2>/dev/null
  du -sh $OUT/pmc_mfma_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG 2>/dev/null
  # summarised HERE, on the library these counters were taken on (the JSON carries its per-object build digests; bench.py refuses a mismatch)
  python $R0/tools/pmc_traffic.py $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/traffic_pmc_$TAG.json > /dev/null 2>&1; echo "traffic summary rc=$?"
  python $R0/tools/pmc_mfma.py $OUT/pmc_mfma_$TAG $OUT/mfma_pmc_$TAG.json "$BENCH" > /dev/null 2>&1; echo "mfma summary rc=$?"
  # the bench below quotes THESE counters (same library build, same box): profiles/<round>_{traffic,mfma}_pmc[_bf16].json
  RT=${TAG%%_*}; SFX=""; case "$TAG" in *_bf16) SFX="_bf16";; esac
  cp $OUT/traffic_pmc_$TAG.json $R0/profiles/${RT}_traffic_pmc$SFX.json; cp $OUT/mfma_pmc_$TAG.json $R0/profiles/${RT}_mfma_pmc$SFX.json
fi
cd $R0

cd $R0
timeout 900 python bench.py --dump-shapes $OUT/shapes_$TAG.txt "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
python - <<PY
import json
try:
    j = json.loads([l for l in open("$OUT/bench_$TAG.json") if l.startswith("{")][-1])
    print("fps", j["value"], "ms/step", j["ms_per_step"], "roofline", {k: j.get("roofline", {}).get(k) for k in ("achieved", "frac", "achieved_algorithmic", "share_of_step_time")},
          "pcie", j.get("value_incl_pcie"), "cons", j.get("batch_consistency", {}).get("max_lsb_vs_b1"), "cpu", (j.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("bench parse failed", e)
PY
head -12 $OUT/shapes_$TAG.txt
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {} | cut -c1-160
du -sh $OUT/prof_$TAG
