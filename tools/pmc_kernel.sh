#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> "<counters>" <cmd...>   -- one rocprofv3 --pmc pass, per-kernel averages printed
TAG=$1; CNT=$2; shift 2
OUT=$PWD/gpurun_out/pmck_$TAG; rm -rf $OUT
REPO=$PWD; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc $CNT -d $OUT -o run --output-format csv -- "$@" > $OUT.log 2>&1
cd $REPO
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "at::" in n or "hip" in n[:4]: continue
        n = n.replace("(anonymous namespace)::", "").split("(")[0][:60]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in agg.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
