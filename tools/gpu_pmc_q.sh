#!/bin/bash
# quick SQ-counter pass of one workload: tools/gpu_pmc_q.sh <tag> <bench flags...>   (env is passed through)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PM_FOLD_CLEAR=1 timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/log 2>&1
f=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python - $f $TAG <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "pm_" in k: print(sys.argv[2], k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, len(next(iter(d.values()))))
PY
rm -rf $OUT/p
