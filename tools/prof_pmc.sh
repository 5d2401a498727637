#!/bin/bash
# PMC passes for the two frame kernels (run on the GPU box through gpurun).
# Counters are collected in their own runs, with --kernel-trace only (never with
# sys/hip/hsa tracing), as the pool requires.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
i=0
shift
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pass$i -- $CMD > $OUT/pass$i.log 2>&1
  f=$(find $OUT/pass$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "pm_" not in k: continue
    print(k, {c: round(sum(v)/len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
  else
    tail -5 $OUT/pass$i.log
  fi
done
