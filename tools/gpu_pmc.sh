#!/bin/bash
# one PMC pass of the default bench per counter set: tools/gpu_pmc.sh "FETCH_SIZE" "WRITE_SIZE" ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
{
for set in "$@"; do
  rm -rf /tmp/qp
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/qp -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 ${PM_PMC_FLAGS:-} > /tmp/qp.log 2>&1
  f=$(find /tmp/qp -name "*counter_collection.csv" | head -1)
  python - $f <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
    if "pm_" in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    for c, v in d.items(): print(k, c, round(sum(v) / len(v), 1), "x", len(v))
PY
done
} > $ROOT/gpurun_out/pmc.log 2>&1
cat $ROOT/gpurun_out/pmc.log
