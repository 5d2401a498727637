#!/bin/bash
# quick parity + bench + one PMC pass (HBM traffic) of the default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/quickpmc
export TMPDIR=/tmp
{
PM_FUZZ=${PM_FUZZ:-100} bash tools/gpu_quick.sh
cd /tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/qp_$set -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-config5 > /tmp/qp_$set.log 2>&1
  f=$(find /tmp/qp_$set -name "*counter_collection.csv" | head -1)
  python - $f $set <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
    if "pm_" in k: agg[k].append(float(r["Counter_Value"]))
for k, v in agg.items(): print(sys.argv[2], k, round(sum(v) / len(v), 1), "KiB x", len(v))
PY
done
} > $GRAFT_REPO_ROOT/gpurun_out/call17.log 2>&1
tail -20 $GRAFT_REPO_ROOT/gpurun_out/call17.log
