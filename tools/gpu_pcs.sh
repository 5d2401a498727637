#!/bin/bash
# developer profile: PC sampling (host trap) of one bench workload: tools/gpu_pcs.sh <tag> <interval> <bench flags...>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
IV=$1; shift
OUT=$ROOT/gpurun_out/pcs_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>&1 | grep -i -A6 "pc sampl\|PC_SAMPL\|host_trap\|stochastic" | head -30 > $OUT/avail.txt
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval $IV --kernel-trace --output-format csv -d $OUT/p -- python $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/log 2>&1
echo "rc=$?" >> $OUT/log
find $OUT/p -type f | head -20 >> $OUT/log
f=$(find $OUT/p -name "*pc_sampling*csv" | head -1)
if [ -n "$f" ]; then
  head -3 $f > $OUT/head.txt
  wc -l $f >> $OUT/head.txt
  python - $f $OUT <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
c = collections.Counter(); n = 0
cols = None
for r in rows:
    cols = cols or list(r.keys())
    n += 1
    key = (r.get("Instruction_Comment") or "", r.get("Instruction") or "")
    c[key] += 1
open(sys.argv[2] + "/hist.txt", "w").write("cols %s samples %d\n" % (cols, n) + "\n".join("%7d %s | %s" % (v, k[0], k[1]) for k, v in c.most_common(4000)))
PY
fi
rm -rf $OUT/p
tail -5 $OUT/log; cat $OUT/avail.txt | head -10; head -c 1500 $OUT/head.txt
