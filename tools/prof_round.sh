#!/bin/bash
# Round profile (run on the GPU box through gpurun):  tools/prof_round.sh <tag>
#   1. bench.py (default flags) -> gpurun_out/<tag>/bench.json
#   2. rocprofv3 --kernel-trace --stats of the same command -> kernel_stats.csv (+ that run's own
#      JSON line -> bench_traced.json); again with PM_FRAME_STREAMS=1 PM_SLOTS=1 -> serial_kernel_stats.csv
#      (+ bench_serial_traced.json): launches that do not overlap
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE, then SQ counters), each in its own run with
#      --kernel-trace only, -> pmc_passN.csv + pmc_summary.json
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py"
WL=${PM_PROF_FLAGS:---no-config5}   # the traced / counted runs hold ONE workload: kernel averages are per configuration
timeout 900 $BENCH ${PM_PROF_MAIN:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.json
# the traced run's own JSON line is kept next to the stats: same process, same launches
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH --no-cpu-baseline $WL > $OUT/trace.log 2>&1
grep '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench_traced.json
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats.csv && head -8 $OUT/kernel_stats.csv
find $OUT/trace -name "*kernel_trace.csv" -size +30M -delete
# the same command with frames SERIALIZED (one stream, one frame slot): every launch alone on the GPU, so the
# stats' AverageNs is a duration that fits inside a step (with four frames in flight the launches stretch each
# other); this is the trace roofline.frac / kernel_ms of the bench line agree with
PM_FRAME_STREAMS=1 PM_SLOTS=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_serial -- $BENCH --no-cpu-baseline $WL > $OUT/trace_serial.log 2>&1
grep '^{"metric"' $OUT/trace_serial.log | tail -1 > $OUT/bench_serial_traced.json
f=$(find $OUT/trace_serial -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/serial_kernel_stats.csv && head -5 $OUT/serial_kernel_stats.csv
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  # (PM_FOLD_CLEAR=1: every frame of the counted runs is two launches, clearing inside the tile kernel's -- the
  #  per-launch averages then add up to one frame; by default frames behind other frames clear in a launch of their own)
  PM_FOLD_CLEAR=1 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- $BENCH --steps 20 --warmup 5 --no-cpu-baseline $WL > $OUT/pmc$i.log 2>&1
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/pmc_pass$i.csv || tail -5 $OUT/pmc$i.log
done
python - $OUT <<'PY'
import csv, sys, collections, glob, json, os
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc_pass*.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].strip()
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"launches": len(next(iter(d.values())))} for k, d in agg.items() if "pm_" in k}
json.dump(summ, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
PY
rm -rf $OUT/trace $OUT/trace_serial $OUT/pmc[0-9]
# (what travels back is capped at 64 MiB for the whole of gpurun_out/: the per-launch counter rows are summarised above)
[ "${PM_PROF_KEEP_PASSES:-0}" = "1" ] || rm -f $OUT/pmc_pass*.csv
ls -la $OUT
