#!/bin/bash
# copy a profile round's summaries from gpurun_out/<tag>* into profiles/ (tracked): tools/collect_profiles.sh r04_a
set -u
cd "$(dirname "$0")/.."
TAG=$1
for d in gpurun_out/${TAG} gpurun_out/${TAG}_cfg2 gpurun_out/${TAG}_cfg4 gpurun_out/${TAG}_cfg5 gpurun_out/${TAG}_held1 gpurun_out/${TAG}_held2 gpurun_out/${TAG}_held3; do
  [ -d $d ] || continue
  t=$(basename $d)
  for f in bench.json bench_traced.json bench_serial_traced.json kernel_stats.csv serial_kernel_stats.csv pmc_summary.json; do
    [ -s $d/$f ] && cp $d/$f profiles/${t}_$f
  done
done
python tools/make_traffic.py profiles/${TAG}_pmc_summary.json profiles/hbm_traffic.json ${TAG} >/dev/null
# (the other workloads' PMC digests, what `bench.py --workload <w>` copies its traffic / issue blocks from)
for w in cfg2:config2 cfg4:config4 cfg5:config5 held1:held1 held2:held2 held3:held3; do
  s=${w%%:*}; n=${w##*:}
  [ -s profiles/${TAG}_${s}_pmc_summary.json ] && python tools/make_traffic.py profiles/${TAG}_${s}_pmc_summary.json profiles/hbm_traffic_${n}.json ${TAG}_${s} >/dev/null
done
ls profiles | grep "^${TAG}" | wc -l
