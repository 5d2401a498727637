#!/bin/bash
# copy a profile round's summaries from gpurun_out/<tag>* into profiles/ (tracked): tools/collect_profiles.sh r04_a
set -u
cd "$(dirname "$0")/.."
TAG=$1
for d in gpurun_out/${TAG} gpurun_out/${TAG}_cfg2 gpurun_out/${TAG}_cfg4 gpurun_out/${TAG}_cfg5; do
  [ -d $d ] || continue
  t=$(basename $d)
  for f in bench.json bench_traced.json bench_serial_traced.json kernel_stats.csv serial_kernel_stats.csv pmc_summary.json; do
    [ -s $d/$f ] && cp $d/$f profiles/${t}_$f
  done
done
python tools/make_traffic.py profiles/${TAG}_pmc_summary.json profiles/hbm_traffic.json ${TAG} >/dev/null
ls profiles | grep "^${TAG}" | wc -l
