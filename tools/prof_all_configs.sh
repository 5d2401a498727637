#!/bin/bash
# Profile rounds of all bench workloads: tools/prof_all_configs.sh <tag>  -> tools/prof_round.sh for the Tiger (<tag>) and configs 2, 4, 5 (<tag>_cfgN)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
TAG=${1:-r03_a}
bash tools/prof_round.sh $TAG 2>&1 | grep -v '^ \|^{\|^}' | tail -3
for cfg in 2 4 5; do
  steps=200; [ $cfg = 2 ] && steps=1000
  PM_PROF_MAIN="--workload config$cfg --steps $steps" PM_PROF_FLAGS="--workload config$cfg --steps $steps" bash tools/prof_round.sh ${TAG}_cfg$cfg 2>&1 | grep -v '^ \|^{\|^}' | tail -2
done
# the held-out workloads (piet_metal_amd/workloads.py, heldout_workloads): scenes no threshold was chosen on
for h in ${PM_PROF_HELD:-1 2 3}; do
  PM_PROF_MAIN="--workload held$h --steps 300" PM_PROF_FLAGS="--workload held$h --steps 300" bash tools/prof_round.sh ${TAG}_held$h 2>&1 | grep -v '^ \|^{\|^}' | tail -2
done
