#!/bin/bash
# Profile rounds of all bench workloads: tools/prof_round.sh for the Tiger (r02_c) and configs 2, 4, 5 (r02c_cfgN)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
bash tools/prof_round.sh r02_c 2>&1 | grep -v '^ \|^{\|^}' | tail -3
for cfg in 2 4 5; do
  steps=200; [ $cfg = 2 ] && steps=1000
  PM_PROF_MAIN="--workload config$cfg --steps $steps" PM_PROF_FLAGS="--workload config$cfg --steps $steps" bash tools/prof_round.sh r02c_cfg$cfg 2>&1 | grep -v '^ \|^{\|^}' | tail -2
done
