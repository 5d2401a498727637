#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for cfg in config5 config4; do PM_TL_WORKLOAD=$cfg timeout 200 python tools/chain_timeline.py 2>&1 | grep -v amdgpu.ids; done
for k in 1 2 3 5; do PM_BIN_WG_PER_CU=$k PM_TL_WORKLOAD=config3 timeout 200 python tools/chain_timeline.py 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r4b.log 2>&1
cat gpurun_out/r4b.log
