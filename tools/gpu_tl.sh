#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
{ for cfg in ${PM_TL_CFGS:-config3}; do PM_TL_WORKLOAD=$cfg timeout 200 python tools/bin_timeline.py 2>&1 | grep -v amdgpu.ids; done; } > gpurun_out/tl.log 2>&1; cat gpurun_out/tl.log
