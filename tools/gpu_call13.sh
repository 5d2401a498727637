#!/bin/bash
# fused coarse+fine A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "== fused =="; PM_FUSED=1 PM_FUZZ=100 bash tools/gpu_quick.sh
echo "== unfused =="; PM_FUSED=0 PM_FUZZ=50 bash tools/gpu_quick.sh
} > gpurun_out/call13.log 2>&1
tail -40 gpurun_out/call13.log
