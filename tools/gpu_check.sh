#!/bin/bash
# quick GPU check + tile timeline (developer loop)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
PM_FUZZ=${PM_FUZZ:-100} bash tools/gpu_quick.sh
timeout 120 python tools/tile_timeline.py | grep -v "^  slot [0-9]* tile\|^   "
} > gpurun_out/check.log 2>&1
tail -30 gpurun_out/check.log
