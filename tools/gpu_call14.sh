#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
timeout 120 python tools/tile_timeline.py
} > gpurun_out/call14.log 2>&1
tail -40 gpurun_out/call14.log
