#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "== tests"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "== held-out policy"; timeout 900 python tools/heldout_policy.py gpurun_out/held_policy.json
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp5.log
cat gpurun_out/exp5.log
