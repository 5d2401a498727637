#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "== tests"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "== policy, all workloads"; timeout 1500 python tools/heldout_policy.py --all gpurun_out/held_policy_b.json
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp6.log
cat gpurun_out/exp6.log
