#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "== tests"; timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert|Fatal|fault" | tail -12
echo "=== config3 timeline"; python tools/one_launch_timeline.py
echo "=== config2 timeline"; PM_TL_WORKLOAD=config2 python tools/one_launch_timeline.py
echo "=== A/B"; PM_AB_ROUNDS=2 python tools/one_launch_ab.py config3 config2 tiger1440
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp2.log
cat gpurun_out/exp2.log
