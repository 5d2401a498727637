#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "== tests (one launch for lone frames)"; PM_ONE_LAUNCH=1 timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "=== A/B"; PM_AB_ROUNDS=2 python tools/one_launch_ab.py config3 config2 tiger1440
echo "=== fuzz one-launch"; PM_ONE_LAUNCH=1 timeout 300 python tests/dev/fuzz_parity.py 51000 150 --ext 2>&1 | tail -2
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp3.log
cat gpurun_out/exp3.log
