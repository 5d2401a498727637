#!/bin/bash
# GPU session of the one-launch work: parity suite, then the A/B of one launch against two.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-one}
{
echo "== tests"; timeout ${PM_TEST_TIMEOUT:-900} python -m pytest tests -x -q -m gpu ${PM_K:+-k "$PM_K"} 2>&1 | grep -E "passed|failed|error|Error|assert|Fatal|fault" | tail -12
echo "== A/B"; timeout 600 python tools/one_launch_ab.py ${PM_AB_CASES:-config3 config2} 2>&1 | grep -v amdgpu.ids
} > gpurun_out/$TAG.log 2>&1
tail -60 gpurun_out/$TAG.log
