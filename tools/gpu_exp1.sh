#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "=== base config2 timeline"; PM_TL_WORKLOAD=config2 python tools/one_launch_timeline.py
echo "=== wg4 config2 timeline"; PM_LIB_DEV=1 PM_LIB_VARIANT=wg4 PM_FRAME_WG_PER_CU=4 PM_TL_WORKLOAD=config2 python tools/one_launch_timeline.py
echo "=== split config2 timeline (phase 2 only)"; PM_ONE_LAUNCH_SPLIT=1 PM_TL_WORKLOAD=config2 python tools/one_launch_timeline.py
echo "=== split config3 timeline (phase 2 only)"; PM_ONE_LAUNCH_SPLIT=1 PM_TL_WORKLOAD=config3 python tools/one_launch_timeline.py
echo "=== binwt A/B config3"; PM_LIB_DEV=1 PM_LIB_VARIANT=binwt PM_AB_ROUNDS=1 python tools/one_launch_ab.py config3
echo "=== wg4 A/B config2"; PM_LIB_DEV=1 PM_LIB_VARIANT=wg4 PM_FRAME_WG_PER_CU=4 PM_AB_ROUNDS=1 python tools/one_launch_ab.py config2
echo "=== split A/B"; PM_ONE_LAUNCH_SPLIT=1 PM_AB_ROUNDS=1 python tools/one_launch_ab.py config3 config2
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp1.log
cat gpurun_out/exp1.log
