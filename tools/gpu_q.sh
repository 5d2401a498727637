#!/bin/bash
# quick GPU check: parity subset + bench summary (+ optional bin timeline): tools/gpu_q.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-q}
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${PM_K:-reference_scenes or random_scenes or baseline_configs or both_fine}" 2>&1 | tail -2
timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
if [ "${PM_Q_TL:-1}" = "1" ]; then PM_TL_WORKLOAD=config3 timeout 200 python tools/bin_timeline.py 2>&1 | grep -v amdgpu.ids | head -14; fi
for cfg in ${PM_Q_CFGS:-}; do
timeout 600 python bench.py --workload $cfg --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
} > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
