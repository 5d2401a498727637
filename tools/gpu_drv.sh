#!/bin/bash
# the driver's bench command a few times + overflow tests: tools/gpu_drv.sh <tag>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
TAG=${1:-drv}
{
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "overflow or pipelined or flight or depths" 2>&1 | tail -2
for i in 1 2 3; do
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'ms_per_step', j['ms_per_step'], 'cfg5', j['config5']['value'], j['config5']['sustained_mpix_s'], j['config5']['first_frame_ms'])"
done
} > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
