#!/bin/bash
# A/B of an environment switch: tools/gpu_ab.sh VAR v1 v2 ...  (quick parity subset + bench summary per value)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
VAR=$1; shift
{
for v in "$@"; do
echo "== $VAR=$v"
env $VAR=$v timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_scenes or random_scenes or baseline_configs or flight or both_fine" 2>&1 | tail -1
env $VAR=$v timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
} > gpurun_out/ab.log 2>&1
cat gpurun_out/ab.log
