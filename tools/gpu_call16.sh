#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for w in 4 5; do
echo "== PM_FINE_WG_PER_CU=$w"
PM_FINE_WG_PER_CU=$w timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
PM_FINE_WG_PER_CU=5 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reference_scenes or random_scenes or baseline_configs" 2>&1 | tail -2
} > gpurun_out/call16.log 2>&1
tail -40 gpurun_out/call16.log
