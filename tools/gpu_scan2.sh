#!/bin/bash
# tools/gpu_scan2.sh "VAR=v1 VAR2=w1" "VAR=v2" ... : one bench summary per environment
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for e in "$@"; do
echo -n "[$e]  "
env $e timeout 600 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 ${PM_SCAN_FLAGS:-} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
} > gpurun_out/scan.log 2>&1
cat gpurun_out/scan.log
