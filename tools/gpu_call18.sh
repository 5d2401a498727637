#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for w in 4 5; do
for cfg in config3 config4 config5 config2; do
echo "== PM_FINE_WG_PER_CU=$w $cfg"
PM_FINE_WG_PER_CU=$w timeout 600 python bench.py --workload $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
done
} > gpurun_out/call18.log 2>&1
cat gpurun_out/call18.log
