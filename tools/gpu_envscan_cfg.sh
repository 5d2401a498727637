#!/bin/bash
# scan of environment settings over several workloads: PM_SCAN_CFGS="config2 config5" tools/gpu_envscan_cfg.sh "A=1" "A=2" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for v in "$@"; do
for cfg in ${PM_SCAN_CFGS:-config3}; do
echo -n "[$v] $cfg  "
env $v timeout 600 python bench.py --workload $cfg --steps 200 --warmup 20 --no-cpu-baseline --no-config5 2>/dev/null >/tmp/scan_out.json
python -c "
import json,sys
j=json.loads(open('/tmp/scan_out.json').read().strip().splitlines()[-1]); print('t_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
done
} > gpurun_out/envscan_cfg.log 2>&1
cat gpurun_out/envscan_cfg.log
