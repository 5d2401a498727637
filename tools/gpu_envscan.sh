#!/bin/bash
# scan of environment settings over the default bench: tools/gpu_envscan.sh "A=1 B=2" "A=2" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
for v in "$@"; do
echo -n "[$v]  "
env $v timeout 600 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 ${PM_SCAN_FLAGS:-} 2>&1 >/tmp/scan_out.json | grep "^pm:" | tail -1 | tr '\n' ' '
python -c "
import json,sys
j=json.loads(open('/tmp/scan_out.json').read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
} > gpurun_out/envscan.log 2>&1
cat gpurun_out/envscan.log
