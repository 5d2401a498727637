#!/bin/bash
# waves per strip row for frames behind other frames (PM_BIN_WAVES_INFLIGHT=1 / 4): sustained rates
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
for bw in 1 4; do for cfg in config3 config2; do
PM_BIN_WAVES_INFLIGHT=$bw timeout 600 python bench.py --workload $cfg --steps 1000 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight bw $bw $cfg value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'])"
PM_BIN_WAVES_INFLIGHT=$bw timeout 600 python bench.py --workload $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   20 steps: value', j['value'], 'sustained', j['sustained_mpix_s'])"
done; done
