#!/bin/bash
# waves per strip row in pm_bin_kernel (PM_BIN_WAVES=1 / 4): parity subset + the four configurations, both ways
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
PM_BIN_WAVES=1 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${PM_K:-reference_scenes or random_scenes or baseline_configs or per_row or bands or view_changes}" 2>&1 | tail -2
for bw in ${PM_BW_LIST:-1 4}; do
for cfg in ${PM_Q_CFGS:-config5 config4 config3 config2}; do
PM_BIN_WAVES=$bw timeout 600 python bench.py --workload $cfg --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bw $bw $cfg value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done
done
} > gpurun_out/bw.log 2>&1
cat gpurun_out/bw.log
