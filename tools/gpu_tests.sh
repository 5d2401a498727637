#!/bin/bash
# full -m gpu suite + default bench line: tools/gpu_tests.sh [tag]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${1:-tests}
{
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/${TAG}_bench.err | tee gpurun_out/${TAG}_bench.json | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'], 'frac_serial', j['roofline']['frac_serial_frame'], 'rccl', j['config'].get('rccl_lib'), j['config'].get('rccl_ranks'), 'cfg5', j['config5']['value'], j['config5']['t_frame_ms'], j['config5']['first_frame_ms'])"
tail -3 gpurun_out/${TAG}_bench.err
} > gpurun_out/$TAG.log 2>&1
tail -40 gpurun_out/$TAG.log
