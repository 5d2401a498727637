#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/c8
echo "== new tests =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "even_odd or nested or malformed" 2>&1 | tail -15 | tee gpurun_out/c8/new.log
echo "== full gpu suite =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/c8/suite.log
echo "== fuzz ext =="
timeout 600 python tests/dev/fuzz_parity.py 10000 300 --ext 2>&1 | tail -3 | tee gpurun_out/c8/fuzz_ext.log
timeout 600 python tests/dev/fuzz_parity.py 11000 200 2>&1 | tail -3 | tee gpurun_out/c8/fuzz.log
echo "== bench =="
timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], j['roofline']['kernels_alone_ms'], j['scene'])"
