#!/bin/bash
# quick GPU check: parity subset + fuzz + lone-frame timeline + bench summary
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${PM_K:-reference_scenes or random_scenes or longer_than or many_items or baseline_configs or both_fine or even_odd}" 2>&1 | tail -3
timeout 600 python tests/dev/fuzz_parity.py ${PM_SEED:-12000} ${PM_FUZZ:-150} --ext 2>&1 | tail -1
timeout 100 python tools/frame_timeline.py 2>/dev/null
timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
