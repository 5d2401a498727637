#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/c9
echo "== new tests =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "coverage or full_size_goldens" 2>&1 | tail -15 | tee gpurun_out/c9/new.log
echo "== full gpu suite =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/c9/suite.log
echo "== bin sort A/B =="
timeout 100 python tools/frame_timeline.py
