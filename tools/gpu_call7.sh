#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/c7
echo "== full gpu suite =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/c7/suite.log
echo "== fuzz =="
timeout 600 python tests/dev/fuzz_parity.py 9000 300 2>&1 | tail -3 | tee gpurun_out/c7/fuzz.log
echo "== profile round =="
bash tools/prof_round.sh r02_a 2>&1 | tail -60
