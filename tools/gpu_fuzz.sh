#!/bin/bash
# fuzz both library builds against the oracle: tools/gpu_fuzz.sh <tag> <count>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
TAG=${1:-fuzz}; N=${2:-1000}
{
echo "== strict build (every LdsBarrier a __syncthreads), plain scenes"; PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py 410000 $N 2>&1 | tail -2
echo "== strict build, extensions"; PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py 420000 $N --ext 2>&1 | tail -2
echo "== product build, plain scenes"; timeout 3000 python tests/dev/fuzz_parity.py 430000 $N 2>&1 | tail -2
echo "== product build, extensions"; timeout 3000 python tests/dev/fuzz_parity.py 440000 $N --ext 2>&1 | tail -2
echo "== product build, flatten"; timeout 1200 python tests/dev/fuzz_flatten.py 450000 300 2>&1 | tail -2
} > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
