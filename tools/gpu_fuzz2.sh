#!/bin/bash
# fuzz of the final kernels, both builds and both binning modes: tools/gpu_fuzz2.sh <tag> <count>
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
TAG=${1:-fuzz2}; N=${2:-500}
{
echo "== strict build (every LdsBarrier a __syncthreads), plain scenes"; PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py 510000 $N 2>&1 | tail -1
echo "== strict build, extensions, a wave per strip row (PM_BIN_WAVES=1)"; PM_BIN_WAVES=1 PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py 520000 $N --ext 2>&1 | tail -1
echo "== product build, plain scenes"; timeout 3000 python tests/dev/fuzz_parity.py 530000 $N 2>&1 | tail -1
echo "== product build, extensions"; timeout 3000 python tests/dev/fuzz_parity.py 540000 $N --ext 2>&1 | tail -1
echo "== product build, plain scenes, a wave per strip row (PM_BIN_WAVES=1)"; PM_BIN_WAVES=1 timeout 3000 python tests/dev/fuzz_parity.py 550000 $N 2>&1 | tail -1
echo "== product build, extensions, a wave per strip row, one workgroup per CU (chains of rows)"; PM_BIN_WAVES=1 PM_BIN_WG_PER_CU=1 timeout 3000 python tests/dev/fuzz_parity.py 560000 $N --ext 2>&1 | tail -1
echo "== product build, flatten"; timeout 1200 python tests/dev/fuzz_flatten.py 570000 300 2>&1 | tail -1
} > gpurun_out/$TAG.log 2>&1
cat gpurun_out/$TAG.log
