#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c3
mkdir -p $OUT /tmp/pb
cd $ROOT
export TMPDIR=/tmp
echo "== issue probe =="
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/pb/issue_probe tools/probes/issue_probe.hip 2>/dev/null
timeout 300 /tmp/pb/issue_probe 2>&1 | tee $OUT/issue_probe.log
