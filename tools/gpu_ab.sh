#!/bin/bash
# Developer A/B of two library builds on ONE box, alternating (boxes differ by +-0.5 us, runs on one box by +-0.15):
#   tools/gpu_ab.sh <variant> <rounds> [bench flags]
# <variant> = a second library piet_metal_amd/lib/libpiet_metal_amd_<variant>.so built by hand from modified sources (the
# objects of `hipcc -c` with the Makefile's flags, linked like the product) and allowed in piet_metal_amd/_lib.py's
# PM_LIB_VARIANT check for the session -- never committed, never loaded by the product.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
V=$1; R=$2; shift; shift
for i in $(seq 1 $R); do for v in "" $V; do
PM_LIB_VARIANT=$v timeout 300 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[%-4s]' % '$v', 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'])"
done; done
