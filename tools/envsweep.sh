#!/bin/bash
# Developer experiment: the lone-frame digest under a list of environment settings (one per argument, "A=1 B=2" each) -> gpurun_out/envsweep.log
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
W=${PM_SWEEP_WORKLOAD:-config3}
for e in "$@"; do
  for rep in 1 2; do
  env $e timeout 600 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 --workload $W 2>/dev/null | python -c '
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]
print(sys.argv[1], "| value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", r.get("kernels_alone_ms"))' "$e"
  done
done 2>&1 | tee gpurun_out/envsweep.log
