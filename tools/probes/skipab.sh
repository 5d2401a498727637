#!/bin/bash
# Developer: cost attribution by leaving parts out (libraries from tools/build_variant.sh x<part> "-DPM_EXP_SKIP_<PART>"; pixels are wrong by design)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
digest='import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]
print("t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", r.get("kernels_alone_ms"))'
for wl in "$@"; do for v in ${PM_SKIP_VARIANTS:-"" u2 "" u2}; do
  PM_LIB_DEV=1 PM_LIB_VARIANT=${v#base} timeout 300 python bench.py --workload $wl --steps 200 --warmup 20 --no-cpu-baseline --no-config5 2>/dev/null | python -c "$digest" | sed "s/^/[$wl ${v:-base}] /"
done; done | tee gpurun_out/skipab.log
