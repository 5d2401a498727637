#!/bin/bash
# Developer loop on the GPU box: parity of a few scenes, the lone-frame bench digest, the two timelines -> gpurun_out/quick_<tag>.log
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-x}; shift
{ echo "== parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "${PM_QUICK_K:-baseline_configs or reference_scenes or random_scenes or wave_per_strip_row or strict_barrier}" 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== bench"; timeout 600 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 2>/dev/null | python -c '
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]
print("value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", r.get("kernels_alone_ms"))'
  for w in "$@"; do
    echo "== bench $w"; timeout 600 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 --workload $w 2>/dev/null | python -c '
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]
print("value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", r.get("kernels_alone_ms"))'
  done
  echo "== frame timeline"; timeout 100 python tools/frame_timeline.py 2>/dev/null
  echo "== bin timeline"; timeout 200 python tools/bin_timeline.py 2>&1 | grep -v amdgpu.ids
  echo "== tile timeline"; timeout 200 python tools/tile_timeline.py 2>&1 | grep -v "amdgpu.ids\|^  slot [0-9]* tile"
} > gpurun_out/quick_$TAG.log 2>&1
head -60 gpurun_out/quick_$TAG.log
