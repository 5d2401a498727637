#!/bin/bash
# round 4, call A: baseline of the round + instruction-cache probe
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== icache probe"; timeout 120 tools/probes/bin/icache_probe
echo "== bench config3"; timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'], 'scene', j['scene'])"
echo "== frame timeline"; timeout 100 python tools/frame_timeline.py 2>/dev/null
for cfg in config3; do
echo "== bin timeline $cfg"; PM_TL_WORKLOAD=$cfg timeout 200 python tools/bin_timeline.py 2>&1 | grep -v amdgpu.ids
echo "== tile timeline $cfg"; PM_TL_WORKLOAD=$cfg timeout 200 python tools/tile_timeline.py 2>&1 | grep -v "amdgpu.ids\|^  slot [0-9]* tile"
done
} > gpurun_out/r4a.log 2>&1
tail -120 gpurun_out/r4a.log
