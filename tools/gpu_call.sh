#!/bin/bash
# One GPU session of the developer loop: full -m gpu suite, timelines of the Tiger and the dense config, short bench.
#   gpurun --timeout 900 -- 'bash tools/gpu_call.sh <tag>'
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
TAG=${1:-call}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
[ "${PM_CALL_TESTS:-1}" = "1" ] && echo "== tests" && timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench config3"; timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'], 'scene', j['scene'])"
for cfg in ${PM_CALL_CFGS:-config4}; do
echo "== bench $cfg"; timeout 600 python bench.py --workload $cfg --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value', j['value'], 't_frame', j['t_frame_ms'], 'sustained', j['sustained_mpix_s'], 'alone', j['roofline']['kernels_alone_ms'], 'scene', j['scene'])"
done
echo "== frame timeline"; timeout 100 python tools/frame_timeline.py 2>/dev/null
if [ "${PM_CALL_TIMELINES:-1}" = "1" ]; then
for cfg in config3 config4; do
echo "== bin timeline $cfg"; PM_TL_WORKLOAD=$cfg timeout 200 python tools/bin_timeline.py 2>&1 | grep -v amdgpu.ids
echo "== tile timeline $cfg"; PM_TL_WORKLOAD=$cfg timeout 200 python tools/tile_timeline.py 2>&1 | grep -v "amdgpu.ids\|^  slot [0-9]* tile"
done
fi
} > gpurun_out/$TAG.log 2>&1
tail -150 gpurun_out/$TAG.log
