#!/bin/bash
# ONE runner for the developer's GPU sessions (each a `gpurun -- 'bash tools/gpu.sh <task> [args]'`); output -> gpurun_out/<task>[_<tag>].log.
#   tests [k-expr]        the -m gpu suite (optionally -k <expr>), smoke(), the driver's bench command
#   fuzz <n> [seed]       n random scenes x {strict, product} builds x {plain, extensions, a wave per strip row, one launch} + flatten
#   bench [flags...]      bench.py with the given flags, a one-line digest of its JSON
#   ab <variant> <rounds> [flags]   two library builds alternating on this box (tools/build_variant.sh makes the second)
#   timelines [cfg...]    the lone frame taken apart: per strip row, per tile, per kernel (config3 config2 ...)
#   one [cfg...]          one launch per frame against two (tools/one_launch_ab.py) + its per-workgroup timeline
#   policy [--all]        every switch of pm_create on the held-out workloads (tools/heldout_policy.py)
#   profile <tag>         tools/prof_all_configs.sh <tag> (+ the held-out workloads): the round's committed evidence
#   py <script> [args]    any script of tools/ or tests/dev/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
TASK=${1:-tests}; shift
digest='import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]
print("value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "ms_per_step", j["ms_per_step"], "alone", r.get("kernels_alone_ms"), "frac", r.get("frac"), "frac_frame", r.get("frac_frame"), "cfg5", (j.get("config5") or {}).get("value"))'
case $TASK in
tests)
  { timeout 1800 python -m pytest tests -x -q -m gpu ${1:+-k "$1"} 2>&1 | grep -v amdgpu.ids | tail -15
    echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
    echo "== bench (driver flags)"; timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>gpurun_out/tests_bench.err | tee gpurun_out/tests_bench.json | python -c "$digest"
  } > gpurun_out/tests.log 2>&1; tail -40 gpurun_out/tests.log ;;
fuzz)
  N=${1:-500}; S=${2:-610000}
  { echo "== strict build, plain"; PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py $S $N 2>&1 | tail -1
    echo "== strict build, extensions, a wave per strip row"; PM_BIN_WAVES=1 PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py $((S+10000)) $N --ext 2>&1 | tail -1
    echo "== strict build, one launch per frame, extensions"; PM_ONE_LAUNCH=1 PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py $((S+20000)) $N --ext 2>&1 | tail -1
    echo "== product build, plain"; timeout 3000 python tests/dev/fuzz_parity.py $((S+30000)) $N 2>&1 | tail -1
    echo "== product build, extensions"; timeout 3000 python tests/dev/fuzz_parity.py $((S+40000)) $N --ext 2>&1 | tail -1
    echo "== product build, one launch per frame, plain"; PM_ONE_LAUNCH=1 timeout 3000 python tests/dev/fuzz_parity.py $((S+50000)) $N 2>&1 | tail -1
    echo "== product build, one launch per frame on a quarter grid (chains of strip rows), extensions"; PM_ONE_LAUNCH=1 PM_FRAME_WG_PER_CU=1 timeout 3000 python tests/dev/fuzz_parity.py $((S+60000)) $N --ext 2>&1 | tail -1
    echo "== product build, a wave per strip row, one workgroup per CU"; PM_BIN_WAVES=1 PM_BIN_WG_PER_CU=1 timeout 3000 python tests/dev/fuzz_parity.py $((S+70000)) $N --ext 2>&1 | tail -1
    echo "== product build, row lists in parts of 13 items, every frame behind a scene's first on the one-wave tile kernel, extensions"; PM_ROW_LIST_MIN_ITEMS=1 PM_ROW_LIST_PART_ITEMS=13 PM_DENSE_FACTOR=100000 timeout 3000 python tests/dev/fuzz_parity.py $((S+90000)) $N --ext 2>&1 | tail -1
    echo "== product build, every strip row cut in two, clearing inside the binning launch, extensions"; PM_BIN_SPLIT=2 PM_FOLD_CLEAR=4 timeout 3000 python tests/dev/fuzz_parity.py $((S+110000)) $N --ext 2>&1 | tail -1
    echo "== strict build, every strip row cut in two, a wave per strip row, clearing inside the binning launch"; PM_BIN_SPLIT=2 PM_BIN_WAVES=1 PM_FOLD_CLEAR=4 PM_LIB_VARIANT=strict timeout 3000 python tests/dev/fuzz_parity.py $((S+120000)) $N 2>&1 | tail -1
    echo "== product build, five frames per scene, strip rows of 24 slots and more cut by the frames' own report"; PM_FUZZ_FRAMES=5 PM_BIN_SPLIT_SLOTS=24 timeout 3000 python tests/dev/fuzz_parity.py $((S+130000)) $N --ext 2>&1 | tail -1
    echo "== product build, flatten, block-parallel sums above 64 elements"; PM_SCAN_SPLIT=64 timeout 1200 python tests/dev/fuzz_flatten.py $((S+100000)) 300 2>&1 | tail -1
    echo "== product build, flatten"; timeout 1200 python tests/dev/fuzz_flatten.py $((S+80000)) 300 2>&1 | tail -1
  } > gpurun_out/fuzz.log 2>&1; cat gpurun_out/fuzz.log ;;
bench)
  timeout 1500 python bench.py "$@" 2>gpurun_out/bench.err | tee gpurun_out/bench.json | python -c "$digest"; tail -3 gpurun_out/bench.err ;;
ab)
  V=$1; R=$2; shift; shift
  for i in $(seq 1 $R); do for v in "" $V; do
    PM_LIB_DEV=1 PM_LIB_VARIANT=$v timeout 300 python bench.py --steps 300 --warmup 40 --no-cpu-baseline --no-config5 "$@" 2>/dev/null | python -c "$digest" | sed "s/^/[${v:-base}] /"
  done; done | tee gpurun_out/ab.log ;;
timelines)
  { for cfg in ${@:-config3}; do
      echo "== frame timeline"; timeout 100 python tools/frame_timeline.py 2>/dev/null
      echo "== bin timeline $cfg"; PM_TL_WORKLOAD=$cfg timeout 200 python tools/bin_timeline.py 2>&1 | grep -v amdgpu.ids
      echo "== tile timeline $cfg"; PM_TL_WORKLOAD=$cfg timeout 200 python tools/tile_timeline.py 2>&1 | grep -v "amdgpu.ids\|^  slot [0-9]* tile"
    done; } > gpurun_out/timelines.log 2>&1; tail -150 gpurun_out/timelines.log ;;
one)
  { timeout 600 python tools/one_launch_ab.py ${@:-config3 config2} 2>&1 | grep -v amdgpu.ids
    for cfg in ${@:-config3 config2}; do echo "== one-launch timeline $cfg"; PM_ONE_LAUNCH=1 PM_TL_WORKLOAD=$cfg timeout 200 python tools/one_launch_timeline.py 2>&1 | grep -v amdgpu.ids; done
  } > gpurun_out/one.log 2>&1; cat gpurun_out/one.log ;;
policy)
  timeout 2400 python tools/heldout_policy.py "$@" gpurun_out/held_policy.json 2>&1 | grep -v amdgpu.ids | tee gpurun_out/policy.log ;;
profile)
  bash tools/prof_all_configs.sh ${1:-r05_a} ;;
py)
  timeout 2400 python "$@" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/py.log ;;
*) echo "unknown task $TASK"; exit 2 ;;
esac
