#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c2
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== quick parity =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "both_fine or reference_scenes or random_scenes or longer_than or many_items or baseline_configs" 2>&1 | tail -5 | tee $OUT/quick.log
echo "== fuzz =="
timeout 600 python tests/dev/fuzz_parity.py 6000 300 2>&1 | tail -5 | tee $OUT/fuzz.log
echo "== bench =="
timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 > $OUT/bench_sparse.json 2> $OUT/bench_sparse.err; tail -c 300 $OUT/bench_sparse.err
python - <<'PY'
import json,os
for n in ("sparse",):
    try:
        j=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/c2/bench_%s.json"%n)).read().strip().splitlines()[-1])
        print(n, "value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", j["roofline"]["kernels_alone_ms"], "inflight", j["roofline"]["kernels_ms"])
    except Exception as e: print(n, "ERR", e)
PY
echo "== timelines =="
timeout 300 python tools/tile_timeline.py 2>&1 | tee $OUT/tile_timeline.log
echo "== config times =="
timeout 600 python tools/config_times.py 2>&1 | tee $OUT/config_times.log
