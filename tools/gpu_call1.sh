#!/bin/bash
# GPU call 1 (round 2): parity of the sparse fine kernel, A/B bench, timelines
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c1
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== quick parity (both fine kernels) =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "both_fine or reference_scenes or random_scenes or longer_than or many_items" 2>&1 | tail -5 | tee $OUT/quick.log
echo "== fuzz 150 =="
timeout 600 python tests/dev/fuzz_parity.py 5000 150 2>&1 | tail -5 | tee $OUT/fuzz.log
echo "== bench A/B =="
PM_FINE_SPARSE=0 timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 > $OUT/bench_dense.json 2> $OUT/bench_dense.err; tail -c 300 $OUT/bench_dense.err
PM_FINE_SPARSE=1 timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 > $OUT/bench_sparse.json 2> $OUT/bench_sparse.err; tail -c 300 $OUT/bench_sparse.err
python - <<'PY'
import json,os
for n in ("dense","sparse"):
    try:
        j=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),"gpurun_out/c1/bench_%s.json"%n)).read().strip().splitlines()[-1])
        print(n, "value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", j["roofline"]["kernels_alone_ms"], "inflight", j["roofline"]["kernels_ms"], "scene", j["scene"])
    except Exception as e: print(n, "ERR", e)
PY
echo "== timelines (sparse) =="
timeout 300 python tools/tile_timeline.py 2>&1 | tee $OUT/tile_timeline.log
timeout 300 python tools/bin_timeline.py 2>&1 | tee $OUT/bin_timeline.log
echo "== full gpu suite =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/suite.log
echo "== config times =="
timeout 600 python tools/config_times.py 2>&1 | tee $OUT/config_times.log
