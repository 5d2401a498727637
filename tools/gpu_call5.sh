#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c6
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
echo "== quick parity =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "both_fine or pipelined or overflow or reference_scenes or random_scenes or longer_than or many_items or baseline_configs" 2>&1 | tail -3 | tee $OUT/quick.log
echo "== fuzz =="
timeout 600 python tests/dev/fuzz_parity.py 8000 200 2>&1 | tail -3 | tee $OUT/fuzz.log
for H in 32 48 24; do
echo "== bench PM_HEAVY_STREAM=$H =="
PM_HEAVY_STREAM=$H timeout 600 python bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-config5 > $OUT/bench_$H.json 2> $OUT/bench_$H.err; tail -c 300 $OUT/bench_$H.err
python - $OUT/bench_$H.json <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", j["value"], "t_frame", j["t_frame_ms"], "sustained", j["sustained_mpix_s"], "alone", j["roofline"]["kernels_alone_ms"], "inflight", j["roofline"]["kernels_ms"])
except Exception as e: print("ERR", e)
PY
done
echo "== timelines =="
timeout 300 python tools/tile_timeline.py 2>&1 | tail -22 | tee $OUT/tile_timeline.log
timeout 300 python tools/config_times.py 2>&1 | tee $OUT/config_times.log
