run() { echo "== HWQ=$1 B=$2 C=$3 F=$4 S=$5"; GPU_MAX_HW_QUEUES=$1 PM_BIN_STREAMS=$2 PM_COARSE_STREAMS=$3 PM_FINE_STREAMS=$4 PM_SLOTS=$5 timeout 300 python bench.py --steps 2000 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
run 4 2 1 1 4
run 4 2 1 1 5
run 4 2 1 1 6
run 4 2 1 1 8
run 4 3 1 1 5
run 4 3 1 1 8
run 8 3 1 1 8
run 4 2 2 1 6
run 4 1 1 1 2
run 4 1 1 1 4
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
