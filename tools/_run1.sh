timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 1000 --warmup 30 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernels_ms'], d['roofline']['kernels_alone_ms'])"; done
