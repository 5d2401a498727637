set -x
timeout 300 python tools/gpu_check.py > gpurun_out/check.log 2>&1; echo check rc=$?
tail -25 gpurun_out/check.log
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>&1 | tail -2
