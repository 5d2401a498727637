#!/bin/bash
# full GPU verification: pytest -m gpu, smoke(), fuzzers, default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python tests/dev/fuzz_parity.py ${PM_SEED:-31000} ${PM_FUZZ:-400} --ext 2>&1 | tail -1
timeout 600 python tests/dev/fuzz_parity.py ${PM_SEED2:-52000} ${PM_FUZZ2:-300} 2>&1 | tail -1
timeout 600 python tests/dev/fuzz_flatten.py 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
} > gpurun_out/full.log 2>&1
tail -30 gpurun_out/full.log
