#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/c10
echo "== new tests =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "second_svg or animation or cli" 2>&1 | tail -25 | tee gpurun_out/c10/new.log
echo "== full gpu suite =="
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/c10/suite.log
echo "== animation rate at 4K =="
timeout 300 python -m piet_metal_amd.cli tiger /tmp/spin.png --width 3840 --height 2160 --frames 24 2>&1 | tail -2
python - <<'PY'
import time, numpy as np, piet_metal_amd as pm
from piet_metal_amd import cli
wl = pm.workloads.tiger(3840, 2160)
r = pm.Renderer(0); r.resize(wl.width, wl.height); r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale); r.render(); r.sync()
for name, fn in (("reflatten (resident paths)", lambda a: r.reflatten(a, wl.width_scale)), ("flatten_and_encode (upload)", lambda a: r.flatten_and_encode(wl.paths, a, wl.width_scale))):
    t0 = time.perf_counter(); n = 100
    for k in range(n):
        fn(cli.spin_affine(wl.affine, 0.01 * k, 1920, 1080)); r.render()
    r.sync(); dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: {dt:.3f} ms per re-encoded 4K frame ({1e3/dt:.0f} fps)", r.scene_timings())
PY
