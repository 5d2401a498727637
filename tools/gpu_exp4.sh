#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; export TMPDIR=/tmp; mkdir -p gpurun_out
{
echo "=== tile timeline config3"; PM_TL_WORKLOAD=config3 timeout 200 python tools/tile_timeline.py 2>&1 | grep -v "^  slot [0-9]* tile" | tail -60
echo "=== bin timeline config3"; PM_TL_WORKLOAD=config3 timeout 200 python tools/bin_timeline.py 2>&1 | tail -30
echo "=== batch start/stop cost"; python - <<'PY'
import time, sys, os
sys.path.insert(0, os.getcwd())
import torch
import piet_metal_amd as pm
wl = pm.workloads.tiger(3840, 2160)
r = pm.Renderer(0); r.resize(wl.width, wl.height); r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
for _ in range(300): r.render()
r.sync(); torch.cuda.synchronize()
for K in (1, 5, 20, 100):
    ts = []
    for rep in range(30):
        r.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): r.render()
        t1 = time.perf_counter()
        r.sync()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
    ts.sort(key=lambda x: x[3])
    m = ts[len(ts) // 2]
    print("K=%3d submit %.1f us  pm_sync %.1f us  torch.sync %.1f us  total %.1f us (%.1f per step)" % (K, m[0]*1e6, m[1]*1e6, m[2]*1e6, m[3]*1e6, m[3]*1e6/K))
PY
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp4.log
cat gpurun_out/exp4.log
