"""world_size-2 `gloo` test of the multi-GPU path's host logic (band split +
the one framebuffer gather).  On a CPU box the bands come from the oracle's
per-band render (test infrastructure standing in for the HIP kernels); the code
under test is piet_metal_amd/dist.py, identical for gloo and nccl(RCCL)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, width, height, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch

    from oracle import pmo
    from piet_metal_amd import dist as pmd

    r, w, _ = pmd.init_process_group("gloo")
    assert (r, w) == (rank, world)
    scene = pmo.scene_cardioid()
    layout = pmd.band_layout(height, world)
    r0, r1, rows = layout[rank]
    pad = pmd.padded_band_rows(height, world)
    band = torch.zeros((pad, width, 4), dtype=torch.uint8)
    P = pmo.Ptcl(scene, width, height)
    band[:rows] = torch.from_numpy(P.render_rows(r0, r1))
    full = pmd.gather_framebuffer(band, height, dst=0)
    if rank == 0:
        np.save(out_path, full.numpy())
    else:
        assert full is None
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,height", [(2, 1000), (3, 250)])
def test_band_gather_gloo(tmp_path, pmo, world, height):
    import torch.multiprocessing as mp

    width = 1300
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), width, height, out), nprocs=world, join=True)
    full = np.load(out)
    want = pmo.render(pmo.scene_cardioid(), width, height)
    assert full.shape == want.shape and np.array_equal(full, want)


def test_band_layout_covers_viewport():
    sys.path.insert(0, ROOT)
    from piet_metal_amd import dist as pmd

    for height in (16, 17, 250, 1080, 2160, 8192):
        for world in (1, 2, 4, 8):
            lay = pmd.band_layout(height, world)
            assert lay[0][0] == 0 and lay[-1][1] == (height + 15) // 16
            assert all(a[1] == b[0] for a, b in zip(lay, lay[1:]))
            assert sum(p for _, _, p in lay) == height
