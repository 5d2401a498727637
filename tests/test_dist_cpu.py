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
    # uneven, cost-balanced style cuts: the exchange must not depend on equal band sizes
    tiles_y = (height + 15) // 16
    cuts = pmd.balanced_cuts([b[0] for b in pmd.band_layout(height, world)] + [tiles_y], [1.0 + 3.0 * k for k in range(world)])
    layout = pmd.band_layout(height, world, cuts)
    r0, r1, rows = layout[rank]
    P = pmo.Ptcl(scene, width, height)
    mine = torch.from_numpy(P.render_rows(r0, r1))
    if rank == 0:  # the root renders straight into its rows of the final image
        full = torch.zeros((height, width, 4), dtype=torch.uint8)
        band = full[r0 * 16 : r0 * 16 + rows]
        band.copy_(mine)
    else:
        full, band = None, mine
    got = pmd.gather_bands(band, layout, height, dst=0, full=full)
    # the all-gather alternative must assemble the same image on every rank
    pad = torch.zeros((pmd.padded_band_rows(height, world, cuts), width, 4), dtype=torch.uint8)
    pad[:rows] = mine
    ag = pmd.allgather_bands(pad, layout, height)
    if rank == 0:
        assert got is full and torch.equal(ag, full)
        np.save(out_path, full.numpy())
    else:
        assert got is None
        np.save(out_path + f".{rank}.npy", ag.numpy())
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
    for k in range(1, world):
        assert np.array_equal(np.load(out + f".{k}.npy"), want)


def test_band_layout_covers_viewport():
    sys.path.insert(0, ROOT)
    from piet_metal_amd import dist as pmd

    for height in (16, 17, 250, 1080, 2160, 8192):
        for world in (1, 2, 4, 8):
            lay = pmd.band_layout(height, world)
            assert lay[0][0] == 0 and lay[-1][1] == (height + 15) // 16
            assert all(a[1] == b[0] for a, b in zip(lay, lay[1:]))
            assert sum(p for _, _, p in lay) == height


def test_balanced_cuts_converge_and_stay_valid():
    sys.path.insert(0, ROOT)
    from piet_metal_amd import dist as pmd

    true = np.interp(np.arange(135) + 0.5, [0, 30, 60, 75, 100, 135], [0.2, 1, 5, 6, 2, 0.2])  # Tiger-like: busy in the middle
    for world in (2, 4, 8):
        cuts = [b[0] for b in pmd.band_layout(2160, world)] + [135]
        for _ in range(3):
            ms = [float(true[a:b].sum()) for a, b in zip(cuts[:-1], cuts[1:])]
            cuts = pmd.balanced_cuts(cuts, ms)
            assert cuts[0] == 0 and cuts[-1] == 135 and all(b > a for a, b in zip(cuts, cuts[1:]))
        ms = [float(true[a:b].sum()) for a, b in zip(cuts[:-1], cuts[1:])]
        assert max(ms) / (sum(ms) / world) < 1.15
    # degenerate: more ranks than rows keeps the split untouched; zero times do not divide by zero
    assert pmd.balanced_cuts([0, 1, 2], [0.0, 0.0]) == [0, 1, 2]
    assert pmd.balanced_cuts([0, 2, 4, 6], [0.0, 0.0, 0.0])[-1] == 6


@pytest.mark.parametrize("world,root", [(2, 0), (3, 1)])
def test_c_abi_gather_across_processes_with_a_mock_rccl(built, tmp_path, world, root):
    """pm_comm_create / pm_gather -- the C-ABI collective SCALE runs would time -- across real processes on
    a GPU-less box: the library is the CPU emulation of tests/emu ("device" pointers are host pointers),
    RCCL is tests/mock_rccl bound through PM_RCCL_LIB (messages travel as files).  Uneven bands; the
    root's band copied into place (last frame) and already in place (pm_render_to into its rows of the
    image); a band table that does not match is refused.  What it cannot show is RCCL itself: the first
    real N > 1 run is the driver's."""
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU box: test_c_abi_gather_single_rank runs the real library against the real RCCL")
    from oracle import pmo

    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "emu")])
    shim = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", shim, os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.c")])
    box = tmp_path / "box"
    box.mkdir()
    width, height = 400, 300
    env = dict(os.environ, PM_RCCL_LIB=shim, PM_MOCK_RCCL_DIR=str(box), PM_WARMUP="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "cabi_gather_worker.py"), str(k), str(world), str(width), str(height), str(box), str(root)],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for k in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for k, p in enumerate(procs):
        assert p.returncode == 0, f"rank {k}: {outs[k][-2000:]}"
    want = pmo.render(pmo.scene_cardioid(), width, height)
    assert np.array_equal(np.load(box / "full_a.npy"), want)
    assert np.array_equal(np.load(box / "full_b.npy"), want)
    assert np.array_equal(np.load(box / "full_c.npy"), want), "four sub-bands per rank, one communicator (bench.py --gather-chunks)"
    assert not [f for f in os.listdir(box) if f.startswith("msg_")], "every message was consumed"


@pytest.mark.parametrize("fail,requested", [("all", "auto"), ("1", "auto"), ("", "auto"), ("all", "cabi")])
def test_a_job_agrees_on_one_gather_when_rccl_cannot_make_a_communicator(built, tmp_path, fail, requested):
    """bench.py --gpus N must print a complete line whatever the node's RCCL does (the first real N > 1 run is the driver's): the ranks
    exchange rank 0's id, and THEN ncclCommInitRank fails -- everywhere, or on one rank of three -- in the mock RCCL
    (PM_MOCK_RCCL_FAIL_INIT).  Every rank ends with the same answer (piet_metal_amd.dist.agree_on_c_abi_gather, the function bench.py
    calls): "sendrecv" and the reason with --gather-impl auto, an error that names the cause with --gather-impl cabi, "cabi" and a
    gathered frame equal to the oracle's when nothing fails.  Three processes over gloo, the CPU emulation as the library."""
    import json
    import socket
    import subprocess

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU box: the real RCCL is bound there")
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "emu")])
    shim = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", shim, os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.c")])
    box = tmp_path / "box"
    box.mkdir()
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    world = 3
    env = dict(os.environ, PM_RCCL_LIB=shim, PM_MOCK_RCCL_DIR=str(box), PM_WARMUP="0", PM_MOCK_RCCL_FAIL_INIT=fail)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "cabi_agree_worker.py"), str(k), str(world), port, requested],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    lines = []
    for k, p in enumerate(procs):
        assert p.returncode == 0, f"rank {k}: {outs[k][1][-2000:]}"
        lines.append(json.loads(outs[k][0].strip().splitlines()[-1]))  # a valid JSON line from every rank
    if requested == "cabi":
        assert all("error" in ln and "pm_comm_create" in ln["error"] for ln in lines), lines
        return
    impls = {ln["config"]["gather_impl"] for ln in lines}
    assert impls == ({"cabi"} if fail == "" else {"sendrecv"}), lines
    if fail == "":
        assert all(ln["gathered_frame_equals_oracle"] for ln in lines) and not any("gather_fallback" in ln["config"] for ln in lines)
    else:
        assert all(ln["config"]["gather_fallback"] for ln in lines), lines
        failed = [ln for ln in lines if "ncclCommInitRank" in ln["config"]["gather_fallback"]]
        assert len(failed) == (world if fail == "all" else 1), lines  # the ranks whose communicator failed say so; the others name their own
        # failed exchange (a peer that never posted its send) or "another rank"
    assert all(ln["config"]["t_gather_wire_floor_ms"] > 0 for ln in lines)
