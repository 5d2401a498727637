"""Worker process of tests/test_dist_cpu.py::test_a_job_agrees_on_one_gather_when_rccl_cannot_make_a_communicator: one rank of the
agreement bench.py --gpus N runs before its first step (piet_metal_amd.dist.agree_on_c_abi_gather) on a GPU-less box -- the library is
the CPU emulation of tests/emu, RCCL is tests/mock_rccl (PM_RCCL_LIB) with ncclCommInitRank failing where PM_MOCK_RCCL_FAIL_INIT says,
the ranks talk over gloo.  Prints the part of bench.py's JSON line the agreement decides."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, requested = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    from emu_swap import swap_in_emulated_library

    swap_in_emulated_library(build=False)
    import torch
    import torch.distributed as dist

    import piet_metal_amd as pm
    from cabi_gather_worker import HostBuf
    from oracle import pmo
    from piet_metal_amd import dist as pmd

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    width, height = 320, 240
    layout = pmd.band_layout(height, world)
    r0, r1, rows = layout[rank]
    r = pm.Renderer(0)
    r.resize(width, height)
    r.set_scene_bytes(pmo.scene_cardioid())
    r.set_band(r0, r1)
    full = np.zeros((height, width, 4), np.uint8) if rank == 0 else None

    def all_min(v):
        t = torch.tensor([v], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return int(t.item())

    def broadcast_id(make_id):
        box = [make_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def try_exchange(comm):
        r.render()
        comm.gather(layout, root=0, full=HostBuf(full) if full is not None else None)
        r.sync()

    try:
        impl, comm, why = pmd.agree_on_c_abi_gather(requested, world, pm.Comm.unique_id, broadcast_id,
                                                     lambda uid: pm.Comm(r, uid, rank, world), try_exchange, all_min)
        line = {"rank": rank, "n_gpus": world, "config": {"gather_impl": impl, **({"gather_fallback": why} if why else {}),
                                                         "t_gather_wire_floor_ms": round(pmd.gather_wire_floor_ms(layout, width, world), 6)}}
        if impl == "cabi":
            ok = full is None or np.array_equal(full, pmo.render(pmo.scene_cardioid(), width, height))
            line["gathered_frame_equals_oracle"] = bool(ok)
    except RuntimeError as e:
        line = {"rank": rank, "n_gpus": world, "error": str(e)}
    dist.barrier()
    print(json.dumps(line), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
