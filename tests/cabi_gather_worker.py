"""Worker process of tests/test_dist_cpu.py::test_c_abi_gather_across_processes_with_a_mock_rccl:
one rank of pm_comm_create / pm_gather (the C-ABI form of the band gather) on a GPU-less box -- the
library is the CPU emulation of tests/emu, RCCL is tests/mock_rccl (PM_RCCL_LIB)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class HostBuf:
    """What Comm.gather expects of a tensor, over a numpy array ("device" memory of the emulation)."""

    def __init__(self, a):
        self.a = a

    def data_ptr(self):
        return self.a.ctypes.data

    def stride(self, k):
        return self.a.strides[k]


def main():
    rank, world, width, height, box = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    from emu_swap import swap_in_emulated_library

    swap_in_emulated_library(build=False)
    import piet_metal_amd as pm
    from oracle import pmo
    from piet_metal_amd import _lib
    from piet_metal_amd import dist as pmd

    scene = pmo.scene_cardioid()
    tiles_y = (height + 15) // 16
    cuts = pmd.balanced_cuts([b[0] for b in pmd.band_layout(height, world)] + [tiles_y], [1.0 + 2.5 * k for k in range(world)])
    layout = pmd.band_layout(height, world, cuts)
    r0, r1, rows = layout[rank]
    r = pm.Renderer(0)
    r.resize(width, height)
    r.set_scene_bytes(scene)
    r.set_band(r0, r1)
    uid_path = os.path.join(box, "uid.bin")
    if rank == 0:
        uid = pm.Comm.unique_id()
        with open(uid_path + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_path + ".tmp", uid_path)
    else:
        t0 = time.time()
        while not os.path.exists(uid_path):
            assert time.time() - t0 < 60, "no id from rank 0"
            time.sleep(0.01)
        uid = open(uid_path, "rb").read()
    comm = pm.Comm(r, uid, rank, world)
    root = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    # what bench.py records in its line: the library that got bound and the communicator's size
    info = comm.info()
    assert info["rccl_ranks"] == world, info
    assert os.path.samefile(info["rccl_lib"], os.environ["PM_RCCL_LIB"]) and pm.Comm.library_path() == info["rccl_lib"], info
    # 1. every rank's LAST FRAME (the context's own framebuffer) -> the root's image
    r.render()
    full = np.zeros((height, width, 4), np.uint8) if rank == root else None
    comm.gather(layout, root=root, full=HostBuf(full) if full is not None else None)
    r.sync()
    if rank == root:
        np.save(os.path.join(box, "full_a.npy"), full)
    # 2. the root renders straight into its rows of the image (pm_render_to: the band is already in
    #    place, pm_gather must not touch it), the others hand a caller-owned band over
    band = np.zeros((max(rows, 1) if rank != root else 0, width, 4), np.uint8)
    full2 = np.full((height, width, 4), 7, np.uint8) if rank == root else None
    target = full2[r0 * 16 : r0 * 16 + rows] if rank == root else band
    _lib.check(_lib.load().pm_render_to(r._h, target.ctypes.data, width * 4, None), "pm_render_to")
    r.sync()
    comm.gather(layout, root=root, full=HostBuf(full2) if full2 is not None else None, band=HostBuf(target))
    r.sync()
    if rank == root:
        np.save(os.path.join(box, "full_b.npy"), full2)
    # 3. a band table that does not name this context's band is refused before anything is sent
    bad = [(b[0] + (1 if k == rank else 0), b[1], b[2]) for k, b in enumerate(layout)]
    try:
        comm.gather(bad, root=root, full=HostBuf(full2) if full2 is not None else None, band=HostBuf(target))
        refused = False
    except pm.PietMetalError as e:
        refused = e.status == _lib.PM_ERR_INVALID
    assert refused
    # 4. a root whose own band overlaps its rows of the image without being them: the error comes AFTER the
    #    root has posted its receives -- the peers' sends complete (and the test below finds no message left)
    if rows >= 2 or rank != root:
        shifted = full2[r0 * 16 + 1 : r0 * 16 + 1 + rows] if rank == root else target
        if rank == root and shifted.shape[0] < rows:  # (the last band: shift up instead)
            shifted = full2[r0 * 16 - 1 : r0 * 16 - 1 + rows]
        try:
            comm.gather(layout, root=root, full=HostBuf(full2) if full2 is not None else None, band=HostBuf(np.ascontiguousarray(shifted) if rank != root else shifted))
            late = rank != root
        except pm.PietMetalError as e:
            late = rank == root and e.status == _lib.PM_ERR_INVALID
        assert late
        r.sync()
    # 5. the gather pipelined under the render (bench.py --gather-chunks): every band in four sub-bands, a context each,
    #    ONE communicator; sub-band k of every rank lands in its rows of the image (empty sub-bands send nothing)
    chunks = 4
    subs = []
    for lay in pmd.sub_band_layouts(layout, height, chunks):
        s0, s1, srows = lay[rank]
        q = None
        if s1 > s0:
            q = pm.Renderer(0)
            q.resize(width, height)
            q.set_scene_bytes(scene)
            q.set_band(s0, s1)
        subs.append((q, lay, s0, srows))
    full3 = np.full((height, width, 4), 9, np.uint8) if rank == root else None
    band3 = np.zeros((max(rows, 1), width, 4), np.uint8)
    for q, lay, s0, srows in subs:
        if rank == root:
            target3 = full3[s0 * 16 : s0 * 16 + srows]
        else:
            target3 = band3[(s0 - r0) * 16 : (s0 - r0) * 16 + srows]
        if q is not None:
            _lib.check(_lib.load().pm_render_to(q._h, target3.ctypes.data, width * 4, None), "pm_render_to")
            q.sync()
        comm.gather(lay, root=root, full=HostBuf(full3) if full3 is not None else None, band=HostBuf(target3) if q is not None else HostBuf(band3), renderer=q or r)
        (q or r).sync()
    if rank == root:
        np.save(os.path.join(box, "full_c.npy"), full3)
    for q, *_ in subs:
        if q is not None:
            q.close()
    comm.close()
    r.close()
    print("rank", rank, "ok", flush=True)


if __name__ == "__main__":
    main()
