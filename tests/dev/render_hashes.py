"""TEST HELPER: renders BASELINE configs through whatever library variant PM_LIB_VARIANT names and prints one
sha256 per config (pixels) -- tests/test_gpu_parity.py::test_strict_barrier_build_renders_the_same_bytes runs it
once per build and compares the lines."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import piet_metal_amd as pm  # noqa: E402


def main():
    wls = {"config2": lambda: pm.workloads.tiger(1920, 1080, fills_only=True), "config3": lambda: pm.workloads.tiger(3840, 2160),
           "config4": pm.workloads.config4_blobs}
    print("lib", os.path.basename(pm._lib.LIB_PATH))
    with pm.Renderer(0) as r:
        for name in sys.argv[1:] or list(wls):
            wl = wls[name]()
            r.resize(wl.width, wl.height)
            r.flatten_and_encode(wl.paths, wl.affine, wl.width_scale)
            for _ in range(3):  # (frames in flight too: every slot)
                r.render()
            r.sync()
            px = r.read_pixels()
            print(name, hashlib.sha256(np.ascontiguousarray(px).tobytes()).hexdigest())


if __name__ == "__main__":
    main()
