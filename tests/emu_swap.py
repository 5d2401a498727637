"""TEST INFRASTRUCTURE: make piet_metal_amd's ctypes layer call the CPU emulation of the library
(tests/emu/) in THIS process.  Used by tests/conftest.py (PM_TEST_EMU=1) and by worker processes of the
multi-process tests; refused on a box with a GPU.  The product never imports this."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def swap_in_emulated_library(build: bool = True):
    import torch

    if torch.cuda.is_available():
        raise RuntimeError("the emulated library is for GPU-less boxes: run the real one")
    # PM_EMU_SUFFIX=_O0 (+ PM_EMU_STRICT=1): the unoptimized build, where every source-level
    # cross-lane operation is one call site and a divergent one can be told from a duplicated one;
    # _small: a tiny LDS survivor list, so that small scenes reach pm_bin_kernel's spill path
    suffix = os.environ.get("PM_EMU_SUFFIX", "")
    opt = ["OPT=-O0", "SUFFIX=_O0"] if suffix == "_O0" else []
    if suffix == "_small":
        opt = ["SUFFIX=_small", "DEFS=-DPM_BIN_SURV_LDS=24"]
    if build:
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tests", "emu")] + opt)
    from piet_metal_amd import _lib

    emu = ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "_build" + suffix, "libpiet_metal_amd_emu.so"))
    for name, (restype, argtypes) in _lib.SIGNATURES.items():
        fn = getattr(emu, name)
        fn.restype = restype
        fn.argtypes = argtypes
    _lib._lib = emu
    return emu
