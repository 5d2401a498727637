"""ctypes binding of the C ABI declared in include/piet_metal_amd.h.

This is exactly the binding a reference-side maintainer would write against the
shared library (see INTEGRATION.md for the Rust / Objective-C equivalents).  There
is no Python or CPU fallback: if the HIP library is missing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PM_LIB_VARIANT=strict (tests / fuzzing only): the build whose LDS-only barriers are full __syncthreads()
# (csrc/pm_kernels_common.h, LdsBarrier) -- the same C ABI, the same kernels otherwise
_VARIANT = os.environ.get("PM_LIB_VARIANT", "")
if _VARIANT not in ("", "strict") and not (os.environ.get("PM_LIB_DEV") == "1" and _VARIANT.isalnum()):  # (PM_LIB_DEV: a developer's hand-built A/B library)
    raise ImportError(f"PM_LIB_VARIANT={_VARIANT!r}: only 'strict' exists")
LIB_PATH = os.path.join(_HERE, "lib", "libpiet_metal_amd" + ("_" + _VARIANT if _VARIANT else "") + ".so")

PM_OK = 0
PM_ERR_INVALID = -1
PM_ERR_NO_DEVICE = -2
PM_ERR_HIP = -3
PM_ERR_CAPACITY = -4
PM_ERR_SCENE = -5
PM_ERR_PARSE = -6

PM_EL_MOVE, PM_EL_LINE, PM_EL_QUAD, PM_EL_CURVE, PM_EL_CLOSE = range(5)
PM_PATH_FILL, PM_PATH_STROKE, PM_PATH_EVEN_ODD, PM_PATH_COMPOUND = 1, 2, 4, 8
PM_FILL_EVEN_ODD = 1
PM_FILL_COMPOUND = 2
PM_SVG_REJECT_ARC_PATHS = 1
PM_SVG_SPEC_DEFAULTS = 2
PM_SVG_FLAT_GRADIENTS = 4
PM_FMT_RGBA8, PM_FMT_BGRA8 = 0, 1


class PathEl(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("pad", C.c_uint32), ("p", C.c_double * 6)]


class Path(C.Structure):
    _fields_ = [
        ("el_begin", C.c_uint32),
        ("el_end", C.c_uint32),
        ("flags", C.c_uint32),
        ("fill_rgba", C.c_uint32),
        ("stroke_rgba", C.c_uint32),
        ("stroke_width", C.c_float),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("tiles_x", C.c_uint32),
        ("tiles_y", C.c_uint32),
        ("band_row0", C.c_uint32),
        ("band_row1", C.c_uint32),
        ("n_items", C.c_uint32),
        ("queued_tiles", C.c_uint32),
        ("arena_used_dwords", C.c_uint32),
        ("arena_cap_dwords", C.c_uint32),
        ("overflow", C.c_uint32),
        ("scene_bytes", C.c_uint32),
        ("heavy_tiles", C.c_uint32),
        ("ptcl_used_cmds", C.c_uint32),
    ]


class SceneTimings(C.Structure):
    _fields_ = [("flatten_encode_ms", C.c_float), ("scene_index_ms", C.c_float), ("arena_setup_ms", C.c_float)]


class Cmd(C.Structure):
    _fields_ = [("tag", C.c_uint32), ("body", C.c_uint32 * 5)]


assert C.sizeof(PathEl) == 56 and C.sizeof(Path) == 24 and C.sizeof(Cmd) == 24

# name -> (restype, argtypes); every symbol include/piet_metal_amd.h declares
SIGNATURES = {
    "init_test_scene": (None, [C.c_void_p, C.c_ssize_t]),
    "pm_encoder_new": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "pm_encoder_free": (None, [C.c_void_p]),
    "pm_encoder_alloc": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "pm_encoder_write_struct": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "pm_encoder_encode_points": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]),
    "pm_encoder_begin_group": (C.c_int, [C.c_void_p, C.c_size_t]),
    "pm_encoder_end_group": (C.c_int, [C.c_void_p]),
    "pm_encoder_circle": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double]),
    "pm_encoder_ellipse": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    "pm_encoder_fill_compound": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]),
    "pm_encoder_fill_path": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]),
    "pm_encoder_stroke_path": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_float]),
    "pm_encoder_stroke_line": (C.c_int, [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_float, C.c_uint32]),
    "pm_encoder_fill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32]),
    "pm_encoder_fill_rule": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]),
    "pm_encoder_polyline": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_float]),
    "pm_encoder_bytes_used": (C.c_size_t, [C.c_void_p]),
    "pm_scene_cardioid": (C.c_int64, [C.c_void_p, C.c_size_t]),
    "pm_scene_path_test": (C.c_int64, [C.c_void_p, C.c_size_t]),
    "pm_svg_parse": (C.c_void_p, [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(C.c_int)]),
    "pm_svg_tiger": (C.c_void_p, [C.c_int, C.POINTER(C.c_int)]),
    "pm_svg_free": (None, [C.c_void_p]),
    "pm_svg_n_paths": (C.c_size_t, [C.c_void_p]),
    "pm_svg_n_els": (C.c_size_t, [C.c_void_p]),
    "pm_svg_viewbox": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pm_svg_paths": (C.POINTER(Path), [C.c_void_p]),
    "pm_svg_els": (C.POINTER(PathEl), [C.c_void_p]),
    "pm_parse_color": (C.c_uint32, [C.c_char_p]),
    "pm_create": (C.c_void_p, [C.c_int, C.POINTER(C.c_int)]),
    "pm_destroy": (None, [C.c_void_p]),
    "pm_last_error": (C.c_char_p, []),
    "pm_abi_version": (C.c_uint32, []),
    "pm_resize": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "pm_set_band": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "pm_scene_buffer": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pm_scene_reserve": (C.c_int, [C.c_void_p, C.c_size_t]),
    "pm_upload_scene": (C.c_int, [C.c_void_p, C.c_size_t]),
    "pm_flatten_and_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_double), C.c_float, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "pm_reflatten": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_float, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "pm_download_scene": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pm_render": (C.c_int, [C.c_void_p]),
    "pm_render_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_sync": (C.c_int, [C.c_void_p]),
    "pm_set_target_format": (C.c_int, [C.c_void_p, C.c_int]),
    "pm_read_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "pm_framebuffer_device_ptr": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]),
    "pm_scene_device_ptr": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "pm_time_frames": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pm_frame_latency": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pm_tile_kernel_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "pm_binning_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "pm_binning_plan_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "pm_one_launch_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int)]),
    "pm_time_one_launch": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "pm_debug_time_frame": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pm_time_frames_pipelined": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "pm_debug_frame_timeline": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "pm_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "pm_fill_coverage": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "pm_layout_selfcheck": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "pm_get_scene_timings": (C.c_int, [C.c_void_p, C.POINTER(SceneTimings)]),
    "pm_get_binning_plans": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32)]),
    "pm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "pm_comm_create": (C.c_void_p, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "pm_comm_destroy": (None, [C.c_void_p]),
    "pm_comm_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]),
    "pm_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "pm_debug_capture_ptcl": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pm_debug_time_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "pm_debug_time_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
}

_lib = None


class PietMetalError(RuntimeError):
    def __init__(self, status: int, what: str):
        super().__init__(f"{what}: status {status}: {last_error()}")
        self.status = status


PM_ABI_VERSION = 600  # include/piet_metal_amd.h


def load() -> C.CDLL:
    """Load libpiet_metal_amd.so (built by __graft_entry__.build()); fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(piet_metal_amd has no CPU/Python fallback)"
        )
    # One HIP runtime per process: the PyTorch-ROCm wheel bundles its own libamdhip64 /
    # libhsa-runtime64.  If it is loaded after ours (from /opt/rocm) the second runtime
    # finds no GPU, so when torch is installed let it load first; our library's
    # NEEDED libamdhip64.so.7 then binds to the copy that is already resident.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.pm_abi_version() != PM_ABI_VERSION:  # (include/piet_metal_amd.h: the struct layouts this binding hands over)
        raise ImportError(f"{LIB_PATH}: ABI {lib.pm_abi_version()}, this binding is for {PM_ABI_VERSION} -- rebuild the library")
    _lib = lib
    return lib


def last_error() -> str:
    s = load().pm_last_error()
    return s.decode("utf-8", "replace") if s else ""


def check(status: int, what: str) -> None:
    if status != PM_OK:
        raise PietMetalError(status, what)
