"""piet_metal_amd: MI355X-native (gfx950) drop-in for piet-metal's compute path.

Host-side mirror of the reference interfaces for this path:
  Encoder / scene_* / PathSet   src/lib.rs (Encoder, test scenes, make_tiger input)
  Renderer                      TestApp/PietRenderer.{h,m}
All rendering happens in lib/libpiet_metal_amd.so (hand-written HIP kernels); the
import fails loudly if that library has not been built.
"""
from . import _lib
from ._lib import PietMetalError
from .encoder import Encoder, PathSet, parse_color, scene_cardioid, scene_path_test
from .renderer import Comm, Renderer, init_test_scene
from . import workloads

_lib.load()  # no library => ImportError here, never a silent fallback

__all__ = [
    "Comm", "Encoder", "PathSet", "Renderer", "PietMetalError", "init_test_scene", "parse_color",
    "scene_cardioid", "scene_path_test", "workloads",
]
