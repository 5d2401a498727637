"""The layout generator: pm_layoutgen turns the `piet_gpu!`-style description
piet_metal_amd/layout/piet_layout.pgpu into piet_metal_amd/csrc/pm_layout_gen.h, the HIP / C++
target the reference's piet-gpu-derive (MSL / HLSL only, loaders and writers unfinished,
piet-gpu-derive/src/lib.rs:24-27, :1051-1117) does not have."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "piet_metal_amd", "csrc", "pm_layoutgen.cpp")
DESC = os.path.join(ROOT, "piet_metal_amd", "layout", "piet_layout.pgpu")
HDR = os.path.join(ROOT, "piet_metal_amd", "csrc", "pm_layout_gen.h")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("layoutgen") / "pm_layoutgen")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-o", exe, SRC])
    return exe


def test_committed_header_is_what_the_generator_prints(tool):
    out = subprocess.run([tool, DESC], capture_output=True, text=True, check=True).stdout
    assert out == open(HDR).read(), "pm_layout_gen.h is stale: make -C piet_metal_amd/csrc regen-layout"
    # and it is a self-contained header for a plain host compiler
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-x", "c++", "-include", HDR, "/dev/null"])


def test_generator_layout_rules_and_errors(tool, tmp_path):
    d = tmp_path / "t.pgpu"
    d.write_text("""
    piet_gpu! { mod demo {
        struct A { a: u8, b: u32, c: [u16; 4], d: [f32; 2], e: i16 }   // not a variant: no tag
        struct B { x: f32 }
        struct C { r: Ref<A>, v: [u8; 4] }
        enum E { First(B), Unit, Third(C) = 7, Fourth }
    } }""")
    out = subprocess.run([tool, str(d)], capture_output=True, text=True, check=True).stdout
    for needle in [
        "static_assert(offsetof(APacked, a) == 0", "static_assert(offsetof(APacked, b) == 4",
        "static_assert(offsetof(APacked, c) == 8", "static_assert(offsetof(APacked, d) == 16",
        "static_assert(offsetof(APacked, e) == 24", "constexpr uint32_t A_SIZE = 28;",
        "static_assert(offsetof(BPacked, x) == 4",  # variant: the tag comes first
        "static_assert(offsetof(CPacked, r) == 4", "static_assert(offsetof(CPacked, v) == 8", "constexpr uint32_t C_SIZE = 12;",
        "constexpr uint32_t E_First = 1;", "constexpr uint32_t E_Unit = 2;", "constexpr uint32_t E_Third = 7;", "constexpr uint32_t E_Fourth = 8;",
        "constexpr uint32_t E_SIZE = 16;", "BPacked B_load(const E &s)", "E E_Third_pack(ARef r, pm_u8x4 v)", "E_write_tag(",
    ]:
        assert needle in out, needle
    (tmp_path / "p.cc").write_text('#include "t.h"\nint main() { return 0; }\n')
    (tmp_path / "t.h").write_text(out)
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", str(tmp_path), "-o", str(tmp_path / "p"), str(tmp_path / "p.cc")])
    for bad in ["mod m { struct S { a: u64 } }", "mod m { enum E { V(Nope) } }", "mod m { struct S { a: Ref<Nope> } }", "mod m { struct S { a: [f32; 9] } }", "mod m { struct"]:
        d.write_text(bad)
        assert subprocess.run([tool, str(d)], capture_output=True).returncode == 2, bad


def test_generated_accessors_agree_with_the_encoder_and_the_command_lists(pm, pmo):
    """pm_layout_selfcheck: every item of real scenes and every command of real per-tile lists
    through the generated readers / loaders / writers, byte for byte."""
    from test_host_cpu import encode_ops, extend_ops, random_ops

    lib = pm._lib.load()
    scenes = [pmo.scene_cardioid(), pmo.scene_path_test(), encode_ops(pm, extend_ops(5, random_ops(5, 120, extent=300.0)))]
    for scene in scenes:
        P = pmo.Ptcl(scene, 320, 304)
        cmds = np.concatenate([P.cmds(tx, ty) for ty in range(P.tiles_y) for tx in range(P.tiles_x)]).astype(np.uint32)
        P.close()
        assert len(cmds) > 100
        assert lib.pm_layout_selfcheck(scene.ctypes.data, scene.size, cmds.ctypes.data, len(cmds)) == 0
    # it does notice disagreement: a command with a stray word where the layout has padding
    bad = np.array([[3, 0xdead, 0, 0, 0, 0]], np.uint32)  # Line: body[0] is padding
    assert lib.pm_layout_selfcheck(None, 0, bad.ctypes.data, 1) < 0
