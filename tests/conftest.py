import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure both shared libraries exist (no-op when __graft_entry__.build() already ran)."""
    lib = os.path.join(ROOT, "piet_metal_amd", "lib", "libpiet_metal_amd.so")
    ora = os.path.join(ROOT, "oracle", "libpmo_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__ as g

        g.build()
    return True


def _swap_in_the_emulated_library():
    """PM_TEST_EMU=1 (test infrastructure, GPU-less boxes only): the gpu-marked parity tests run
    against tests/emu/_build/libpiet_metal_amd_emu.so -- the kernel sources of piet_metal_amd/csrc
    compiled as plain C++ and run lane by lane on the CPU with wave64 semantics (tests/emu/).  It
    checks kernel LOGIC against the oracle before a GPU is spent on it; the product never loads it,
    and with a GPU present this switch is refused."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_swap import swap_in_emulated_library

    return swap_in_emulated_library()


@pytest.fixture(scope="session")
def pm(built):
    import piet_metal_amd

    if os.environ.get("PM_TEST_EMU") == "1":
        _swap_in_the_emulated_library()
    return piet_metal_amd


@pytest.fixture(scope="session")
def pmo(built):
    from oracle import pmo as m

    m.load()
    return m


@pytest.fixture(scope="session")
def renderer(pm):
    r = pm.Renderer(0)
    yield r
    r.close()
