import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session", autouse=True)
def _native_pieces_built():
    """The suites need the in-tree native build (libkcfx.so, libkcfhost.so, bin/kcfx_synth, bin/kcftools-amd). It is
    normally there (`__graft_entry__.build()`); a fresh checkout gets it here, once."""
    from kcftools_amd import engine
    need = [engine.LIB_PATH, os.path.join(ROOT, "kcftools_amd", "libkcfhost.so"),
            os.path.join(ROOT, "kcftools_amd", "bin", "kcfx_synth"), os.path.join(ROOT, "kcftools_amd", "bin", "kcftools-amd")]
    if os.environ.get("KCFX_LIB") is None and not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()
