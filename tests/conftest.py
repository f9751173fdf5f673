import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the switches that change what the library builds (KCFX_TABLE_*, KCFX_LOAD_*, KCFX_GENERIC_KERNEL, KCFX_REF_PLAN ...) are test
# hooks: the library reads them only when this is set too (kcfx_internal.h: hook()); the suites force table forms with them
os.environ.setdefault("KCFX_TEST_HOOKS", "1")


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


@pytest.fixture(autouse=True)
def _one_shot_screens_take_both_reference_forms(request, monkeypatch):
    """kcfx_screen keeps its one-shot reference as text (the screening launch plans its tiles itself); a resident reference
    is planned once by ref_plan_kernel. Two kernels for the same answer: in the GPU suites every KmcDb.screen(...) call also
    screens a planned Reference of the same text and windows and must get the identical records -- so every parity case
    written against the one-shot call covers both forms."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import numpy as np
    from kcftools_amd import engine
    one_shot = engine.KmcDb.screen

    def screen(self, text, win_begin, win_len, min_count=1):
        got = one_shot(self, text, win_begin, win_len, min_count)
        if len(text) > 2_000_000_000:                       # (the wheat-scale cases: their memory is budgeted for one reference)
            return got
        ref = engine.Reference(text, win_begin, win_len, self.info.kmer_length, device=self.info.device, expected_screens=0)
        try:
            if 19 <= self.info.kmer_length <= 31 and len(text) > 0 and "KCFX_REF_PLAN" not in os.environ:
                assert ref.planned
            planned = ref.screen(self, min_count)
        finally:
            ref.close()
        assert np.array_equal(got, planned), "one-shot (text-only) and planned reference disagree"
        return got

    monkeypatch.setattr(engine.KmcDb, "screen", screen)
    yield
