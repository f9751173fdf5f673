"""What hipcc made of the kernels, read from the gfx950 code object inside libkcfx.so (no GPU needed): the screening
kernels must not spill (no scratch), must fit the register budget their occupancy is planned with (<= 64 VGPRs = 8 waves per
SIMD) and the LDS a workgroup takes must allow 8 workgroups per CU. A change that silently breaks one
of these costs tens of per cent on the GPU and is invisible to the parity tests."""
import os
import re
import shutil
import subprocess

import pytest

from kcftools_amd import engine

LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    objdump, readelf = os.path.join(LLVM, "llvm-objdump"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf of the ROCm toolchain not found")
    d = tmp_path_factory.mktemp("co")
    so = shutil.copy(engine.LIB_PATH, d / "libkcfx.so")
    subprocess.run([objdump, "--offloading", str(so)], check=True, capture_output=True, cwd=d)
    co = [f for f in os.listdir(d) if "gfx950" in f]
    assert len(co) == 1, os.listdir(d)                     # one architecture, one code object: gfx950 only
    notes = subprocess.run([readelf, "--notes", str(d / co[0])], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        out[name] = {k: int(re.search(r"\." + k + r":\s+(\d+)", block).group(1))
                     for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    return out


def test_screening_kernels_do_not_spill_and_fit_their_occupancy(kernels):
    hot = {n: m for n, m in kernels.items() if re.search(r"screen_k31c_kernel|screen_k31_kernel|screen_k31_multi|screen_tiles_kernel|reduce_windows", n)}
    # k = 19 .. 31 x canonical / forward-only x planned / text-only reference, the line-table kernels, the general kernel, the reduction
    assert sum("screen_k31c_kernel" in n for n in hot) == 52 and len(hot) >= 62
    text_only = lambda name: "screen_k31c_kernel" in name and name.endswith("Lb0EEEvPKhNS_7RefPlanEPKNS_8TileDescENS_9TableViewEiPmPl")
    for name, m in hot.items():
        assert m["private_segment_fixed_size"] == 0, (name, m)           # no scratch
        # 8 waves per SIMD (<= 64 VGPRs); the multi-sample line-table kernel is planned for 6 (amdgpu_waves_per_eu(6, 8): <= 80);
        # the text-only form of the cell-table kernel (round 5: ONE wave per tile that plans 32-position strips, kcfx_plan.h, and
        # then walks the tile's 32 words) for 7: <= 72 -- at 64 it spills, at 80 it loses a wave and 3 % (profiles/r05_notes.md)
        assert m["vgpr_count"] <= (80 if "screen_k31_multi" in name else 72 if text_only(name) else 64), (name, m)
    for name, m in hot.items():
        if "screen_k31c_kernel" in name:
            # planned reference: two waves per workgroup, 5.7 KB of LDS (codes both ways, anchors, the two queues of each wave) -- 16
            # workgroups per CU. text-only: one wave per workgroup, the planner's exchange rows UNDER the codes, anchors and queues:
            # 4.7 KB (round 6: + 512 B for the queue of the k-mers their second line sent on) -- 28 workgroups (7 waves per SIMD)
            # take 132 KB of a CU's 160
            assert m["group_segment_fixed_size"] <= (4864 if text_only(name) else 5888), (name, m)
        elif "screen_" in name:
            assert m["group_segment_fixed_size"] <= 20480, (name, m)     # 160 KB LDS per CU / 8 workgroups
    # a planned reference against a table slice (round 6): the list of the slice's positions beside the plan and the queues -- 9.7 KB,
    # 16 workgroups of two waves per CU = the 8 waves per SIMD its 64 registers allow. With the hits as one byte per position in
    # LDS (12 KB, 13 workgroups) a pass took 1.12 ms where this form takes 0.90 (profiles/r06_notes.md section 11)
    sliced = {n: m for n, m in kernels.items() if "screen_k31c_slice_kernel" in n}
    assert len(sliced) == 26                                             # k = 19 .. 31 x canonical / forward-only
    for name, m in sliced.items():
        assert m["private_segment_fixed_size"] == 0 and m["vgpr_count"] <= 64 and m["group_segment_fixed_size"] <= 10240, (name, m)


def test_no_kernel_of_the_library_uses_scratch(kernels):
    """(the table-build kernels walk a k-mer's candidate cells with a visitor, cells_visit, instead of an array)"""
    assert len(kernels) > 30
    spilling = {n: m["private_segment_fixed_size"] for n, m in kernels.items() if m["private_segment_fixed_size"]}
    assert not spilling, spilling
