"""CPU-side hardening (sanitizers; GPU AddressSanitizer does not exist on the pool):

* the oracle (oracle/kcf_oracle.c) and the host modules of the product (kcftools_amd/csrc/host/*.cpp) are built with
  -fsanitize=address,undefined and the known-answer, cross-model and host-side suites are run through those builds;
* the .kmc_pre parser of the product (kcftools_amd/csrc/kmc_pre.cpp = what kcfx_db_open / kcfx_db_header run on the host,
  KMC.java:107-168) is fed truncated and corrupted prefix files: under the sanitizers through a small driver, and through
  the shipped library's kcfx_db_header: a status code every time, never a crash, never a sanitizer report."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
GOLDEN_PRE = sorted(f for f in os.listdir(G) if f.endswith(".kmc_pre"))


def _san_env():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("gcc has no libasan.so here")
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:exitcode=97"        # (CPython itself is not leak-clean)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1:exitcode=98"
    return env


def _run_suite_under_sanitizers(files, extra_env):
    env = _san_env()
    env.update(extra_env)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + files,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
    assert p.returncode == 0, out[-3000:]
    return out


def test_oracle_suites_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libkcforacle_asan.so"], check=True, capture_output=True)
    so = os.path.join(ROOT, "oracle", "libkcforacle_asan.so")
    out = _run_suite_under_sanitizers(["tests/test_oracle_kats.py", "tests/test_oracle_vs_pymodel.py"], {"KCF_ORACLE_SO": so})
    assert " passed" in out


def test_host_modules_under_asan_ubsan():
    subprocess.run(["make", "-C", os.path.join(ROOT, "kcftools_amd", "csrc"), "../../build/libkcfhost_asan.so"], check=True,
                   capture_output=True)
    so = os.path.join(ROOT, "build", "libkcfhost_asan.so")
    out = _run_suite_under_sanitizers(["tests/test_host_side.py"], {"KCF_HOST_SO": so})
    assert " passed" in out


# ---- malformed .kmc_pre ------------------------------------------------------------------------------------------------
def _mutants(data: bytes, rng: np.random.Generator, n: int):
    """Truncations, byte flips (biased towards the header at the end and the markers), extreme values in the header's
    32-bit fields, and a corrupted header-length word."""
    size = len(data)
    hdr_off = int.from_bytes(data[-8:-4], "little")
    hdr = size - hdr_off - 8
    extremes = (0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 31, 32, 33, 64, 0x200, 0x201)
    for i in range(n):
        kind = i % 5
        b = bytearray(data)
        if kind == 0:                                   # truncated anywhere (a few: right around the header)
            cut = int(rng.integers(0, size)) if i % 10 else max(0, size - int(rng.integers(1, 90)))
            b = b[:cut]
        elif kind == 1:                                 # random byte flips in the trailing 128 bytes
            for _ in range(int(rng.integers(1, 9))):
                b[size - 1 - int(rng.integers(0, min(128, size)))] = int(rng.integers(0, 256))
        elif kind == 2:                                 # one header field := an extreme value
            f = int(rng.integers(0, 17))
            v = extremes[int(rng.integers(0, len(extremes)))]
            b[hdr + 4 * f:hdr + 4 * f + 4] = int(v).to_bytes(4, "little")
        elif kind == 3:                                 # the header-length word itself
            v = (extremes + (hdr_off - 1, hdr_off + 1, size, size - 8, size - 12))[int(rng.integers(0, len(extremes) + 5))]
            b[size - 8:size - 4] = int(v & 0xFFFFFFFF).to_bytes(4, "little")
        else:                                           # LUT damage: one entry non-monotonic / past the total / flipped
            pos = 4 + 8 * int(rng.integers(0, max(1, (size - 200) // 8)))
            b[pos:pos + 8] = int(rng.integers(0, 2**63)).to_bytes(8, "little")
        yield bytes(b)


def test_malformed_kmc_pre_never_crashes_the_parser_under_asan_ubsan(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "kcftools_amd", "csrc"), "../../build/fuzz_kmc_pre_asan"], check=True,
                   capture_output=True)
    exe = os.path.join(ROOT, "build", "fuzz_kmc_pre_asan")
    rng = np.random.default_rng(20261003)
    paths, pristine = [], []
    for g in GOLDEN_PRE:
        data = open(os.path.join(G, g), "rb").read()
        p = tmp_path / f"ok_{g}"
        p.write_bytes(data)
        pristine.append(str(p))
        for j, m in enumerate(_mutants(data, rng, 400)):
            q = tmp_path / f"m{j}_{g}"
            q.write_bytes(m)
            paths.append(str(q))
    paths.append(str(tmp_path / "does_not_exist.kmc_pre"))
    (tmp_path / "empty.kmc_pre").write_bytes(b"")
    paths.append(str(tmp_path / "empty.kmc_pre"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:exitcode=97", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")
    statuses = []
    for lo in range(0, len(pristine + paths), 200):
        chunk = (pristine + paths)[lo:lo + 200]
        p = subprocess.run([exe] + chunk, capture_output=True, text=True, env=env, timeout=600)
        assert p.returncode == 0 and "Sanitizer" not in p.stderr and "runtime error" not in p.stderr, p.stderr[-3000:]
        lines = p.stdout.strip().splitlines()
        assert len(lines) == len(chunk)
        statuses += [int(ln.split()[0]) for ln in lines]
    assert statuses[:len(pristine)] == [0] * len(pristine)              # the golden files themselves parse
    assert set(statuses) <= {0, 2, 3}                                   # KCFX_OK / KCFX_ERR_IO / KCFX_ERR_FORMAT
    rejected = sum(1 for s in statuses[len(pristine):] if s != 0)
    assert rejected > len(paths) // 3, (rejected, len(paths))           # the mutations do reach the checks


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture])
@given(st.data())
def test_malformed_kmc_pre_through_kcfx_db_header(tmp_path, data):
    """The same through the shipped library (host-only entry point: no GPU is touched)."""
    from kcftools_amd import engine
    L = engine.lib()
    g = data.draw(st.sampled_from(GOLDEN_PRE))
    raw = bytearray(open(os.path.join(G, g), "rb").read())
    size = len(raw)
    op = data.draw(st.sampled_from(["cut", "flip", "field", "lut"]))
    if op == "cut":
        raw = raw[:data.draw(st.integers(0, size - 1))]
    elif op == "flip":
        for _ in range(data.draw(st.integers(1, 6))):
            raw[size - 1 - data.draw(st.integers(0, 99))] = data.draw(st.integers(0, 255))
    elif op == "field":
        hdr = size - int.from_bytes(raw[-8:-4], "little") - 8
        f = data.draw(st.integers(0, 16))
        raw[hdr + 4 * f:hdr + 4 * f + 4] = data.draw(st.integers(0, 2**32 - 1)).to_bytes(4, "little")
    else:
        pos = 4 + 8 * data.draw(st.integers(0, (size - 200) // 8))
        raw[pos:pos + 8] = data.draw(st.integers(0, 2**64 - 1)).to_bytes(8, "little")
    prefix = tmp_path / "fuzz"
    (tmp_path / "fuzz.kmc_pre").write_bytes(bytes(raw))
    info = engine.DbInfo()
    rc = L.kcfx_db_header(str(prefix).encode(), C.byref(info))
    assert rc in (0, 2, 3), (rc, L.kcfx_last_error())
    if rc:
        assert L.kcfx_last_error()                                     # a message, not a crash
    else:
        assert 1 <= info.kmer_length <= 32 and info.version == 0x200 and info.lut_suspect == 0


def test_lut_damage_is_refused_unless_lenient(tmp_path, monkeypatch):
    """a non-monotonic LUT gives records arbitrary prefixes -- refused by default, KCFX_LENIENT_LUT=1 restores
    the reference's unchecked read (KMC.java:153-163) and flags the handle."""
    from kcftools_amd import engine
    L = engine.lib()
    g = [f for f in GOLDEN_PRE if "p3" in f][0]
    raw = bytearray(open(os.path.join(G, g), "rb").read())
    raw[4 + 8 * 5:4 + 8 * 6] = (2**40).to_bytes(8, "little")           # entry 5 far beyond its neighbours and the total
    (tmp_path / "bad.kmc_pre").write_bytes(bytes(raw))
    info = engine.DbInfo()
    monkeypatch.delenv("KCFX_LENIENT_LUT", raising=False)
    assert L.kcfx_db_header(str(tmp_path / "bad").encode(), C.byref(info)) == 3
    assert b"LUT" in L.kcfx_last_error()
    monkeypatch.setenv("KCFX_LENIENT_LUT", "1")
    assert L.kcfx_db_header(str(tmp_path / "bad").encode(), C.byref(info)) == 0
    assert info.lut_suspect == 1
