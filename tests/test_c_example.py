"""The drop-in boundary from plain C: examples/screen_windows.c includes include/kcfx.h as C99 (-pedantic -Werror), links
libkcfx.so and nothing else of this repository. CPU: it compiles and links (the header is C, every symbol it uses is
exported). GPU: its output equals what the Python binding gets, and the oracle."""
import os
import subprocess

import numpy as np
import pytest

from kcftools_amd import engine, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def example(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("cex") / "screen_windows")
    libdir = os.path.dirname(engine.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-O2", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "screen_windows.c"), "-L" + libdir, "-lkcfx", "-Wl,-rpath," + libdir,
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    return exe


def test_header_is_c99_and_the_example_links(example):
    assert os.access(example, os.X_OK)
    p = subprocess.run([example], capture_output=True, text=True)          # no arguments: usage, exit 2, no GPU touched
    assert p.returncode == 2 and "usage:" in p.stderr


@pytest.mark.gpu
def test_c_example_output_equals_binding_and_oracle(example, oracle, tmp_path):
    k = 31
    ref = synth.random_reference(200_000, 77)
    sample = synth.mutate_sample(ref, 78)
    prm = synth.KmcParams(k=k, p=3, counter_size=2, sig_len=7, n_bins=8, max_count=65535)
    kmers, counts = synth.build_sample_db(sample, prm, 79, noise_kmers=3000)
    prefix = str(tmp_path / "db")
    synth.write_kmc(prefix, kmers, counts, prm)
    seq = tmp_path / "seq.txt"
    seq.write_bytes(b"\n".join(ref[i:i + 70] for i in range(0, len(ref), 70)) + b"\n")      # line-wrapped: newlines are ignored
    p = subprocess.run([example, prefix, str(seq), "10000", "2"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rows = [ln.split(",") for ln in p.stdout.strip().splitlines()[1:]]
    s, e = engine.windows(len(ref), 10_000, 0, k)
    assert len(rows) == len(s)
    want = oracle.OracleDb(prefix).screen(ref, s, e - s, 2)
    got = engine.KmcDb.open(prefix).screen(ref, s, e - s, 2)
    sc, md = engine.score(got)
    cols = ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right", "kmer_count_sum")
    for i, r in enumerate(rows):
        assert (int(r[0]), int(r[1])) == (int(s[i]), int(e[i]))
        for j, f in enumerate(cols):
            assert int(r[2 + j]) == int(want[f][i]) == int(got[f][i]), (i, f)
        assert float(r[10]) == sc[i] and float(r[11]) == md[i]             # %.17g round-trips a double
