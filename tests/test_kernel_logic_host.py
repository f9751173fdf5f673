"""CPU lane-by-lane emulation of screen_tiles_kernel (tests/host_harness) against the oracle.

The harness is compiled from the same __host__ __device__ helpers the HIP kernel uses (kcfx_device.h,
kcfx_internal.h): base packing, k-mer extraction, ballot-mask gap accounting, the tile monoid. Only the
table probe is substituted (counts come from the oracle). This is a pre-GPU logic check, not a parity
claim -- the parity tests proper are tests/test_gpu_parity.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from kcftools_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
HDIR = os.path.join(HERE, "host_harness")

SUMMARY = np.dtype([("n", "<i4"), ("obs", "<i4"), ("lead", "<i4"), ("trail", "<i4"), ("var_in", "<i4"),
                    ("inner_in", "<i4"), ("runs", "<i4"), ("flags", "<u4"), ("count_sum", "<i8")])
RESULT = np.dtype([("total_kmers", "<i4"), ("eff_length", "<i4"), ("obs", "<i4"), ("var", "<i4"), ("inner", "<i4"),
                   ("left", "<i4"), ("right", "<i4"), ("_pad", "<i4"), ("kmer_count_sum", "<i8")])
FIELDS = ("total_kmers", "eff_length", "obs", "var", "inner", "left", "right", "kmer_count_sum")


@pytest.fixture(scope="module")
def harness():
    so = os.path.join(HDIR, "libharness.so")
    src = os.path.join(HDIR, "harness.cpp")
    deps = [src] + [os.path.join(HERE, "..", "kcftools_amd", "csrc", f) for f in ("kcfx_device.h", "kcfx_internal.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        hipcc = "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("hipcc not available")
        subprocess.run([hipcc, "-O1", "-std=c++17", "-fPIC", "-x", "hip", "--offload-arch=gfx950", "--cuda-host-only", "-shared", "-o", so, src],
                       check=True, capture_output=True)
    L = C.CDLL(so)
    L.harness_extract.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.harness_screen_tile.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.harness_combine.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.harness_identity.argtypes = [C.c_void_p]
    L.harness_finalize.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.harness_hash32.argtypes = [C.c_uint64]
    L.harness_hash32.restype = C.c_uint32
    L.harness_minimizers.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.harness_revcomp_word.argtypes = [C.c_uint64, C.c_int]
    L.harness_revcomp_word.restype = C.c_uint64
    L.harness_gap_both.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.harness_cells.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.harness_cells_of_key.argtypes = [C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.harness_cell_alt.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.harness_cell_alt.restype = C.c_int
    L.harness_cell_second_line.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
    L.harness_cell_second_line.restype = None
    L.harness_cell_rmask.argtypes = [C.c_int, C.c_int]
    L.harness_cell_rmask.restype = C.c_uint32
    L.harness_cell_third_line.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.harness_cell_third_line.restype = C.c_int
    L.harness_cell_line_index.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
    L.harness_cell_line_index.restype = None
    L.harness_plan_strips.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.harness_anchor_reference.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    return L


def padded(text: bytes) -> np.ndarray:
    return np.frombuffer(text + b"\0" * 128, dtype=np.uint8).copy()


def plan(nk: int, tile_k: int):
    if nk <= 0:
        return []
    nt = -(-nk // tile_k)
    per = min(tile_k, -(-(-(-nk // nt)) // 64) * 64)
    return [(s, min(per, nk - s)) for s in range(0, nk, per)]


def emulate_window(L, buf, begin, length, k, both, count_of, min_count, cuts=None):
    """fold the window's tiles through the harness; `cuts` overrides the tile plan (arbitrary split points)."""
    nk = length - k + 1
    acc = np.zeros(1, dtype=SUMMARY)
    L.harness_identity(acc.ctypes.data)
    tiles = plan(nk, L.harness_tile_k()) if cuts is None else cuts
    for s, n in tiles:
        fwd = np.zeros(n, dtype=np.uint64)
        rc = np.zeros(n, dtype=np.uint64)
        val = np.zeros(n, dtype=np.uint8)
        L.harness_extract(buf.ctypes.data, begin + s, n, k, fwd.ctypes.data, rc.ctypes.data, val.ctypes.data)
        key = np.minimum(fwd, rc) if both else fwd
        counts = np.array([count_of(int(x)) if v else -99 for x, v in zip(key, val)], dtype=np.int32)
        out = np.zeros(1, dtype=SUMMARY)
        L.harness_screen_tile(buf.ctypes.data, begin + s, n, k, counts.ctypes.data, min_count, out.ctypes.data)
        nxt = np.zeros(1, dtype=SUMMARY)
        L.harness_combine(acc.ctypes.data, out.ctypes.data, k, nxt.ctypes.data)
        acc = nxt
    res = np.zeros(1, dtype=RESULT)
    L.harness_finalize(acc.ctypes.data, k, res.ctypes.data)
    return {f: int(res[f][0]) for f in FIELDS}


def test_pack_and_extract_match_oracle(harness, oracle):
    rng = np.random.default_rng(1)
    alphabet = np.frombuffer(b"ACGTacgtNnRYKM-*\x00\xff@[`{AAAACCCCGGGGTTTT", dtype=np.uint8)
    for k in (31, 32, 15, 7, 1):
        text = alphabet[rng.integers(0, len(alphabet), 700)].tobytes()
        # long valid stretches too
        text = text[:200] + bytes(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), 300)) + text[200:]
        buf = padded(text)
        for begin in (0, 1, 17, 31, 32, 33, 95):
            n = len(text) - begin - k + 1
            fwd = np.zeros(n, dtype=np.uint64)
            rc = np.zeros(n, dtype=np.uint64)
            val = np.zeros(n, dtype=np.uint8)
            harness.harness_extract(buf.ctypes.data, begin, n, k, fwd.ctypes.data, rc.ctypes.data, val.ctypes.data)
            up = text.upper()
            want_valid = np.array([all(c in b"ACGT" for c in up[begin + j:begin + j + k]) for j in range(n)])
            assert np.array_equal(val.astype(bool), want_valid), (k, begin)
            # the valid ones, in order, are Fasta.getKmersList (oracle)
            lst = np.zeros(max(1, n), dtype=np.uint64)
            m = oracle.lib().kcfo_kmers_list(text[begin:], len(text) - begin, k, lst.ctypes.data)
            assert m == int(want_valid.sum())
            assert np.array_equal(fwd[want_valid], lst[:m])
            for j in np.nonzero(want_valid)[0][:60]:
                assert int(rc[j]) == oracle.lib().kcfo_kmer_revcomp(int(fwd[j]), k)


@pytest.mark.parametrize("k,both", [(31, True), (31, False), (15, True), (32, True)])
def test_emulated_kernel_matches_oracle(harness, oracle, tmp_path, k, both):
    p = 3 if (k - 3) % 4 == 0 else 4
    ref = synth.random_reference(40_000, 77 + k, n_run_every=6000, max_n_run=300, lower_frac=0.2, iupac_rate=1e-3)
    sample = synth.mutate_sample(ref, 5, sub_rate=1.2e-2, indel_rate=1e-3, big_del_per_mb=100, big_del_len=2500)
    prm = synth.KmcParams(k=k, p=p, counter_size=2, sig_len=7, n_bins=8, both_strands=both)
    kmers, counts = synth.build_sample_db(sample, prm, 9, noise_kmers=500)
    prefix = str(tmp_path / "db")
    synth.write_kmc(prefix, kmers, counts, prm)
    db = oracle.OracleDb(prefix)
    table = dict(zip((int(x) for x in kmers), (int(c) for c in counts)))
    count_of = lambda w: table.get(w, 0)
    buf = padded(ref)
    wins = [(0, 10_000), (9_970, 10_000), (123, 4126), (5, 4127), (777, 31), (778, k), (800, k - 1 if k > 1 else 1),
            (1000, 200), (20_001, 12_345), (39_000, 1000), (0, 40_000)]
    for begin, length in wins:
        for mc in (1, 4):
            got = emulate_window(harness, buf, begin, length, k, both, count_of, mc)
            want = db.process_window(ref[begin:begin + length], mc)
            assert got == want, (begin, length, mc)
    # arbitrary split points give the same answer as one pass (monoid associativity + carries across tiles)
    rng = np.random.default_rng(4)
    begin, length = 3_000, 9_000
    nk = length - k + 1
    want = db.process_window(ref[begin:begin + length], 1)
    for _ in range(12):
        pts = sorted(set(int(x) for x in rng.integers(1, nk, size=int(rng.integers(6, 14)))))
        edges = [0] + pts + [nk]
        cuts = [(a, b - a) for a, b in zip(edges[:-1], edges[1:]) if b - a <= harness.harness_tile_k()]
        if sum(n for _, n in cuts) != nk:
            continue
        assert emulate_window(harness, buf, begin, length, k, both, count_of, 1, cuts=cuts) == want


def test_emulated_kernel_extreme_patterns(harness, oracle):
    """alternating hit/miss, all-miss, all-hit, N-separated runs: via a hand-made count function"""
    k = 31
    rng = np.random.default_rng(8)
    text = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 9000))
    text = text[:2500] + b"N" * 40 + text[2540:6000] + b"n" + text[6001:]
    buf = padded(text)
    lst = np.zeros(len(text), dtype=np.uint64)
    m = oracle.lib().kcfo_kmers_list(text, len(text), k, lst.ctypes.data)
    order = {int(w): i for i, w in enumerate(lst[:m])}       # random text: k-mers are distinct
    assert len(order) == m
    patterns = {
        "alternate": lambda i: i % 2,
        "all_miss": lambda i: 0,
        "all_hit": lambda i: 3,
        "blocks": lambda i: 0 if (i // 97) % 3 == 0 else 2,
        "sparse_hits": lambda i: 1 if i % 641 == 5 else 0,
        "gap29_30_31": lambda i: 0 if (i % 200) < 29 + (i // 200) % 3 else 1,
    }
    for name, fn in patterns.items():
        count_of = lambda w, fn=fn: fn(order[w])
        counts = np.array([fn(i) for i in range(m)], dtype=np.int32)
        want = oracle.accumulate(counts, k, 1)
        want["eff_length"] = oracle.lib().kcfo_eff_length(text, len(text), k)
        got = emulate_window(harness, buf, 0, len(text), k, False, count_of, 1)
        assert got == want, name


def test_minimizer_build_side_equals_query_side(harness, oracle):
    """The table is bucketed by minimizer_of_kmer(stored word); screen_tiles derives the bucket from the reference text.
    They must agree for every valid k-mer, on either strand, or lookups would silently miss."""
    rng = np.random.default_rng(21)
    text = bytes(rng.choice(np.frombuffer(b"ACGTacgt", dtype=np.uint8), 3000))
    text = text[:900] + b"AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA" + b"ATATATATATATATATATATATATATATATATATATATAT" + text[990:]
    buf = padded(text)
    for k, M in ((31, 16), (31, 13), (31, 20), (31, 31), (32, 16), (23, 12), (15, 11), (15, 15), (31, 1)):
        for begin in (0, 5, 33):
            n = min(2048, len(text) - begin - k + 1)
            q = np.zeros(n, dtype=np.uint32)
            bf = np.zeros(n, dtype=np.uint32)
            bc = np.zeros(n, dtype=np.uint32)
            harness.harness_minimizers(buf.ctypes.data, begin, n, k, M, q.ctypes.data, bf.ctypes.data, bc.ctypes.data)
            assert np.array_equal(q, bf), (k, M, begin)
            assert np.array_equal(q, bc), (k, M, begin)
    # consecutive k-mers share their minimizer in runs: mean run length ~ (w + 1) / 2
    n = 2048
    q = np.zeros(n, dtype=np.uint32)
    t1 = np.zeros(n, dtype=np.uint32)
    t2 = np.zeros(n, dtype=np.uint32)
    harness.harness_minimizers(buf.ctypes.data, 1000, n, 31, 16, q.ctypes.data, t1.ctypes.data, t2.ctypes.data)
    runs = 1 + int((q[1:] != q[:-1]).sum())
    assert 6.0 < n / runs < 12.0
    for w in (0x1B00000000000000, 0xFFFFFFFFFFFFFFFC, 0x123456789ABCDEF0 & ~3):
        assert harness.harness_revcomp_word(w, 31) == oracle.lib().kcfo_kmer_revcomp(w, 31)


def test_scalar_gap_loop_equals_lane_parallel_form(harness):
    """gap_close_scalar (what the kernel runs on the scalar unit) against the lane-parallel formulation, on random
    and adversarial ballot masks, with invalid (transparent) positions and carries"""
    rng = np.random.default_rng(99)
    out = np.zeros(4, dtype=np.int32)
    cases = []
    for _ in range(4000):
        dens_v, dens_h = rng.choice([0.05, 0.5, 0.9, 1.0]), rng.choice([0.0, 0.1, 0.5, 0.9, 1.0])
        V = int.from_bytes(np.packbits(rng.random(64) < dens_v).tobytes(), "little")
        H = int.from_bytes(np.packbits(rng.random(64) < dens_h).tobytes(), "little") & V
        cases.append((V, H))
    full = (1 << 64) - 1
    cases += [(full, 0x5555555555555555), (full, 0xAAAAAAAAAAAAAAAA), (full, full), (full, 0), (0, 0), (full, 1 << 63), (full, 1),
              (0x8000000000000001, 0x8000000000000001), (0xFFFF0000FFFF0000, 0x8000000080000000)]
    for V, H in cases:
        for cg in (0, 1, 29, 30, 31, 100):
            for cany in (0, 1):
                harness.harness_gap_both(V, H, cg, cany, 31, out.ctypes.data)
                assert (out[0], out[1]) == (out[2], out[3]), (hex(V), hex(H), cg, cany)


@pytest.mark.parametrize("ext", [0, 1])
def test_cell_table_build_side_covers_query_side(harness, oracle, ext):
    """FORMAT_CELLS (kcfx_internal.h): the cell screen_k31c derives for a reference k-mer -- line from the tagged sliding
    minimum, cell index from the minimizer's offset on the strand where it is canonical, quotiented key (w, r) -- must be one
    of the cells the build stores the canonical k-mer in, for every k-mer, alignment and strand, including repeated
    16-mers (ties), reverse-palindromic 16-mers and low-complexity text. And distinct k-mers never share a full
    (line, cell, w, r): that is what makes the 48-bit compare exact."""
    # ext = 1: the same for an EXTENDED table (databases of >= 0.5 G k-mers: the first-choice line depends on the flank bases next to
    # the minimizer too, cell_first_ext) -- build and query must still meet, the entries must still be injective
    harness.harness_set_ext(ext)
    rng = np.random.default_rng(33)
    acgt = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    pal = b"ACGTACGTACGTACGT"            # its own reverse complement (and so are its even-length prefixes)
    pal2 = b"AAAACCCCGGGGTTTT"
    text = bytes(rng.choice(acgt, 6000))
    text = (text[:500] + b"A" * 80 + b"AT" * 40 + b"ACG" * 30 + text[500:1500] + pal + text[1500:1600] + pal + pal2 +
            text[1600:1700] + pal2 * 3 + text[1700:2500] + text[2000:2300] + b"N" + text[2500:])
    buf = padded(text)
    seen = {}
    # both = 0: a forward-only database stores (and is asked for) the k-mer as the text reads it; bit 14 of r tells it from
    # its reverse complement, so such a table starts at lbits 18
    for k, lbits, both in ((31, 17, 1), (31, 19, 1), (31, 27, 1), (31, 30, 1), (30, 21, 1), (29, 19, 1), (29, 26, 1), (28, 20, 1),
                           (27, 17, 1), (27, 22, 1), (31, 18, 0), (31, 19, 0), (31, 30, 0), (29, 19, 0), (27, 18, 0), (28, 24, 0),
                           (26, 19, 1), (25, 19, 1), (25, 28, 0), (24, 21, 1), (23, 19, 1), (23, 18, 0),
                           (22, 19, 1), (21, 19, 1), (21, 22, 0), (20, 19, 0), (19, 19, 1), (19, 18, 0)):
        for begin in (0, 1, 7, 33, 64, 501):
            n = min(2048, len(text) - begin - k + 1)
            ncand = np.zeros(n, dtype=np.int32)
            probe = np.zeros(4 * n, dtype=np.uint32)
            bad = harness.harness_cells(buf.ctypes.data, begin, n, k, lbits, both, ncand.ctypes.data, probe.ctypes.data)
            assert bad == 0, (k, lbits, both, begin)
            assert ncand.min() >= 1 and ncand.max() <= 32
            if lbits == 19:
                assert (ncand > 1).sum() > (50 if both else 25)          # the tie / palindrome paths were exercised
                assert (ncand == 1).mean() > 0.8
    # injectivity over every cell of every distinct canonical k-mer of the text (valid positions only)
    out = np.zeros(4 * 32, dtype=np.uint32)
    alt3 = np.zeros(3, dtype=np.uint32)
    for k, lbits, both in ((31, 17, 1), (31, 24, 1), (29, 20, 1), (27, 18, 1), (31, 18, 0), (29, 21, 0), (25, 20, 1), (23, 18, 0), (21, 19, 1), (19, 18, 0),
                           (27, 21, 1), (25, 27, 1), (25, 22, 0), (21, 24, 1), (19, 23, 0)):
        lst = np.zeros(len(text), dtype=np.uint64)
        m = oracle.lib().kcfo_kmers_list(text, len(text), k, lst.ctypes.data)
        if both:
            canon = np.unique(np.minimum(lst[:m], synth.revcomp(lst[:m], k)))
        else:                                          # a forward-only database may hold a k-mer AND its reverse complement
            canon = np.unique(np.concatenate([lst[:m], synth.revcomp(lst[:m], k)]))
        seen, seen_alt, seen_2nd = {}, {}, {}
        thirds, seconds = set(), set()
        W = k - 15
        # (free-cell second choices are a build-time option, retired by default: CELL_ALT_FREE_MAX_W = 0)
        free_scheme = k - 15 <= harness.harness_alt_free_max_w() and lbits - (17 if both else 18) >= 4
        rmask = int(harness.harness_cell_rmask(lbits, both))
        q4 = np.zeros(4, dtype=np.uint32)
        for c in canon:
            nc = harness.harness_cells_of_key(int(c), k, lbits, both, out.ctypes.data)
            for i in range(nc):
                cell = tuple(int(x) for x in out[4 * i:4 * i + 4])
                assert seen.setdefault(cell, int(c)) == int(c), (k, lbits, hex(int(c)), hex(seen[cell]))
                line, sh, w, r = cell
                assert sh < 2 * W and r & 0x8000 == 0
                # free-cell second choices (cell_alt_of: the cell, the r field stored there, the r bits compared): what is stored
                # there must give the first-choice cell back, or two k-mers with equal (w, r) at different offsets of one line
                # could not be told apart
                has = harness.harness_cell_alt(sh, w, r, k, lbits, both, alt3.ctypes.data)
                assert bool(has) == free_scheme
                if has:
                    a_sh, a_r, a_mask = (int(x) for x in alt3)
                    assert a_sh % 2 == 0 and 2 * W <= a_sh < 32 and a_r & 0x8000 and a_r & ~a_mask == 0      # a cell that is nobody's first choice
                    alt = (line, a_sh, w, a_r)
                    assert seen_alt.setdefault(alt, int(c)) == int(c), (k, lbits, hex(int(c)), hex(seen_alt[alt]))
                # second-line entries: same cell index and flank, (line, r) from a second bijection of the minimizer, marked bit 15:
                # injective over the k-mers, never equal to a first-choice entry (those have bit 15 clear) and, sitting in a cell
                # < W, never in a free cell
                harness.harness_cell_second_line(line, sh, w, r, lbits, q4.ctypes.data)
                l2, sh2, w2, r2 = (int(x) for x in q4)
                assert (sh2, w2) == (sh, w) and r2 & 0x8000 and r2 & ~rmask == 0 and l2 < (1 << lbits)
                assert both or (r2 & 0x4000) == (r & 0x4000)             # forward-only: the strand mark travels with the entry
                second = (l2, sh2, w2, r2)
                assert seen_2nd.setdefault(second, int(c)) == int(c), (k, lbits, hex(int(c)), hex(seen_2nd[second]))
                # third-line entries (tables with >= 3 spare r bits): line and r from a bijection of the minimizer that the flank's
                # bases next to it select; marked ALT and the top spare bit, so never a first-choice or second-line entry; injective
                has3 = harness.harness_cell_third_line(line, sh, w, r, k, lbits, both, q4.ctypes.data)
                assert bool(has3) == (lbits - (17 if both else 18) >= 3)
                if has3:
                    l3, sh3, w3, r3 = (int(x) for x in q4)
                    third_mark = 0x4000 if both else 0x2000
                    assert (sh3, w3) == (sh, w) and r3 & 0x8000 and r3 & third_mark and r3 & ~rmask == 0 and l3 < (1 << lbits)
                    assert r2 & third_mark == 0 and r & third_mark == 0
                    third = (l3, sh3, w3, r3)
                    assert seen_2nd.setdefault(third, int(c)) == int(c), (k, lbits, hex(int(c)))
                    thirds.add(l3)
                    seconds.add(l2)


@pytest.mark.parametrize("lbits", [17, 22, 29])
def test_region_tables_address_lines_not_identities(harness, lbits):
    """cell_line_index: a table of R regions of 2^lbits lines. The physical line keeps the logical line in its low lbits (what a
    cell compares -- w, r, the second and third lines -- is untouched), the region in front of it is a function of (line, the
    identifying bits of r) alone -- spare, mark and filter bits of r do not move a k-mer -- stays below R, and takes every value about
    equally often, also among the probes of ONE logical line (otherwise a line's tenants would all land in the same region)."""
    rng = np.random.default_rng(lbits)
    n = 200_000
    idm = 0xFFFFFFFF >> lbits
    line = rng.integers(0, 1 << lbits, n, dtype=np.uint32)
    line[: n // 4] = line[0]                                            # many probes of one logical line
    r = rng.integers(0, 1 << 16, n, dtype=np.uint32) if lbits >= 17 else None
    g_low = rng.integers(0, idm + 1, n, dtype=np.uint32)
    r = (r & ~np.uint32(idm & 0xFFFF)) | (g_low & np.uint32(0xFFFF))       # identifying bits + random mark / filter bits above them
    out = np.zeros(n, dtype=np.uint32)
    harness.harness_cell_line_index(line.ctypes.data, r.ctypes.data, n, lbits, 1, out.ctypes.data)
    assert np.array_equal(out, line)
    for R in (3, 5, 7):
        harness.harness_cell_line_index(line.ctypes.data, r.ctypes.data, n, lbits, R, out.ctypes.data)
        assert np.array_equal(out & np.uint32((1 << lbits) - 1), line)
        region = out >> np.uint32(lbits)
        assert region.max() < R
        # the bits of r above the identifying ones do not matter
        out2 = np.zeros(n, dtype=np.uint32)
        r2 = (r & np.uint32(idm & 0xFFFF)) | np.uint32(0x8000 & ~idm)
        harness.harness_cell_line_index(line.ctypes.data, r2.ctypes.data, n, lbits, R, out2.ctypes.data)
        assert np.array_equal(out, out2)
        for sel in (slice(n // 4, None), slice(0, n // 4)):
            if lbits == 29 and sel.start == 0:
                continue                                                # (3 identifying bits: eight tenants per line, nothing to count)
            cnt = np.bincount(region[sel], minlength=R)
            assert cnt.min() > 0.8 * cnt.sum() / R and cnt.max() < 1.2 * cnt.sum() / R, (R, cnt)


@pytest.fixture(autouse=True)
def _plain_tables_by_default(harness):
    yield
    harness.harness_set_ext(0)


def _planner_text(rng):
    acgt = np.frombuffer(b"ACGTacgt", dtype=np.uint8)
    pal = b"ACGTACGTACGTACGT"
    text = bytes(rng.choice(acgt, 9000))
    return (text[:700] + b"A" * 90 + b"AT" * 45 + b"ACG" * 33 + text[700:1500] + pal + text[1500:1600] + pal * 2 + b"N" + text[1600:2500] +
            b"NNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNNN" + text[2500:4000] + b"R" + text[4000:4016] + b"y" + text[4016:6000] + text[3000:3400] + b"\xff" +
            text[6000:])


@pytest.mark.parametrize("L", [32, 16])
def test_strip_planner_matches_the_definition(harness, L):
    """kcfx_plan.h (the code ref_plan_kernel and the text-only screening kernels run, strip by strip with the device's three
    neighbour exchanges emulated): every anchor byte -- valid bit, minimizer offset in N, strand -- against the position-by-
    position definition (cell_order27 over the k-mer's 16-mers, the one cells_visit builds the table with), and the codes against
    pack16. Origins at every alignment a tile can have, every k a cell table exists for, N-runs, IUPAC codes, soft-masking,
    low-complexity runs and palindromes, the end of the text (zero padding) included."""
    rng = np.random.default_rng(77)
    text = _planner_text(rng)
    buf = padded(text)
    readable = len(text) + 128
    cands = np.zeros(16, dtype=np.int32)
    for k in (31, 30, 27, 25, 22, 21, 19):
        for origin in ((0, 16, 48, 1008, 2016) if L == 16 else (0, 32, 2016, 4032)):
            n_strips = 64
            codes = np.zeros(n_strips * L // 16, dtype=np.uint32)
            anchors = np.zeros(n_strips * L, dtype=np.uint8)
            assert harness.harness_plan_strips(buf.ctypes.data, readable, origin, n_strips, k, L, codes.ctypes.data, anchors.ctypes.data) == 0
            good = n_strips - (1 if L == 32 else 2)             # the last strips miss their neighbours, as the last lanes of a wave do
            n_tied = 0
            for j in range(good * L):
                pos = origin + j
                if pos + k > len(text):
                    assert anchors[j] & 0x80 == 0, (k, origin, j)
                    continue
                n = harness.harness_anchor_reference(buf.ctypes.data, pos, k, cands.ctypes.data)
                assert n >= 1
                if cands[0] & 0x80:
                    assert int(anchors[j]) in set(int(c) for c in cands[:n]), (k, L, origin, j, hex(anchors[j]), [hex(int(c)) for c in cands[:n]])
                else:
                    assert anchors[j] & 0x80 == 0, (k, origin, j)
                n_tied += n > 1
            assert n_tied < 0.25 * good * L                     # (the text starts with low-complexity runs: ties are the point there)
            # codes: 16 bases per word, first base in the top bits, A C G T = 0 1 2 3 whatever the case
            code_of = {65: 0, 67: 1, 71: 2, 84: 3, 97: 0, 99: 1, 103: 2, 116: 3}
            for w in range(n_strips * L // 16):
                for i in range(16):
                    p0 = origin + 16 * w + i
                    if p0 < len(text) and text[p0] in code_of:
                        assert (int(codes[w]) >> (30 - 2 * i)) & 3 == code_of[text[p0]]


def test_minimizer_order_keeps_runs_long(harness):
    """The minimizer order (cell_order27: open syncmers first, closed ones second) exists to make consecutive k-mers agree on
    their minimizer for longer: on uniform text a random order gives 2 / 17 = 0.1177 minimizer runs per 31-mer, this one
    0.0965 (numpy simulation in profiles/r03_notes.md, r04_notes.md). Measured here on what the device code itself derives (the query side of
    harness_cells): a run ends where (line, r) changes. A regression of the class function would show as a density near 0.118."""
    rng = np.random.default_rng(2024)
    text = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 40_000))
    buf = padded(text)
    k, lbits = 31, 27
    runs, kmers = 0, 0
    for begin in range(0, len(text) - 2048 - k, 2048):
        n = 2048
        ncand = np.zeros(n, dtype=np.int32)
        probe = np.zeros(4 * n, dtype=np.uint32)
        assert harness.harness_cells(buf.ctypes.data, begin, n, k, lbits, 1, ncand.ctypes.data, probe.ctypes.data) == 0
        p = probe.reshape(n, 4)
        ident = p[:, 0].astype(np.uint64) << np.uint64(16) | (p[:, 3] & 0x3FFF).astype(np.uint64)      # (line, rest of G32)
        runs += 1 + int(np.count_nonzero(ident[1:] != ident[:-1]))
        kmers += n
    density = runs / kmers
    assert 0.090 < density < 0.104, density
