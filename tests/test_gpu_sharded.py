"""kcfx_ctx_create + kcfx_screen_sharded (the multi-GPU entry of the C ABI) on ONE GPU: the same device listed two or
three times gives two or three worker threads, each with its own resident reference -- the control flow of a node with
several GPUs. Every slice must equal the unsharded kcfx_screen_ref and the oracle, in both partitionings."""
import os

import numpy as np
import pytest

from kcftools_amd import engine, synth
from test_gpu_parity import assert_same, make_db

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def job(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("sharded")
    k = 31
    ref = synth.random_reference(300_000, 71, n_run_every=40_000, max_n_run=700)
    specs = [dict(cs=2, both=True), dict(cs=1, both=True), dict(cs=4, both=True), dict(cs=2, both=False), dict(cs=2, both=True)]
    prefixes = [make_db(tmp, f"s{i}", ref, 900 + i, k=k, p=3, cs=sp["cs"], m=7, bins=4, both=sp["both"], noise=4_000)[0]
                for i, sp in enumerate(specs)]
    rng = np.random.default_rng(5)
    # tiling windows + gene-like irregular ones (short, empty, overlapping), unsorted
    s, e = engine.windows(len(ref), 10_000, 0, k)
    lens = np.concatenate([e - s, rng.integers(0, 9_000, 40), [0, 30, 31]])
    begins = np.concatenate([s, rng.integers(0, len(ref) - 9_000, 40), [5, 77, 123]])
    want = np.stack([engine.KmcDb.open(p).screen(ref, begins, lens, 1) for p in prefixes])
    return k, ref, prefixes, begins, lens, want


@pytest.mark.parametrize("devices,mode", [((0, 0), engine.SHARD_SAMPLES), ((0, 0), engine.SHARD_WINDOWS), ((0, 0, 0), engine.SHARD_WINDOWS),
                                          ((0,), engine.SHARD_AUTO), ((0, 0), engine.SHARD_AUTO), ((0,) * 7, engine.SHARD_AUTO),
                                          ((0, 0, 0), engine.SHARD_GRID), ((0,) * 8, engine.SHARD_GRID), ((0,) * 11, engine.SHARD_GRID)])
def test_sharded_equals_unsharded(oracle, job, devices, mode):
    k, ref, prefixes, begins, lens, want = job
    got, st = engine.screen_sharded(prefixes, k, ref, begins, lens, 1, devices=devices, mode=mode)
    assert got.shape == want.shape
    for i in range(len(prefixes)):
        assert_same(got[i], want[i], f"sample {i}, devices {devices}, mode {mode}")
    # five samples: on 7 devices the automatic choice is the samples x windows grid (a device opens two databases, not five)
    auto = engine.SHARD_SAMPLES if len(prefixes) >= len(devices) else engine.SHARD_GRID
    assert st.shard_mode_used == (mode or auto) and st.n_devices == len(devices)
    assert st.seconds_total > 0 and st.seconds_screen_sum > 0 and st.seconds_db_open_sum > 0
    # one sample against the oracle, to anchor the comparison
    assert_same(got[0], oracle.OracleDb(prefixes[0]).screen(ref, begins, lens, 1, threads=4), "vs oracle")


def test_sharded_edge_cases_and_errors(job, tmp_path):
    k, ref, prefixes, begins, lens, want = job
    # one sample on several devices: the automatic choice is the window-sharded mode
    _, st1 = engine.screen_sharded(prefixes[:1], k, ref, begins, lens, 1, devices=(0, 0, 0))
    assert st1.shard_mode_used == engine.SHARD_WINDOWS
    # the grid with fewer window-samples than devices, and with windows that hold no k-mer at a cut
    got, _ = engine.screen_sharded(prefixes[:3], k, ref, begins[-3:], lens[-3:], 1, devices=(0,) * 12, mode=engine.SHARD_GRID)
    for i in range(3):
        assert_same(got[i], want[i][-3:], "grid: 3 short windows x 3 samples on 12 shards")
    # fewer windows than devices, -c 2, no samples, no windows
    got, _ = engine.screen_sharded(prefixes[:1], k, ref, begins[:2], lens[:2], 1, devices=(0, 0, 0, 0), mode=engine.SHARD_WINDOWS)
    assert_same(got[0], want[0][:2], "2 windows on 4 shards")
    got2, _ = engine.screen_sharded(prefixes[:2], k, ref, begins, lens, 2, devices=(0, 0))
    for i in range(2):
        assert_same(got2[i], engine.KmcDb.open(prefixes[i]).screen(ref, begins, lens, 2), "-c 2")
    assert engine.screen_sharded([], k, ref, begins, lens, 1, devices=(0, 0))[0].shape == (0, len(begins))
    assert engine.screen_sharded(prefixes[:1], k, ref, [], [], 1, devices=(0, 0))[0].shape == (1, 0)
    # a missing database, a k mismatch, a bad window, a bad threshold: a status and a message, never an exit
    with pytest.raises(engine.KcfxError) as ei:
        engine.screen_sharded([prefixes[0], str(tmp_path / "missing"), prefixes[1]], k, ref, begins, lens, 1, devices=(0, 0))
    assert ei.value.code == 2 and "device 0" in str(ei.value)
    with pytest.raises(engine.KcfxError) as ei:
        engine.screen_sharded(prefixes[:1], 23, ref, begins, lens, 1, devices=(0,))
    assert "Kmer size mismatch" in str(ei.value)
    with pytest.raises(engine.KcfxError):
        engine.screen_sharded(prefixes[:1], k, ref, [len(ref) - 5], [50], 1, devices=(0,))
    with pytest.raises(engine.KcfxError):
        engine.screen_sharded(prefixes[:1], k, ref, begins, lens, 0, devices=(0,))
    with pytest.raises(engine.KcfxError):
        engine.screen_sharded(prefixes[:1], k, ref, begins, lens, 1, devices=(0, 99))


def test_closed_tables_are_kept_for_the_next_database_and_released_on_request(oracle, tmp_path):
    """kcfx_pool.cpp: the table of a closed database stays with the library for the next database of the same size (a run over
    many samples; hipMalloc after hipFree of 16 GiB stalls for seconds on this platform) and goes back to the driver with
    kcfx_release_cached -- or by itself when an allocation would fail otherwise."""
    import gc
    import torch
    ref = synth.random_reference(300_000, 3)
    sample = synth.mutate_sample(ref, 4)
    prm = synth.KmcParams(k=31, p=3, counter_size=2, sig_len=7, n_bins=4, max_count=65535)
    kmers, counts = synth.build_sample_db(sample, prm, 5, noise_kmers=1000)
    prefix = str(tmp_path / "db")
    synth.write_kmc(prefix, kmers, counts, prm)
    os.environ["KCFX_TABLE_LBITS"] = "21"            # a 256 MiB table: large enough to be kept
    try:
        engine.release_cached()
        free0 = torch.cuda.mem_get_info()[0]
        s, e = engine.windows(len(ref), 10_000, 0, 31)
        want = oracle.OracleDb(prefix).screen(ref, s, e - s, 1)
        for _ in range(3):                           # the same block serves all three
            db = engine.KmcDb.open(prefix)
            assert db.info.table_bytes >= 256 << 20
            got = db.screen(ref, s, e - s, 1)
            assert all(np.array_equal(got[f], want[f]) for f in engine.INT_FIELDS)
            del db
            gc.collect()
        kept = free0 - torch.cuda.mem_get_info()[0]
        assert kept >= 256 << 20, kept
        engine.release_cached()
        assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20
    finally:
        del os.environ["KCFX_TABLE_LBITS"]


def test_kept_blocks_and_another_allocator_in_the_process(tmp_path):
    """include/kcfx.h, kcfx_release_cached: the blocks the library keeps for the next database are invisible to another
    allocator in the process (PyTorch here). Under memory pressure that allocator fails while the blocks sit in the pool, and
    succeeds once kcfx_release_cached has given them back -- the documented contract, exercised."""
    import gc
    import torch
    ref = synth.random_reference(200_000, 13)
    prm = synth.KmcParams(k=31, p=3, counter_size=2, sig_len=7, n_bins=4, max_count=65535)
    kmers, counts = synth.build_sample_db(synth.mutate_sample(ref, 14), prm, 15, noise_kmers=1000)
    prefix = str(tmp_path / "db")
    synth.write_kmc(prefix, kmers, counts, prm)
    os.environ["KCFX_TABLE_LBITS"] = "26"            # an 8 GiB table
    try:
        engine.release_cached()
        torch.cuda.empty_cache()
        db = engine.KmcDb.open(prefix)
        assert db.info.table_bytes >= 8 << 30
        del db
        gc.collect()                                 # closed: the 8 GiB block is kept by the library
        free_now = torch.cuda.mem_get_info()[0]
        want = free_now + (4 << 30)                  # more than the driver has, less than driver + pool
        with pytest.raises(torch.OutOfMemoryError):
            torch.empty(want, dtype=torch.uint8, device="cuda:0")
        engine.release_cached()
        t = torch.empty(want, dtype=torch.uint8, device="cuda:0")
        assert t.numel() == want
        del t
        torch.cuda.empty_cache()
    finally:
        del os.environ["KCFX_TABLE_LBITS"]


@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)])
def test_cohort_tables_across_devices(oracle, tmp_path_factory, devices):
    """KCFX_SHARD_COHORT: the samples in cohort tables of up to eight, tables x window ranges over the devices -- eleven samples =
    one table of eight + one of three; on (0, 0) each worker builds one, on (0, 0, 0) a worker builds two. Every (window, sample)
    result must equal the single-sample path and the oracle. A group that cannot have a cohort table (a 4-byte-counter sample in
    it) is screened sample by sample inside the same call."""
    tmp = tmp_path_factory.mktemp("cohort_sharded")
    k = 31
    ref = synth.random_reference(260_000, 81, n_run_every=50_000, max_n_run=600)
    prefixes = [make_db(tmp, f"c{i}", ref, 700 + i, k=k, p=3, cs=2 if i % 3 else 1, m=7, bins=4, both=True, noise=3_000)[0] for i in range(11)]
    s, e = engine.windows(len(ref), 10_000, 0, k)
    rng = np.random.default_rng(6)
    lens = np.concatenate([e - s, rng.integers(0, 7_000, 25), [0, 30, 31]])
    begins = np.concatenate([s, rng.integers(0, len(ref) - 7_000, 25), [5, 77, 123]])
    want = np.stack([engine.KmcDb.open(p).screen(ref, begins, lens, 1) for p in prefixes])
    got, st = engine.screen_sharded(prefixes, k, ref, begins, lens, 1, devices=devices, mode=engine.SHARD_COHORT)
    assert st.shard_mode_used == engine.SHARD_COHORT and got.shape == want.shape
    for i in range(len(prefixes)):
        assert_same(got[i], want[i], f"cohort-sharded, sample {i}, devices {devices}")
    assert_same(got[9], oracle.OracleDb(prefixes[9]).screen(ref, begins, lens, 1, threads=4), "vs oracle (a sample of the second table)")
    if devices == (0, 0):
        # -c 3, and a group with a 4-byte-counter sample: no cohort table for it, same results
        wide = make_db(tmp, "wide", ref, 777, k=k, p=3, cs=4, m=7, bins=4, both=True, noise=3_000)[0]
        mixed = prefixes[:3] + [wide]
        got3, st3 = engine.screen_sharded(mixed, k, ref, begins, lens, 3, devices=devices, mode=engine.SHARD_COHORT)
        assert st3.shard_mode_used == engine.SHARD_COHORT
        for i, p in enumerate(mixed):
            assert_same(got3[i], engine.KmcDb.open(p).screen(ref, begins, lens, 3), f"mixed counter widths, sample {i}")


@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0), (0,) * 5])
def test_table_partitioned_over_devices(oracle, job, devices):
    """KCFX_SHARD_TABLE: every sample's table in len(devices) slices by minimizer (here: on one GPU), every slice screened against
    the whole window list, the tiles' bitmaps merged and reduced once == the unsharded call == the oracle. Canonical and
    forward-only databases, 1- / 2- / 4-byte counters (wide counters live in the slices' stashes), gene-like windows."""
    k, ref, prefixes, begins, lens, want = job
    got, st = engine.screen_sharded(prefixes, k, ref, begins, lens, 1, devices=devices, mode=engine.SHARD_TABLE)
    assert st.shard_mode_used == engine.SHARD_TABLE and st.n_devices == len(devices)
    for i in range(len(prefixes)):
        assert_same(got[i], want[i], f"sample {i}, table slices on {devices}")
    assert_same(got[3], oracle.OracleDb(prefixes[3]).screen(ref, begins, lens, 1, threads=4), "forward-only sample vs oracle")
    got2, _ = engine.screen_sharded(prefixes[:2], k, ref, begins, lens, 3, devices=devices, mode=engine.SHARD_TABLE)
    for i in range(2):
        assert_same(got2[i], engine.KmcDb.open(prefixes[i]).screen(ref, begins, lens, 3), "-c 3")


@pytest.mark.parametrize("k,both,cs", [(31, True, 2), (31, False, 2), (25, True, 1), (19, False, 4)])
def test_a_slice_walks_its_own_positions(oracle, tmp_path, monkeypatch, k, both, cs):
    """Round 6: which slice holds a k-mer starts from two hash bits of its minimizer's order key (the slice class), and the anchor
    byte of every planned reference position carries them -- a planned reference's pass against a slice lists the positions whose
    class the slice can own and walks the list (screen_k31c_slice_kernel) where round 5 walked every position with the other
    slices' lanes idle (KCFX_SLICE_WALK_ALL, and what a text-only reference still does). The three forms give the same tile
    bitmaps and count sums for every slice, bit for bit; the slices merged == the whole table == the oracle. 2 / 4 slices (the
    class is the slice), 8 (two slices per class), 3 and 5 (classes shared unevenly); plain, extended and three-region tables;
    the build with and without the probes it keeps between its passes (the class rides in their flag byte)."""
    import torch
    L = 260_000
    ref = synth.random_reference(L, 5 + k, n_run_every=50_000, max_n_run=500, lower_frac=0.1, iupac_rate=2e-4)
    prefix, *_ = make_db(tmp_path, "own", ref, 31 + k, k=k, p={31: 3, 25: 5, 19: 3}[k], cs=cs, m=7, bins=4, both=both, noise=30_000,
                         sub_rate=5e-3, indel_rate=5e-4)
    s, e = engine.windows(len(ref), 7_000, 0, k)
    rng = np.random.default_rng(k)
    lens = np.concatenate([e - s, rng.integers(0, 5_000, 30), [0, k - 1, k, 2048 + k - 1]]).astype(np.int32)
    begins = np.concatenate([s, rng.integers(0, L - 5_000, 30), [3, 9, L - k, 1000]]).astype(np.int64)
    want = oracle.OracleDb(prefix).screen(ref, begins, lens, 1, threads=4)
    st = torch.cuda.current_stream().cuda_stream
    planned = engine.Reference(ref, begins, lens, k)
    text_only = engine.Reference(ref, begins, lens, k, expected_screens=1)
    assert planned.planned and not text_only.planned
    nb, ns = planned.partial_bytes()

    def export(r):
        b = torch.zeros(nb // 8, dtype=torch.int64, device="cuda:0")
        c = torch.zeros(ns // 8, dtype=torch.int64, device="cuda:0")
        r.export_partial_async(b.data_ptr(), c.data_ptr(), st)
        torch.cuda.synchronize()
        return b, c

    for variant, n_list in (({}, (2, 4, 8, 3)), ({"KCFX_TABLE_EXT": "1", "KCFX_TABLE_REGIONS": "3"}, (4, 5)), ({"KCFX_LOAD_NO_PROBES": "1"}, (8,))):
        for name, v in variant.items():
            monkeypatch.setenv(name, v)
        for n in n_list:
            total = None
            sizes = []
            for i in range(n):
                db = engine.KmcDb.open(prefix, n_slices=n, slice=i)
                sizes.append(int(db.info.stored_kmers))
                planned.screen_partial(db, 1)
                listed = export(planned)
                monkeypatch.setenv("KCFX_SLICE_WALK_ALL", "1")
                planned.screen_partial(db, 1)
                walked = export(planned)
                monkeypatch.delenv("KCFX_SLICE_WALK_ALL")
                assert torch.equal(listed[0], walked[0]) and torch.equal(listed[1], walked[1]), (variant, n, i)
                text_only.screen_partial(db, 2)
                t2 = text_only.finish()
                planned.screen_partial(db, 2)                                       # (-c 2: the counters' side of the match)
                assert_same(planned.finish(), t2, f"slice {i} of {n} alone, -c 2: planned (listed) vs text-only")
                planned.import_partial_async(listed[0].data_ptr(), listed[1].data_ptr(), st)
                if total is not None:
                    planned.merge_partial_async(total[0].data_ptr(), total[1].data_ptr(), st)
                total = export(planned)
                db.close()
            assert_same(planned.finish(), want, f"{n} slices merged vs oracle ({variant})")
            assert max(sizes) < (1.0 / n + 0.08) * sum(sizes) + 2_000, (n, sizes)     # the classes are even: no slice much over its share
        for name in variant:
            monkeypatch.delenv(name)


def test_slice_loaders_share_one_file_stream(oracle, job, monkeypatch):
    """Round 6: the G loaders of a table-sharded sample read the .kmc_suf ONCE between them (a cache of pinned chunks: whoever asks
    first reads, the siblings copy from the same buffer). With the file streamed once per build pass and chunks of 4 099 records
    (a few hundred chunks per file) four slices on the one GPU: the same records as the unsharded call, most chunk requests served
    by a sibling's read, and far fewer bytes read than four loaders of their own would have."""
    k, ref, prefixes, begins, lens, want = job
    monkeypatch.setenv("KCFX_LOAD_STREAM_TWICE", "1")          # (the tier of a database that does not fit: the file once per pass)
    monkeypatch.setenv("KCFX_LOAD_CHUNK_RECORDS", "4099")
    devices = [0, 0, 0, 0]
    got, st = engine.screen_sharded(prefixes[:1], k, ref, begins, lens, 1, devices=devices, mode=engine.SHARD_TABLE)
    assert st.shard_mode_used == engine.SHARD_TABLE
    assert_same(got[0], want[0], "four slice loaders on one file stream")
    nbytes, reads, shared = engine.shard_slice_io()
    suf = os.path.getsize(prefixes[0] + ".kmc_suf")
    assert reads > 0 and shared > reads, (nbytes, reads, shared)      # most requests found their chunk already read
    passes = nbytes / suf
    assert 3.5 <= passes <= 9.0, passes                         # ~4 - 5 passes of ONE stream (count, claim, write, settle [, stash]); 4 loaders alone: 16 - 20
    # one device: no cache at all
    got1, _ = engine.screen_sharded(prefixes[:1], k, ref, begins, lens, 1, devices=[0], mode=engine.SHARD_TABLE)
    assert_same(got1[0], want[0], "one slice")
    assert engine.shard_slice_io() == (0, 0, 0)


def test_table_slices_piece_by_piece(oracle, job, tmp_path):
    """The entry points behind KCFX_SHARD_TABLE one by one: kcfx_db_open_slice (the slices hold every stored k-mer exactly once, up
    to the few with several first-choice cells; a look-up finds a k-mer on the slice that holds it), kcfx_screen_ref_partial on a
    planned and on a text-only reference, merge through another reference handle and through exported device buffers (what an RCCL
    gather carries; the torch model of the merge beside the kernel), kcfx_ref_finish. Whole-table entry points refuse a slice."""
    import torch
    from kcftools_amd import shard
    k, ref, prefixes, begins, lens, want = job
    for pi in (0, 2, 3):                                       # 2-byte canonical, 4-byte counters, forward-only
        prefix = prefixes[pi]
        whole = engine.KmcDb.open(prefix)
        n = 3
        slices = [engine.KmcDb.open(prefix, n_slices=n, slice=i) for i in range(n)]
        infos = [d.info for d in slices]
        assert all(i.table_slices == n and i.table_slice == j and i.table_format == 1 for j, i in enumerate(infos))
        stored = sum(int(i.stored_kmers) for i in infos)
        assert whole.info.stored_kmers <= stored <= whole.info.stored_kmers + 2 * whole.info.multi_cell_kmers + 4
        assert max(int(i.stored_kmers) for i in infos) < 0.45 * whole.info.stored_kmers      # thirds, roughly
        assert all(i.table_lines < whole.info.table_lines or whole.info.table_lines <= (1 << 18) for i in infos)
        # KMC.getCount through the slices: a present k-mer is found on (at least) one, an absent one nowhere
        if pi == 0:
            lp, kmers, counts, _ = make_db(tmp_path, "lookup", ref, 900, k=k, p=3, cs=2, m=7, bins=4, both=True, noise=4_000)   # (sample 0 again: its k-mers)
            absent = np.random.default_rng(4).integers(0, 1 << 62, 5_000, dtype=np.uint64) << np.uint64(2)
            absent = np.minimum(absent, synth.revcomp(absent, k))
            keys = np.concatenate([kmers[np.random.default_rng(3).choice(len(kmers), min(20_000, len(kmers)), replace=False)], absent])
            full = whole.lookup(keys)
            per = np.stack([d.lookup(keys) for d in slices])
            assert np.array_equal(per.max(0), full)                                # found where it is held, with its counter
            assert np.all((per != 0).sum(0) >= (full != 0)) and np.all((per != 0).sum(0)[full == 0] == 0)
            assert ((per != 0).sum(0) > 1).mean() < 1e-3                           # several slices only for tied minimizers
        for planned in (True, False):
            refs = [engine.Reference(ref, begins, lens, k, expected_screens=0 if planned else 1) for _ in range(n)]
            assert refs[0].planned == planned
            for r, d in zip(refs, slices):
                r.screen_partial(d, 1)
            # one slice alone is a subset of the hits
            alone = refs[1].finish()
            assert np.all(alone["obs"] <= want[pi]["obs"]) and np.array_equal(alone["total_kmers"], want[pi]["total_kmers"])
            # (a) through exported device buffers, merged by the kernel and by the torch model
            nb, ns = refs[0].partial_bytes()
            bufs = []
            for r in refs:
                b = torch.zeros(nb // 8, dtype=torch.int64, device="cuda:0")
                c = torch.zeros(ns // 8, dtype=torch.int64, device="cuda:0")
                r.export_partial_async(b.data_ptr(), c.data_ptr(), torch.cuda.current_stream().cuda_stream)
                bufs.append((b, c))
            torch.cuda.synchronize()
            mb, mc = bufs[0][0].clone(), bufs[0][1].clone()
            for b, c in bufs[1:]:
                shard.merge_table_partials(mb, mc, b, c)
            # (b) in place, through the other handles
            refs[0].merge_from(refs[1])
            refs[0].merge_from(refs[2])
            assert_same(refs[0].finish(), want[pi], f"slices merged handle to handle (sample {pi}, planned={planned})")
            eb = torch.zeros_like(mb)
            ec = torch.zeros_like(mc)
            refs[0].export_partial_async(eb.data_ptr(), ec.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert torch.equal(eb, mb) and torch.equal(ec, mc)               # kernel merge == torch model
            # (c) the exported buffers of slices 0 and 2 into slice 1's reference (any order)
            refs[1].merge_partial_async(bufs[2][0].data_ptr(), bufs[2][1].data_ptr(), torch.cuda.current_stream().cuda_stream)
            refs[1].merge_partial_async(bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            assert_same(refs[1].finish(), want[pi], "slices merged from exported buffers")
            # (d) round 6, range by range: "rank" q owns a window range, takes that range's tiles of every slice's export (what the
            # W gathers of shard.reduce_scatter_table_partials deliver), imports its own into a reference of JUST those windows,
            # merges the others and finishes: its windows' records, no buffer of the whole list anywhere
            st = torch.cuda.current_stream().cuda_stream
            for q in range(n):
                lo, hi = shard.window_range_of_rank(lens, k, n, q)
                t_lo, t_hi = refs[0].tile_range(lo, hi)
                own = engine.Reference(ref, begins[lo:hi], lens[lo:hi], k, expected_screens=0 if planned else 1)    # (the same FORM: a text-only reference cuts its tiles at 2 001 k-mers, a planned one at 2 048)
                assert own.tile_range(0, hi - lo) == (0, t_hi - t_lo)
                parts = [(b[t_lo * 64:t_hi * 64].contiguous(), c[t_lo:t_hi].contiguous()) for b, c in bufs]
                own.import_partial_async(parts[q][0].data_ptr(), parts[q][1].data_ptr(), st)
                for r_ in range(n):
                    if r_ != q:
                        own.merge_partial_async(parts[r_][0].data_ptr(), parts[r_][1].data_ptr(), st)
                torch.cuda.synchronize()
                if hi > lo:
                    assert_same(own.finish(), want[pi][lo:hi], f"range {q} of {n} merged range by range")
                own.close()
        # a slice is not a database: the whole-table entry points say so
        with pytest.raises(engine.KcfxError) as ei:
            refs[0].screen(slices[0], 1)
        assert ei.value.code == 1 and "slice" in str(ei.value)
        with pytest.raises(engine.KcfxError):
            slices[0].screen(ref, begins, lens, 1)
    with pytest.raises(engine.KcfxError):
        engine.KmcDb.open(prefixes[0], n_slices=3, slice=3)
    with pytest.raises(engine.KcfxError):
        engine.KmcDb.open(prefixes[0], n_slices=65, slice=0)
