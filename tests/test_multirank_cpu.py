"""World-size-2 (and 3) `gloo` tests of the multi-GPU plumbing: the sharding plans and the final gather of the
per-(window, sample) records. On CPU the per-rank records come from the oracle (this is test infrastructure);
on the GPU box bench.py runs the same functions with the HIP engine and backend "nccl" (= RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from kcftools_amd import shard, synth

K = 31


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def make_inputs(tmpdir, n_samples, write=True):
    """write=True: the parent, ONCE, before the ranks start; a rank passes False and finds the databases written (every rank writing
    the same files while its siblings read them was a race: a rank now and then opened a half-written .kmc_suf and died)"""
    ref = synth.random_reference(60_000, 7, n_run_every=9000, max_n_run=300)
    prefixes = []
    for s in range(n_samples):
        prefix = os.path.join(tmpdir, f"s{s}")
        if write:
            prm = synth.KmcParams(k=K, p=3, counter_size=2, sig_len=5, n_bins=4)
            kmers, counts = synth.build_sample_db(synth.mutate_sample(ref, 100 + s), prm, 200 + s, noise_kmers=500)
            synth.write_kmc(prefix, kmers, counts, prm)
        prefixes.append(prefix)
    starts, ends = synth.tiling_windows(len(ref), 5_000, K)
    return ref, prefixes, starts, ends - starts


def worker(rank, world, port, tmpdir, n_samples, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import binding as oracle
    ref, prefixes, wb, wl = make_inputs(tmpdir, n_samples, write=False)
    if mode == "samples":
        mine = shard.samples_of_rank(n_samples, world, rank)
        recs = [oracle.OracleDb(prefixes[s]).screen(ref, wb, wl, 1) for s in mine]
        local = np.concatenate(recs) if recs else np.zeros(0, dtype=oracle.RESULT_DTYPE)
    elif mode == "cohort":
        # cohort tables x window ranges: a rank's units are (group of up to eight samples, window range); on the GPU one cohort
        # table answers for the whole group, here the oracle does, sample by sample
        recs = []
        for first, n, lo, hi in shard.cohort_segments_of_rank(wl, K, n_samples, world, rank):
            recs += [oracle.OracleDb(prefixes[s]).screen(ref, wb[lo:hi], wl[lo:hi], 1) for s in range(first, first + n)]
        local = np.concatenate(recs) if recs else np.zeros(0, dtype=oracle.RESULT_DTYPE)
    else:
        lo, hi = shard.window_range_of_rank(wl, K, world, rank)
        local = oracle.OracleDb(prefixes[0]).screen(ref, wb[lo:hi], wl[lo:hi], 1)
    t = torch.from_numpy(local.view(np.int32).copy())
    parts = shard.gather_results(t, dst=0)
    if rank == 0:
        q.put([p.numpy().copy() for p in parts])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_samples,mode", [(2, 4, "samples"), (2, 1, "windows"), (3, 1, "windows"), (2, 3, "samples"),
                                                  (2, 3, "cohort"), (3, 10, "cohort")])
def test_sharded_run_equals_single_process(tmp_path, world, n_samples, mode):
    from oracle import binding as oracle
    import queue as _queue
    import time
    ctx = mp.get_context("spawn")
    oracle.build()                                 # (and the oracle's library: the ranks only load it)
    ref, prefixes, wb, wl = make_inputs(str(tmp_path), n_samples)
    parts = None
    for attempt in range(3):                       # retries: the rendezvous port can be taken between probe and bind
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=worker, args=(r, world, port, str(tmp_path), n_samples, mode, q)) for r in range(world)]
        for p in procs:
            p.start()
        deadline = time.time() + 240
        while parts is None and time.time() < deadline:
            try:
                parts = q.get(timeout=2)
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break                            # a rank died: do not wait for the others
        for p in procs:
            p.join(timeout=90 if parts is not None else 1)     # (process teardown on a loaded 8-core box can take a while)
            if p.is_alive():
                p.kill()
                p.join()
        if parts is not None:
            assert all(p.exitcode == 0 for p in procs)
            break
    assert parts is not None, "multi-rank run did not produce a result"
    if mode == "samples":
        for r in range(world):
            got = parts[r].view(oracle.RESULT_DTYPE)
            want = [oracle.OracleDb(prefixes[s]).screen(ref, wb, wl, 1) for s in shard.samples_of_rank(n_samples, world, r)]
            want = np.concatenate(want) if want else np.zeros(0, dtype=oracle.RESULT_DTYPE)
            assert np.array_equal(got, want)
    elif mode == "cohort":
        segs = [shard.cohort_segments_of_rank(wl, K, n_samples, world, r) for r in range(world)]
        got = shard.assemble_cohort_shards([p.view(oracle.RESULT_DTYPE) for p in parts], segs, n_samples, len(wb))
        for s in range(n_samples):
            assert np.array_equal(got[s], oracle.OracleDb(prefixes[s]).screen(ref, wb, wl, 1)), s
        if n_samples > 8:
            assert {f for sg in segs for f, *_ in sg} == {0, 8}          # two tables
    else:
        got = shard.merge_window_shards([p.view(oracle.RESULT_DTYPE) for p in parts])
        want = oracle.OracleDb(prefixes[0]).screen(ref, wb, wl, 1)
        assert np.array_equal(got, want)


def test_window_ranges_partition_and_balance():
    rng = np.random.default_rng(0)
    for n, world in [(1, 2), (7, 8), (100, 3), (3052, 8), (27000, 8), (0, 4)]:
        wl = np.concatenate([rng.integers(1, 200, n // 2), rng.integers(200, 60_000, n - n // 2)]) if n else np.zeros(0, dtype=np.int64)
        ranges = [shard.window_range_of_rank(wl, K, world, r) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
        work = np.maximum(wl - K + 1, 0)
        if n >= 1000:
            per = [int(work[lo:hi].sum()) for lo, hi in ranges]
            assert max(per) <= 1.05 * (sum(per) / world) + int(work.max())
    assert shard.samples_of_rank(8, 8, 3) == [3] and shard.samples_of_rank(32, 8, 7) == [7, 15, 23, 31]
    assert shard.samples_of_rank(3, 8, 5) == []


# ---- table slices: the one data-path exchange (shard.gather_table_partials + merge_table_partials) -------------------------------
def _table_partials(world, n_tiles=53, seed=19):
    """What `world` slices of a table would hold after a pass over n_tiles tiles: the same validity words everywhere, every hit bit
    on exactly one slice (a k-mer's minimizer belongs to one), count sums that add up. Returns (valid, hits, per-rank hit words,
    per-rank sums)."""
    rng = np.random.default_rng(seed)
    W = shard.TILE_WORDS
    valid = rng.integers(0, 1 << 63, (n_tiles, W), dtype=np.int64) | rng.integers(0, 1 << 63, (n_tiles, W), dtype=np.int64)
    hits = valid & rng.integers(0, 1 << 63, (n_tiles, W), dtype=np.int64)
    owner = rng.integers(0, world, (n_tiles, W, 63))                          # which slice holds the minimizer of each position
    bit = (np.int64(1) << np.arange(63, dtype=np.int64))
    per_hits = [((hits[:, :, None] & bit) * (owner == r)).sum(-1).astype(np.int64) for r in range(world)]
    per_sums = [rng.integers(0, 1 << 40, n_tiles, dtype=np.int64) for _ in range(world)]
    return valid, hits, per_hits, per_sums


def table_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    valid, hits, per_hits, per_sums = _table_partials(world)
    bits = torch.from_numpy(np.concatenate([valid, per_hits[rank]], axis=1).reshape(-1).copy())     # per tile: 32 validity, 32 hit words
    sums = torch.from_numpy(per_sums[rank].copy())
    gb, gs = shard.gather_table_partials(bits, sums, dst=0)
    if rank == 0:
        mb, mc = gb[0].clone(), gs[0].clone()
        for b, c in zip(gb[1:], gs[1:]):
            shard.merge_table_partials(mb, mc, b, c)
        q.put((mb.numpy().copy(), mc.numpy().copy()))
    else:
        assert gb is None and gs is None
    dist.barrier()
    dist.destroy_process_group()


def table_rs_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    valid, hits, per_hits, per_sums = _table_partials(world)
    n_tiles = valid.shape[0]
    bits = torch.from_numpy(np.concatenate([valid, per_hits[rank]], axis=1).reshape(-1).copy())
    sums = torch.from_numpy(per_sums[rank].copy())
    # uneven ranges, one of them empty when there are three ranks (a rank may own no window)
    cuts = [0, n_tiles // 3, n_tiles] if world == 2 else [0, n_tiles // 4, n_tiles // 4, n_tiles]
    mb, mc = shard.reduce_scatter_table_partials(bits, sums, cuts)
    q.put((rank, cuts, mb.numpy().copy(), mc.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_table_slices_meet_range_by_range(world):
    """Round 6, the other form of the exchange: rank q owns a range of the tiles (the tiles of a window range) and receives that range
    from every rank -- 1 / W of the bitmaps each -- instead of rank 0 receiving everything. Every rank ends up with the merged words
    of its own range: validity as it was, hits OR-ed over the slices, count sums added; the ranges tile the list, one may be empty."""
    import queue as _queue
    import time
    ctx = mp.get_context("spawn")
    got = {}
    for attempt in range(3):
        got = {}
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=table_rs_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        deadline = time.time() + 240
        while len(got) < world and time.time() < deadline:
            try:
                r, cuts, mb, mc = q.get(timeout=2)
                got[r] = (cuts, mb, mc)
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
        for p in procs:
            p.join(timeout=90 if len(got) == world else 1)
            if p.is_alive():
                p.kill()
                p.join()
        if len(got) == world:
            assert all(p.exitcode == 0 for p in procs)
            break
    assert len(got) == world, "multi-rank run did not produce a result"
    valid, hits, per_hits, per_sums = _table_partials(world)
    W = shard.TILE_WORDS
    for r in range(world):
        cuts, mb, mc = got[r]
        lo, hi = cuts[r], cuts[r + 1]
        mb = mb.reshape(-1, 2 * W)
        assert mb.shape[0] == hi - lo == len(mc)
        assert np.array_equal(mb[:, :W], valid[lo:hi]) and np.array_equal(mb[:, W:], hits[lo:hi])
        assert np.array_equal(mc, np.sum(per_sums, axis=0)[lo:hi])


@pytest.mark.parametrize("world", [2, 3])
def test_table_slices_meet_on_rank_zero(world):
    """bench.py --shard table / a job with one process per GPU: every rank screens the whole window list against ITS slice of the
    table, the tiles' bitmaps and count sums are gathered to rank 0 (RCCL on GPUs, gloo here) and merged -- hits OR-ed, sums added,
    validity words as they were -- before the one reduce_windows. (On the GPU the merge is kcfx_ref_merge_partial_async;
    tests/test_gpu_sharded.py checks that kernel against this torch model.)"""
    import queue as _queue
    import time
    ctx = mp.get_context("spawn")
    got = None
    for attempt in range(3):
        q = ctx.Queue()
        port = free_port()
        procs = [ctx.Process(target=table_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        deadline = time.time() + 240
        while got is None and time.time() < deadline:
            try:
                got = q.get(timeout=2)
            except _queue.Empty:
                if any(p.exitcode not in (None, 0) for p in procs):
                    break
        for p in procs:
            p.join(timeout=90 if got is not None else 1)
            if p.is_alive():
                p.kill()
                p.join()
        if got is not None:
            assert all(p.exitcode == 0 for p in procs)
            break
    assert got is not None, "multi-rank run did not produce a result"
    valid, hits, per_hits, per_sums = _table_partials(world)
    bits, sums = got
    bits = bits.reshape(-1, 2 * shard.TILE_WORDS)
    assert np.array_equal(bits[:, :shard.TILE_WORDS], valid) and np.array_equal(bits[:, shard.TILE_WORDS:], hits)
    assert np.array_equal(sums, np.sum(per_sums, axis=0))
