"""Multi-GPU plumbing for the getVariations hot path: one process per GPU, torch.distributed (RCCL on GPUs,
gloo in the CPU tests).

The path shards into independent units -- every (window, sample) result depends only on that window's text and
that sample's database (GetVariants.java:138-150 processes windows independently; utils/run_kcftools.sh:124-140
runs samples in separate processes) -- so there is NO data-path collective. Two partitionings (SURVEY.md 8 e):

  * samples >= ranks : sample-sharded. Rank r owns samples r, r + W, r + 2W, ...; the reference is replicated.
  * samples <  ranks : window-sharded. Contiguous window ranges balanced by the number of k-mer start positions;
                       the sample database is replicated.

  * cohort tables   : (on request) the samples in groups of eight, each group ONE cohort table (kcfx_cohort_open); the
                       group-major sequence of (group, window) units is cut into equal contiguous parts, one per rank --
                       8 samples on 8 ranks: every rank the same table and an eighth of the windows; 32 samples: four tables,
                       two ranks each. The plan is kcfx_shard_plan's (KCFX_SHARD_COHORT), host arithmetic inside libkcfx.

  * table slices    : (on request, round 5) a database too large for one device at a density worth having: its TABLE is cut
                       into one slice per rank by minimizer (kcfx_db_open_slice), every rank screens the whole window list
                       against its slice, and the tiles' hit bitmaps (512 B per tile) + count sums meet -- the one data-path
                       exchange of this engine -- either on rank 0 (gather_table_partials: a gather of the partial results, OR /
                       add, then ONE reduce_windows) or, round 6, RANGE BY RANGE (reduce_scatter_table_partials: rank q owns a
                       window range and receives only that range's tiles from everybody, 1 / W of the bitmaps each -- no rank
                       carries (W - 1) x the whole; then the ordinary final gather of 40-byte records).

Otherwise the only exchange is the final gather of the 40-byte per-(window, sample) records to rank 0.
"""
from __future__ import annotations

import numpy as np

RESULT_WORDS = 10   # kcfx_result = 10 x int32 (the int64 count sum occupies the last two)


def samples_of_rank(n_samples: int, world: int, rank: int) -> list[int]:
    return list(range(rank, n_samples, world))


def window_range_of_rank(win_len: np.ndarray, k: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous [lo, hi) window range of `rank`, balanced by k-mer start positions (work), not by window count:
    gene-model windows span 31 bp .. 100 kbp."""
    work = np.maximum(np.asarray(win_len, dtype=np.int64) - k + 1, 0)
    total = int(work.sum())
    n = len(work)
    if world <= 1 or n == 0:
        return 0, n
    cum = np.cumsum(work)
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(cum, target, side="left")) + (1 if total else 0))
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts[rank], cuts[rank + 1]


def gather_results(local, dst: int = 0, group=None):
    """Gather every rank's result records (torch int32 tensor [n_local * 10], device or host) to `dst`.
    Ranks may hold different numbers of records (window sharding): sizes are exchanged first, the payload is
    padded to the maximum, and `dst` trims. Returns a list of per-rank tensors on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes) if sizes else 0
    padded = local
    if local.numel() < m:
        padded = torch.zeros(m, dtype=local.dtype, device=local.device)
        padded[: local.numel()] = local
    out = [torch.zeros(m, dtype=local.dtype, device=local.device) for _ in range(world)] if rank == dst else None
    dist.gather(padded, out, dst=dst, group=group)
    if rank != dst:
        return None
    return [o[:s] for o, s in zip(out, sizes)]


def merge_window_shards(parts: list[np.ndarray]) -> np.ndarray:
    """Window-sharded run: concatenate the per-rank record arrays in rank order (ranges are contiguous)."""
    return np.concatenate(parts) if parts else np.zeros(0)


def cohort_segments_of_rank(win_len: np.ndarray, k: int, n_samples: int, world: int, rank: int) -> list[tuple[int, int, int, int]]:
    """The (first sample, samples in the group, w_lo, w_hi) units rank `rank` screens when the samples are held in cohort
    tables (kcfx_shard_plan, KCFX_SHARD_COHORT: the partitioning kcfx_screen_sharded uses inside one process)."""
    from . import engine
    plan, _ = engine.shard_plan(win_len, k, n_samples, world, engine.SHARD_COHORT)
    return [(s, min(engine.COHORT_MAX, n_samples - s), lo, hi) for dev, s, lo, hi in plan if dev == rank]


def assemble_cohort_shards(parts: list[np.ndarray], segments: list[list[tuple[int, int, int, int]]], n_samples: int,
                           n_windows: int) -> np.ndarray:
    """Rank r's records = its segments in order, each [samples of the group, windows of the range] row-major: back into one
    [n_samples, n_windows] array."""
    out = np.zeros((n_samples, n_windows), dtype=parts[0].dtype if parts else np.int32)
    for part, segs in zip(parts, segments):
        at = 0
        for first, n, lo, hi in segs:
            m = n * (hi - lo)
            out[first:first + n, lo:hi] = part[at:at + m].reshape(n, hi - lo)
            at += m
        assert at == len(part)
    return out


# ---- table slices: the partial results of a pass (kcfx_ref_export_partial_async) ------------------------------------------------
TILE_WORDS = 32               # per tile: 32 validity words, then 32 hit words (kcfx_internal.h: TILE_BITS_U64 = 64)


def gather_table_partials(bits, sums, dst: int = 0, group=None):
    """Every rank's tile bitmaps (int64 tensor [n_tiles * 64]) and count sums (int64 [n_tiles]) to `dst`: lists of per-rank
    tensors there, (None, None) elsewhere. Every rank screens the same tiles, so the sizes agree."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    gb = [torch.zeros_like(bits) for _ in range(world)] if rank == dst else None
    gs = [torch.zeros_like(sums) for _ in range(world)] if rank == dst else None
    dist.gather(bits, gb, dst=dst, group=group)
    dist.gather(sums, gs, dst=dst, group=group)
    return gb, gs


def merge_table_partials(bits, sums, other_bits, other_sums):
    """What kcfx_ref_merge_partial_async does on the device, in torch (the CPU tests' model and a fallback-free reference for the
    GPU test): the hit words of `other` OR-ed into `bits`, its count sums added; validity words untouched. In place."""
    b = bits.view(-1, 2 * TILE_WORDS)
    b[:, TILE_WORDS:] |= other_bits.view(-1, 2 * TILE_WORDS)[:, TILE_WORDS:]
    sums += other_sums
    return bits, sums


def reduce_scatter_table_partials(bits, sums, tile_cuts, group=None):
    """The slices' results meet range by range: tile_cuts[q] .. tile_cuts[q + 1] are the tiles of the windows rank q owns
    (len(tile_cuts) == world + 1; engine.Reference.tile_range of window_range_of_rank). Every rank sends rank q that range of its
    bitmaps and count sums (W gathers of 1 / W of the data each: the traffic of a reduce-scatter, on backends with and without
    all_to_all) and merges what it receives for its own range -- hit words OR-ed, count sums added, validity words as they are.
    Returns (bits, sums) of this rank's tile range, merged; a rank with an empty range gets empty tensors."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert len(tile_cuts) == world + 1
    W2 = 2 * TILE_WORDS
    mine_b = mine_s = None
    for q in range(world):
        lo, hi = int(tile_cuts[q]), int(tile_cuts[q + 1])
        pb = bits[lo * W2:hi * W2].contiguous()
        ps = sums[lo:hi].contiguous()
        gb = [torch.zeros_like(pb) for _ in range(world)] if rank == q else None
        gs = [torch.zeros_like(ps) for _ in range(world)] if rank == q else None
        if hi > lo:                                       # (every rank skips an empty range alike)
            dist.gather(pb, gb, dst=q, group=group)
            dist.gather(ps, gs, dst=q, group=group)
        if rank == q:
            mine_b, mine_s = gb[rank].clone(), gs[rank].clone()
            for r in range(world):
                if r != rank and hi > lo:
                    merge_table_partials(mine_b, mine_s, gb[r], gs[r])
    return mine_b, mine_s
