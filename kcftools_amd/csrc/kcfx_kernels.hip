// kcfx_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of the getVariations engine.
//
//   decode_records   : KMC .kmc_suf records + LUT -> (k-mer word, count)         (load time; KMC.java:427-450)
//   cells_*          : build the cell table (FORMAT_CELLS, kcfx_internal.h) from (k-mer word, count) pairs   (load time)
//   table_*          : build the 12-slot line table (FORMAT_LINES) from the same pairs                         (load time)
//   lookup_*         : batched KMC.getCount                                      (KMC.java:292-326)
//   screen_k31c<K>   : THE hot kernel (k = 27 .. 31, cell table; BASELINE.json runs K = 31). ASCII window text -> 2-bit pack
//                      in LDS -> per lane: both strands of the k-mer, tagged minimizer, its cell -> ONE 8-byte load per
//                      choice of cell, software-pipelined one word ahead -> wave ballots
//                                                     (GetVariants.java:202-227, Fasta.java:90-167, Kmer.java)
//   screen_k31       : the round-1 kernel (k = 31, line table): minimizer runs -> cooperative 128-byte line fetches ->
//                      12-way compare; now for forward-only databases and 3- / 4-byte counters
//   screen_k31_multi : the same for up to 8 resident line tables per pass over the reference (SURVEY f-4)
//   screen_tiles     : the general form (any k <= 32, both line-table modes)
//   reduce_windows   : hit bitmaps -> gap monoid -> kcfx_result per (window, sample)   (GetVariants.java:228-258)
//
// Integer / index work: no MFMA. Lane <-> consecutive k-mer start positions, so one __ballot is a
// 64-position slice of the window's hit bitmap.
#include <hip/hip_ext.h>

#include <cstdlib>

#include "kcfx_device.h"
#include "kcfx_plan.h"

#include <type_traits>

namespace kcfx {

// wave total of a u32 in six DPP adds (row_shr 1/2/4/8 leave each row's sum in its lane 15; row_bcast15 / row_bcast31
// carry it across rows); the result is lane 63's value, returned to every lane as a scalar
__device__ __forceinline__ uint32_t wave_sum_u32_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1, 3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2, 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ======================================================================================================
// table build
// ======================================================================================================
// every key slot = EMPTY_KEY (all ones), counts / meta = 0
__global__ void table_clear_kernel(uint4* lines, uint64_t n_vec, int32_t key_pieces) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_vec; i += stride) {
        const uint32_t v = ((int32_t)(i & 7) < key_pieces) ? 0xFFFFFFFFu : 0u;
        lines[i] = make_uint4(v, v, v, v);
    }
}

// Build pass 1 (minimizer mode): how many k-mers have each line as their minimizer home
__global__ void table_count_kernel(uint8_t* lines, TableView tv, int32_t k, const uint64_t* __restrict__ keys, int64_t n) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t line = minimizer_line(tv, keys[i], k);
        atomicAdd(reinterpret_cast<uint32_t*>(lines + (size_t)line * LINE_BYTES) + META_DWORD, 1u);
    }
}

__global__ void table_mark_heavy_kernel(uint8_t* lines, uint32_t n_lines) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n_lines; i += stride) {
        uint32_t* meta = reinterpret_cast<uint32_t*>(lines + (size_t)i * LINE_BYTES) + META_DWORD;
        *meta = (*meta > HEAVY_KEYS) ? BUILD_HEAVY : 0u;
    }
}

// Build pass 2. The meta dword is {bit 31 heavy mark, bits 0..23 arrival counter}; table_finalize turns it into
// {occupied slots, overflow flag, heavy flag}.
template <bool WIDE>
__global__ void table_insert_kernel(uint8_t* lines, TableView tv, int32_t k, const uint64_t* __restrict__ keys,
                                    const uint32_t* __restrict__ counts, int64_t n) {
    constexpr uint32_t S = WIDE ? SLOTS_U32 : SLOTS_U16;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t key = keys[i];
        const uint32_t c = counts[i];
        uint32_t line;
        if (tv.mode == MODE_MINIMIZER) {
            line = minimizer_line(tv, key, k);
            const uint32_t m = *(reinterpret_cast<const uint32_t*>(lines + (size_t)line * LINE_BYTES) + META_DWORD);
            if (m & BUILD_HEAVY) line = keyhash_line(tv, key);
        } else {
            line = keyhash_line(tv, key);
        }
        for (;;) {
            uint8_t* base = lines + (size_t)line * LINE_BYTES;
            const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t*>(base) + META_DWORD, 1u) & BUILD_COUNT_MASK;
            if (slot < S) {
                reinterpret_cast<uint64_t*>(base)[slot] = key;
                if (WIDE)
                    reinterpret_cast<uint32_t*>(base + 8 * SLOTS_U32)[slot] = c;
                else
                    reinterpret_cast<uint16_t*>(base + 8 * SLOTS_U16)[slot] = (uint16_t)c;
                break;
            }
            line = (line + 1 == tv.n_lines) ? 0 : line + 1;
        }
    }
}

template <bool WIDE>
__global__ void table_finalize_kernel(uint8_t* lines, uint32_t n_lines, unsigned long long* stats) {
    constexpr uint32_t S = WIDE ? SLOTS_U32 : SLOTS_U16;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n_lines; i += stride) {
        uint32_t* meta = reinterpret_cast<uint32_t*>(lines + (size_t)i * LINE_BYTES) + META_DWORD;
        const uint32_t raw = *meta;
        const uint32_t arrivals = raw & BUILD_COUNT_MASK;
        // a HEAVY line also carries the overflow bit: probes that miss there stay pending and are redirected
        *meta = ((arrivals > S) ? (S | META_OVERFLOW) : arrivals) | ((raw & BUILD_HEAVY) ? (META_HEAVY | META_OVERFLOW) : 0u);
        if (raw & BUILD_HEAVY) atomicAdd(&stats[0], 1ull);      // heavy minimizer home lines
        if (arrivals > S) atomicAdd(&stats[1], 1ull);           // overflowed lines
    }
}

// KMC.java:427-450 (dumpKmerTable): record j belongs to the last LUT slot i with lut[i] <= j;
// k-mer = prefix bases (i mod 4^p) || suffix bases; counter little-endian (KMC.java:395-401).
__global__ void decode_records_kernel(const uint8_t* __restrict__ records, int64_t first_record, int64_t n,
                                      int32_t record_size, int32_t suffix_bytes, int32_t counter_size, int32_t p,
                                      const uint64_t* __restrict__ lut, int64_t lut_entries,
                                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ counts_out) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const uint64_t j = (uint64_t)(first_record + t);
        // upper_bound(lut, j) - 1
        int64_t lo = 0, hi = lut_entries;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (lut[mid] <= j) lo = mid + 1; else hi = mid;
        }
        const uint64_t slot = (uint64_t)(lo - 1);
        const uint64_t prefix = p ? (slot & ((1ull << (2 * p)) - 1)) : 0;
        const uint8_t* r = records + t * record_size;
        uint64_t suf = 0;
        for (int b = 0; b < suffix_bytes; ++b) suf |= (uint64_t)r[b] << (56 - 8 * b);
        uint64_t key = suf >> (2 * p);
        if (p) key |= prefix << (64 - 2 * p);
        uint32_t c = 0;
        for (int b = 0; b < counter_size; ++b) c |= (uint32_t)r[suffix_bytes + b] << (8 * b);
        keys_out[t] = key;
        counts_out[t] = c;
    }
}

// ======================================================================================================
// lookup (batched KMC.getCount; not the hot path): 8 consecutive lanes share one key and fetch its line together
// ======================================================================================================
template <bool WIDE>
__global__ __launch_bounds__(256) void lookup_kernel(TableView tv, int32_t k, const uint64_t* __restrict__ keys,
                                                     int64_t n, int32_t* __restrict__ out) {
    const int64_t gtid = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> 3;
    const int piece = threadIdx.x & 7;
    const int lane = threadIdx.x & 63;
    for (int64_t i = gtid >> 3; i < n; i += stride) {     // (gtid >> 3) is uniform over each group of 8 lanes
        const uint64_t key = keys[i];
        uint32_t line = (tv.mode == MODE_MINIMIZER) ? minimizer_line(tv, key, k) : keyhash_line(tv, key);
        bool at_home = (tv.mode == MODE_MINIMIZER);
        int32_t result = 0;
        for (;;) {
            const uint4 mine = reinterpret_cast<const uint4*>(tv.lines + (size_t)line * LINE_BYTES)[piece];
            uint4 q[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int src = (lane & ~7) | s;
                q[s].x = __shfl(mine.x, src, 64);
                q[s].y = __shfl(mine.y, src, 64);
                q[s].z = __shfl(mine.z, src, 64);
                q[s].w = __shfl(mine.w, src, 64);
            }
            uint32_t c, meta;
            const bool any = match_line<WIDE, true>(q, key, c, meta);
            if (any) result = (int32_t)c;        // Java int (KMC.java:395-401)
            if (!probe_next(tv, at_home, any, meta, key, line)) break;
            at_home = false;
        }
        if (piece == 0) out[i] = result;
    }
}


// ======================================================================================================
// FORMAT_CELLS build + lookup (layout and exactness argument: kcfx_internal.h; cell derivation: kcfx_device.h)
// ======================================================================================================
// Who gets a contested cell. A reference window asks for the k-mers of a genome, which come in runs that share their
// minimizer; an isolated k-mer (a sequencing error, a contaminant) is asked for rarely. So: first the k-mer whose minimizer
// is shared by more k-mers of the database (class "crowded": >= CELL_CROWD of them in its sub-bucket = line + 2 more bits of
// the minimizer's mix), then "seen more than once", then -- round 4 -- the RUN the k-mer belongs to (cell_bid: one run takes every
// cell two runs contest), then the larger counter. Exactness never depends on this -- it only decides who lives in a cell and
// who elsewhere; on the bench database the classes halve the stash look-ups of a screening pass.
constexpr uint32_t CELL_CROWD = 3;
constexpr uint32_t CELL_COUNT_MAX = 0xFFFFu;                // what the 16-bit count field of a cell holds
// The key table of a COHORT (one table for S samples: every sample's records go through the same passes, the counters live
// beside the cells; presence != 0): the count field of a cell only says "present" (non-zero) and ranks the bids -- a record
// bids with its own counter, capped at what the field holds, so that the k-mers of the genome (seen depth times in every
// sample) keep their cells against the k-mers one sample saw once, exactly as in a single-sample table -- and the copies of a
// k-mer that several samples hold, bidding with different counters, are ONE k-mer: identity is compared without the count.
// The rank of a bid is not the record's own counter but (samples that hold the k-mer, then the counter), which the cohort
// loader has put into the upper half of the counts word (cohort_prio_kernel): at a site where the samples differ, the allele
// most samples carry -- as a rule the one the reference carries too, i.e. the one a screening pass asks for -- keeps the
// first-choice cell; its rarer siblings (same minimizer, same offset, one base of the flank different) go to second choices
// and the stash. (With the counter alone as the rank half of the reference's k-mers lived in the stash: 32 M HBM requests per
// pass of the bench cohort instead of 12 M.)
__device__ __forceinline__ uint32_t cell_build_count(uint32_t c, int32_t presence) { return presence ? (c >> 16) : c; }
// the k-mer of this record cannot have a cell: its counter does not fit (c = cell_build_count: a single sample's counter, or a cohort
// record's rank, where cohort_prio_kernel marks "some sample's counter is wider than the cohort's 8-bit cell counters")
__device__ __forceinline__ bool cell_is_wide(uint32_t c, int32_t presence) { return presence ? c == COHORT_RANK_WIDE : c > CELL_COUNT_MAX; }
// cell word `have` (identity bits kc, flags and filter bits excluded) is the k-mer whose word is `mine`
__device__ __forceinline__ bool cell_same(uint64_t have, uint64_t mine, uint64_t kc, int32_t presence) {
    if (!presence) return (have & kc) == mine;
    const uint64_t id = kc & 0x0000FFFFFFFFFFFFull;
    return (have >> 48) != 0 && (have & id) == (mine & id);
}
// The crowding counters: 2^pop_bits bytes, indexed by the top pop_bits of G32. pop_bits = 32 (4 GiB for the length of the build):
// a bucket is ONE minimizer. With line + 2 bits (round 2) an isolated k-mer that shared the line of a genomic run was classed
// with the run and, where its counter happened to be the larger, took the genomic k-mer's cell (profiles/r03_notes.md).
// kcfx_db_open falls back to lbits + 2 when the device cannot spare 4 GiB.
// WHICH k-mers are counted together (round 6): those of one LOCUS, not of one minimizer. The order prefers open syncmers, so the
// minimizers of any set of k-mers concentrate on ~1 / 11 of the canonical 16-mers -- the 123 M uniformly random k-mers of the bench
// database share their 16-mer with 0.6 other k-mers each, and a genomic run has as many strangers with ITS minimizer, in its line,
// with its rest of G32. Counted per minimizer such a stranger is "crowded" like the run, carries the run's own hash, and wins the
// cell it wants on a coin toss: 3.5 % of the reference's present k-mers sat outside their first-choice cell in the sparsest table
// (0.66 M of the 0.71 M queued look-ups of a pass were k-mers that ARE in the database; at 0.28 per cell with per-line counters
// 2.0 M of 2.2 M: profiles/r06_notes.md section 2). The key is the minimizer TOGETHER with the three bases next to it -- the
// bijection of G32 that an extended table takes its lines from (cell_g32_ext: the bases in front of the minimizer for the
// thirteen offsets that have them, else those behind it). The k-mers of a run share it (two groups per run at most), a stranger
// with the same 16-mer has other neighbours 63 times in 64. In an extended table (line, r) already are that key.
// by_minimizer: the plain G32 -- what the choice of an extended table is estimated from (kcfx_db_open: ext_by_crowding), and the
// A/B hook KCFX_CROWD_BY_MINIMIZER.
__device__ __forceinline__ uint32_t cell_crowd_key(const CellProbe& p, int32_t lbits, bool ext, bool by_minimizer) {
    // (r may carry CELL_R_FLIP above its 32 - lbits identifying bits: masked, or a forward-only database's reverse-strand k-mers
    // would be counted in another minimizer's bucket)
    const uint32_t g = (p.line << (32 - lbits)) | (p.r & (0xFFFFFFFFu >> lbits));
    return (ext || by_minimizer) ? g : cell_g32_ext(g, p.w & ~CELL_FLAG_MASK, p.sh);
}
__device__ __forceinline__ uint64_t cell_sub_bucket(const CellProbe& p, int32_t lbits, int32_t pop_bits, bool ext, bool by_minimizer) {
    return (uint64_t)(cell_crowd_key(p, lbits, ext, by_minimizer) >> (32 - pop_bits));
}

// Pass 0: k-mers per sub-bucket, one byte each (a byte that overflows spills into its neighbour: the classes are a
// heuristic, see above)
__global__ void cells_count_kernel(uint32_t* pop32, int32_t pop_bits, int32_t k, int32_t lbits, int32_t both, const uint64_t* __restrict__ keys,
                                   const uint32_t* __restrict__ counts, int64_t n, int32_t presence, int32_t ext, int32_t by_minimizer) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (counts[i] == 0) continue;                           // (presence mode: the same test)
        cells_visit(keys[i], k, lbits, both != 0, ext != 0, [&](const CellProbe& p, int a) {
            if (a != 0) return;                                 // the first of its cells names the k-mer's sub-bucket
            const uint64_t sub = cell_sub_bucket(p, lbits, pop_bits, ext != 0, by_minimizer != 0);
            atomicAdd(&pop32[sub >> 2], 1u << (8 * (uint32_t)(sub & 3)));
        });
    }
}

// ---- the cells of a record, worked out ONCE ----------------------------------------------------------------------------
// cells_visit costs ~600 instructions per k-mer (26 canonical 6-mer hashes, 16 order keys, the probe of every tied offset) and
// the build walks the records five times: on the bench database that was 5 x 14 ms of a 105 ms kcfx_db_open, all of it
// instruction issue. When the decoded records stay resident between the passes (they do unless the database is larger than
// half the device), the first pass keeps what it derived -- G32 and the flank in one 64-bit word, cell index and two flags in a
// byte: 9 bytes per record -- and the later passes read that instead. A k-mer with MORE than one cell (tied or palindromic
// minimizers: a few in 10^5) is flagged and walked by cells_visit as before.
struct RecProbes {
    const uint64_t* pw;     // G32 | flank << 32        (null: nothing was kept -- every pass derives the cells itself)
    const uint8_t* ps;      // cell index | REC_MULTI | REC_FLIP | slice class << 6
    int32_t ext;            // the table is an extended one (what is kept is the extended G32: cell_first_ext)
    int32_t regions;        // the table has R regions of 2^lbits lines (cell_line_index): where a probe's line IS
    int32_t slicing;        // 0, or (slices << 8 | slice): the table holds one slice of the database (cell_slice_of) -- a pass acts
                            // on the cells of a record that are this slice's and leaves the others to the devices that hold theirs
    __device__ __forceinline__ bool mine(const CellProbe& p, int32_t lbits) const {
        return slicing == 0 || cell_slice_of(p.cls, p.line, p.r, lbits, (uint32_t)slicing >> 8) == ((uint32_t)slicing & 0xFFu);
    }
};
// byte address of a probe's cell in the table
__device__ __forceinline__ size_t cell_byte_at(const CellProbe& p, int32_t lbits, int32_t regions) {
    return (size_t)cell_line_index(p.line, p.r, lbits, regions) * LINE_BYTES + 4 * p.sh;
}
constexpr uint8_t REC_MULTI = 0x10, REC_FLIP = 0x20;
constexpr uint32_t REC_CLASS_SHIFT = 6;                    // bits 7..6: order_slice_class of the minimizer
template <class F>
__device__ __forceinline__ int record_cells(const RecProbes& rp, int64_t i, uint64_t key, int32_t k, int32_t lbits, bool both, F&& f) {
    if (rp.pw) {
        const uint32_t s = rp.ps[i];
        if (!(s & REC_MULTI)) {
            const uint64_t w = rp.pw[i];
            const uint32_t g = (uint32_t)w;
            CellProbe p;
            p.line = g >> (32 - lbits);
            p.r = (g & (0xFFFFFFFFu >> lbits)) | ((s & REC_FLIP) ? CELL_R_FLIP : 0u);
            p.w = (uint32_t)(w >> 32);
            p.sh = 2u * (s & 15u);
            p.cls = s >> REC_CLASS_SHIFT;
            f(p, 0);
            return 1;
        }
    }
    return cells_visit(key, k, lbits, both, rp.ext != 0, f);
}

// Pass 0 when the records stay resident: cells_count_kernel + the probes kept
__global__ void cells_resolve_kernel(uint32_t* pop32, int32_t pop_bits, int32_t k, int32_t lbits, int32_t both, const uint64_t* __restrict__ keys,
                                     const uint32_t* __restrict__ counts, int64_t n, uint64_t* __restrict__ pw, uint8_t* __restrict__ ps, int32_t ext, int32_t by_minimizer) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (counts[i] == 0) {                                   // never stored (every later pass skips it on its counter)
            pw[i] = 0ull;
            ps[i] = 0;
            continue;
        }
        CellProbe first{0, 0, 0, 0, 0};
        const int nc = cells_visit(keys[i], k, lbits, both != 0, ext != 0, [&](const CellProbe& p, int a) {
            if (a == 0) first = p;
        });
        const uint64_t sub = cell_sub_bucket(first, lbits, pop_bits, ext != 0, by_minimizer != 0);
        atomicAdd(&pop32[sub >> 2], 1u << (8 * (uint32_t)(sub & 3)));
        const uint32_t g = (first.line << (32 - lbits)) | (first.r & (0xFFFFFFFFu >> lbits));
        pw[i] = (uint64_t)g | ((uint64_t)first.w << 32);
        ps[i] = (uint8_t)((first.sh >> 1) | (nc > 1 ? REC_MULTI : 0) | ((first.r & CELL_R_FLIP & ~(0xFFFFFFFFu >> lbits)) ? REC_FLIP : 0) |
                          (first.cls << REC_CLASS_SHIFT));
    }
}

// Pass 1 (cls = 1: claim) and pass 2 (cls = 0: write, after ALL of pass 1). Claim: every stored k-mer bids for its first-choice
// cell(s) with a 64-bit atomic max of its RANK -- not of its cell word: [crowded : 1][counter >= 2 : 1][run : 14][counter : 16]
// [locus : 8][hash of (w, r) : 24] (round 6; before: [hash : 32]). "run" is a hash of the minimizer (of G32: line and the identifying bits of r), the same for every k-mer of
// a minimizer run: where two runs share a line the SAME run wins every cell both want, so a screening pass over the winner's locus
// finds all of its k-mers at home and only the loser's locus pays the request for a second line (with the counter as the rank both
// runs lost about half of the contested cells and both loci paid: round 4, profiles/r04_notes.md). "Seen more than once" stays
// above the run -- the k-mers of the genome keep their cells against a run of sequencing errors whatever the hashes say; a finer
// counter class would split a run wherever the coverage steps -- and the counter itself below it: among k-mers of ONE minimizer (the alleles of a site, an error and its original:
// same cell) the most abundant one still wins. Write: a k-mer that finds its own rank in the cell stores its cell word there. Two
// k-mers with the same rank (a 32-bit hash collision among the bidders of one cell) both store; the pass after this one
// (cells_settle_kernel) looks at the cell WORDS and treats whoever is not standing in its cell as a loser, so exactness never
// depends on the bidding.
// (a cohort's key table, presence != 0: c is the rank cohort_prio_kernel made -- samples that hold the k-mer << 12 | counter -- and
// "seen more than once" reads "held by more than one sample": the k-mers the samples share, which are the ones a reference
// asks for, stay above a run that is private to one sample. With the run directly under the class the bench cohort screened
// in 0.292 ms instead of 0.217: half of the shared runs lost their contests to private ones.)
// (round 6: the crowding class is counted per LOCUS -- cell_crowd_key -- and the locus also breaks the ties that are left when two
// k-mers of ONE minimizer have the same counter: of two loci that contain the same 16-mer with equally abundant k-mers ONE keeps
// every cell both want, as of two runs that share a line, instead of splitting them k-mer by k-mer on a hash. It stands BELOW the
// counter: among the copies of a repeat family the consensus k-mer -- the one most copies carry, the one a reference asks for most
// often -- must keep its cell against a rare variant whatever the hashes say. With the locus in the run's place the
// repeat-structured genome of the bench sent 1.6 M k-mers per pass to the stash instead of 0.5 M: profiles/r06_notes.md section 2.)
// rank = [crowded : 1][counter >= 2 : 1][run : 14] [counter : 16] [locus : 8][hash of (w, r, cell) : 24]
__device__ __forceinline__ unsigned long long cell_bid(const CellProbe& p, uint32_t c, bool crowded, int32_t lbits, int32_t presence, bool ext, bool by_minimizer) {
    const uint32_t g = (p.line << (32 - lbits)) | (p.r & (0xFFFFFFFFu >> lbits));
    const uint32_t run = (g * 0x9E3779B1u) >> 18;
    const bool more = presence ? (c >> 12) >= 2u : c >= 2u;
    const uint32_t top = ((crowded ? 1u : 0u) << 15) | ((more ? 1u : 0u) << 14) | run;
    const uint32_t locus = by_minimizer ? 0u : (cell_crowd_key(p, lbits, ext, false) * 0x85EBCA6Bu) >> 24;
    const uint32_t idh = ((p.w & ~CELL_FLAG_MASK) * 0x85EBCA6Bu) ^ ((p.r & 0xFFFFu) * 0xC2B2AE35u) ^ (p.sh * 0x27D4EB2Fu);
    return ((unsigned long long)top << 48) | ((unsigned long long)(c & 0xFFFFu) << 32) | ((unsigned long long)locus << 24) | (by_minimizer ? idh : idh >> 8);
}
__global__ void cells_insert_kernel(uint8_t* lines, const uint32_t* __restrict__ pop32, int32_t pop_bits, int32_t cls, int32_t k, int32_t lbits, int32_t both,
                                    const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, int64_t n, int32_t presence, RecProbes rp, int32_t by_minimizer) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t c = cell_build_count(counts[i], presence);
        if (c == 0) continue;                       // can never be observed (count >= -c >= 1 is needed): not stored
        if (cell_is_wide(c, presence)) continue;    // a counter that does not fit a cell (3- / 4-byte; a cohort's: > 8 bits): lives in the stash (settle)
        bool crowded = false;
        record_cells(rp, i, keys[i], k, lbits, both != 0, [&](const CellProbe& p, int a) {
            if (a == 0) {
                const uint64_t sub = cell_sub_bucket(p, lbits, pop_bits, rp.ext != 0, by_minimizer != 0);
                crowded = ((pop32[sub >> 2] >> (8 * (uint32_t)(sub & 3))) & 0xFFu) >= CELL_CROWD;
            }
            if (!rp.mine(p, lbits)) return;             // (a table slice: the crowding counters are of the whole database, the cells of this slice)
            unsigned long long* cell = reinterpret_cast<unsigned long long*>(lines + cell_byte_at(p, lbits, rp.regions));
            const unsigned long long bid = cell_bid(p, c, crowded, lbits, presence, rp.ext != 0, by_minimizer != 0);
            if (cls) atomicMax(cell, bid);
            else if (__atomic_load_n(cell, __ATOMIC_RELAXED) == bid) __atomic_store_n(cell, (unsigned long long)cell_word(p.w, p.r, c), __ATOMIC_RELAXED);
        });
    }
}

// where a k-mer stands after the build passes so far: in its first-choice cell (0), in a free cell of the same line (1: k < 31,
// cell_alt_free), in the same cell of its second line (2) or of its third line (4: cell_third_line), or nowhere (STAND_NONE: the stash)
constexpr int STAND_FIRST = 0, STAND_FREE = 1, STAND_SECOND_LINE = 2, STAND_NONE = 3, STAND_THIRD_LINE = 4;
__device__ __forceinline__ int cell_standing(const uint8_t* lines, const CellProbe& p, uint32_t c, int32_t lbits, bool both, int W,
                                             int32_t presence = 0, int32_t regions = 1) {
    const uint64_t kc = cell_keycount_mask(lbits, both);              // (flags and filter bits are not part of the k-mer)
    const uint8_t* line = lines + (size_t)cell_line_index(p.line, p.r, lbits, regions) * LINE_BYTES;
    const uint64_t first = __atomic_load_n(reinterpret_cast<const uint64_t*>(line + 4 * p.sh), __ATOMIC_RELAXED);
    if (cell_same(first, cell_word(p.w, p.r, c), kc, presence)) return STAND_FIRST;
    if (cell_alt_free(W, lbits, both)) {
        const CellAlt a = cell_alt_of(p.sh, p.w, p.r, W, lbits, both);
        const uint64_t second = __atomic_load_n(reinterpret_cast<const uint64_t*>(line + 4 * a.sh), __ATOMIC_RELAXED);
        if (cell_same(second, cell_word(p.w, a.r, c), 0xFFFF0000FFFFFFFCull | ((uint64_t)a.rmask << 32), presence)) return STAND_FREE;
    }
    const CellProbe q = cell_second_line(p, lbits);
    const uint64_t there = __atomic_load_n(reinterpret_cast<const uint64_t*>(lines + cell_byte_at(q, lbits, regions)), __ATOMIC_RELAXED);
    if (cell_same(there, cell_word(q.w, q.r, c), kc, presence)) return STAND_SECOND_LINE;
    if (cell_has_third(lbits, both)) {
        const CellProbe t = cell_third_line(p, lbits, both, W);
        const uint64_t third = __atomic_load_n(reinterpret_cast<const uint64_t*>(lines + cell_byte_at(t, lbits, regions)), __ATOMIC_RELAXED);
        if (cell_same(third, cell_word(t.w, t.r, c), kc, presence)) return STAND_THIRD_LINE;
    }
    return STAND_NONE;
}

// Pass 2 (after ALL bidding has completed): a k-mer that lost its bid takes, in this order, a free cell of its line (k < 31), the
// same cell of its second line (flagging its first-choice cell: "someone who belongs here lives elsewhere", with its filter
// bit, cell_filter_pos) -- each only if still empty -- or is counted for the stash (then the second-line cell is flagged
// too). stats[0] = k-mers for the stash, [1] = k-mers stored (count != 0), [2] = k-mers with more than one first-choice cell,
// [3] = k-mers placed outside their first-choice cell
// (lost: when the load cannot keep its records between the passes, the k-mers this pass counts for the stash are also WRITTEN DOWN --
// key and counter, appended by the wave -- so that filling the stash needs no further pass over the file: LostList, kcfx_internal.h)
__global__ void cells_settle_kernel(uint8_t* lines, int32_t k, int32_t lbits, int32_t both_i, const uint64_t* __restrict__ keys,
                                    const uint32_t* __restrict__ counts, int64_t n, unsigned long long* stats, int32_t presence, RecProbes rp, LostList ll) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    uint32_t n_lost = 0, n_stored = 0, n_multi = 0, n_second = 0;      // per lane; one atomic per wave at the end
    const bool both = both_i != 0;
    for (; i < n; i += stride) {
        const uint32_t c = cell_build_count(counts[i], presence);
        if (c == 0) continue;
        bool lost = false;
        const bool wide = cell_is_wide(c, presence);
        int nc = 0;                                     // its cells in this table (all of them unless the table is a slice)
        record_cells(rp, i, keys[i], k, lbits, both, [&](const CellProbe& p, int) {
            if (!rp.mine(p, lbits)) return;
            ++nc;
            uint8_t* line = lines + (size_t)cell_line_index(p.line, p.r, lbits, rp.regions) * LINE_BYTES;
            // "someone who belongs here lives elsewhere" + this k-mer's filter bit IN THAT CELL: keyed by how its flank differs from the
            // occupant's (cell_filter_pos) -- the occupant as it stands now, which is final (see there); every bit for an empty cell
            auto flag = [&](unsigned long long* cell, uint32_t w) {
                unsigned long long mark = CELL_FLAG_STASH;
                if (cell_spare_bits(lbits, both) > 0) {
                    const unsigned long long now = __atomic_load_n(cell, __ATOMIC_RELAXED);
                    mark |= (now >> 48) == 0 ? (unsigned long long)cell_filter_all(lbits, both) << 32
                                             : 1ull << (32 + cell_filter_pos(w, (uint32_t)now, lbits, both));
                }
                atomicOr(cell, mark);
            };
            if (wide) {
                // never bid: every cell a query could derive for it -- first choice and second line -- is flagged (whoever lives
                // there, or nobody), with its filter bit
                const CellProbe q = cell_second_line(p, lbits);
                flag(reinterpret_cast<unsigned long long*>(line + 4 * p.sh), p.w);
                flag(reinterpret_cast<unsigned long long*>(lines + cell_byte_at(q, lbits, rp.regions)), q.w);
                if (cell_has_third(lbits, both)) {
                    const CellProbe t = cell_third_line(p, lbits, both, cell_nw(k));
                    flag(reinterpret_cast<unsigned long long*>(lines + cell_byte_at(t, lbits, rp.regions)), t.w);
                }
                lost = true;
                return;
            }
            if (cell_standing(lines, p, c, lbits, both, cell_nw(k), presence, rp.regions) != STAND_NONE) return;   // (already placed: the same word twice in the input)
            // an empty cell (count field 0: nobody's first choice landed there, no other loser took it) is taken with a CAS that
            // keeps the flag and filter bits it may carry
            auto take = [&](unsigned long long* cell, unsigned long long mine, unsigned long long kcm) -> bool {
                unsigned long long old = __atomic_load_n(cell, __ATOMIC_RELAXED);
                while ((old >> 48) == 0) {
                    const unsigned long long prev = atomicCAS(cell, old, mine | (old & ~kcm));
                    if (prev == old) return true;
                    old = prev;
                }
                return cell_same(old, mine, kcm, presence);                   // (a copy of the same word got there first)
            };
            if (cell_alt_free(cell_nw(k), lbits, both)) {                    // k < 31: a free cell of the same line (looked at in the main pass)
                const CellAlt a = cell_alt_of(p.sh, p.w, p.r, cell_nw(k), lbits, both);
                if (take(reinterpret_cast<unsigned long long*>(line + 4 * a.sh), cell_word(p.w, a.r, c),
                         0xFFFF0000FFFFFFFCull | ((unsigned long long)a.rmask << 32))) {
                    ++n_second;
                    return;
                }
            }
            // the same cell of its second line: the first-choice cell says "look on" (flag + filter bit)
            const CellProbe q = cell_second_line(p, lbits);
            unsigned long long* there = reinterpret_cast<unsigned long long*>(lines + cell_byte_at(q, lbits, rp.regions));
            flag(reinterpret_cast<unsigned long long*>(line + 4 * p.sh), p.w);
            if (take(there, cell_word(q.w, q.r, c), cell_keycount_mask(lbits, both))) {
                ++n_second;
                return;
            }
            flag(there, q.w);                                                // ... and so does the second-line cell (occupied: take() failed)
            if (cell_has_third(lbits, both)) {
                // the same cell of its third line: where the losers of DIFFERENT loci with this minimizer part ways
                const CellProbe t = cell_third_line(p, lbits, both, cell_nw(k));
                unsigned long long* third = reinterpret_cast<unsigned long long*>(lines + cell_byte_at(t, lbits, rp.regions));
                if (take(third, cell_word(t.w, t.r, c), cell_keycount_mask(lbits, both))) {
                    ++n_second;
                    return;
                }
                flag(third, t.w);                                            // ... and the third-line cell: the stash
            }
            lost = true;
        });
        if (nc == 0) continue;                          // (a table slice: none of its cells is here)
        if (lost && ll.keys) {
            // one atomic per wave: the lanes that are here take consecutive entries; an entry beyond the capacity is counted and not
            // written (the load then fills the stash from the file as before)
            const uint64_t here = __builtin_amdgcn_ballot_w64(true);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(here >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)here, 0u));
            unsigned long long base = 0;
            if (rank == 0) base = atomicAdd(ll.n, (unsigned long long)__popcll(here));
            const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)base), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            const unsigned long long at = (((unsigned long long)bhi << 32) | blo) + rank;
            if (at < ll.cap) {
                ll.keys[at] = keys[i];
                ll.counts[at] = counts[i];
            }
        }
        n_lost += lost ? 1u : 0u;
        ++n_stored;
        n_multi += nc > 1 ? 1u : 0u;
    }
    // (150 M atomics on four addresses took 80 ms; a wave total each takes none)
    const uint32_t t0 = wave_sum_u32_dpp(n_lost), t1 = wave_sum_u32_dpp(n_stored), t2 = wave_sum_u32_dpp(n_multi), t3 = wave_sum_u32_dpp(n_second);
    if ((threadIdx.x & 63) == 0) {
        if (t0) atomicAdd(&stats[0], (unsigned long long)t0);
        if (t1) atomicAdd(&stats[1], (unsigned long long)t1);
        if (t2) atomicAdd(&stats[2], (unsigned long long)t2);
        if (t3) atomicAdd(&stats[3], (unsigned long long)t3);
    }
}

// Pass 3 (after pass 2 has completed): a k-mer that stands in neither cell of one of its candidates goes to the stash.
__global__ void cells_stash_kernel(const uint8_t* __restrict__ lines, int32_t k, int32_t lbits, int32_t both, const uint64_t* __restrict__ keys,
                                   const uint32_t* __restrict__ counts, int64_t n, uint4* stash, uint32_t mask, int32_t presence, RecProbes rp) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t c = cell_build_count(counts[i], presence);
        if (c == 0) continue;
        const uint64_t key = keys[i];
        const bool wide = cell_is_wide(c, presence);    // (counters wider than a cell's 16 bits; a cohort's: than its 8)
        bool lost = false;
        if (!wide || rp.slicing != 0)
            record_cells(rp, i, key, k, lbits, both != 0, [&](const CellProbe& p, int) {
                if (!rp.mine(p, lbits)) return;
                lost |= wide || cell_standing(lines, p, c, lbits, both != 0, cell_nw(k), presence, rp.regions) == STAND_NONE;
            });
        else lost = true;
        if (!lost) continue;
        uint32_t slot = stash_slot(key, mask);
        for (;;) {
            unsigned long long* kp = reinterpret_cast<unsigned long long*>(stash + slot);
            const unsigned long long prev = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (prev == EMPTY_KEY) {
                reinterpret_cast<uint32_t*>(kp)[2] = c;
                break;
            }
            if (prev == key) break;                 // the same word twice in the input
            slot = (slot + 1) & mask;
        }
    }
}

// stored counter of `key` in the stash, 0 when absent
__device__ __forceinline__ uint32_t stash_find(const TableView& tv, uint64_t key) {
    uint32_t slot = stash_slot(key, tv.stash_mask);
    for (;;) {
        const uint4 e = tv.stash[slot];
        const uint64_t ek = u64_of(e.x, e.y);
        if (ek == key) return e.z;
        if (ek == EMPTY_KEY) return 0u;
        slot = (slot + 1) & tv.stash_mask;
    }
}

// batched KMC.getCount on a FORMAT_CELLS table, one lane per key (utility path)
__global__ __launch_bounds__(256) void lookup_cells_kernel(TableView tv, int32_t k, int32_t both_i, const uint64_t* __restrict__ keys, int64_t n,
                                                           int32_t* __restrict__ out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t key = keys[i];
        int32_t result = 0;
        // a both-strands database stores only canonical words; a k-mer word has its low 64 - 2k bits clear
        const bool both = both_i != 0;
        if ((!both || key <= revcomp_word(key, k)) && (key & ~(~0ull << (64 - 2 * k))) == 0) {
            CellProbe c0{0, 0, 0, 0, 0};                            // any one of its cells decides: the first
            cells_visit(key, k, tv.lbits, both, tv.ext != 0, [&](const CellProbe& p, int a) {
                if (a == 0) c0 = p;
            });
            const uint8_t* line = tv.lines + (size_t)cell_line_index(c0.line, c0.r, tv.lbits, tv.regions) * LINE_BYTES;
            const uint64_t first = *reinterpret_cast<const uint64_t*>(line + 4 * c0.sh);
            const uint64_t KEY_BITS = cell_keycount_mask(tv.lbits, both) & 0x0000FFFFFFFFFFFFull;     // w without the flags, r without the filter
            // "someone who belongs in this cell lives elsewhere" and, when the table has filter bits, "with my filter bit"
            auto sent_on = [&](uint64_t cell) {            // (w is the same in every line of the k-mer; the occupant is the cell's own)
                return (cell & CELL_FLAG_STASH) &&
                       (cell_spare_bits(tv.lbits, both) == 0 || ((cell >> (32 + cell_filter_pos(c0.w, (uint32_t)cell, tv.lbits, both))) & 1));
            };
            bool found = false;
            if (((first ^ cell_word(c0.w, c0.r, 0)) & KEY_BITS) == 0 && (first >> 48) != 0) {
                result = (int32_t)(first >> 48);
                found = true;
            } else if (cell_alt_free(cell_nw(k), tv.lbits, both)) {
                const CellAlt a = cell_alt_of(c0.sh, c0.w, c0.r, cell_nw(k), tv.lbits, both);
                const uint64_t second = *reinterpret_cast<const uint64_t*>(line + 4 * a.sh);
                const uint64_t KEY_BITS_ALT = 0x00000000FFFFFFFCull | ((uint64_t)a.rmask << 32);
                if (((second ^ cell_word(c0.w, a.r, 0)) & KEY_BITS_ALT) == 0 && (second >> 48) != 0) {
                    result = (int32_t)(second >> 48);
                    found = true;
                }
            }
            if (!found && sent_on(first)) {
                const CellProbe q = cell_second_line(c0, tv.lbits);
                const uint64_t there = *reinterpret_cast<const uint64_t*>(tv.lines + cell_byte_at(q, tv.lbits, tv.regions));
                if (((there ^ cell_word(q.w, q.r, 0)) & KEY_BITS) == 0 && (there >> 48) != 0) {
                    result = (int32_t)(there >> 48);
                } else if (sent_on(there)) {
                    bool on = true;
                    if (cell_has_third(tv.lbits, both)) {
                        const CellProbe t = cell_third_line(c0, tv.lbits, both, cell_nw(k));
                        const uint64_t third = *reinterpret_cast<const uint64_t*>(tv.lines + cell_byte_at(t, tv.lbits, tv.regions));
                        if (((third ^ cell_word(t.w, t.r, 0)) & KEY_BITS) == 0 && (third >> 48) != 0) {
                            result = (int32_t)(third >> 48);
                            on = false;
                        } else {
                            on = sent_on(third);
                        }
                    }
                    if (on) result = (int32_t)stash_find(tv, key);
                }
            }
        }
        out[i] = result;
    }
}

// ---- cohort tables: S samples in ONE cell table (SURVEY f-4 taken into the table; Cohort.java:71-119, run_kcftools.sh:124-140) ----
// The key table is an ordinary FORMAT_CELLS table built from the records of all S samples with every count field = 1
// (presence); the counters live beside it, COHORT_STRIDE u16 per cell (one 16-byte load per lane) and per stash slot.
// A reference k-mer is then located ONCE for all samples -- one line request per minimizer run plus the lines of the
// counters its cells span -- instead of once per sample.
// slot of `key` in the stash, ~0u when absent
__device__ __forceinline__ uint32_t stash_find_slot(const TableView& tv, uint64_t key) {
    uint32_t slot = stash_slot(key, tv.stash_mask);
    for (;;) {
        const uint4 e = tv.stash[slot];
        const uint64_t ek = u64_of(e.x, e.y);
        if (ek == key) return slot;
        if (ek == EMPTY_KEY) return ~0u;
        slot = (slot + 1) & tv.stash_mask;
    }
}

// Votes: an open-addressing table {u64 key, u32 samples that hold it, u32 ~(largest counter any of them has)} over the records of
// all samples (the stash's entry layout and probing), then every record's bid rank = min(samples, 15) << 12 | min(counter, 4095)
// into the upper half of its counts word (counters of a cohort are at most 16 bits wide). A k-mer that SOME sample counted
// COHORT_CELL_MAX + 1 times or more does not fit the table's 8-bit counters: every one of its records gets the rank COHORT_RANK_WIDE
// (no real rank: that would take 15 samples) -- it never bids, flags the cells it would have and lives in the stash, whose
// counters are 16 bits wide (as a k-mer with a 3- / 4-byte counter does in a single sample's table).
__global__ void cohort_vote_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ counts, int64_t n, uint4* votes, uint32_t mask) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        if (counts[i] == 0) continue;
        const uint64_t key = keys[i];
        uint32_t slot = stash_slot(key, mask);
        for (;;) {
            unsigned long long* kp = reinterpret_cast<unsigned long long*>(votes + slot);
            const unsigned long long prev = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (prev == EMPTY_KEY || prev == key) {
                atomicAdd(reinterpret_cast<uint32_t*>(kp) + 2, 1u);
                atomicMin(reinterpret_cast<uint32_t*>(kp) + 3, ~(counts[i] & 0xFFFFu));     // (the entry's last word: ~ the largest counter)
                break;
            }
            slot = (slot + 1) & mask;
        }
    }
}
__global__ void cohort_prio_kernel(const uint64_t* __restrict__ keys, uint32_t* counts, int64_t n, const uint4* __restrict__ votes, uint32_t mask) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t c = counts[i] & 0xFFFFu;
        if (c == 0) continue;
        const uint64_t key = keys[i];
        uint32_t slot = stash_slot(key, mask), v = 1, widest = c;
        for (;;) {
            const uint4 e = votes[slot];
            const uint64_t ek = u64_of(e.x, e.y);
            if (ek == key) {
                // entries were zero-filled apart from the key (0xFF..): the vote word started at 0xFFFFFFFF and wrapped
                v = e.z + 1u;
                widest = ~e.w;
                break;
            }
            if (ek == EMPTY_KEY) break;
            slot = (slot + 1) & mask;
        }
        const uint32_t rank = widest > COHORT_CELL_MAX ? COHORT_RANK_WIDE : (min(v, 15u) << 12 | min(c, 4095u));
        counts[i] = c | (rank << 16);
    }
}

// After the key table is complete: sample `s` writes its counters where its k-mers stand -- in every cell copy (a k-mer with
// tied minimizers has several), or in the stash. Records of one sample are distinct k-mers and samples write different
// columns, so there is no race.
__global__ void cohort_fill_kernel(TableView tv, int32_t k, int32_t both_i, const uint64_t* __restrict__ keys,
                                   const uint32_t* __restrict__ counts, int64_t n, int32_t sample, uint8_t* cell_counts,
                                   uint16_t* stash_counts, RecProbes rp) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const bool both = both_i != 0;
    for (; i < n; i += stride) {
        const uint32_t c = counts[i] & 0xFFFFu;                 // (the upper half is the bid rank)
        if (c == 0) continue;
        const uint64_t key = keys[i];
        bool stashed = false;
        record_cells(rp, i, key, k, tv.lbits, both, [&](const CellProbe& p, int) {
            const int st = cell_standing(tv.lines, p, 1u, tv.lbits, both, cell_nw(k), 1, tv.regions);
            if (st == STAND_NONE) {
                stashed = true;
                return;
            }
            const uint32_t sh = st == STAND_FREE ? cell_alt_of(p.sh, p.w, p.r, cell_nw(k), tv.lbits, both).sh : p.sh;
            const CellProbe at = st == STAND_SECOND_LINE ? cell_second_line(p, tv.lbits)
                                 : st == STAND_THIRD_LINE ? cell_third_line(p, tv.lbits, both, cell_nw(k)) : p;
            const size_t cell = (size_t)cell_line_index(at.line, at.r, tv.lbits, tv.regions) * 16 + (sh >> 1);
            cell_counts[cell * COHORT_STRIDE + sample] = (uint8_t)c;      // (<= COHORT_CELL_MAX: a k-mer with a wider counter anywhere has no cell)
        });
        if (stashed) {
            const uint32_t slot = stash_find_slot(tv, key);
            if (slot != ~0u) stash_counts[(size_t)slot * COHORT_STRIDE + sample] = (uint16_t)c;
        }
    }
}

// ======================================================================================================
// screening: wave-level helpers
// ======================================================================================================
constexpr int MAX_CHUNKS = (31 + TILE_K + 31 + 31) / 32 + 1;   // 32-base chunks a tile can span
constexpr int WAVES = BLOCK / 64;

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// __ballot() takes an int: hipcc materialises the predicate (v_cndmask 0/1) and compares it again, two VALU
// instructions per ballot in a VALU-bound kernel. The builtin reads the condition mask directly.
__device__ __forceinline__ uint64_t wave_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// wave-private LDS hand-off: LDS operations of one wave execute in order, so only the COMPILER has to be kept from
// moving accesses across this point. (A wavefront-scope fence also works but makes hipcc wait for vmcnt(0), i.e. for
// the wave's outstanding global stores.)
__device__ __forceinline__ void wave_lds_fence() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// ======================================================================================================
// the probe: KMC.getCount for the 64 k-mers of one word (shared by all screening kernels)
// ======================================================================================================
// Table lines land in LDS by LDS-DMA (global_load_lds_dwordx4): no destination VGPRs, no ds_write. Ballots are taken of
// single compares and combined on the scalar unit (a ballot of a conjunction costs hipcc a v_cndmask + v_cmp to
// re-materialise the predicate).
// LDS-DMA puts lane i's 16 bytes at base + 16 i. Nine lanes per run, the ninth idle, leave a 16-byte hole after every
// line: lines start 144 bytes apart, on distinct LDS banks for the match reads (a 128-byte pitch costs 4x the bank
// conflict cycles, SQ_LDS_BANK_CONFLICT). 7 runs per instruction, 2 instructions per chunk.
constexpr int K31_RUNS = 14;
constexpr int K31_PITCH = 9;                                   // uint4 per buffered line

__device__ __forceinline__ uint32_t wave_shr1(uint32_t x) {     // lane l gets lane l-1's value, lane 0 gets 0 (bound_ctrl)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, true);
}

// Per-lane constants of the cooperative line fetch
struct K31Fetch {
    int fetch_run;        // lane / 9: the run (0..6, 7 = idle) this lane fetches for
    int piece;            // its 16-byte piece of the line (8 = idle)
    bool fetcher;
    uint4* lines;         // this wave's line buffers (2 x 63 uint4)
    uint32_t* runline;    // this wave's run -> line index list (65 entries)
};

// KMC.getCount for the 64 k-mers of one word against one table: `key` (canonical word), `home` (first line to look at), `V`
// (ballot of valid k-mers). Returns the stored counter, 0 when absent / invalid.
// Cooperative probe rounds: each round fetches, for every run of adjacent pending lanes with the same target line, that
// line ONCE (8 lanes x 16 B = one 128-byte request, LDS-DMA) into wave-private LDS, then every lane compares its key
// against its run's line. Probe state lives in VGPR integers and is updated with selects: lane-varying bools carried
// through the loops would be kept as SGPR masks and cost three scalar instructions per update and block.
// CHECK_NS: k = 32 (the all-ones word is a k-mer, see match_line).
//
// One round for the lanes in PM (= ballot of st == 1, not empty): st becomes 2 (found, cnt set), 1 (the line overflowed:
// tgt moved on to the next line) or 0 (absent); meta = meta dword of the line a lane looked at.
template <bool WIDE, bool CHECK_NS>
__device__ __forceinline__ void probe_round(const TableView& tv, const K31Fetch& f, uint64_t PM, uint64_t key, uint32_t& tgt,
                                            uint32_t& st, uint32_t& cnt, uint32_t& meta_seen) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    // heads of runs: pending lanes whose predecessor is not pending (a scalar shift of PM) or aims at another line
    const uint64_t HM = PM & (~(PM << 1) | wave_ballot(wave_shr1(tgt) != tgt));
    const int nruns = __popcll(HM);
    // index of this lane's run = heads at or below it, minus one
    const uint64_t H1 = HM >> 1;
    const int my_run = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(H1 >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)H1, (uint32_t)((int)(HM & 1) - 1)));
    // every pending lane stores its run's line index (same value within a run); the others hit a dummy slot
    f.runline[(st == 1u) ? my_run : 64] = tgt;
    wave_lds_fence();
    for (int c0 = 0; c0 < nruns; c0 += K31_RUNS) {
        const int r0 = c0 + f.fetch_run, r1 = r0 + 7;
        // 8 lanes x 16 B = one 128-byte request per run
        if (f.fetcher && r0 < nruns)
            __builtin_amdgcn_global_load_lds((gptr_t)(tv.lines + (size_t)f.runline[r0] * LINE_BYTES + 16 * f.piece),
                                             (lptr_t)f.lines, 16, 0, 0);
        if (f.fetcher && r1 < nruns)
            __builtin_amdgcn_global_load_lds((gptr_t)(tv.lines + (size_t)f.runline[r1] * LINE_BYTES + 16 * f.piece),
                                             (lptr_t)(f.lines + 7 * K31_PITCH), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        // every lane matches (idle lanes against buffer 0); the outcome is merged with selects
        const uint32_t rel = (uint32_t)(my_run - c0);
        const bool mine = (st == 1u) && (rel < (uint32_t)K31_RUNS);
        const uint4* src = f.lines + (mine ? rel : 0u) * K31_PITCH;
        uint32_t c, meta;
        const bool any = match_line_lds<WIDE, CHECK_NS>(src, key, c, meta);
        const uint32_t ov = (meta >> 8) & 1u;                       // overflowed line: look at the next one
        const uint32_t nst = any ? 2u : ov;
        const uint32_t ntgt = (tgt + 1 == tv.n_lines) ? 0u : tgt + 1;
        cnt = (mine && any) ? c : cnt;                              // Java int (KMC.java:395-401)
        tgt = (mine && nst == 1u) ? ntgt : tgt;
        meta_seen = mine ? meta : meta_seen;
        st = mine ? nst : st;
        wave_lds_fence();
    }
}

// HEAVY minimizer home (table_finalize also sets its overflow bit, so lanes that missed there are still pending): every
// k-mer homed there by minimizer lives at its key-hash line instead -- jump, do not walk. `home_meta` = meta dword of the
// home line, PM = pending lanes after the first round.
__device__ __forceinline__ void heavy_redirect(const TableView& tv, int lane, uint64_t PM, uint32_t home_meta, uint64_t key,
                                               uint32_t& tgt) {
    const uint64_t RM = PM & wave_ballot((home_meta & META_HEAVY) != 0u);
    if (RM) {
        // rare: keep the key hash (three multiplies) from being hoisted into every word
        uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
        asm volatile("" : "+v"(klo), "+v"(khi));
        if ((RM >> lane) & 1) tgt = keyhash_line(tv, u64_of(klo, khi));
    }
}

// All rounds of one word, one after the other. MINI: minimizer table (HEAVY homes exist).
template <bool WIDE, bool CHECK_NS, bool MINI>
__device__ __forceinline__ uint32_t probe_word(const TableView& tv, const K31Fetch& f, int lane, uint64_t key,
                                               uint32_t home, uint64_t V) {
    uint32_t tgt = home;                                    // minimizer tables: shared by the whole minimizer run
    uint32_t st = (uint32_t)((V >> lane) & 1);              // 1 = pending, 2 = found, 0 = absent / invalid
    uint32_t cnt = 0;
    uint32_t meta = 0;
    for (int round = 0;; ++round) {
        const uint64_t PM = wave_ballot(st == 1u);
        if (!PM) break;
        if (MINI && round == 1) heavy_redirect(tv, lane, PM, meta, key, tgt);
        probe_round<WIDE, CHECK_NS>(tv, f, PM, key, tgt, st, cnt, meta);
    }
    return cnt;
}

__device__ __forceinline__ K31Fetch fetch_setup(uint4* lines, uint32_t* runline, int lane) {
    K31Fetch f;
    f.fetch_run = (lane * 57) >> 9;
    f.piece = lane - 9 * f.fetch_run;
    f.fetcher = f.piece < 8 && f.fetch_run < 7;
    f.lines = lines;
    f.runline = runline;
    return f;
}

// The general kernel: any k <= 32, any minimizer length, both line-table modes. A k = 31 line table
// (k = 31, M = 16, minimizer table) runs screen_k31_kernel below instead; KCFX_GENERIC_KERNEL=1 forces this one.
template <bool WIDE, int MODE>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void screen_tiles_kernel(const uint8_t* __restrict__ text,
                                                             const TileDesc* __restrict__ tiles, TableView tv,
                                                             int32_t k, int32_t both_strands, int32_t min_count,
                                                             uint64_t* __restrict__ bits,
                                                             int64_t* __restrict__ count_sums) {
    constexpr bool MINI = (MODE == MODE_MINIMIZER);
    constexpr int LINEBUF_U32 = WAVES * (2 * 64) * 4;      // per wave: 2 DMA instructions x 64 lanes x 16 B
    // s_ha: M-mer hashes, then prefix minima, then (dead) the fetched table lines; s_hb: suffix minima, then minimizers
    constexpr int HMIN = TILE_K + 96;                      // + (wlen - 1) + padding of the last block (wlen <= 32)
    constexpr int HBIG = HMIN > LINEBUF_U32 ? HMIN : LINEBUF_U32;
    constexpr int HLEN = MINI ? HBIG : LINEBUF_U32;
    __shared__ uint32_t s_fwx[2 * (MAX_CHUNKS + 2) + 1];   // [0] is the lead word bits_at may touch
    __shared__ uint32_t s_cwx[2 * (MAX_CHUNKS + 2) + 1];
    __shared__ uint32_t s_vbx[MAX_CHUNKS + 2 + 1];
    uint32_t* const s_fw = s_fwx + 1;
    uint32_t* const s_cw = s_cwx + 1;
    uint32_t* const s_vb = s_vbx + 1;
    __shared__ __attribute__((aligned(16))) uint32_t s_ha[HLEN];
    __shared__ __attribute__((aligned(16))) uint32_t s_hb[MINI ? HMIN : 1];
    __shared__ uint32_t s_runline[WAVES][65];              // line index of each run, per wave (+1 dummy slot)
    __shared__ int64_t s_sum[WAVES];

    const int M = tv.mm_len;
    const int wlen = k - M + 1;                            // M-mers per k-mer
    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;

    // ---- phase 1: ASCII -> 2-bit codes + validity bits in LDS (coalesced 16-byte loads) --------------
    const int64_t a0 = td.text_off & ~(int64_t)31;
    const int off = (int)(td.text_off - a0);
    const int nbases = off + td.n_kmers + k - 1;
    const int nchunks = (nbases + 31) >> 5;
    if (tid == 0) s_fwx[0] = s_cwx[0] = s_vbx[0] = 0;
    for (int c = tid; c < nchunks + 2; c += BLOCK) {
        uint32_t f0 = 0, f1 = 0, v0 = 0, v1 = 0;
        if (c < nchunks) {
            const uint4* p = reinterpret_cast<const uint4*>(text + a0 + 32 * (int64_t)c);
            const uint4 x = p[0], y = p[1];
            pack16(x.x, x.y, x.z, x.w, f0, v0);
            pack16(y.x, y.y, y.z, y.w, f1, v1);
        }
        s_fw[2 * c] = f0;
        s_fw[2 * c + 1] = f1;
        s_cw[2 * c] = comp_swap(f0);
        s_cw[2 * c + 1] = comp_swap(f1);
        s_vb[c] = (v0 << 16) | v1;
    }
    __syncthreads();

    // ---- phase 1b (minimizer mode): h[p] = hash of the canonical M-mer at p, then the sliding minimum over the
    //      wlen M-mers of every k-mer in ~3 min per element (van Herk / Gil-Werman): per block of wlen positions
    //      running prefix minima (in place) and suffix minima; window [p, p+wlen-1] = min(suffix[p], prefix[p+wlen-1]).
    //      The result lands in s_hb; s_ha is dead afterwards and holds the fetched table lines. -----------------------
    const uint32_t* s_fin = s_hb;                          // minimizer (minimum hash) of the k-mer at each position
    uint4* s_line = reinterpret_cast<uint4*>(s_ha);        // [WAVES][2 * 64] fetched lines (LDS-DMA layout, K31Fetch)
    if (MINI) {
        const int n_m = td.n_kmers + wlen - 1;             // every M-mer of every k-mer of the tile
        const int n_blk = (n_m + wlen - 1) / wlen;
        for (int p = tid; p < n_blk * wlen; p += BLOCK) {
            uint32_t h = 0xFFFFFFFFu;                      // padding of the last block
            if (p < n_m) {
                uint64_t f, r;
                extract_pair(s_fw, s_cw, off + p, M, f, r);
                h = mmer_hash(f, r);
            }
            s_ha[p] = h;
        }
        __syncthreads();
        for (int b = tid; b < n_blk; b += BLOCK) {
            const int base = b * wlen;
            uint32_t run = 0xFFFFFFFFu;
            for (int i = wlen - 1; i >= 0; --i) {
                run = min(run, s_ha[base + i]);
                s_hb[base + i] = run;
            }
            run = 0xFFFFFFFFu;
            for (int i = 0; i < wlen; ++i) {
                run = min(run, s_ha[base + i]);
                s_ha[base + i] = run;
            }
        }
        __syncthreads();
        for (int p = tid; p < td.n_kmers; p += BLOCK) s_hb[p] = min(s_hb[p], s_ha[p + wlen - 1]);
        __syncthreads();
    }

    // ---- phase 2: each wave walks a contiguous range of 64-position words ------------------------------
    // the wave index is uniform: say so, or the word loop's counter and bounds live in VGPRs (and spill)
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + WAVES - 1) / WAVES;
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);
    const K31Fetch fetch = fetch_setup(s_line + wave * (2 * 64), s_runline[wave], lane);

    int64_t sum_acc = 0;                                   // per lane
    uint64_t* my_bits = bits + (size_t)blockIdx.x * TILE_BITS_U64;

    for (int w = w_begin; w < w_end; ++w) {
        const int j = w * 64 + lane;
        const int jc = min(j, td.n_kmers - 1);
        uint64_t fwd, rc;
        const bool ok = extract_kmer(s_fw, s_cw, s_vb, off + jc, k, fwd, rc);
        const uint64_t V = wave_ballot(ok) & wave_ballot(j < td.n_kmers);
        // Kmer(Kmer, bothStrands): canonical = min(fwd, revcomp) unsigned (Kmer.java:72-79,406-414)
        const uint64_t key = (both_strands && rc < fwd) ? rc : fwd;
        // first line to look at: the minimizer's (shared by the whole run this k-mer belongs to) or the key's own
        const uint32_t home = MINI ? line_of_minimizer(s_fin[jc], tv.n_lines) : keyhash_line(tv, key);
        const uint32_t cnt = probe_word<WIDE, true, MINI>(tv, fetch, lane, key, home, V);
        // cnt is 0 unless the k-mer was found and min_count >= 1 (checked by the API), so this is "found && count >= -c"
        const bool hit = (int32_t)cnt >= min_count;              // GetVariants.java:224
        const uint64_t H = wave_ballot(hit);
        if (lane == 0) {                                         // this word of the window's hit bitmap
            my_bits[w] = V;
            my_bits[TILE_WORDS + w] = H;
        }
        sum_acc += hit ? (int64_t)(int32_t)cnt : 0;              // GetVariants.java:226
    }

    const int64_t count_sum = wave_sum_i64(sum_acc);
    if (lane == 0) s_sum[wave] = count_sum;
    __syncthreads();
    if (tid == 0) {
        int64_t t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += s_sum[i];
        count_sums[blockIdx.x] = t;
    }
}

// ======================================================================================================
// screen_k31: the k = 31 / M = 16 line-table specialisation of screen_tiles (round 1's hot kernel)
// ======================================================================================================
// Same inputs, outputs and probe semantics as screen_tiles_kernel<*, MODE_MINIMIZER, 31>. SQ counters show the path is
// VALU-bound (profiles/r01_notes.md), so this kernel is about instruction count:
//   * minimizers never touch LDS: with M = 16 a k-mer spans exactly 16 M-mers = one DPP row, so the van Herk prefix /
//     suffix minima are two 4-step v_min_u32 row scans in registers and the sliding minimum of the k-mer at lane l is
//     min(S[l], P[l + 15]) (one ds_bpermute pair; lanes 49..63 take P from the next word). No phase 1b, no barriers.
//   * one word index / shift pair addresses every code-bit read of a word: its own low halves and, 64 positions on, the
//     leading halves of the next word, which feed that word's M-mer hashes now and its k-mers one iteration later.
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_or_ones(uint32_t x) {   // value of the DPP source lane; all ones where there is none
    return (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)x, CTRL, 0xf, 0xf, false);
}
// inclusive prefix (P) and suffix (S) minima within each row of 16 lanes. All ones is the identity of min, which lets
// hipcc fold each step into ONE v_min_u32_dpp -- provided the min sits in the same block as the DPP move: the empty asm
// pins the results here (otherwise the last step, whose result is only needed one word later, is sunk across the loop's
// back edge and stays a v_mov + v_mov_dpp + v_min triple).
__device__ __forceinline__ void row_scan_min(uint32_t h, uint32_t& P, uint32_t& S) {
    P = h;
    P = min(P, dpp_or_ones<0x111>(P));                         // row_shr:1
    P = min(P, dpp_or_ones<0x112>(P));
    P = min(P, dpp_or_ones<0x114>(P));
    P = min(P, dpp_or_ones<0x118>(P));
    S = h;
    S = min(S, dpp_or_ones<0x101>(S));                         // row_shl:1
    S = min(S, dpp_or_ones<0x102>(S));
    S = min(S, dpp_or_ones<0x104>(S));
    S = min(S, dpp_or_ones<0x108>(S));
    asm volatile("" : "+v"(P), "+v"(S));
}
// LDS of the k31 kernels: packed code / validity bits of the tile ...
struct K31Pack {
    // The word loop reads code bits up to 64 positions past the tile's last k-mer without clamping (what it reads there
    // only reaches lanes that are invalid anyway): MAX_CHUNKS + 2 chunks cover every such read of a full tile.
    static_assert(2 * (MAX_CHUNKS + 2) > ((2 * (31 + TILE_K - 1) + 31) >> 5) + 4, "code arrays too short for unclamped reads");
    // code words of both strands, interleaved (.x forward, .y complement-swapped): one 8-byte LDS read serves both
    __attribute__((aligned(8))) uint2 fcx[2 * (MAX_CHUNKS + 2) + 1];   // [0] is the lead word a bit read may touch
    uint32_t vbx[MAX_CHUNKS + 2 + 1];
    uint32_t dirty[WAVES];                                  // per packing wave: some base of its chunks is not A/C/G/T
    int64_t sum[WAVES];
};
// ... + the per-wave line buffers of the line-table kernels
struct K31Shared {
    K31Pack pk;
    __attribute__((aligned(16))) uint4 lines[WAVES][2 * 64]; // DMA layout: line r at 144 r (7 per instruction)
    uint32_t runline[WAVES][65];                            // line index of each run (+1 dummy slot)
};

// phase 1: ASCII -> 2-bit codes + validity bits in LDS (as screen_tiles_kernel); returns the tile's bit offset `off`
template <int NT = BLOCK>
__device__ __forceinline__ int k31_pack_tile(K31Pack& sh, const uint8_t* __restrict__ text, const TileDesc& td, int tid) {
    constexpr int k = 31;
    const int64_t a0 = td.text_off & ~(int64_t)31;
    const int off = (int)(td.text_off - a0);
    const int nbases = off + td.n_kmers + k - 1;
    const int nchunks = (nbases + 31) >> 5;
    uint2* const s_fc = sh.fcx + 1;
    uint32_t* const s_vb = sh.vbx + 1;
    if (tid == 0) {
        sh.fcx[0] = make_uint2(0u, 0u);
        sh.vbx[0] = 0;
    }
    bool bad = false;
    for (int c = tid; c < nchunks + 2; c += NT) {
        uint32_t f0 = 0, f1 = 0, v0 = 0, v1 = 0;
        if (c < nchunks) {
            const uint4* p = reinterpret_cast<const uint4*>(text + a0 + 32 * (int64_t)c);
            const uint4 x = p[0], y = p[1];
            uint32_t d0, d1;
            pack16_codes(x.x, x.y, x.z, x.w, f0, d0);
            pack16_codes(y.x, y.y, y.z, y.w, f1, d1);
            v0 = 0xFFFFu;
            v1 = 0xFFFFu;
            if ((d0 | d1) != 0u) {                              // rare: an N, an IUPAC code, a non-letter in this chunk
                uint32_t g0, g1;
                pack16(x.x, x.y, x.z, x.w, g0, v0);
                pack16(y.x, y.y, y.z, y.w, g1, v1);
                bad = true;
            }
        }
        s_fc[2 * c] = make_uint2(f0, comp_swap(f0));
        s_fc[2 * c + 1] = make_uint2(f1, comp_swap(f1));
        s_vb[c] = (v0 << 16) | v1;
    }
    // every wave writes its own flag (also the waves that packed nothing): no initialisation race
    const uint64_t any_bad = wave_ballot(bad);
    if ((tid & 63) == 0) sh.dirty[tid >> 6] = any_bad != 0;
    __syncthreads();
    return off;
}
// after k31_pack_tile: every base of the tile's chunks is A/C/G/T (then validity is a matter of the tile's length only;
// conservative: the chunks cover a few bases around the tile too)
template <int NW = WAVES>
__device__ __forceinline__ bool k31_tile_clean(const K31Pack& sh) {
    uint32_t d = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) d |= sh.dirty[i];
    return __builtin_amdgcn_readfirstlane((int)d) == 0;
}

// The words of a tile as one wave sees them (line-table kernels): both strands of the k-mer, minimizer and validity ballot of 64
// consecutive positions per step, everything in registers. Carried from one word to the next: the leading code words of both
// strands at this lane's position and the row scans of the M-mer hashes they give. (Cell tables do not come through here: their
// minimizers are the planner's business -- kcfx_plan.h -- whether the reference was planned or is planned tile by tile.)
struct K31Words {
    // Between consecutive words every code-bit address advances by the same amount (64 positions = 4 code dwords of each
    // strand = 2 validity dwords) and the shifts do not change: the addresses are carried, not recomputed.
    const uint2* p_fc;          // entry holding bit 2q - 1 of this lane's position q in the current word
    const uint32_t* p_vb;       // dword holding validity bit q - 1
    uint32_t sh_fc, sh_vb;      // alignbit shifts of the two streams
    int n_kmers, lane, next15;
    bool clean;                 // no invalid base anywhere in the tile: validity = "inside the tile"
    uint32_t Fc, Cc, Pc, Sc;

    // what a position feeds into the sliding minimum, from the leading 32 code bits of both strands: the hash of its canonical
    // 16-mer (its reverse complement: the same bits of the complement-swapped stream, reversed)
    __device__ __forceinline__ uint32_t mmer_of(uint32_t f0, uint32_t c0) const { return mhash32(((uint64_t)min(f0, brev32(c0))) << 32); }
    __device__ __forceinline__ void start(const K31Pack& sh, int off, int n_kmers_, int lane_, int w_begin) {
        n_kmers = n_kmers_;
        lane = lane_;
        clean = k31_tile_clean<WAVES>(sh);
        next15 = ((lane + 15) & 63) << 2;                   // ds_bpermute address of lane + 15 (mod 64)
        const int q = off + w_begin * 64 + lane;
        const int b0 = 2 * q - 1;                            // bits_at(., 2q) on both strands at once
        p_fc = sh.fcx + 1 + (b0 >> 5);
        sh_fc = 31u - (uint32_t)(b0 & 31);
        const int bv = q - 1;                                // bits_at(s_vb, q)
        p_vb = sh.vbx + 1 + (bv >> 5);
        sh_vb = 31u - (uint32_t)(bv & 31);
        const uint2 x = p_fc[0], y = p_fc[1];
        Fc = alignbit32(x.x, y.x, sh_fc);
        Cc = alignbit32(x.y, y.y, sh_fc);
        row_scan_min(mmer_of(Fc, Cc), Pc, Sc);
    }
    // word w (call with consecutive w): the k-mer as the text reads it and its reverse complement (Kmer.java:232-252,
    // :300-338), the minimum over its 16 M-mers, validity
    __device__ __forceinline__ void next(int w, uint64_t& fwd, uint64_t& rc, uint32_t& mn, uint64_t& V) {
        constexpr int k = 31;
        // bits [2q+32, 2q+64) of this word (one entry on) and bits [2(q+64), 2(q+64)+32) of the next (four entries on)
        // share the carried shift
        constexpr uint32_t lo_mask = ~0u << (64 - 2 * k);
        const uint2 x0 = p_fc[1], x1 = p_fc[2], x3 = p_fc[4], x4 = p_fc[5];
        const uint32_t f1 = alignbit32(x0.x, x1.x, sh_fc) & lo_mask;
        const uint32_t c1 = alignbit32(x0.y, x1.y, sh_fc) & lo_mask;
        const uint32_t Fn = alignbit32(x3.x, x4.x, sh_fc);
        const uint32_t Cn = alignbit32(x3.y, x4.y, sh_fc);
        uint32_t Pn = 0, Sn = 0;
        row_scan_min(mmer_of(Fn, Cn), Pn, Sn);
        // extract_kmer (kcfx_device.h) from the four halves
        fwd = u64_of(f1, Fc);
        rc = u64_of(brev32(Cc), brev32(c1)) << (64 - 2 * k);
        // the position is a k-mer of this tile (a scalar mask: only the tile's last word is ragged) and all its k bases
        // are A/C/G/T
        const int rem = n_kmers - w * 64;
        V = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
        if (!clean) V &= wave_ballot((alignbit32(p_vb[0], p_vb[1], sh_vb) >> (32 - k)) == (uint32_t)((1ull << k) - 1));
        // minimum over the k-mer's 16 M-mers = min(S[lane], P[lane + 15])
        const uint32_t pa = (uint32_t)__builtin_amdgcn_ds_bpermute(next15, (int)Pc);
        const uint32_t pb = (uint32_t)__builtin_amdgcn_ds_bpermute(next15, (int)Pn);
        mn = min(Sc, lane < 49 ? pa : pb);
        Fc = Fn;
        Cc = Cn;
        Pc = Pn;
        Sc = Sn;
        p_fc += 4;
        p_vb += 2;
    }
};

// Probe rounds after the first are sparse: an instrumented build counts 13.7 pending lanes per word after round 0 (21 %:
// a minimizer run of up to 16 k-mers overflows its 12-slot home line 40 % of the time, and every k-mer of such a run
// that is not in the home line -- pushed out, or absent from the database -- must look at the next line), yet 91 % of the
// words would run a full-width second round for them. Instead the pending k-mers of successive words are queued,
// compacted, in wave-private LDS and probed 64 at a time; their hits are OR-ed into the wave's hit words afterwards.
constexpr int K31_WORDS_PER_WAVE = (TILE_WORDS + WAVES - 1) / WAVES;
constexpr int K31_QUEUE = 128;                              // < 64 left over + <= 64 new entries
struct K31Queue {
    uint64_t key[K31_QUEUE];
    uint32_t tgt[K31_QUEUE];
    uint32_t id[K31_QUEUE];                                 // (word of the wave << 6) | lane
    unsigned long long hit[K31_WORDS_PER_WAVE];             // hit ballots of the wave's words
};

// One probe round for the first min(qn, 64) queued k-mers; those that must go on are queued again. Returns the new qn.
template <bool WIDE>
__device__ __forceinline__ int k31_deferred_round(const TableView& tv, const K31Fetch& f, K31Queue& q, int lane, int qn,
                                                  int32_t min_count, typename std::conditional<WIDE, int64_t, uint32_t>::type& sum_acc) {
    const int nb = min(qn, 64), rest = qn - nb;
    const uint64_t key = q.key[lane];
    uint32_t tgt = q.tgt[lane];
    const uint32_t id = q.id[lane];
    // the entries behind the batch move to the front
    const uint64_t k2 = q.key[64 + lane];
    const uint32_t t2 = q.tgt[64 + lane], i2 = q.id[64 + lane];
    wave_lds_fence();
    if (lane < rest) {
        q.key[lane] = k2;
        q.tgt[lane] = t2;
        q.id[lane] = i2;
    }
    uint32_t st = lane < nb ? 1u : 0u, cnt = 0, meta = 0;
    const uint64_t PM = nb == 64 ? ~0ull : ((1ull << nb) - 1);
    probe_round<WIDE, false>(tv, f, PM, key, tgt, st, cnt, meta);
    const bool hit = (int32_t)cnt >= min_count;             // cnt is 0 unless found; GetVariants.java:224
    if (hit) atomicOr(&q.hit[id >> 6], 1ull << (id & 63));
    if (WIDE) sum_acc += hit ? (int64_t)(int32_t)cnt : 0;   // GetVariants.java:226
    else sum_acc += hit ? cnt : 0u;
    const uint64_t again = wave_ballot(st == 1u);
    if (again) {
        const int pos = rest + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(again >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)again, 0u));
        if (st == 1u) {
            q.key[pos] = key;
            q.tgt[pos] = tgt;
            q.id[pos] = id;
        }
    }
    wave_lds_fence();
    return rest + __popcll(again);
}

template <bool WIDE, bool BOTH>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void screen_k31_kernel(
    const uint8_t* __restrict__ text, const TileDesc* __restrict__ tiles, TableView tv, int32_t min_count,
    uint64_t* __restrict__ bits, int64_t* __restrict__ count_sums) {
    __shared__ K31Shared sh;
    __shared__ K31Queue s_queue[WAVES];
    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int off = k31_pack_tile(sh.pk, text, td, tid);

    // ---- phase 2: each wave walks a contiguous range of 64-position words ----------------------------------
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + WAVES - 1) / WAVES;
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);
    const K31Fetch fetch = fetch_setup(sh.lines[wave], sh.runline[wave], lane);
    K31Queue& q = s_queue[wave];
    int qn = 0;                                             // queued k-mers (wave-uniform)

    // u16 counters: a wave's total over its <= TILE_WORDS words fits 32 bits; 4-byte counters are Java ints (may be < 0)
    static_assert((uint64_t)TILE_WORDS * 64 * 65535 < (1ull << 32), "32-bit count sums");
    typename std::conditional<WIDE, int64_t, uint32_t>::type sum_acc = 0;   // per lane
    uint64_t* my_bits = bits + (size_t)blockIdx.x * TILE_BITS_U64;
    K31Words words;
    if (w_begin < w_end) words.start(sh.pk, off, td.n_kmers, lane, w_begin);

    for (int w = w_begin; w < w_end; ++w) {
        uint64_t fwd, rc, V;
        uint32_t mn;
        words.next(w, fwd, rc, mn, V);
        // Kmer(Kmer, bothStrands): canonical = min(fwd, revcomp) unsigned (Kmer.java:72-79,406-414)
        const uint64_t key = (BOTH && rc < fwd) ? rc : fwd;
        uint32_t tgt = line_of_minimizer(mn, tv.n_lines);   // home line, shared by the whole minimizer run
        uint32_t st = (uint32_t)((V >> lane) & 1);          // 1 = pending, 2 = found, 0 = absent / invalid
        uint32_t cnt = 0, meta = 0;
        if (V) probe_round<WIDE, false>(tv, fetch, V, key, tgt, st, cnt, meta);      // the home lines
        // cnt is 0 unless the k-mer was found and min_count >= 1 (checked by the API), so this is "found && count >= -c"
        const bool hit = (int32_t)cnt >= min_count;                         // GetVariants.java:224
        const uint64_t H = wave_ballot(hit);
        if (lane == 0) {
            my_bits[w] = V;                                                 // validity word of the window's bitmap
            q.hit[w - w_begin] = H;                                         // hits so far; queued k-mers may add theirs
        }
        if (WIDE) sum_acc += hit ? (int64_t)(int32_t)cnt : 0;               // GetVariants.java:226
        else sum_acc += hit ? cnt : 0u;
        // the rest of this word's probing is queued
        const uint64_t PM = wave_ballot(st == 1u);
        if (PM) {
            heavy_redirect(tv, lane, PM, meta, key, tgt);
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(PM >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)PM, 0u));
            if (st == 1u) {
                q.key[pos] = key;
                q.tgt[pos] = tgt;
                q.id[pos] = (uint32_t)(((w - w_begin) << 6) | lane);
            }
            qn += __popcll(PM);
        }
        wave_lds_fence();
        while (qn >= 64) qn = k31_deferred_round<WIDE>(tv, fetch, q, lane, qn, min_count, sum_acc);
    }
    while (qn > 0) qn = k31_deferred_round<WIDE>(tv, fetch, q, lane, qn, min_count, sum_acc);
    if (lane < w_end - w_begin) my_bits[TILE_WORDS + w_begin + lane] = q.hit[lane];   // hit words of the window's bitmap

    int64_t count_sum;
    if constexpr (WIDE) count_sum = wave_sum_i64(sum_acc);
    else count_sum = (int64_t)wave_sum_u32_dpp(sum_acc);
    if (lane == 0) sh.pk.sum[wave] = count_sum;
    __syncthreads();
    if (tid == 0) {
        int64_t t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += sh.pk.sum[i];
        count_sums[blockIdx.x] = t;
    }
}

// ======================================================================================================
// screen_k31c: the same tile walk against a FORMAT_CELLS table (k = 31, both strands, counters <= 2 bytes)
// ======================================================================================================
// One line per minimizer run as before (same LDS-DMA fetch), but a lane reads exactly ONE cell of its run's line -- the
// cell numbered by the minimizer's offset in the k-mer -- and compares 32 + 16 quotient bits: 3 LDS reads and 3 compares
// where screen_k31 runs 8 x ds_read_b128 and 12 x v_cmp_eq_u64 per round, and no second round over lines: a k-mer whose
// cell is flagged "someone who belongs here lives in the stash" (a few per cent) is queued and looked up in the stash
// 64 at a time, each lane fetching its own 16-byte entry.
constexpr int CELLS_WAVES = CELLS_BLOCK / 64;
constexpr int CELLS_WORDS_PER_WAVE = (TILE_WORDS + CELLS_WAVES - 1) / CELLS_WAVES;
static_assert(CELLS_WAVES >= 1 && CELLS_WAVES <= WAVES && CELLS_WORDS_PER_WAVE <= 64, "cell-table kernel: workgroup shape");
template <int WPW = CELLS_WORDS_PER_WAVE>
struct K31CQueueT {
    uint32_t id[K31_QUEUE];                                 // (word of the wave << 6) | lane
    uint32_t id2[128];                                      // the k-mers their second line sent on (k31c_second_round): third line + stash, once per tile
    unsigned long long hit[WPW];                            // hit ballots of the wave's words
    unsigned long long vw[WPW];                             // their validity ballots (written out with the hits)
};
typedef K31CQueueT<> K31CQueue;
// A text-only reference (K1 - K3 in one launch) is screened by ONE wave per tile: the wave plans the tile -- a strip of 32
// positions per lane, neighbours through wave-private LDS rows, fences instead of barriers: ref_plan_kernel's form of the
// planner, 28 VALU per position where two waves of 16-position strips spend 33 -- and then walks all of its words. No barrier
// anywhere in the launch; a wave's fixed costs are paid once per 32 words. Lane 63's strip only lends to lane 62:
// 63 x 32 positions, up to 15 of them alignment in front of the tile.
constexpr int TEXT_BLOCK = 64;
#ifndef KCFX_TEXT_WAVES_MIN
#define KCFX_TEXT_WAVES_MIN 7
#endif
static_assert(TEXT_TILE_K + 15 <= 32 * 63 && TEXT_TILE_K + 15 + 30 <= 32 * 64 && TEXT_TILE_K <= TILE_K, "text-only tiles");

// The probe of a word is split in two so that the cell loads of word w + 1 are in flight while word w is matched: with
// ~65 VALU instructions per word the kernel is no longer issue-bound, and a wave that waits for its own loads before it
// derives the next word leaves the memory system well below the request rate it can sustain (8 waves per SIMD is the cap).
// Every lane loads its own 8-byte cell; the lanes of a minimizer run aim at the same 128-byte line, which goes to HBM once.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));   // a native vector (HIP's uint2 is a union-based struct)
typedef uint32_t u32x4n __attribute__((ext_vector_type(4)));
// Table cells, second-line cells, stash entries and cohort counters are read ONCE per pass: loaded with the non-temporal hint
// (`nt` on the instruction) they do not push the lines that ARE read again -- the reference plan, the tile bitmaps -- out of the L2
// and the Infinity Cache. Same-box A/B on the bench workload: 0.0808 - 0.0845 ms against 0.0879 - 0.0918 ms per launch.
__device__ __forceinline__ u32x2 nt_load2(const void* p) {
    typedef const __attribute__((address_space(1))) u32x2* gp;
    return __builtin_nontemporal_load((gp)(uint64_t)p);
}
__device__ __forceinline__ u32x4n nt_load4(const void* p) {
    typedef const __attribute__((address_space(1))) u32x4n* gp;
    return __builtin_nontemporal_load((gp)(uint64_t)p);
}

struct CellWord {
    uint64_t V;            // valid lanes (a table slice: whose minimizer this slice holds)
    uint64_t Vfull;        // valid lanes whatever the slice: the validity word of the bitmap
    uint32_t w, r;         // what this lane's cell must hold (CellProbe)
    uint32_t anc;          // low five bits: 2 * cell index | (N is the reverse complement) -- what a queued k-mer needs to derive its probe again
    uint32_t r_alt;        // (k <= 30 only) what the r field of its free-cell entry must be (cell_alt_of)
    u32x2 cell, alt;       // what its first-choice cell and (k <= 30 only) its free cell hold
};
constexpr uint32_t QUEUE_ANCHOR_SHIFT = 16;                 // queue id = (word of the wave << 6) | lane | anc << 16

// stage A: the load of every lane's cell (k <= 30: and of its free cell in the same line), issued and not waited for. The same
// load instructions for every word, every lane active (an invalid lane reads line 0): the compiler can then count what is
// in flight -- s_waitcnt vmcnt(n) where word w is consumed while word w + 1's loads stay out -- instead of falling back to
// vmcnt(0) as it must around a conditional load.
template <int W, bool BOTH>
__device__ __forceinline__ void cells_issue(const TableView& tv, uint64_t V, const CellProbe& p, uint32_t anc, CellWord& cw) {
    const bool valid = __builtin_amdgcn_inverse_ballot_w64(V);
    cw.V = V;
    cw.w = p.w;
    cw.r = p.r;
    cw.anc = anc;
    uint32_t line = valid ? p.line : 0u;
    asm volatile("" : "+v"(line));                              // (select the 32-bit line number, not the 64-bit byte offset after the shift)
    // lines are 128-byte aligned (and so is the table): the cell's offset in the line (8 * cell = 4 * sh) is OR-ed into the low
    // dword of the line's byte offset: base + ((32 * line + sh) << 2) = v_lshlrev_b64, v_or, v_lshl_add_u64 (whose shift goes up to 4)
    // (the two halves of 32 * line + sh separately -- v_lshl_or_b32, v_lshrrev_b32 -- not a 64-bit shift of a zero-extended line:
    // one instruction less)
    uint64_t l32 = u64_of((line << 5) | p.sh, line >> 27);
    asm volatile("" : "+v"(l32));                               // (or the compiler re-associates it into a shift by 7 and two 64-bit adds)
    const uint8_t* a0 = reinterpret_cast<const uint8_t*>(reinterpret_cast<const uint32_t*>(tv.lines) + l32);
    // (nt_load2 goes through a global-address-space pointer: through a generic one the load could become a flat_load, which the
    // compiler cannot count in vmcnt and waits for with vmcnt(0) -- the end of the software pipeline)
    // (k = 31 only: with a second load of the same line behind it -- the free cell of k < 31 -- the hint costs that load its hit:
    // k = 25, 0.131 ms with it, 0.121 ms without)
    // (round 5 tried the ordinary policy for the lanes next to a word border -- where a minimizer run straddles two words and its
    // line is wanted twice -- so that the second word would find it in the L2: 0.0835 against 0.0801 ms, profiles/r05_notes.md)
    if constexpr (W > CELL_ALT_FREE_MAX_W) cw.cell = nt_load2(a0);
    else cw.cell = *reinterpret_cast<const u32x2*>(a0);
    if constexpr (W <= CELL_ALT_FREE_MAX_W) {
        // (on a table without the free-cell scheme this reads some cell of the line and cells_match ignores it)
        const CellAlt a = cell_alt_of(p.sh, p.w, p.r, W, tv.lbits, BOTH);
        cw.r_alt = a.r;
        cw.alt = *reinterpret_cast<const u32x2*>(tv.lines + (((uint64_t)line << 7) | (uint64_t)(a.sh << 2)));
    }
}

// stage B: cnt = the stored counter if the cell (k <= 30: or the free cell) holds this lane's k-mer (else 0); pend != 0: the cell
// is flagged "someone who belongs here lives elsewhere" and this flank's filter bit is set (if cnt == 0 the k-mer is queued
// and looked up in its second line, then in the stash)
template <bool BOTH, int W>
__device__ __forceinline__ void cells_match(const CellWord& cw, int32_t lbits, uint32_t& cnt, uint32_t& pend, uint64_t* matched = nullptr) {
    const uint32_t rmask = cell_rmask(lbits, BOTH);             // (wave-uniform: scalar registers)
    // (cell_filter_pos with its two wave-uniform terms pinned to scalar registers: one v_add instead of a sub + add pair)
    uint32_t rb = (uint32_t)(32 - lbits);
    asm volatile("" : "+s"(rb));
    const uint32_t nspare = cell_spare_bits(lbits, BOTH);
    // a table without filter bits sends every miss in a flagged cell to the stash: OR-ed in as a scalar, so that the word loop
    // has neither a branch nor a materialised predicate for it
    const uint32_t unfiltered = nspare > 0 ? 0u : 1u;
    // (every lane, valid or not -- an invalid one has loaded cell 0 of line 0: the caller masks the ballots with V, which costs
    // scalar instructions only, where a predicated region costs the moves that initialise cnt and pend around it)
    {
        const uint32_t x = cw.cell.x, y = cw.cell.y;
        // (cw.r has no bits outside rmask and bit 15 clear; xor + and fuse into one v_bitop3)
        // (an invalid lane has loaded cell 0 of line 0 and compared it with whatever its bases encode: never a match -- the
        // validity mask is AND-ed to the compares' on the scalar unit)
        // (ballots AND-ed, not `&&`: both compares for every lane, no exec juggling for a short circuit)
        // (d: how the cell's flank differs from this lane's, flags aside -- one v_bitop3; zero = the same flank; also the filter's key)
        const uint32_t d = (x & ~CELL_FLAG_MASK) ^ cw.w;
        const uint64_t FM = cw.V & wave_ballot(d == 0u) & wave_ballot(((y ^ cw.r) & rmask) == 0);
        const bool first = __builtin_amdgcn_inverse_ballot_w64(FM);
        if (matched) *matched = FM;                         // (k <= 30 with the retired free-cell scheme: first-choice matches only)
        if constexpr (W <= CELL_ALT_FREE_MAX_W) {
            const bool free_scheme = cell_alt_free(W, lbits, BOTH);                              // (wave-uniform)
            const uint32_t rmask_alt = rmask | (0xFu << (32 - lbits));
            const uint32_t xa = cw.alt.x, ya = cw.alt.y;        // (only loaded in builds with the free-cell scheme)
            const bool second = free_scheme && ((xa ^ cw.w) <= CELL_FLAG_MASK) && (((ya ^ cw.r_alt) & rmask_alt) == 0);
            cnt = first ? (y >> 16) : second ? (ya >> 16) : 0u;
        } else {
            cnt = first ? (y >> 16) : 0u;
        }
        // the stash only if the cell is flagged AND (when the table has filter bits) the bit of this flank is set.
        // "cnt == 0", not "no match": an EMPTY cell can be flagged (a k-mer whose counter does not fit 16 bits never takes
        // a cell) and would "match" the k-mer whose w and r are all zero
        // (cell_filter_pos: the bit is chosen by how this flank differs from the occupant's -- d, cell_filter_key)
        const uint32_t mine = (y >> (rb + __umulhi(d * 0x9E3779B1u, nspare))) | unfiltered;
        pend = x & mine & CELL_FLAG_STASH;
    }
}

// ---- the planned reference ------------------------------------------------------------------------------------------
// Which 16-mer of a reference k-mer is its minimizer, on which strand it is canonical and whether all k bases are A/C/G/T
// does not depend on the sample: kcfx_ref_create works it out ONCE per reference (ref_plan_kernel: StripPlanner, kcfx_plan.h)
// and keeps, beside the text,
//   codes    2 bits per base, 16 bases per dword, first base in the top bits (what the pack phase used to rebuild in LDS
//            for every tile of every launch), and
//   anchors  one byte per k-mer start position: bit 7 = the k-mer is valid, bits 4..1 = 2 * (the cell index: the minimizer's
//            offset in N), bit 0 = N is the reverse complement of the text's k-mer (cell_probe_of's sh and flip).
// A screening launch then copies a tile's codes and anchors into LDS and starts at the flank: no ASCII, no hashes, no scans,
// no cross-lane traffic in the word loop -- ~30 VALU instructions per word instead of 53 before the cell loads are issued,
// and a minimizer ORDER of any cost (kcfx_device.h: cell_order27) is free for the launches that follow.
// In LDS the tile's codes stand TWICE (round 5): as the text reads them (fw) and as the reverse complement of the whole tile
// (rw: word j = brev(comp_swap(word NW - 1 - j))), so that the reverse complement of the k-mer at packed position q is the
// window of rw at T - q - K (T = 16 NW bases) and both strands are read the same way: three dwords, two v_alignbit. The anchor
// byte says which strand a lane needs, so a lane reads ONE of them (v_bfi of the two LDS addresses) where round 4 read a pair
// of streams and built both k-mers -- 9 VALU per word instead of 14, half the LDS bytes.
struct K31Plan {
    static constexpr int NW = 2 * (MAX_CHUNKS + 2);         // code words: 16 NW bases from the tile's aligned start
    static constexpr int T = 16 * NW;
    static_assert(NW > ((2 * (31 + TILE_K - 1) + 31) >> 5) + 4, "code arrays too short for unclamped reads");
    static_assert(31 + TILE_K - 1 + CELL_K_MAX + 1 <= T, "every k-mer's reverse complement starts at a position >= 1 of rw");
    __attribute__((aligned(16))) uint32_t fw[1 + NW + 3];   // [0] is the lead word (position 0 reads bit -1); fw[1 + c] = code word c
    __attribute__((aligned(16))) uint32_t rw[NW + 4];       // the reverse complement of the tile, mirrored (+ the word read past it)
    __attribute__((aligned(16))) uint4 an[(32 + TILE_K + 15) / 16 + 1]; // anchor bytes of the tile from its 32-aligned start
    int64_t sum[WAVES];
    __device__ __forceinline__ void put_code(int c, uint32_t code) {       // code word c of the tile, both ways
        fw[1 + c] = code;
        rw[NW - 1 - c] = brev32(comp_swap(code));
    }
};

// phase 1 of a planned launch: codes (both ways) and anchor bytes of the tile into LDS; returns `off`
template <int NT>
__device__ __forceinline__ int k31_load_plan_tile(K31Plan& sh, const RefPlan& plan, const TileDesc& td, int k, int tid) {
    const int64_t a0 = td.text_off & ~(int64_t)31;
    const int off = (int)(td.text_off - a0);
    const int nbases = off + td.n_kmers + k - 1;
    const int nchunks = (nbases + 31) >> 5;
    if (tid == 0) sh.fw[0] = 0u;
    const uint2* const codes = reinterpret_cast<const uint2*>(plan.codes) + (a0 >> 5);       // 32 bases = 8 bytes
    for (int c = tid; c < nchunks + 2; c += NT) {
        const uint2 x = codes[c];
        sh.put_code(2 * c, x.x);
        sh.put_code(2 * c + 1, x.y);
    }
    const uint4* const an = reinterpret_cast<const uint4*>(plan.anchors + a0);
    const int n16 = (off + td.n_kmers + 15) >> 4;
    for (int c = tid; c < n16; c += NT) sh.an[c] = an[c];
    __syncthreads();
    return off;
}

// ---- the planner on the device: one strip per lane, exchanges through wave-private LDS ------------------------------
// A wave plans 64 consecutive strips; what a strip needs of its right-hand neighbour travels through a row of 16 dwords per
// lane (the first code words: dwords 0..2; the first 10 hashes: 4..13; then, over both, the first W - 1 tagged keys). The lanes
// of a wave execute their LDS instructions in order, so a compiler fence between a row's write and the neighbour's read is
// all the synchronisation there is. Lane 63 has no neighbour in the wave: its results are not used, the next wave (or the next
// workgroup) starts at its strip again -- 1 / 64 of the work, against a barrier per exchange.
// (BLOCKWIDE: the strips of a whole workgroup are neighbours -- thread t's right-hand neighbour is thread t + 1, across the
// wave boundary too -- and a barrier stands where the wave-private form has a fence: the text-only screening kernels, whose
// tile is ONE run of strips.)
constexpr int PLAN_ROWS = 66;                               // 64 lanes + the rows lane 63 and 62 read past the end (garbage, unused)
// (rows 64 bytes apart: every fourth lane's 16-byte piece falls on the same four banks, a four-way conflict on each b128 access of the
// exchanges -- 5.9 M conflict cycles per launch of the one-wave-per-tile kernel. Rows of 80 bytes have none, and were slower: 0.0911
// against 0.0884 ms -- the 5.3 KB of LDS per wave cost more occupancy than the conflicts cost time; profiles/r05_notes.md)
typedef uint32_t PlanRows[PLAN_ROWS][16];
typedef uint32_t PlanRowsBlock[CELLS_BLOCK + 2][16];
template <int K, int L, bool BLOCKWIDE = false>
struct PlanWaveIo {
    using P = StripPlanner<K, L>;
    uint32_t (*rows)[16];                                   // this wave's rows (BLOCKWIDE: the workgroup's)
    int lane;                                               // (BLOCKWIDE: the thread index)
    static __device__ __forceinline__ void sync() {
        if constexpr (BLOCKWIDE) __syncthreads();
        else wave_lds_fence();
    }
    const uint8_t* text;                                    // the strip's first base (16-byte aligned); null: behind the readable text -> zeros
    __device__ __forceinline__ void load_text(uint32_t (&x)[L / 4]) const {
#pragma unroll
        for (int i = 0; i < L / 16; ++i) {
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (text) v = reinterpret_cast<const uint4*>(text)[i];
            x[4 * i] = v.x;
            x[4 * i + 1] = v.y;
            x[4 * i + 2] = v.z;
            x[4 * i + 3] = v.w;
        }
    }
    __device__ __forceinline__ void exchange_first(const typename P::First& mine, typename P::First (&nb)[P::VN]) const {
        *reinterpret_cast<uint4*>(rows[lane]) = make_uint4(mine.F0, mine.G0, mine.valid, 0u);
        sync();
#pragma unroll
        for (int i = 0; i < P::VN; ++i) {
            const uint4 v = *reinterpret_cast<const uint4*>(rows[lane + 1 + i]);
            nb[i].F0 = v.x;
            nb[i].G0 = v.y;
            nb[i].valid = v.z;
        }
    }
    __device__ __forceinline__ void exchange_hashes(const uint32_t (&u)[L + P::UH], uint32_t (&uh)[P::UH]) const {
        static_assert(P::UH == 10, "row layout");
        uint4* mine = reinterpret_cast<uint4*>(rows[lane]);
        mine[1] = make_uint4(u[0], u[1], u[2], u[3]);
        mine[2] = make_uint4(u[4], u[5], u[6], u[7]);
        *reinterpret_cast<uint2*>(mine + 3) = make_uint2(u[8], u[9]);
        sync();
        const uint4* theirs = reinterpret_cast<const uint4*>(rows[lane + 1]);
        const uint4 a = theirs[1], b = theirs[2];
        const uint2 c = *reinterpret_cast<const uint2*>(theirs + 3);
        uh[0] = a.x; uh[1] = a.y; uh[2] = a.z; uh[3] = a.w;
        uh[4] = b.x; uh[5] = b.y; uh[6] = b.z; uh[7] = b.w;
        uh[8] = c.x; uh[9] = c.y;
    }
    __device__ __forceinline__ void exchange_keys(const uint32_t (&t)[L + P::TH], uint32_t (&th)[P::TH]) const {
        sync();                                   // (every lane has read its neighbour's hashes: the rows are free)
        uint4* mine = reinterpret_cast<uint4*>(rows[lane]);
#pragma unroll
        for (int i = 0; i < (P::TH + 3) / 4; ++i)           // (t has at least L >= 16 entries: whole uint4s)
            mine[i] = make_uint4(t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]);
        sync();
        const uint4* theirs = reinterpret_cast<const uint4*>(rows[lane + 1]);
#pragma unroll
        for (int i = 0; i < (P::TH + 3) / 4; ++i) {
            const uint4 v = theirs[i];
            if (4 * i < P::TH) th[4 * i] = v.x;
            if (4 * i + 1 < P::TH) th[4 * i + 1] = v.y;
            if (4 * i + 2 < P::TH) th[4 * i + 2] = v.z;
            if (4 * i + 3 < P::TH) th[4 * i + 3] = v.w;
        }
        sync();                                   // (the rows may be reused by the caller)
    }
    // every base this WAVE looked at is A/C/G/T: the valid-start masks are all ones (wave-uniform, so a branch costs nothing).
    // (The last VN strips read rows behind the last one written -- their results are not used and neither is their vote.)
    __device__ __forceinline__ bool all_clean(bool mine) const {
        return wave_ballot(!mine && lane < (BLOCKWIDE ? CELLS_BLOCK : 64) - P::VN) == 0ull;
    }
};

// phase 1 of a launch on a text-only reference (KCFX_REF_PLAN=0, or kcfx_screen with a one-shot text): the tile is planned
// here, straight into the LDS layout a planned reference is copied into -- K1 to K3 in one launch. The workgroup's threads
// plan one strip of 16 positions each from the tile's 16-aligned start, neighbours across the wave boundary (PlanWaveIo,
// BLOCKWIDE). The last TWO strips depend on neighbours that do not exist (the last one's keys need hashes of bases behind it, the
// one before needs the last one's keys): anchors for 16 (CELLS_BLOCK - 2) positions, codes for 16 CELLS_BLOCK bases. Such a
// reference is therefore cut into tiles of at most TEXT_TILE_K k-mers (kcfx_ref_create): off <= 15 in front, and the last
// k-mer's 30 bases behind it must have codes.
static_assert(TEXT_TILE_K + 15 <= 16 * (CELLS_BLOCK - 2) && TEXT_TILE_K + 15 + 30 <= 16 * CELLS_BLOCK && TEXT_TILE_K <= TILE_K, "text-only tiles");
template <int K>
__device__ __forceinline__ int k31_plan_tile(K31Plan& sh, PlanRowsBlock& rows, const uint8_t* __restrict__ text, const TileDesc& td, int tid) {
    using P = StripPlanner<K, 16>;
    const int64_t a0 = td.text_off & ~(int64_t)15;
    const int off = (int)(td.text_off - a0);
    // bases the tile's k-mers, their 16-mers' keys and those keys' 6-mers reach: nothing behind them is read (the text is
    // readable 128 bytes past its end, a tile ends at most there minus K - 1)
    const int need = off + td.n_kmers + 30;
    PlanWaveIo<K, 16, true> io{rows, tid, 16 * tid < need ? text + a0 + 16 * (int64_t)tid : nullptr};
    const typename P::Result res = P::run(io);
    if (tid == 0) sh.fw[0] = 0u;
    sh.put_code(tid, res.codes[0]);
    if (tid < CELLS_BLOCK - 2) sh.an[tid] = make_uint4(res.anchors[0], res.anchors[1], res.anchors[2], res.anchors[3]);
    __syncthreads();
    return off;
}

// The same for the ONE wave that screens a text-only tile on its own (screen_k31c_kernel, PLANNED = false): a strip of 32 positions
// per lane from the tile's 16-aligned start, wave-private rows, fences. `rows` may overlap `sh` and the queues: every lane has
// its results in registers before the first of them is written (one wave: its LDS instructions execute in order).
template <int K>
__device__ __forceinline__ int k31_plan_tile_wave(K31Plan& sh, PlanRows& rows, const uint8_t* __restrict__ text, const TileDesc& td, int lane) {
    using P = StripPlanner<K, 32>;
    const int64_t a0 = td.text_off & ~(int64_t)15;
    const int off = (int)(td.text_off - a0);
    // bases the tile's k-mers, their 16-mers' keys and those keys' 6-mers reach: nothing behind them is read (the text is
    // readable 128 bytes past its end, a tile ends at most there minus K - 1)
    const int need = off + td.n_kmers + 30;
    PlanWaveIo<K, 32, false> io{rows, lane, 32 * lane < need ? text + a0 + 32 * (int64_t)lane : nullptr};
    const typename P::Result res = P::run(io);
    wave_lds_fence();
    if (lane == 0) sh.fw[0] = 0u;
    sh.put_code(2 * lane, res.codes[0]);
    sh.put_code(2 * lane + 1, res.codes[1]);
    if (lane < 63) {
        sh.an[2 * lane] = make_uint4(res.anchors[0], res.anchors[1], res.anchors[2], res.anchors[3]);
        sh.an[2 * lane + 1] = make_uint4(res.anchors[4], res.anchors[5], res.anchors[6], res.anchors[7]);
    }
    wave_lds_fence();
    return off;
}

// The words of a tile from a planned reference: N (the strand of the k-mer on which its minimizer is canonical), 2 * the
// minimizer's offset there, the flip mask, validity. Nothing is carried between words but two LDS addresses.
template <int K>
struct PlannedWords {
    // byte offsets from the start of the tile's LDS block (K31Plan) of the dword holding bit 2q - 1 of this lane's position q in
    // the current word, as the text reads it / of bit 2q' - 1 in the mirrored reverse complement (q' = T - q - K); the shifts
    // that go with them do not change from word to word (a word is 128 bits on)
    uint32_t a_f, a_r;          // (LDS addresses, whole: the three reads of a word then need no address arithmetic)
    uint32_t s_f, s_r;
    const int8_t* p_an;         // this lane's anchor byte
    int n_kmers;
    typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
    static __device__ __forceinline__ uint32_t lds_address(const uint32_t* p) { return (uint32_t)(uintptr_t)(lds_u32)p; }
    __device__ __forceinline__ void start(const K31Plan& sh, int off, int n_kmers_, int lane, int w_begin) {
        n_kmers = n_kmers_;
        const int q = off + w_begin * 64 + lane;
        const int b0 = 2 * q - 1;
        a_f = lds_address(sh.fw + 1 + (b0 >> 5));
        s_f = 31u - (uint32_t)(b0 & 31);
        const int b1 = 2 * (K31Plan::T - q - K) - 1;
        a_r = lds_address(sh.rw + (b1 >> 5));
        s_r = 31u - (uint32_t)(b1 & 31);
        p_an = reinterpret_cast<const int8_t*>(sh.an) + q;
    }
    // (*anchor: the anchor byte as loaded, sign-extended -- its low five bits are what a queued k-mer keeps: CellWord::anc)
    __device__ __forceinline__ void next(int w, uint64_t& N, uint32_t& sh2, uint32_t& fm, uint64_t& V, uint32_t* anchor = nullptr) {
        constexpr uint32_t lo_mask = ~0u << (64 - 2 * K);
        const int32_t a = (int32_t)p_an[0];                  // sign-extended: valid <=> negative
        asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(fm) : "v"(a));  // all ones: N is the reverse complement
        // the k-mer as the text reads it (Kmer.java:232-252) or its reverse complement (:300-338): whichever has the minimizer
        // in canonical form -- one stream per lane
        const uint32_t at = (a_f & ~fm) | (a_r & fm);        // v_bfi_b32
        const uint32_t sx = (s_f & ~fm) | (s_r & fm);
        const lds_u32 p = (lds_u32)(uintptr_t)at;
        const uint32_t y0 = p[0], y1 = p[1], y2 = p[2];
        N = u64_of(alignbit32(y1, y2, sx) & lo_mask, alignbit32(y0, y1, sx));
        sh2 = (uint32_t)a & 30u;
        if (anchor) *anchor = (uint32_t)a;
        const int rem = n_kmers - w * 64;
        V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & wave_ballot(a < 0);
        a_f += 16u;
        a_r -= 16u;
        p_an += 64;
    }
};

// cell_probe_of when the anchor byte has said where the minimizer sits (sh2 = 2 * its offset in N, fm = the flip mask)
template <bool BOTH>
__device__ __forceinline__ CellProbe cell_probe_planned(uint64_t N, uint32_t sh2, uint32_t fm, int32_t lbits, int32_t ext, uint32_t* g_out = nullptr) {
    CellProbe p;
    uint32_t cm;
    p.w = cell_flank(N, sh2, 16, &cm);
    uint32_t g = cell_g32(cm);
    if (ext) g = cell_g32_ext(g, p.w, sh2);                 // (wave-uniform: an extended table, cell_first_ext)
    if (g_out) *g_out = g;                                  // (line : r in one word: what regions and slices hash)
    p.sh = sh2;
    p.line = g >> (32 - lbits);
    p.r = (g & (0xFFFFFFFFu >> lbits)) | (BOTH ? 0u : (fm & CELL_R_FLIP));
    p.cls = 0;                                              // (the anchor byte carries it: anchor_slice_class)
    return p;
}

// A queued k-mer, derived again from the packed tile (the queue holds where it sits and its anchor bits): its probe and the
// word the database stores it under
template <int K, bool BOTH, class Lds>
__device__ __forceinline__ CellProbe k31_probe_at(const Lds& sh, int q, uint32_t anc, int32_t lbits, int32_t ext, uint64_t& key) {
    constexpr uint32_t lo_mask = ~0u << (64 - 2 * K);
    const int b0 = 2 * q - 1, b1 = 2 * (Lds::T - q - K) - 1;
    const uint32_t* pf = sh.fw + 1 + (b0 >> 5);
    const uint32_t* pr = sh.rw + (b1 >> 5);
    const uint32_t sf = 31u - (uint32_t)(b0 & 31), sr = 31u - (uint32_t)(b1 & 31);
    const uint64_t fwd = u64_of(alignbit32(pf[1], pf[2], sf) & lo_mask, alignbit32(pf[0], pf[1], sf));
    const uint64_t rc = u64_of(alignbit32(pr[1], pr[2], sr) & lo_mask, alignbit32(pr[0], pr[1], sr));
    key = (BOTH && rc < fwd) ? rc : fwd;                    // canonical (Kmer.java:72-79), or as the text reads it (:57-65)
    const uint32_t fm = 0u - (anc & 1u);
    return cell_probe_planned<BOTH>(fm ? rc : fwd, anc & 30u, fm, lbits, ext);
}

#ifdef KCFX_EXP_STATS
// experiment build only (make fast FASTFLAGS=-DKCFX_EXP_STATS): what the queued look-ups of a pass are made of
// [0] k-mers queued  [1] found in their second line  [2] sent on from it  [3] found in their third line  [4] looked up in the stash
// [5] found there  [6] second-line requests (lanes whose line differs from the lane before)  [7] words with a queued k-mer
// [8] k-mers found in their first-choice cell  [9] deferred rounds
__device__ unsigned long long g_exp[16];
extern "C" int32_t kcfx_exp_stats(unsigned long long* out, int32_t reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_exp), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) {
        const unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_exp), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#define EXP_COUNT(i, pred) do { if (pred) atomicAdd(&g_exp[i], 1ull); } while (0)
#define EXP_ROUND(i, pred) do { if constexpr (EXPS) { if (pred) atomicAdd(&g_exp[i], 1ull); } } while (0)
#else
#define EXP_COUNT(i, pred) do { } while (0)
#define EXP_ROUND(i, pred) do { } while (0)
#endif
// The first min(qn, 64) queued k-mers -- their first-choice cell said "someone who belongs here lives elsewhere", with their filter
// bit: the same cell of their second line (the losers of a minimizer sit there together: neighbouring lanes, one line), and if
// that cell sends them on too, the stash. Calls found(id, counter) for the lanes that find theirs; returns the new qn.
template <int K, bool BOTH, bool EXPS = false, class Lds, class Found>
__device__ __forceinline__ int k31c_deferred_round(const TableView& tv, const Lds& sh, uint32_t* ids, int lane, int qn, int pos0, Found&& found) {
    const int nb = min(qn, 64), rest = qn - nb;
    const uint32_t id = ids[lane];
    const uint32_t i2 = ids[64 + lane];
    wave_lds_fence();
    if (lane < rest) ids[lane] = i2;                        // the entries behind the batch move to the front
    if (lane < nb) {
        uint64_t key;
        const CellProbe p = k31_probe_at<K, BOTH>(sh, pos0 + (int)(id & 0xFFFFu), id >> QUEUE_ANCHOR_SHIFT, tv.lbits, tv.ext, key);
        const CellProbe q2 = cell_second_line(p, tv.lbits);
        const size_t cell = (size_t)cell_line_index(q2.line, q2.r, tv.lbits, tv.regions) * 16 + (q2.sh >> 1);
        const uint2 e = reinterpret_cast<const uint2*>(tv.lines)[cell];      // (plain: with the non-temporal hint this one is slower)
        const uint32_t rmask = cell_rmask(tv.lbits, BOTH);
        const uint32_t nspare = cell_spare_bits(tv.lbits, BOTH);
        // a cell that does not hold the k-mer sends it on if it is flagged and (a table with filter bits) the k-mer's bit there is
        // set: w is the same in every line of the k-mer, its bit in a cell depends on that cell's occupant (cell_filter_pos).
        // (level after level, no nesting and no short circuits: the structurizer made a loop with spilled masks of the nested form)
        auto holds = [&](const uint2& c, const CellProbe& q) {
            return (((c.x ^ q.w) <= CELL_FLAG_MASK) & (((c.y ^ q.r) & rmask) == 0) & ((c.y >> 16) != 0u)) != 0;
        };
        auto sends_on = [&](const uint2& c, const CellProbe& q) {
            const uint32_t bit = nspare == 0 ? 1u : (c.y >> cell_filter_pos(q.w, c.x, tv.lbits, BOTH)) & 1u;
            return (c.x & CELL_FLAG_STASH & bit) != 0u;                 // (CELL_FLAG_STASH is bit 0)
        };
        const bool hit2 = holds(e, q2);
        if (hit2) found(id, e.y >> 16, (uint32_t)cell, false);
        bool on = !hit2 && sends_on(e, q2);
#ifdef KCFX_EXP_STATS
        if constexpr (EXPS) {
            const uint32_t l2 = cell_line_index(q2.line, q2.r, tv.lbits, tv.regions);
            const uint32_t prev = __shfl_up(l2, 1);
            EXP_ROUND(0, true);
            EXP_ROUND(1, hit2);
            EXP_ROUND(2, on);
            EXP_ROUND(6, lane == 0 || prev != l2);
            EXP_ROUND(9, lane == 0);
        }
#endif
#if defined(KCFX_EXP_CUT) && KCFX_EXP_CUT <= 1
        on = false;                                         // (experiment build: nothing beyond the second line)
#endif
        if (cell_has_third(tv.lbits, BOTH)) {                // (wave-uniform)
            if (on) {
                // its third line: the losers of this minimizer's other loci are not here
                const CellProbe q3 = cell_third_line(p, tv.lbits, BOTH, cell_nw(K));
                const size_t cell3 = (size_t)cell_line_index(q3.line, q3.r, tv.lbits, tv.regions) * 16 + (q3.sh >> 1);
                const uint2 e3 = reinterpret_cast<const uint2*>(tv.lines)[cell3];
                const bool hit3 = holds(e3, q3);
                if (hit3) found(id, e3.y >> 16, (uint32_t)cell3, false);
                on = !hit3 && sends_on(e3, q3);
                EXP_ROUND(3, hit3);
            }
        }
#if defined(KCFX_EXP_CUT) && KCFX_EXP_CUT <= 2
        on = false;                                         // (experiment build: no stash)
#endif
        if (on) {
            const uint32_t slot = stash_find_slot(tv, key);
            if (slot != ~0u) found(id, tv.stash[slot].z, slot, true);
            EXP_ROUND(4, true);
            EXP_ROUND(5, slot != ~0u);
        }
    }
    wave_lds_fence();
    return rest;
}

// Round 6: the same look-ups in two stages. A wave that walks a queued batch through second line -> third line -> stash waits
// for three to five dependent loads in a row with nothing else of its own in flight, and with one wave per tile (the text-only
// form) that chain is a fifth of the wave's life at 0.28 k-mers per cell: experiment builds that stop after the second line run
// the one-pass kernel in 0.080 ms instead of 0.0875 - 0.090, although third lines and stash are 5 % of the requests
// (profiles/r06_notes.md section 9). So a batch only takes its SECOND-LINE step here; the lanes that are sent on (a fifth of them)
// are written to a second queue, and that queue is walked ONCE per tile (k31c_rest_round): one third-line load for all of them,
// then the stash for those still sent on, two slots per lane at once. Same cells, same order per k-mer, same results.
// Returns the new qn; qn2 = entries in ids2 (<= 128: the caller empties it before a batch of 64 could overflow it).
template <int K, bool BOTH, class Lds, class Found>
__device__ __forceinline__ int k31c_second_round(const TableView& tv, const Lds& sh, uint32_t* ids, uint32_t* ids2, int& qn2, int lane, int qn, int pos0, Found&& found) {
    const int nb = min(qn, 64), rest = qn - nb;
    const uint32_t id = ids[lane];
    const uint32_t i2 = ids[64 + lane];
    wave_lds_fence();
    if (lane < rest) ids[lane] = i2;                        // the entries behind the batch move to the front
    bool on = false;
    if (lane < nb) {
        uint64_t key;
        const CellProbe p = k31_probe_at<K, BOTH>(sh, pos0 + (int)(id & 0xFFFFu), id >> QUEUE_ANCHOR_SHIFT, tv.lbits, tv.ext, key);
        const CellProbe q2 = cell_second_line(p, tv.lbits);
        const size_t cell = (size_t)cell_line_index(q2.line, q2.r, tv.lbits, tv.regions) * 16 + (q2.sh >> 1);
        const uint2 e = reinterpret_cast<const uint2*>(tv.lines)[cell];
        const uint32_t rmask = cell_rmask(tv.lbits, BOTH);
        const uint32_t nspare = cell_spare_bits(tv.lbits, BOTH);
        const bool hit2 = (((e.x ^ q2.w) <= CELL_FLAG_MASK) & (((e.y ^ q2.r) & rmask) == 0) & ((e.y >> 16) != 0u)) != 0;
        if (hit2) found(id, e.y >> 16, (uint32_t)cell, false);
        const uint32_t bit = nspare == 0 ? 1u : (e.y >> cell_filter_pos(q2.w, e.x, tv.lbits, BOTH)) & 1u;
        on = !hit2 && (e.x & CELL_FLAG_STASH & bit) != 0u;
#if defined(KCFX_EXP_CUT) && KCFX_EXP_CUT <= 1
        on = false;                                         // (experiment build: nothing beyond the second line)
#endif
    }
    const uint64_t ON = wave_ballot(on);
    if (ON) {
        const int at = qn2 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(ON >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)ON, 0u));
        if (on) ids2[at] = id;
        qn2 += __popcll(ON);
    }
    wave_lds_fence();
    return rest;
}
// the second queue, once per tile (or when a batch could overflow it): third line, then the stash
template <int K, bool BOTH, class Lds, class Found>
__device__ __forceinline__ void k31c_rest_round(const TableView& tv, const Lds& sh, const uint32_t* ids2, int qn2, int lane, int pos0, Found&& found) {
    if (lane < qn2) {
        const uint32_t id = ids2[lane];
        uint64_t key;
        const CellProbe p = k31_probe_at<K, BOTH>(sh, pos0 + (int)(id & 0xFFFFu), id >> QUEUE_ANCHOR_SHIFT, tv.lbits, tv.ext, key);
        const uint32_t rmask = cell_rmask(tv.lbits, BOTH);
        const uint32_t nspare = cell_spare_bits(tv.lbits, BOTH);
        bool on = true;
        if (cell_has_third(tv.lbits, BOTH)) {               // (wave-uniform)
            const CellProbe q3 = cell_third_line(p, tv.lbits, BOTH, cell_nw(K));
            const size_t cell3 = (size_t)cell_line_index(q3.line, q3.r, tv.lbits, tv.regions) * 16 + (q3.sh >> 1);
            const uint2 e3 = reinterpret_cast<const uint2*>(tv.lines)[cell3];
            const bool hit3 = (((e3.x ^ q3.w) <= CELL_FLAG_MASK) & (((e3.y ^ q3.r) & rmask) == 0) & ((e3.y >> 16) != 0u)) != 0;
            if (hit3) found(id, e3.y >> 16, (uint32_t)cell3, false);
            const uint32_t bit = nspare == 0 ? 1u : (e3.y >> cell_filter_pos(q3.w, e3.x, tv.lbits, BOTH)) & 1u;
            on = !hit3 && (e3.x & CELL_FLAG_STASH & bit) != 0u;
        }
        if (on) {
            // linear probing at load <= 0.25 (0.8 at worst): the slot and its successor in one round trip, further ones (rare) one by one
            uint32_t slot = stash_slot(key, tv.stash_mask);
            const uint4 a = tv.stash[slot], b = tv.stash[(slot + 1) & tv.stash_mask];
            const uint64_t ka = u64_of(a.x, a.y), kb = u64_of(b.x, b.y);
            uint32_t got = ~0u, cnt = 0u;
            if (ka == key) {
                got = slot;
                cnt = a.z;
            } else if (ka != EMPTY_KEY) {
                if (kb == key) {
                    got = (slot + 1) & tv.stash_mask;
                    cnt = b.z;
                } else if (kb != EMPTY_KEY) {
                    slot = (slot + 2) & tv.stash_mask;
                    for (;;) {
                        const uint4 e = tv.stash[slot];
                        const uint64_t ek = u64_of(e.x, e.y);
                        if (ek == key) {
                            got = slot;
                            cnt = e.z;
                            break;
                        }
                        if (ek == EMPTY_KEY) break;
                        slot = (slot + 1) & tv.stash_mask;
                    }
                }
            }
            if (got != ~0u) found(id, cnt, got, true);
        }
    }
    wave_lds_fence();
}

template <int K, bool BOTH, bool PLANNED>
__global__ __launch_bounds__(PLANNED ? CELLS_BLOCK : TEXT_BLOCK) __attribute__((amdgpu_waves_per_eu(PLANNED ? 8 : KCFX_TEXT_WAVES_MIN, 8))) void screen_k31c_kernel(
    const uint8_t* __restrict__ text, RefPlan plan, const TileDesc* __restrict__ tiles, TableView tv, int32_t min_count,
    uint64_t* __restrict__ bits, int64_t* __restrict__ count_sums) {
    // (this kernel's own workgroup shapes. Planned reference: kcfx_internal.h; text-only: one wave per tile, see TEXT_BLOCK)
    constexpr int WAVES = PLANNED ? CELLS_WAVES : TEXT_BLOCK / 64;
    constexpr int WPW = (TILE_WORDS + WAVES - 1) / WAVES;   // words a wave can have
    using Lds = K31Plan;
    using Queue = K31CQueueT<WPW>;
    // (a text-only reference is planned tile by tile: the planner's exchange rows are dead before the codes, the anchors and the
    // queue are first written, and lie under all three)
    struct Work {
        Queue queue[WAVES];                                 // (first: its words are written once per word of the loop, with short offsets)
        Lds sh;
    };
    union Scratch {
        Work w;
        typename std::conditional<PLANNED, uint32_t, PlanRows>::type rows;
    };
    __shared__ __attribute__((aligned(16))) Scratch s_scratch;
    Lds& sh = s_scratch.w.sh;
    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    int off;
    if constexpr (PLANNED) off = k31_load_plan_tile<CELLS_BLOCK>(sh, plan, td, K, tid);
    else off = k31_plan_tile_wave<K>(sh, s_scratch.rows, text, td, tid);
    // From here on this wave issues table loads: it goes first, and the waves of the SIMD that are still planning their tiles
    // (pure arithmetic, nothing in flight) fill the issue slots it leaves while it waits. Same-box A/B on the bench workload:
    // 0.0880 - 0.0883 ms per launch against 0.0902 - 0.0913 without (profiles/r05_notes.md)
    if constexpr (!PLANNED) __builtin_amdgcn_s_setprio(3);

    const int lane = tid & 63, wave = PLANNED ? __builtin_amdgcn_readfirstlane(tid >> 6) : 0;
    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + WAVES - 1) / WAVES;
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);
    Queue& q = s_scratch.w.queue[wave];
    int qn = 0;                                             // queued k-mers (wave-uniform)
    int qn2 = 0;                                            // ... of them, sent on by their second line: third line + stash once per tile
    const int pos0 = off + w_begin * 64;                    // packed position of the wave's first k-mer

    uint32_t sum_acc = 0;                                   // per lane; u16 counters: fits (static_assert in screen_k31_kernel)
    int64_t big_acc = 0;                                    // counters beyond 16 bits come from the stash only, rarely
    uint64_t* my_bits = bits + (size_t)blockIdx.x * TILE_BITS_U64;
    PlannedWords<K> words;
    if (w_begin < w_end) words.start(sh, off, td.n_kmers, lane, w_begin);

    // a queued k-mer found in its second line or in the stash
    auto found = [&](uint32_t id, uint32_t cnt, uint32_t, bool) {
        if ((int32_t)cnt < min_count) return;               // GetVariants.java:224
        atomicOr(&q.hit[(id >> 6) & 0x3FFu], 1ull << (id & 63));
        // only a 3- / 4-byte counter (they live in the stash) exceeds 16 bits: its own 64-bit sum (GetVariants.java:226)
        if (cnt > CELL_COUNT_MAX) big_acc += (int64_t)(int32_t)cnt;
        else sum_acc += cnt;
    };
    // the second queue: third line, then stash, 64 entries at a time (normally once, at the end of the tile)
    auto flush_rest = [&]() {
        for (int off = 0; off < qn2; off += 64) k31c_rest_round<K, BOTH>(tv, sh, q.id2 + off, min(qn2 - off, 64), lane, pos0, found);
        qn2 = 0;
    };
    // stage A of word w: keys, minimizer, cell, fetches issued
    auto issue = [&](int w, CellWord& cw) {
        uint64_t N, V;
        uint32_t sh2, fm, a;
        words.next(w, N, sh2, fm, V, &a);
        uint32_t g;
        CellProbe p = cell_probe_planned<BOTH>(N, sh2, fm, tv.lbits, tv.ext, &g);
        cw.Vfull = V;
        // (wave-uniform: the table is one slice of the database -- the k-mers whose minimizer another device holds are that
        // device's to look up: no load, no hit, no queue entry here; the tiles' bitmaps are OR-ed afterwards)
        // (the text-only form: a planned reference against a slice has its own kernel below, which walks the slice's positions only)
        if (tv.slices > 1) V &= wave_ballot(cell_slice_of_g((a >> 5) & 3u, g, (uint32_t)tv.slices) == (uint32_t)tv.slice);
        if (tv.regions > 1) p.line |= cell_region_of(g, (uint32_t)tv.regions) << tv.lbits;  // (wave-uniform: cell_line_index; from here on the line is only an address)
        cells_issue<cell_nw(K), BOTH>(tv, V, p, a, cw);          // (the anchor byte whole: masked when -- if -- the k-mer is queued)
    };
    // stage B of word w: cells matched, ballots out, stash candidates queued
    auto retire = [&](int w, const CellWord& cw) {
        uint32_t cnt = 0, pend = 0;
        uint64_t M = 0;
        cells_match<BOTH, cell_nw(K)>(cw, tv.lbits, cnt, pend, &M);
        // cnt is 0 unless the k-mer was found at a valid position, and min_count >= 1 (checked by the API), so this is "found && count >= -c"
        const uint64_t H = wave_ballot((int32_t)cnt >= min_count);          // GetVariants.java:224 (valid positions only: cells_match)
        if (lane == 0) {
            q.vw[w - w_begin] = cw.Vfull;                                   // validity word of the window's bitmap
            q.hit[w - w_begin] = H;                                         // hits so far; queued k-mers may add theirs
        }
        // GetVariants.java:226: the counters of the hits. Every found counter is added; the lanes that found theirs below -c (none
        // at the default -c 1; decided on the scalar unit from the two masks) take it back
        sum_acc += cnt;
        if constexpr (cell_nw(K) > CELL_ALT_FREE_MAX_W) {
            const uint64_t low = M & ~H;
            if (low) sum_acc -= __builtin_amdgcn_inverse_ballot_w64(low) ? cnt : 0u;
        } else {
            sum_acc -= __builtin_amdgcn_inverse_ballot_w64(H) ? 0u : cnt;
        }
        // looked up further: flagged cell + filter bit, and not a hit here (a k-mer found below -c is not stored anywhere else
        // either: its look-up would end as the miss it already is -- harmless, and one compare less than "cnt == 0")
#if defined(KCFX_EXP_CUT) && KCFX_EXP_CUT == 0
        const uint64_t PM = 0;                              // (experiment build: first-choice cells only)
#else
        const uint64_t PM = wave_ballot(pend != 0u) & ~H & cw.V;
#endif
        EXP_COUNT(8, __builtin_amdgcn_inverse_ballot_w64(M));
        if (PM) {
            EXP_COUNT(7, lane == 0);
            const bool pending = __builtin_amdgcn_inverse_ballot_w64(PM);
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(PM >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)PM, 0u));
            if (pending) q.id[pos] = (uint32_t)(((w - w_begin) << 6) | lane) | ((cw.anc & 31u) << QUEUE_ANCHOR_SHIFT);
            qn += __popcll(PM);
        }
        wave_lds_fence();
        while (qn >= 64) {
            if (qn2 > 64) flush_rest();                     // (a batch of 64 may send all of its lanes on: room for them first)
            qn = k31c_second_round<K, BOTH>(tv, sh, q.id, q.id2, qn2, lane, qn, pos0, found);
        }
    };
    // D words of cell loads in flight per wave (CELLS_PIPE): a ring of D CellWords in fixed registers -- the loop is unrolled D
    // times -- and no condition around an issue in the steady state (see cells_issue); the first D - 1 issues and the last
    // <= 2 D - 2 words are peeled (there a skipped issue makes the compiler wait for everything in flight, which by then is old).
    // Before the cell load became non-temporal D = 2, 3, 4, 6 measured the same; since then the memory system takes more than two
    // words per wave can keep in flight: D = 2 / 3 / 4 / 5 / 6 = 0.0816 / 0.0790 / 0.0779 / 0.0783 / 0.084 ms (profiles/r03_notes.md).
    constexpr int D = CELLS_PIPE;
    CellWord cw[D];
    int base = w_begin;
#pragma unroll
    for (int i = 0; i < D - 1; ++i)
        if (w_begin + i < w_end) issue(w_begin + i, cw[i]);
    for (; base + 2 * D - 2 < w_end; base += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            issue(base + i + D - 1, cw[(i + D - 1) % D]);
            retire(base + i, cw[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2 * D - 2; ++i) {
        if (base + i + D - 1 < w_end) issue(base + i + D - 1, cw[(i + D - 1) % D]);
        if (base + i < w_end) retire(base + i, cw[i % D]);
    }
    while (qn > 0) {
        if (qn2 > 64) flush_rest();
        qn = k31c_second_round<K, BOTH>(tv, sh, q.id, q.id2, qn2, lane, qn, pos0, found);
    }
    flush_rest();
    if (lane < w_end - w_begin) {                           // the wave's words of the window's bitmap
        my_bits[w_begin + lane] = q.vw[lane];
        my_bits[TILE_WORDS + w_begin + lane] = q.hit[lane];
    }

    // (counters beyond 16 bits exist only in 3- / 4-byte-counter databases: their 64-bit wave sum -- twelve cross-lane moves --
    // is taken only when some lane has one)
    int64_t count_sum = (int64_t)wave_sum_u32_dpp(sum_acc);
    if (wave_ballot(big_acc != 0) != 0) count_sum += wave_sum_i64(big_acc);
    if constexpr (WAVES == 1) {
        if (lane == 0) count_sums[blockIdx.x] = count_sum;
    } else {
        if (lane == 0) sh.sum[wave] = count_sum;
        __syncthreads();
        if (tid == 0) {
            int64_t t = 0;
#pragma unroll
            for (int i = 0; i < WAVES; ++i) t += sh.sum[i];
            count_sums[blockIdx.x] = t;
        }
    }
}

// ======================================================================================================
// screen_k31c_slice: a planned reference against ONE SLICE of a database (round 6)
// ======================================================================================================
// Every device of a node walks the whole reference against its slice of the table. In screen_k31c_kernel that walk costs a slice
// what it costs a whole table -- keys, minimizer, cell address, match and ballots for all 64 lanes of every word, with the 1 / S of
// them that are this slice's switched on: a pass against one of four slices took 5.0 - 5.15 ms where the whole table's took 9.9,
// all of it instruction issue. Here the anchor bytes decide first: a wave reads the anchor of every position of its words (one
// LDS byte each), keeps the positions whose minimizer's slice class this slice can own (cell_slice_classes: exact with 2 or 4
// slices) in a list in LDS -- ~10 VALU instructions per word -- and then walks the LIST 64 entries at a time with the code of
// screen_k31c_kernel's word loop (same probe, same cell loads in a ring of CELLS_PIPE, same match, same queues for the k-mers that
// are looked up further). A lane's k-mer now sits at any position of the wave's words, so it derives its LDS addresses from the
// position (~12 instructions more per list word), and a hit is OR-ed into the tile's hit word in LDS (one byte per position,
// gathered at the end, costs 2 KB more LDS per workgroup = 13 workgroups per CU instead of 16: 1.12 ms where this form takes 0.90);
// the validity words do not depend on the slice and are written by the first phase. 3.34 ms for that pass (profiles/r06_notes.md
// section 11): 0.55 x the instructions, and a wave's life is now mostly its waits.
constexpr int SLICE_LIST = CELLS_WORDS_PER_WAVE * 64;       // positions of a wave's words
struct SliceWave {
    uint16_t list[SLICE_LIST];                              // (word of the wave << 6) | lane of the positions this slice may own, ascending
};
static_assert(SLICE_LIST <= 1024, "a position of the wave in 10 bits (queue ids keep 16)");

// N, sh2, fm (PlannedWords::next) of the k-mer at packed position q of the tile, from its anchor byte
template <int K>
__device__ __forceinline__ uint64_t planned_kmer_at(const K31Plan& sh, int q, uint32_t fm) {
    typedef const __attribute__((address_space(3))) uint32_t* lds_u32;
    constexpr uint32_t lo_mask = ~0u << (64 - 2 * K);
    // the strand the anchor names: position q of fw (lead word in front) or position T - q - K of rw
    const int qq = fm ? K31Plan::T - q - K : q;
    const int b = 2 * qq - 1;
    const uint32_t base = fm ? PlannedWords<K>::lds_address(sh.rw) : PlannedWords<K>::lds_address(sh.fw + 1);
    const lds_u32 p = (lds_u32)(uintptr_t)(base + 4u * (uint32_t)(b >> 5));
    const uint32_t sx = 31u - (uint32_t)(b & 31);
    const uint32_t y0 = p[0], y1 = p[1], y2 = p[2];
    return u64_of(alignbit32(y1, y2, sx) & lo_mask, alignbit32(y0, y1, sx));
}

template <int K, bool BOTH>
__global__ __launch_bounds__(CELLS_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void screen_k31c_slice_kernel(
    RefPlan plan, const TileDesc* __restrict__ tiles, TableView tv, uint32_t classes, int32_t min_count,
    uint64_t* __restrict__ bits, int64_t* __restrict__ count_sums) {
    constexpr int WAVES = CELLS_WAVES;
    using Lds = K31Plan;
    __shared__ Lds sh;
    __shared__ K31CQueue s_queue[WAVES];
    __shared__ SliceWave s_slice[WAVES];
    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int off = k31_load_plan_tile<CELLS_BLOCK>(sh, plan, td, K, tid);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + WAVES - 1) / WAVES;
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);
    K31CQueue& q = s_queue[wave];
    SliceWave& sw = s_slice[wave];
    const int pos0 = off + w_begin * 64;                    // packed position of the wave's first k-mer
    const int8_t* const an = reinterpret_cast<const int8_t*>(sh.an) + pos0;
    uint64_t* my_bits = bits + (size_t)blockIdx.x * TILE_BITS_U64;

    // ---- phase 1: the anchors of the wave's words -> validity words + the list of the positions this slice may own ----
    if (lane < CELLS_WORDS_PER_WAVE) q.hit[lane] = 0ull;    // the hit words: the lanes that find their k-mer OR their bit in
    int n_list = 0;                                         // (wave-uniform)
    {
        // (all the anchor reads first, then the words: one LDS round trip per wave, not one per word; a word behind the wave's
        // last reads anchors of the next wave's positions or the array's tail -- inside sh.an either way -- and is not looked at)
        static_assert(sizeof(sh.an) >= 32 + CELLS_WORDS_PER_WAVE * 64 * CELLS_WAVES, "phase 1 reads the anchors of whole words");
        int32_t a[CELLS_WORDS_PER_WAVE];
#pragma unroll
        for (int i = 0; i < CELLS_WORDS_PER_WAVE; ++i) a[i] = (int32_t)an[(i << 6) + lane];     // sign-extended: valid <=> negative
#pragma unroll
        for (int i = 0; i < CELLS_WORDS_PER_WAVE; ++i) {
            if (w_begin + i >= w_end) break;
            const int rem = td.n_kmers - (w_begin + i) * 64;                                    // >= 1
            const uint64_t V = (rem >= 64 ? ~0ull : ((1ull << rem) - 1ull)) & wave_ballot(a[i] < 0);
            if (lane == 0) q.vw[i] = V;
            const uint32_t cls = __builtin_amdgcn_ubfe((uint32_t)a[i], 5u, 2u);
            const uint64_t O = V & wave_ballot(__builtin_amdgcn_ubfe(classes, cls, 1u) != 0u);
            if (O) {
                const int at = n_list + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(O >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)O, 0u));
                if (__builtin_amdgcn_inverse_ballot_w64(O)) sw.list[at] = (uint16_t)((i << 6) | lane);
                n_list += __popcll(O);
            }
        }
    }
    wave_lds_fence();

    // ---- phase 2: the list, 64 positions per word ----
    int qn = 0, qn2 = 0;
    uint32_t sum_acc = 0;
    int64_t big_acc = 0;
    auto found = [&](uint32_t id, uint32_t cnt, uint32_t, bool) {
        if ((int32_t)cnt < min_count) return;               // GetVariants.java:224
        atomicOr(&q.hit[(id >> 6) & (uint32_t)(CELLS_WORDS_PER_WAVE - 1)], 1ull << (id & 63));
        if (cnt > CELL_COUNT_MAX) big_acc += (int64_t)(int32_t)cnt;     // GetVariants.java:226
        else sum_acc += cnt;
    };
    auto flush_rest = [&]() {
        for (int o = 0; o < qn2; o += 64) k31c_rest_round<K, BOTH>(tv, sh, q.id2 + o, min(qn2 - o, 64), lane, pos0, found);
        qn2 = 0;
    };
    const bool by_class = (tv.slices & (tv.slices - 1)) == 0 && tv.slices <= 4;     // the class alone names the slice
    struct ListWord {
        CellWord cw;
        uint32_t id;
    };
    // a list word in three steps -- its entries, their anchors, their k-mers (three dependent LDS reads) -- and the probe + load
    auto entry_of = [&](int c) -> uint32_t {                // (behind the list's end: some position of the wave, masked out of V)
        return sw.list[(64 * c + lane) & (SLICE_LIST - 1)] & (uint32_t)(SLICE_LIST - 1);
    };
    auto launch = [&](int c, uint32_t id, int32_t a, uint64_t N, ListWord& lw) {
        const int n = max(0, min(n_list - 64 * c, 64));
        uint64_t V = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
        uint32_t fm;
        asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(fm) : "v"(a));
        const uint32_t sh2 = (uint32_t)a & 30u;
        uint32_t g;
        CellProbe p = cell_probe_planned<BOTH>(N, sh2, fm, tv.lbits, tv.ext, &g);
        // (3, 5, 6, 7 or more than 4 slices: some classes are shared by two slices -- the hash of (line : r) decides)
        if (!by_class) V &= wave_ballot(cell_slice_of_g(((uint32_t)a >> 5) & 3u, g, (uint32_t)tv.slices) == (uint32_t)tv.slice);
        if (tv.regions > 1) p.line |= cell_region_of(g, (uint32_t)tv.regions) << tv.lbits;
        lw.id = id;
        cells_issue<cell_nw(K), BOTH>(tv, V, p, (uint32_t)a, lw.cw);
    };
    auto kmer_of = [&](uint32_t id, int32_t a) -> uint64_t {
        uint32_t fm;
        asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(fm) : "v"(a));
        return planned_kmer_at<K>(sh, pos0 + (int)id, fm);
    };
    auto issue = [&](int c, ListWord& lw) {
        const uint32_t id = entry_of(c);
        const int32_t a = (int32_t)an[id];
        launch(c, id, a, kmer_of(id, a), lw);
    };
    auto retire = [&](int, const ListWord& lw) {
        const CellWord& cw = lw.cw;
        uint32_t cnt = 0, pend = 0;
        uint64_t M = 0;
        cells_match<BOTH, cell_nw(K)>(cw, tv.lbits, cnt, pend, &M);
        const uint64_t H = wave_ballot((int32_t)cnt >= min_count);          // GetVariants.java:224 (cnt is 0 outside cw.V: cells_match)
        if (__builtin_amdgcn_inverse_ballot_w64(H)) atomicOr(&q.hit[lw.id >> 6], 1ull << (lw.id & 63));
        sum_acc += cnt;                                                     // GetVariants.java:226 (see screen_k31c_kernel)
        if constexpr (cell_nw(K) > CELL_ALT_FREE_MAX_W) {
            const uint64_t low = M & ~H;
            if (low) sum_acc -= __builtin_amdgcn_inverse_ballot_w64(low) ? cnt : 0u;
        } else {
            sum_acc -= __builtin_amdgcn_inverse_ballot_w64(H) ? 0u : cnt;
        }
#if defined(KCFX_EXP_CUT) && KCFX_EXP_CUT == 0
        const uint64_t PM = 0;                              // (experiment build: first-choice cells only)
#else
        const uint64_t PM = wave_ballot(pend != 0u) & ~H & cw.V;
#endif
        if (PM) {
            const bool pending = __builtin_amdgcn_inverse_ballot_w64(PM);
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(PM >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)PM, 0u));
            if (pending) q.id[pos] = lw.id | ((cw.anc & 31u) << QUEUE_ANCHOR_SHIFT);
            qn += __popcll(PM);
        }
        wave_lds_fence();
        while (qn >= 64) {
            if (qn2 > 64) flush_rest();
            qn = k31c_second_round<K, BOTH>(tv, sh, q.id, q.id2, qn2, lane, qn, pos0, found);
        }
    };
    const int c_end = (n_list + 63) >> 6;
    constexpr int D = CELLS_PIPE;                           // (5 and 6 list words in flight measure the same: 0.87 / 0.89 ms against 0.88)
    ListWord lw[D];
    int base = 0;
    {
        // the first D - 1 list words side by side, no condition between them (one block of code: the three LDS round trips of
        // each overlap with the others'; a word behind the list's end loads line 0 with every lane masked and is never retired)
        uint32_t id[D - 1];
        int32_t a[D - 1];
        uint64_t N[D - 1];
#pragma unroll
        for (int i = 0; i < D - 1; ++i) id[i] = entry_of(i);
#pragma unroll
        for (int i = 0; i < D - 1; ++i) a[i] = (int32_t)an[id[i]];
#pragma unroll
        for (int i = 0; i < D - 1; ++i) N[i] = kmer_of(id[i], a[i]);
#pragma unroll
        for (int i = 0; i < D - 1; ++i) launch(i, id[i], a[i], N[i], lw[i]);
    }
    for (; base + 2 * D - 2 < c_end; base += D) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            issue(base + i + D - 1, lw[(i + D - 1) % D]);
            retire(base + i, lw[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 2 * D - 2; ++i) {
        if (base + i + D - 1 < c_end) issue(base + i + D - 1, lw[(i + D - 1) % D]);
        if (base + i < c_end) retire(base + i, lw[i % D]);
    }
    while (qn > 0) {
        if (qn2 > 64) flush_rest();
        qn = k31c_second_round<K, BOTH>(tv, sh, q.id, q.id2, qn2, lane, qn, pos0, found);
    }
    flush_rest();
    wave_lds_fence();

    // ---- the wave's words of the tile's bitmaps ----
    if (lane < w_end - w_begin) {
        my_bits[w_begin + lane] = q.vw[lane];
        my_bits[TILE_WORDS + w_begin + lane] = q.hit[lane];
    }
    int64_t count_sum = (int64_t)wave_sum_u32_dpp(sum_acc);
    if (wave_ballot(big_acc != 0) != 0) count_sum += wave_sum_i64(big_acc);
    if (lane == 0) sh.sum[wave] = count_sum;
    __syncthreads();
    if (tid == 0) {
        int64_t t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += sh.sum[i];
        count_sums[blockIdx.x] = t;
    }
}

// ======================================================================================================
// screen_cohort: the tile walk of screen_k31c against a cohort table -- S <= 8 samples per k-mer look-up
// ======================================================================================================
// Three stages per word, software-pipelined: keys of word w + 2 issued | cells of word w + 1 matched and the 16 bytes of its
// lanes' counters requested | counters of word w turned into S hit ballots and S count sums. Everything before the counters
// is screen_k31c's code (K31Words, cell_probe_of, the two cell loads, cells_match): per sample a word costs one extract, one
// compare, one ballot word to LDS and one masked add.
struct CohortQueue {
    uint32_t id[K31_QUEUE];                                           // (word of the wave << 6) | lane
    unsigned long long vw[CELLS_WORDS_PER_WAVE];                      // validity ballots of the wave's words
    unsigned long long hit[COHORT_STRIDE][CELLS_WORDS_PER_WAVE];      // hit bits of the QUEUED k-mers per sample (OR-ed in by the lanes that find theirs)
    unsigned long long own[COHORT_STRIDE][CELLS_WORDS_PER_WAVE];      // hit ballots of the words themselves per sample (written once, plain stores)
};
struct CohortKeys {                       // stage 1 -> 2
    uint64_t V;
    uint32_t w, r;
    uint32_t anc;                         // anchor bits of the k-mer (CellWord::anc)
    uint32_t r_alt;                       // (k <= 30 only) the r field of the free-cell entry (cell_alt_of)
    uint32_t dsh;                         // (k <= 30 only) sh ^ (2 * index of the free cell)
    uint64_t l32;                         // 32 * line + 2 * first-choice cell: byte offset of the cell / 4, and of its counters / 8
    u32x2 cell, alt;
};
struct CohortCounts {                     // stage 2 -> 3
    uint64_t V, M;                        // valid lanes; lanes whose k-mer stands in a cell
    u32x2 c;                              // the 8 u8 counters of this lane's cell (meaningless outside M)
};
typedef const __attribute__((address_space(1))) u32x2* gcell_ptr_t;

template <int K, bool BOTH, bool PLANNED>
__global__ __launch_bounds__(CELLS_BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void screen_cohort_kernel(
    const uint8_t* __restrict__ text, RefPlan plan, const TileDesc* __restrict__ tiles, int32_t n_tiles, CohortView cv, int32_t min_count,
    uint64_t* __restrict__ bits, int64_t* __restrict__ count_sums) {
    constexpr int W = cell_nw(K);                           // 16-mers per k-mer = cells of a line a k-mer can sit in
    constexpr int WAVES = CELLS_WAVES;
    using Lds = K31Plan;
    __shared__ Lds sh;
    union Scratch {                                         // (text-only reference: the planner's rows are dead before the queues live)
        CohortQueue queue[WAVES];
        typename std::conditional<PLANNED, uint32_t, PlanRowsBlock>::type rows;
    };
    __shared__ __attribute__((aligned(16))) Scratch s_scratch;
    __shared__ uint32_t s_sums[WAVES][COHORT_STRIDE];
    const TableView& tv = cv.tv;
    const int S = cv.n_samples;
    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int off;
    if constexpr (PLANNED) off = k31_load_plan_tile<CELLS_BLOCK>(sh, plan, td, K, tid);
    else off = k31_plan_tile<K>(sh, s_scratch.rows, text, td, tid);
    CohortQueue& q = s_scratch.queue[wave];
    // the queued k-mers' hit bits are OR-ed into zeroed words (a word's own ballots live beside them: CohortQueue::own); wave-private
    for (int i = lane; i < COHORT_STRIDE * CELLS_WORDS_PER_WAVE; i += 64) (&q.hit[0][0])[i] = 0ull;
    wave_lds_fence();

    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + WAVES - 1) / WAVES;
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);
    int qn = 0;
    const int pos0 = off + w_begin * 64;
    uint32_t sum[COHORT_STRIDE];
#pragma unroll
    for (int s = 0; s < COHORT_STRIDE; ++s) sum[s] = 0;
    PlannedWords<K> words;
    if (w_begin < w_end) words.start(sh, off, td.n_kmers, lane, w_begin);
    const uint8_t* const count_base = reinterpret_cast<const uint8_t*>(cv.counts);

    // counter s of the eight a lane has loaded: 8 bits each in a cell, 16 bits each in a stash entry
    auto counter = [&](const u32x2& c, int s) -> uint32_t { return ((s >> 2) == 0 ? c.x : c.y) >> (8 * (s & 3)) & 0xFFu; };
    auto counter16 = [&](const uint4& c, int s) -> uint32_t {
        const uint32_t d = (s >> 1) == 0 ? c.x : (s >> 1) == 1 ? c.y : (s >> 1) == 2 ? c.z : c.w;
        return (s & 1) ? (d >> 16) : (d & 0xFFFFu);
    };
    // stage 1: keys, minimizer, cell -> the two cell loads (cells_issue of screen_k31c, keeping the cell's offset for stage 2)
    auto issue = [&](int w, CohortKeys& kw) {
        uint64_t V;
        CellProbe p;
        uint32_t anc;
        {
            uint64_t N;
            uint32_t sh2, fm;
            words.next(w, N, sh2, fm, V);
            p = cell_probe_planned<BOTH>(N, sh2, fm, tv.lbits, tv.ext);
            anc = sh2 | (fm & 1u);
        }
        const bool valid = __builtin_amdgcn_inverse_ballot_w64(V);
        kw.V = V;
        kw.w = p.w;
        kw.r = p.r;
        kw.anc = anc;
        uint32_t line = valid ? p.line : 0u;
        asm volatile("" : "+v"(line));
        uint64_t l32 = u64_of((line << 5) | p.sh, line >> 27);
        asm volatile("" : "+v"(l32));
        kw.l32 = l32;
        const uint8_t* a0 = reinterpret_cast<const uint8_t*>(reinterpret_cast<const uint32_t*>(tv.lines) + l32);
        if constexpr (W > CELL_ALT_FREE_MAX_W) kw.cell = nt_load2(a0);       // (non-temporal where nothing reads the line again: see cells_issue)
        else kw.cell = *(gcell_ptr_t)a0;
        if constexpr (W <= CELL_ALT_FREE_MAX_W) {           // k < 31: the free cell of the same line differs in the cell index only
            const CellAlt a = cell_alt_of(p.sh, p.w, p.r, W, tv.lbits, BOTH);
            kw.r_alt = a.r;
            kw.dsh = p.sh ^ a.sh;
            const uint64_t a1 = (uint64_t)a0 ^ (uint64_t)(kw.dsh << 2);
            kw.alt = *(gcell_ptr_t)a1;
        }
    };
    // the queued k-mers (their cell sent them on, with their filter bit): looked up 64 at a time in their second line, then in
    // the stash (k31c_deferred_round); a lane that finds its k-mer fetches the counters of that cell / stash slot
    auto found = [&](uint32_t id, uint32_t, uint32_t where, bool in_stash) {
        uint4 c16 = make_uint4(0u, 0u, 0u, 0u);
        u32x2 c8;
        c8.x = c8.y = 0u;
        if (in_stash) c16 = cv.stash_counts[where];
        else {
            const uint2 t = cv.counts[where];
            c8.x = t.x;
            c8.y = t.y;
        }
#pragma unroll
        for (int s = 0; s < COHORT_STRIDE; ++s) {
            const uint32_t cs = in_stash ? counter16(c16, s) : counter(c8, s);
            if (s < S && (int32_t)cs >= min_count) {               // GetVariants.java:224-227
                atomicOr(&q.hit[s][(id >> 6) & 0x3FFu], 1ull << (id & 63));
                sum[s] += cs;
            }
        }
    };
    auto stash_round = [&]() { qn = k31c_deferred_round<K, BOTH>(tv, sh, q.id, lane, qn, pos0, found); };
    // stage 2: cells matched; the counters of every lane's cell requested (first choice unless the second one matched)
    auto match = [&](int w, const CohortKeys& kw, CohortCounts& cw) {
        uint32_t cnt = 0, pend = 0;
        {
            CellWord m;
            m.V = kw.V;
            m.w = kw.w;
            m.r = kw.r;
            if constexpr (W <= CELL_ALT_FREE_MAX_W) {
                m.r_alt = kw.r_alt;
                m.alt = kw.alt;
            }
            m.cell = kw.cell;
            cells_match<BOTH, W>(m, tv.lbits, cnt, pend);
        }
        cw.V = kw.V;
        cw.M = wave_ballot(cnt != 0u) & kw.V;
        uint64_t l32 = kw.l32;
        if constexpr (W <= CELL_ALT_FREE_MAX_W) {                                  // the counters of the free cell if the k-mer stands there
            const bool first = ((kw.cell.x ^ kw.w) <= CELL_FLAG_MASK) && (((kw.cell.y ^ kw.r) & cell_rmask(tv.lbits, BOTH)) == 0);
            l32 ^= (uint64_t)(first ? 0u : kw.dsh);
        }
        if constexpr (W > CELL_ALT_FREE_MAX_W) cw.c = nt_load2(count_base + (l32 << 2));           // 8 bytes per cell, cell = l32 / 2
        else cw.c = *(gcell_ptr_t)(count_base + (l32 << 2));
        const uint64_t PM = wave_ballot(pend != 0u) & wave_ballot(cnt == 0u) & kw.V;
        if (PM) {
            const bool pending = __builtin_amdgcn_inverse_ballot_w64(PM);
            const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(PM >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)PM, 0u));
            if (pending) q.id[pos] = (uint32_t)(((w - w_begin) << 6) | lane) | (kw.anc << QUEUE_ANCHOR_SHIFT);
            qn += __popcll(PM);
        }
        wave_lds_fence();
        while (qn >= 64) stash_round();
    };
    // stage 3: S hit ballots and count sums of the word
    // (the S ballots of a word go to LDS in ONE store: sample s's ballot is put into lane s of a register pair -- v_writelane, two per
    // sample -- where an atomicOr by "one lane" per sample cost the election of that lane, a branch and an LDS atomic each: 74 M VALU
    // instructions per launch of the bench cohort, the kernel's bound, of which a quarter went there)
    auto consume = [&](int w, const CohortCounts& cw) {
        if (lane == 0) q.vw[w - w_begin] = cw.V;
        uint32_t h_lo = 0u, h_hi = 0u;
        static_for<0, COHORT_STRIDE>([&](auto ic) {
            constexpr int s = ic;
            if (s < S) {
                const uint32_t cs = counter(cw.c, s);
                const uint64_t H = wave_ballot((int32_t)cs >= min_count) & cw.M;   // GetVariants.java:224
                const bool hit = __builtin_amdgcn_inverse_ballot_w64(H);
                asm("v_writelane_b32 %0, %1, %2" : "+v"(h_lo) : "s"((uint32_t)H), "n"(s));
                asm("v_writelane_b32 %0, %1, %2" : "+v"(h_hi) : "s"((uint32_t)(H >> 32)), "n"(s));
                sum[s] += hit ? cs : 0u;                                           // GetVariants.java:226
            }
        });
        if (lane < S) q.own[lane][w - w_begin] = u64_of(h_lo, h_hi);
    };
    CohortKeys ka, kb;
    CohortCounts ca, cb;
    int w = w_begin;
    const int nw = w_end - w_begin;
    if (nw == 1) {
        issue(w, ka);
        match(w, ka, ca);
        consume(w, ca);
    } else if (nw > 1) {
        issue(w, ka);
        issue(w + 1, kb);
        match(w, ka, ca);
        for (; w + 3 < w_end; w += 2) {                     // at the top: keys of w + 1 (kb) and counters of w (ca) are in flight
            issue(w + 2, ka);
            match(w + 1, kb, cb);
            consume(w, ca);
            issue(w + 3, kb);
            match(w + 2, ka, ca);
            consume(w + 1, cb);
        }
        if (w + 2 < w_end) {
            issue(w + 2, ka);
            match(w + 1, kb, cb);
            consume(w, ca);
            match(w + 2, ka, ca);
            consume(w + 1, cb);
            consume(w + 2, ca);
        } else {
            match(w + 1, kb, cb);
            consume(w, ca);
            consume(w + 1, cb);
        }
    }
    while (qn > 0) stash_round();
    wave_lds_fence();
    const int64_t bits_stride = (int64_t)n_tiles * TILE_BITS_U64;
    for (int s = 0; s < S; ++s) {                           // per sample: the wave's words of the tile's hit bitmap
        uint64_t* my_bits = bits + (size_t)s * bits_stride + (size_t)blockIdx.x * TILE_BITS_U64;
        if (lane < w_end - w_begin) {
            if (s == 0) my_bits[w_begin + lane] = q.vw[lane];          // (once: reduce_windows_kernel reads them for every sample, shared_valid)
            my_bits[TILE_WORDS + w_begin + lane] = q.hit[s][lane] | q.own[s][lane];
        }
    }
#pragma unroll
    for (int s = 0; s < COHORT_STRIDE; ++s) {
        const uint32_t t = wave_sum_u32_dpp(sum[s]);
        if (lane == 0) s_sums[wave][s] = t;
    }
    __syncthreads();
    if (tid < S) {
        uint64_t t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += s_sums[i][tid];
        count_sums[(size_t)tid * n_tiles + blockIdx.x] = (int64_t)t;
    }
}

// ---- kcfx_ref_create: the plan of a reference (see K31Plan) ---------------------------------------------------------
// One WAVE per 63 strips of 32 text positions (StripPlanner<K, 32>; lane 63 only lends its first words to lane 62, see
// PlanWaveIo): 2 016 positions per wave, four independent waves per workgroup, no barrier. The anchor bytes leave through the
// wave's rows so that a store instruction writes 1 KB of consecutive bytes. Positions whose k-mer runs past the text read the
// zero padding behind it (or, further out, are handed zeros) and come out invalid.
constexpr int PLAN_BLOCK = 256, PLAN_WAVES = PLAN_BLOCK / 64, PLAN_L = 32, PLAN_WAVE_POS = 63 * PLAN_L;
template <int K>
__global__ __launch_bounds__(PLAN_BLOCK) void ref_plan_kernel(const uint8_t* __restrict__ text, int64_t text_len, int64_t text_readable,
                                                             uint32_t* __restrict__ codes, int64_t codes_words,
                                                             uint8_t* __restrict__ anchors, int64_t anchor_bytes) {
    using P = StripPlanner<K, PLAN_L>;
    __shared__ __attribute__((aligned(16))) PlanRows s_rows[PLAN_WAVES];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t origin = ((int64_t)blockIdx.x * PLAN_WAVES + wave) * PLAN_WAVE_POS;
    if (origin >= text_len) return;                         // (wave-uniform; the waves of a workgroup do not meet again)
    const int64_t s0 = origin + (int64_t)lane * PLAN_L;
    PlanWaveIo<K, PLAN_L> io{s_rows[wave], lane, s0 + PLAN_L <= text_readable ? text + s0 : nullptr};
    const typename P::Result res = P::run(io);
    if (lane < 63 && (s0 >> 4) + P::NF <= codes_words)
        *reinterpret_cast<uint2*>(codes + (s0 >> 4)) = make_uint2(res.codes[0], res.codes[1]);
    // anchors: lane l's 32 bytes to bytes 32 l .. of the rows, read back 16 bytes per lane and instruction
    uint4* const flat = reinterpret_cast<uint4*>(s_rows[wave]);
    flat[2 * lane] = make_uint4(res.anchors[0], res.anchors[1], res.anchors[2], res.anchors[3]);
    flat[2 * lane + 1] = make_uint4(res.anchors[4], res.anchors[5], res.anchors[6], res.anchors[7]);
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = 64 * i + lane;                    // 16-byte piece of the wave's 2 016 anchor bytes
        const int64_t at = origin + 16 * (int64_t)piece;
        if (piece < PLAN_WAVE_POS / 16 && at + 16 <= anchor_bytes) *reinterpret_cast<uint4*>(anchors + at) = flat[piece];
    }
}

// ---- S samples in one pass over the reference (SURVEY f-4) ------------------------------------------------------
// The same tile walk, but every word's keys / minimizers / validity are derived once and probed against S resident
// tables: pack, strands, hashing and the sliding minimum are amortised over the samples. Sample s writes its own hit
// bitmap (bits + s * bits_stride, the validity words replicated) and count sums (count_sums + s * n_tiles), so
// reduce_windows runs unchanged per sample (blockIdx.y).
constexpr int MULTI_MAX = SCREEN_MULTI_MAX;
struct MultiTables {
    TableView tv[MULTI_MAX];
    int32_t n;
};

// The later probe rounds of all samples share one queue per wave: an entry carries its sample in the top 3 bits of its line
// number (tables of the multi-sample pass have fewer than 2^29 lines and 2-byte counters, screen_multi_supported), so
// run detection, the run list and the re-queue work on the tagged value; the table base and size come from a small LDS
// table per lane.
constexpr int MULTI_TAG_SHIFT = 29;
constexpr uint32_t MULTI_LINE_MASK = (1u << MULTI_TAG_SHIFT) - 1;
struct MultiTab {
    const uint8_t* lines;
    uint32_t n_lines, pad;
};
struct K31MultiQueue {
    uint64_t key[K31_QUEUE];
    uint32_t tgt[K31_QUEUE];                                 // (sample << 29) | line
    uint16_t id[K31_QUEUE];                                  // (word of the wave << 6) | lane
    unsigned long long hit[MULTI_MAX][K31_WORDS_PER_WAVE];   // hit ballots per sample and word of the wave
};

// k31_deferred_round for entries of mixed samples; count sums go to sums[sample] (LDS atomics from the hit lanes)
__device__ __forceinline__ int k31_deferred_round_multi(const MultiTab* tab, const K31Fetch& f, K31MultiQueue& q, int lane,
                                                        int qn, int32_t min_count, unsigned long long* sums) {
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int nb = min(qn, 64), rest = qn - nb;
    const uint64_t key = q.key[lane];
    uint32_t tgt = q.tgt[lane];
    const uint32_t id = q.id[lane];
    const uint64_t k2 = q.key[64 + lane];
    const uint32_t t2 = q.tgt[64 + lane], i2 = q.id[64 + lane];
    wave_lds_fence();
    if (lane < rest) {                                       // the entries behind the batch move to the front
        q.key[lane] = k2;
        q.tgt[lane] = t2;
        q.id[lane] = (uint16_t)i2;
    }
    uint32_t st = lane < nb ? 1u : 0u, cnt = 0;
    const uint64_t PM = nb == 64 ? ~0ull : ((1ull << nb) - 1);
    const uint32_t smp = tgt >> MULTI_TAG_SHIFT;
    const uint32_t n_lines = tab[lane < nb ? smp : 0u].n_lines;
    // one probe round (probe_round with the table looked up per run)
    const uint64_t HM = PM & (~(PM << 1) | wave_ballot(wave_shr1(tgt) != tgt));
    const int nruns = __popcll(HM);
    const uint64_t H1 = HM >> 1;
    const int my_run = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(H1 >> 32),
                                                      __builtin_amdgcn_mbcnt_lo((uint32_t)H1, (uint32_t)((int)(HM & 1) - 1)));
    f.runline[(st == 1u) ? my_run : 64] = tgt;
    wave_lds_fence();
    for (int c0 = 0; c0 < nruns; c0 += K31_RUNS) {
        const int r0 = c0 + f.fetch_run, r1 = r0 + 7;
        if (f.fetcher && r0 < nruns) {
            const uint32_t v = f.runline[r0];
            __builtin_amdgcn_global_load_lds(
                (gptr_t)(tab[v >> MULTI_TAG_SHIFT].lines + (size_t)(v & MULTI_LINE_MASK) * LINE_BYTES + 16 * f.piece),
                (lptr_t)f.lines, 16, 0, 0);
        }
        if (f.fetcher && r1 < nruns) {
            const uint32_t v = f.runline[r1];
            __builtin_amdgcn_global_load_lds(
                (gptr_t)(tab[v >> MULTI_TAG_SHIFT].lines + (size_t)(v & MULTI_LINE_MASK) * LINE_BYTES + 16 * f.piece),
                (lptr_t)(f.lines + 7 * K31_PITCH), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wave_lds_fence();
        const uint32_t rel = (uint32_t)(my_run - c0);
        const bool mine = (st == 1u) && (rel < (uint32_t)K31_RUNS);
        const uint4* src = f.lines + (mine ? rel : 0u) * K31_PITCH;
        uint32_t c, meta;
        const bool any = match_line_lds<false, false>(src, key, c, meta);
        const uint32_t ov = (meta >> 8) & 1u;
        const uint32_t nst = any ? 2u : ov;
        const uint32_t line = tgt & MULTI_LINE_MASK;
        const uint32_t ntgt = (tgt & ~MULTI_LINE_MASK) | ((line + 1 == n_lines) ? 0u : line + 1);
        cnt = (mine && any) ? c : cnt;
        tgt = (mine && nst == 1u) ? ntgt : tgt;
        st = mine ? nst : st;
        wave_lds_fence();
    }
    const bool hit = (int32_t)cnt >= min_count;             // cnt is 0 unless found; GetVariants.java:224
    if (hit) {
        atomicOr(&q.hit[smp][id >> 6], 1ull << (id & 63));
        atomicAdd(&sums[smp], (unsigned long long)cnt);     // GetVariants.java:226
    }
    const uint64_t again = wave_ballot(st == 1u);
    if (again) {
        const int pos = rest + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(again >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)again, 0u));
        if (st == 1u) {
            q.key[pos] = key;
            q.tgt[pos] = tgt;
            q.id[pos] = (uint16_t)id;
        }
    }
    wave_lds_fence();
    return rest + __popcll(again);
}

template <bool BOTH>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(6, 8))) void screen_k31_multi_kernel(
    const uint8_t* __restrict__ text, const TileDesc* __restrict__ tiles, int32_t n_tiles, MultiTables mt, int32_t min_count,
    uint64_t* __restrict__ bits, int64_t bits_stride, int64_t* __restrict__ count_sums) {
    __shared__ K31Shared sh;
    __shared__ K31MultiQueue s_queue[WAVES];
    __shared__ unsigned long long s_msum[WAVES][MULTI_MAX];
    __shared__ MultiTab s_tab[MULTI_MAX];
    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;
    if (tid < MULTI_MAX) {
        const TableView t = mt.tv[tid < mt.n ? tid : 0];
        s_tab[tid].lines = t.lines;
        s_tab[tid].n_lines = t.n_lines;
    }
    const int off = k31_pack_tile(sh.pk, text, td, tid);       // (block barrier inside)

    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + WAVES - 1) / WAVES;
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);
    const K31Fetch fetch = fetch_setup(sh.lines[wave], sh.runline[wave], lane);
    K31MultiQueue& q = s_queue[wave];
    int qn = 0;
    if (lane < MULTI_MAX) s_msum[wave][lane] = 0;
    wave_lds_fence();
    K31Words words;
    if (w_begin < w_end) words.start(sh.pk, off, td.n_kmers, lane, w_begin);

    for (int w = w_begin; w < w_end; ++w) {
        uint64_t fwd, rc, V;
        uint32_t mn;
        words.next(w, fwd, rc, mn, V);
        const uint64_t key = (BOTH && rc < fwd) ? rc : fwd;     // Kmer.java:72-79
        if (lane == 0) {
            for (int s = 0; s < mt.n; ++s) bits[(size_t)s * bits_stride + (size_t)blockIdx.x * TILE_BITS_U64 + w] = V;
        }
        for (int s = 0; s < mt.n; ++s) {
            const TableView tv = mt.tv[s];
            uint32_t tgt = line_of_minimizer(mn, tv.n_lines);
            uint32_t st = (uint32_t)((V >> lane) & 1), cnt = 0, meta = 0;
            if (V) probe_round<false, false>(tv, fetch, V, key, tgt, st, cnt, meta);      // the home lines
            const bool hit = (int32_t)cnt >= min_count;                     // GetVariants.java:224
            const uint64_t H = wave_ballot(hit);
            // the word's count sum goes straight to the wave's per-sample total (no per-lane accumulator per sample)
            const uint32_t wsum = wave_sum_u32_dpp(hit ? cnt : 0u);
            if (lane == 0) {
                q.hit[s][w - w_begin] = H;                                  // hits so far; queued k-mers may add theirs
                s_msum[wave][s] += wsum;
            }
            const uint64_t PM = wave_ballot(st == 1u);
            if (PM) {
                heavy_redirect(tv, lane, PM, meta, key, tgt);
                const int pos = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(PM >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)PM, 0u));
                if (st == 1u) {
                    q.key[pos] = key;
                    q.tgt[pos] = ((uint32_t)s << MULTI_TAG_SHIFT) | tgt;
                    q.id[pos] = (uint16_t)(((w - w_begin) << 6) | lane);
                }
                qn += __popcll(PM);
            }
            wave_lds_fence();
            while (qn >= 64) qn = k31_deferred_round_multi(s_tab, fetch, q, lane, qn, min_count, s_msum[wave]);
        }
    }
    while (qn > 0) qn = k31_deferred_round_multi(s_tab, fetch, q, lane, qn, min_count, s_msum[wave]);
    // hit words of every sample's bitmap
    const int nw = w_end - w_begin;
    for (int i = lane; i < mt.n * nw; i += 64) {
        const int s = i / nw, wl = i - s * nw;
        bits[(size_t)s * bits_stride + (size_t)blockIdx.x * TILE_BITS_U64 + TILE_WORDS + w_begin + wl] = q.hit[s][wl];
    }
    __syncthreads();
    if (tid < mt.n) {
        unsigned long long t = 0;
#pragma unroll
        for (int i = 0; i < WAVES; ++i) t += s_msum[i][tid];
        count_sums[(size_t)tid * n_tiles + blockIdx.x] = (int64_t)t;
    }
}

// hit bitmap -> TileSummary (gap monoid + EFFLEN runs), GetVariants.java:228-252. SUM_LANES lanes per tile: each walks
// TILE_WORDS / SUM_LANES consecutive words, then the pieces are folded in order with the monoid (one lane per tile walked
// 32 dependent words and took 20 us for 15 k tiles).
constexpr int SUM_LANES = 8;
static_assert(TILE_WORDS % SUM_LANES == 0 && 64 % SUM_LANES == 0, "reduce_windows_kernel piece size");

__device__ __forceinline__ TileSummary shfl_down_summary(const TileSummary& s, int off) {
    TileSummary r;
    r.n = __shfl_down(s.n, off, 64);
    r.obs = __shfl_down(s.obs, off, 64);
    r.lead = __shfl_down(s.lead, off, 64);
    r.trail = __shfl_down(s.trail, off, 64);
    r.var_in = __shfl_down(s.var_in, off, 64);
    r.inner_in = __shfl_down(s.inner_in, off, 64);
    r.runs = __shfl_down(s.runs, off, 64);
    r.flags = (uint32_t)__shfl_down((int)s.flags, off, 64);
    const uint32_t lo = (uint32_t)__shfl_down((int)(uint32_t)s.count_sum, off, 64);
    const uint32_t hi = (uint32_t)__shfl_down((int)(uint32_t)((uint64_t)s.count_sum >> 32), off, 64);
    r.count_sum = (int64_t)(((uint64_t)hi << 32) | lo);
    return r;
}

// One wave per (window, sample): lane l takes piece l % 8 of tile l / 8 of the current group of 8 tiles, the 64 pieces
// are folded in order with shuffles, groups are chained on lane 0, and lane 0 writes the kcfx_result
// (GetVariants.java:248-258). One launch instead of a per-tile and a per-window kernel.
constexpr int REDUCE_BLOCK = 256;
__global__ __launch_bounds__(REDUCE_BLOCK) void reduce_windows_kernel(const TileDesc* __restrict__ tiles, int32_t n_tiles,
                                                                      const uint64_t* __restrict__ bits,
                                                                      const int64_t* __restrict__ count_sums,
                                                                      const int32_t* __restrict__ win_tile0, int32_t n_windows,
                                                                      int32_t k, kcfx_result* __restrict__ out, int32_t shared_valid) {
    constexpr int PW = TILE_WORDS / SUM_LANES;             // words per piece
    constexpr int GROUP = 64 / SUM_LANES;                  // tiles per wave step
    // blockIdx.y = sample of a multi-sample pass: its bitmaps, count sums and results follow the previous sample's
    // (shared_valid: the validity words of a tile are the same for every sample and were written once, into sample 0's block --
    // a cohort pass, whose kernel derives them once)
    const uint64_t* const vbits = bits + (shared_valid ? 0 : (size_t)blockIdx.y * n_tiles * TILE_BITS_U64);
    bits += (size_t)blockIdx.y * n_tiles * TILE_BITS_U64;
    count_sums += (size_t)blockIdx.y * n_tiles;
    out += (size_t)blockIdx.y * n_windows;
    const int lane = threadIdx.x & 63;
    const int32_t w = blockIdx.x * (REDUCE_BLOCK / 64) + (threadIdx.x >> 6);
    if (w >= n_windows) return;
    const int32_t t_begin = win_tile0[w], t_end = win_tile0[w + 1];
    const int sub = lane % SUM_LANES;
    TileSummary total = summary_identity();                // meaningful on lane 0
    for (int32_t t0 = t_begin; t0 < t_end; t0 += GROUP) {
        const int32_t t = t0 + lane / SUM_LANES;
        TileSummary acc = summary_identity();
        if (t < t_end) {
            // everything the piece needs is requested at once (words beyond the tile's last are never looked at): the kernel's
            // time is its chain of dependent loads -- 9.4 us with one round trip per word, tile descriptor and count sum
            const uint64_t* b = bits + (size_t)t * TILE_BITS_U64 + sub * PW;
            const uint64_t* bv = vbits + (size_t)t * TILE_BITS_U64 + sub * PW;
            const int32_t nk = tiles[t].n_kmers;
            const int64_t cs = count_sums[t];
            uint64_t Vl[PW], Hl[PW];
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                Vl[i] = bv[i];
                Hl[i] = b[TILE_WORDS + i];
            }
#pragma unroll
            for (int i = 0; i < PW; ++i) asm volatile("" : "+v"(Vl[i]), "+v"(Hl[i]));      // (or the compiler sinks the word loads behind
                                                                                           // the wait for n_kmers: loads of a
                                                                                           // const __restrict__ array move freely)
            const int n_local = min(nk - sub * PW * 64, PW * 64);
            acc = summarize_piece<PW>(Vl, Hl, n_local, k, sub == 0 ? cs : 0);
        }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const TileSummary nb = shfl_down_summary(acc, off);
            if ((lane & (2 * off - 1)) == 0) acc = summary_combine(acc, nb, k);
        }
        if (lane == 0) total = summary_combine(total, acc, k);
    }
    if (lane == 0) {
        if (total.flags & F_EMPTY) total.flags = 0;   // window shorter than k: empty list => all zeros (GetVariants.java:219)
        out[w] = summary_finalize(total, k);
    }
}

// ======================================================================================================
// launchers
// ======================================================================================================
static inline int grid_for(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_table_clear(uint8_t* lines, uint32_t n_lines, int32_t wide, hipStream_t st) {
    const uint64_t n_vec = (uint64_t)n_lines * (LINE_BYTES / 16);
    hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for((int64_t)n_vec, 256)), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(lines), n_vec, wide ? SLOTS_U32 / 2 : SLOTS_U16 / 2);
    return hipGetLastError();
}

hipError_t launch_decode_records(const uint8_t* records, int64_t first_record, int64_t n, int32_t record_size,
                                 int32_t suffix_bytes, int32_t counter_size, int32_t lut_prefix_length,
                                 const uint64_t* lut, int64_t lut_entries, uint64_t* keys_out, uint32_t* counts_out,
                                 hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_records_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, records, first_record,
                       n, record_size, suffix_bytes, counter_size, lut_prefix_length, lut, lut_entries, keys_out,
                       counts_out);
    return hipGetLastError();
}

hipError_t launch_table_count(uint8_t* lines, TableView tv, int32_t k, const uint64_t* keys, int64_t n, hipStream_t st) {
    if (n <= 0 || tv.mode != MODE_MINIMIZER) return hipSuccess;
    hipLaunchKernelGGL(table_count_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, lines, tv, k, keys, n);
    return hipGetLastError();
}

hipError_t launch_table_mark_heavy(uint8_t* lines, uint32_t n_lines, hipStream_t st) {
    hipLaunchKernelGGL(table_mark_heavy_kernel, dim3(grid_for(n_lines, 256)), dim3(256), 0, st, lines, n_lines);
    return hipGetLastError();
}

hipError_t launch_table_insert(uint8_t* lines, TableView tv, int32_t k, const uint64_t* keys, const uint32_t* counts,
                               int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 g(grid_for(n, 256, 256 * 16)), b(256);
    if (tv.wide)
        hipLaunchKernelGGL(table_insert_kernel<true>, g, b, 0, st, lines, tv, k, keys, counts, n);
    else
        hipLaunchKernelGGL(table_insert_kernel<false>, g, b, 0, st, lines, tv, k, keys, counts, n);
    return hipGetLastError();
}

hipError_t launch_table_finalize(uint8_t* lines, uint32_t n_lines, int32_t wide, unsigned long long* stats, hipStream_t st) {
    const dim3 g(grid_for(n_lines, 256)), b(256);
    if (wide)
        hipLaunchKernelGGL(table_finalize_kernel<true>, g, b, 0, st, lines, n_lines, stats);
    else
        hipLaunchKernelGGL(table_finalize_kernel<false>, g, b, 0, st, lines, n_lines, stats);
    return hipGetLastError();
}

hipError_t launch_cells_count(uint32_t* pop32, int32_t pop_bits, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                              hipStream_t st, int32_t presence, int32_t ext, int32_t by_minimizer) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cells_count_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, pop32, pop_bits, k, lbits, both, keys, counts, n, presence, ext, by_minimizer);
    return hipGetLastError();
}

hipError_t launch_cells_resolve(uint32_t* pop32, int32_t pop_bits, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                                uint64_t* pw, uint8_t* ps, hipStream_t st, int32_t ext, int32_t by_minimizer) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cells_resolve_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, pop32, pop_bits, k, lbits, both, keys, counts, n, pw, ps, ext, by_minimizer);
    return hipGetLastError();
}

hipError_t launch_cells_insert(uint8_t* lines, const uint32_t* pop32, int32_t pop_bits, int32_t cls, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys,
                               const uint32_t* counts, int64_t n, hipStream_t st, int32_t presence, const uint64_t* pw, const uint8_t* ps, int32_t ext, int32_t regions, int32_t slicing, int32_t by_minimizer) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cells_insert_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, lines, pop32, pop_bits, cls, k, lbits, both, keys, counts, n, presence,
                       RecProbes{pw, ps, ext, regions, slicing}, by_minimizer);
    return hipGetLastError();
}

size_t cells_pop_bytes(int32_t pop_bits) { return (size_t)1 << pop_bits; }

// how many minimizers have c k-mers, c = 0 .. 255: the histogram of the build's crowding counters (one byte per minimizer), from
// which kcfx_db_open estimates BEFORE the bids how many k-mers a plain table would leave outside their first-choice cells
__global__ void pop_histogram_kernel(const uint32_t* __restrict__ pop32, int64_t n_words, unsigned long long* __restrict__ hist) {
    __shared__ uint32_t h[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n_words; i += stride) {
        const uint32_t w = pop32[i];
        if (w == 0u) continue;                                  // (most minimizers of a 2^32 space are nobody's)
        for (int b = 0; b < 4; ++b) {
            const uint32_t c = (w >> (8 * b)) & 0xFFu;
            if (c) atomicAdd(&h[c], 1u);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 256; j += blockDim.x)
        if (h[j]) atomicAdd(&hist[j], (unsigned long long)h[j]);
}
hipError_t launch_pop_histogram(const uint32_t* pop32, int32_t pop_bits, unsigned long long* hist256, hipStream_t st) {
    const int64_t n_words = (int64_t)(cells_pop_bytes(pop_bits) / 4);
    hipLaunchKernelGGL(pop_histogram_kernel, dim3(4096), dim3(256), 0, st, pop32, n_words, hist256);
    return hipGetLastError();
}


hipError_t launch_cells_settle(uint8_t* lines, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                               unsigned long long* stats, hipStream_t st, int32_t presence, const uint64_t* pw, const uint8_t* ps, int32_t ext, int32_t regions, int32_t slicing, LostList lost) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cells_settle_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, lines, k, lbits, both, keys, counts, n, stats, presence, RecProbes{pw, ps, ext, regions, slicing}, lost);
    return hipGetLastError();
}

hipError_t launch_cells_stash(const uint8_t* lines, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                              uint4* stash, uint32_t stash_mask, hipStream_t st, int32_t presence, const uint64_t* pw, const uint8_t* ps, int32_t ext, int32_t regions, int32_t slicing) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cells_stash_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, lines, k, lbits, both, keys, counts, n, stash,
                       stash_mask, presence, RecProbes{pw, ps, ext, regions, slicing});
    return hipGetLastError();
}

hipError_t launch_lookup(TableView tv, int32_t k, const uint64_t* keys, int64_t n, int32_t* counts_out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (tv.fmt == FORMAT_CELLS) {
        hipLaunchKernelGGL(lookup_cells_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, tv, k, tv.both, keys, n, counts_out);
        return hipGetLastError();
    }
    const dim3 g(grid_for(n * 8, 256, 256 * 16)), b(256);
    if (tv.wide)
        hipLaunchKernelGGL(lookup_kernel<true>, g, b, 0, st, tv, k, keys, n, counts_out);
    else
        hipLaunchKernelGGL(lookup_kernel<false>, g, b, 0, st, tv, k, keys, n, counts_out);
    return hipGetLastError();
}

hipError_t launch_reduce_windows(const TileDesc* tiles, int32_t n_tiles, int32_t n_samples, const uint64_t* bits,
                                 const int64_t* count_sums, const int32_t* win_tile0, int32_t n_windows, int32_t k,
                                 kcfx_result* out, hipStream_t st, int32_t shared_valid) {
    if (n_windows <= 0 || n_samples <= 0) return hipSuccess;
    const int wpb = REDUCE_BLOCK / 64;
    hipLaunchKernelGGL(reduce_windows_kernel, dim3((n_windows + wpb - 1) / wpb, n_samples), dim3(REDUCE_BLOCK), 0, st, tiles,
                       n_tiles, bits, count_sums, win_tile0, n_windows, k, out, shared_valid);
    return hipGetLastError();
}

// Another table slice's results for the same tiles into these: the hit words OR-ed (a k-mer is looked up on exactly one slice),
// the count sums added; the validity words are the same on both sides. One thread per 64-bit word of the hit half.
__global__ void merge_partial_kernel(uint64_t* __restrict__ bits, int64_t* __restrict__ sums, const uint64_t* __restrict__ obits,
                                     const int64_t* __restrict__ osums, int64_t n_tiles) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_tiles * TILE_WORDS) return;
    const int64_t tile = i / TILE_WORDS, w = i % TILE_WORDS;
    const int64_t at = tile * TILE_BITS_U64 + TILE_WORDS + w;
    bits[at] |= obits[at];
    if (w == 0) sums[tile] += osums[tile];
}
hipError_t launch_merge_partial(uint64_t* bits, int64_t* sums, const uint64_t* obits, const int64_t* osums, int32_t n_tiles, hipStream_t st) {
    if (n_tiles <= 0) return hipSuccess;
    const int64_t n = (int64_t)n_tiles * TILE_WORDS;
    hipLaunchKernelGGL(merge_partial_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, bits, sums, obits, osums, (int64_t)n_tiles);
    return hipGetLastError();
}

// whole tiles of TILE_K positions + what a screening tile reads past its last k-mer (two code chunks, one 16-byte anchor piece)
size_t ref_plan_codes_bytes(int64_t text_len) { return (size_t)((text_len + TILE_K - 1) / TILE_K * TILE_K + 256) / 4; }
size_t ref_plan_anchor_bytes(int64_t text_len) { return (size_t)((text_len + TILE_K - 1) / TILE_K * TILE_K + 128); }
hipError_t launch_ref_plan(const uint8_t* text, int64_t text_len, int32_t k, uint32_t* codes, uint8_t* anchors, hipStream_t st,
                           hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (text_len <= 0) return hipSuccess;
    const int64_t n_waves = (text_len + PLAN_WAVE_POS - 1) / PLAN_WAVE_POS;
    const dim3 g((unsigned)((n_waves + PLAN_WAVES - 1) / PLAN_WAVES)), b(PLAN_BLOCK);
    // (the text is readable 128 bytes past its end: kcfx_ref_create)
    const int64_t readable = text_len + 128, cw = (int64_t)(ref_plan_codes_bytes(text_len) / 4), ab = (int64_t)ref_plan_anchor_bytes(text_len);
#define KCFX_LAUNCHP(KK) hipExtLaunchKernelGGL((ref_plan_kernel<KK>), g, b, 0, st, ev_start, ev_stop, 0, text, text_len, readable, codes, cw, anchors, ab)
    switch (k) {
#ifndef KCFX_ONLY_K31
        case 19: KCFX_LAUNCHP(19); break;
        case 20: KCFX_LAUNCHP(20); break;
        case 21: KCFX_LAUNCHP(21); break;
        case 22: KCFX_LAUNCHP(22); break;
        case 23: KCFX_LAUNCHP(23); break;
        case 24: KCFX_LAUNCHP(24); break;
        case 25: KCFX_LAUNCHP(25); break;
        case 26: KCFX_LAUNCHP(26); break;
        case 27: KCFX_LAUNCHP(27); break;
        case 28: KCFX_LAUNCHP(28); break;
        case 29: KCFX_LAUNCHP(29); break;
        case 30: KCFX_LAUNCHP(30); break;
#endif
        case 31: KCFX_LAUNCHP(31); break;
        default: return hipErrorInvalidValue;
    }
#undef KCFX_LAUNCHP
    return hipGetLastError();
}

// KCFX_SLICE_WALK_ALL (test hook): a planned reference against a table slice in screen_k31c_kernel's form -- every position of every
// word with the foreign lanes switched off, what the text-only reference does -- for the A/B and the parity of the two forms
static bool slice_walk_all() { return hook("KCFX_SLICE_WALK_ALL") != nullptr; }     // (read per launch: a test switches it inside one process)
hipError_t launch_screen_tiles(const uint8_t* text, RefPlan plan, const TileDesc* tiles, int32_t n_tiles, TableView tv, int32_t k,
                               int32_t both_strands, int32_t min_count, uint64_t* bits, int64_t* count_sums,
                               hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n_tiles <= 0) return hipSuccess;
    const dim3 g(n_tiles), b(BLOCK), bc(CELLS_BLOCK);
    // ev_start / ev_stop (may be null) take the kernel's own begin / end timestamps (hipExtLaunchKernelGGL), which is
    // what rocprofv3 --kernel-trace reports; events recorded around the launch would add the dispatch gaps
#define KCFX_LAUNCH(W, MD) \
    hipExtLaunchKernelGGL((screen_tiles_kernel<W, MD>), g, b, 0, st, ev_start, ev_stop, 0, text, tiles, tv, k, both_strands, min_count, bits, count_sums)
#define KCFX_LAUNCH31(W, BS) \
    hipExtLaunchKernelGGL((screen_k31_kernel<W, BS>), g, b, 0, st, ev_start, ev_stop, 0, text, tiles, tv, min_count, bits, count_sums)
    const bool generic_only = tv.generic != 0;                                      // A/B and test hook, fixed at kcfx_db_open
    if (tv.fmt == FORMAT_CELLS) {
        // a planned reference against a table slice: the walk over the slice's positions only (screen_k31c_slice_kernel)
        const bool by_list = tv.slices > 1 && !slice_walk_all();
        const uint32_t slice_classes = cell_slice_classes((uint32_t)tv.slices, (uint32_t)tv.slice);
#define KCFX_LAUNCHC1(KK, BS, PL) \
    hipExtLaunchKernelGGL((screen_k31c_kernel<KK, BS, PL>), g, (PL) ? bc : dim3(TEXT_BLOCK), 0, st, ev_start, ev_stop, 0, text, plan, tiles, tv, min_count, bits, count_sums)
#define KCFX_LAUNCHS1(KK, BS) \
    hipExtLaunchKernelGGL((screen_k31c_slice_kernel<KK, BS>), g, bc, 0, st, ev_start, ev_stop, 0, plan, tiles, tv, slice_classes, min_count, bits, count_sums)
#define KCFX_LAUNCHC(KK)                                                                  \
    do {                                                                                  \
        if (plan.codes && by_list) { if (both_strands) KCFX_LAUNCHS1(KK, true); else KCFX_LAUNCHS1(KK, false); }   \
        else if (plan.codes) { if (both_strands) KCFX_LAUNCHC1(KK, true, true); else KCFX_LAUNCHC1(KK, false, true); }   \
        else            { if (both_strands) KCFX_LAUNCHC1(KK, true, false); else KCFX_LAUNCHC1(KK, false, false); } \
    } while (0)
        switch (k) {
#ifndef KCFX_ONLY_K31
            case 19: KCFX_LAUNCHC(19); break;
            case 20: KCFX_LAUNCHC(20); break;
            case 21: KCFX_LAUNCHC(21); break;
            case 22: KCFX_LAUNCHC(22); break;
            case 23: KCFX_LAUNCHC(23); break;
            case 24: KCFX_LAUNCHC(24); break;
            case 25: KCFX_LAUNCHC(25); break;
            case 26: KCFX_LAUNCHC(26); break;
            case 27: KCFX_LAUNCHC(27); break;
            case 28: KCFX_LAUNCHC(28); break;
            case 29: KCFX_LAUNCHC(29); break;
            case 30: KCFX_LAUNCHC(30); break;
#endif
            case 31: KCFX_LAUNCHC(31); break;
            default: return hipErrorInvalidValue;
        }
#undef KCFX_LAUNCHC
#undef KCFX_LAUNCHC1
#undef KCFX_LAUNCHS1
    } else if (tv.mode == MODE_MINIMIZER && k == 31 && tv.mm_len == 16 && !generic_only) {
        if (tv.wide) { if (both_strands) KCFX_LAUNCH31(true, true); else KCFX_LAUNCH31(true, false); }
        else         { if (both_strands) KCFX_LAUNCH31(false, true); else KCFX_LAUNCH31(false, false); }
    } else if (tv.mode == MODE_MINIMIZER) {
        if (tv.wide) KCFX_LAUNCH(true, MODE_MINIMIZER); else KCFX_LAUNCH(false, MODE_MINIMIZER);
    } else {
        if (tv.wide) KCFX_LAUNCH(true, MODE_KEYHASH); else KCFX_LAUNCH(false, MODE_KEYHASH);
    }
#undef KCFX_LAUNCH31
#undef KCFX_LAUNCH
    return hipGetLastError();
}

hipError_t launch_cohort_votes(const uint64_t* keys, uint32_t* counts, int64_t n, uint4* votes, uint32_t mask, int32_t phase, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if (phase == 0)
        hipLaunchKernelGGL(cohort_vote_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, keys, counts, n, votes, mask);
    else
        hipLaunchKernelGGL(cohort_prio_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, keys, counts, n, votes, mask);
    return hipGetLastError();
}

hipError_t launch_cohort_fill(TableView tv, int32_t k, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                              int32_t sample, uint8_t* cell_counts, uint16_t* stash_counts, hipStream_t st, const uint64_t* pw, const uint8_t* ps) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(cohort_fill_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, tv, k, both, keys, counts, n, sample,
                       cell_counts, stash_counts, RecProbes{pw, ps, tv.ext, tv.regions, 0});
    return hipGetLastError();
}

hipError_t launch_screen_cohort(const uint8_t* text, RefPlan plan, const TileDesc* tiles, int32_t n_tiles, CohortView cv, int32_t k,
                                int32_t both_strands, int32_t min_count, uint64_t* bits, int64_t* count_sums, hipStream_t st,
                                hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n_tiles <= 0 || cv.n_samples <= 0) return hipSuccess;
    if (cv.n_samples > COHORT_STRIDE || cv.tv.fmt != FORMAT_CELLS) return hipErrorInvalidValue;
    const dim3 g(n_tiles), b(CELLS_BLOCK);
#define KCFX_LAUNCHCO1(KK, BS, PL) \
    hipExtLaunchKernelGGL((screen_cohort_kernel<KK, BS, PL>), g, b, 0, st, ev_start, ev_stop, 0, text, plan, tiles, n_tiles, cv, min_count, bits, count_sums)
#define KCFX_LAUNCHCO(KK)                                                                  \
    do {                                                                                   \
        if (plan.codes) { if (both_strands) KCFX_LAUNCHCO1(KK, true, true); else KCFX_LAUNCHCO1(KK, false, true); }   \
        else            { if (both_strands) KCFX_LAUNCHCO1(KK, true, false); else KCFX_LAUNCHCO1(KK, false, false); } \
    } while (0)
    switch (k) {                                            // every k a cell table exists for
#ifndef KCFX_ONLY_K31
        case 19: KCFX_LAUNCHCO(19); break;
        case 20: KCFX_LAUNCHCO(20); break;
        case 21: KCFX_LAUNCHCO(21); break;
        case 22: KCFX_LAUNCHCO(22); break;
        case 23: KCFX_LAUNCHCO(23); break;
        case 24: KCFX_LAUNCHCO(24); break;
        case 25: KCFX_LAUNCHCO(25); break;
        case 26: KCFX_LAUNCHCO(26); break;
        case 27: KCFX_LAUNCHCO(27); break;
        case 28: KCFX_LAUNCHCO(28); break;
        case 29: KCFX_LAUNCHCO(29); break;
        case 30: KCFX_LAUNCHCO(30); break;
#endif
        case 31: KCFX_LAUNCHCO(31); break;
        default: return hipErrorInvalidValue;
    }
#undef KCFX_LAUNCHCO
#undef KCFX_LAUNCHCO1
    return hipGetLastError();
}

bool screen_multi_supported(const TableView& tv, int32_t k) {
    // 2-byte counters and < 2^29 lines (64 GiB): the shared queue of the multi-sample kernel tags line numbers with the sample
    return tv.fmt == FORMAT_LINES && tv.mode == MODE_MINIMIZER && k == 31 && tv.mm_len == 16 && !tv.wide &&
           tv.n_lines < (1u << MULTI_TAG_SHIFT);
}

hipError_t launch_screen_multi(const uint8_t* text, const TileDesc* tiles, int32_t n_tiles, const TableView* tvs,
                               int32_t n_tables, int32_t both_strands, int32_t min_count, uint64_t* bits, int64_t* count_sums,
                               hipStream_t st, hipEvent_t ev_start, hipEvent_t ev_stop) {
    if (n_tiles <= 0 || n_tables <= 0) return hipSuccess;
    if (n_tables > MULTI_MAX) return hipErrorInvalidValue;
    MultiTables mt;
    for (int i = 0; i < MULTI_MAX; ++i) mt.tv[i] = tvs[i < n_tables ? i : 0];
    mt.n = n_tables;
    const int64_t stride = (int64_t)n_tiles * TILE_BITS_U64;
    if (both_strands)
        hipExtLaunchKernelGGL((screen_k31_multi_kernel<true>), dim3(n_tiles), dim3(BLOCK), 0, st, ev_start, ev_stop, 0, text,
                              tiles, n_tiles, mt, min_count, bits, stride, count_sums);
    else
        hipExtLaunchKernelGGL((screen_k31_multi_kernel<false>), dim3(n_tiles), dim3(BLOCK), 0, st, ev_start, ev_stop, 0, text,
                              tiles, n_tiles, mt, min_count, bits, stride, count_sums);
    return hipGetLastError();
}

}  // namespace kcfx
