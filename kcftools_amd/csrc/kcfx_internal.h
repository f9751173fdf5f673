// kcfx_internal.h -- structures shared by the host side and the HIP kernels of libkcfx.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

#include "../../include/kcfx.h"

namespace kcfx {

// ----------------------------------------------------------------------------------------------------
// HBM table: open addressing at 128-byte line granularity. Measured on MI355X (profiles/r01_membench*.log):
// a random access costs the same for 16..128 contiguous bytes (~50 G lines/s at any footprint) provided ONE
// wave instruction fetches the whole line (8 lanes x 16 B); per-lane multi-load probing collapses to ~19 G/s
// once the table exceeds ~3 GiB. So a line is 128 B and is always fetched cooperatively.
//
//   LAYOUT_U16 (counter_size <= 2): 12 x u64 key | 12 x u16 count | u32 meta | u32 pad   = 96 + 24 + 4 + 4
//   LAYOUT_U32 (counter_size 3, 4): 10 x u64 key | 10 x u32 count | u32 meta | u32 pad   = 80 + 40 + 4 + 4
//   meta (dword 30): bits 0..7 = occupied slots, bit 8 = overflow (some key whose home is this line or an
//   earlier one was pushed to the next line) => a probe that misses continues with the next line iff bit 8.
//
// Home line of a key (k-mer word):
//   MODE_KEYHASH   : umulhi(hash32(key), n_lines)                      -- every k-mer probes its own line
//   MODE_MINIMIZER : umulhi(remix32(minimizer(key)), n_lines)          -- minimizer = min over the k-M+1 M-mers of
//                    hash32(canonical M-mer): strand-symmetric, and consecutive reference k-mers share it in runs of
//                    ~(k-M+2)/2, so ONE line fetch serves a whole run of k-mers.
// Exactness never depends on the home function: the full 64-bit key is compared.
// ----------------------------------------------------------------------------------------------------
constexpr int LINE_BYTES = 128;
constexpr int SLOTS_U16 = 12;
constexpr int SLOTS_U32 = 10;
constexpr int META_DWORD = 30;
constexpr uint32_t META_OVERFLOW = 0x100u;
// bit 9: HEAVY -- more than HEAVY_KEYS k-mers have this line as their MINIMIZER home (a low-complexity repeat can put
// thousands of k-mers under one minimizer). Their chain would be walked line by line, so all of them are homed by
// their own key hash instead, and a probe that starts at a HEAVY minimizer home jumps to hash32(key)'s line.
constexpr uint32_t META_HEAVY = 0x200u;
constexpr uint32_t HEAVY_KEYS = 4 * SLOTS_U16;
// build-time encoding of the meta dword: bits 0..23 arrival counter, bit 31 heavy mark
constexpr uint32_t BUILD_COUNT_MASK = 0x00FFFFFFu, BUILD_HEAVY = 0x80000000u;
constexpr int MODE_KEYHASH = 0, MODE_MINIMIZER = 1;

// ----------------------------------------------------------------------------------------------------
// FORMAT_CELLS (k = 19 .. 31 -- BASELINE.json's canonical k = 31 among them; forward-only databases: CELL_R_FLIP in kcfx_device.h):
// the line a k-mer lives in is still chosen by its minimizer -- always a 16-mer: a k-mer spans W = k - 15 of them (16 for
// k = 31, 10 for KMC's default k = 25) -- but INSIDE the line the k-mer has exactly one
// cell: cell index = offset of the minimizer in the k-mer (0 .. W - 1), read on the strand where the minimizer 16-mer is in
// canonical form ("N orientation"). The W k-mers that can share a minimizer occurrence have W distinct offsets, so a
// run of consecutive reference k-mers maps onto distinct cells of ONE line, and a probe is ONE 8-byte load per lane --
// the lanes of a run hit the same 128-byte line, which the memory pipeline fetches once -- and two compares: no
// cooperative line fetch, no LDS staging, no 12-way 64-bit compare. A cell holds a quotiented key, exact by construction:
//
//   128-byte line = 16 x 8-byte cell;  cell = u32 w | u16 r | u16 count
//     w     = the flank bits (the k - 16 <= 15 bases of N around the minimizer, laid out relative to it: cell_flank) | bit 0: some k-mer that maps
//             to this cell lives elsewhere (second line or stash) | bit 1: the occupant is of the "crowded" class (build time only)
//     r     = low (32 - lbits) bits of G32 = a 32-bit mix of the canonical minimizer 16-mer; the line number is the top
//             lbits of G32, and G32 <-> minimizer is a bijection (cell_g32)
//             (at most 15 bits: lbits >= 17) | bit 15: this k-mer does NOT sit in its first-choice cell | the lbits - 17 bits in
//             between: in a flagged cell, a one-hash Bloom filter of the k-mers that were sent on, keyed by how their flank
//             differs from the occupant's (cell_filter_pos) -- a query that misses in a flagged cell looks further only if its bit is set
//     count = the stored counter; 0 = empty cell (a k-mer whose counter is 0 can never be observed -- GetVariants.java:224
//             needs count >= -c >= 1 -- so it is not stored). A counter beyond 65535 (3- / 4-byte counters only) does not
//             fit: that k-mer never takes a cell, it flags the cells it would have -- occupied or empty -- and lives in
//             the stash, whose entries carry 32-bit counts
//   (line, cell, w, r) -> N -> canonical k-mer is injective, so equality of (w, r) in the k-mer's cell IS equality of
//   the k-mer. Which 16-mer of a k-mer is its minimizer: the smallest under an order that puts open syncmers first
//   (cell_order27: 0.097 minimizer runs per k-mer instead of 0.118). When several k-mers want one cell, a k-mer whose minimizer
//   is shared by >= 3 k-mers of the database (the genome's runs -- what a reference window asks for) beats an isolated one, then
//   a counter >= 2 one of 1, then the k-mer of the higher-ranked RUN (a hash of the minimizer: where two runs share a line one of
//   them keeps every contested cell -- cell_bid, kcfx_kernels.hip), then the larger counter (64-bit atomic max: deterministic, and
//   in a real database the genomic k-mer beats its sequencing-error neighbours, which share its minimizer and offset). A loser takes, in this order: (a retired build-time
//   option, KCFX_ALT_FREE_MAX_W: for k < 31 a free cell W .. 15 of the same line;) the same cell of its SECOND LINE (top lbits of G32' = a second bijection of the
//   minimizer, r from G32' with bit 15 set: cell_second_line) -- the losers of a minimizer meet again there, one request for
//   all of them; (round 4, tables with >= 3 spare r bits) the same cell of its THIRD LINE -- chosen by the minimizer AND the three
//   flank bases next to it (cell_third_line: different loci that share a 16-mer part ways there; r carries one more mark); the stash: open addressing over 16-byte entries {u64 key, u32 count, u32 pad}, slot = hash32(key) & mask,
//   probed by the lanes individually. Both steps only into EMPTY cells, and each is announced in the cell before it
//   (flag + filter bit), so a query follows exactly the path the build took.
// ----------------------------------------------------------------------------------------------------
constexpr int FORMAT_LINES = 0, FORMAT_CELLS = 1;
constexpr int CELL_K_MIN = 19, CELL_K_MAX = 31;               // cell tables: 16-base minimizers, k - 15 = 4 .. 16 of them per k-mer
constexpr int CELL_LBITS_MIN = 17, CELL_LBITS_MAX = 31;     // r must fit 15 bits (bit 15 marks an entry outside its first-choice cell); n_lines fits 32
                                                            // (forward-only databases: 18, bit 14 of r is taken too -- cell_lbits_min)
__host__ __device__ inline int32_t cell_lbits_min(bool both) { return both ? CELL_LBITS_MIN : CELL_LBITS_MIN + 1; }
constexpr uint32_t CELL_FLAG_STASH = 1u, CELL_FLAG_MASK = 3u;
// a cell as one 64-bit word: w in bits 0..31, r in 32..47, count in 48..63
__host__ __device__ inline uint64_t cell_word(uint32_t w, uint32_t r, uint32_t count) {
    return (uint64_t)w | ((uint64_t)(r & 0xFFFFu) << 32) | ((uint64_t)(count & 0xFFFFu) << 48);
}

// TABLE SLICES (TableView::slices = S; round 5): a database whose table does not fit one device at a density worth having is
// partitioned over the S devices of a node BY MINIMIZER -- device s holds the cells of the k-mers whose minimizer belongs to s, in a
// table of 1 / S of the lines (kcfx_db_open_slice); every device screens the whole reference for the k-mers whose minimizer is its
// own, and the per-tile hit bitmaps and count sums are OR-ed / added before ONE reduce_windows (kcfx_shard.cpp, KCFX_SHARD_TABLE).
// Which slice: the top of a 32-bit word whose two leading bits are the minimizer's SLICE CLASS (order_slice_class, kcfx_device.h:
// two hash bits of its order key, which the anchor byte of every reference position carries) and whose other thirty are a hash of
// the 32 bits that name the line and r of the FIRST choice. So a k-mer's second and third lines, its filter bits and its stash
// entry live where its first-choice cell does, the k-mers of a minimizer run stay together (an extended table cuts a run in two
// at most, as it does for its lines), and a pass knows from the anchor bytes alone which positions cannot be its slice's
// (cell_slice_classes: with 2 or 4 slices the class IS the slice; with 8, two slices share a class) -- round 6: the planned
// reference's pass against a slice walks only those (screen_k31c_slice_kernel). A k-mer with several first-choice cells (tied
// minimizers: same key, same class) is stored per cell, each copy where a query that derives THAT cell will look.
__host__ __device__ inline uint32_t cell_slice_of_g(uint32_t cls, uint32_t g, uint32_t n_slices) {        // g = the 32 bits (line : r) themselves
    const uint32_t h = (g ^ (g >> 13)) * 0x7FEB352Du;
    const uint32_t u = (cls << 30) | ((h ^ (h >> 15)) >> 2);
    return (uint32_t)(((uint64_t)u * n_slices) >> 32);
}
__host__ __device__ inline uint32_t cell_slice_of(uint32_t cls, uint32_t line, uint32_t r, int32_t lbits, uint32_t n_slices) {
    return cell_slice_of_g(cls, (line << (32 - lbits)) | (r & (0xFFFFFFFFu >> lbits)), n_slices);
}
// bit c: a minimizer of slice class c can belong to slice `slice` of n_slices
__host__ __device__ inline uint32_t cell_slice_classes(uint32_t n_slices, uint32_t slice) {
    uint32_t m = 0;
    for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t lo = (uint32_t)(((uint64_t)(c << 30) * n_slices) >> 32), hi = (uint32_t)(((uint64_t)((c << 30) | 0x3FFFFFFFu) * n_slices) >> 32);
        if (lo <= slice && slice <= hi) m |= 1u << c;
    }
    return m;
}

struct TableView {
    const uint8_t* lines;
    uint32_t n_lines;
    int32_t wide;        // 0: LAYOUT_U16, 1: LAYOUT_U32
    int32_t mode;        // MODE_KEYHASH / MODE_MINIMIZER
    int32_t mm_len;      // M (minimizer length, <= min(k, 32)); unused in MODE_KEYHASH
    int32_t fmt;         // FORMAT_LINES (above) / FORMAT_CELLS
    int32_t lbits;       // FORMAT_CELLS: n_lines == regions << lbits
    uint32_t stash_mask; // FORMAT_CELLS: stash slots - 1 (a power of two)
    const uint4* stash;
    int32_t generic;     // screen with the general kernel whatever k / M (test and A/B hook, fixed when the handle is made)
    int32_t both;        // the database is canonical (both strands); 0: k-mers are looked up as the text reads them
    int32_t ext;         // FORMAT_CELLS: an extended table -- the first-choice line depends on the flank too (cell_first_ext, kcfx_device.h)
    int32_t regions;     // FORMAT_CELLS: 1, or R regions of 2^lbits lines each (cell_line_index, kcfx_device.h)
    int32_t slices;      // FORMAT_CELLS: 1, or the table holds slice `slice` of `slices` of the database (cell_slice_of, kcfx_device.h)
    int32_t slice;
};

// ----------------------------------------------------------------------------------------------------
// Cohort table (kcfx_cohort_open): ONE FORMAT_CELLS key table for S <= COHORT_STRIDE samples -- every stored k-mer's count
// field is 1 -- and the counters beside it: counts[cell][COHORT_STRIDE] u8 (cell = 16 * line + cell index: 8 bytes per
// cell, 128 bytes per line -- the ~10 cells of a minimizer run are ONE more line request; as u16 they were 1.75),
// stash_counts[slot][COHORT_STRIDE] u16. Counters are at most 2 bytes wide (counter_size <= 2); a k-mer that some sample counted
// more than COHORT_CELL_MAX times lives in the stash with all its counters (cohort_prio_kernel).
// ----------------------------------------------------------------------------------------------------
constexpr int COHORT_STRIDE = 8;
constexpr uint32_t COHORT_CELL_MAX = 255u;        // what a cell's 8-bit counter of one sample holds (0 = the sample does not have the k-mer)
constexpr uint32_t COHORT_RANK_WIDE = 0xFFFFu;    // the bid rank of a record whose k-mer has a wider counter in some sample: never bids, lives in the stash
struct CohortView {
    TableView tv;
    const uint2* counts;
    const uint4* stash_counts;
    int32_t n_samples;
};

// ----------------------------------------------------------------------------------------------------
// Tile plan: every window is cut into tiles of at most TILE_K consecutive k-mer start positions; one
// workgroup screens one tile and emits its hit bitmap; reduce_windows summarises the bitmaps piecewise (TileSummary)
// and folds a window's pieces in order.
// ----------------------------------------------------------------------------------------------------
#ifndef KCFX_BLOCK
#define KCFX_BLOCK 256
#endif
constexpr int BLOCK = KCFX_BLOCK;     // 4 wavefronts (line-table kernels)
// The cell-table kernel runs 2 wavefronts x 16 words per tile: a wave's fixed cost (its first word's strands, scans and cell
// loads: ~112 VALU instructions) is paid once per 16 words instead of once per 8 -- the kernel sits at the VALU issue limit,
// and same-box A/B on the bench workload gives 0.1039 ms per launch against 0.1118 ms with 4 x 8 (profiles/r03_notes.md)
#ifndef KCFX_CELLS_BLOCK
#define KCFX_CELLS_BLOCK 128
#endif
constexpr int CELLS_BLOCK = KCFX_CELLS_BLOCK;
// words of cell loads a wave of the (planned) cell-table kernel keeps in flight
#ifndef KCFX_CELLS_PIPE
#define KCFX_CELLS_PIPE 4
#endif
constexpr int CELLS_PIPE = KCFX_CELLS_PIPE;
#ifndef KCFX_TILE_K
#define KCFX_TILE_K 2048
#endif
constexpr int TILE_K = KCFX_TILE_K;   // k-mer start positions per tile (multiple of 64)
constexpr int TILE_WORDS = TILE_K / 64;
// per tile: TILE_WORDS valid-ballots then TILE_WORDS hit-ballots (screen kernel -> reduce_windows)
constexpr int TILE_BITS_U64 = 2 * TILE_WORDS;
// A reference that is kept as text only (kcfx_ref_opts.expected_screens = 1, KCFX_REF_PLAN=0) is planned tile by tile inside the
// screening launch. screen_k31c_kernel: ONE wave per tile, a strip of 32 positions per lane, lane 63's strip only lends to its
// neighbour (k31_plan_tile_wave); screen_cohort_kernel: every thread of the workgroup a strip of 16, the last two strips only
// lend (k31_plan_tile). Either way that covers tiles of at most this many k-mers (up to 15 bases of alignment in front). Not a
// multiple of 64: a 10 kb window is 5 x 1 994 k-mers, and only a tile's LAST word has to be ragged.
constexpr int TEXT_TILE_K = (16 * (CELLS_BLOCK - 2) < 32 * 63 ? 16 * (CELLS_BLOCK - 2) : 32 * 63) - 15;

// The sample-independent half of a cell-table look-up, worked out once per reference (launch_ref_plan, k = 19 .. 31):
// 2-bit codes of the text (16 bases per dword, first base in the top bits) and one anchor byte per k-mer start position
// (bit 7 valid, bits 4..1 twice the cell index, bit 0 "the canonical strand is the reverse complement"). Both arrays are
// readable ref_plan_*_bytes(text_len) bytes and zero behind the text.
struct RefPlan {
    const uint32_t* codes = nullptr;
    const uint8_t* anchors = nullptr;
};

struct TileDesc {
    int64_t text_off;    // offset of the tile's first base in the padded device text
    int32_t n_kmers;     // k-mer start positions in this tile (1..TILE_K)
    int32_t window;
};

// The gap monoid of SURVEY.md section 8 a-7 plus what EFFLEN needs (a-8): summarises a contiguous range of
// k-mer start positions of one window.
struct TileSummary {
    int32_t n;           // valid k-mers
    int32_t obs;         // hits
    int32_t lead;        // misses before the first hit (== n when there is no hit)
    int32_t trail;       // misses after the last hit  (== n when there is no hit)
    int32_t var_in;      // interior miss-runs (between two hits of this range)
    int32_t inner_in;    // sum of getDistance over them
    int32_t runs;        // maximal runs of valid k-mer start positions
    uint32_t flags;      // bit0 any hit, bit1 first position valid, bit2 last position valid, bit3 empty range
    int64_t count_sum;   // sum of counts over hits
};
constexpr uint32_t F_ANY = 1u, F_FIRST = 2u, F_LAST = 4u, F_EMPTY = 8u;

// GetVariants.java:267-273
__host__ __device__ inline int32_t gap_distance(int32_t k, int32_t gap) {
    int32_t d = gap - (k - 1);
    if (d <= 0) {
        d = d + 1;
        d = d < 0 ? -d : d;
    }
    return d;
}

__host__ __device__ inline TileSummary summary_identity() {
    TileSummary s;
    s.n = s.obs = s.lead = s.trail = s.var_in = s.inner_in = s.runs = 0;
    s.flags = F_EMPTY;
    s.count_sum = 0;
    return s;
}

// A followed by B (adjacent position ranges of the same window).
__host__ __device__ inline TileSummary summary_combine(const TileSummary& a, const TileSummary& b, int32_t k) {
    if (a.flags & F_EMPTY) return b;
    if (b.flags & F_EMPTY) return a;
    TileSummary c;
    c.n = a.n + b.n;
    c.obs = a.obs + b.obs;
    c.count_sum = a.count_sum + b.count_sum;
    c.runs = a.runs + b.runs - (((a.flags & F_LAST) && (b.flags & F_FIRST)) ? 1 : 0);
    c.flags = (a.flags & F_FIRST) | (b.flags & F_LAST) | ((a.flags | b.flags) & F_ANY);
    c.var_in = a.var_in + b.var_in;
    c.inner_in = a.inner_in + b.inner_in;
    const bool ha = a.flags & F_ANY, hb = b.flags & F_ANY;
    if (ha && hb) {
        int32_t g = a.trail + b.lead;
        if (g > 0) {
            c.var_in += 1;
            c.inner_in += gap_distance(k, g);
        }
        c.lead = a.lead;
        c.trail = b.trail;
    } else if (ha) {
        c.lead = a.lead;
        c.trail = a.trail + b.n;
    } else if (hb) {
        c.lead = a.n + b.lead;
        c.trail = b.trail;
    } else {
        c.lead = c.n;
        c.trail = c.n;
    }
    return c;
}

__host__ __device__ inline kcfx_result summary_finalize(const TileSummary& s, int32_t k) {
    kcfx_result r;
    const bool any = s.flags & F_ANY;
    r.total_kmers = s.n;
    r.eff_length = s.n + (k - 1) * s.runs;       // sum over ACGT runs >= k of their length (Fasta.java:140-167)
    r.obs = s.obs;
    r.left = any ? s.lead : 0;
    r.right = any ? s.trail : s.n;
    r.var = s.var_in + ((any && s.lead > 0) ? 1 : 0) + (r.right > 0 ? 1 : 0);
    r.inner = s.inner_in;
    r._pad = 0;
    r.kmer_count_sum = s.count_sum;
    return r;
}

// ---- error plumbing ---------------------------------------------------------------------------------
void set_error(const std::string& msg);
int32_t hip_fail(hipError_t e, const char* what);
int32_t exception_to_status(const char* fn) noexcept;
// kcfx_pool.cpp: hipMalloc / hipFree with large released blocks kept per device for the next database (see there)
hipError_t dev_alloc(void** p, size_t bytes);
template <class T> inline hipError_t dev_alloc(T** p, size_t bytes) { return dev_alloc(reinterpret_cast<void**>(p), bytes); }
hipError_t dev_free(void* p);                // waits for the device first, as hipFree does
hipError_t dev_free_idle(void* p);           // the caller has already waited for every user of the block (its own stream)
size_t free_device_bytes();                  // hipMemGetInfo's free + what the pool keeps on the current device
size_t pooled_device_bytes(int dev);
// Switches that change what the library BUILDS (table size / format / regions, load tiers, kernel selection) are test and A/B
// hooks, not configuration: they are read only when KCFX_TEST_HOOKS is set in the environment as well, so that a host process (a JVM)
// that happens to inherit one of them gets the documented behaviour. hook(name) = getenv(name) under that condition, else null.
// (Always read: KCFX_VERBOSE, KCFX_LOAD_TRACE / KCFX_REF_TRACE / KCFX_CLI_TRACE / KCFX_POOL_TRACE, KCFX_NO_POOL, KCFX_POOL_FRACTION,
// KCFX_LENIENT_LUT -- they report, or change where memory goes, never what a look-up returns or costs.)
const char* hook(const char* name);
// one .kmc_suf read once for the table slices that several devices build at the same time (kcfx_api.cpp: SuffixCache)
struct SuffixCache;
SuffixCache* suffix_cache_create(const char* kmc_prefix, int32_t consumers);     // null: the loaders read for themselves
void suffix_cache_destroy(SuffixCache* c);
void suffix_cache_stats(const SuffixCache* c, int64_t* reads, int64_t* hits, int64_t* bytes_read);
void db_close_idle(kcfx_db* db);     // kcfx_db_close without the device-wide wait: every launch that read the table has been waited for
int32_t db_open_slice_shared(const char* kmc_prefix, int32_t device, const kcfx_db_opts* opts, int32_t n_slices, int32_t slice, SuffixCache* shared, kcfx_db** out);   // kcfx_db_open_slice with the loaders' shared cache (or null)
void pool_release(int dev);                  // dev < 0: every device
double pool_set_fraction(double f);          // the bound on what is kept (share of a device); returns the previous one
void pool_trim();                            // release what is over the bound now
hipError_t host_alloc(void** p, size_t bytes);    // pinned staging buffers, a few kept
template <class T> inline hipError_t host_alloc(T** p, size_t bytes) { return host_alloc(reinterpret_cast<void**>(p), bytes); }
hipError_t host_free(void* p);
void host_pool_release();    // inside catch (...): last_error := what was thrown, returns KCFX_ERR_HOST

#define KCFX_HIP(call)                                         \
    do {                                                       \
        hipError_t e__ = (call);                               \
        if (e__ != hipSuccess) return kcfx::hip_fail(e__, #call); \
    } while (0)

// ---- kernel launchers (kcfx_kernels.hip) ------------------------------------------------------------
// table build
hipError_t launch_table_clear(uint8_t* lines, uint32_t n_lines, int32_t wide, hipStream_t st);
hipError_t launch_decode_records(const uint8_t* records, int64_t first_record, int64_t n, int32_t record_size,
                                 int32_t suffix_bytes, int32_t counter_size, int32_t lut_prefix_length,
                                 const uint64_t* lut, int64_t lut_entries, uint64_t* keys_out, uint32_t* counts_out,
                                 hipStream_t st);
hipError_t launch_table_count(uint8_t* lines, TableView tv, int32_t k, const uint64_t* keys, int64_t n, hipStream_t st);
hipError_t launch_table_mark_heavy(uint8_t* lines, uint32_t n_lines, hipStream_t st);
hipError_t launch_table_insert(uint8_t* lines, TableView tv, int32_t k, const uint64_t* keys,
                               const uint32_t* counts, int64_t n, hipStream_t st);
hipError_t launch_table_finalize(uint8_t* lines, uint32_t n_lines, int32_t wide, unsigned long long* stats, hipStream_t st);
// FORMAT_CELLS build, five passes over the k-mers, each after ALL of the one before: count per minimizer sub-bucket
// (cells_pop_bytes of zeroed scratch), the bids for the first-choice cells (cls 1: every k-mer claims with its rank), the winners'
// cell words (cls 0: written where a k-mer finds its own rank), settle the losers into
// second-choice cells or flag them for the stash (stats[0] = how many), fill the stash
size_t cells_pop_bytes(int32_t pop_bits);
// minimizers per crowding count 0 .. 255 (hist256: 256 zeroed counters on the device), see kcfx_db_open's early choice of an extended table
hipError_t launch_pop_histogram(const uint32_t* pop32, int32_t pop_bits, unsigned long long* hist256, hipStream_t st);          // pop_bits = 32 (one counter per minimizer) or lbits + 2
// (both = the database is canonical; see CELL_R_FLIP in kcfx_device.h for what a forward-only one changes)
// (presence != 0: every non-zero counter is stored as 1 -- the key table of a cohort, kcfx_cohort_open)
hipError_t launch_cells_count(uint32_t* pop32, int32_t pop_bits, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                              hipStream_t st, int32_t presence = 0, int32_t ext = 0, int32_t by_minimizer = 0);
// (by_minimizer: the crowding counters / the run of a bid keyed by the minimizer alone instead of by the locus -- cell_crowd_key, kcfx_kernels.hip)
// (pw / ps: the cells of the records as launch_cells_resolve kept them -- 8 + 1 bytes per record, see RecProbes in kcfx_kernels.hip --
// or null: the pass derives them itself with cells_visit. launch_cells_resolve = launch_cells_count + keeping them.)
hipError_t launch_cells_resolve(uint32_t* pop32, int32_t pop_bits, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                                uint64_t* pw, uint8_t* ps, hipStream_t st, int32_t ext = 0, int32_t by_minimizer = 0);
hipError_t launch_cells_insert(uint8_t* lines, const uint32_t* pop32, int32_t pop_bits, int32_t cls, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys,
                               const uint32_t* counts, int64_t n, hipStream_t st, int32_t presence = 0, const uint64_t* pw = nullptr, const uint8_t* ps = nullptr, int32_t ext = 0, int32_t regions = 1, int32_t slicing = 0, int32_t by_minimizer = 0);
// what the settle pass counts for the stash, written down (round 5): keys / counts of `cap` entries and the number appended so far
// (it keeps counting beyond cap: n > cap = the list is incomplete and the stash is filled from the records instead). keys == null: no list.
struct LostList {
    uint64_t* keys = nullptr;
    uint32_t* counts = nullptr;
    unsigned long long* n = nullptr;
    unsigned long long cap = 0;
};
hipError_t launch_cells_settle(uint8_t* lines, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                               unsigned long long* stats, hipStream_t st, int32_t presence = 0, const uint64_t* pw = nullptr, const uint8_t* ps = nullptr, int32_t ext = 0, int32_t regions = 1, int32_t slicing = 0,
                               LostList lost = LostList());
hipError_t launch_cells_stash(const uint8_t* lines, int32_t k, int32_t lbits, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                              uint4* stash, uint32_t stash_mask, hipStream_t st, int32_t presence = 0, const uint64_t* pw = nullptr, const uint8_t* ps = nullptr, int32_t ext = 0, int32_t regions = 1, int32_t slicing = 0);
// (slicing: 0, or slices << 8 | slice -- the table is one slice of the database, kcfx_db_open_slice: the passes act on this slice's cells only)
// cohort tables: one sample's counters into the finished key table; S samples screened in one pass (bits / count_sums laid out
// per sample as launch_screen_multi's: sample s at bits + s * n_tiles * TILE_BITS_U64, count_sums + s * n_tiles)
// phase 0: every record votes for its k-mer in `votes` (2^n slots of 16 bytes, all bytes 0xFF); phase 1: the bid rank of every
// record (samples holding its k-mer, then its counter) into the upper half of its counts word
hipError_t launch_cohort_votes(const uint64_t* keys, uint32_t* counts, int64_t n, uint4* votes, uint32_t mask, int32_t phase, hipStream_t st);
hipError_t launch_cohort_fill(TableView tv, int32_t k, int32_t both, const uint64_t* keys, const uint32_t* counts, int64_t n,
                              int32_t sample, uint8_t* cell_counts, uint16_t* stash_counts, hipStream_t st,
                              const uint64_t* pw = nullptr, const uint8_t* ps = nullptr);
hipError_t launch_screen_cohort(const uint8_t* text, RefPlan plan, const TileDesc* tiles, int32_t n_tiles, CohortView cv, int32_t k,
                                int32_t both_strands, int32_t min_count, uint64_t* bits, int64_t* count_sums, hipStream_t st,
                                hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// the plan of a reference (RefPlan): `text` must be readable 128 bytes past text_len (zeros), the outputs sized as below
size_t ref_plan_codes_bytes(int64_t text_len);
size_t ref_plan_anchor_bytes(int64_t text_len);
hipError_t launch_ref_plan(const uint8_t* text, int64_t text_len, int32_t k, uint32_t* codes, uint8_t* anchors, hipStream_t st,
                           hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// lookup
hipError_t launch_lookup(TableView tv, int32_t k, const uint64_t* keys, int64_t n, int32_t* counts_out, hipStream_t st);
// screen
// (plan: what kcfx_ref_create prepared for cell tables -- null members: the kernels work from the text)
hipError_t launch_screen_tiles(const uint8_t* text, RefPlan plan, const TileDesc* tiles, int32_t n_tiles, TableView tv, int32_t k,
                               int32_t both_strands, int32_t min_count, uint64_t* bits, int64_t* count_sums,
                               hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// hit bitmaps + count sums of n_samples consecutive samples -> kcfx_result per (sample, window)
// table slices: another slice's tile bitmaps (hit halves OR-ed) and count sums (added) into these
hipError_t launch_merge_partial(uint64_t* bits, int64_t* sums, const uint64_t* obits, const int64_t* osums, int32_t n_tiles, hipStream_t st);
hipError_t launch_reduce_windows(const TileDesc* tiles, int32_t n_tiles, int32_t n_samples, const uint64_t* bits,
                                 const int64_t* count_sums, const int32_t* win_tile0, int32_t n_windows, int32_t k,
                                 kcfx_result* out, hipStream_t st, int32_t shared_valid = 0);
// S <= SCREEN_MULTI_MAX tables in one pass over the reference (k = 31, M = 16 minimizer tables with one strandedness);
// sample s writes bits + s * n_tiles * TILE_BITS_U64 and count_sums + s * n_tiles
constexpr int SCREEN_MULTI_MAX = 8;
bool screen_multi_supported(const TableView& tv, int32_t k);
hipError_t launch_screen_multi(const uint8_t* text, const TileDesc* tiles, int32_t n_tiles, const TableView* tvs,
                               int32_t n_tables, int32_t both_strands, int32_t min_count, uint64_t* bits, int64_t* count_sums,
                               hipStream_t st, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);


}  // namespace kcfx
