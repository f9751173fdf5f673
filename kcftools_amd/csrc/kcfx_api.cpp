// kcfx_api.cpp -- host side of libkcfx.so: the C ABI of include/kcfx.h.
//
// Handles mirror the reference's objects: kcfx_db <-> `KMC` (Data/KMC.java), kcfx_ref <-> the window list plus the
// FastaIndex-backed window text (Plugins/GetVariants.java:117-121,188-197). There is no CPU compute path here:
// every compute entry point launches the HIP kernels of kcfx_kernels.hip or fails.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <atomic>
#include <thread>
#include <vector>

#include "kcfx_internal.h"
#include "kmc_pre.h"

namespace kcfx {

static thread_local std::string g_error;
void set_error(const std::string& msg) { g_error = msg; }
int32_t hip_fail(hipError_t e, const char* what) {
    g_error = std::string("HIP error: ") + hipGetErrorString(e) + " in " + what;
    (void)hipGetLastError();
    return KCFX_ERR_DEVICE;
}
// Nothing may propagate out of an extern "C" entry point (the callers are C, ctypes, JNI): every one that allocates host
// memory, builds strings or starts threads is a function-try-block ending in this. Called inside catch (...).
int32_t exception_to_status(const char* fn) noexcept {
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            set_error(std::string(fn) + ": out of host memory");
        } catch (const std::exception& e) {
            set_error(std::string(fn) + ": " + e.what());
        } catch (...) {
            set_error(std::string(fn) + ": unexpected C++ exception");
        }
    } catch (...) {                                   // no memory even for the message
    }
    return KCFX_ERR_HOST;
}

static int32_t fail(int32_t code, const std::string& msg) {
    g_error = msg;
    return code;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace kcfx

using namespace kcfx;

// ------------------------------------------------------------------------------------------------------
// streams: creating one costs ~90 ms on this stack (a hardware queue behind it), which would dominate kcfx_db_open;
// handles borrow from a per-device pool and hand the stream back (idle) when they are destroyed
// ------------------------------------------------------------------------------------------------------
namespace {
std::mutex g_stream_mu;
std::vector<std::pair<int, hipStream_t>> g_stream_pool;

hipError_t stream_acquire(int device, hipStream_t* out) {
    {
        std::lock_guard<std::mutex> lk(g_stream_mu);
        for (size_t i = 0; i < g_stream_pool.size(); ++i)
            if (g_stream_pool[i].first == device) {
                *out = g_stream_pool[i].second;
                g_stream_pool.erase(g_stream_pool.begin() + (ptrdiff_t)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(out, hipStreamNonBlocking);
}
// databases only use their stream while loading and in kcfx_db_lookup: a few shared streams per device, for the process. A load takes
// one that no other load is running on (round 6: with ONE stream per device two loaders serialised completely -- every wait of one
// stood behind the other's queued work -- which is what round 4 measured as "two loads do not overlap"); a second stream is created
// only when two loads really run at the same time, and so is each stream's companion for the uploads (the copier of load_suffix_file).
struct DbStream {
    int device;
    hipStream_t st, copier;
    bool loading;
};
std::vector<DbStream> g_db_streams;
hipError_t db_stream(int device, hipStream_t* out, hipStream_t* copier_out, bool for_a_load) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& e : g_db_streams)
        if (e.device == device && !(for_a_load && e.loading)) {
            *out = e.st;
            if (copier_out) *copier_out = e.copier;
            if (for_a_load) e.loading = true;
            return hipSuccess;
        }
    hipStream_t copier = nullptr;
    hipError_t e = hipStreamCreateWithFlags(out, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&copier, hipStreamNonBlocking);
    if (e == hipSuccess) g_db_streams.push_back(DbStream{device, *out, copier, for_a_load});
    if (copier_out) *copier_out = copier;
    return e;
}
void db_stream_load_done(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& e : g_db_streams)
        if (e.st == st) e.loading = false;
}
hipStream_t db_stream_copier(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    for (auto& e : g_db_streams)
        if (e.st == st) return e.copier;
    return nullptr;
}
void stream_release(int device, hipStream_t st) {
    if (!st) return;
    (void)hipStreamSynchronize(st);
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_pool.emplace_back(device, st);
}
}  // namespace

// ------------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------------
struct kcfx_db {
    int32_t device = 0;
    kcfx_db_info_t info{};
    uint8_t* d_lines = nullptr;
    uint32_t n_lines = 0;
    int32_t wide = 0;
    int32_t mode = MODE_MINIMIZER;
    int32_t mm_len = 0;
    int32_t generic = 0;                 // KCFX_GENERIC_KERNEL at open time: screen with the general kernel
    int64_t expected_queries = 0;        // kcfx_db_opts: sizes the table (cells_lbits)
    int32_t lbits_cap = 0;               // > 0: cell tables of at most 2^lbits_cap lines (a retry after the policy's size did not fit)
    // kcfx_db_open allocates and clears the cell table (up to 48 GiB: hipMalloc alone can take a second when the driver scrubs
    // the block first) on a thread of its own while the .kmc_suf is read, decoded and counted: nothing touches the lines
    // before the first bidding pass, which waits for it (table_ready)
    int32_t ext = 0;                     // FORMAT_CELLS: an extended table (TableView::ext)
    bool ext_auto = false;               // ... decided by the build itself: plain first, rebuilt extended if too many k-mers lost their cell
    int32_t line_footprint = LINE_BYTES;   // bytes per table line the handle will hold in all (a cohort's key table: + 128 B of counters)
    std::thread alloc_thread;
    int32_t alloc_rc = KCFX_OK;          // written by alloc_thread, read after the join
    std::string alloc_error;
    bool alloc_oom = false;
    std::atomic<bool> alloc_go{false};       // the loader has taken its own buffers: the table's allocation may start (it holds the driver for seconds)
    std::atomic<bool> alloc_landed{false};   // alloc_thread: the table's memory has been taken (hipMemGetInfo no longer counts it as free)
    bool table_arrived = false;          // table_ready() has succeeded: a later device error is not about the table's size
    int32_t fmt = FORMAT_LINES;          // FORMAT_CELLS: d_lines holds 1 << lbits cell lines, d_stash the overflow k-mers
    int32_t lbits = 0;
    int32_t regions = 1;                 // FORMAT_CELLS: n_lines = regions << lbits (cell_line_index, kcfx_device.h)
    uint4* d_stash = nullptr;
    uint32_t stash_mask = 0;
    unsigned long long* d_stats = nullptr;   // FORMAT_CELLS build counters (cells_settle_kernel)
    uint32_t* d_pop = nullptr;               // FORMAT_CELLS build scratch: k-mers per minimizer (freed after the build)
    int32_t pop_bits = 32;                   // 2^pop_bits byte counters: 32 = one per minimizer, lbits + 2 when 4 GiB cannot be had
    int32_t crowd_by_minimizer = 0;          // the counters (and the run of a bid) keyed by the minimizer alone, as before round 6 (hook KCFX_CROWD_BY_MINIMIZER; cell_crowd_key)
    hipStream_t stream = nullptr;
    int32_t n_slices = 1, slice = 0;         // kcfx_db_open_slice: the table holds slice `slice` of `n_slices` of the database (cell_slice_of)
    int32_t slicing() const { return n_slices > 1 ? (n_slices << 8 | slice) : 0; }
    TableView view() const { return TableView{d_lines, n_lines, wide, mode, mm_len, fmt, lbits, stash_mask, d_stash, generic, info.both_strands, ext, regions, n_slices, slice}; }
};

struct kcfx_ref {
    int32_t device = 0;
    int32_t k = 0;
    int32_t n_windows = 0;
    int32_t n_tiles = 0;
    int64_t n_slots = 0;
    int64_t text_len = 0;
    uint8_t* d_text = nullptr;
    uint32_t* d_codes = nullptr;       // RefPlan (k = 19 .. 31: what a cell table's look-ups need of the reference, worked out once;
    uint8_t* d_anchors = nullptr;      // both point into the block of d_text)
    RefPlan plan() const {
        RefPlan p;
        p.codes = d_codes;
        p.anchors = d_anchors;
        return p;
    }
    TileDesc* d_tiles = nullptr;
    int32_t* d_win_tile0 = nullptr;
    uint64_t* d_bits = nullptr;        // per tile: valid / hit ballots of every 64-position word
    int64_t* d_csum = nullptr;         // per tile: sum of counts over hits
    kcfx_result* d_res = nullptr;      // used by the synchronous entry point
    // multi-sample pass (allocated on first use; scratch sized for SCREEN_MULTI_MAX samples)
    uint64_t* d_mbits = nullptr;
    int64_t* d_mcsum = nullptr;
    int32_t multi_cap = 0;             // samples the multi-sample scratch holds
    kcfx_result* d_mres = nullptr;     // synchronous multi entry point: n_db * n_windows records
    size_t mres_bytes = 0;
    hipStream_t stream = nullptr;
    uint64_t win_hash = 0;             // fingerprint of the window list (begin, len): two references are mergeable when they agree (kcfx_ref_merge_from)
    hipEvent_t partial_done = nullptr; // recorded behind every partial screen on the stream it was launched on: the merge waits for it
    // measurement
    bool prof = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> plan_events;     // kcfx_ref_plan_async launches, when prof is on
    size_t plan_events_used = 0;
};

struct kcfx_cohort {
    int32_t device = 0;
    int32_t n_samples = 0;
    kcfx_db* keys = nullptr;           // the key table: an ordinary FORMAT_CELLS handle whose count fields are all 1
    uint8_t* d_counts = nullptr;       // [16 * lines][COHORT_STRIDE], 8 bits each
    uint16_t* d_scounts = nullptr;     // [stash slots][COHORT_STRIDE]
    kcfx_cohort_info_t info{};
};

namespace {

struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int32_t check_device(int32_t device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(KCFX_ERR_DEVICE, "no HIP device available (libkcfx has no CPU fallback)");
    }
    if (device < 0 || device >= n) return fail(KCFX_ERR_ARG, "device index out of range");
    return KCFX_OK;
}

// Lines are filled sparsely on purpose. The probe costs the same whatever the table size (one line per fetch, 8 lanes
// at once), and every overflowed line sends its k-mers to a further probe round (measured on the bench workload with
// the queued-rounds kernel: 0.199 / 0.210 / 0.221 / 0.239 ms at load 0.15 / 0.20 / 0.30 / 0.40), so a sparse table is
// worth its HBM while HBM is plentiful (288 GB): the sparsest of 0.15 / 0.35 / 0.60 whose table stays under 48 / 128 GiB
// and under 60 % of the free device memory (several samples may be resident at once). KCFX_TABLE_LOAD overrides.
double table_load_factor(int64_t n_keys, int slots) {
    const char* s = hook("KCFX_TABLE_LOAD");
    if (s) {
        const double a = atof(s);
        if (a >= 0.05 && a <= 0.9) return a;
    }
    const size_t free_b = free_device_bytes();
    const double GiB = (double)(1ull << 30);
    const double room = 0.6 * (double)free_b;
    auto bytes_at = [&](double a) { return (double)n_keys / (slots * a) * LINE_BYTES; };
    if (bytes_at(0.15) <= 48.0 * GiB && bytes_at(0.15) <= room) return 0.15;
    if (bytes_at(0.35) <= 128.0 * GiB && bytes_at(0.35) <= room) return 0.35;
    return 0.6;
}

// KCFX_TABLE_MODE=keyhash selects per-k-mer hashing (A/B against the default minimizer bucketing)
int32_t table_mode() {
    const char* s = hook("KCFX_TABLE_MODE");
    return (s && (!strcmp(s, "keyhash") || !strcmp(s, "0"))) ? MODE_KEYHASH : MODE_MINIMIZER;
}

// Minimizer length M: long enough that distinct minimizers rarely share an M-mer by chance (4^M >> 16 x keys),
// short enough to keep runs long (run length ~ (k - M + 2) / 2).
int32_t minimizer_len(int32_t k, int64_t n_keys) {
    const char* s = hook("KCFX_MM_LEN");
    int m;
    if (s && atoi(s) > 0) {
        m = atoi(s);
    } else {
        m = 11;
        while (m < 20 && (double)(1ull << (2 * m)) < 16.0 * (double)std::max<int64_t>(n_keys, 1)) ++m;
        // k = 31 always takes M = 16, the length screen_k31_kernel is built around (a k-mer spans 16 M-mers = one DPP
        // row). Beyond ~270 M keys unrelated k-mers start to share minimizer values, which only merges buckets the
        // way any hash collision does; the general kernel a longer M would select costs 1.8x the instructions
        // (measured on a 600 M-k-mer database: 0.34 ms with M = 17 against 0.25 ms with M = 16).
        if (k == 31) m = 16;
    }
    return std::max(1, std::min(m, std::min(k, 32)));
}

// largest lbits a k-mer length allows: the line number is the top of a 32-bit mix of the (k - 15)-base minimizer
int32_t cells_lbits_cap(int32_t) { return std::min<int32_t>(CELL_LBITS_MAX, 2 * 16 - 2); }

// FORMAT_CELLS (kcfx_internal.h) is what the BASELINE configuration gets: k = 27 .. 31 (31 in BASELINE.json), both strands or forward-only,
// any counter size, minimizer bucketing. KCFX_TABLE_FORMAT=lines keeps the 12-slot line table (A/B and test
// hook, read at open time).
bool cells_eligible(const kcfx_db* db, int32_t counter_size) {
    const char* s = hook("KCFX_TABLE_FORMAT");
    if (s && !strcmp(s, "lines")) return false;
    const int32_t k = db->info.kmer_length;
    // (3- and 4-byte counters too: a counter beyond 65535 does not fit a cell, that k-mer lives in the stash)
    // (forward-only databases too: bit 14 of r tells a k-mer from its reverse complement)
    if (k < CELL_K_MIN || k > CELL_K_MAX || table_mode() != MODE_MINIMIZER ||
        hook("KCFX_MM_LEN") != nullptr)
        return false;
    // the minimizer is k - 15 bases: a line number cannot carry more than its 2 (k - 15) bits
    return db->info.total_kmers / std::max(1, db->n_slices) <= (int64_t)16 << cells_lbits_cap(k);   // (a slice holds 1 / n of the k-mers)
}

// 2^lbits lines of 16 cells: memory (and the time the driver takes to hand it out) against the speed of a screening pass.
// A denser table sends more k-mers to second / third lines and the stash, and every such look-up is a random request of its own
// (bench workload, 150 M k-mers, screening kernel on a planned reference, same box, round 6 -- the committed profiles/r06_density.csv
// is a later run on another box: 0.0763 / 0.0785 / 0.0819 / 0.0944 ms; the round-5 code: 0.0797 / 0.0833 / 0.0921 / 0.1142 ms):
//     2^27 lines  0.070 per cell  17.2 GB   5.6 % outside their first-choice cell  0.03 % in the stash  0.0750 ms per pass
//     2^26        0.140            8.6 GB   8.7 %                                   0.19 %               0.0773 ms
//     2^25        0.280            4.4 GB  13.6 %                                   1.09 %               0.0810 ms
//     2^24        0.559            3.2 GB  18.9 %                                   6.00 % (1 GiB stash) 0.0973 ms
// What the memory costs is not the clearing (a few ms of hipMemset) but the ALLOCATION on a device that other processes have used:
// the driver hands out memory that a finished process has just given back only after clearing it, at 28 - 42 GB/s, and while it
// does so the uploads of the loading pass stand still (profiles/r06_notes.md section 1: the one-sample job end to end, 2^27 lines:
// 0.47 - 0.53 s on an idle device, 2.8 - 3.1 s right after a process that had written 64 GiB; 2^25 lines: 0.45 - 0.55 s either way.
// Rounds 4 and 5 gave every handle 2^27 lines "because the sizes are indistinguishable end to end" -- they are, on a device nobody
// else has just used; the review's box was not one: 0.98 - 1.49 s in three runs of four).
// So the size follows the look-ups the handle will serve (kcfx_db_opts.expected_queries = reference k-mers x passes):
//     cost(L) = bytes(L) / ALLOC_BYTES_PER_S + expected_queries x seconds_per_query(k-mers per cell at L)
// minimised over the sizes from "at most ~0.3 per cell" (the density the large databases get as well: cells_regions) to ~0.075 per
// cell. One pass of a 30 Mb reference: 2^25 lines; the sparse table pays for its 13 GB after ~10^12 look-ups (a resident engine
// that screens thousands of references, a cohort of hundreds of samples' worth of windows). expected_queries = 0 (unknown: a
// handle somebody keeps) gets the sparse table, as kcfx_db_open always did. KCFX_TABLE_LBITS overrides.
constexpr double QUERIES_FOR_SPARSE = 1e10;             // (pop_alloc: below this the build takes the cheaper crowding counters)
constexpr double ALLOC_BYTES_PER_S = 35e9;              // what a used device hands out per second (profiles/r06_e2e_policy.log)
constexpr double DENSITY_DENSE = 0.30, DENSITY_SPARSE = 0.075;
// seconds per look-up at d k-mers per cell: the table above, linear in log2(d) between its rows (below 0.07 nothing more to gain)
double seconds_per_query(double d) {
    static const double dd[4] = {0.070, 0.140, 0.280, 0.559}, tt[4] = {0.0750e-3 / 30.0e6, 0.0773e-3 / 30.0e6, 0.0810e-3 / 30.0e6, 0.0973e-3 / 30.0e6};
    if (d <= dd[0]) return tt[0];
    for (int i = 1; i < 4; ++i)
        if (d <= dd[i]) return tt[i - 1] + (tt[i] - tt[i - 1]) * std::log2(d / dd[i - 1]);
    return tt[3] * (d / dd[3]);                         // (denser than anything the policy picks by itself)
}
int32_t cells_lbits(int64_t n_keys, int32_t k, bool both, int64_t expected_queries, int32_t line_footprint, size_t free_b) {
    const int32_t cap = cells_lbits_cap(k);
    const int32_t lmin = cell_lbits_min(both);
    const char* s = hook("KCFX_TABLE_LBITS");
    if (s && atoi(s) >= lmin && atoi(s) <= cap) return atoi(s);
    // A table above 48 GiB is halved while that leaves it at <= ~0.5 k-mers per cell (the 2.1 G-k-mer test database: 2^28 lines), and
    // beyond that only when it does not fit 60 % of the free memory: a database of billions of GENOMIC k-mers needs the lines --
    // at ~1 per cell a quarter of it lives in the stash, at ~0.5 a twelfth (profiles/r04_notes.md) -- and a second database that
    // does not fit beside it is waited for by the sharded loader (kcfx_shard.cpp).
    const double soft_cap = 48.0 * (double)(1ull << 30), hard_cap = 0.6 * (double)free_b;
    auto per_cell = [&](int32_t l) { return (double)n_keys / (16.0 * (double)(1ull << l)); };
    auto bytes = [&](int32_t l) { return (double)(1ull << l) * line_footprint; };
    int32_t L = lmin;
    while (L < cap && per_cell(L) > DENSITY_SPARSE) ++L;
    if (expected_queries > 0) {
        // the cheapest size for that many look-ups, from the last power of two ABOVE ~0.3 per cell (cells_regions then grows it in
        // quarters to the first size at or under 0.3: the dense end of the policy is the same for every database) upwards
        int32_t Ld = lmin;
        while (Ld < L && per_cell(Ld + 1) > DENSITY_DENSE) ++Ld;
        // (a cohort's key table has no regions to grow in: its dense end is the first power of two at or under ~0.3 per cell -- at the
        // 0.56 per cell below that the bench cohort kept 6 % of its records in the stash instead of 2 %)
        if (line_footprint != LINE_BYTES && Ld < L && per_cell(Ld) > DENSITY_DENSE) ++Ld;
        int32_t best = Ld;
        double best_cost = 0;
        for (int32_t l = Ld; l <= L; ++l) {
            // (what the handle gets at l: the region step of cells_regions when l is left above 0.3 per cell)
            double lines = (double)(1ull << l);
            if (per_cell(l) > DENSITY_DENSE && line_footprint == LINE_BYTES)
                for (const double f : {1.25, 1.5, 1.75, 2.0}) {
                    lines = f * (double)(1ull << l);
                    if ((double)n_keys / (16.0 * lines) <= DENSITY_DENSE) break;
                }
            const double cost = lines * line_footprint / ALLOC_BYTES_PER_S + (double)expected_queries * seconds_per_query((double)n_keys / (16.0 * lines));
            if (l == Ld || cost < best_cost) {
                best = l;
                best_cost = cost;
            }
        }
        L = best;
    }
    while (L > lmin && bytes(L) > soft_cap && 16.0 * (double)(1ull << (L - 1)) * 0.5 >= (double)n_keys) --L;
    while (L > lmin && bytes(L) > hard_cap && 16.0 * (double)(1ull << (L - 1)) >= (double)n_keys) --L;
    return L;
}

// Regions (kcfx_device.h: cell_line_index). Where the caps of cells_lbits left the table above 0.3 k-mers per cell it grows in
// quarters of its power of two -- 1.25 / 1.5 / 1.75 / 2 / 2.5 / 3 / 3.5 x 2^L0 lines = 5 / 3 / 7 / 1 / 5 / 3 / 7 regions of 2^(L0 - 2) ..
// 2^(L0 + 1) lines -- to the first size at or under 0.3 per cell, or the largest one its limit allows: a table under 48 GiB up
// to 48 GiB (a 2.1 G-k-mer database: 3 x 2^27 lines, 0.33 per cell instead of 0.49), a larger one up to 80 % of the free memory (8 G
// genomic k-mers: 3 x 2^29 lines = 192 GiB, 0.31 per cell instead of 0.47; the stash, the crowding counters and the streaming
// buffers of such a build take ~25 GiB beside it). Past ~0.2 per cell a genomic database gains nothing more
// (profiles/r04_notes.md). Cohort tables, retries after an allocation that did not fit and KCFX_TABLE_LBITS keep one region;
// KCFX_TABLE_REGIONS=R (1, 3, 5, 7; with KCFX_TABLE_LBITS=L: R << L lines) decides by hand (tests, A/B).
void cells_regions(int64_t n_keys, int32_t k, bool both, int32_t L0, size_t free_b, int32_t* regions, int32_t* lbits) {
    *regions = 1;
    *lbits = L0;
    const int32_t lmin = cell_lbits_min(both), cap = cells_lbits_cap(k);
    if (const char* rs = hook("KCFX_TABLE_REGIONS")) {
        const int v = atoi(rs);
        if ((v == 3 || v == 5 || v == 7) && ((uint64_t)v << L0) < (1ull << 32)) *regions = v;
        return;
    }
    auto density = [&](double lines) { return (double)n_keys / (16.0 * lines); };
    if (hook("KCFX_TABLE_LBITS") != nullptr || density((double)(1ull << L0)) <= 0.30) return;
    const double one = (double)(1ull << L0) * LINE_BYTES, gib48 = 48.0 * (double)(1ull << 30), most = 0.80 * (double)free_b;
    const double limit = one < gib48 ? std::min(gib48, most) : most;        // (a table under 48 GiB grows up to them)
    static const struct { int32_t r, dl; } steps[] = {{5, -2}, {3, -1}, {7, -2}, {1, 1}, {5, -1}, {3, 0}, {7, -1}};   // ascending sizes
    for (const auto& st : steps) {
        const int32_t L = L0 + st.dl;
        if (L < lmin || L > cap || ((uint64_t)st.r << L) >= (1ull << 32)) continue;
        const double lines = (double)((uint64_t)st.r << L);
        if (lines * LINE_BYTES > limit) break;
        *regions = st.r;
        *lbits = L;
        if (density(lines) <= 0.30) break;
    }
}

// the deferred half of cells_alloc: join the allocation thread, report what it found
int32_t table_ready(kcfx_db* db) {
    db->alloc_go = true;
    if (db->alloc_thread.joinable()) db->alloc_thread.join();
    if (db->alloc_rc != KCFX_OK) return fail(db->alloc_rc, db->alloc_error);
    db->table_arrived = true;
    return KCFX_OK;
}

// the crowding counters of the build: one byte per minimizer (4 GiB) if the device can spare them beside the table it is about
// to hold, else one per line + 2 bits (round 2's classes)
int32_t pop_alloc(kcfx_db* db) {
    db->pop_bits = 32;
    hipError_t e = hipErrorOutOfMemory;
    // A handle that serves few look-ups (kcfx_db_opts.expected_queries: the command-line driver's one pass) takes the fallback by
    // policy: allocating 4 GiB costs 30 ms of its 0.5 s in a fresh process on an idle device and 120 ms on a used one, the
    // look-ups it saves a fraction of a millisecond.
    // (KCFX_CROWD_PER_LINE=1: the fallback by choice -- test hook, read when the handle is made)
    const bool few = db->expected_queries > 0 && (double)db->expected_queries < QUERIES_FOR_SPARSE;
    if (!few && hook("KCFX_CROWD_PER_LINE") == nullptr && free_device_bytes() > ((size_t)LINE_BYTES * db->n_lines) + ((size_t)6 << 30))
        e = dev_alloc(&db->d_pop, cells_pop_bytes(db->pop_bits));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        // (round 6: 2^30 counters at least -- 1 GiB, 30 ms of allocation where 4 GiB cost 120 -- since the counters are keyed by the
        // locus: at lbits + 2 bits a bucket of the bench database at 0.28 per cell holds 1.1 k-mers and every stranger looks crowded --
        // 2.2 M queued look-ups per pass; 2^29 / 2^30 / 2^31 / 2^32 counters: 0.92 / 0.88 / 0.86 / 0.86 M, profiles/r06_notes.md section 2)
        db->pop_bits = std::min(32, std::max(30, db->lbits + 2));
        if (const char* pb = hook("KCFX_POP_BITS")) db->pop_bits = std::max(db->lbits + 2, std::min(32, atoi(pb)));      // (experiments: the resolution of the cheap counters)
        if (dev_alloc(&db->d_pop, cells_pop_bytes(db->pop_bits)) != hipSuccess) {      // (a device that cannot spare even that: one counter per line + 2 bits)
            (void)hipGetLastError();
            db->pop_bits = std::min(32, db->lbits + 2);
            KCFX_HIP(dev_alloc(&db->d_pop, cells_pop_bytes(db->pop_bits)));
        }
    }
    KCFX_HIP(hipMemsetAsync(db->d_pop, 0, cells_pop_bytes(db->pop_bits), db->stream));
    return KCFX_OK;
}

int32_t cells_alloc(kcfx_db* db, int64_t n_keys, bool deferred = false) {
    db->fmt = FORMAT_CELLS;
    db->wide = 0;
    db->mode = MODE_MINIMIZER;
    db->mm_len = 16;                                  // cell tables: 16-base minimizers whatever k (k - 15 of them per k-mer)
    db->info.table_mode = MODE_MINIMIZER;
    db->info.minimizer_length = db->mm_len;
    db->info.table_format = FORMAT_CELLS;
    // Extended tables (kcfx_device.h: cell_first_ext) are for databases whose runs outnumber the open-syncmer 16-mers their
    // minimizers concentrate on. Whether that is so shows in the build: kcfx_db_open builds a database of >= 0.5 G k-mers as a
    // plain table up to the settle pass, and starts over as an extended one if more than EXT_MOVED_SHARE of the k-mers did not get
    // their first-choice cell (load_suffix_file). The query side pays ~1.3 line requests per run in an extended table, so a
    // database that does not need it must not get it: the bench cohort (1.2 G records, mostly k-mers private to one sample) screens
    // in 0.21 ms plain and 0.28 ms extended. KCFX_TABLE_EXT=0 / 1 decides by hand (tests, A/B). Cohort tables stay plain.
    db->crowd_by_minimizer = hook("KCFX_CROWD_BY_MINIMIZER") != nullptr;
    db->ext = 0;
    db->ext_auto = n_keys >= 500000000ll && db->line_footprint == LINE_BYTES;
    if (const char* xe = hook("KCFX_TABLE_EXT")) {
        db->ext = atoi(xe) != 0;
        db->ext_auto = false;
    }
    hipError_t e = hipErrorOutOfMemory;
    const bool both = db->info.both_strands != 0;
    const size_t free_b = free_device_bytes();
    int32_t L0 = cells_lbits(n_keys, db->info.kmer_length, both, db->expected_queries, db->line_footprint, free_b);
    if (db->lbits_cap > 0) L0 = std::min(L0, db->lbits_cap);
    db->regions = 1;
    int32_t R = 1, Lr = L0;
    if (db->line_footprint == LINE_BYTES && db->lbits_cap == 0) cells_regions(n_keys, db->info.kmer_length, both, L0, free_b, &R, &Lr);
    db->alloc_rc = KCFX_OK;
    db->alloc_oom = false;
    db->alloc_landed = false;
    db->alloc_go = false;
    db->table_arrived = false;
    if (deferred && Lr >= cell_lbits_min(both) && (double)n_keys <= 16.0 * (double)((uint64_t)R << Lr)) {
        // the policy's size, allocated and cleared beside the file streaming; if it does not fit, kcfx_db_open starts over
        // with denser tables, synchronously
        db->lbits = Lr;
        db->regions = R;
        db->n_lines = (uint32_t)R << Lr;
        KCFX_HIP(dev_alloc(&db->d_stats, 4 * sizeof(unsigned long long)));
        KCFX_HIP(hipMemsetAsync(db->d_stats, 0, 4 * sizeof(unsigned long long), db->stream));
        if (int32_t rc = pop_alloc(db)) return rc;
        db->info.table_bytes = (int64_t)LINE_BYTES * (int64_t)db->n_lines;
        db->info.table_lines = db->n_lines;
        db->info.slots_per_line = 16;
        db->alloc_thread = std::thread([db]() {
            try {
                // (the driver serves one allocation at a time and a table of 200 GB holds it for 5 - 6 s: the loader's own buffers --
                // a few hundred MB, or the resident records -- go first, or its first pass over the file starts that much later;
                // alloc_go comes from load_suffix_file, from whoever waits for the table, or after five seconds)
                const bool early = hook("KCFX_LOAD_TABLE_EARLY") != nullptr;     // (the table before everything else, as in rounds 1 - 4: profiles/r05_notes.md section 5)
                for (int waited = 0; !early && !db->alloc_go && waited < 25000; ++waited) std::this_thread::sleep_for(std::chrono::microseconds(200));
                hipError_t e2 = hipSetDevice(db->device);
                if (e2 == hipSuccess) e2 = dev_alloc(&db->d_lines, (size_t)LINE_BYTES * db->n_lines);
                if (e2 == hipSuccess) db->alloc_landed = true;
                if (e2 == hipSuccess) e2 = hipMemsetAsync(db->d_lines, 0, (size_t)LINE_BYTES * db->n_lines, db->stream);
                if (e2 != hipSuccess) {
                    db->alloc_oom = (e2 == hipErrorOutOfMemory);
                    db->alloc_rc = KCFX_ERR_DEVICE;
                    db->alloc_error = std::string("HIP error: ") + hipGetErrorString(e2) + " in the allocation of the k-mer table";
                    (void)hipGetLastError();
                }
            } catch (...) {
                db->alloc_rc = KCFX_ERR_HOST;
                db->alloc_error = "kcfx_db_open: exception on the table allocation thread";
            }
        });
        return KCFX_OK;
    }
    if (R > 1 && (double)n_keys <= 16.0 * (double)((uint64_t)R << Lr)) {       // the regions first; if they do not fit: powers of two
        e = dev_alloc(&db->d_lines, ((size_t)LINE_BYTES * R) << Lr);
        if (e == hipSuccess) {
            db->lbits = Lr;
            db->regions = R;
        } else {
            (void)hipGetLastError();
            db->d_lines = nullptr;
        }
    }
    // (the power of two at or below the policy's size, then denser ones)
    const int32_t Lf = std::min(cells_lbits_cap(db->info.kmer_length), R == 1 ? Lr : Lr + (R == 3 ? 1 : 2));
    for (int32_t L = Lf; e != hipSuccess && L >= cell_lbits_min(both); --L) {
        if ((double)n_keys > 16.0 * (double)(1ull << L))            // fewer cells than k-mers: not a cell table any more
            return fail(KCFX_ERR_DEVICE, "no cell table of 2^" + std::to_string(L) + " or more lines fits the device memory (or KCFX_TABLE_LBITS)");
        e = dev_alloc(&db->d_lines, (size_t)LINE_BYTES << L);
        if (e == hipErrorOutOfMemory) {                     // denser tables before giving up
            (void)hipGetLastError();
            db->d_lines = nullptr;
            continue;
        }
        db->lbits = L;
        break;
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc of the k-mer table");
    db->n_lines = (uint32_t)db->regions << db->lbits;
    KCFX_HIP(hipMemsetAsync(db->d_lines, 0, (size_t)LINE_BYTES * db->n_lines, db->stream));
    KCFX_HIP(dev_alloc(&db->d_stats, 4 * sizeof(unsigned long long)));
    KCFX_HIP(hipMemsetAsync(db->d_stats, 0, 4 * sizeof(unsigned long long), db->stream));
    if (int32_t rc = pop_alloc(db)) return rc;
    db->info.table_bytes = (int64_t)LINE_BYTES * (int64_t)db->n_lines;
    db->info.table_lines = db->n_lines;
    db->info.slots_per_line = 16;
    return KCFX_OK;
}

// between the two build passes of a FORMAT_CELLS table: read how many k-mers lost their cell, size and clear the stash.
// Open addressing with linear probing (stash_find): load <= 0.25 (~1.2 entries looked at per lookup) while that fits in
// half of the free device memory (KCFX_STASH_BUDGET=<bytes> overrides the budget), else denser, down to load 0.8; a stash
// that does not fit even then fails with KCFX_ERR_DEVICE and the caller rebuilds the database as a 12-slot line table.
int32_t cells_stash_alloc(kcfx_db* db) {
    unsigned long long h[4] = {0, 0, 0, 0};
    KCFX_HIP(hipMemcpyAsync(h, db->d_stats, sizeof h, hipMemcpyDeviceToHost, db->stream));
    KCFX_HIP(hipStreamSynchronize(db->stream));
    db->info.stash_kmers = (int64_t)h[0];
    db->info.stored_kmers = (int64_t)h[1];
    db->info.multi_cell_kmers = (int64_t)h[2];
    db->info.second_choice_kmers = (int64_t)h[3];
    db->info.extended = db->ext;
    db->info.regions = db->regions;
    const size_t free_b = free_device_bytes();
    double budget = 0.5 * (double)free_b;
    if (const char* s = hook("KCFX_STASH_BUDGET")) budget = std::max(0.0, atof(s));
    uint64_t slots = 1024;
    while (slots < 4 * h[0]) slots <<= 1;
    while (slots > 1024 && (double)slots * sizeof(uint4) > budget && (double)(slots >> 1) * 0.8 >= (double)h[0]) slots >>= 1;
    if ((double)slots * sizeof(uint4) > budget && slots > 1024)
        return fail(KCFX_ERR_DEVICE, "the stash of the cell table (" + std::to_string(h[0]) + " k-mers) does not fit the device memory budget");
    if (slots > (1ull << 32)) return fail(KCFX_ERR_DEVICE, "stash too large for a 32-bit slot index");
    db->stash_mask = (uint32_t)(slots - 1);
    KCFX_HIP(dev_alloc(&db->d_stash, (size_t)slots * sizeof(uint4)));
    KCFX_HIP(hipMemsetAsync(db->d_stash, 0xFF, (size_t)slots * sizeof(uint4), db->stream));
    db->info.stash_bytes = (int64_t)(slots * sizeof(uint4));
    db->info.table_bytes += db->info.stash_bytes;
    return KCFX_OK;
}

// back to "no table" (the handle keeps its stream and header fields): before a rebuild in the other format
void table_release(kcfx_db* db) {
    db->alloc_go = true;
    if (db->alloc_thread.joinable()) db->alloc_thread.join();
    (void)hipStreamSynchronize(db->stream);
    if (db->d_lines) (void)dev_free(db->d_lines);
    if (db->d_stash) (void)dev_free(db->d_stash);
    if (db->d_stats) (void)dev_free(db->d_stats);
    if (db->d_pop) (void)dev_free(db->d_pop);
    db->d_lines = nullptr;
    db->d_stash = nullptr;
    db->d_stats = nullptr;
    db->d_pop = nullptr;
    db->fmt = FORMAT_LINES;
    db->lbits = 0;
    db->regions = 1;
    db->stash_mask = 0;
    db->n_lines = 0;
    db->info.table_format = FORMAT_LINES;
    db->info.table_bytes = db->info.stash_bytes = 0;
    db->info.stash_kmers = db->info.stored_kmers = db->info.multi_cell_kmers = db->info.second_choice_kmers = 0;
    (void)hipGetLastError();
}

int32_t table_alloc(kcfx_db* db, int64_t n_keys, int32_t counter_size, bool allow_cells = true, bool deferred = false) {
    db->generic = hook("KCFX_GENERIC_KERNEL") != nullptr;   // A/B and test hook: read when the handle is made, never per launch
    if (allow_cells && cells_eligible(db, counter_size) && !db->generic) return cells_alloc(db, n_keys, deferred);
    db->wide = counter_size > 2 ? 1 : 0;
    db->mode = table_mode();
    db->mm_len = minimizer_len(db->info.kmer_length, n_keys);
    db->info.table_mode = db->mode;
    db->info.minimizer_length = db->mode == MODE_MINIMIZER ? db->mm_len : 0;
    const int slots = db->wide ? SLOTS_U32 : SLOTS_U16;
    // the policy's load factor first; if the device cannot give that much memory after all, denser tables before giving up
    const double first = table_load_factor(n_keys, slots);
    const double tries[4] = {first, 0.35, 0.6, 0.85};
    hipError_t e = hipErrorOutOfMemory;
    for (int i = 0; i < 4 && e != hipSuccess; ++i) {
        if (i > 0 && tries[i] <= first) continue;           // only ever denser than the first attempt
        double want = (double)n_keys / (slots * tries[i]);
        if (want < 64) want = 64;
        if (want > 4294967295.0) return fail(KCFX_ERR_ARG, "database too large for a 32-bit line index");
        db->n_lines = (uint32_t)want + 1;
        e = dev_alloc(&db->d_lines, (size_t)db->n_lines * LINE_BYTES);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            db->d_lines = nullptr;
            continue;
        }
        break;
    }
    if (e != hipSuccess) return hip_fail(e, "hipMalloc of the k-mer table");
    KCFX_HIP(launch_table_clear(db->d_lines, db->n_lines, db->wide, db->stream));
    db->info.table_bytes = (int64_t)db->n_lines * LINE_BYTES;
    db->info.table_lines = db->n_lines;
    db->info.slots_per_line = slots;
    return KCFX_OK;
}

// table_finalize + statistics (heavy / overflowed line counts) into db->info
int32_t table_finish(kcfx_db* db) {
    if (db->fmt == FORMAT_CELLS) {
        KCFX_HIP(hipStreamSynchronize(db->stream));
        if (db->d_stats) (void)dev_free_idle(db->d_stats);
        if (db->d_pop) (void)dev_free_idle(db->d_pop);
        db->d_stats = nullptr;
        db->d_pop = nullptr;
        return KCFX_OK;
    }
    unsigned long long* d_stats = nullptr;
    unsigned long long h_stats[2] = {0, 0};
    KCFX_HIP(dev_alloc(&d_stats, sizeof h_stats));
    hipError_t e = hipMemsetAsync(d_stats, 0, sizeof h_stats, db->stream);
    if (e == hipSuccess) e = launch_table_finalize(db->d_lines, db->n_lines, db->wide, d_stats, db->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_stats, d_stats, sizeof h_stats, hipMemcpyDeviceToHost, db->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
    (void)dev_free(d_stats);
    if (e != hipSuccess) return hip_fail(e, "table_finalize");
    db->info.heavy_lines = (int64_t)h_stats[0];
    db->info.overflow_lines = (int64_t)h_stats[1];
    return KCFX_OK;
}

// idle: the caller has waited for every launch that read the table (kcfx_screen_sharded's worker: its screens are synchronous) --
// no device-wide synchronisation, which would wait for whatever the NEXT databases' loaders have in flight on the device and kept
// the worker's "in use" slot taken for the length of their loads: the reason two loaders never overlapped (profiles/r06_notes.md 7)
void db_free(kcfx_db* db, bool idle = false) {
    if (!db) return;
    db->alloc_go = true;
    if (db->alloc_thread.joinable()) db->alloc_thread.join();
    DeviceGuard g(db->device);
    if (db->stream) (void)hipStreamSynchronize(db->stream);   // shared per device (db_stream), never destroyed
    auto drop = [&](void* p) { if (p) (void)(idle ? dev_free_idle(p) : dev_free(p)); };
    drop(db->d_lines);
    drop(db->d_stash);
    drop(db->d_stats);
    drop(db->d_pop);
    delete db;
}

void ref_free(kcfx_ref* r) {
    if (!r) return;
    DeviceGuard g(r->device);
    for (auto& ev : r->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    for (auto& ev : r->plan_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (r->partial_done) (void)hipEventDestroy(r->partial_done);
    if (r->d_text) (void)dev_free(r->d_text);
    if (r->d_tiles) (void)dev_free(r->d_tiles);
    if (r->d_win_tile0) (void)dev_free(r->d_win_tile0);
    if (r->d_bits) (void)dev_free(r->d_bits);
    if (r->d_csum) (void)dev_free(r->d_csum);
    if (r->d_res) (void)dev_free(r->d_res);
    if (r->d_mbits) (void)dev_free(r->d_mbits);
    if (r->d_mcsum) (void)dev_free(r->d_mcsum);
    if (r->d_mres) (void)dev_free(r->d_mres);
    stream_release(r->device, r->stream);
    delete r;
}

// KCFX_LOAD_TRACE=1: phase timings of kcfx_db_open on stderr
struct LoadTrace {
    const char* tag = "load";
    bool on = getenv("KCFX_LOAD_TRACE") != nullptr;
    double t = now_s();
    LoadTrace() = default;
    LoadTrace(const char* tag_, const char* env) : tag(tag_), on(getenv(env) != nullptr) {}
    void mark(const char* what) {
        if (!on) return;
        const double n = now_s();
        fprintf(stderr, "[kcfx %s] %-28s %8.1f ms\n", tag, what, (n - t) * 1e3);
        t = n;
    }
};

}  // namespace

// ---- one .kmc_suf read ONCE for the slices that a node's devices build at the same time (round 6) --------------------------------
// kcfx_db_open_slice makes every device walk the whole file in every build pass; G loaders of their own read G x the file through one
// page cache (46 GB/s on the boxes seen: eight slices of a 65 GB file = 11 s of reading per pass where one device takes 1.4 s).
// The loaders of one call share this instead: a small cache of pinned (portable) chunk buffers keyed by the chunk's index in the
// file. The first loader that asks for a chunk reads it (parallel_pread), the others find it there and start their own
// host-to-device copy from the same buffer -- every device has its own link. A loader holds at most two chunks (its double
// buffering); a buffer is reused when nobody holds it. Loaders that drift apart by more than the cache holds simply read for
// themselves again: nothing waits for a loader that never comes (the passes a slice makes may differ from its siblings').
namespace kcfx {
struct SuffixCache {
    struct Buf {
        uint8_t* p = nullptr;
        int64_t chunk = -1;
        int refs = 0;                    // loaders copying from it now
        int taken = 0;                   // loaders that have had it since it was read
        bool ready = false, loading = false;
        uint64_t stamp = 0;
    };
    int fd = -1;
    size_t cap = 0;                      // bytes per buffer
    int consumers = 1;                   // loaders still at work (leave())
    std::vector<Buf> bufs;
    std::mutex mu;
    std::condition_variable cv;
    uint64_t clock = 0;
    int64_t reads = 0, hits = 0, bytes_read = 0;
    // the chunk's bytes in pinned memory, or null on a read error; *slot names the buffer for release().
    // A buffer that not every loader has had yet is given up only after a wait: the loaders walk the file in step (a leader waits
    // for the others at most a few chunks ahead) instead of reading the same chunks one after the other -- but nobody waits for
    // a loader that never comes (a slice that makes a pass its siblings do not: it reads for itself, after 50 ms).
    const uint8_t* acquire(int64_t chunk, int64_t offset, int64_t bytes, int* slot) {
        std::unique_lock<std::mutex> lk(mu);
        int patience = 1;
        for (;;) {
            int mine = -1, spare = -1, busy_spare = -1;
            for (int i = 0; i < (int)bufs.size(); ++i) {
                Buf& b = bufs[i];
                if (b.chunk == chunk && (b.ready || b.loading)) mine = i;
                else if (b.refs == 0 && !b.loading) {
                    if (b.chunk < 0 || b.taken >= consumers) {
                        if (spare < 0 || b.stamp < bufs[spare].stamp) spare = i;
                    } else if (busy_spare < 0 || b.stamp < bufs[busy_spare].stamp) busy_spare = i;
                }
            }
            if (mine >= 0) {
                if (bufs[mine].loading) {
                    cv.wait(lk);
                    continue;
                }
                bufs[mine].refs++;
                bufs[mine].taken++;
                bufs[mine].stamp = ++clock;
                ++hits;
                *slot = mine;
                return bufs[mine].p;
            }
            if (spare < 0 && busy_spare >= 0 && patience > 0) {      // (somebody has not had the oldest chunk yet: give them a moment)
                if (cv.wait_for(lk, std::chrono::milliseconds(50)) == std::cv_status::timeout) --patience;
                continue;
            }
            if (spare < 0) spare = busy_spare;
            if (spare < 0) {                               // (every buffer is held: 2 per loader at most, the cache has 2 G + 2)
                cv.wait(lk);
                continue;
            }
            Buf& b = bufs[spare];
            b.chunk = chunk;
            b.ready = false;
            b.loading = true;
            b.taken = 0;
            lk.unlock();
            const bool ok = (size_t)bytes <= cap && parallel_pread(fd, b.p, bytes, offset);
            lk.lock();
            b.loading = false;
            if (!ok) {
                b.chunk = -1;
                cv.notify_all();
                return nullptr;
            }
            b.ready = true;
            b.refs = 1;
            b.taken = 1;
            b.stamp = ++clock;
            ++reads;
            bytes_read += bytes;
            *slot = spare;
            cv.notify_all();
            return b.p;
        }
    }
    void release(int slot) {
        std::lock_guard<std::mutex> lk(mu);
        if (slot >= 0 && slot < (int)bufs.size() && bufs[slot].refs > 0) --bufs[slot].refs;
        cv.notify_all();
    }
    void leave() {                       // a loader is done (or has failed): the others stop counting on it
        std::lock_guard<std::mutex> lk(mu);
        if (consumers > 1) --consumers;
        cv.notify_all();
    }
};

int64_t load_chunk_records() {
    // records per chunk (KCFX_LOAD_CHUNK_RECORDS: test hook -- a few thousand records per chunk make a small database take many chunks,
    // both buffers reused, chunks at byte offsets that are no multiple of 4)
    int64_t ch = 4ll << 20;
    if (const char* ce = hook("KCFX_LOAD_CHUNK_RECORDS")) ch = std::max<int64_t>(64, std::min<int64_t>(ch, atoll(ce)));
    return ch;
}

SuffixCache* suffix_cache_create(const char* kmc_prefix, int32_t consumers) {
    KmcHeader h;
    std::string perr;
    if (parse_prefix_file(std::string(kmc_prefix) + ".kmc_pre", h, perr) != KCFX_OK) return nullptr;
    std::unique_ptr<SuffixCache> c(new SuffixCache());
    c->fd = open((std::string(kmc_prefix) + ".kmc_suf").c_str(), O_RDONLY);
    if (c->fd < 0) return nullptr;
    const int64_t ch = load_chunk_records();
    c->cap = (size_t)std::max<int64_t>(16, std::min<int64_t>(ch, std::max<int64_t>(h.total, 1)) * h.record_size);
    c->consumers = std::max(1, consumers);
    c->bufs.resize((size_t)(2 * std::max(1, consumers) + 4));
    for (auto& b : c->bufs)
        if (hipHostMalloc(reinterpret_cast<void**>(&b.p), c->cap, hipHostMallocPortable) != hipSuccess) {
            (void)hipGetLastError();
            suffix_cache_destroy(c.release());
            return nullptr;
        }
    return c.release();
}
void suffix_cache_destroy(SuffixCache* c) {
    if (!c) return;
    for (auto& b : c->bufs)
        if (b.p) (void)hipHostFree(b.p);
    if (c->fd >= 0) close(c->fd);
    delete c;
}
void suffix_cache_stats(const SuffixCache* c, int64_t* reads, int64_t* hits, int64_t* bytes_read) {
    if (reads) *reads = c ? c->reads : 0;
    if (hits) *hits = c ? c->hits : 0;
    if (bytes_read) *bytes_read = c ? c->bytes_read : 0;
}
}  // namespace kcfx

namespace {

// Stream the .kmc_suf records through the GPU once: upload -> decode_records -> (key, count) arrays.
//   minimizer tables need two passes over the k-mers (count per home line, mark heavy homes, insert). When the decoded
//   records fit beside the table (12 B per k-mer) they stay in HBM between the passes and the file is read once;
//   otherwise the file is streamed twice.
int32_t load_suffix_file(kcfx_db* db, const std::string& path, const KmcHeader& h, SuffixCache* shared = nullptr) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(KCFX_ERR_IO, "Error reading suffix buffers from file " + path);
    struct stat st;
    if (fstat(fd, &st) != 0 || (int64_t)st.st_size < 4 + h.total * h.record_size) {
        close(fd);
        return fail(KCFX_ERR_IO, "suffix file shorter than total_kmers * record_size: " + path);
    }
    char marker[4];
    if (pread(fd, marker, 4, 0) != 4 || memcmp(marker, "KMCS", 4) != 0) {
        close(fd);
        return fail(KCFX_ERR_FORMAT, "missing KMCS marker in " + path);
    }
    LoadTrace tr;
    const int64_t CH = load_chunk_records();    // records per chunk
    const int64_t chunk_bytes = CH * h.record_size;
    uint8_t* h_buf[2] = {nullptr, nullptr};
    uint8_t* d_rec[2] = {nullptr, nullptr};
    uint64_t* d_lut = nullptr;
    uint64_t* d_keys = nullptr;
    uint32_t* d_counts = nullptr;
    uint64_t* d_pw = nullptr;          // the cells of the resident records, kept by the first pass for the later ones (RecProbes)
    uint8_t* d_ps = nullptr;
    uint32_t* d_pop2 = nullptr;        // the crowding counters of the OTHER form of the table (extended), counted in the same first pass
    LostList lost;                     // the k-mers the settle pass counts for the stash, written down (records not resident: see below)
    hipEvent_t done[2] = {nullptr, nullptr};       // the copy of chunk buffer [i] to the device has finished (recorded on `copier`)
    hipEvent_t decoded[2] = {nullptr, nullptr};    // the decode of d_rec[i] has finished (recorded on db->stream)
    // The uploads run on a stream of their own (round 5): chunk n + 1 crosses PCIe while chunk n is decoded and applied -- on
    // db->stream alone the copy waited for the table pass of the chunk before it, and a streamed pass over a 65 GB file took 2.5 s
    // where the link and the page cache deliver it in 1.5 s (tools: 56 GB/s H2D, 60 GB/s pread, 45 GB/s both at once).
    hipStream_t copier = nullptr;
    int held[2] = {-1, -1};                        // (shared chunk cache: the buffers this loader's two turns copy from)
    int32_t rc = KCFX_OK;
    auto cleanup = [&]() {
        // the temporaries are used on db->stream and the copier only: wait for those, not for the device (the other worker's
        // screening kernel, the next database's uploads)
        if (copier) (void)hipStreamSynchronize(copier);
        (void)hipStreamSynchronize(db->stream);
        for (int i = 0; i < 2; ++i)
            if (shared && held[i] >= 0) {
                shared->release(held[i]);
                held[i] = -1;
            }
        copier = nullptr;                                     // (the stream pair stays with the process: db_stream)
        if (lost.keys) (void)dev_free_idle(lost.keys);
        if (lost.counts) (void)dev_free_idle(lost.counts);
        if (lost.n) (void)dev_free_idle(lost.n);
        if (d_pop2) (void)dev_free_idle(d_pop2);
        if (d_pw) (void)dev_free_idle(d_pw);
        if (d_ps) (void)dev_free_idle(d_ps);
        for (int i = 0; i < 2; ++i) {
            if (h_buf[i]) (void)host_free(h_buf[i]);
            if (d_rec[i]) (void)dev_free_idle(d_rec[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            if (decoded[i]) (void)hipEventDestroy(decoded[i]);
        }
        if (d_lut) (void)dev_free_idle(d_lut);
        if (d_keys) (void)dev_free_idle(d_keys);
        if (d_counts) (void)dev_free_idle(d_counts);
        close(fd);
    };
#define LOAD_HIP(call)                                  \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) {                        \
            rc = hip_fail(e__, #call);                  \
            cleanup();                                  \
            return rc;                                  \
        }                                               \
    } while (0)
    const bool cells = (db->fmt == FORMAT_CELLS);
    const bool two_phase = (db->mode == MODE_MINIMIZER);
    // keep every decoded record resident between the phases?
    // (the table may still be on its way -- table_ready -- and must not be counted as free room; once its allocation has landed
    // hipMemGetInfo no longer counts it and subtracting it again -- round 4 did -- left a wheat-scale open no tier that fit. `landed`
    // is read BEFORE the free memory: an allocation that lands between the two reads is then subtracted once too often -- the
    // conservative side -- never counted as free)
    const bool landed = db->alloc_landed.load();
    size_t free_b = free_device_bytes();
    if (db->alloc_thread.joinable() && !landed) free_b -= std::min<size_t>(free_b, (size_t)LINE_BYTES * db->n_lines);
    const bool resident = two_phase && hook("KCFX_LOAD_STREAM_TWICE") == nullptr &&
                          (double)h.total * 12.0 + (double)(2 * chunk_bytes) + (double)(64 << 20) < 0.5 * (double)free_b;
    // (Round 5 built a tier between "everything decoded stays" and "the file is streamed once per pass" -- the RECORDS resident as the
    // file holds them, decoded again per pass -- and measured it slower than streaming at the size it was meant for: allocating 65 GB
    // costs 4 - 5 s on this platform, 20.0 s against 18.7 s for the wheat-scale open, profiles/r05_notes.md section 5. Removed in round 6.)
    const int64_t n_chunk_max = std::min<int64_t>(CH, std::max<int64_t>(h.total, 1));
    const int64_t n_keys_buf = resident ? std::max<int64_t>(h.total, 1) : n_chunk_max;
    for (int i = 0; i < 2; ++i) {
        if (!shared) LOAD_HIP(host_alloc(&h_buf[i], (size_t)std::min(chunk_bytes, std::max<int64_t>(h.total * h.record_size, 16))));
        LOAD_HIP(dev_alloc(&d_rec[i], (size_t)n_chunk_max * h.record_size + 16));
        LOAD_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
        LOAD_HIP(hipEventCreateWithFlags(&decoded[i], hipEventDisableTiming));
    }
    copier = db_stream_copier(db->stream);                  // (created with the handle's stream; null only if that failed: uploads on db->stream then)
    LOAD_HIP(dev_alloc(&d_lut, (size_t)std::max<int64_t>(h.lut_entries, 1) * 8));
    LOAD_HIP(dev_alloc(&d_keys, (size_t)n_keys_buf * 8));
    LOAD_HIP(dev_alloc(&d_counts, (size_t)n_keys_buf * 4));
    // cell tables, records resident: 9 more bytes per record buy four of the five passes their ~600 instructions per k-mer
    // (KCFX_LOAD_NO_PROBES=1: every pass derives the cells itself, as a streamed-twice load has to -- A/B and test hook)
    const bool keep_probes = cells && resident && hook("KCFX_LOAD_NO_PROBES") == nullptr &&
                             (double)h.total * 21.0 + (double)(2 * chunk_bytes) + (double)(64 << 20) < 0.5 * (double)free_b;
    if (keep_probes) {
        LOAD_HIP(dev_alloc(&d_pw, (size_t)n_keys_buf * 8));
        LOAD_HIP(dev_alloc(&d_ps, (size_t)n_keys_buf));
    }
    // (a database that may want an extended table -- ext_auto -- counts its k-mers per minimizer BOTH ways in the first pass, the
    // second way under the file's copies: the choice then costs no pass of its own)
    const bool early_ext = cells && db->ext_auto && !db->ext && db->pop_bits == 32;
    // what pass 0 counts its k-mers by (cell_crowd_key): the LOCUS -- except while the choice of an extended table is still to be made,
    // which is estimated from the k-mers per MINIMIZER; the locus counters are then the second set (d_pop2: the same key as an
    // extended table's), or counted in a pass of their own once the choice is made
    int32_t count_by_min = (early_ext || db->crowd_by_minimizer) ? 1 : 0;
    if (early_ext && !resident) {
        if (dev_alloc(&d_pop2, cells_pop_bytes(db->pop_bits)) != hipSuccess) {
            (void)hipGetLastError();
            d_pop2 = nullptr;
        } else {
            LOAD_HIP(hipMemsetAsync(d_pop2, 0, cells_pop_bytes(db->pop_bits), db->stream));
        }
    }
    if (h.lut_entries) LOAD_HIP(hipMemcpyAsync(d_lut, h.lut_bytes, (size_t)h.lut_entries * 8, hipMemcpyHostToDevice, db->stream));
    db->alloc_go = true;                                    // (the table's allocation, if it is still to come: from here on)
    tr.mark("buffers + LUT upload");

    // One pass of the build over a batch of decoded records. Line tables: 0 = count per home line (minimizer tables only),
    // 1 = insert. Cell tables: 0 = count per minimizer sub-bucket, 1 / 2 = claim the first-choice cells / write the winners' cell words, 3 = settle
    // the losers, 4 = fill the stash.
    auto apply = [&](int phase, const uint64_t* keys, const uint32_t* counts, int64_t n) -> hipError_t {
        if (cells) {
            const int32_t kk = db->info.kmer_length;
            // (the kept cells of these records: the batch's offset in the resident arrays is its offset in d_keys)
            uint64_t* pw = d_pw ? d_pw + (keys - d_keys) : nullptr;
            uint8_t* ps = d_ps ? d_ps + (keys - d_keys) : nullptr;
            if (phase == 0) {
                if (d_pop2 && !db->ext) {                   // ... and per locus = per minimizer of the extended form (see early_ext)
                    const hipError_t e0 = launch_cells_count(d_pop2, db->pop_bits, kk, db->lbits, db->info.both_strands, keys, counts, n, db->stream, 0, 1);
                    if (e0 != hipSuccess) return e0;
                }
                return pw ? launch_cells_resolve(db->d_pop, db->pop_bits, kk, db->lbits, db->info.both_strands, keys, counts, n, pw, ps, db->stream, db->ext, count_by_min)
                          : launch_cells_count(db->d_pop, db->pop_bits, kk, db->lbits, db->info.both_strands, keys, counts, n, db->stream, 0, db->ext, count_by_min);
            }
            if (phase == 1 || phase == 2) return launch_cells_insert(db->d_lines, db->d_pop, db->pop_bits, phase == 1 ? 1 : 0, kk, db->lbits, db->info.both_strands, keys, counts, n, db->stream, 0, pw, ps, db->ext, db->regions, db->slicing(), db->crowd_by_minimizer);
            if (phase == 3) return launch_cells_settle(db->d_lines, kk, db->lbits, db->info.both_strands, keys, counts, n, db->d_stats, db->stream, 0, pw, ps, db->ext, db->regions, db->slicing(), lost);
            return launch_cells_stash(db->d_lines, kk, db->lbits, db->info.both_strands, keys, counts, n, db->d_stash, db->stash_mask, db->stream, 0, pw, ps, db->ext, db->regions, db->slicing());
        }
        if (phase == 0) return launch_table_count(db->d_lines, db->view(), h.k, keys, n, db->stream);
        return launch_table_insert(db->d_lines, db->view(), h.k, keys, counts, n, db->stream);
    };
    auto stream_file = [&](int phase, bool keep) -> int32_t {
        int64_t first = 0;
        int turn = 0;
        bool used[2] = {false, false};
        hipStream_t const up = copier ? copier : db->stream;
        if (copier) {                                       // whatever db->stream still does with d_rec / the LUT upload comes first
            LOAD_HIP(hipEventRecord(decoded[0], db->stream));
            LOAD_HIP(hipStreamWaitEvent(copier, decoded[0], 0));
        }
        double t_wait = 0, t_read = 0, t_copy = 0, t_pass0 = tr.on ? now_s() : 0;     // (KCFX_LOAD_TRACE: where the host side of a pass spends its time)
        while (first < h.total) {
            const int64_t n = std::min(CH, h.total - first);
            const int64_t bytes = n * h.record_size;
            const double t_a = tr.on ? now_s() : 0;
            if (used[turn]) LOAD_HIP(hipEventSynchronize(done[turn]));      // the copy out of h_buf[turn] has finished
            const double t_b = tr.on ? now_s() : 0;
            const uint8_t* src = h_buf[turn];
            if (shared) {                                   // the chunk from the cache its sibling loaders share (read by whoever asks first)
                if (held[turn] >= 0) shared->release(held[turn]);
                held[turn] = -1;
                src = shared->acquire(first / CH, 4 + first * h.record_size, bytes, &held[turn]);
            } else if (!parallel_pread(fd, h_buf[turn], bytes, 4 + first * h.record_size)) {
                src = nullptr;
            }
            if (!src) {
                cleanup();
                return fail(KCFX_ERR_IO, "short read on " + path);
            }
            const double t_c = tr.on ? now_s() : 0;
            if (tr.on) {
                t_wait += t_b - t_a;
                t_read += t_c - t_b;
            }
            // d_rec[turn] was last read by the decode two chunks ago: before this copy in stream order, or (the copier) waited for
            uint8_t* const rec_at = d_rec[turn];
            if (copier && used[turn]) LOAD_HIP(hipStreamWaitEvent(copier, decoded[turn], 0));
            LOAD_HIP(hipMemcpyAsync(rec_at, src, (size_t)bytes, hipMemcpyHostToDevice, up));
            LOAD_HIP(hipEventRecord(done[turn], up));
            if (copier) LOAD_HIP(hipStreamWaitEvent(db->stream, done[turn], 0));
            used[turn] = true;
            if (tr.on) t_copy += now_s() - t_c;
            uint64_t* keys = d_keys + (keep ? first : 0);
            uint32_t* counts = d_counts + (keep ? first : 0);
            LOAD_HIP(launch_decode_records(rec_at, first, n, h.record_size, h.suffix_bytes, h.counter_size, h.p, d_lut,
                                           h.lut_entries, keys, counts, db->stream));
            if (copier) LOAD_HIP(hipEventRecord(decoded[turn], db->stream));
            LOAD_HIP(apply(phase, keys, counts, n));
            first += n;
            turn ^= 1;
        }
        if (shared)                                         // (the pass's last two chunks: back to the cache once their copies are done)
            for (int i = 0; i < 2; ++i)
                if (held[i] >= 0) {
                    LOAD_HIP(hipEventSynchronize(done[i]));
                    shared->release(held[i]);
                    held[i] = -1;
                }
        if (tr.on) fprintf(stderr, "[kcfx load]   pass %d over the file: host %.0f ms = %.0f waiting for the upload two chunks back + %.0f reading + %.0f enqueueing the upload + %.0f launching\n", phase,
                           (now_s() - t_pass0) * 1e3, t_wait * 1e3, t_read * 1e3, t_copy * 1e3, (now_s() - t_pass0 - t_wait - t_read - t_copy) * 1e3);
        return KCFX_OK;
    };
    // a later pass: straight from HBM when the decoded records were kept (launches of at most 2^30 records), else the
    // file is streamed again
    auto later_pass = [&](int phase) -> int32_t {
        if (resident) {
            // (round 6 tried launches of 8 M records instead of one per pass, so that another load's small per-chunk kernels would not
            // wait at the boundary of a 5 - 7 ms kernel: two loads on two streams then overlapped a little -- 128 - 166 ms for a pair
            // against 167 alone -- and a single load got no faster; the timeline says why two loaders gain so little: a load keeps
            // the copy engine busy for 43 ms and the CUs for 45 ms of its ~75, tools/load_overlap_probe.py, profiles/r06_notes.md 7)
            for (int64_t first = 0; first < h.total; first += (1ll << 30)) {
                const int64_t n = std::min<int64_t>(1ll << 30, h.total - first);
                LOAD_HIP(apply(phase, d_keys + first, d_counts + first, n));
            }
            return KCFX_OK;
        }
        LOAD_HIP(hipStreamSynchronize(db->stream));
        return stream_file(phase, false);
    };
    // Records that are not resident between the passes (the file is streamed once per pass): the settle pass writes the k-mers it counts
    // for the stash down, and the stash is filled from that list instead of from one more stream of the file (an 8 G-k-mer database:
    // 1.4 s of 12.6). Room for a sixteenth of the database, within a twentieth of the free memory (the stash itself comes out of the
    // same memory next); a list that overflows is ignored. KCFX_LOAD_NO_LOST_LIST=1: never (A/B and test hook).
    auto lost_list_alloc = [&]() {
        if (!cells || resident || lost.keys || hook("KCFX_LOAD_NO_LOST_LIST")) return;
        unsigned long long cap = (unsigned long long)std::max<int64_t>(h.total / 16, 1024);
        if (const char* s = hook("KCFX_LOAD_LOST_CAP")) cap = std::max(1ull, strtoull(s, nullptr, 10));          // (test hook: an overflowing list)
        else cap = std::min<unsigned long long>(cap, (unsigned long long)(0.05 * (double)free_device_bytes() / 12.0));
        if (cap == 0) return;
        if (dev_alloc(&lost.keys, (size_t)cap * 8) != hipSuccess || dev_alloc(&lost.counts, (size_t)cap * 4) != hipSuccess ||
            dev_alloc(&lost.n, sizeof(unsigned long long)) != hipSuccess || hipMemsetAsync(lost.n, 0, sizeof(unsigned long long), db->stream) != hipSuccess) {
            (void)hipGetLastError();
            if (lost.keys) (void)dev_free_idle(lost.keys);
            if (lost.counts) (void)dev_free_idle(lost.counts);
            if (lost.n) (void)dev_free_idle(lost.n);
            lost = LostList();
            return;
        }
        lost.cap = cap;
    };
    // Does this database want an EXTENDED table? Round 4 found out by building a plain one (four passes) and starting over when more
    // than EXT_MOVED_SHARE of the k-mers sat outside their first-choice cells. The crowding counters of pass 0 say as much before
    // any bid: a minimizer with c k-mers is c / L runs of ~L consecutive offsets (L = 0.65 W: the run length of the order) that
    // fall on the W cells of a line independently, so W (1 - (1 - L / W)^(c / L)) of its k-mers find a cell of their own and the rest
    // are displaced (8 G genomic k-mers, four loci per preferred 16-mer: c ~ 41 -> 61 %, measured 65 %; the bench database: ~0).
    // An estimate that decides between two forms, not a count: the check after the build still catches what it misses.
    // Only with one counter per minimizer (pop_bits == 32); the check after the build stays as the safety net.
    auto ext_by_crowding = [&](double* share) -> int32_t {
        unsigned long long* d_hist = nullptr;
        unsigned long long hist[256];
        LOAD_HIP(dev_alloc(&d_hist, sizeof hist));
        hipError_t e2 = hipMemsetAsync(d_hist, 0, sizeof hist, db->stream);
        if (e2 == hipSuccess) e2 = launch_pop_histogram(db->d_pop, db->pop_bits, d_hist, db->stream);
        if (e2 == hipSuccess) e2 = hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, db->stream);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(db->stream);
        (void)dev_free_idle(d_hist);
        LOAD_HIP(e2);
        const double W = (double)(db->info.kmer_length - 15), L = 0.65 * W;
        double all = 0, moved = 0;
        for (int c = 1; c < 256; ++c) {
            const double n = (double)hist[c];
            const double home = std::min((double)c, W * (1.0 - std::pow(1.0 - L / W, (double)c / L)));
            all += n * c;
            moved += n * ((double)c - home);
        }
        // ... and a k-mer that keeps its place among the k-mers of its own minimizer still meets those of OTHER minimizers that
        // hash to its line and cell: d = k-mers per cell of them on average, and of two runs that contest a line one keeps every
        // cell (cell_bid) -- half of the contested k-mers move (the 2.1 G-k-mer database of random k-mers at 0.33 per cell: 11 %
        // by its own minimizers, 49 % outside their cells in the plain table that round 4 built first)
        // (a slice's table has 1 / n_slices of the lines and holds 1 / n_slices of the k-mers the counters saw: the counters are of the
        // whole database)
        const double d = all / (double)std::max(1, db->n_slices) / (16.0 * (double)db->n_lines);
        const double own = all > 0 ? moved / all : 0.0;
        *share = own + (1.0 - own) * 0.5 * (1.0 - std::exp(-d));
        return KCFX_OK;
    };

    if (!two_phase) {
        if ((rc = stream_file(1, false))) return rc;
    } else {
        {
            // Loaders of one process take turns on the file's way into the device (round 6): page cache -> pinned memory is a memcpy
            // that the box's cores run at ~48 GB/s in all, so two loads that stream at the same time each take twice as long -- and
            // stay in step for good, streaming together and then building together. With the turn, B streams while A's build passes
            // run on the GPU (no host work), and the two alternate: profiles/r06_notes.md section 7.
            static std::mutex file_turn;
            std::unique_lock<std::mutex> turn(file_turn, std::defer_lock);
            if (!shared) turn.lock();                       // (slice loaders share one stream of the file instead: SuffixCache)
            if ((rc = stream_file(0, resident))) return rc;
        }
        tr.mark("file -> decode -> first pass (host)");
        if (cells) {
            if ((rc = table_ready(db))) {               // (the lines are first touched by the bids)
                cleanup();
                return rc;
            }
            tr.mark("wait for the table");
            constexpr double EXT_MOVED_SHARE = 0.15;
            if (early_ext) {
                double share = 0;
                if ((rc = ext_by_crowding(&share))) return rc;
                if (tr.on) fprintf(stderr, "[kcfx load] crowding counters: %.1f %% of the k-mers would sit outside their first-choice cells in a plain table\n", 100.0 * share);
                if (share > EXT_MOVED_SHARE) db->ext = 1;   // nothing but the counters has been built yet
                if (db->ext || !db->crowd_by_minimizer) {
                    // the bids want the counters per LOCUS (an extended table's own key): counted beside the others, or now
                    count_by_min = db->crowd_by_minimizer;
                    if (d_pop2) {
                        LOAD_HIP(hipStreamSynchronize(db->stream));
                        std::swap(db->d_pop, d_pop2);
                    } else {                                // (records resident, or no room for a second set: once more)
                        LOAD_HIP(hipMemsetAsync(db->d_pop, 0, cells_pop_bytes(db->pop_bits), db->stream));
                        if ((rc = later_pass(0))) return rc;
                    }
                    if (db->ext) tr.mark("extended table chosen by the crowding counters");
                }
                if (d_pop2) {
                    LOAD_HIP(hipStreamSynchronize(db->stream));
                    (void)dev_free_idle(d_pop2);
                    d_pop2 = nullptr;
                }
            }
            for (int phase = 1; phase <= 3; ++phase) {
                if (phase == 3) lost_list_alloc();
                if ((rc = later_pass(phase))) return rc;
            }
            if (db->ext_auto && !db->ext) {
                // how many k-mers sit outside their first-choice cell? (d_stats: [0] for the stash, [1] stored, [3] second / third lines)
                unsigned long long st4[4] = {0, 0, 0, 0};
                LOAD_HIP(hipMemcpyAsync(st4, db->d_stats, sizeof st4, hipMemcpyDeviceToHost, db->stream));
                LOAD_HIP(hipStreamSynchronize(db->stream));
                if (tr.on) fprintf(stderr, "[kcfx load] plain table: %.1f %% of the k-mers outside their first-choice cells\n", 100.0 * (double)(st4[0] + st4[3]) / (double)std::max<unsigned long long>(st4[1], 1));
                if ((double)(st4[0] + st4[3]) > EXT_MOVED_SHARE * (double)std::max<unsigned long long>(st4[1], 1)) {
                    db->ext = 1;                            // start over: lines, counters and statistics cleared, the four passes again
                    count_by_min = db->crowd_by_minimizer;
                    LOAD_HIP(hipMemsetAsync(db->d_lines, 0, (size_t)LINE_BYTES * db->n_lines, db->stream));
                    LOAD_HIP(hipMemsetAsync(db->d_pop, 0, cells_pop_bytes(db->pop_bits), db->stream));
                    LOAD_HIP(hipMemsetAsync(db->d_stats, 0, 4 * sizeof(unsigned long long), db->stream));
                    tr.mark("plain table: too many k-mers outside their cell");
                    if (lost.n) LOAD_HIP(hipMemsetAsync(lost.n, 0, sizeof(unsigned long long), db->stream));
                    for (int phase = 0; phase <= 3; ++phase)
                        if ((rc = later_pass(phase))) return rc;
                    tr.mark("rebuilt as an extended table");
                }
            }
            if ((rc = cells_stash_alloc(db))) {
                cleanup();
                return rc;
            }
            tr.mark("bids + settle + stash alloc");
            // the stash: from the list the settle pass wrote, when it holds every k-mer that pass counted; else one more pass over the records
            unsigned long long n_listed = ~0ull;
            if (lost.keys) {
                LOAD_HIP(hipMemcpyAsync(&n_listed, lost.n, sizeof n_listed, hipMemcpyDeviceToHost, db->stream));
                LOAD_HIP(hipStreamSynchronize(db->stream));
            }
            if (lost.keys && n_listed <= lost.cap && n_listed == (unsigned long long)db->info.stash_kmers) {
                if (tr.on) fprintf(stderr, "[kcfx load] the stash is filled from the %llu k-mers the settle pass wrote down\n", n_listed);
                LOAD_HIP(launch_cells_stash(db->d_lines, db->info.kmer_length, db->lbits, db->info.both_strands, lost.keys, lost.counts, (int64_t)n_listed,
                                            db->d_stash, db->stash_mask, db->stream, 0, nullptr, nullptr, db->ext, db->regions, db->slicing()));
            } else if ((rc = later_pass(4))) return rc;
        } else {
            LOAD_HIP(launch_table_mark_heavy(db->d_lines, db->n_lines, db->stream));
            if ((rc = later_pass(1))) return rc;
        }
    }
    LOAD_HIP(hipStreamSynchronize(db->stream));
    tr.mark("insert + sync");
#undef LOAD_HIP
    cleanup();
    tr.mark("free buffers");
    return KCFX_OK;
}

// One .kmc_suf decoded into device arrays (keys_out / counts_out hold h.total entries): pread -> pinned double buffer -> H2D ->
// decode_records_kernel, on `st`. The cohort loader's form of load_suffix_file's streaming (no table pass rides on it).
int32_t decode_file_to_device(const std::string& path, const KmcHeader& h, uint64_t* keys_out, uint32_t* counts_out, hipStream_t st) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(KCFX_ERR_IO, "Error reading suffix buffers from file " + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (int64_t)sb.st_size < 4 + h.total * h.record_size) {
        close(fd);
        return fail(KCFX_ERR_IO, "suffix file shorter than total_kmers * record_size: " + path);
    }
    char marker[4];
    if (pread(fd, marker, 4, 0) != 4 || memcmp(marker, "KMCS", 4) != 0) {
        close(fd);
        return fail(KCFX_ERR_FORMAT, "missing KMCS marker in " + path);
    }
    const int64_t CH = 4ll << 20;
    const int64_t n_chunk_max = std::min<int64_t>(CH, std::max<int64_t>(h.total, 1));
    uint8_t* h_buf[2] = {nullptr, nullptr};
    uint8_t* d_rec[2] = {nullptr, nullptr};
    uint64_t* d_lut = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int32_t rc = KCFX_OK;
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(st);
        for (int i = 0; i < 2; ++i) {
            if (h_buf[i]) (void)host_free(h_buf[i]);
            if (d_rec[i]) (void)dev_free_idle(d_rec[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
        }
        if (d_lut) (void)dev_free_idle(d_lut);
        close(fd);
    };
#define DEC_HIP(call)                                   \
    do {                                                \
        hipError_t e__ = (call);                        \
        if (e__ != hipSuccess) {                        \
            rc = hip_fail(e__, #call);                  \
            cleanup();                                  \
            return rc;                                  \
        }                                               \
    } while (0)
    for (int i = 0; i < 2; ++i) {
        DEC_HIP(host_alloc(&h_buf[i], (size_t)std::max<int64_t>(n_chunk_max * h.record_size, 16)));
        DEC_HIP(dev_alloc(&d_rec[i], (size_t)n_chunk_max * h.record_size + 16));
        DEC_HIP(hipEventCreateWithFlags(&done[i], hipEventDisableTiming));
    }
    DEC_HIP(dev_alloc(&d_lut, (size_t)std::max<int64_t>(h.lut_entries, 1) * 8));
    if (h.lut_entries) DEC_HIP(hipMemcpyAsync(d_lut, h.lut_bytes, (size_t)h.lut_entries * 8, hipMemcpyHostToDevice, st));
    int turn = 0;
    bool used[2] = {false, false};
    for (int64_t first = 0; first < h.total;) {
        const int64_t n = std::min(CH, h.total - first);
        const int64_t bytes = n * h.record_size;
        if (used[turn]) DEC_HIP(hipEventSynchronize(done[turn]));
        if (!parallel_pread(fd, h_buf[turn], bytes, 4 + first * h.record_size)) {
            cleanup();
            return fail(KCFX_ERR_IO, "short read on " + path);
        }
        DEC_HIP(hipMemcpyAsync(d_rec[turn], h_buf[turn], (size_t)bytes, hipMemcpyHostToDevice, st));
        DEC_HIP(hipEventRecord(done[turn], st));
        used[turn] = true;
        DEC_HIP(launch_decode_records(d_rec[turn], first, n, h.record_size, h.suffix_bytes, h.counter_size, h.p, d_lut, h.lut_entries,
                                      keys_out + first, counts_out + first, st));
        first += n;
        turn ^= 1;
    }
#undef DEC_HIP
    cleanup();
    return KCFX_OK;
}

}  // namespace

static void info_from_header(const KmcHeader& h, kcfx_db_info_t& info) {
    info.kmer_length = h.k;
    info.mode = h.mode;
    info.counter_size = h.counter_size;
    info.lut_prefix_length = h.p;
    info.signature_length = h.sig_len;
    info.min_count = h.min_count;
    info.max_count = h.max_count;
    info.both_strands = h.both;
    info.version = h.version;
    info.n_bins = h.n_bins;
    info.total_kmers = h.total;
    info.lut_suspect = h.lut_suspect ? 1 : 0;
}

// a handle under construction: freed on every exit path (error return or exception) unless released
struct DbHold {
    kcfx_db* p = nullptr;
    ~DbHold() { db_free(p); }
    kcfx_db* release() { kcfx_db* q = p; p = nullptr; return q; }
};
struct RefHold {
    kcfx_ref* p = nullptr;
    ~RefHold() { ref_free(p); }
    kcfx_ref* release() { kcfx_ref* q = p; p = nullptr; return q; }
};

// ------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------
#define KCFX_CATCH(fn) catch (...) { return kcfx::exception_to_status(fn); }
extern "C" {

const char* kcfx_last_error(void) { return g_error.c_str(); }
const char* kcfx_version(void) { return "kcfx 0.6.0 (gfx950)"; }

int32_t kcfx_device_count(int32_t* n) {
    if (!n) return fail(KCFX_ERR_ARG, "null argument");
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return KCFX_OK;
}

int32_t kcfx_db_open(const char* kmc_prefix, int32_t device, kcfx_db** out) { return kcfx_db_open_opts(kmc_prefix, device, nullptr, out); }

int32_t kcfx_db_open_opts(const char* kmc_prefix, int32_t device, const kcfx_db_opts* opts, kcfx_db** out) {
    return kcfx_db_open_slice(kmc_prefix, device, opts, 1, 0, out);
}

int32_t kcfx_table_slice_of(uint32_t g32, int32_t lbits, int32_t n_slices, int32_t slice_class) {
    if (n_slices < 1 || n_slices > KCFX_SLICES_MAX || lbits < CELL_LBITS_MIN || lbits > CELL_LBITS_MAX || slice_class < 0 || slice_class > 3) return -1;
    return (int32_t)cell_slice_of((uint32_t)slice_class, g32 >> (32 - lbits), g32, lbits, (uint32_t)n_slices);
}

int32_t kcfx_db_open_slice(const char* kmc_prefix, int32_t device, const kcfx_db_opts* opts, int32_t n_slices, int32_t slice, kcfx_db** out) {
    return kcfx::db_open_slice_shared(kmc_prefix, device, opts, n_slices, slice, nullptr, out);
}

}  // extern "C"

int32_t kcfx::db_open_slice_shared(const char* kmc_prefix, int32_t device, const kcfx_db_opts* opts, int32_t n_slices, int32_t slice, SuffixCache* shared, kcfx_db** out) try {
    struct Leave {                              // (the siblings' shared file stream stops counting on this loader when it is done, or has failed)
        SuffixCache* c;
        ~Leave() { if (c) c->leave(); }
    } leave{shared};
    if (!kmc_prefix || !out) return fail(KCFX_ERR_ARG, "null argument");
    if (opts && opts->expected_queries < 0) return fail(KCFX_ERR_ARG, "kcfx_db_opts: negative expected_queries");
    if (n_slices < 1 || n_slices > KCFX_SLICES_MAX || slice < 0 || slice >= n_slices) return fail(KCFX_ERR_ARG, "kcfx_db_open_slice: slice out of range (1 .. 64 slices)");
    *out = nullptr;
    const double t0 = now_s();
    KmcHeader h;
    LoadTrace tr;
    // The first open of a process on a device brings up the device's context and the loads' stream pair: 60 - 70 ms of the
    // runtime's, none of it this thread's to do -- a helper takes it while the .kmc_pre is read and checked (11 - 17 ms of host work
    // for the bench database; every later open finds the streams made and the helper returns at once)
    hipStream_t load_stream = nullptr;
    hipError_t stream_rc = hipSuccess;
    bool device_ok = true;
    std::thread bring_up([&] {
        DeviceGuard hg(device);
        device_ok = hg.ok;
        if (device_ok) stream_rc = db_stream(device, &load_stream, nullptr, true);
    });
    struct Joined {                             // (whatever the parse throws)
        std::thread& t;
        ~Joined() { if (t.joinable()) t.join(); }
    } joined{bring_up};
    std::string perr;
    const int32_t parse_rc = parse_prefix_file(std::string(kmc_prefix) + ".kmc_pre", h, perr);
    tr.mark("parse .kmc_pre");
    bring_up.join();
    struct LoadDone {                           // (the stream is another load's to take again when this one is over, however it ends)
        hipStream_t st;
        ~LoadDone() { if (st) db_stream_load_done(st); }
    } load_done{device_ok && stream_rc == hipSuccess ? load_stream : nullptr};
    if (parse_rc) return fail(parse_rc, perr);   // (a file that cannot be read is reported before a device that cannot be used)
    if (int32_t rc = check_device(device)) return rc;
    DeviceGuard g(device);
    if (!g.ok || !device_ok) return fail(KCFX_ERR_DEVICE, "hipSetDevice failed");
    if (stream_rc != hipSuccess) return hip_fail(stream_rc, "hipStreamCreate");
    DbHold hold;
    kcfx_db* db = hold.p = new kcfx_db();
    db->device = device;
    info_from_header(h, db->info);
    db->info.device = device;
    db->expected_queries = opts ? opts->expected_queries : 0;
    db->n_slices = n_slices;
    db->slice = slice;
    db->info.table_slices = n_slices;
    db->info.table_slice = slice;
    db->stream = load_stream;
    tr.mark("device + stream");
    // a slice holds 1 / n of the k-mers (a hash of the minimizer decides: + 2 % and a few thousand for the unevenness) -- the table is
    // sized for that many, every pass still walks the whole file
    const int64_t n_mine = n_slices > 1 ? std::min<int64_t>(h.total, h.total / n_slices + h.total / (50 * n_slices) + 4096) : h.total;
    if (n_slices > 1) {
        const int32_t kk = db->info.kmer_length;
        if (kk < CELL_K_MIN || kk > CELL_K_MAX || hook("KCFX_TABLE_FORMAT") || hook("KCFX_GENERIC_KERNEL") || hook("KCFX_MM_LEN") || table_mode() != MODE_MINIMIZER)
            return fail(KCFX_ERR_ARG, "table slices need a cell table (k = 19 .. 31, the default table format)");
    }
    int32_t rc = table_alloc(db, n_mine, h.counter_size, true, hook("KCFX_SYNC_ALLOC") == nullptr);
    tr.mark("table alloc (deferred)");
    if (!rc) rc = load_suffix_file(db, std::string(kmc_prefix) + ".kmc_suf", h, shared);
    // (the allocation thread may still be running when the streaming failed first -- an I/O error, or the decoded arrays lost the
    // race for the memory to the table: join it before its outcome is read)
    if (rc) db->alloc_go = true;
    if (rc && db->alloc_thread.joinable()) db->alloc_thread.join();
    // (a denser table helps when the table itself did not fit, or when the streaming's buffers lost the race for the memory to
    // it -- not when a later step failed: a stash that does not fit only grows with a denser table)
    if (rc == KCFX_ERR_DEVICE && db->fmt == FORMAT_CELLS && (db->regions > 1 || db->lbits - 1 >= cell_lbits_min(db->info.both_strands != 0)) &&
        (db->alloc_oom || (db->alloc_rc == KCFX_OK && !db->table_arrived) || (db->regions > 1 && db->table_arrived))) {
        // (... or a table of several regions -- grown up to 80 % of the free memory -- fitted and what comes after it did not: the
        // stash, the crowding counters, the streaming buffers. The power of two below it leaves them a third or more of the table.)
        // the policy's table did not fit next to what else lives on the device: denser cell tables, allocated synchronously
        // (a table of R regions that did not fit: the power of two below it -- (3 << L) -> 2^(L + 1), (5 | 7) << L -> 2^(L + 2) -- as one region)
        const int32_t cap = db->regions > 1 ? std::min(cells_lbits_cap(db->info.kmer_length), db->lbits + (db->regions == 3 ? 1 : 2)) : db->lbits - 1;
        table_release(db);
        db->lbits_cap = cap;
        rc = table_alloc(db, n_mine, h.counter_size, true, false);
        if (!rc) rc = load_suffix_file(db, std::string(kmc_prefix) + ".kmc_suf", h, shared);
        tr.mark("denser cell table after an out-of-memory");
    }
    if (rc == KCFX_ERR_DEVICE && db->fmt == FORMAT_CELLS && !(opts && opts->no_fallback) && n_slices == 1) {     // (a slice is a cell table or nothing)
        // the cell table or its stash did not fit (a database of many billions of k-mers next to another resident one):
        // the 12-slot line table stores a k-mer in 11-30 bytes instead of 100+, at ~0.55x the screening speed
        const std::string why = g_error;
        table_release(db);
        // (not an error and not reported on stderr -- this is a library behind a C / JNI ABI: kcfx_db_info's table_format says it)
        if (getenv("KCFX_VERBOSE")) fprintf(stderr, "kcfx: %s: cell table not built (%s); building a line table instead\n", kmc_prefix, why.c_str());
        rc = table_alloc(db, h.total, h.counter_size, false);
        if (!rc) rc = load_suffix_file(db, std::string(kmc_prefix) + ".kmc_suf", h, shared);
        tr.mark("rebuilt as a line table");
    }
    if (!rc) rc = table_finish(db);
    tr.mark("finalize");
    if (!rc && getenv("KCFX_VERBOSE"))
        fprintf(stderr, "kcfx: %s: table at %p, %u lines (%d region(s) of 2^%d), stash at %p\n", kmc_prefix, (void*)db->d_lines, db->n_lines, db->regions, db->lbits, (void*)db->d_stash);
    if (rc) return rc;
    db->info.load_seconds = now_s() - t0;
    *out = hold.release();
    return KCFX_OK;
} KCFX_CATCH("kcfx_db_open_slice")

extern "C" {

int32_t kcfx_table_plan(int64_t total_kmers, int32_t kmer_length, int32_t both_strands, int64_t free_bytes,
                        int64_t* lines, int32_t* regions) {
    return kcfx_table_plan_for(total_kmers, kmer_length, both_strands, free_bytes, 0, lines, regions);
}

int32_t kcfx_table_plan_for(int64_t total_kmers, int32_t kmer_length, int32_t both_strands, int64_t free_bytes, int64_t expected_queries,
                            int64_t* lines, int32_t* regions) try {
    if (total_kmers < 0 || free_bytes < 0 || expected_queries < 0) return fail(KCFX_ERR_ARG, "kcfx_table_plan: negative argument");
    if (kmer_length < 1 || kmer_length > 32) return fail(KCFX_ERR_FORMAT, "unsupported k-mer length (1..32)");
    int64_t n_lines = 0;
    int32_t R = 1;
    const bool both = both_strands != 0;
    if (kmer_length >= CELL_K_MIN && kmer_length <= CELL_K_MAX && total_kmers <= (int64_t)16 << cells_lbits_cap(kmer_length)) {
        const int32_t L0 = cells_lbits(total_kmers, kmer_length, both, expected_queries, LINE_BYTES, (size_t)free_bytes);
        int32_t Lr = L0;
        cells_regions(total_kmers, kmer_length, both, L0, (size_t)free_bytes, &R, &Lr);
        n_lines = (int64_t)R << Lr;
        if ((double)total_kmers > 16.0 * (double)n_lines) {      // (the device cannot hold a cell per k-mer)
            n_lines = 0;
            R = 1;
        }
    }
    if (lines) *lines = n_lines;
    if (regions) *regions = R;
    return KCFX_OK;
} KCFX_CATCH("kcfx_table_plan_for")

int32_t kcfx_db_build(const uint64_t* kmers, const uint32_t* counts, int64_t n, int32_t kmer_length,
                      int32_t counter_size, int32_t both_strands, int32_t device, kcfx_db** out) try {
    if (!out || n < 0 || (n > 0 && (!kmers || !counts))) return fail(KCFX_ERR_ARG, "null argument");
    *out = nullptr;
    if (kmer_length < 1 || kmer_length > 32) return fail(KCFX_ERR_FORMAT, "unsupported k-mer length (1..32)");
    if (counter_size < 0 || counter_size > 4) return fail(KCFX_ERR_FORMAT, "unsupported counter size");
    const double t0 = now_s();
    if (int32_t rc = check_device(device)) return rc;
    DeviceGuard g(device);
    if (!g.ok) return fail(KCFX_ERR_DEVICE, "hipSetDevice failed");
    DbHold hold;
    kcfx_db* db = hold.p = new kcfx_db();
    db->device = device;
    db->info.kmer_length = kmer_length;
    db->info.counter_size = counter_size;
    db->info.both_strands = both_strands ? 1 : 0;
    db->info.version = 0x200;
    db->info.total_kmers = n;
    db->info.device = device;
    struct Scratch {                                   // (freed on every exit path)
        uint64_t* keys = nullptr;
        uint32_t* counts = nullptr;
        ~Scratch() {
            if (keys) (void)dev_free(keys);
            if (counts) (void)dev_free(counts);
        }
    } scratch;
    uint64_t*& d_keys = scratch.keys;
    uint32_t*& d_counts = scratch.counts;
    int32_t rc = KCFX_OK;
    hipError_t e = db_stream(device, &db->stream, nullptr, true);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
    struct LoadDone {                           // (the stream is another load's to take again when this one is over, however it ends)
        hipStream_t st;
        ~LoadDone() { db_stream_load_done(st); }
    } load_done{db->stream};
    if (n > 0) {
        // the stored counter is counter_size bytes wide (KMC.java:395-401)
        std::vector<uint32_t> masked;
        const uint32_t* src = counts;
        if (counter_size < 4) {
            const uint32_t mask = counter_size ? ((1u << (8 * counter_size)) - 1u) : 0u;
            masked.resize((size_t)n);
            for (int64_t i = 0; i < n; ++i) masked[(size_t)i] = counts[i] & mask;
            src = masked.data();
        }
        if ((e = dev_alloc(&d_keys, (size_t)n * 8)) != hipSuccess || (e = dev_alloc(&d_counts, (size_t)n * 4)) != hipSuccess ||
            (e = hipMemcpyAsync(d_keys, kmers, (size_t)n * 8, hipMemcpyHostToDevice, db->stream)) != hipSuccess ||
            (e = hipMemcpyAsync(d_counts, src, (size_t)n * 4, hipMemcpyHostToDevice, db->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(db->stream)) != hipSuccess)            // (`masked` goes out of scope below)
            return hip_fail(e, "kcfx_db_build upload");
    }
    auto build = [&](bool allow_cells) -> int32_t {
        int32_t rc = table_alloc(db, n, counter_size, allow_cells);
        if (rc || n == 0) return rc;
        if (db->fmt == FORMAT_CELLS) {
            if ((e = launch_cells_count(db->d_pop, db->pop_bits, kmer_length, db->lbits, db->info.both_strands, d_keys, d_counts, n, db->stream, 0, db->ext, db->crowd_by_minimizer)) != hipSuccess ||
                (e = launch_cells_insert(db->d_lines, db->d_pop, db->pop_bits, 1, kmer_length, db->lbits, db->info.both_strands, d_keys, d_counts, n, db->stream, 0, nullptr, nullptr, db->ext, db->regions, 0, db->crowd_by_minimizer)) != hipSuccess ||
                (e = launch_cells_insert(db->d_lines, db->d_pop, db->pop_bits, 0, kmer_length, db->lbits, db->info.both_strands, d_keys, d_counts, n, db->stream, 0, nullptr, nullptr, db->ext, db->regions, 0, db->crowd_by_minimizer)) != hipSuccess ||
                (e = launch_cells_settle(db->d_lines, kmer_length, db->lbits, db->info.both_strands, d_keys, d_counts, n, db->d_stats, db->stream, 0, nullptr, nullptr, db->ext, db->regions)) != hipSuccess)
                return hip_fail(e, "kcfx_db_build insert");
            if ((rc = cells_stash_alloc(db))) return rc;
            if ((e = launch_cells_stash(db->d_lines, db->info.kmer_length, db->lbits, db->info.both_strands, d_keys, d_counts, n, db->d_stash, db->stash_mask, db->stream, 0, nullptr, nullptr, db->ext, db->regions)) != hipSuccess ||
                (e = hipStreamSynchronize(db->stream)) != hipSuccess)
                return hip_fail(e, "kcfx_db_build stash");
        } else {
            if ((e = launch_table_count(db->d_lines, db->view(), kmer_length, d_keys, n, db->stream)) != hipSuccess ||
                (db->mode == MODE_MINIMIZER && (e = launch_table_mark_heavy(db->d_lines, db->n_lines, db->stream)) != hipSuccess) ||
                (e = launch_table_insert(db->d_lines, db->view(), kmer_length, d_keys, d_counts, n, db->stream)) != hipSuccess ||
                (e = hipStreamSynchronize(db->stream)) != hipSuccess)
                return hip_fail(e, "kcfx_db_build insert");
        }
        return KCFX_OK;
    };
    rc = build(true);
    if (rc == KCFX_ERR_DEVICE && db->fmt == FORMAT_CELLS) {      // as in kcfx_db_open: the line table needs far less memory
        table_release(db);
        rc = build(false);
    }
    if (!rc && n == 0 && db->fmt == FORMAT_CELLS) rc = cells_stash_alloc(db);
    if (!rc) rc = table_finish(db);
    if (rc) return rc;
    db->info.load_seconds = now_s() - t0;
    *out = hold.release();
    return KCFX_OK;
} KCFX_CATCH("kcfx_db_build")

int32_t kcfx_db_header(const char* kmc_prefix, kcfx_db_info_t* info) try {
    if (!kmc_prefix || !info) return fail(KCFX_ERR_ARG, "null argument");
    KmcHeader h;
    {
        std::string perr;
        if (int32_t rc = parse_prefix_file(std::string(kmc_prefix) + ".kmc_pre", h, perr)) return fail(rc, perr);
    }
    memset(info, 0, sizeof *info);
    info_from_header(h, *info);
    info->device = -1;
    return KCFX_OK;
} KCFX_CATCH("kcfx_db_header")

int32_t kcfx_db_info(const kcfx_db* db, kcfx_db_info_t* info) {
    if (!db || !info) return fail(KCFX_ERR_ARG, "null argument");
    *info = db->info;
    return KCFX_OK;
}

int32_t kcfx_db_lookup(kcfx_db* db, const uint64_t* kmers, int64_t n, int32_t* counts_out) try {
    if (!db || n < 0 || (n > 0 && (!kmers || !counts_out))) return fail(KCFX_ERR_ARG, "null argument");
    if (n == 0) return KCFX_OK;
    DeviceGuard g(db->device);
    uint64_t* d_k = nullptr;
    int32_t* d_c = nullptr;
    hipError_t e;
    int32_t rc = KCFX_OK;
    if ((e = dev_alloc(&d_k, (size_t)n * 8)) != hipSuccess || (e = dev_alloc(&d_c, (size_t)n * 4)) != hipSuccess ||
        (e = hipMemcpyAsync(d_k, kmers, (size_t)n * 8, hipMemcpyHostToDevice, db->stream)) != hipSuccess ||
        (e = launch_lookup(db->view(), db->info.kmer_length, d_k, n, d_c, db->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(counts_out, d_c, (size_t)n * 4, hipMemcpyDeviceToHost, db->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(db->stream)) != hipSuccess)
        rc = hip_fail(e, "kcfx_db_lookup");
    if (d_k) (void)dev_free(d_k);
    if (d_c) (void)dev_free(d_c);
    return rc;
} KCFX_CATCH("kcfx_db_lookup")

void kcfx_db_close(kcfx_db* db) { db_free(db); }

}  // extern "C"
// kcfx_db_close for a caller that has already waited for every launch that used the table (no device-wide wait: see db_free)
void kcfx::db_close_idle(kcfx_db* db) { db_free(db, true); }
extern "C" {

int32_t kcfx_release_cached(int32_t device) try {
    kcfx::pool_release(device);
    kcfx::host_pool_release();
    return KCFX_OK;
} KCFX_CATCH("kcfx_release_cached")

double kcfx_set_cache_fraction(double fraction) {
    const double prev = kcfx::pool_set_fraction(fraction);
    if (fraction >= 0.0 && fraction < prev) kcfx::pool_trim();
    return prev;
}

int32_t kcfx_ref_create(int32_t device, int32_t kmer_length, const uint8_t* text, int64_t text_len,
                        const int64_t* win_begin, const int32_t* win_len, int32_t n_windows, kcfx_ref** out) {
    return kcfx_ref_create_opts(device, kmer_length, text, text_len, win_begin, win_len, n_windows, nullptr, out);
}

int32_t kcfx_ref_create_opts(int32_t device, int32_t kmer_length, const uint8_t* text, int64_t text_len,
                             const int64_t* win_begin, const int32_t* win_len, int32_t n_windows, const kcfx_ref_opts* opts,
                             kcfx_ref** out) try {
    if (opts && opts->expected_screens < 0) return fail(KCFX_ERR_ARG, "kcfx_ref_opts: negative expected_screens");
    if (!out || text_len < 0 || n_windows < 0 || (text_len > 0 && !text) || (n_windows > 0 && (!win_begin || !win_len)))
        return fail(KCFX_ERR_ARG, "null argument");
    *out = nullptr;
    if (kmer_length < 1 || kmer_length > 32) return fail(KCFX_ERR_FORMAT, "unsupported k-mer length (1..32)");
    const int32_t k = kmer_length;
    // KCFX_REF_PLAN=0 in the environment keeps the reference as text only (A/B and test hook, read here, never per launch):
    // cell-table launches then plan every tile themselves and take tiles of at most TEXT_TILE_K k-mers (k31_plan_tile)
    // A reference that ONE database will be screened against (kcfx_ref_opts.expected_screens == 1: one sample per device in the
    // sharded call, the one-shot kcfx_screen) is kept as text only too: the one launch that plans its tiles itself is faster than
    // plan + screen as two (0.095 against 0.027 + 0.078 ms on the bench workload) and needs no plan buffers.
    const char* plan_env = hook("KCFX_REF_PLAN");
    const bool cell_k = k >= CELL_K_MIN && k <= CELL_K_MAX;
    bool planned = cell_k && text_len > 0 && !(opts && opts->expected_screens == 1);
    if (plan_env) planned = planned && plan_env[0] != '0';
    if (plan_env && plan_env[0] == '1') planned = cell_k && text_len > 0;
    const int64_t tile_k = (cell_k && !planned) ? TEXT_TILE_K : TILE_K;
    LoadTrace trace("ref", "KCFX_REF_TRACE");               // KCFX_REF_TRACE=1: phase timings on stderr
    // tile plan: every window is cut into near-equal tiles of <= tile_k k-mer start positions (multiples of 64)
    std::vector<TileDesc> tiles;
    std::vector<int32_t> win_tile0((size_t)n_windows + 1, 0);
    int64_t n_slots = 0;
    for (int32_t w = 0; w < n_windows; ++w) {
        if (win_begin[w] < 0 || win_len[w] < 0 || win_begin[w] + (int64_t)win_len[w] > text_len)
            return fail(KCFX_ERR_ARG, "window " + std::to_string(w) + " lies outside the text");   // FastaIndex.java:131-134
        win_tile0[(size_t)w] = (int32_t)tiles.size();
        const int64_t nk = (int64_t)win_len[w] - k + 1;
        if (nk <= 0) continue;
        n_slots += nk;
        const int64_t nt = (nk + tile_k - 1) / tile_k;
        int64_t per = (nk + nt - 1) / nt;
        if ((per + 63) / 64 * 64 <= tile_k) per = (per + 63) / 64 * 64;       // whole words where that fits (only a tile's last word may be ragged)
        for (int64_t s = 0; s < nk; s += per) {
            TileDesc t;
            t.text_off = win_begin[w] + s;
            t.n_kmers = (int32_t)std::min<int64_t>(per, nk - s);
            t.window = w;
            tiles.push_back(t);
        }
        if (tiles.size() > 0x7FFFFFF0u) return fail(KCFX_ERR_ARG, "too many tiles");
    }
    win_tile0[(size_t)n_windows] = (int32_t)tiles.size();
    trace.mark("tile list (host)");

    if (int32_t rc = check_device(device)) return rc;
    DeviceGuard g(device);
    if (!g.ok) return fail(KCFX_ERR_DEVICE, "hipSetDevice failed");
    RefHold hold;
    kcfx_ref* r = hold.p = new kcfx_ref();
    r->device = device;
    r->k = k;
    r->n_windows = n_windows;
    r->n_tiles = (int32_t)tiles.size();
    r->n_slots = n_slots;
    r->text_len = text_len;
    {   // (FNV-1a over the window list)
        uint64_t hsh = 1469598103934665603ull;
        for (int32_t w = 0; w < n_windows; ++w) {
            hsh = (hsh ^ (uint64_t)win_begin[w]) * 1099511628211ull;
            hsh = (hsh ^ (uint64_t)(uint32_t)win_len[w]) * 1099511628211ull;
        }
        r->win_hash = hsh;
    }
    // One block: the text (+ 128 zero bytes: the pack phase reads whole 32-byte chunks) and, for cell tables (k = 19 .. 31), the
    // reference plan behind it -- anchors, then codes, each 256-byte aligned.
    auto up256 = [](size_t n) { return (n + 255) & ~(size_t)255; };
    const size_t text_alloc = up256((size_t)text_len + 128);
    const size_t ab = planned ? up256(ref_plan_anchor_bytes(text_len)) : 0, cb = planned ? up256(ref_plan_codes_bytes(text_len)) : 0;
    hipError_t e;
    if ((e = stream_acquire(device, &r->stream)) != hipSuccess ||
        (e = dev_alloc(&r->d_text, text_alloc + ab + cb)) != hipSuccess ||
        (e = hipMemsetAsync(r->d_text + text_len, 0, text_alloc - (size_t)text_len + ab + cb, r->stream)) != hipSuccess ||
        (text_len > 0 && (e = hipMemcpyAsync(r->d_text, text, (size_t)text_len, hipMemcpyHostToDevice, r->stream)) != hipSuccess) ||
        (e = dev_alloc(&r->d_tiles, std::max<size_t>(tiles.size(), 1) * sizeof(TileDesc))) != hipSuccess ||
        (e = dev_alloc(&r->d_win_tile0, win_tile0.size() * 4)) != hipSuccess ||
        (e = dev_alloc(&r->d_bits, std::max<size_t>(tiles.size(), 1) * TILE_BITS_U64 * 8)) != hipSuccess ||
        (e = dev_alloc(&r->d_csum, std::max<size_t>(tiles.size(), 1) * 8)) != hipSuccess ||
        (e = dev_alloc(&r->d_res, std::max<size_t>((size_t)n_windows, 1) * sizeof(kcfx_result))) != hipSuccess ||
        (!tiles.empty() && (e = hipMemcpyAsync(r->d_tiles, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, r->stream)) != hipSuccess) ||
        (e = hipMemcpyAsync(r->d_win_tile0, win_tile0.data(), win_tile0.size() * 4, hipMemcpyHostToDevice, r->stream)) != hipSuccess)
        return hip_fail(e, "kcfx_ref_create");
    trace.mark("allocations + copies queued");
    if (planned) {                                          // on the same stream, behind the text upload
        r->d_anchors = r->d_text + text_alloc;
        r->d_codes = reinterpret_cast<uint32_t*>(r->d_text + text_alloc + ab);
        if ((e = launch_ref_plan(r->d_text, text_len, k, r->d_codes, r->d_anchors, r->stream)) != hipSuccess)
            return hip_fail(e, "kcfx_ref_create (reference plan)");
    }
    if ((e = hipStreamSynchronize(r->stream)) != hipSuccess) return hip_fail(e, "kcfx_ref_create");
    trace.mark(planned ? "uploads + plan done" : "uploads done");
    if (getenv("KCFX_VERBOSE"))
        fprintf(stderr, "kcfx: reference: text at %p (%zu bytes with the plan), tiles at %p, bitmaps at %p, sums at %p, results at %p\n", (void*)r->d_text,
                text_alloc + ab + cb, (void*)r->d_tiles, (void*)r->d_bits, (void*)r->d_csum, (void*)r->d_res);
    *out = hold.release();
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_create_opts")

// The sample-independent half of a pass (SURVEY's K1: 2-bit pack, validity, minimizer choice) over the resident text again, on
// the caller's stream: what a job that screens ONE sample per reference pays per pass. Idempotent (same text, same plan).
int32_t kcfx_ref_plan_async(kcfx_ref* ref, void* hip_stream) try {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    if (!ref->d_codes) return fail(KCFX_ERR_ARG, "the reference has no plan (text-only: k outside 19 .. 31, or KCFX_REF_PLAN=0)");
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    DeviceGuard g(ref->device);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ref->prof) {
        if (ref->plan_events_used == ref->plan_events.size()) {
            hipEvent_t a, b;
            KCFX_HIP(hipEventCreate(&a));
            KCFX_HIP(hipEventCreate(&b));
            ref->plan_events.emplace_back(a, b);
        }
        ev0 = ref->plan_events[ref->plan_events_used].first;
        ev1 = ref->plan_events[ref->plan_events_used].second;
        ref->plan_events_used++;
    }
    KCFX_HIP(launch_ref_plan(ref->d_text, ref->text_len, ref->k, ref->d_codes, ref->d_anchors, st, ev0, ev1));
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_plan_async")

int32_t kcfx_ref_is_planned(const kcfx_ref* ref) { return ref && ref->d_codes ? 1 : 0; }

int32_t kcfx_ref_n_windows(const kcfx_ref* ref) { return ref ? ref->n_windows : 0; }
int64_t kcfx_ref_n_kmer_slots(const kcfx_ref* ref) { return ref ? ref->n_slots : 0; }
void kcfx_ref_destroy(kcfx_ref* ref) { ref_free(ref); }

// the screening launch of one database against a resident reference (+ reduce_windows unless `partial`: the table is one slice of
// its database and the tiles' bitmaps are merged with the other slices' first)
static int32_t screen_ref_launch(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, void* out_device, void* hip_stream, bool partial) {
    if (!db || !ref || (!out_device && !partial)) return fail(KCFX_ERR_ARG, "null argument");
    if (db->device != ref->device) return fail(KCFX_ERR_ARG, "database and reference live on different devices");
    if (db->info.kmer_length != ref->k) return fail(KCFX_ERR_ARG, "reference was planned for a different k-mer length");
    if (min_kmer_count < 1) return fail(KCFX_ERR_ARG, "Minimum kmer count should be at least 1");   // GetVariants.java:383-385
    if (db->n_slices > 1 && !partial)
        return fail(KCFX_ERR_ARG, "the handle holds one slice of its database (kcfx_db_open_slice): use kcfx_screen_ref_partial_async + kcfx_ref_merge_* + kcfx_ref_finish_async");
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    DeviceGuard g(db->device);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ref->prof && ref->n_tiles > 0) {                   // (an event that is never recorded cannot be read back)
        if (ref->events_used == ref->events.size()) {
            hipEvent_t a, b;
            KCFX_HIP(hipEventCreate(&a));
            KCFX_HIP(hipEventCreate(&b));
            ref->events.emplace_back(a, b);
        }
        ev0 = ref->events[ref->events_used].first;
        ev1 = ref->events[ref->events_used].second;
        ref->events_used++;
    }
    KCFX_HIP(launch_screen_tiles(ref->d_text, ref->plan(), ref->d_tiles, ref->n_tiles, db->view(), ref->k, db->info.both_strands,
                                 min_kmer_count, ref->d_bits, ref->d_csum, st, ev0, ev1));
    if (partial) {                                         // (kcfx_screen_ref_partial_async: the bitmaps stay where they are)
        // ... and a merge out of this reference (kcfx_ref_merge_from) is ordered behind this launch whatever stream it ran on
        if (!ref->partial_done) KCFX_HIP(hipEventCreateWithFlags(&ref->partial_done, hipEventDisableTiming));
        KCFX_HIP(hipEventRecord(ref->partial_done, st));
        return KCFX_OK;
    }
    KCFX_HIP(launch_reduce_windows(ref->d_tiles, ref->n_tiles, 1, ref->d_bits, ref->d_csum, ref->d_win_tile0, ref->n_windows,
                                   ref->k, reinterpret_cast<kcfx_result*>(out_device), st));
    return KCFX_OK;
}

int32_t kcfx_screen_ref_async(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, void* out_device, void* hip_stream) try {
    return screen_ref_launch(db, ref, min_kmer_count, out_device, hip_stream, false);
} KCFX_CATCH("kcfx_screen_ref_async")

// ---- a table partitioned over devices (kcfx_db_open_slice): screen against one slice, merge, reduce once ---------------------
// The screening kernel alone: the tiles' bitmaps (validity + the hits THIS slice found) and count sums stay in the reference's
// scratch. Works for a whole table too (one slice of one).
int32_t kcfx_screen_ref_partial_async(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, void* hip_stream) try {
    if (!db || !ref) return fail(KCFX_ERR_ARG, "null argument");
    if (db->fmt != FORMAT_CELLS) return fail(KCFX_ERR_ARG, "a partial screen needs a cell table (k = 19 .. 31)");
    return screen_ref_launch(db, ref, min_kmer_count, nullptr, hip_stream, true);
} KCFX_CATCH("kcfx_screen_ref_partial_async")

int64_t kcfx_ref_partial_bytes(const kcfx_ref* ref, int64_t* bitmap_bytes, int64_t* sum_bytes) {
    if (!ref) return 0;
    const int64_t b = (int64_t)std::max(ref->n_tiles, 1) * TILE_BITS_U64 * 8, c = (int64_t)std::max(ref->n_tiles, 1) * 8;
    if (bitmap_bytes) *bitmap_bytes = b;
    if (sum_bytes) *sum_bytes = c;
    return b + c;
}

// the reference's partial results copied into caller-owned DEVICE buffers (e.g. tensors handed to an RCCL gather)
int32_t kcfx_ref_export_partial_async(kcfx_ref* ref, void* bitmaps_device, void* sums_device, void* hip_stream) try {
    if (!ref || !bitmaps_device || !sums_device) return fail(KCFX_ERR_ARG, "null argument");
    int64_t b = 0, c = 0;
    kcfx_ref_partial_bytes(ref, &b, &c);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    DeviceGuard g(ref->device);
    KCFX_HIP(hipMemcpyAsync(bitmaps_device, ref->d_bits, (size_t)b, hipMemcpyDeviceToDevice, st));
    KCFX_HIP(hipMemcpyAsync(sums_device, ref->d_csum, (size_t)c, hipMemcpyDeviceToDevice, st));
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_export_partial_async")

// caller-owned device buffers INTO the reference's scratch (the inverse of the export): what a rank that owns a range of the windows
// starts its merge from, when the slices' results meet range by range instead of on one root (shard.py: reduce_scatter_table_partials)
int32_t kcfx_ref_import_partial_async(kcfx_ref* ref, const void* bitmaps_device, const void* sums_device, void* hip_stream) try {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    if (ref->n_tiles == 0) return KCFX_OK;                 // (a rank that owns no window: nothing to bring in, whatever the pointers are)
    if (!bitmaps_device || !sums_device) return fail(KCFX_ERR_ARG, "null argument");
    int64_t b = 0, c = 0;
    kcfx_ref_partial_bytes(ref, &b, &c);
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    DeviceGuard g(ref->device);
    KCFX_HIP(hipMemcpyAsync(ref->d_bits, bitmaps_device, (size_t)b, hipMemcpyDeviceToDevice, st));
    KCFX_HIP(hipMemcpyAsync(ref->d_csum, sums_device, (size_t)c, hipMemcpyDeviceToDevice, st));
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_import_partial_async")

// the tiles of a window range: windows [w_lo, w_hi) of this reference are its tiles [*t_lo, *t_hi) -- the units a rank exchanges
int32_t kcfx_ref_tile_range(const kcfx_ref* ref, int32_t w_lo, int32_t w_hi, int32_t* t_lo, int32_t* t_hi) try {
    if (!ref || !t_lo || !t_hi || w_lo < 0 || w_hi < w_lo || w_hi > ref->n_windows) return fail(KCFX_ERR_ARG, "kcfx_ref_tile_range: bad window range");
    if (ref->n_windows == 0) {
        *t_lo = *t_hi = 0;
        return KCFX_OK;
    }
    int32_t v[2] = {0, 0};
    DeviceGuard g(ref->device);
    KCFX_HIP(hipMemcpy(&v[0], ref->d_win_tile0 + w_lo, sizeof(int32_t), hipMemcpyDeviceToHost));
    KCFX_HIP(hipMemcpy(&v[1], ref->d_win_tile0 + w_hi, sizeof(int32_t), hipMemcpyDeviceToHost));
    *t_lo = v[0];
    *t_hi = v[1];
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_tile_range")

// another slice's partial results (device buffers on the reference's device, same tile list) into this reference's: hits OR-ed,
// count sums added; the validity words are the same on every slice
int32_t kcfx_ref_merge_partial_async(kcfx_ref* ref, const void* bitmaps_device, const void* sums_device, void* hip_stream) try {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    if (ref->n_tiles == 0) return KCFX_OK;                 // (no tiles: nothing to merge)
    if (!bitmaps_device || !sums_device) return fail(KCFX_ERR_ARG, "null argument");
    DeviceGuard g(ref->device);
    KCFX_HIP(launch_merge_partial(ref->d_bits, ref->d_csum, reinterpret_cast<const uint64_t*>(bitmaps_device),
                                  reinterpret_cast<const int64_t*>(sums_device), ref->n_tiles, reinterpret_cast<hipStream_t>(hip_stream)));
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_merge_partial_async")

// the same from another reference handle of the same text and windows, on this device or on another one of the process (peer copy
// into a scratch buffer first); synchronous: returns when dst holds the merged results
int32_t kcfx_ref_merge_from(kcfx_ref* dst, kcfx_ref* src) try {
    if (!dst || !src) return fail(KCFX_ERR_ARG, "null argument");
    if (dst->n_tiles != src->n_tiles || dst->n_windows != src->n_windows || dst->k != src->k || dst->text_len != src->text_len || dst->win_hash != src->win_hash)
        return fail(KCFX_ERR_ARG, "kcfx_ref_merge_from: the two references do not describe the same windows");
    if (dst == src || dst->n_tiles == 0) return KCFX_OK;
    int64_t b = 0, c = 0;
    kcfx_ref_partial_bytes(dst, &b, &c);
    {
        // src's partial screen -- on its own stream or on a caller's (kcfx_screen_ref_partial_async records partial_done there) -- first
        DeviceGuard gs(src->device);
        if (src->partial_done) KCFX_HIP(hipEventSynchronize(src->partial_done));
        KCFX_HIP(hipStreamSynchronize(src->stream));
    }
    DeviceGuard g(dst->device);
    if (dst->partial_done) KCFX_HIP(hipStreamWaitEvent(dst->stream, dst->partial_done, 0));    // (dst's own partial screen, wherever it was launched)
    if (dst->device == src->device) {
        KCFX_HIP(launch_merge_partial(dst->d_bits, dst->d_csum, src->d_bits, src->d_csum, dst->n_tiles, dst->stream));
        KCFX_HIP(hipStreamSynchronize(dst->stream));
        return KCFX_OK;
    }
    uint8_t* tmp = nullptr;
    KCFX_HIP(dev_alloc(&tmp, (size_t)(b + c)));
    hipError_t e = hipMemcpyPeerAsync(tmp, dst->device, src->d_bits, src->device, (size_t)b, dst->stream);
    if (e == hipSuccess) e = hipMemcpyPeerAsync(tmp + b, dst->device, src->d_csum, src->device, (size_t)c, dst->stream);
    if (e == hipSuccess) e = launch_merge_partial(dst->d_bits, dst->d_csum, reinterpret_cast<const uint64_t*>(tmp), reinterpret_cast<const int64_t*>(tmp + b), dst->n_tiles, dst->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(dst->stream);
    (void)dev_free_idle(tmp);
    if (e != hipSuccess) return hip_fail(e, "kcfx_ref_merge_from");
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_merge_from")

// reduce_windows over what the reference's scratch holds now (one slice's results or the merged ones): 40 bytes per window to out_device
int32_t kcfx_ref_finish_async(kcfx_ref* ref, void* out_device, void* hip_stream) try {
    if (!ref || !out_device) return fail(KCFX_ERR_ARG, "null argument");
    DeviceGuard g(ref->device);
    KCFX_HIP(launch_reduce_windows(ref->d_tiles, ref->n_tiles, 1, ref->d_bits, ref->d_csum, ref->d_win_tile0, ref->n_windows,
                                   ref->k, reinterpret_cast<kcfx_result*>(out_device), reinterpret_cast<hipStream_t>(hip_stream)));
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_finish_async")

// synchronous forms on the handles' own streams (what kcfx_screen_sharded's workers call)
int32_t kcfx_screen_ref_partial(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count) try {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    if (int32_t rc = kcfx_screen_ref_partial_async(db, ref, min_kmer_count, ref->stream)) return rc;
    DeviceGuard g(ref->device);
    KCFX_HIP(hipStreamSynchronize(ref->stream));
    return KCFX_OK;
} KCFX_CATCH("kcfx_screen_ref_partial")

int32_t kcfx_ref_finish(kcfx_ref* ref, kcfx_result* out) try {
    if (!ref || (!out && ref->n_windows > 0)) return fail(KCFX_ERR_ARG, "null argument");
    if (ref->n_windows == 0) return KCFX_OK;
    if (int32_t rc = kcfx_ref_finish_async(ref, ref->d_res, ref->stream)) return rc;
    DeviceGuard g(ref->device);
    KCFX_HIP(hipMemcpyAsync(out, ref->d_res, (size_t)ref->n_windows * sizeof(kcfx_result), hipMemcpyDeviceToHost, ref->stream));
    KCFX_HIP(hipStreamSynchronize(ref->stream));
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_finish")

// SURVEY f-4: S resident databases, one pass over the reference. Databases the multi-sample kernel cannot take
// (k != 31, key-hash tables) go through the single-sample path into their slice of the output.
int32_t kcfx_screen_ref_multi_async(kcfx_db* const* dbs, int32_t n_db, kcfx_ref* ref, int32_t min_kmer_count,
                                    void* out_device, void* hip_stream) try {
    if (!ref || n_db < 0 || (n_db > 0 && (!dbs || !out_device))) return fail(KCFX_ERR_ARG, "null argument");
    if (min_kmer_count < 1) return fail(KCFX_ERR_ARG, "Minimum kmer count should be at least 1");   // GetVariants.java:383-385
    for (int32_t i = 0; i < n_db; ++i) {
        if (!dbs[i]) return fail(KCFX_ERR_ARG, "null database handle");
        if (dbs[i]->device != ref->device) return fail(KCFX_ERR_ARG, "database and reference live on different devices");
        if (dbs[i]->info.kmer_length != ref->k) return fail(KCFX_ERR_ARG, "reference was planned for a different k-mer length");
    }
    if (n_db == 0 || ref->n_windows == 0) return KCFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    kcfx_result* out = reinterpret_cast<kcfx_result*>(out_device);
    DeviceGuard g(ref->device);
    // scratch of the multi-sample kernel: only if some database can take it, sized for the largest batch this call forms
    // (512 B per tile and sample: 29 GB for eight samples of a 14.5 Gb reference -- not to be allocated on spec)
    // Cell tables (k = 19 .. 31: the BASELINE configuration) take one screening launch per sample -- their probe is one load per k-mer
    // already -- but their window reduces are batched (round 6): the launches of up to eight consecutive samples write their tiles'
    // bitmaps side by side into the multi-sample scratch and ONE reduce_windows launch (grid.y = samples) turns them into records, where
    // every sample used to pay a reduce and the gap in front of it (7 + ~4 us per sample beside a 80 us pass of a 30 Mb reference). The
    // scratch is bounded at 2 GiB: a 14.5 Gb reference (3.7 GB of bitmaps per sample) keeps the one-by-one form.
    int32_t cell_batch = 1;
    {
        int32_t eligible = 0, cells = 0;
        for (int32_t i = 0; i < n_db; ++i) {
            eligible += screen_multi_supported(dbs[i]->view(), ref->k) ? 1 : 0;
            cells += (dbs[i]->fmt == FORMAT_CELLS && dbs[i]->n_slices == 1) ? 1 : 0;
        }
        const double per_sample = (double)std::max(ref->n_tiles, 1) * (TILE_BITS_U64 * 8 + 8);
        if (cells >= 2) cell_batch = (int32_t)std::max(1.0, std::min({(double)SCREEN_MULTI_MAX, (double)cells, std::floor((double)(2ull << 30) / per_sample)}));
        if (int32_t rc = kcfx_ref_reserve_multi(ref, std::max(eligible, cell_batch >= 2 ? cell_batch : 0))) return rc;
    }
    std::vector<char> done(n_db, 0);
    for (int32_t first = 0; first < n_db; ++first) {
        if (done[first]) continue;
        kcfx_db* d0 = dbs[first];
        if (cell_batch >= 2 && d0->fmt == FORMAT_CELLS && d0->n_slices == 1) {
            // the run of consecutive cell tables that starts here, up to cell_batch of them
            int32_t nb = 0;
            while (first + nb < n_db && nb < cell_batch && nb < ref->multi_cap && !done[first + nb] && dbs[first + nb]->fmt == FORMAT_CELLS && dbs[first + nb]->n_slices == 1) ++nb;
            for (int32_t b = 0; b < nb; ++b) {
                kcfx_db* d = dbs[first + b];
                hipEvent_t ev0 = nullptr, ev1 = nullptr;
                if (ref->prof && ref->n_tiles > 0) {
                    if (ref->events_used == ref->events.size()) {
                        hipEvent_t ea, eb;
                        KCFX_HIP(hipEventCreate(&ea));
                        KCFX_HIP(hipEventCreate(&eb));
                        ref->events.emplace_back(ea, eb);
                    }
                    ev0 = ref->events[ref->events_used].first;
                    ev1 = ref->events[ref->events_used].second;
                    ref->events_used++;
                }
                KCFX_HIP(launch_screen_tiles(ref->d_text, ref->plan(), ref->d_tiles, ref->n_tiles, d->view(), ref->k, d->info.both_strands, min_kmer_count,
                                             ref->d_mbits + (size_t)b * ref->n_tiles * TILE_BITS_U64, ref->d_mcsum + (size_t)b * ref->n_tiles, st, ev0, ev1));
                done[first + b] = 1;
            }
            KCFX_HIP(launch_reduce_windows(ref->d_tiles, ref->n_tiles, nb, ref->d_mbits, ref->d_mcsum, ref->d_win_tile0, ref->n_windows, ref->k,
                                           out + (size_t)first * ref->n_windows, st));
            continue;
        }
        if (!screen_multi_supported(d0->view(), ref->k)) {
            if (int32_t rc = kcfx_screen_ref_async(d0, ref, min_kmer_count, out + (size_t)first * ref->n_windows, hip_stream))
                return rc;
            done[first] = 1;
            continue;
        }
        // batch: the next databases of the same strandedness the kernel can take, in input order
        int32_t idx[SCREEN_MULTI_MAX];
        TableView tvs[SCREEN_MULTI_MAX];
        int32_t nb = 0;
        for (int32_t j = first; j < n_db && nb < ref->multi_cap; ++j) {
            if (done[j] || !screen_multi_supported(dbs[j]->view(), ref->k)) continue;
            if (dbs[j]->info.both_strands != d0->info.both_strands) continue;
            idx[nb] = j;
            tvs[nb++] = dbs[j]->view();
            done[j] = 1;
        }
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (ref->prof && ref->n_tiles > 0) {
            if (ref->events_used == ref->events.size()) {
                hipEvent_t a, b;
                KCFX_HIP(hipEventCreate(&a));
                KCFX_HIP(hipEventCreate(&b));
                ref->events.emplace_back(a, b);
            }
            ev0 = ref->events[ref->events_used].first;
            ev1 = ref->events[ref->events_used].second;
            ref->events_used++;
        }
        KCFX_HIP(launch_screen_multi(ref->d_text, ref->d_tiles, ref->n_tiles, tvs, nb, d0->info.both_strands, min_kmer_count,
                                     ref->d_mbits, ref->d_mcsum, st, ev0, ev1));
        // consecutive output slices take one launch; scattered ones (mixed batches) one each
        for (int32_t b = 0; b < nb;) {
            int32_t e = b + 1;
            while (e < nb && idx[e] == idx[e - 1] + 1) ++e;
            KCFX_HIP(launch_reduce_windows(ref->d_tiles, ref->n_tiles, e - b, ref->d_mbits + (size_t)b * ref->n_tiles * TILE_BITS_U64,
                                           ref->d_mcsum + (size_t)b * ref->n_tiles, ref->d_win_tile0, ref->n_windows, ref->k,
                                           out + (size_t)idx[b] * ref->n_windows, st));
            b = e;
        }
    }
    return KCFX_OK;
} KCFX_CATCH("kcfx_screen_ref_multi_async")

int32_t kcfx_ref_reserve_multi(kcfx_ref* ref, int32_t n_samples) try {
    if (!ref || n_samples < 0) return fail(KCFX_ERR_ARG, "null argument");
    const int32_t want = std::min<int32_t>(n_samples, SCREEN_MULTI_MAX);
    if (want <= ref->multi_cap) return KCFX_OK;
    DeviceGuard g(ref->device);
    if (ref->d_mbits) (void)dev_free(ref->d_mbits);
    if (ref->d_mcsum) (void)dev_free(ref->d_mcsum);
    ref->d_mbits = nullptr;
    ref->d_mcsum = nullptr;
    ref->multi_cap = 0;
    const size_t nt = std::max<size_t>(ref->n_tiles, 1);
    KCFX_HIP(dev_alloc(&ref->d_mbits, nt * (size_t)want * TILE_BITS_U64 * 8));
    KCFX_HIP(dev_alloc(&ref->d_mcsum, nt * (size_t)want * sizeof(int64_t)));
    ref->multi_cap = want;
    return KCFX_OK;
} KCFX_CATCH("kcfx_ref_reserve_multi")

int32_t kcfx_screen_ref_multi(kcfx_db* const* dbs, int32_t n_db, kcfx_ref* ref, int32_t min_kmer_count, kcfx_result* out) try {
    if (!ref || n_db < 0 || (n_db > 0 && !dbs)) return fail(KCFX_ERR_ARG, "null argument");
    if (n_db == 0 || ref->n_windows == 0) {
        if (min_kmer_count < 1) return fail(KCFX_ERR_ARG, "Minimum kmer count should be at least 1");
        return KCFX_OK;
    }
    if (!out) return fail(KCFX_ERR_ARG, "null argument");
    DeviceGuard g(ref->device);
    const size_t bytes = (size_t)n_db * ref->n_windows * sizeof(kcfx_result);
    if (ref->mres_bytes < bytes) {          // sized for the largest n_db seen so far
        if (ref->d_mres) (void)dev_free(ref->d_mres);
        ref->d_mres = nullptr;
        ref->mres_bytes = 0;
        KCFX_HIP(dev_alloc(&ref->d_mres, bytes));
        ref->mres_bytes = bytes;
    }
    if (int32_t rc = kcfx_screen_ref_multi_async(dbs, n_db, ref, min_kmer_count, ref->d_mres, ref->stream)) return rc;
    KCFX_HIP(hipMemcpyAsync(out, ref->d_mres, bytes, hipMemcpyDeviceToHost, ref->stream));
    KCFX_HIP(hipStreamSynchronize(ref->stream));
    return KCFX_OK;
} KCFX_CATCH("kcfx_screen_ref_multi")

int32_t kcfx_screen_ref(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, kcfx_result* out) try {
    if (!db || !ref || (!out && ref->n_windows > 0)) return fail(KCFX_ERR_ARG, "null argument");
    if (ref->n_windows == 0) return KCFX_OK;
    if (int32_t rc = kcfx_screen_ref_async(db, ref, min_kmer_count, ref->d_res, ref->stream)) return rc;
    DeviceGuard g(db->device);
    KCFX_HIP(hipMemcpyAsync(out, ref->d_res, (size_t)ref->n_windows * sizeof(kcfx_result), hipMemcpyDeviceToHost, ref->stream));
    KCFX_HIP(hipStreamSynchronize(ref->stream));
    return KCFX_OK;
} KCFX_CATCH("kcfx_screen_ref")

int32_t kcfx_screen(kcfx_db* db, const uint8_t* text, int64_t text_len, const int64_t* win_begin,
                    const int32_t* win_len, int32_t n_windows, int32_t min_kmer_count, kcfx_result* out) try {
    if (!db) return fail(KCFX_ERR_ARG, "null argument");
    kcfx_ref* ref = nullptr;
    kcfx_ref_opts one;
    one.expected_screens = 1;                               // (one database against this text: planned inside the launch)
    one.reserved = 0;
    if (int32_t rc = kcfx_ref_create_opts(db->device, db->info.kmer_length, text, text_len, win_begin, win_len, n_windows, &one, &ref))
        return rc;
    const int32_t rc = kcfx_screen_ref(db, ref, min_kmer_count, out);
    kcfx_ref_destroy(ref);
    return rc;
} KCFX_CATCH("kcfx_screen")

// ---- cohort table ------------------------------------------------------------------------------------------------------
int32_t kcfx_cohort_open(const char* const* kmc_prefixes, int32_t n_samples, int32_t device, const kcfx_db_opts* opts,
                         kcfx_cohort** out) try {
    if (!out || !kmc_prefixes) return fail(KCFX_ERR_ARG, "null argument");
    *out = nullptr;
    if (n_samples < 1 || n_samples > KCFX_COHORT_MAX)
        return fail(KCFX_ERR_ARG, "a cohort table holds 1 .. " + std::to_string(KCFX_COHORT_MAX) + " samples (screen larger cohorts in batches)");
    if (opts && opts->expected_queries < 0) return fail(KCFX_ERR_ARG, "kcfx_db_opts: negative expected_queries");
    const double t0 = now_s();
    std::vector<KmcHeader> hs((size_t)n_samples);
    int64_t N = 0;
    for (int32_t s = 0; s < n_samples; ++s) {
        if (!kmc_prefixes[s]) return fail(KCFX_ERR_ARG, "null database prefix");
        std::string perr;
        if (int32_t rc = parse_prefix_file(std::string(kmc_prefixes[s]) + ".kmc_pre", hs[(size_t)s], perr)) return fail(rc, perr);
        const KmcHeader& h = hs[(size_t)s];
        if (h.k < CELL_K_MIN || h.k > CELL_K_MAX)
            return fail(KCFX_ERR_ARG, "cohort tables need k = 19 .. 31 (" + std::string(kmc_prefixes[s]) + " has k = " + std::to_string(h.k) + ")");
        if (h.k != hs[0].k) return fail(KCFX_ERR_ARG, "Kmer size mismatch between the KMC databases");          // KCFHeader.equals (Cohort.java:71-119)
        if (h.counter_size < 1 || h.counter_size > 2)
            return fail(KCFX_ERR_ARG, "cohort tables need counters of 1 or 2 bytes (" + std::string(kmc_prefixes[s]) + ")");
        if (h.both != hs[0].both) return fail(KCFX_ERR_ARG, "cohort tables need one strandedness across the samples");
        N += h.total;
    }
    if (int32_t rc = check_device(device)) return rc;
    DeviceGuard g(device);
    if (!g.ok) return fail(KCFX_ERR_DEVICE, "hipSetDevice failed");
    struct Hold {                                          // everything built so far is released on every exit path
        kcfx_cohort* c = nullptr;
        uint64_t* d_keys = nullptr;
        uint32_t* d_counts = nullptr;
        uint64_t* d_pw = nullptr;                          // the records' cells, kept by the first build pass (RecProbes)
        uint8_t* d_ps = nullptr;
        ~Hold() {
            if (d_keys) (void)dev_free(d_keys);
            if (d_counts) (void)dev_free(d_counts);
            if (d_pw) (void)dev_free(d_pw);
            if (d_ps) (void)dev_free(d_ps);
            kcfx_cohort_close(c);
        }
    } hold;
    kcfx_cohort* co = hold.c = new kcfx_cohort();
    co->device = device;
    co->n_samples = n_samples;
    kcfx_db* db = co->keys = new kcfx_db();
    db->device = device;
    info_from_header(hs[0], db->info);
    db->info.total_kmers = N;
    db->info.counter_size = 2;
    db->info.device = device;
    db->expected_queries = opts ? opts->expected_queries : 0;
    db->line_footprint = LINE_BYTES + 16 * COHORT_STRIDE;           // cells + 8 bytes of counters per cell: the size policy budgets both
    hipError_t e = db_stream(device, &db->stream, nullptr, true);
    if (e != hipSuccess) return hip_fail(e, "hipStreamCreate");
    struct LoadDone {                           // (the stream is another load's to take again when this one is over, however it ends)
        hipStream_t st;
        ~LoadDone() { db_stream_load_done(st); }
    } load_done{db->stream};
    const size_t nn = (size_t)std::max<int64_t>(N, 1);
    if ((e = dev_alloc(&hold.d_keys, nn * 8)) != hipSuccess || (e = dev_alloc(&hold.d_counts, nn * 4)) != hipSuccess)
        return hip_fail(e, "kcfx_cohort_open: the decoded records of all samples (12 bytes each) must fit the device");
    int64_t off = 0;
    for (int32_t s = 0; s < n_samples; ++s) {
        if (int32_t rc = decode_file_to_device(std::string(kmc_prefixes[s]) + ".kmc_suf", hs[(size_t)s], hold.d_keys + off, hold.d_counts + off, db->stream))
            return rc;
        off += hs[(size_t)s].total;
    }
    // bid ranks: how many samples hold each k-mer (a temporary open-addressing table over all records, ~2 slots per record)
    {
        uint64_t slots = 1024;
        while (slots < 2 * (uint64_t)N) slots <<= 1;
        if (slots > (1ull << 32)) slots = 1ull << 32;
        if ((double)slots < 1.25 * (double)N) return fail(KCFX_ERR_DEVICE, "kcfx_cohort_open: too many records for the vote table");
        uint4* d_votes = nullptr;
        if ((e = dev_alloc(&d_votes, (size_t)slots * sizeof(uint4))) != hipSuccess) return hip_fail(e, "kcfx_cohort_open: vote table");
        e = hipMemsetAsync(d_votes, 0xFF, (size_t)slots * sizeof(uint4), db->stream);
        for (int phase = 0; phase < 2 && e == hipSuccess; ++phase)
            for (int64_t first = 0; first < N && e == hipSuccess; first += (1ll << 30))
                e = launch_cohort_votes(hold.d_keys + first, hold.d_counts + first, std::min<int64_t>(1ll << 30, N - first), d_votes,
                                        (uint32_t)(slots - 1), phase, db->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(db->stream);
        else (void)hipStreamSynchronize(db->stream);       // (a failed launch: earlier kernels may still be writing the block)
        (void)dev_free(d_votes);                           // (64 GiB at most: straight back to the driver, the counters below need the room)
        if (e != hipSuccess) return hip_fail(e, "kcfx_cohort_open: votes");
    }
    // the key table: the ordinary build passes over the records of all samples; a cell's count field = the rank its k-mer bid with
    if (int32_t rc = table_alloc(db, N, 2, true, false)) return rc;
    if (db->fmt != FORMAT_CELLS) return fail(KCFX_ERR_ARG, "cohort tables need a cell table (KCFX_TABLE_FORMAT / KCFX_GENERIC_KERNEL / too many k-mers)");
    const int32_t k = hs[0].k, both = db->info.both_strands;
    // the cells of every record once, for the four later passes and the S counter passes (9 bytes per record, if they fit)
    if (hook("KCFX_LOAD_NO_PROBES") == nullptr && (double)nn * 9.0 < 0.25 * (double)free_device_bytes()) {
        if (dev_alloc(&hold.d_pw, nn * 8) != hipSuccess || dev_alloc(&hold.d_ps, nn) != hipSuccess) {
            (void)hipGetLastError();
            if (hold.d_pw) (void)dev_free(hold.d_pw);
            hold.d_pw = nullptr;
            hold.d_ps = nullptr;
        }
    }
    auto pass = [&](int phase) -> hipError_t {
        for (int64_t first = 0; first < N; first += (1ll << 30)) {
            const int64_t n = std::min<int64_t>(1ll << 30, N - first);
            const uint64_t* kk = hold.d_keys + first;
            const uint32_t* cc = hold.d_counts + first;
            uint64_t* pw = hold.d_pw ? hold.d_pw + first : nullptr;
            uint8_t* ps = hold.d_ps ? hold.d_ps + first : nullptr;
            hipError_t e2 = hipSuccess;
            if (phase == 0)
                e2 = pw ? launch_cells_resolve(db->d_pop, db->pop_bits, k, db->lbits, both, kk, cc, n, pw, ps, db->stream, db->ext, db->crowd_by_minimizer)
                        : launch_cells_count(db->d_pop, db->pop_bits, k, db->lbits, both, kk, cc, n, db->stream, 1, db->ext, db->crowd_by_minimizer);
            else if (phase == 1 || phase == 2) e2 = launch_cells_insert(db->d_lines, db->d_pop, db->pop_bits, phase == 1 ? 1 : 0, k, db->lbits, both, kk, cc, n, db->stream, 1, pw, ps, db->ext, db->regions, 0, db->crowd_by_minimizer);
            else if (phase == 3) e2 = launch_cells_settle(db->d_lines, k, db->lbits, both, kk, cc, n, db->d_stats, db->stream, 1, pw, ps, db->ext, db->regions);
            else e2 = launch_cells_stash(db->d_lines, k, db->lbits, both, kk, cc, n, db->d_stash, db->stash_mask, db->stream, 1, pw, ps, db->ext, db->regions);
            if (e2 != hipSuccess) return e2;
        }
        return hipSuccess;
    };
    for (int phase = 0; phase <= 3; ++phase)
        if ((e = pass(phase)) != hipSuccess) return hip_fail(e, "kcfx_cohort_open: key table pass");
    if (int32_t rc = cells_stash_alloc(db)) return rc;       // (sized for the records that lost: a k-mer of several samples counts once per sample)
    if ((e = pass(4)) != hipSuccess) return hip_fail(e, "kcfx_cohort_open: stash pass");
    if (int32_t rc = table_finish(db)) return rc;
    // the counters: COHORT_STRIDE u8 per cell, COHORT_STRIDE u16 per stash slot
    const size_t cell_bytes = ((size_t)db->n_lines * 16) * COHORT_STRIDE, stash_bytes = ((size_t)db->stash_mask + 1) * COHORT_STRIDE * 2;
    if ((e = dev_alloc(&co->d_counts, cell_bytes)) != hipSuccess || (e = dev_alloc(&co->d_scounts, stash_bytes)) != hipSuccess ||
        (e = hipMemsetAsync(co->d_counts, 0, cell_bytes, db->stream)) != hipSuccess ||
        (e = hipMemsetAsync(co->d_scounts, 0, stash_bytes, db->stream)) != hipSuccess)
        return hip_fail(e, "kcfx_cohort_open: the counters (16 bytes per cell) do not fit the device");
    off = 0;
    for (int32_t s = 0; s < n_samples; ++s) {
        if ((e = launch_cohort_fill(db->view(), k, both, hold.d_keys + off, hold.d_counts + off, hs[(size_t)s].total, s, co->d_counts,
                                    co->d_scounts, db->stream, hold.d_pw ? hold.d_pw + off : nullptr, hold.d_ps ? hold.d_ps + off : nullptr)) != hipSuccess)
            return hip_fail(e, "kcfx_cohort_open: counter pass");
        off += hs[(size_t)s].total;
    }
    if ((e = hipStreamSynchronize(db->stream)) != hipSuccess) return hip_fail(e, "kcfx_cohort_open");
    co->info.n_samples = n_samples;
    co->info.kmer_length = k;
    co->info.both_strands = both;
    co->info.device = device;
    co->info.total_records = N;
    co->info.table_lines = db->n_lines;
    co->info.table_bytes = db->info.table_bytes + (int64_t)cell_bytes + (int64_t)stash_bytes;
    co->info.stash_records = db->info.stash_kmers;
    co->info.load_seconds = now_s() - t0;
    hold.c = nullptr;
    *out = co;
    return KCFX_OK;
} KCFX_CATCH("kcfx_cohort_open")

int32_t kcfx_cohort_info(const kcfx_cohort* cohort, kcfx_cohort_info_t* info) {
    if (!cohort || !info) return fail(KCFX_ERR_ARG, "null argument");
    *info = cohort->info;
    return KCFX_OK;
}

void kcfx_cohort_close(kcfx_cohort* cohort) {
    if (!cohort) return;
    {
        DeviceGuard g(cohort->device);
        if (cohort->d_counts) (void)dev_free(cohort->d_counts);
        if (cohort->d_scounts) (void)dev_free(cohort->d_scounts);
    }
    db_free(cohort->keys);
    delete cohort;
}

int32_t kcfx_cohort_screen_ref_async(kcfx_cohort* cohort, kcfx_ref* ref, int32_t min_kmer_count, void* out_device, void* hip_stream) try {
    if (!cohort || !ref || !out_device) return fail(KCFX_ERR_ARG, "null argument");
    if (cohort->device != ref->device) return fail(KCFX_ERR_ARG, "cohort table and reference live on different devices");
    if (ref->k != cohort->info.kmer_length) return fail(KCFX_ERR_ARG, "reference was planned for a different k-mer length");
    if (min_kmer_count < 1) return fail(KCFX_ERR_ARG, "Minimum kmer count should be at least 1");   // GetVariants.java:383-385
    if (ref->n_windows == 0) return KCFX_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    DeviceGuard g(ref->device);
    if (int32_t rc = kcfx_ref_reserve_multi(ref, cohort->n_samples)) return rc;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ref->prof && ref->n_tiles > 0) {
        if (ref->events_used == ref->events.size()) {
            hipEvent_t a, b;
            KCFX_HIP(hipEventCreate(&a));
            KCFX_HIP(hipEventCreate(&b));
            ref->events.emplace_back(a, b);
        }
        ev0 = ref->events[ref->events_used].first;
        ev1 = ref->events[ref->events_used].second;
        ref->events_used++;
    }
    CohortView cv{cohort->keys->view(), reinterpret_cast<const uint2*>(cohort->d_counts), reinterpret_cast<const uint4*>(cohort->d_scounts),
                  cohort->n_samples};
    KCFX_HIP(launch_screen_cohort(ref->d_text, ref->plan(), ref->d_tiles, ref->n_tiles, cv, ref->k, cohort->info.both_strands, min_kmer_count,
                                  ref->d_mbits, ref->d_mcsum, st, ev0, ev1));
    KCFX_HIP(launch_reduce_windows(ref->d_tiles, ref->n_tiles, cohort->n_samples, ref->d_mbits, ref->d_mcsum, ref->d_win_tile0,
                                   ref->n_windows, ref->k, reinterpret_cast<kcfx_result*>(out_device), st, 1));
    return KCFX_OK;
} KCFX_CATCH("kcfx_cohort_screen_ref_async")

int32_t kcfx_cohort_screen_ref(kcfx_cohort* cohort, kcfx_ref* ref, int32_t min_kmer_count, kcfx_result* out) try {
    if (!cohort || !ref) return fail(KCFX_ERR_ARG, "null argument");
    if (ref->n_windows == 0) {
        if (min_kmer_count < 1) return fail(KCFX_ERR_ARG, "Minimum kmer count should be at least 1");
        return KCFX_OK;
    }
    if (!out) return fail(KCFX_ERR_ARG, "null argument");
    DeviceGuard g(ref->device);
    const size_t bytes = (size_t)cohort->n_samples * ref->n_windows * sizeof(kcfx_result);
    if (ref->mres_bytes < bytes) {
        if (ref->d_mres) (void)dev_free(ref->d_mres);
        ref->d_mres = nullptr;
        ref->mres_bytes = 0;
        KCFX_HIP(dev_alloc(&ref->d_mres, bytes));
        ref->mres_bytes = bytes;
    }
    if (int32_t rc = kcfx_cohort_screen_ref_async(cohort, ref, min_kmer_count, ref->d_mres, ref->stream)) return rc;
    KCFX_HIP(hipMemcpyAsync(out, ref->d_mres, bytes, hipMemcpyDeviceToHost, ref->stream));
    KCFX_HIP(hipStreamSynchronize(ref->stream));
    return KCFX_OK;
} KCFX_CATCH("kcfx_cohort_screen_ref")

// Data.java:87 and :95-107. Separate statements + -ffp-contract=off: strict left-to-right IEEE double, no FMA.
int32_t kcfx_score(const kcfx_result* res, int64_t n, double wi, double wt, double wr, double* score, double* mean_count) {
    if (n < 0 || (n > 0 && !res)) return fail(KCFX_ERR_ARG, "null argument");
    const double weights[3] = {wi, wt, wr};                                 // GetVariants.java:388-390
    const bool bad = (weights[0] + weights[1] + weights[2] != 1.0);         // Data.java:101
    for (int64_t i = 0; i < n; ++i) {
        const kcfx_result& r = res[i];
        if (mean_count) mean_count[i] = (r.obs > 0) ? (double)r.kmer_count_sum / r.obs : 0.00;   // :87
        if (!score) continue;
        if (r.obs == 0 || r.total_kmers == 0 || r.eff_length == 0) {      // :96-98
            score[i] = 0;
            continue;
        }
        const int32_t tail = r.left + r.right;                              // :170-172
        const double a = weights[2] * ((double)r.obs / r.total_kmers);
        const double b = weights[0] * (1.0f - ((double)r.inner / r.eff_length));
        const double c = weights[1] * (1.0f - ((double)tail / r.eff_length));
        double s = a + b;
        s = s + c;
        s = s * 100.0f;
        score[i] = s;
    }
    if (bad) return fail(KCFX_ERR_WEIGHTS, "Weights should sum to 1.0");
    return KCFX_OK;
}

// GetVariants.java:293-320
int32_t kcfx_windows(int32_t seq_len, int32_t window, int32_t step, int32_t kmer_length, int32_t* starts, int32_t* ends,
                     int64_t cap, int64_t* n_out) try {
    if (!n_out || kmer_length < 1) return fail(KCFX_ERR_ARG, "null argument");
    if (window <= 0) return fail(KCFX_ERR_ARG, "Window size is required for window model");   // GetVariants.java:360-362
    const int32_t k = kmer_length;
    int64_t n = 0;
    auto emit = [&](int64_t s, int64_t e) {
        if (n < cap) {
            if (starts) starts[n] = (int32_t)s;
            if (ends) ends[n] = (int32_t)e;
        }
        ++n;
    };
    if (step > 0) {
        for (int64_t pos = 0; pos < seq_len; pos += step) {
            const int64_t e = std::min<int64_t>(pos + window, seq_len);
            if (e - pos >= k) emit(pos, e);
        }
    } else {
        if (window <= k - 1) return fail(KCFX_ERR_ARG, "window size must exceed k-1 in tiling mode (the reference never terminates)");
        int64_t last_end = 0;
        while (last_end < seq_len) {
            const int64_t s = std::max<int64_t>(0, last_end - k + 1);
            const int64_t e = std::min<int64_t>(s + window, seq_len);
            if (e - s >= k) emit(s, e);
            last_end = e;
        }
    }
    *n_out = n;
    return KCFX_OK;
} KCFX_CATCH("kcfx_windows")

int32_t kcfx_prof_enable(kcfx_ref* ref, int32_t enable) {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    ref->prof = enable != 0;
    return KCFX_OK;
}

int32_t kcfx_prof_read_plan(kcfx_ref* ref, double* kernel_ms_sum, int64_t* launches, int32_t reset) try {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    DeviceGuard g(ref->device);
    double sum = 0;
    for (size_t i = 0; i < ref->plan_events_used; ++i) {
        KCFX_HIP(hipEventSynchronize(ref->plan_events[i].second));
        float ms = 0;
        KCFX_HIP(hipEventElapsedTime(&ms, ref->plan_events[i].first, ref->plan_events[i].second));
        sum += ms;
    }
    if (kernel_ms_sum) *kernel_ms_sum = sum;
    if (launches) *launches = (int64_t)ref->plan_events_used;
    if (reset) ref->plan_events_used = 0;
    return KCFX_OK;
} KCFX_CATCH("kcfx_prof_read_plan")

int32_t kcfx_prof_read(kcfx_ref* ref, double* kernel_ms_sum, int64_t* launches, int32_t reset) try {
    if (!ref) return fail(KCFX_ERR_ARG, "null argument");
    DeviceGuard g(ref->device);
    double sum = 0;
    for (size_t i = 0; i < ref->events_used; ++i) {
        KCFX_HIP(hipEventSynchronize(ref->events[i].second));
        float ms = 0;
        KCFX_HIP(hipEventElapsedTime(&ms, ref->events[i].first, ref->events[i].second));
        sum += ms;
    }
    if (kernel_ms_sum) *kernel_ms_sum = sum;
    if (launches) *launches = (int64_t)ref->events_used;
    if (reset) ref->events_used = 0;
    return KCFX_OK;
} KCFX_CATCH("kcfx_prof_read")

}  // extern "C"
