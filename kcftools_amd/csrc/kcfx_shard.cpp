// kcfx_shard.cpp -- the multi-GPU entry points of the C ABI (include/kcfx.h: kcfx_ctx_*, kcfx_screen_sharded).
//
// The path shards into independent units: a (window, sample) result depends only on that window's text and that sample's
// database (GetVariants.java:138-150 hands windows to a thread pool one by one; utils/run_kcftools.sh:124-140 runs the
// samples as separate processes). So there is no exchange step: one worker thread per device, each with its own resident
// reference, and either
//   samples >= devices : sample-sharded -- device g screens samples g, g + G, ...; every device holds the whole window list
//   one sample         : window-sharded -- device g screens a contiguous window range (balanced by k-mer start positions)
//                        and uploads only the text that range covers.
//   1 < samples < devices : grid -- the sample-major sequence of (sample, window) units is cut into equal contiguous parts: a
//                        device opens at most two databases instead of all of them (KCFX_SHARD_WINDOWS, still selectable,
//                        screens a window range of EVERY sample on every device).
//   cohort tables       : (KCFX_SHARD_COHORT, on request) the samples in groups of eight, each group ONE cohort table
//                        (kcfx_cohort_open: one look-up per reference k-mer for the eight); the group-major sequence of
//                        (group, window) units is cut like the grid's. 8 samples on 8 devices: every device the same table and
//                        an eighth of the windows; 32 samples: four tables, two devices each.
//   table partitioned   : (KCFX_SHARD_TABLE, on request) every sample's TABLE is cut into G slices by minimizer (kcfx_db_open_slice);
//                        device g holds slice g, screens the whole window list against it with the lanes of foreign minimizers
//                        idle, the tiles' hit bitmaps are OR-ed on the first device (kcfx_ref_merge_from: a peer copy + one
//                        small kernel -- the one exchange step on this path) and reduced there. For a database that a single
//                        device can only hold densely (screen_table_sharded below).
// Inside a worker the next sample's database is opened (file read, upload, table build) by a loader thread while the
// current one is screened: at most two databases are resident per device, and a sample costs max(load, screen), not the sum.
// Built on the public entry points only (kcfx_db_open / kcfx_ref_create / kcfx_screen_ref); HIP is never touched here.
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kcfx_internal.h"

struct kcfx_ctx {
    std::vector<int32_t> devices;
};

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int32_t fail(int32_t code, const std::string& msg) {
    kcfx::set_error(msg);
    return code;
}

// contiguous [lo, hi) window ranges with equal numbers of k-mer start positions (gene-model windows span 31 bp .. 100 kbp)
std::vector<int32_t> window_cuts(const int32_t* win_len, int32_t n, int32_t k, int32_t parts) {
    std::vector<int64_t> cum((size_t)n + 1, 0);
    for (int32_t i = 0; i < n; ++i) cum[(size_t)i + 1] = cum[(size_t)i] + std::max<int64_t>(0, (int64_t)win_len[i] - k + 1);
    const int64_t total = cum[(size_t)n];
    std::vector<int32_t> cuts((size_t)parts + 1, n);
    cuts[0] = 0;
    for (int32_t p = 1; p < parts; ++p) {
        const int64_t target = total * p / parts;
        const int32_t c = (int32_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        cuts[(size_t)p] = std::max(cuts[(size_t)p - 1], std::min(c, n));
    }
    return cuts;
}

struct Segment {
    int32_t sample;                   // the sample screened (cohort mode: the first of the group) ...
    int32_t w_lo, w_hi;               // ... against this window range
    int32_t n = 1;                    // samples of the unit: 1, or the 2 .. KCFX_COHORT_MAX of a cohort table
};

struct Worker {
    int32_t device = 0;
    std::vector<Segment> segs;        // what this device screens, in order
    // outcome
    int32_t rc = KCFX_OK;
    std::string error;
    double t_ref = 0, t_open = 0, t_screen = 0, t_wait = 0;
};

// Who screens what: plan[g] = the (sample, window range) segments of device g, in order. Pure host arithmetic (also behind
// kcfx_shard_plan, which the CPU tests drive): every (sample, window) pair lies in exactly one segment.
std::vector<std::vector<Segment>> plan_segments(int32_t mode, int32_t G, int32_t n_samples, const int32_t* win_len, int32_t n_windows,
                                                int32_t kmer_length) {
    std::vector<std::vector<Segment>> plan((size_t)G);
    if (mode == KCFX_SHARD_TABLE) {
        // every device screens every sample's whole window list against ITS slice of that sample's table
        for (int32_t g = 0; g < G; ++g)
            for (int32_t s = 0; s < n_samples; ++s) plan[(size_t)g].push_back(Segment{s, 0, n_windows});
    } else if (mode == KCFX_SHARD_SAMPLES) {
        for (int32_t g = 0; g < G; ++g)
            for (int32_t s = g; s < n_samples; s += G) plan[(size_t)g].push_back(Segment{s, 0, n_windows});
    } else if (mode == KCFX_SHARD_WINDOWS) {
        const std::vector<int32_t> cuts = window_cuts(win_len, n_windows, kmer_length, G);
        for (int32_t g = 0; g < G; ++g)
            for (int32_t s = 0; s < n_samples; ++s)
                if (cuts[(size_t)g + 1] > cuts[(size_t)g]) plan[(size_t)g].push_back(Segment{s, cuts[(size_t)g], cuts[(size_t)g + 1]});
    } else {
        // KCFX_SHARD_GRID: the sample-major sequence of (sample, window) units -- S runs of the window list -- is cut into G
        // contiguous parts of equal k-mer count: a device screens the tail of one sample's windows and the head of the next
        // one's, so it opens at most ceil(S / G) + 1 databases (2 when S < G) where KCFX_SHARD_WINDOWS opens all S on every
        // device, and the parts are as even as a single sample's window cuts.
        // KCFX_SHARD_COHORT: the same over GROUPS of up to KCFX_COHORT_MAX consecutive samples, each screened as one cohort
        // table (a look-up costs about the same whatever the group's size: the units weigh what their windows weigh).
        const int32_t per_unit = mode == KCFX_SHARD_COHORT ? KCFX_COHORT_MAX : 1;
        const int32_t n_units = (n_samples + per_unit - 1) / per_unit;
        // cut positions in units of k-mers over n_units repetitions: part g = [g, g + 1) * n_units / G of the n_units-fold sequence
        std::vector<int64_t> cum((size_t)n_windows + 1, 0);
        for (int32_t i = 0; i < n_windows; ++i) cum[(size_t)i + 1] = cum[(size_t)i] + std::max<int64_t>(0, (int64_t)win_len[i] - kmer_length + 1);
        const int64_t per_sample = cum[(size_t)n_windows];
        auto locate = [&](int64_t pos, int32_t& unit, int32_t& w) {      // position in the n_units-fold k-mer sequence -> (unit, first window at or after it)
            if (per_sample <= 0) {
                unit = n_units;
                w = 0;
                return;
            }
            unit = (int32_t)std::min<int64_t>(pos / per_sample, n_units);
            const int64_t r = pos - (int64_t)unit * per_sample;
            w = unit >= n_units ? 0 : (int32_t)(std::lower_bound(cum.begin(), cum.end(), r) - cum.begin());
            if (w > n_windows) w = n_windows;
        };
        auto segment_of = [&](int32_t unit, int32_t lo, int32_t hi) {
            Segment sg;
            sg.sample = unit * per_unit;
            sg.w_lo = lo;
            sg.w_hi = hi;
            sg.n = std::min(per_unit, n_samples - sg.sample);
            return sg;
        };
        const int64_t total = per_sample * n_units;
        int32_t ps = 0, pw = 0;                                           // where the previous part ended
        for (int32_t g = 0; g < G; ++g) {
            int32_t es = n_units, ew = 0;
            if (g + 1 < G) locate(total * (g + 1) / G, es, ew);
            if (per_sample <= 0) {                                        // no k-mers at all: device 0 takes everything
                if (g == 0)
                    for (int32_t u = 0; u < n_units; ++u) plan[0].push_back(segment_of(u, 0, n_windows));
                continue;
            }
            // from (ps, pw) up to (es, ew)
            for (int32_t u = ps; u <= es && u < n_units; ++u) {
                const int32_t lo = (u == ps) ? pw : 0, hi = (u == es) ? ew : n_windows;
                if (hi > lo) plan[(size_t)g].push_back(segment_of(u, lo, hi));
            }
            ps = es;
            pw = ew;
        }
    }

    return plan;
}

// what the slice loaders of the most recent KCFX_SHARD_TABLE call read from the host side: bytes of .kmc_suf, chunks read, chunks that
// a sibling loader had already read (kcfx_shard_slice_io)
std::mutex g_slice_io_mu;
int64_t g_slice_io[3] = {0, 0, 0};

// KCFX_SHARD_TABLE: sample after sample, the sample's table in G slices on the G devices. Each device keeps ONE reference of the
// whole window list for the call (planned once when several samples follow, text-only for one); per sample a thread per device
// opens its slice, screens, closes; then the slices' bitmaps meet on the first device and are reduced there.
int32_t screen_table_sharded(const std::vector<int32_t>& devices, const char* const* kmc_prefixes, int32_t n_samples, int32_t kmer_length,
                             const uint8_t* text, int64_t text_len, const int64_t* win_begin, const int32_t* win_len, int32_t n_windows,
                             int32_t min_kmer_count, kcfx_result* out, kcfx_shard_stats* stats) {
    const int32_t G = (int32_t)devices.size();
    if (G > KCFX_SLICES_MAX) return fail(KCFX_ERR_ARG, "KCFX_SHARD_TABLE: at most " + std::to_string(KCFX_SLICES_MAX) + " devices");
    struct Dev {
        kcfx_ref* ref = nullptr;
        int32_t rc = KCFX_OK;
        std::string error;
        double t_ref = 0, t_open = 0, t_screen = 0;
    };
    std::vector<Dev> dev((size_t)G);
    struct Cleanup {
        std::vector<Dev>& d;
        ~Cleanup() { for (auto& x : d) kcfx_ref_destroy(x.ref); }
    } cleanup{dev};
    auto in_parallel = [&](const std::function<void(int32_t)>& body) {
        auto guarded = [&](int32_t g) {
            try {
                body(g);
            } catch (...) {
                dev[(size_t)g].rc = kcfx::exception_to_status("kcfx_screen_sharded (table slices)");
                try { dev[(size_t)g].error = kcfx_last_error(); } catch (...) {}
            }
        };
        if (G == 1) {
            guarded(0);
            return;
        }
        std::vector<std::thread> th;
        th.reserve((size_t)G);
        try {
            for (int32_t g = 0; g < G; ++g) th.emplace_back(guarded, g);
        } catch (...) {
            dev[th.size()].rc = KCFX_ERR_HOST;
            dev[th.size()].error = "could not start a worker thread";
        }
        for (auto& x : th) x.join();
    };
    auto first_error = [&]() -> int32_t {
        for (int32_t g = 0; g < G; ++g)
            if (dev[(size_t)g].rc != KCFX_OK) return fail(dev[(size_t)g].rc, "device " + std::to_string(devices[(size_t)g]) + " (slice " + std::to_string(g) + "): " + dev[(size_t)g].error);
        return KCFX_OK;
    };
    int64_t queries = 0;
    for (int32_t w = 0; w < n_windows; ++w) queries += std::max<int64_t>(0, (int64_t)win_len[w] - kmer_length + 1);
    {
        std::lock_guard<std::mutex> lk(g_slice_io_mu);
        g_slice_io[0] = g_slice_io[1] = g_slice_io[2] = 0;
    }
    in_parallel([&](int32_t g) {
        Dev& d = dev[(size_t)g];
        const double t = now_s();
        kcfx_ref_opts ro;
        ro.expected_screens = n_samples == 1 ? 1 : 0;
        ro.reserved = 0;
        d.rc = kcfx_ref_create_opts(devices[(size_t)g], kmer_length, text, text_len, win_begin, win_len, n_windows, &ro, &d.ref);
        if (d.rc) d.error = kcfx_last_error();
        d.t_ref = now_s() - t;
    });
    if (int32_t rc = first_error()) return rc;
    double t_merge = 0;
    for (int32_t s = 0; s < n_samples; ++s) {
        // the G loaders of this sample walk the same file in the same passes: ONE read of every chunk for all of them (SuffixCache,
        // kcfx_api.cpp) -- each device then copies it over its own link. Null (no pinned memory to be had): every loader reads for itself.
        struct CacheHold {
            kcfx::SuffixCache* c;
            ~CacheHold() { kcfx::suffix_cache_destroy(c); }
        } cache{G > 1 ? kcfx::suffix_cache_create(kmc_prefixes[s], G) : nullptr};
        in_parallel([&](int32_t g) {
            Dev& d = dev[(size_t)g];
            kcfx_db_opts opts;
            opts.expected_queries = std::max<int64_t>(1, queries);
            opts.no_fallback = 1;
            opts.reserved = 0;
            kcfx_db* db = nullptr;
            double t = now_s();
            d.rc = kcfx::db_open_slice_shared(kmc_prefixes[s], devices[(size_t)g], &opts, G, g, cache.c, &db);
            d.t_open += now_s() - t;
            if (!d.rc) {
                kcfx_db_info_t info;
                kcfx_db_info(db, &info);
                t = now_s();
                if (info.kmer_length != kmer_length) {
                    d.rc = KCFX_ERR_ARG;
                    d.error = "Kmer size mismatch between the KMC databases";              // KCFHeader.equals (Cohort.java:71-119)
                } else {
                    d.rc = kcfx_screen_ref_partial(db, d.ref, min_kmer_count);
                    if (d.rc) d.error = kcfx_last_error();
                }
                d.t_screen += now_s() - t;
            } else {
                d.error = kcfx_last_error();
            }
            kcfx_db_close(db);
        });
        if (int32_t rc = first_error()) return rc;
        if (cache.c && (getenv("KCFX_LOAD_TRACE") || getenv("KCFX_VERBOSE"))) {
            int64_t reads = 0, hits = 0, bytes = 0;
            kcfx::suffix_cache_stats(cache.c, &reads, &hits, &bytes);
            fprintf(stderr, "[kcfx load] %s: %d slice loaders shared one file stream: %lld chunks read (%.2f GB), %lld served from the cache\n", kmc_prefixes[s], G,
                    (long long)reads, (double)bytes / 1e9, (long long)hits);
        }
        {
            int64_t reads = 0, hits = 0, bytes = 0;
            kcfx::suffix_cache_stats(cache.c, &reads, &hits, &bytes);
            std::lock_guard<std::mutex> lk(g_slice_io_mu);
            g_slice_io[0] += bytes;
            g_slice_io[1] += reads;
            g_slice_io[2] += hits;
        }
        const double t = now_s();
        for (int32_t g = 1; g < G; ++g)
            if (int32_t rc = kcfx_ref_merge_from(dev[0].ref, dev[(size_t)g].ref)) return rc;
        if (int32_t rc = kcfx_ref_finish(dev[0].ref, out + (size_t)s * n_windows)) return rc;
        t_merge += now_s() - t;
    }
    if (stats) {
        for (const Dev& d : dev) {
            stats->seconds_ref = std::max(stats->seconds_ref, d.t_ref);
            stats->seconds_db_open_sum += d.t_open;
            stats->seconds_screen_sum += d.t_screen;
            stats->seconds_exposed_load = std::max(stats->seconds_exposed_load, d.t_open);   // (nothing hides a slice's load here)
        }
        stats->seconds_screen_sum += t_merge;
    }
    return KCFX_OK;
}

}  // namespace

extern "C" {

int32_t kcfx_ctx_create(const int32_t* devices, int32_t n_devices, kcfx_ctx** out) try {
    if (!out || n_devices < 1 || !devices) return fail(KCFX_ERR_ARG, "kcfx_ctx_create: at least one device is required");
    *out = nullptr;
    int32_t have = 0;
    if (kcfx_device_count(&have) != KCFX_OK || have <= 0)
        return fail(KCFX_ERR_DEVICE, "no HIP device available (libkcfx has no CPU fallback)");
    for (int32_t i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= have) return fail(KCFX_ERR_ARG, "device index out of range");
    kcfx_ctx* c = new kcfx_ctx();
    c->devices.assign(devices, devices + n_devices);      // the same device may be listed twice (two shards on one GPU)
    *out = c;
    return KCFX_OK;
} catch (...) { return kcfx::exception_to_status("kcfx_ctx_create"); }

int32_t kcfx_ctx_n_devices(const kcfx_ctx* ctx) { return ctx ? (int32_t)ctx->devices.size() : 0; }

int32_t kcfx_shard_plan(int32_t shard_mode, int32_t n_devices, int32_t n_samples, const int32_t* win_len, int32_t n_windows,
                        int32_t kmer_length, int32_t* seg_device, int32_t* seg_sample, int32_t* seg_w_lo, int32_t* seg_w_hi,
                        int64_t cap, int64_t* n_segments, int32_t* mode_used) try {
    if (n_devices < 1 || n_samples < 0 || n_windows < 0 || kmer_length < 1 || !n_segments || (n_windows > 0 && !win_len))
        return fail(KCFX_ERR_ARG, "kcfx_shard_plan: bad argument");
    if (shard_mode < KCFX_SHARD_AUTO || shard_mode > KCFX_SHARD_TABLE) return fail(KCFX_ERR_ARG, "unknown shard mode");
    int32_t mode = shard_mode;
    if (mode == KCFX_SHARD_AUTO)
        mode = (n_samples >= n_devices || n_windows == 0) ? KCFX_SHARD_SAMPLES : (n_samples == 1 ? KCFX_SHARD_WINDOWS : KCFX_SHARD_GRID);
    if (mode_used) *mode_used = mode;
    int64_t n = 0;
    if (n_samples > 0 && n_windows > 0) {
        const std::vector<std::vector<Segment>> plan = plan_segments(mode, n_devices, n_samples, win_len, n_windows, kmer_length);
        for (int32_t g = 0; g < n_devices; ++g)
            for (const Segment& sg : plan[(size_t)g]) {
                if (n < cap) {
                    if (seg_device) seg_device[n] = g;
                    if (seg_sample) seg_sample[n] = sg.sample;
                    if (seg_w_lo) seg_w_lo[n] = sg.w_lo;
                    if (seg_w_hi) seg_w_hi[n] = sg.w_hi;
                }
                ++n;
            }
    }
    *n_segments = n;
    return KCFX_OK;
} catch (...) { return kcfx::exception_to_status("kcfx_shard_plan"); }

void kcfx_ctx_destroy(kcfx_ctx* ctx) { delete ctx; }

int32_t kcfx_shard_slice_io(int64_t* bytes_read, int64_t* chunks_read, int64_t* chunks_shared) {
    std::lock_guard<std::mutex> lk(g_slice_io_mu);
    if (bytes_read) *bytes_read = g_slice_io[0];
    if (chunks_read) *chunks_read = g_slice_io[1];
    if (chunks_shared) *chunks_shared = g_slice_io[2];
    return KCFX_OK;
}

int32_t kcfx_screen_sharded(kcfx_ctx* ctx, const char* const* kmc_prefixes, int32_t n_samples, int32_t kmer_length,
                            const uint8_t* text, int64_t text_len, const int64_t* win_begin, const int32_t* win_len,
                            int32_t n_windows, int32_t min_kmer_count, int32_t shard_mode, kcfx_result* out,
                            kcfx_shard_stats* stats) try {
    if (!ctx || n_samples < 0 || n_windows < 0 || text_len < 0 || (n_samples > 0 && !kmc_prefixes) ||
        (n_windows > 0 && (!win_begin || !win_len)) || (text_len > 0 && !text) ||
        ((int64_t)n_samples * n_windows > 0 && !out))
        return fail(KCFX_ERR_ARG, "null argument");
    if (min_kmer_count < 1) return fail(KCFX_ERR_ARG, "Minimum kmer count should be at least 1");   // GetVariants.java:383-385
    if (shard_mode < KCFX_SHARD_AUTO || shard_mode > KCFX_SHARD_TABLE) return fail(KCFX_ERR_ARG, "unknown shard mode");
    for (int32_t s = 0; s < n_samples; ++s)
        if (!kmc_prefixes[s]) return fail(KCFX_ERR_ARG, "null database prefix");
    for (int32_t w = 0; w < n_windows; ++w)
        if (win_begin[w] < 0 || win_len[w] < 0 || win_begin[w] + (int64_t)win_len[w] > text_len)
            return fail(KCFX_ERR_ARG, "window " + std::to_string(w) + " lies outside the text");      // FastaIndex.java:131-134
    const double t0 = now_s();
    const int32_t G = (int32_t)ctx->devices.size();
    int32_t mode = shard_mode;
    if (mode == KCFX_SHARD_AUTO)
        mode = (n_samples >= G || n_windows == 0) ? KCFX_SHARD_SAMPLES : (n_samples == 1 ? KCFX_SHARD_WINDOWS : KCFX_SHARD_GRID);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->shard_mode_used = mode;
        stats->n_devices = G;
    }
    if (n_samples == 0 || n_windows == 0) return KCFX_OK;
    // one database after the other on the same devices: a released table stays for the next one whatever its size (a 200 GB table
    // that went back to the driver made the next open wait for the driver's clearing, kcfx_pool.cpp); the process's own bound
    // comes back -- and what is over it goes -- when the call returns
    struct KeepLarge {
        double prev = kcfx::pool_set_fraction(0.9);
        ~KeepLarge() {
            kcfx::pool_set_fraction(prev);
            kcfx::pool_trim();
        }
    } keep_large;
    if (mode == KCFX_SHARD_TABLE) {
        const int32_t rc = screen_table_sharded(ctx->devices, kmc_prefixes, n_samples, kmer_length, text, text_len, win_begin, win_len, n_windows,
                                                min_kmer_count, out, stats);
        if (stats) stats->seconds_total = now_s() - t0;
        return rc;
    }

    std::vector<Worker> workers((size_t)G);
    {
        std::vector<std::vector<Segment>> plan = plan_segments(mode, G, n_samples, win_len, n_windows, kmer_length);
        for (int32_t g = 0; g < G; ++g) {
            workers[(size_t)g].device = ctx->devices[(size_t)g];
            workers[(size_t)g].segs = std::move(plan[(size_t)g]);
        }
    }

    // (a worker never lets an exception escape its thread: it becomes the worker's status)
    auto run_body = [&](Worker& wk) {
        if (wk.segs.empty()) return;
        auto failed = [&](int32_t rc) {
            wk.rc = rc;
            wk.error = kcfx_last_error();
        };
        // the resident reference of a window range: the text that range covers (all of it when the whole list is screened here);
        // consecutive segments of one range (the sample-sharded and window-sharded modes) share it
        struct RefHold {                                 // destroyed on every exit path
            kcfx_ref* p = nullptr;
            int32_t lo = -1, hi = -1;
            ~RefHold() { kcfx_ref_destroy(p); }
        } ref_hold;
        auto ref_for = [&](int32_t lo, int32_t hi) -> int32_t {
            if (ref_hold.p && ref_hold.lo == lo && ref_hold.hi == hi) return KCFX_OK;
            kcfx_ref_destroy(ref_hold.p);
            ref_hold.p = nullptr;
            const int32_t nw = hi - lo;
            int64_t t_lo = text_len, t_hi = 0;
            for (int32_t w = lo; w < hi; ++w) {
                t_lo = std::min(t_lo, win_begin[w]);
                t_hi = std::max(t_hi, win_begin[w] + (int64_t)win_len[w]);
            }
            if (t_hi < t_lo) t_lo = t_hi = 0;
            std::vector<int64_t> wb((size_t)nw);
            for (int32_t i = 0; i < nw; ++i) wb[(size_t)i] = win_begin[lo + i] - t_lo;
            const double t = now_s();
            // (how many of this worker's segments screen exactly this window range: one -> a text-only reference, planned inside
            // its single launch; several -> planned once for all of them)
            kcfx_ref_opts ro;
            ro.expected_screens = 0;
            ro.reserved = 0;
            int32_t uses = 0;
            for (const Segment& sg2 : wk.segs)
                if (sg2.w_lo == lo && sg2.w_hi == hi) uses += (sg2.n > 1 ? 2 : 1);      // (a cohort table that cannot be built screens its samples one by one)
            if (uses == 1) ro.expected_screens = 1;
            const int32_t rc = kcfx_ref_create_opts(wk.device, kmer_length, text + t_lo, t_hi - t_lo, wb.data(), win_len + lo, nw, &ro, &ref_hold.p);
            wk.t_ref += now_s() - t;
            ref_hold.lo = lo;
            ref_hold.hi = hi;
            return rc;
        };
        // k-mer look-ups a segment's database will serve: one pass over its window range (sizes the table, kcfx_db_opts)
        auto seg_queries = [&](const Segment& sg) {
            int64_t q = 0;
            for (int32_t w = sg.w_lo; w < sg.w_hi; ++w) q += std::max<int64_t>(0, (int64_t)win_len[w] - kmer_length + 1);
            return std::max<int64_t>(1, q);
        };
        // The loaders open one database (or cohort table) per segment, in segment order, while at most MAX_RESIDENT tables exist on
        // the device: the one being screened + the ones ready + the ones being built (counted when a loader CLAIMS a segment, so
        // that the table sizing of kcfx_db_open -- 60 % / 80 % of the free memory -- never meets a third table). ONE loader by default: a load is a file streamed from the page cache through pinned buffers (~40 ms for
        // 1.2 GB: ~30 GB/s whatever the number of reader threads) followed by the build passes on the GPU (~22 ms), and two loads
        // in flight take exactly twice as long each -- measured, profiles/r04_notes.md: the host-side copy is the shared resource,
        // not the GPU (making the files take turns on that path changes nothing either: profiles/r04_notes.md). KCFX_SHARD_LOADERS=
        // 1 .. 4 keeps the experiment repeatable. Cohort tables are opened one at a time
        // (their vote table alone can take tens of GiB).
        std::mutex mu, cohort_mu;
        std::condition_variable cv;
        struct Loaded {                                  // what a loader hands over for a segment
            kcfx_db* db = nullptr;                       // a sample's table, or
            kcfx_cohort* co = nullptr;                   // a group's cohort table, or
            bool one_by_one = false;                     // neither: the cohort table could not be built -- screen the group's samples singly
            std::string why;
        };
        std::map<size_t, Loaded> ready;                  // by segment index
        const size_t n_segs = wk.segs.size();
        size_t next_load = 0, next_screen = 0;           // the next segment a loader claims / the screening thread waits for
        constexpr int MAX_RESIDENT = 2;
        bool stop = false;
        int in_use = 0;                                  // databases the screening thread holds (0 or 1)
        int loading = 0;                                 // segments claimed by a loader and not yet handed over
        auto resident = [&] { return in_use + (int)ready.size() + loading; };
        int loaders_alive = 0;
        int32_t load_rc = KCFX_OK;
        std::string load_error;
        auto loader_body = [&]() {
            try {
            for (;;) {
                size_t i;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || load_rc != KCFX_OK || next_load >= n_segs || resident() < MAX_RESIDENT; });
                    if (stop || load_rc != KCFX_OK || next_load >= n_segs) break;
                    i = next_load++;
                    ++loading;
                }
                const double t1 = now_s();
                kcfx_db* db = nullptr;
                Loaded ld;
                // one pass of this segment's window range per database (GetVariants.java:129-159): the table is sized for that.
                // First without the line-table fallback: a cell table that does not fit NEXT TO the databases already resident
                // (two wheat-scale tables do not always fit) is worth waiting for -- the line table screens at ~0.55x the
                // speed and its odd size defeats the block pool. When this segment is the next to be screened and nothing is
                // being screened, it is opened again with the fallback allowed.
                kcfx_db_opts opts;
                opts.expected_queries = seg_queries(wk.segs[i]);
                opts.no_fallback = 1;
                opts.reserved = 0;
                const char* prefix = kmc_prefixes[wk.segs[i].sample];
                auto my_turn = [&](std::unique_lock<std::mutex>& lk) {
                    cv.wait(lk, [&] { return stop || (in_use == 0 && next_screen == i); });
                    return !stop;
                };
                int32_t rc;
                if (wk.segs[i].n > 1) {
                    // a cohort table for the group (key table + 16 bytes of counters per cell: ~3 x a single table of the union).
                    // If the device cannot hold it beside what is resident: once more when it is this segment's turn. If it cannot
                    // be built at all (3- / 4-byte counters, mixed strandedness, k outside 19 .. 31, too large): the group's
                    // samples are screened one by one, which is what the other modes do.
                    opts.no_fallback = 0;
                    {
                        std::lock_guard<std::mutex> one(cohort_mu);
                        rc = kcfx_cohort_open(kmc_prefixes + wk.segs[i].sample, wk.segs[i].n, wk.device, &opts, &ld.co);
                    }
                    if (rc == KCFX_ERR_DEVICE) {
                        std::unique_lock<std::mutex> lk(mu);
                        if (my_turn(lk)) {
                            lk.unlock();
                            std::lock_guard<std::mutex> one(cohort_mu);
                            rc = kcfx_cohort_open(kmc_prefixes + wk.segs[i].sample, wk.segs[i].n, wk.device, &opts, &ld.co);
                        }
                    }
                    if (rc == KCFX_ERR_DEVICE || rc == KCFX_ERR_ARG) {
                        ld.one_by_one = true;
                        ld.why = kcfx_last_error();
                        rc = KCFX_OK;
                    }
                } else {
                    rc = kcfx_db_open_opts(prefix, wk.device, &opts, &db);
                    if (rc == KCFX_ERR_DEVICE) {
                        std::unique_lock<std::mutex> lk(mu);
                        if (my_turn(lk)) {
                            lk.unlock();
                            opts.no_fallback = 0;
                            rc = kcfx_db_open_opts(prefix, wk.device, &opts, &db);
                        }
                    }
                    ld.db = db;
                }
                std::lock_guard<std::mutex> lk(mu);
                wk.t_open += now_s() - t1;
                --loading;
                if (rc) {
                    if (load_rc == KCFX_OK) {
                        load_rc = rc;
                        load_error = kcfx_last_error();
                    }
                    cv.notify_all();
                    break;
                }
                ready[i] = ld;
                cv.notify_all();
            }
            } catch (...) {
                const int32_t rc = kcfx::exception_to_status("kcfx_screen_sharded (loader)");
                std::lock_guard<std::mutex> lk(mu);
                if (load_rc == KCFX_OK) {
                    load_rc = rc;
                    try { load_error = kcfx_last_error(); } catch (...) {}
                }
            }
            std::lock_guard<std::mutex> lk(mu);
            --loaders_alive;
            cv.notify_all();
        };
        int n_loaders = 1;
        if (const char* e = kcfx::hook("KCFX_SHARD_LOADERS")) n_loaders = std::max(1, std::min(4, atoi(e)));
        n_loaders = (int)std::min<size_t>((size_t)n_loaders, std::max<size_t>(n_segs, 1));
        std::vector<std::thread> loaders;
        // on every way out of this scope (normal, error, exception): stop the loaders, join them, close what they had ready
        struct Joiner {
            std::vector<std::thread>& ts;
            std::mutex& mu;
            std::condition_variable& cv;
            bool& stop;
            std::map<size_t, Loaded>& ready;
            ~Joiner() {
                {
                    std::lock_guard<std::mutex> lk(mu);
                    stop = true;
                    cv.notify_all();
                }
                for (auto& t : ts)
                    if (t.joinable()) t.join();
                for (auto& kv : ready) {
                    kcfx_db_close(kv.second.db);
                    kcfx_cohort_close(kv.second.co);
                }
            }
        } joiner{loaders, mu, cv, stop, ready};
        loaders.reserve((size_t)n_loaders);
        {
            std::lock_guard<std::mutex> lk(mu);
            loaders_alive = n_loaders;
        }
        for (int t = 0; t < n_loaders; ++t) {
            try {
                loaders.emplace_back(loader_body);
            } catch (...) {                              // (a thread that could not be started: the others carry on)
                std::lock_guard<std::mutex> lk(mu);
                --loaders_alive;
            }
        }
        for (size_t i = 0; i < wk.segs.size() && wk.rc == KCFX_OK; ++i) {
            const Segment& sg = wk.segs[i];
            if (int32_t rc = ref_for(sg.w_lo, sg.w_hi)) {      // (beside the loader's first open)
                failed(rc);
                break;
            }
            Loaded ld;
            {
                const double t1 = now_s();
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return ready.count(i) != 0 || load_rc != KCFX_OK || loaders_alive == 0; });
                wk.t_wait += now_s() - t1;
                if (ready.count(i) == 0) {              // a loader failed (or none could be started)
                    wk.rc = load_rc ? load_rc : KCFX_ERR_DEVICE;
                    wk.error = load_rc ? load_error : "could not start a loader thread";
                    break;
                }
                ld = ready[i];
                ready.erase(i);
                in_use = 1;
                next_screen = i + 1;
                cv.notify_all();
            }
            const double t1 = now_s();
            auto screen_one = [&](kcfx_db* db, int32_t sample) {
                kcfx_db_info_t info;
                kcfx_db_info(db, &info);
                if (info.kmer_length != kmer_length) {
                    wk.rc = KCFX_ERR_ARG;
                    wk.error = "Kmer size mismatch between the KMC databases";          // KCFHeader.equals (Cohort.java:71-119)
                } else if (int32_t rc = kcfx_screen_ref(db, ref_hold.p, min_kmer_count, out + (size_t)sample * n_windows + sg.w_lo)) {
                    failed(rc);
                }
            };
            if (ld.co) {
                // one look-up per reference k-mer for the whole group; the table's rows go to the group's rows of `out`
                kcfx_cohort_info_t ci;
                kcfx_cohort_info(ld.co, &ci);
                const size_t nw = (size_t)(sg.w_hi - sg.w_lo);
                std::vector<kcfx_result> rows((size_t)sg.n * nw);
                if (ci.kmer_length != kmer_length) {
                    wk.rc = KCFX_ERR_ARG;
                    wk.error = "Kmer size mismatch between the KMC databases";
                } else if (int32_t rc = kcfx_cohort_screen_ref(ld.co, ref_hold.p, min_kmer_count, rows.data())) {
                    failed(rc);
                } else {
                    for (int32_t j = 0; j < sg.n; ++j)
                        memcpy(out + (size_t)(sg.sample + j) * n_windows + sg.w_lo, rows.data() + (size_t)j * nw, nw * sizeof(kcfx_result));
                }
                wk.t_screen += now_s() - t1;
                kcfx_cohort_close(ld.co);
            } else if (ld.one_by_one) {
                if (getenv("KCFX_VERBOSE")) fprintf(stderr, "[kcfx] no cohort table for samples %d .. %d (%s): screened one by one\n", sg.sample, sg.sample + sg.n - 1, ld.why.c_str());
                for (int32_t j = 0; j < sg.n && wk.rc == KCFX_OK; ++j) {
                    kcfx_db_opts opts;
                    opts.expected_queries = seg_queries(sg);
                    opts.no_fallback = 0;
                    opts.reserved = 0;
                    kcfx_db* db = nullptr;
                    const double t2 = now_s();
                    if (int32_t rc = kcfx_db_open_opts(kmc_prefixes[sg.sample + j], wk.device, &opts, &db)) {
                        failed(rc);
                        break;
                    }
                    wk.t_open += now_s() - t2;
                    wk.t_wait += now_s() - t2;
                    screen_one(db, sg.sample + j);
                    kcfx::db_close_idle(db);
                }
                wk.t_screen += now_s() - t1;
            } else {
                screen_one(ld.db, sg.sample);
                wk.t_screen += now_s() - t1;
                kcfx::db_close_idle(ld.db);               // (kcfx_screen_ref has returned: nothing in flight reads this table)
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                in_use = 0;
                cv.notify_all();
            }
        }
    };
    auto run = [&](Worker& wk) {
        try {
            run_body(wk);
        } catch (...) {
            wk.rc = kcfx::exception_to_status("kcfx_screen_sharded (worker)");
            try { wk.error = kcfx_last_error(); } catch (...) {}
        }
    };

    if (G == 1) {
        run(workers[0]);
    } else {
        std::vector<std::thread> th;
        th.reserve((size_t)G);
        try {
            for (int32_t g = 0; g < G; ++g) th.emplace_back(run, std::ref(workers[(size_t)g]));
        } catch (...) {                                  // a thread could not be started: its share is not screened
            workers[th.size()].rc = kcfx::exception_to_status("kcfx_screen_sharded (thread start)");
            workers[th.size()].error = "could not start a worker thread";
        }
        for (auto& x : th) x.join();
    }
    if (stats) {
        stats->seconds_total = now_s() - t0;
        for (const Worker& wk : workers) {
            stats->seconds_ref = std::max(stats->seconds_ref, wk.t_ref);
            stats->seconds_db_open_sum += wk.t_open;
            stats->seconds_screen_sum += wk.t_screen;
            stats->seconds_exposed_load = std::max(stats->seconds_exposed_load, wk.t_wait);
        }
    }
    for (const Worker& wk : workers)
        if (wk.rc != KCFX_OK) return fail(wk.rc, "device " + std::to_string(wk.device) + ": " + wk.error);
    return KCFX_OK;
} catch (...) { return kcfx::exception_to_status("kcfx_screen_sharded"); }

}  // extern "C"
