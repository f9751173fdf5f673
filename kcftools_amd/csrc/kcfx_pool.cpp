// kcfx_pool.cpp -- device memory for the library's large buffers (tables, stashes, decoded records, reference text).
//
// A run over many samples opens and closes one database after the other on the same device, each with a table of the
// same size (2^lbits lines). hipFree / hipMalloc of 16 GiB is not free on this platform: measured on MI355X, every
// second or third hipMalloc of a 16 GiB table that follows the hipFree of the previous one takes 1.5 - 1.9 s (the driver
// hands the pages back cleared), against a 0.1 s database load. So large blocks that are released are kept, per device, and
// handed out again when a request of (about) the same size comes; everything the library clears or overwrites before use
// anyway. Bounded (pool_max_blocks, pool_max_fraction of the device), emptied when an allocation fails, released on
// request (kcfx_release_cached), disabled by KCFX_NO_POOL, bounded lower by KCFX_POOL_FRACTION. Memory policies that look at the free device memory count the
// kept blocks as free (free_device_bytes).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "kcfx_internal.h"

namespace kcfx {

const char* hook(const char* name) {
    static const bool on = getenv("KCFX_TEST_HOOKS") != nullptr;
    return on ? getenv(name) : nullptr;
}

namespace {

size_t pool_min_bytes() {                               // smaller requests go straight to hipMalloc (KCFX_POOL_MIN_KB: experiments)
    static const size_t n = [] {
        const char* e = hook("KCFX_POOL_MIN_KB");
        return e && atoi(e) > 0 ? (size_t)atoi(e) << 10 : (size_t)8 << 20;
    }();
    return n;
}
size_t pool_max_blocks() {                              // kept blocks per device (KCFX_POOL_BLOCKS: experiments)
    static const size_t n = [] {
        const char* e = hook("KCFX_POOL_BLOCKS");
        return e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)16;
    }();
    return n;
}
// of the device's memory. Round 5 raised the bound from 0.4 to 0.9 for every process: a table of 206 GB -- an 8 G-k-mer database --
// went back to the driver on close, the next open of the run got the same pages back at once, and its first pass over the file
// then crept along for six seconds with the GPU idle and the uploads stuck in hipMemcpyAsync -- the driver clears released memory in
// the background on the engines the uploads use (kept, the second open of such a database takes 7 s instead of 12.8). That is the
// right bound for a job that opens one large database after the other and a hostile default for a library loaded into a host that
// shares the device: round 6 -- half of the device by default, kcfx_set_cache_fraction for the jobs that want more
// (kcfx_screen_sharded raises it to 0.9 for the length of its call and trims when it returns). What is kept is released when any
// allocation of the library fails and counts as free memory in every policy.
std::atomic<double> g_fraction{-1.0};
double pool_max_fraction() {                            // (KCFX_POOL_FRACTION=<0..1>: the process-wide default)
    const double now = g_fraction.load();
    if (now >= 0.0) return now;
    static const double f = [] {
        const char* e = getenv("KCFX_POOL_FRACTION");
        const double v = e ? atof(e) : 0.5;
        return v >= 0.0 && v <= 1.0 ? v : 0.5;
    }();
    return f;
}

struct Block {
    void* p;
    size_t bytes;
    int dev;
};

std::mutex g_mu;
std::unordered_map<void*, Block> g_live;                // large blocks in use
std::vector<Block> g_spare;                             // large blocks kept (oldest first)

bool pool_on() {
    static const bool on = getenv("KCFX_NO_POOL") == nullptr;
    return on;
}

struct OnDevice {
    int prev = -1;
    explicit OnDevice(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(dev);
    }
    ~OnDevice() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// (the device is idle with respect to the block: the callers release after their stream is done, and hipFree -- which this
// replaces -- synchronises the whole device anyway)
void really_free(const Block& b) {
    OnDevice g(b.dev);
    if (getenv("KCFX_POOL_TRACE")) fprintf(stderr, "[kcfx pool] hipFree of %zu bytes\n", b.bytes);
    (void)hipFree(b.p);
}

size_t pooled_locked(int dev) {
    size_t s = 0;
    for (const Block& b : g_spare)
        if (b.dev == dev) s += b.bytes;
    return s;
}

}  // namespace

double pool_set_fraction(double f) {
    const double prev = pool_max_fraction();
    if (f >= 0.0 && f <= 1.0) g_fraction.store(f);
    return prev;
}

// the oldest kept blocks of every device go until each device is within the bound again (after the bound was lowered)
void pool_trim() {
    std::vector<Block> out;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        std::vector<int> devs;
        for (const Block& b : g_spare)
            if (std::find(devs.begin(), devs.end(), b.dev) == devs.end()) devs.push_back(b.dev);
        for (int dev : devs) {
            size_t total_b = 0, free_b = 0;
            {
                OnDevice g(dev);
                if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)256 << 30;
            }
            while ((double)pooled_locked(dev) > pool_max_fraction() * (double)total_b) {
                bool dropped = false;
                for (size_t i = 0; i < g_spare.size(); ++i)
                    if (g_spare[i].dev == dev) {
                        out.push_back(g_spare[i]);
                        g_spare.erase(g_spare.begin() + (long)i);
                        dropped = true;
                        break;
                    }
                if (!dropped) break;
            }
        }
    }
    for (const Block& b : out) really_free(b);
}

size_t pooled_device_bytes(int dev) {
    std::lock_guard<std::mutex> lk(g_mu);
    return pooled_locked(dev);
}

void pool_release(int dev) {
    std::vector<Block> out;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_spare.size();) {
            if (dev < 0 || g_spare[i].dev == dev) {
                out.push_back(g_spare[i]);
                g_spare.erase(g_spare.begin() + (long)i);
            } else {
                ++i;
            }
        }
    }
    for (const Block& b : out) really_free(b);
}

size_t free_device_bytes() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return (size_t)256 << 30;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return free_b;
    return free_b + pooled_device_bytes(dev);
}

hipError_t dev_alloc(void** p, size_t bytes) {
    *p = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    const bool big = pool_on() && bytes >= pool_min_bytes();
    if (big) {
        std::lock_guard<std::mutex> lk(g_mu);
        long best = -1;                                   // smallest kept block that fits without wasting a quarter
        for (size_t i = 0; i < g_spare.size(); ++i) {
            const Block& b = g_spare[i];
            if (b.dev == dev && b.bytes >= bytes && b.bytes - bytes <= bytes / 4 && (best < 0 || b.bytes < g_spare[(size_t)best].bytes))
                best = (long)i;
        }
        if (best >= 0) {
            const Block b = g_spare[(size_t)best];
            g_spare.erase(g_spare.begin() + best);
            g_live[b.p] = b;
            *p = b.p;
            return hipSuccess;
        }
    }
    if (getenv("KCFX_POOL_TRACE")) fprintf(stderr, "[kcfx pool] hipMalloc of %zu bytes\n", bytes);
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipErrorOutOfMemory && pooled_device_bytes(dev) > 0) {
        (void)hipGetLastError();
        pool_release(dev);                               // what is kept must never be the reason for a failure
        e = hipMalloc(p, bytes);
    }
    if (e == hipSuccess && big) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_live[*p] = Block{*p, bytes, dev};
    }
    return e;
}

// idle = the caller has already waited for everything that used the block (the loader's temporaries live on ONE stream, which
// it synchronises before it frees them): no device-wide synchronisation, which would wait for the OTHER worker's screening
// kernel and for queued uploads of the next database (kcfx_screen_sharded overlaps exactly those)
static hipError_t dev_free_impl(void* p, bool idle);
hipError_t dev_free(void* p) { return dev_free_impl(p, false); }
hipError_t dev_free_idle(void* p) { return dev_free_impl(p, true); }
static hipError_t dev_free_impl(void* p, bool idle) {
    if (!p) return hipSuccess;
    Block b{nullptr, 0, 0};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_live.find(p);
        if (it != g_live.end()) {
            b = it->second;
            g_live.erase(it);
        }
    }
    if (!b.p) {                                          // a small block
        if (getenv("KCFX_POOL_TRACE")) fprintf(stderr, "[kcfx pool] hipFree of a small block\n");
        return hipFree(p);
    }
    if (!idle) {
        OnDevice g(b.dev);
        (void)hipDeviceSynchronize();                    // as hipFree would have: nothing queued may still touch the block
    }
    size_t total_b = 0, free_b = 0;
    {
        OnDevice g(b.dev);
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)256 << 30;
    }
    std::vector<Block> out;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_spare.push_back(b);
        // over the bounds: the oldest blocks of this device go
        for (;;) {
            size_t n = 0;
            for (const Block& s : g_spare) n += s.dev == b.dev ? 1 : 0;
            if (n <= pool_max_blocks() && (double)pooled_locked(b.dev) <= pool_max_fraction() * (double)total_b) break;
            bool dropped = false;
            for (size_t i = 0; i < g_spare.size(); ++i)
                if (g_spare[i].dev == b.dev) {
                    out.push_back(g_spare[i]);
                    g_spare.erase(g_spare.begin() + (long)i);
                    dropped = true;
                    break;
                }
            if (!dropped) break;
        }
    }
    for (const Block& s : out) really_free(s);
    return hipSuccess;
}

// Pinned host staging buffers of the database loader (two per load, a few tens of MB): hipHostMalloc / hipHostFree cost
// milliseconds each, so up to four released ones are kept for the next load (any device).
namespace {
struct HostBlock {
    void* p;
    size_t bytes;
};
std::vector<HostBlock> g_host_spare;
std::unordered_map<void*, size_t> g_host_live;
}  // namespace

hipError_t host_alloc(void** p, size_t bytes) {
    *p = nullptr;
    if (pool_on()) {
        std::lock_guard<std::mutex> lk(g_mu);
        for (size_t i = 0; i < g_host_spare.size(); ++i)
            if (g_host_spare[i].bytes >= bytes && g_host_spare[i].bytes <= 2 * bytes) {
                *p = g_host_spare[i].p;
                g_host_live[*p] = g_host_spare[i].bytes;
                g_host_spare.erase(g_host_spare.begin() + (long)i);
                return hipSuccess;
            }
    }
    const hipError_t e = hipHostMalloc(p, bytes, hipHostMallocDefault);
    if (e == hipSuccess && pool_on()) {
        std::lock_guard<std::mutex> lk(g_mu);
        g_host_live[*p] = bytes;
    }
    return e;
}

hipError_t host_free(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_host_live.find(p);
        if (it != g_host_live.end()) {
            const size_t bytes = it->second;
            g_host_live.erase(it);
            if (g_host_spare.size() < 4) {
                g_host_spare.push_back(HostBlock{p, bytes});
                return hipSuccess;
            }
        }
    }
    if (getenv("KCFX_POOL_TRACE")) fprintf(stderr, "[kcfx pool] hipHostFree\n");
    return hipHostFree(p);
}

void host_pool_release() {
    std::vector<HostBlock> out;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        out.swap(g_host_spare);
    }
    for (const HostBlock& b : out) (void)hipHostFree(b.p);
}

}  // namespace kcfx
