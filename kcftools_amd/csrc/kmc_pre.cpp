// kmc_pre.cpp -- see kmc_pre.h
#include "kmc_pre.h"

#include <fcntl.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/kcfx.h"

namespace kcfx {

static int32_t fail(std::string& err, int32_t code, const std::string& msg) {
    err = msg;
    return code;
}

static uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

// pread `bytes` at `offset` into dst with a few threads (page cache -> pinned memory is a memcpy: one thread moves
// ~5 GB/s, the .kmc_suf of a 150 M-k-mer sample is 1.2 GB)
bool parallel_pread(int fd, uint8_t* dst, int64_t bytes, int64_t offset) {
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(8, bytes >> 22));      // >= 4 MiB per thread (4 .. 32 threads measure the same:
                                                                                     // the upload over PCIe, not this copy, paces a load)
    std::atomic<bool> ok{true};
    auto work = [&](int t) {
        const int64_t lo = bytes * t / T, hi = bytes * (t + 1) / T;
        int64_t got = lo;
        while (got < hi) {
            const ssize_t r = pread(fd, dst + got, (size_t)(hi - got), offset + got);
            if (r <= 0) {
                ok = false;
                return;
            }
            got += r;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    return ok;
}

// whole file into an uninitialised buffer (no zero-fill pass)
static int32_t read_whole(const std::string& path, std::unique_ptr<uint8_t[]>& out, int64_t& size, std::string& err) {
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(err, KCFX_ERR_IO, "Error reading prefix file " + path);
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        return fail(err, KCFX_ERR_IO, "Error reading prefix file " + path);
    }
    size = (int64_t)st.st_size;
    out.reset(new uint8_t[(size_t)std::max<int64_t>(size, 1)]);
    const bool ok = parallel_pread(fd, out.get(), size, 0);
    close(fd);
    if (!ok) return fail(err, KCFX_ERR_IO, "short read on " + path);
    return KCFX_OK;
}

// KMC.java:107-168 readPrefixFile
int32_t parse_prefix_file(const std::string& path, KmcHeader& h, std::string& err) {
    int64_t size = 0;
    if (int32_t rc = read_whole(path, h.file, size, err)) return rc;
    const uint8_t* pre = h.file.get();
    if (size < 4 + 8 + 68 + 8) return fail(err, KCFX_ERR_FORMAT, "prefix file too small: " + path);
    if (memcmp(pre, "KMCP", 4) != 0 || memcmp(pre + size - 4, "KMCP", 4) != 0)
        return fail(err, KCFX_ERR_FORMAT, "missing KMCP markers in " + path);
    const int64_t header_offset = rd32(pre + size - 8);                    // :117-118
    const int64_t pos = size - header_offset - 8;                          // :121
    if (header_offset < 68 || pos < 4 + 8) return fail(err, KCFX_ERR_FORMAT, "bad header offset in " + path);
    const uint8_t* q = pre + pos;
    h.k = (int32_t)rd32(q + 0);                                            // :124
    h.mode = (int32_t)rd32(q + 4);
    h.counter_size = (int32_t)rd32(q + 8);
    h.p = (int32_t)rd32(q + 12);
    h.sig_len = (int32_t)rd32(q + 16);
    h.min_count = (int32_t)rd32(q + 20);
    h.max_count = (int32_t)rd32(q + 24);
    h.total = (int64_t)rd64(q + 28);
    h.both = (q[36] == 0) ? 1 : 0;                                         // :133 (inverted flag)
    h.version = (int32_t)rd32(q + 64);                                     // :139
    if (h.version != 0x200) return fail(err, KCFX_ERR_FORMAT, "KMC version is not 0x200");   // :140-142
    if (h.k < 1 || h.k > 32) return fail(err, KCFX_ERR_FORMAT, "unsupported k-mer length " + std::to_string(h.k) + " (1..32)");
    if (h.p < 0 || h.p > 15 || h.p > h.k || (h.k - h.p) % 4 != 0)
        return fail(err, KCFX_ERR_FORMAT, "unsupported LUT prefix length " + std::to_string(h.p) + " for k=" + std::to_string(h.k));
    if (h.counter_size < 0 || h.counter_size > 4) return fail(err, KCFX_ERR_FORMAT, "unsupported counter size");
    if (h.sig_len < 1 || h.sig_len > 14) return fail(err, KCFX_ERR_FORMAT, "unsupported signature length");
    if (h.total < 0) return fail(err, KCFX_ERR_FORMAT, "negative k-mer total");
    const int64_t map_size = (1ll << (2 * h.sig_len)) + 1;                 // :145
    const int64_t map_start = size - header_offset - 8 - map_size * 4;     // :146
    const int64_t lut_size = 1ll << (2 * h.p);                             // :154
    if (map_start < 4 + 8) return fail(err, KCFX_ERR_FORMAT, "prefix file too small for its signature map");
    h.n_bins = (int32_t)((map_start - 8 - 4) / (lut_size * 8));            // :156
    const int64_t entries = (int64_t)h.n_bins * lut_size;                  // :160
    h.lut_bytes = pre + 4;                                                 // :153,:161-163
    h.lut_entries = entries;
    // Sanity of the LUT: a KMC writer stores the prefix sums of the record counts, i.e. a non-decreasing run from 0 that
    // never points past total_kmers. decode_records_kernel finds a record's (bin, prefix) by bisection over these values, so a
    // LUT that is not monotonic would hand records arbitrary prefixes -- wrong k-mers, silently. The reference reads the
    // values without looking at them (KMC.java:153-163) and is as undefined on such a file; here the file is refused, unless
    // KCFX_LENIENT_LUT=1 asks for the reference's behaviour (the handle then carries lut_suspect).
    uint64_t prev = 0;
    bool monotonic = true;
    for (int64_t i = 0; i < entries; ++i) {
        const uint64_t v = h.lut_at(i);
        monotonic = monotonic && v >= prev;
        prev = v;
    }
    const char* why = nullptr;
    if (!monotonic) why = "LUT is not monotonic";
    else if (entries && h.lut_at(entries - 1) > (uint64_t)h.total) why = "LUT points past total_kmers";
    else if (h.total > 0 && entries > 0 && h.lut_at(0) != 0) why = "LUT does not start at record 0";
    if (why) {
        const char* lenient = getenv("KCFX_LENIENT_LUT");
        if (!(lenient && !strcmp(lenient, "1"))) return fail(err, KCFX_ERR_FORMAT, std::string(why) + " in " + path + " (KCFX_LENIENT_LUT=1 accepts it)");
        h.lut_suspect = true;
    }
    if (h.total > 0 && entries == 0) return fail(err, KCFX_ERR_FORMAT, "prefix file has no LUT");
    h.suffix_bytes = (h.k - h.p) / 4;
    h.record_size = h.counter_size + h.suffix_bytes;                       // :61
    return KCFX_OK;
}


}  // namespace kcfx
