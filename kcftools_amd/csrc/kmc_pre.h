// kmc_pre.h -- host-only parser of the KMC prefix file (<prefix>.kmc_pre): KMC.java:107-168 readPrefixFile, layout in
// SURVEY.md Appendix A. No HIP in this translation unit: it is built a second time with -fsanitize=address,undefined
// for the malformed-file fuzz of tests/test_sanitizers.py.
#pragma once

#include <stdint.h>
#include <string.h>

#include <memory>
#include <string>

namespace kcfx {

struct KmcHeader {
    int32_t k = 0, mode = 0, counter_size = 0, p = 0, sig_len = 0, min_count = 0, max_count = 0, both = 0, version = 0;
    int32_t n_bins = 0;
    int64_t total = 0;
    int32_t suffix_bytes = 0, record_size = 0;
    // n_bins * 4^p global record indices: a view into `file` (the whole .kmc_pre, 64 MiB for p = 7; file offset 4 is not
    // 8-byte aligned, hence the byte pointer + memcpy accessors)
    std::unique_ptr<uint8_t[]> file;
    const uint8_t* lut_bytes = nullptr;
    int64_t lut_entries = 0;
    // the LUT is not the run of prefix sums a KMC writer produces (KCFX_LENIENT_LUT=1 only; otherwise the file is refused)
    bool lut_suspect = false;
    uint64_t lut_at(int64_t i) const { uint64_t v; memcpy(&v, lut_bytes + 8 * i, 8); return v; }
};

// 0 (KCFX_OK) or KCFX_ERR_IO / KCFX_ERR_FORMAT with a message in `err`. Never throws anything but std::bad_alloc.
int32_t parse_prefix_file(const std::string& path, KmcHeader& h, std::string& err);

// pread `bytes` at `offset` into dst with a few threads
bool parallel_pread(int fd, uint8_t* dst, int64_t bytes, int64_t offset);

}  // namespace kcfx
