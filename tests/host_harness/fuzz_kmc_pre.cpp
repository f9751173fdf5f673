// Driver of the malformed-.kmc_pre fuzz (tests/test_sanitizers.py): parses every file named on the command line with the
// product's own parser (kcftools_amd/csrc/kmc_pre.cpp, built here with -fsanitize=address,undefined) and prints one status
// per file. TEST INFRASTRUCTURE ONLY. A file the parser accepts has its whole LUT read once, so that an out-of-bounds
// view (KMC.java:153-163 derives the LUT's extent from the header's own offsets) is caught by the sanitizer.
#include <cstdio>
#include <string>

#include "../../kcftools_amd/csrc/kmc_pre.h"

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        kcfx::KmcHeader h;
        std::string err;
        const int32_t rc = kcfx::parse_prefix_file(argv[i], h, err);
        uint64_t x = 0;
        if (rc == 0)
            for (int64_t j = 0; j < h.lut_entries; ++j) x ^= h.lut_at(j);
        printf("%d %d %lld %llu %s\n", rc, h.k, (long long)h.lut_entries, (unsigned long long)(x & 1), rc ? err.c_str() : "");
    }
    return 0;
}
