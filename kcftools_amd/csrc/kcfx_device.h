// kcfx_device.h -- device-side building blocks of the getVariations engine (gfx950, wave64).
//
// Everything here is __host__ __device__ where it does not need a GPU builtin, so that the bit-level logic
// (base packing, k-mer extraction, gap accounting) can be unit-tested lane by lane on the CPU; the product
// path only ever runs the __global__ kernels of kcfx_kernels.hip.
#pragma once

#include "kcfx_internal.h"

namespace kcfx {

#define KCFX_HD __host__ __device__ __forceinline__

// ---- hashing ----------------------------------------------------------------------------------------
// 64-bit left-aligned k-mer word -> 32 bits. hi*odd ^ lo is a bijection in lo for fixed hi, then murmur3's
// fmix32 (a bijection on 32 bits) spreads it; 3 x v_mul_lo_u32, no 64-bit multiply.
KCFX_HD uint32_t hash32(uint64_t key) {
    uint32_t lo = (uint32_t)key, hi = (uint32_t)(key >> 32);
    uint32_t h = (hi * 0x9E3779B1u) ^ lo;
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
KCFX_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
KCFX_HD uint32_t home_line(uint64_t key, uint32_t n_lines) { return mulhi32(hash32(key), n_lines); }

// ---- base packing ------------------------------------------------------------------------------------
// funnel shift left: high 32 bits of ((hi:lo) << s), s in 0..31 (lowers to v_alignbit_b32)
KCFX_HD uint32_t fshl(uint32_t hi, uint32_t lo, uint32_t s) {
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) << (s & 31) >> 32);
}

KCFX_HD uint32_t byte_select_acgt(uint32_t codes) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_perm_b32: selector bytes 0..3 pick bytes of the second source
    return __builtin_amdgcn_perm(0u, 0x54474341u, codes);
#else
    uint32_t r = 0;
    for (int b = 0; b < 4; ++b) r |= ((0x54474341u >> (8 * ((codes >> (8 * b)) & 3))) & 0xFFu) << (8 * b);
    return r;
#endif
}

// 4 ASCII bases (little-endian dword, byte 0 first) -> 8 code bits (first base in bits 7:6; A=0 C=1 G=2 T=3,
// Kmer.java:286-294) and 4 validity bits (first base in bit 3). Valid = upper-cased char in {A,C,G,T}
// (Fasta.java:98,132-134): soft-masked bases are valid, N and IUPAC codes are not.
KCFX_HD void pack4(uint32_t x, uint32_t& code8, uint32_t& valid4) {
    // code = ((c >> 1) & 3) ^ ((c >> 2) & 1): A(0x41)->0 C(0x43)->1 G(0x47)->2 T(0x54)->3, case-insensitive
    uint32_t cb = ((x >> 1) & 0x03030303u) ^ ((x >> 2) & 0x01010101u);
    code8 = (cb * 0x40100401u) >> 24;                 // gather byte j's 2 bits to bits [31-2j, 30-2j]
    uint32_t up = x & 0xDFDFDFDFu;                    // upper-case
    uint32_t eq = up ^ byte_select_acgt(cb);          // zero byte <=> the char is exactly the letter of its code
    uint32_t nz = (((eq & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | eq) & 0x80808080u;
    uint32_t ok = (~nz & 0x80808080u) >> 7;           // bits 0, 8, 16, 24
    valid4 = ((ok * 0x08040201u) >> 24) & 0xFu;       // byte j -> bit 3-j
}

// 16 ASCII bases -> 32 code bits (MSB first) + 16 validity bits (MSB first)
KCFX_HD void pack16(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t& code32, uint32_t& valid16) {
    uint32_t c0, c1, c2, c3, v0, v1, v2, v3;
    pack4(x0, c0, v0);
    pack4(x1, c1, v1);
    pack4(x2, c2, v2);
    pack4(x3, c3, v3);
    code32 = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
    valid16 = (v0 << 12) | (v1 << 8) | (v2 << 4) | v3;
}

// code word -> word whose 64-bit bit-reversal yields reverse-complement codes:
// per 2-bit code (b1 b0) store (~b0 ~b1)
KCFX_HD uint32_t comp_swap(uint32_t f) { return ~(((f >> 1) & 0x55555555u) | ((f & 0x55555555u) << 1)); }

KCFX_HD uint32_t brev32(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bitreverse32(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return (x >> 24) | ((x >> 8) & 0xFF00u) | ((x << 8) & 0xFF0000u) | (x << 24);
#endif
}

// The k-mer starting at packed base position q (forward array fw, complement-swapped array cw, validity
// bits vb; all MSB-first). Returns validity (all k bases valid); fwd / rc are left-aligned words
// (Kmer.java:232-252, :300-338).
KCFX_HD bool extract_kmer(const uint32_t* fw, const uint32_t* cw, const uint32_t* vb, int q, int k, uint64_t& fwd,
                          uint64_t& rc) {
    const int w = q >> 4;
    const uint32_t s = (uint32_t)(q & 15) * 2;
    const uint32_t f0 = fw[w], f1 = fw[w + 1], f2 = fw[w + 2];
    const uint32_t g0 = cw[w], g1 = cw[w + 1], g2 = cw[w + 2];
    const uint64_t kmask = ~0ull << (64 - 2 * k);          // top 2k bits
    fwd = ((((uint64_t)fshl(f0, f1, s)) << 32) | fshl(f1, f2, s)) & kmask;
    const uint64_t y = ((((uint64_t)fshl(g0, g1, s)) << 32) | fshl(g1, g2, s)) & kmask;
    // bit-reverse the 64-bit word: reverses the base order and swaps the bits of every pair back
    const uint64_t r = (((uint64_t)brev32((uint32_t)y)) << 32) | brev32((uint32_t)(y >> 32));
    rc = r << (64 - 2 * k);
    const int w2 = q >> 5;
    const uint32_t v = fshl(vb[w2], vb[w2 + 1], (uint32_t)(q & 31));
    return (v >> (32 - k)) == (uint32_t)((1ull << k) - 1);
}

// ---- gap accounting over one 64-position word ----------------------------------------------------------
// H = hit positions, M = valid positions that missed (disjoint; invalid positions are in neither and are
// transparent: GetVariants.java:220-245 iterates the list of VALID k-mers only).
// Wave-uniform carry: cg = misses since the last hit (or since the range start), cany = a hit was seen.
struct GapCarry {
    int32_t cg;
    int32_t lead;
    bool cany;
};

// contribution of the hit at `lane` (must be a set bit of H): does it close an interior gap, and how long?
KCFX_HD int32_t gap_closed_by(uint64_t H, uint64_t M, int lane, const GapCarry& c) {
    const uint64_t below = (1ull << lane) - 1;
    const uint64_t hb = H & below;
    if (hb == 0) return c.cany ? c.cg + __builtin_popcountll(M & below) : 0;
    const int q = 63 - __builtin_clzll(hb);
    const uint64_t between = below & ~((2ull << q) - 1);
    return __builtin_popcountll(M & between);
}

// advance the carry past the word
KCFX_HD void gap_advance(uint64_t H, uint64_t M, GapCarry& c) {
    if (H) {
        if (!c.cany) {
            const int first = __builtin_ctzll(H);
            c.lead = c.cg + __builtin_popcountll(M & ((1ull << first) - 1));
            c.cany = true;
        }
        const int last = 63 - __builtin_clzll(H);
        c.cg = __builtin_popcountll(M & ~((2ull << last) - 1));
    } else {
        c.cg += __builtin_popcountll(M);
    }
}

}  // namespace kcfx
