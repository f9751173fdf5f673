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
// The minimum of ~16 hashes is skewed towards 0: one odd multiply (Fibonacci hashing, a bijection mod 2^32) re-spreads
// it before range reduction. screen_* is VALU-issue bound, so every instruction here is paid per k-mer.
KCFX_HD uint32_t remix32(uint32_t x) { return x * 0x9E3779B1u; }
// line of a minimizer (minimum M-mer hash)
KCFX_HD uint32_t line_of_minimizer(uint32_t mn, uint32_t n_lines) { return mulhi32(remix32(mn), n_lines); }

// M-mer hash: two multiplies. Only its ORDER matters (minimum selection), plus the spread remix32 gives the winner;
// the added constant keeps poly-A (word 0) from hashing to 0 and winning every window.
KCFX_HD uint32_t mhash32(uint64_t word) {
    uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
    uint32_t h = ((hi + 0x7F4A7C15u) * 0x9E3779B1u) ^ lo;
    h ^= h >> 15;
    h *= 0x85EBCA6Bu;
    return h;
}

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

// The codes alone plus ONE word that is zero iff all four characters are A/C/G/T in either case (the per-base validity bits
// cost 9 of pack4's 19 instructions, and all but a few chunks of a genome are clean): pack16_codes for everybody, pack16
// only for the chunks whose word is not zero.
KCFX_HD void pack4_codes(uint32_t x, uint32_t& code8, uint32_t& dirt) {
    const uint32_t cb = ((x >> 1) & 0x03030303u) ^ ((x >> 2) & 0x01010101u);
    code8 = (cb * 0x40100401u) >> 24;
    dirt = (x & 0xDFDFDFDFu) ^ byte_select_acgt(cb);
}
KCFX_HD void pack16_codes(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3, uint32_t& code32, uint32_t& dirt) {
    uint32_t c0, c1, c2, c3, d0, d1, d2, d3;
    pack4_codes(x0, c0, d0);
    pack4_codes(x1, c1, d1);
    pack4_codes(x2, c2, d2);
    pack4_codes(x3, c3, d3);
    code32 = (c0 << 24) | (c1 << 16) | (c2 << 8) | c3;
    dirt = d0 | d1 | d2 | d3;
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

// 32 bits of an MSB-first bit stream starting at bit `bitpos` (>= 0). The stream's array must have ONE readable word in
// front of index 0: the two words are chosen so that the shift lands in 0..31 and maps onto a single v_alignbit_b32
// (a left funnel shift by 0 would need its own case; a shift by 32 does not).
KCFX_HD uint32_t alignbit32(uint32_t hi, uint32_t lo, uint32_t sh) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
    return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (sh & 31));
#endif
}
KCFX_HD uint32_t bits_at(const uint32_t* a, int bitpos) {
    const int b = bitpos - 1;
    const int w = b >> 5;                                   // -1 for bitpos 0
    return alignbit32(a[w], a[w + 1], 31u - (uint32_t)(b & 31));
}

// `len` bases starting at packed base position q as left-aligned words: fwd (Kmer.java:232-252) and its reverse
// complement (Kmer.java:300-338), from the forward code array fw and the complement-swapped array cw (MSB first).
KCFX_HD void extract_pair(const uint32_t* fw, const uint32_t* cw, int q, int len, uint64_t& fwd, uint64_t& rc) {
    const uint64_t mask = ~0ull << (64 - 2 * len);          // top 2*len bits
    fwd = ((((uint64_t)bits_at(fw, 2 * q)) << 32) | bits_at(fw, 2 * q + 32)) & mask;
    const uint64_t y = ((((uint64_t)bits_at(cw, 2 * q)) << 32) | bits_at(cw, 2 * q + 32)) & mask;
    // bit-reverse the 64-bit word: reverses the base order and swaps the bits of every pair back
    const uint64_t r = (((uint64_t)brev32((uint32_t)y)) << 32) | brev32((uint32_t)(y >> 32));
    rc = r << (64 - 2 * len);
}

// all k bases starting at q are A/C/G/T (validity bits vb, MSB first)
KCFX_HD bool kmer_valid(const uint32_t* vb, int q, int k) {
    return (bits_at(vb, q) >> (32 - k)) == (uint32_t)((1ull << k) - 1);
}

KCFX_HD bool extract_kmer(const uint32_t* fw, const uint32_t* cw, const uint32_t* vb, int q, int k, uint64_t& fwd,
                          uint64_t& rc) {
    extract_pair(fw, cw, q, k, fwd, rc);
    return kmer_valid(vb, q, k);
}

// ---- minimizers ---------------------------------------------------------------------------------------
// reverse complement of a left-aligned word of `len` bases (same bit trick as extract_pair)
KCFX_HD uint64_t revcomp_word(uint64_t x, int len) {
    const uint64_t z = ~(((x >> 1) & 0x5555555555555555ull) | ((x & 0x5555555555555555ull) << 1));
    const uint64_t y = z & (~0ull << (64 - 2 * len));
    const uint64_t r = (((uint64_t)brev32((uint32_t)y)) << 32) | brev32((uint32_t)(y >> 32));
    return r << (64 - 2 * len);
}

// hash of the canonical M-mer given its two strands as left-aligned words
KCFX_HD uint32_t mmer_hash(uint64_t fwd, uint64_t rc) { return mhash32(fwd < rc ? fwd : rc); }

// Build side: minimizer of a stored k-mer word = min over its k-M+1 M-mers of mmer_hash. Strand-symmetric, so the
// canonical and the forward form of a k-mer agree, and it equals what screen_tiles derives from the reference text
// (min over s_h[p .. p+k-M]).
KCFX_HD uint32_t minimizer_of_kmer(uint64_t x, int k, int M) {
    const uint64_t y = revcomp_word(x, k);
    const uint64_t mask = ~0ull << (64 - 2 * M);
    uint32_t best = 0xFFFFFFFFu;
    for (int j = 0; j + M <= k; ++j) {
        const uint64_t f = (x << (2 * j)) & mask;
        const uint64_t r = (y << (2 * (k - M - j))) & mask;
        const uint32_t h = mmer_hash(f, r);
        best = h < best ? h : best;
    }
    return best;
}

KCFX_HD uint32_t keyhash_line(const TableView& tv, uint64_t key) { return mulhi32(hash32(key), tv.n_lines); }
KCFX_HD uint32_t minimizer_line(const TableView& tv, uint64_t key, int k) {
    return line_of_minimizer(minimizer_of_kmer(key, k, tv.mm_len), tv.n_lines);
}

// Probe step of lookup_kernel (screen_tiles has the same logic inlined around its chunk loop).
// at_home: the line just matched was the key's minimizer home. Returns true while the probe must go on; tgt is advanced.
KCFX_HD bool probe_next(const TableView& tv, bool at_home, bool found, uint32_t meta, uint64_t key, uint32_t& tgt) {
    if (found || !(meta & META_OVERFLOW)) return false;
    if (at_home && (meta & META_HEAVY))             // every k-mer homed here by minimizer lives at its key-hash line
        tgt = keyhash_line(tv, key);
    else
        tgt = (tgt + 1 == tv.n_lines) ? 0u : tgt + 1;
    return true;
}

// ---- FORMAT_CELLS (kcfx_internal.h): k = 31, M = 16 --------------------------------------------------
// ---- the ORDER that picks a k-mer's minimizer ----
// Not the plain hash: 16-mers that are OPEN SYNCMERS (the smallest of their eleven canonical 6-mers sits in the middle, position 5)
// come first, CLOSED ones (it sits at either end) second, the rest last; inside a class a hash of the 16-mer's eleven 6-mers
// decides. Two open syncmers cannot start fewer than 6 bases apart unless they share their smallest 6-mer's neighbourhood, so
// consecutive k-mers agree on their minimizer for longer: 0.0965 minimizer runs per k-mer instead of 0.1177 (= 2 / 17) on random
// text (k = 31) -- 18 % fewer table lines per reference k-mer, and the line fetches ARE the screening kernel's time.
// Round 4: the whole key is a function of ONE hash per text position -- u[p] = hash of the canonical 6-mer at p -- so that the
// planner (StripPlanner below, ref_plan_kernel) pays one multiply per position where round 3 hashed eleven 6-mers and the 16-mer
// itself for every 16-mer:
//   class   closed (1) if an end holds the minimum of the eleven (min(u0, u10) <= min(u1..u9)), else open (0) if the middle one
//           does (u5 == min(u1..u9)), else 2. Mirror-symmetric, ties included, and the 6-mers are hashed in canonical form: a
//           16-mer and its reverse complement get the same class;
//   inside  the XOR of the eleven hashes (bits 29..5): a sliding xor -- hx[p + 1] = hx[p] ^ u[p] ^ u[p + 11] -- that is symmetric
//           too. u = c6 * odd is linear over the integers, not over GF(2): the xor of eleven of them orders 16-mers as well as the
//           two-multiply hash of the 16-mer did (numpy, 3 Mb: 0.0965 runs per k-mer either way; 1.3e-5 of the windows have two
//           distinct 16-mers tied for the minimum -- every tied offset gets a copy of the k-mer, cells_visit).
// Build (cells_visit: the k-mer's k - 5 hashes once, then a key per 16-mer) and query (StripPlanner) compute the same function
// of the same canonical 6-mers and agree by construction (tests/test_kernel_logic_host.py checks every position of its texts).
constexpr uint32_t SMER6_MUL = 0x9E3779u;                         // 24 bits: v_mul_u32_u24 (the product of a 12-bit 6-mer fits 36 bits; the low 32 are kept)
KCFX_HD uint32_t smer6_hash(uint32_t x, uint32_t rcx) { return (x < rcx ? x : rcx) * SMER6_MUL; }   // distinct for distinct canonical 6-mers
// (spelled out for the device: left to itself the compiler re-associates the sliding minima of kcfx_plan.h into shared pairwise
// minima -- two v_min_u32 per element where one v_min3_u32 does -- and never forms the three-input xor)
KCFX_HD uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c;
#endif
}
KCFX_HD uint32_t xor3u(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return a ^ b ^ c;
#endif
}
constexpr uint32_t ORDER_HASH_MASK = 0x3FFFFFE0u;                 // the 25 hash bits of a key: class above, position label and strand below
constexpr uint32_t ORDER_CLOSED = 1u << 30, ORDER_REST = 2u << 30;
// class bits of a 16-mer from the hashes of its 6-mers: inner = min(u1 .. u9)
KCFX_HD uint32_t cell_class30(uint32_t u0, uint32_t u5, uint32_t u10, uint32_t inner) {
    const uint32_t ends = u0 < u10 ? u0 : u10;
    return ends <= inner ? ORDER_CLOSED : (u5 == inner ? 0u : ORDER_REST);
}
// 27-bit order key of a 16-mer, top-aligned (low 5 bits zero), from u[i] = smer6_hash of the 6-mer at base offset i, i = 0 .. 10
KCFX_HD uint32_t cell_order27(const uint32_t* u) {
    const uint32_t inner = min3u(min3u(u[1], u[2], u[3]), min3u(u[4], u[5], u[6]), min3u(u[7], u[8], u[9]));
    const uint32_t hx = u[0] ^ u[1] ^ u[2] ^ u[3] ^ u[4] ^ u[5] ^ u[6] ^ u[7] ^ u[8] ^ u[9] ^ u[10];
    return (hx & ORDER_HASH_MASK) | cell_class30(u[0], u[5], u[10], inner);
}
// The two lowest hash bits of a minimizer's order key (bits 6..5): its SLICE CLASS. A function of the 16-mer alone, the same for
// a 16-mer and its reverse complement, and both sides have it for nothing -- the build as the minimum it has just taken
// (cells_visit), the query in the byte of the tagged minimum that already carries offset and strand (StripPlanner: bits 6..5 of
// the anchor byte). Which device of a node holds a k-mer (cell_slice_of, kcfx_internal.h) starts from these two bits, so a
// screening pass against a table slice knows from the anchor bytes alone which positions cannot be its own.
KCFX_HD uint32_t order_slice_class(uint32_t key) { return (key >> 5) & 3u; }
// What a position feeds into the sliding minimum: order key | position label (bits 4..1: twice the position mod 16, distinct within
// any 16 consecutive positions) | strand bit (bit 0: the canonical form is the reverse complement of the 16-mer as the
// text reads it). The minimum therefore carries where the minimizer sits and which way round it is; with the label
// pre-doubled the query gets 2 * offset as (mn - 2 * position) & 30, the strand bit riding along untouched.
KCFX_HD uint32_t order27_tagged(uint32_t key, uint32_t fwd_m, uint32_t rc_m, uint32_t label2) {
    return key | (rc_m < fwd_m ? 1u : 0u) | label2;
}
KCFX_HD uint32_t order27_label2(uint32_t pos) { return (pos & 15u) << 1; }
// canonical minimizer 16-mer -> G32: line = top lbits, r = the rest. A bijection on 32 bits, so (line, r) identify the minimizer
// exactly, and independent of the order key: the MINIMUM of 16 keys is skewed towards 0 and keeps only ~24 bits of entropy, which
// would leave most lines of a table of 2^27 or more lines unused. ONE odd multiply (round 5; rounds 2 - 4: multiply, xorshift,
// multiply): the line is the TOP bits of the product, which every bit of the 16-mer reaches, and that is all a line number needs --
// on the bench databases the tables come out marginally better placed (stash 51.1 K k-mers against 52.6 K; k = 25: 7.8 % outside
// their first-choice cell against 8.0 %), and the word loop is three instructions shorter: 0.0855 against 0.0871 ms on the same box.
// (r, the LOW bits, depends on the 16-mer's last bases only: it identifies, it does not need to spread. Everything that hashes G32
// further -- second and third lines, extended tables, regions, slices -- has a mixer of its own.)
KCFX_HD uint32_t cell_g32(uint32_t cm) { return cm * 0x9E3779B1u; }

// Bit 15 of r: the entry does not sit in its first-choice cell. Such an entry can never be mistaken for the k-mer that has the
// same (w, r) and this cell as its first choice. Two kinds, told apart by WHERE they sit (see CellAlt and cell_second_line):
// cells W .. 15 of a line (k < 31) hold free-cell second choices, cells 0 .. W - 1 second-LINE entries.
constexpr uint32_t CELL_R_ALT = 0x8000u;

// The r field of a cell: bits 0 .. 31 - lbits carry the rest of G32 (G32' in a second-line entry), bit 15 the "not in its
// first-choice cell" mark; the lbits - 17 bits in between would be zero. In a cell that is flagged "someone who belongs here
// lives elsewhere" (its second line, the stash) they say WHO: every k-mer that was sent on sets the bit its flank selects (a
// one-hash Bloom filter per cell). A query that finds its cell taken by another k-mer looks further only if its own bit is
// set -- in a k-mer count of sequencing reads nearly every genomic cell is flagged (the error neighbours of the genomic k-mer
// want the same cell and lose to its larger counter), and every absent reference k-mer would otherwise cost a look-up.
// A forward-only database (header byte != 0, KMC.java:133; Kmer.java:57-65 then looks k-mers up as the text reads them)
// may hold a k-mer AND its reverse complement as two records: both have the same N orientation, line, cell, w and
// rest of G32, so bit 14 of r says which of the two this is (CELL_R_FLIP: the stored word is the reverse complement of
// N). Such a table needs lbits >= 18 and has one filter bit less. `both` = the database is canonical (both strands).
constexpr uint32_t CELL_R_FLIP = 0x4000u;
// THIRD LINES (round 4). A k-mer that finds its second-line cell taken as well gets one more line before the stash -- and this
// one is chosen by more than the minimizer: by the three bases of the flank next to it, on the side where the k-mer has more of
// them (cell_third_line). The losers of a run still move together (at most two groups: the k-mers with the minimizer in their
// second half share the bases before it, the others the bases after it), but DIFFERENT loci that contain the same 16-mer part
// ways here -- which the second line, a function of the minimizer alone, cannot do for them: beyond ~0.5 G distinct k-mers the
// 2^31 canonical 16-mers are shared by several loci each, and a third locus of a minimizer used to live in the stash, k-mer by
// k-mer (profiles/r03_config5: 19 - 33 % of a 2.1 G-k-mer database). Such an entry carries ALT and one more mark: the top spare
// bit of r (bit 14, forward-only tables: 13), taken from the filter bits -- tables with fewer than three spare bits have no third
// lines and send on to the stash as before.
KCFX_HD bool cell_has_third(int32_t lbits, bool both) { return lbits - cell_lbits_min(both) >= 3; }
KCFX_HD uint32_t cell_r_third(bool both) { return both ? 0x4000u : 0x2000u; }
KCFX_HD uint32_t cell_rmask(int32_t lbits, bool both) {          // the bits of r that identify
    return (0xFFFFFFFFu >> lbits) | CELL_R_ALT | (both ? 0u : CELL_R_FLIP) | (cell_has_third(lbits, both) ? cell_r_third(both) : 0u);
}
KCFX_HD uint32_t cell_spare_bits(int32_t lbits, bool both) {     // filter bits of a flagged cell
    return (uint32_t)(lbits - cell_lbits_min(both)) - (cell_has_third(lbits, both) ? 1u : 0u);
}
// Bit position inside r of the filter bit of a k-mer with flank w in a flagged cell that holds x (meaningful only when
// cell_spare_bits() > 0). Round 5: the bit is chosen by HOW THE K-MER DIFFERS FROM THE CELL'S OCCUPANT -- the xor of the two flanks,
// which cell_flank lays out relative to the minimizer, so that a base at a given distance from the minimizer is the same two bits
// whatever cell (offset) the k-mer has -- not by its own flank alone. In a database counted from reads the k-mers that lose a
// genomic k-mer's cell are its sequencing-error neighbours, and a reference k-mer that carries a variant is such a neighbour too:
// with the flank as the key every cell of the ~10 that a minimizer run around the variant visits drew its own bit (~30 % set each:
// the run was sent on to its second line with near certainty); with the difference as the key all of them ask for the SAME bit,
// which the handful of error events around that minimizer have set or not -- the whole run is sent on or none of it. Build and
// query agree because a flag is only ever keyed by an occupant that no longer changes (first-choice occupants are final before the
// settle pass, second- / third-line ones once taken); a flag on a still-EMPTY cell sets every filter bit (cell_filter_all). A
// difference in several bases, an unrelated occupant: some bit, the same on both sides. Costs the query nothing: the xor is the
// one its compare needs anyway.
KCFX_HD uint32_t cell_filter_key(uint32_t w, uint32_t x) { return (x & ~CELL_FLAG_MASK) ^ w; }
KCFX_HD uint32_t cell_filter_pos(uint32_t w, uint32_t x, int32_t lbits, bool both) {
    const uint32_t h = cell_filter_key(w, x) * 0x9E3779B1u;
    const uint32_t rb = (uint32_t)(32 - lbits);
    return rb + (uint32_t)(((uint64_t)h * cell_spare_bits(lbits, both)) >> 32);
}
// every filter bit of a cell (as a mask inside r): what flagging an EMPTY cell sets -- whoever takes the cell later is not the
// occupant the flag could have been keyed by
KCFX_HD uint32_t cell_filter_all(int32_t lbits, bool both) {
    return ((1u << cell_spare_bits(lbits, both)) - 1u) << (32 - lbits);
}
// the bits of a 64-bit cell word that identify the k-mer and carry its counter (flags and filter bits excluded)
KCFX_HD uint64_t cell_keycount_mask(int32_t lbits, bool both) {
    return 0xFFFF0000FFFFFFFCull | ((uint64_t)cell_rmask(lbits, both) << 32);
}

// Cell tables exist for k = 19 .. 31: the minimizer is always a 16-mer (32 bits: G32 is a bijection of it), of which a
// k-mer spans W = k - 15 (16 for k = 31: one DPP row, all 16 cells of a line); the flank is k - 16 <= 15 bases = at most 30 bits.
KCFX_HD constexpr int cell_mlen(int) { return 16; }
KCFX_HD constexpr int cell_nw(int k) { return k - 15; }       // 16-mers per k-mer = cells of a line a k-mer can sit in
KCFX_HD constexpr uint32_t cell_mmask(int M) { return 0xFFFFFFFFu << (32 - 2 * M); }

// reverse complement of a left-aligned M-mer (comp_swap also complements the padding; the shift drops it)
KCFX_HD uint32_t rc_mmer(uint32_t m, int M) { return brev32(comp_swap(m)) << (32 - 2 * M); }

// the k - 16 bases of N around the M-mer at base offset sh / 2, laid out RELATIVE TO THE M-MER (round 5): the bases behind it
// left-aligned from bit 31 down (the first one behind it in bits 31..30), the sh / 2 bases in front of it right-aligned above the
// two flag bits (the last one in front of it in bits 3..2), zeros between and in bits 1..0. A base at a given distance from the
// minimizer has the same two bits in every cell of a line -- what cell_filter_key relies on, and what makes "the three bases next
// to the minimizer" (cell_third_line, cell_ext_select) a constant shift. *m_out = that M-mer
KCFX_HD uint32_t cell_flank(uint64_t N, uint32_t sh, int M, uint32_t* m_out) {
    const uint64_t t = N << sh;
    *m_out = (uint32_t)(t >> 32) & cell_mmask(M);
    const uint32_t behind = M == 16 ? (uint32_t)t : (uint32_t)((t << (2 * M)) >> 32);
    const uint32_t front = (uint32_t)(N >> (62u - sh)) & ~3u;     // the top sh bits of N at bits sh + 1 .. 2
    return behind | front;
}

struct CellProbe {
    uint32_t line;      // top lbits of G32
    uint32_t sh;        // 2 * cell index (the cell is at byte 4 * sh of the line)
    uint32_t w;         // flank bits (flag bits clear)
    uint32_t r;         // low 32 - lbits bits of G32
    uint32_t cls;       // build side only (cells_visit, record_cells): the slice class of the k-mer's minimizer (order_slice_class)
};

// Where a k-mer that lost the bid for its first-choice cell may live, in the order a query looks:
//  * (only in builds with KCFX_ALT_FREE_MAX_W > 0 -- retired, see below; k <= 30 and a table with >= 4 spare r bits) in a FREE CELL of the same line: a k-mer spans W = k - 15 <= 15 minimizer
//    positions, so cells W .. 15 of every line are nobody's first choice -- which of them: the leading flank bits. Such a
//    cell does not say where its tenant came from, so the entry carries its first-choice cell index in 4 of the spare r bits
//    (never used for a filter there: only cells 0 .. W - 1 are flagged). Looked at in the main pass: same line, no request.
//  * in its SECOND LINE (every k): the same cell index in the line a second mix of the minimizer names. All the losers of a
//    minimizer move there together -- the k-mers of a locus whose minimizer 16-mer occurs at another locus too, or of a run that
//    shares its line with another run (3 % of a genome's k-mers on the bench database) -- so the queued look-ups of a run's
//    k-mers are ONE line request, where the stash costs one request per k-mer. The first-choice cell is flagged (with the
//    k-mer's filter bit) so that a query knows to look on.
//  * in the stash (then the first-choice AND the second-line cell are flagged).
struct CellAlt {
    uint32_t sh;        // 2 * index of the free cell
    uint32_t r;         // the r field an entry of this k-mer has there
    uint32_t rmask;     // the bits of r that identify it there
};
// RETIRED at the end of round 3 (KCFX_ALT_FREE_MAX_W = 0: no k uses it; 15 = every k < 31 brings it back): the free cell costs a
// second load of the line per word, which keeps the first from being non-temporal -- with second lines to fall back on, every k
// is 8 - 12 % faster without it (k = 21 / 25 / 30: 0.159 / 0.108 / 0.083 ms against 0.179 / 0.121 / 0.090; the stash grows from 0.1 to
// 0.5 - 2 % of the database, which costs less than the load).
#ifndef KCFX_ALT_FREE_MAX_W
#define KCFX_ALT_FREE_MAX_W 0
#endif
constexpr int CELL_ALT_FREE_MAX_W = KCFX_ALT_FREE_MAX_W;
KCFX_HD bool cell_alt_free(int W, int32_t lbits, bool both) { return W <= CELL_ALT_FREE_MAX_W && cell_spare_bits(lbits, both) >= 4; }
// (call only when cell_alt_free)
KCFX_HD CellAlt cell_alt_of(uint32_t sh, uint32_t w, uint32_t r, int W, int32_t lbits, bool both) {
    CellAlt a;
    const uint32_t rb = (uint32_t)(32 - lbits);
    a.sh = 2u * ((uint32_t)W + (((w >> 24) * (uint32_t)(16 - W)) >> 8));
    a.r = r | CELL_R_ALT | ((sh >> 1) << rb);
    a.rmask = cell_rmask(lbits, both) | (0xFu << rb);
    return a;
}
// G32 -> G32': another bijection of the minimizer (xorshifts and odd multiplies), so (second line, r there) identify it as
// exactly as (line, r) do
KCFX_HD uint32_t cell_g32b(uint32_t g) {
    g ^= g >> 16;
    g *= 0x7FEB352Du;
    g ^= g >> 15;
    g *= 0x846CA68Bu;
    g ^= g >> 16;
    return g;
}
// the second-line probe of a first-choice probe: same cell index and flank, line and r from G32', marked CELL_R_ALT
// (compared under cell_rmask like a first-choice entry)
KCFX_HD CellProbe cell_second_line(const CellProbe& p, int32_t lbits) {
    const uint32_t idm = 0xFFFFFFFFu >> lbits;
    const uint32_t g2 = cell_g32b((p.line << (32 - lbits)) | (p.r & idm));
    CellProbe q;
    q.line = g2 >> (32 - lbits);
    q.sh = p.sh;
    q.w = p.w;
    q.r = (g2 & idm) | CELL_R_ALT | (p.r & CELL_R_FLIP & ~idm);    // (a canonical table of 2^17 lines has a G32 bit where FLIP would be)
    q.cls = p.cls;
    return q;
}

// the third-line probe of a first-choice probe (call only when cell_has_third): line and r from a bijection of G32' that the
// flank's bases next to the minimizer select -- side (1 bit: before / after the minimizer, whichever side the k-mer has more
// bases on) and up to three bases there. (sh, w) are compared with the entry, so (line, r, sh, w) still identify the k-mer.
KCFX_HD CellProbe cell_third_line(const CellProbe& p, int32_t lbits, bool both, int W) {
    const uint32_t idm = 0xFFFFFFFFu >> lbits;
    // the flank: the bases in front of the minimizer end at bit 2, those behind it start at bit 31 (cell_flank)
    const bool before = (int)p.sh >= W;                           // offset >= W / 2: more bases in front (sh >= 4 there: W >= 4)
    const uint32_t near6 = before ? (p.w >> 2) & 63u : p.w >> 26;
    const uint32_t e = (near6 << 1) | (before ? 1u : 0u);
    const uint32_t g2 = cell_g32b((p.line << (32 - lbits)) | (p.r & idm));
    const uint32_t g3 = cell_g32b(g2 ^ ((e + 1u) * 0x9E3779B1u));
    CellProbe q;
    q.line = g3 >> (32 - lbits);
    q.sh = p.sh;
    q.w = p.w;
    q.r = (g3 & idm) | CELL_R_ALT | cell_r_third(both) | (p.r & CELL_R_FLIP & ~idm);
    q.cls = p.cls;
    return q;
}

// EXTENDED TABLES (TableView::ext; round 4, databases of >= ~0.5 G k-mers). The order puts open syncmers first, so the
// minimizers of a genome concentrate on ~1 / 11 of the 2^31 canonical 16-mers: a count of a 9 Gb genome has ~780 M runs over ~195 M
// preferred 16-mers, FOUR loci per minimizer -- and the first-choice line, a function of the minimizer alone, holds one of them:
// two thirds of such a database sat outside its first-choice cells (profiles/r04_notes.md). In an extended table the line and r
// of the FIRST choice (and with them of the second and third, which are derived from it) come from a bijection of G32 that the
// flank selects: the three bases in front of the minimizer for a k-mer that has them (offset >= 3: thirteen of the sixteen
// offsets), else the three behind it. A run costs ~1.3 line requests instead of 1, its loci no longer meet. (sh, w) are part of
// what a cell compares, so (line, r, sh, w) still identify the k-mer exactly.
KCFX_HD uint32_t cell_ext_select(uint32_t w, uint32_t sh) {
    const bool before = sh >= 6u;
    const uint32_t near6 = before ? (w >> 2) & 63u : w >> 26;
    return (((near6 << 1) | (before ? 1u : 0u)) + 1u) * 0x9E3779B1u;
}
KCFX_HD uint32_t cell_g32_ext(uint32_t g, uint32_t w, uint32_t sh) { return cell_g32b(g ^ cell_ext_select(w, sh)); }
KCFX_HD CellProbe cell_first_ext(const CellProbe& p, int32_t lbits) {
    const uint32_t idm = 0xFFFFFFFFu >> lbits;
    const uint32_t gx = cell_g32_ext((p.line << (32 - lbits)) | (p.r & idm), p.w, p.sh);
    CellProbe q = p;
    q.line = gx >> (32 - lbits);
    q.r = (gx & idm) | (p.r & ~idm);
    return q;
}

// TABLES OF R x 2^lbits LINES (TableView::regions = R; round 4). A table of 2^lbits lines doubles or halves; between 48 GiB and what
// a device holds that is the difference between 0.47 and 0.94 k-mers per cell (an 8 G-k-mer genomic database: 4 % or 18 % of it in
// the stash, 176 or 102 G k-mers/s). With R > 1 the table is R regions of 2^lbits lines: everything that identifies a k-mer --
// line = top lbits of G32, r = the rest, second and third lines, filter bits -- is as in a table of 2^lbits lines, and only the
// ADDRESS of a line gets a region in front of it, chosen by a hash of the same 32 bits (so the k-mers of a run, and the losers of
// a minimizer in their second line, still share their line). R = 3 on lbits - 1 is 1.5 x 2^lbits lines, R = 5 / 7 on lbits - 2
// are 1.25 / 1.75 x. Costs the query one multiply-high behind a wave-uniform branch.
KCFX_HD uint32_t cell_region_of(uint32_t g, uint32_t regions) {
    const uint32_t h = (g ^ (g >> 15)) * 0x2C1B3C6Du;
    return (uint32_t)(((uint64_t)(h ^ (h >> 12)) * regions) >> 32);
}
KCFX_HD uint32_t cell_line_index(uint32_t line, uint32_t r, int32_t lbits, int32_t regions) {
    if (regions <= 1) return line;
    const uint32_t g = (line << (32 - lbits)) | (r & (0xFFFFFFFFu >> lbits));
    return (cell_region_of(g, (uint32_t)regions) << lbits) | line;
}

// Query side, one lane: F / R = the k-mer as the text reads it and its reverse complement, mn = the tagged minimum over
// the k-mer's 16 M-mers, lane = this lane's position label base (order27_tagged was fed (lane + offset) & 15).
// (lane2 = 2 * position, what order27_label2 was fed -- mod 32 is enough)
KCFX_HD CellProbe cell_probe_of(uint64_t F, uint64_t R, uint32_t mn, uint32_t lane2, int32_t lbits, int M, bool both, int W = 16) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t fm;                                              // (one v_bfe_i32; hipcc rewrites the builtin into v_and + v_sub)
    asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(fm) : "v"(mn));
#else
    const uint32_t fm = 0u - (mn & 1u);                       // all ones: the minimizer is canonical on the other strand
#endif
    const bool flip = fm != 0;
    const uint64_t N = flip ? R : F;
    uint32_t sh;
    if (W == 16) {
        sh = ((mn - lane2) ^ fm) & 30u;                       // 2 * offset of the minimizer in F; in R it sits at 15 - offset
    } else {
        sh = (mn - lane2) & 30u;
        sh = flip ? 2u * (uint32_t)(W - 1) - sh : sh;         // in R it sits at W - 1 - offset
    }
    CellProbe p;
    uint32_t cm;
    p.w = cell_flank(N, sh, M, &cm);
    const uint32_t g = cell_g32(cm);
    p.sh = sh;
    p.line = g >> (32 - lbits);
    p.r = (g & (0xFFFFFFFFu >> lbits)) | ((!both && flip) ? CELL_R_FLIP : 0u);     // forward-only: F itself is looked up
    p.cls = order_slice_class(mn);
    return p;
}

// Build side: every cell a stored k-mer C (canonical in a both-strands database) must be findable in. Normally one; several when the minimum hash
// is attained at more than one offset (a repeated M-mer, or two M-mers that agree in their leading 27 hash bits --
// the query's tie-break depends on the k-mer's alignment in the text, so every tied offset gets a copy), two per offset
// when the M-mer is its own reverse complement (either strand may be the one the text reads). A forward-only database is
// asked for C exactly as stored: one orientation per tied offset (the one the query's tag bit selects: rc(m) < m), marked
// with CELL_R_FLIP when that is the reverse complement.
constexpr int CELL_MAX_CANDS = 32;
// f(const CellProbe&, int index) for every such cell, in a fixed order; returns how many. (A visitor, not an array: the
// kernels that walk a k-mer's cells keep nothing in private memory.)
// (ext: the table is an extended one -- cell_first_ext is applied to every probe)
template <class F>
KCFX_HD int cells_visit(uint64_t C, int32_t k, int32_t lbits, bool both, bool ext, F&& f) {
    const int M = cell_mlen(k), W = cell_nw(k);
    const uint64_t Rc = revcomp_word(C, k);
    // the k-mer's k - 5 canonical 6-mer hashes, once (the 6-mer at base t of C is the reverse complement of the one at base
    // k - 6 - t of Rc), then the order key of each of its W 16-mers. Static indices throughout: registers, not private memory.
    uint32_t U[26], key[16];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int t = 0; t < 26; ++t) {
        const int tr = k - 6 - t;
        U[t] = tr >= 0 ? smer6_hash((uint32_t)(C >> (52 - 2 * t)) & 0xFFFu, (uint32_t)(Rc >> (52 - 2 * tr)) & 0xFFFu) : 0xFFFFFFFFu;
    }
    uint32_t best = 0xFFFFFFFFu;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 16; ++j) {
        key[j] = 0xFFFFFFFFu;                                 // (never a key: those end in five zero bits)
        if (j < W) key[j] = cell_order27(U + j);
        best = key[j] < best ? key[j] : best;
    }
    int n = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < 16; ++j) {
        if (key[j] != best) continue;
        const uint32_t m = (uint32_t)((C << (2 * j)) >> 32) & cell_mmask(M), r = rc_mmer(m, M);
        const uint32_t cm = m < r ? m : r;
        for (int strand = 0; strand < 2; ++strand) {
            // strand 0: N = C (needs m canonical), strand 1: N = revcomp(C) (needs rc(m) canonical)
            if (strand == 0 ? (m > r) : (r > m)) continue;
            if (!both && strand == 1 && r == m) continue;     // (a palindromic M-mer: the query reads it on strand 0)
            const uint64_t N = strand ? Rc : C;
            const uint32_t sh = 2u * (uint32_t)(strand ? W - 1 - j : j);
            CellProbe p;
            uint32_t got;
            p.w = cell_flank(N, sh, M, &got);
            const uint32_t g = cell_g32(cm);
            p.sh = sh;
            p.line = g >> (32 - lbits);
            p.r = (g & (0xFFFFFFFFu >> lbits)) | ((!both && strand == 1) ? CELL_R_FLIP : 0u);
            p.cls = order_slice_class(best);
            f(ext ? cell_first_ext(p, lbits) : p, n);
            ++n;
        }
    }
    return n;
}
// the same into an array of CELL_MAX_CANDS (host-side checks)
KCFX_HD int cells_of_key(uint64_t C, int32_t k, int32_t lbits, CellProbe* out, bool both, bool ext = false) {
    return cells_visit(C, k, lbits, both, ext, [&](const CellProbe& p, int i) { out[i] = p; });
}

KCFX_HD uint32_t stash_slot(uint64_t key, uint32_t mask) { return hash32(key) & mask; }

// ---- one 128-byte line against one key ----------------------------------------------------------------
// q[0..7] = the line as eight 16-byte pieces. Empty slots hold EMPTY_KEY (all ones), which no k-mer word with
// k < 32 can equal (its low 2*(32-k) bits are zero); CHECK_NS additionally masks slots >= the occupied count and
// is needed only for k == 32, where TT..T is the all-ones word. Returns true when the key is present.
constexpr uint64_t EMPTY_KEY = ~0ull;

KCFX_HD uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Same as match_line below, reading the line from LDS piece by piece so that at most two 16-byte pieces are live
// (keeps screen_tiles at <= 64 VGPRs = 8 waves per SIMD). `src` points at the line's 8 pieces.
template <bool WIDE, bool CHECK_NS>
__device__ __forceinline__ bool match_line_lds(const uint4* src, uint64_t key, uint32_t& count, uint32_t& meta) {
    const uint4 tail = src[7];
    meta = tail.z;
    const uint32_t ns = CHECK_NS ? (meta & 0xFFu) : 0xFFu;
    uint32_t c = 0;
    bool any = false, odd = false;
    constexpr int KP = WIDE ? 5 : 6;                          // pieces that hold keys
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const uint4 p = src[i];
        const bool m0 = u64_of(p.x, p.y) == key && ns > (uint32_t)(2 * i);
        const bool m1 = u64_of(p.z, p.w) == key && ns > (uint32_t)(2 * i + 1);
        uint32_t cw;                                          // the dword(s) holding this pair's counts
        if (!WIDE) {
            const uint4 cp = (i < 4) ? src[6] : tail;
            cw = (i == 0) ? cp.x : (i == 1) ? cp.y : (i == 2) ? cp.z : (i == 3) ? cp.w : (i == 4) ? cp.x : cp.y;
            c = (m0 | m1) ? cw : c;
            odd |= m1;
        } else {
            const uint4 cp = (i < 2) ? src[5] : (i < 4) ? src[6] : tail;
            const uint32_t c0 = (i & 1) ? cp.z : cp.x, c1 = (i & 1) ? cp.w : cp.y;
            c = m0 ? c0 : c;
            c = m1 ? c1 : c;
        }
        any |= m0 | m1;
    }
    if (!WIDE) c = odd ? (c >> 16) : (c & 0xFFFFu);
    count = c;
    return any;
}

template <bool WIDE, bool CHECK_NS>
KCFX_HD bool match_line(const uint4 (&q)[8], uint64_t key, uint32_t& count, uint32_t& meta) {
    meta = q[7].z;
    const uint32_t ns = CHECK_NS ? (meta & 0xFFu) : 0xFFu;
    const bool m0 = u64_of(q[0].x, q[0].y) == key && ns > 0, m1 = u64_of(q[0].z, q[0].w) == key && ns > 1;
    const bool m2 = u64_of(q[1].x, q[1].y) == key && ns > 2, m3 = u64_of(q[1].z, q[1].w) == key && ns > 3;
    const bool m4 = u64_of(q[2].x, q[2].y) == key && ns > 4, m5 = u64_of(q[2].z, q[2].w) == key && ns > 5;
    const bool m6 = u64_of(q[3].x, q[3].y) == key && ns > 6, m7 = u64_of(q[3].z, q[3].w) == key && ns > 7;
    const bool m8 = u64_of(q[4].x, q[4].y) == key && ns > 8, m9 = u64_of(q[4].z, q[4].w) == key && ns > 9;
    uint32_t c = 0;
    bool any;
    if (!WIDE) {
        const bool m10 = u64_of(q[5].x, q[5].y) == key && ns > 10, m11 = u64_of(q[5].z, q[5].w) == key && ns > 11;
        // at most one slot matches: pick the dword holding its u16 count, then the half
        c = (m0 | m1) ? q[6].x : c;
        c = (m2 | m3) ? q[6].y : c;
        c = (m4 | m5) ? q[6].z : c;
        c = (m6 | m7) ? q[6].w : c;
        c = (m8 | m9) ? q[7].x : c;
        c = (m10 | m11) ? q[7].y : c;
        const bool odd = m1 | m3 | m5 | m7 | m9 | m11;
        c = odd ? (c >> 16) : (c & 0xFFFFu);
        any = m0 | m1 | m2 | m3 | m4 | m5 | m6 | m7 | m8 | m9 | m10 | m11;
    } else {
        c = m0 ? q[5].x : c;
        c = m1 ? q[5].y : c;
        c = m2 ? q[5].z : c;
        c = m3 ? q[5].w : c;
        c = m4 ? q[6].x : c;
        c = m5 ? q[6].y : c;
        c = m6 ? q[6].z : c;
        c = m7 ? q[6].w : c;
        c = m8 ? q[7].x : c;
        c = m9 ? q[7].y : c;
        any = m0 | m1 | m2 | m3 | m4 | m5 | m6 | m7 | m8 | m9;
    }
    count = c;
    return any;
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
// (lane-parallel form; kept for the CPU emulation tests as the independent cross-check of gap_close_scalar)
KCFX_HD int32_t gap_closed_by(uint64_t H, uint64_t M, int lane, const GapCarry& c) {
    const uint64_t below = (1ull << lane) - 1;
    const uint64_t hb = H & below;
    if (hb == 0) return c.cany ? c.cg + __builtin_popcountll(M & below) : 0;
    const int q = 63 - __builtin_clzll(hb);
    const uint64_t between = below & ~((2ull << q) - 1);
    return __builtin_popcountll(M & between);
}

// Wave-uniform form used by the kernel: all arguments are ballot masks / scalars, so this runs on the scalar unit.
// V = valid positions. A hit closes a gap iff the nearest lower VALID position (looking through invalid ones, and
// through the word boundary via the carry) is a miss. Those hits are found bit-parallel with an add that ripples a
// "previous valid is a miss" bit through each run of invalid positions; the loop then runs once per closed gap
// (typically 0-2 per word), adding interior gaps to var / inner (GetVariants.java:228-238).
KCFX_HD void gap_close_scalar(uint64_t V, uint64_t H, uint64_t M, const GapCarry& c, int32_t k, int32_t& var_in,
                              int32_t& inner_in) {
    const uint64_t T = ~V;
    const uint64_t B = (M << 1) | (c.cg > 0 ? 1ull : 0ull);
    const uint64_t S = (B & V) | (((B & T) + T) & V);
    uint64_t E = H & S;
    uint64_t prev = 0;            // positions up to and including the previous gap-closing hit
    bool use_carry = true;
    while (E) {
        const int e = __builtin_ctzll(E);
        const uint64_t below = (1ull << e) - 1;
        const int32_t g = __builtin_popcountll(M & below & ~prev) + (use_carry ? c.cg : 0);
        const bool interior = c.cany || (H & below) != 0;
        if (interior && g > 0) {
            var_in += 1;
            inner_in += gap_distance(k, g);
        }
        prev = below | (1ull << e);
        use_carry = false;
        E &= E - 1;
    }
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

// ---- a tile's hit bitmap -> its summary -----------------------------------------------------------------
// Vw / Hw: per 64-position word, the ballots of "valid k-mer" and "observed" (hit) written by screen_tiles.
// Sequential over the words of one tile (<= TILE_K / 64); reduce_windows_kernel calls it on 4-word pieces.
KCFX_HD TileSummary summarize_words(const uint64_t* Vw, const uint64_t* Hw, int n_kmers, int k, int64_t count_sum) {
    const int n_words = (n_kmers + 63) >> 6;
    if (n_words == 0) return summary_identity();
    GapCarry carry{0, 0, false};
    int32_t n_valid = 0, obs = 0, runs = 0, var_in = 0, inner_in = 0;
    uint64_t prev_valid = 0, V = 0;
    for (int w = 0; w < n_words; ++w) {
        V = Vw[w];
        const uint64_t H = Hw[w] & V;
        const uint64_t M = V & ~H;
        n_valid += __builtin_popcountll(V);
        obs += __builtin_popcountll(H);
        runs += __builtin_popcountll(V & ~((V << 1) | prev_valid));     // maximal runs of valid start positions
        prev_valid = V >> 63;
        gap_close_scalar(V, H, M, carry, k, var_in, inner_in);          // GetVariants.java:228-244
        gap_advance(H, M, carry);
    }
    TileSummary s;
    s.n = n_valid;
    s.obs = obs;
    s.lead = carry.cany ? carry.lead : n_valid;
    s.trail = carry.cany ? carry.cg : n_valid;
    s.var_in = var_in;
    s.inner_in = inner_in;
    s.runs = runs;
    s.flags = (carry.cany ? F_ANY : 0u) | ((Vw[0] & 1) ? F_FIRST : 0u) | (((V >> ((n_kmers - 1) & 63)) & 1) ? F_LAST : 0u);
    s.count_sum = count_sum;
    return s;
}

// The same over PW words that are already in registers (reduce_windows_kernel requests a piece's 2 x PW ballot words, its tile
// descriptor and its count sum in ONE round trip instead of walking memory word by word: the kernel is a chain of dependent
// loads, not of instructions). Fully unrolled: the arrays are never indexed dynamically.
template <int PW>
KCFX_HD TileSummary summarize_piece(const uint64_t (&Vw)[PW], const uint64_t (&Hw)[PW], int n_kmers, int k, int64_t count_sum) {
    if (n_kmers <= 0) return summary_identity();
    GapCarry carry{0, 0, false};
    int32_t n_valid = 0, obs = 0, runs = 0, var_in = 0, inner_in = 0;
    uint64_t prev_valid = 0, V = 0;
#pragma unroll
    for (int w = 0; w < PW; ++w) {
        if (w * 64 < n_kmers) {
            V = Vw[w];
            const uint64_t H = Hw[w] & V;
            const uint64_t M = V & ~H;
            n_valid += __builtin_popcountll(V);
            obs += __builtin_popcountll(H);
            runs += __builtin_popcountll(V & ~((V << 1) | prev_valid));
            prev_valid = V >> 63;
            gap_close_scalar(V, H, M, carry, k, var_in, inner_in);          // GetVariants.java:228-244
            gap_advance(H, M, carry);
        }
    }
    TileSummary s;
    s.n = n_valid;
    s.obs = obs;
    s.lead = carry.cany ? carry.lead : n_valid;
    s.trail = carry.cany ? carry.cg : n_valid;
    s.var_in = var_in;
    s.inner_in = inner_in;
    s.runs = runs;
    s.flags = (carry.cany ? F_ANY : 0u) | ((Vw[0] & 1) ? F_FIRST : 0u) | (((V >> ((n_kmers - 1) & 63)) & 1) ? F_LAST : 0u);
    s.count_sum = count_sum;
    return s;
}

}  // namespace kcfx
