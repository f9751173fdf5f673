// kcfx_plan.h -- the planner: ASCII reference text -> 2-bit codes + one anchor byte per k-mer start (RefPlan, kcfx_internal.h).
//
// What SURVEY.md calls K1 (Fasta.java:90-127: which positions start a valid k-mer; Kmer.java:232-252: the 2-bit code) plus the
// sample-independent half of the cell-table look-up: which 16-mer of the k-mer is its minimizer and on which strand that 16-mer
// is canonical (cell_order27, kcfx_device.h). Round 3 did this one text position per LANE: eleven 6-mer hashes and a 16-mer hash
// per position, two row scans and two ds_bpermute per word -- 0.101 ms for a 30 Mb reference. Here a lane owns a STRIP of L
// consecutive positions and walks it in registers with compile-time indices:
//   * one hash per position (the canonical 6-mer there); class and order key of every 16-mer are sliding minima / a sliding xor
//     over those (v_min3_u32, v_bitop3), the minimizer of every k-mer a sliding minimum over the tagged keys;
//   * nothing crosses lanes except what a strip needs of its right-hand neighbour -- its first code word, the hashes of its
//     first 10 positions, the tagged keys of its first W - 1 -- three small exchanges (`Io`: wave-private LDS on the device);
//   * the anchor bytes are finished four at a time in one dword (byte-wise subtract, flip and mask).
// About 28 VALU instructions per text position instead of ~130. The code is __host__ __device__: tests/host_harness runs the
// very same function strip by strip on the CPU against the position-by-position definition (cell_probe_of + cells_visit).
#pragma once

#include <type_traits>
#include <utility>

#include "kcfx_device.h"

namespace kcfx {

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1 (indices stay constants: registers, inline constants)
template <int B, int E, class F>
KCFX_HD void static_for(F&& f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// "the canonical form of the 16-mer at this position is its reverse complement": a lane mask in scalar registers on the device
// (the compare writes it there for free and v_addc_co_u32 takes it as its carry-in), a bool on the host
#if defined(__HIP_DEVICE_COMPILE__)
typedef uint64_t StrandFlag;
template <uint32_t LABEL2>
__device__ __forceinline__ uint32_t order_tag(uint32_t key, StrandFlag rc_smaller) {
    uint32_t t;                                             // key + label + carry: the three fields do not overlap
    asm("v_addc_co_u32 %0, vcc, %1, %2, %3" : "=v"(t) : "v"(key), "n"(LABEL2), "s"(rc_smaller) : "vcc");
    return t;
}
__device__ __forceinline__ StrandFlag strand_flag(uint32_t rc_m, uint32_t fwd_m) { return __builtin_amdgcn_ballot_w64(rc_m < fwd_m); }
__device__ __forceinline__ uint32_t low_bytes4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {   // byte 0 of each, a first
    const uint32_t ab = __builtin_amdgcn_perm(b, a, 0x0c0c0400u), cd = __builtin_amdgcn_perm(d, c, 0x0c0c0400u);
    return __builtin_amdgcn_perm(cd, ab, 0x05040100u);
}
#else
typedef bool StrandFlag;
template <uint32_t LABEL2>
inline uint32_t order_tag(uint32_t key, StrandFlag rc_smaller) { return key | LABEL2 | (rc_smaller ? 1u : 0u); }
inline StrandFlag strand_flag(uint32_t rc_m, uint32_t fwd_m) { return rc_m < fwd_m; }
inline uint32_t low_bytes4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return (a & 0xFFu) | ((b & 0xFFu) << 8) | ((c & 0xFFu) << 16) | ((d & 0xFFu) << 24);
}
#endif

// A strip = L consecutive text positions starting at a multiple of 16 of the planner's origin (so the position label of its
// j-th position is the constant 2 * (j & 15)). `Io` hands the strip its text and carries the three exchanges with the strip
// to its right (the next lane); see PlanWave in kcfx_kernels.hip and HostStripIo in tests/host_harness.
template <int K, int L>
struct StripPlanner {
    static_assert(L == 16 || L == 32, "a strip is one or two code words");
    static_assert(K >= CELL_K_MIN && K <= CELL_K_MAX, "cell tables: k = 19 .. 31");
    static constexpr int NF = L / 16;                       // code words of the strip's own bases
    static constexpr int W = cell_nw(K);                    // 16-mers per k-mer: the window of the sliding minimum
    static constexpr int UH = 10;                           // hashes needed of the next strip: a 16-mer spans eleven 6-mers
    static constexpr int TH = W - 1;                        // tagged keys needed of the next strip
    static constexpr int VN = (K - 1 + L - 1) / L;          // following strips a k-mer of this one reaches into (validity)
    static_assert(UH <= L && TH <= L, "one neighbour is enough");

    struct First {                                          // exchange 0: what the strip to the left needs before it can hash
        uint32_t F0, G0;                                    // first code word, forward and reverse-complemented (brev of comp_swap)
        uint32_t valid;                                     // validity bits of the strip's L bases, first base in bit L - 1
    };
    struct Result {
        uint32_t codes[NF];                                 // 2-bit codes, 16 bases per word, first base in the top bits (RefPlan::codes)
        uint32_t anchors[L / 4];                            // anchor bytes of the L positions, little-endian (RefPlan::anchors)
    };

    // positions j .. j + K - 1 all valid, for the 64 positions from the strip's start: V = validity of the 64 bases from there,
    // first base in bit 63 (bases behind bit 0 count as valid: no position that matters reaches them -- L + K - 2 <= 63)
    KCFX_HD static uint64_t valid_starts(uint64_t V) {
        uint64_t r = V;
        int have = 1;
        while (2 * have <= K) {
            r &= (r << have) | ((1ull << have) - 1);
            have *= 2;
        }
        if (have < K) r &= (r << (K - have)) | ((1ull << (K - have)) - 1);
        return r;
    }

    template <class Io>
    KCFX_HD static Result run(Io& io) {
        Result res;
        // ---- the strip's own bases: ASCII -> codes (Kmer.java:286-294) + validity (Fasta.java:98,132-134) ----
        uint32_t F[NF + 1], G[NF + 1], valid = 0;
        uint32_t x[L / 4];
        io.load_text(x);
        static_for<0, NF>([&](auto ic) {
            constexpr int i = ic;
            uint32_t c, dirt, v = 0xFFFFu;
            pack16_codes(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3], c, dirt);
            if (dirt != 0u) {                               // rare: an N, an IUPAC code, a non-letter among these 16
                uint32_t c2;
                pack16(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3], c2, v);
            }
            F[i] = c;
            G[i] = brev32(comp_swap(c));                    // the reverse complement of these 16 bases, left-aligned
            res.codes[i] = c;
            valid = (valid << 16) | v;
        });
        First nb[VN];
        io.exchange_first(First{F[0], G[0], valid}, nb);
        F[NF] = nb[0].F0;
        G[NF] = nb[0].G0;

        // ---- one hash per position: the canonical 6-mer there; and which strand of the 16-mer there is canonical ----
        uint32_t u[L + UH];
        StrandFlag strand[L];
        static_for<0, L>([&](auto jc) {
            constexpr int j = jc, a = j >> 4, b = j & 15;
            // the 16-mer at j and its reverse complement, both left-aligned: one funnel shift each
            const uint32_t F16 = b ? alignbit32(F[a], F[a + 1], 32u - 2u * b) : F[a];
            const uint32_t R16 = b ? alignbit32(G[a + 1], G[a], 2u * b) : G[a];
            u[j] = smer6_hash(F16 >> 20, R16 & 0xFFFu);     // its first 6-mer; reverse-complemented: the last 6 bases of R16
            strand[j] = strand_flag(R16, F16);
        });
        {
            uint32_t uh[UH];
            io.exchange_hashes(u, uh);                      // publishes u[0 .. UH - 1]
            static_for<0, UH>([&](auto ic) { u[L + ic] = uh[ic]; });
        }

        // ---- tagged order key of the 16-mer at every position (cell_order27 | label | strand) ----
        uint32_t t[L + TH];
        {
            uint32_t m3[L + 7];                             // m3[i] = min(u[i .. i + 2]), i = 1 .. L + 6
            static_for<1, L + 7>([&](auto ic) {
                constexpr int i = ic;
                m3[i] = min3u(u[i], u[i + 1], u[i + 2]);
            });
            uint32_t hx = xor3u(xor3u(xor3u(u[0], u[1], u[2]), xor3u(u[3], u[4], u[5]), xor3u(u[6], u[7], u[8])), u[9], u[10]);
            static_for<0, L>([&](auto jc) {
                constexpr int j = jc;
                const uint32_t inner = min3u(m3[j + 1], m3[j + 4], m3[j + 7]);
                const uint32_t key = (hx & ORDER_HASH_MASK) | cell_class30(u[j], u[j + 5], u[j + 10], inner);
                t[j] = order_tag<2u * (j & 15)>(key, strand[j]);
                if constexpr (j + 1 < L) hx = xor3u(hx, u[j], u[j + 11]);
            });
            uint32_t th[TH];
            io.exchange_keys(t, th);                        // publishes t[0 .. TH - 1]
            static_for<0, TH>([&](auto ic) { t[L + ic] = th[ic]; });
        }

        // ---- minimizer of every k-mer: sliding minimum over its W tagged keys ----
        uint32_t mn[L];
        {
            uint32_t a3[L + TH - 2];                        // a3[i] = min(t[i .. i + 2])
            static_for<0, L + TH - 2>([&](auto ic) {
                constexpr int i = ic;
                a3[i] = min3u(t[i], t[i + 1], t[i + 2]);
            });
            if constexpr (W >= 10) {
                uint32_t a9[L + W - 9];                     // a9[i] = min(t[i .. i + 8])
                static_for<0, L + W - 9>([&](auto ic) {
                    constexpr int i = ic;
                    a9[i] = min3u(a3[i], a3[i + 3], a3[i + 6]);
                });
                static_for<0, L>([&](auto jc) {
                    constexpr int j = jc;
                    mn[j] = a9[j] < a9[j + W - 9] ? a9[j] : a9[j + W - 9];
                });
            } else if constexpr (W >= 7) {
                static_for<0, L>([&](auto jc) {
                    constexpr int j = jc;
                    mn[j] = min3u(a3[j], a3[j + 3], a3[j + W - 3]);
                });
            } else {
                static_for<0, L>([&](auto jc) {
                    constexpr int j = jc;
                    mn[j] = a3[j] < a3[j + W - 3] ? a3[j] : a3[j + W - 3];
                });
            }
        }

        // ---- which positions start a valid k-mer (Fasta.java:90-127) ----
        uint64_t V = (uint64_t)valid << (64 - L);
        static_for<0, VN>([&](auto ic) {
            constexpr int i = ic;
            V |= (uint64_t)nb[i].valid << (64 - 2 * L - i * L);
        });
        if constexpr ((VN + 1) * L < 64) V |= (1ull << (64 - (VN + 1) * L)) - 1;
        const uint64_t S = io.all_clean(V == ~0ull) ? ~0ull : valid_starts(V);

        // ---- anchor bytes, four positions per dword: bit 7 valid, bits 6..5 the minimizer's slice class, bits 4..1 twice the
        //      minimizer's offset in N, bit 0 "N is the reverse complement of the k-mer as the text reads it" (cell_probe_of's sh
        //      and flip) ----
        static_for<0, L / 4>([&](auto qc) {
            constexpr int q = qc;
            constexpr uint32_t c0 = (8u * q) & 31u;         // twice the position label of the dword's first position
            constexpr uint32_t C4 = c0 | (((c0 + 2u) & 31u) << 8) | (((c0 + 4u) & 31u) << 16) | (((c0 + 6u) & 31u) << 24);
            uint32_t d = low_bytes4(mn[4 * q], mn[4 * q + 1], mn[4 * q + 2], mn[4 * q + 3]);
            const uint32_t cls = d & 0x60606060u;           // bits 6..5 of the minimum: the minimizer's slice class (order_slice_class)
            // label of the minimum - label of the position, mod 32, in every byte at once (bit 5 keeps the borrows apart)
            d = ((d & 0x1F1F1F1Fu) | 0x20202020u) - C4;
            // bytes whose strand bit is set: the minimizer sits at W - 1 - offset of the reverse complement
            const uint32_t fl = d & 0x01010101u;
            d ^= fl * 30u;
            if constexpr (W < 16) d -= fl * (uint32_t)(32 - 2 * W);
            const uint32_t nib = (uint32_t)(S >> (60 - 4 * q)) & 0xFu;
            res.anchors[q] = (d & 0x1F1F1F1Fu) | cls | ((nib * 0x80402010u) & 0x80808080u);
        });
        return res;
    }
};

}  // namespace kcfx
