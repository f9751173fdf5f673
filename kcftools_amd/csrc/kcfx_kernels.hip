// kcfx_kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64) of the getVariations engine.
//
//   table_*          : build the HBM table from (k-mer word, count) pairs        (load time)
//   decode_records   : KMC .kmc_suf records + LUT -> (k-mer word, count)         (load time; KMC.java:427-450)
//   lookup           : batched KMC.getCount                                      (KMC.java:292-326)
//   screen_tiles     : THE hot kernel. ASCII window text -> 2-bit pack in LDS -> per-lane canonical k-mer ->
//                      one 64-byte line probe -> wave ballots -> gap monoid      (GetVariants.java:202-252,
//                                                                                 Fasta.java:90-167, Kmer.java)
//   finalize_windows : fold a window's tile summaries -> kcfx_result             (GetVariants.java:254-258)
//
// Integer / index work, HBM-bound: no MFMA. Lane <-> consecutive k-mer start positions, so one __ballot is a
// 64-position slice of the window's hit bitmap.
#include <cstdlib>

#include "kcfx_device.h"

namespace kcfx {

// ======================================================================================================
// table build
// ======================================================================================================
__global__ void table_clear_kernel(uint4* lines, uint64_t n_vec) {
    uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n_vec; i += stride) lines[i] = make_uint4(0, 0, 0, 0);
}

// During the build the meta word (byte 60) is a plain arrival counter; table_finalize turns it into
// {occupied slots, overflow flag}.
template <bool WIDE>
__global__ void table_insert_kernel(uint8_t* lines, uint32_t n_lines, const uint64_t* __restrict__ keys,
                                    const uint32_t* __restrict__ counts, int64_t n) {
    constexpr uint32_t S = WIDE ? SLOTS_U32 : SLOTS_U16;
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint64_t key = keys[i];
        const uint32_t c = counts[i];
        uint32_t line = home_line(key, n_lines);
        for (;;) {
            uint8_t* base = lines + (size_t)line * LINE_BYTES;
            const uint32_t slot = atomicAdd(reinterpret_cast<uint32_t*>(base + 60), 1u);
            if (slot < S) {
                reinterpret_cast<uint64_t*>(base)[slot] = key;
                if (WIDE)
                    reinterpret_cast<uint32_t*>(base + 40)[slot] = c;
                else
                    reinterpret_cast<uint16_t*>(base + 48)[slot] = (uint16_t)c;
                break;
            }
            line = (line + 1 == n_lines) ? 0 : line + 1;
        }
    }
}

template <bool WIDE>
__global__ void table_finalize_kernel(uint8_t* lines, uint32_t n_lines) {
    constexpr uint32_t S = WIDE ? SLOTS_U32 : SLOTS_U16;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (; i < n_lines; i += stride) {
        uint32_t* meta = reinterpret_cast<uint32_t*>(lines + (size_t)i * LINE_BYTES + 60);
        const uint32_t arrivals = *meta;
        *meta = (arrivals > S) ? (S | META_OVERFLOW) : arrivals;
    }
}

// KMC.java:427-450 (dumpKmerTable): record j belongs to the last LUT slot i with lut[i] <= j;
// k-mer = prefix bases (i mod 4^p) || suffix bases; counter little-endian (KMC.java:395-401).
__global__ void decode_records_kernel(const uint8_t* __restrict__ records, int64_t first_record, int64_t n,
                                      int32_t record_size, int32_t suffix_bytes, int32_t counter_size, int32_t p,
                                      const uint64_t* __restrict__ lut, int64_t lut_entries,
                                      uint64_t* __restrict__ keys_out, uint32_t* __restrict__ counts_out) {
    int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; t < n; t += stride) {
        const uint64_t j = (uint64_t)(first_record + t);
        // upper_bound(lut, j) - 1
        int64_t lo = 0, hi = lut_entries;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (lut[mid] <= j) lo = mid + 1; else hi = mid;
        }
        const uint64_t slot = (uint64_t)(lo - 1);
        const uint64_t prefix = p ? (slot & ((1ull << (2 * p)) - 1)) : 0;
        const uint8_t* r = records + t * record_size;
        uint64_t suf = 0;
        for (int b = 0; b < suffix_bytes; ++b) suf |= (uint64_t)r[b] << (56 - 8 * b);
        uint64_t key = suf >> (2 * p);
        if (p) key |= prefix << (64 - 2 * p);
        uint32_t c = 0;
        for (int b = 0; b < counter_size; ++b) c |= (uint32_t)r[suffix_bytes + b] << (8 * b);
        keys_out[t] = key;
        counts_out[t] = c;
    }
}

// ======================================================================================================
// probe: one 64-byte line per step; a miss continues to the next line only while the overflow bit is set
// ======================================================================================================
template <bool WIDE, bool SINGLE_ROUND = false>
__device__ __forceinline__ bool probe(const TableView& tv, uint64_t key, int32_t& count) {
    uint32_t line = home_line(key, tv.n_lines);
    const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
    for (;;) {
        const uint4* p = reinterpret_cast<const uint4*>(tv.lines + (size_t)line * LINE_BYTES);
        const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
        // pin the count dwords here: without it LLVM sinks their load below the key compares, i.e. into a second,
        // dependent memory round trip per probe
        asm volatile("" ::"v"(q3.x), "v"(q3.y), "v"(q3.z), "v"(q3.w));
        const uint32_t meta = q3.w;
        const uint32_t ns = meta & 0xFFu;
        // Branch-free match: every loaded dword is consumed unconditionally, so the compiler keeps the four 16-byte
        // loads of the line together in ONE memory round trip (with ?: chains it sinks the count loads into a
        // second, dependent access after the key compares).
        uint32_t c;
        uint32_t any;
#define KCFX_EQ(lo, hi, i) (((((lo) ^ klo) | ((hi) ^ khi)) == 0u && ns > (i)) ? 0xFFFFFFFFu : 0u)
        if (!WIDE) {
            const uint32_t e0 = KCFX_EQ(q0.x, q0.y, 0), e1 = KCFX_EQ(q0.z, q0.w, 1), e2 = KCFX_EQ(q1.x, q1.y, 2);
            const uint32_t e3 = KCFX_EQ(q1.z, q1.w, 3), e4 = KCFX_EQ(q2.x, q2.y, 4), e5 = KCFX_EQ(q2.z, q2.w, 5);
            any = e0 | e1 | e2 | e3 | e4 | e5;
            c = (q3.x & ((e0 & 0xFFFFu) | (e1 & 0xFFFF0000u))) | (q3.y & ((e2 & 0xFFFFu) | (e3 & 0xFFFF0000u))) |
                (q3.z & ((e4 & 0xFFFFu) | (e5 & 0xFFFF0000u)));
            c = (c & 0xFFFFu) | (c >> 16);
        } else {
            const uint32_t e0 = KCFX_EQ(q0.x, q0.y, 0), e1 = KCFX_EQ(q0.z, q0.w, 1), e2 = KCFX_EQ(q1.x, q1.y, 2);
            const uint32_t e3 = KCFX_EQ(q1.z, q1.w, 3), e4 = KCFX_EQ(q2.x, q2.y, 4);
            any = e0 | e1 | e2 | e3 | e4;
            c = (q2.z & e0) | (q2.w & e1) | (q3.x & e2) | (q3.y & e3) | (q3.z & e4);
        }
#undef KCFX_EQ
        const bool found = any != 0u;
        if (found) {
            count = (int32_t)c;           // Java int (KMC.java:395-401)
            return true;
        }
        if (SINGLE_ROUND || !(meta & META_OVERFLOW)) {
            count = 0;
            return false;
        }
        line = (line + 1 == tv.n_lines) ? 0 : line + 1;
    }
}

template <bool WIDE>
__global__ void lookup_kernel(TableView tv, const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ out) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int32_t c;
        probe<WIDE>(tv, keys[i], c);
        out[i] = c;
    }
}

// ======================================================================================================
// screen_tiles
// ======================================================================================================
constexpr int MAX_CHUNKS = (31 + TILE_K + 31 + 31) / 32 + 1;   // 32-base chunks a tile can span

__device__ __forceinline__ int32_t wave_sum_i32(int32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// VAR != 0 only exists in the experiments build (-DKCFX_EXPERIMENTS, libkcfx_exp.so): timing ablations whose
// results are deliberately wrong. 1: single probe round (ignore overflow)  2: no gap accounting
// 3: no table probe  4: no k-mer extraction (synthetic keys)
template <bool WIDE, int VAR>
__global__ __launch_bounds__(BLOCK) void screen_tiles_kernel(const uint8_t* __restrict__ text,
                                                             const TileDesc* __restrict__ tiles, TableView tv,
                                                             int32_t k, int32_t both_strands, int32_t min_count,
                                                             TileSummary* __restrict__ out) {
    __shared__ uint32_t s_fw[2 * (MAX_CHUNKS + 2)];
    __shared__ uint32_t s_cw[2 * (MAX_CHUNKS + 2)];
    __shared__ uint32_t s_vb[MAX_CHUNKS + 2];
    __shared__ TileSummary s_sum[BLOCK / 64];

    const TileDesc td = tiles[blockIdx.x];
    const int tid = threadIdx.x;

    // ---- phase 1: ASCII -> 2-bit codes + validity bits in LDS (coalesced 16-byte loads) --------------
    const int64_t a0 = td.text_off & ~(int64_t)31;
    const int off = (int)(td.text_off - a0);
    const int nbases = off + td.n_kmers + k - 1;
    const int nchunks = (nbases + 31) >> 5;
    for (int c = tid; c < nchunks + 2; c += BLOCK) {
        uint32_t f0 = 0, f1 = 0, v0 = 0, v1 = 0;
        if (c < nchunks) {
            const uint4* p = reinterpret_cast<const uint4*>(text + a0 + 32 * (int64_t)c);
            const uint4 x = p[0], y = p[1];
            pack16(x.x, x.y, x.z, x.w, f0, v0);
            pack16(y.x, y.y, y.z, y.w, f1, v1);
        }
        s_fw[2 * c] = f0;
        s_fw[2 * c + 1] = f1;
        s_cw[2 * c] = comp_swap(f0);
        s_cw[2 * c + 1] = comp_swap(f1);
        s_vb[c] = (v0 << 16) | v1;
    }
    __syncthreads();

    // ---- phase 2: each wave walks a contiguous range of 64-position words ------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int n_words = (td.n_kmers + 63) >> 6;
    const int wpw = (n_words + (BLOCK / 64) - 1) / (BLOCK / 64);
    const int w_begin = min(wave * wpw, n_words);
    const int w_end = min(w_begin + wpw, n_words);

    GapCarry carry;
    carry.cg = 0;
    carry.lead = 0;
    carry.cany = false;
    int32_t n_valid = 0, obs = 0, runs = 0;
    uint32_t prev_valid = 0, first_valid = 0;
    uint64_t last_V = 0;
    int32_t var_acc = 0, inner_acc = 0;
    int64_t sum_acc = 0;

    for (int w = w_begin; w < w_end; ++w) {
        const int j = w * 64 + lane;
        uint64_t fwd, rc;
        bool valid;
        if (VAR == 4) {
            fwd = (uint64_t)hash32((uint64_t)(blockIdx.x * 4096u + j)) << 32 | hash32((uint64_t)j * 77u + blockIdx.x);
            rc = ~fwd;
            valid = true;
        } else {
            valid = extract_kmer(s_fw, s_cw, s_vb, off + min(j, td.n_kmers - 1), k, fwd, rc);
        }
        valid = valid && (j < td.n_kmers);
        // Kmer(Kmer, bothStrands): canonical = min(fwd, revcomp) unsigned (Kmer.java:72-79,406-414)
        const uint64_t key = (both_strands && rc < fwd) ? rc : fwd;
        int32_t cnt = 0;
        bool found = false;
        if (VAR == 3) {
            cnt = (int32_t)(key >> 60);
            found = (key & 0x300) != 0;
        } else if (valid) {
            found = probe<WIDE, VAR == 1>(tv, key, cnt);
        }
        const bool hit = found && (cnt >= min_count);            // GetVariants.java:224

        const uint64_t V = __ballot(valid);
        const uint64_t H = __ballot(hit);
        const uint64_t M = V & ~H;
        n_valid += __popcll(V);
        obs += __popcll(H);
        if (hit) sum_acc += (int64_t)cnt;                          // GetVariants.java:226

        // maximal runs of valid start positions (for EFFLEN)
        runs += __popcll(V & ~((V << 1) | (uint64_t)prev_valid));
        if (w == w_begin) first_valid = (uint32_t)(V & 1);
        prev_valid = (uint32_t)(V >> 63);
        last_V = V;

        // gap accounting (GetVariants.java:228-244)
        if (VAR == 2) {
            var_acc += (int32_t)(M >> lane) & 1;
        } else if (H) {
            if (M) {
                if (hit) {
                    const int32_t g = gap_closed_by(H, M, lane, carry);
                    if (g > 0) {
                        var_acc += 1;
                        inner_acc += gap_distance(k, g);
                    }
                }
            } else if (carry.cany && carry.cg > 0 && lane == 0) {
                // every valid position of this word hit: only its first hit can close a (carried) gap
                var_acc += 1;
                inner_acc += gap_distance(k, carry.cg);
            }
        }
        gap_advance(H, M, carry);
    }

    // ---- wave summary --------------------------------------------------------------------------------
    const int32_t var_in = wave_sum_i32(var_acc);
    const int32_t inner_in = wave_sum_i32(inner_acc);
    const int64_t count_sum = wave_sum_i64(sum_acc);
    if (lane == 0) {
        TileSummary s;
        if (w_begin == w_end) {
            s = summary_identity();
        } else {
            const int last_pos = min(w_end * 64, td.n_kmers) - 1;
            s.n = n_valid;
            s.obs = obs;
            s.lead = carry.cany ? carry.lead : n_valid;
            s.trail = carry.cany ? carry.cg : n_valid;
            s.var_in = var_in;
            s.inner_in = inner_in;
            s.runs = runs;
            s.flags = (carry.cany ? F_ANY : 0u) | (first_valid ? F_FIRST : 0u) |
                      (((last_V >> (last_pos & 63)) & 1) ? F_LAST : 0u);
            s.count_sum = count_sum;
        }
        s_sum[wave] = s;
    }
    __syncthreads();
    if (tid == 0) {
        TileSummary acc = s_sum[0];
#pragma unroll
        for (int i = 1; i < BLOCK / 64; ++i) acc = summary_combine(acc, s_sum[i], k);
        out[blockIdx.x] = acc;
    }
}

// one thread per window: fold its tiles in order (GetVariants.java:248-258)
__global__ void finalize_windows_kernel(const TileSummary* __restrict__ tiles, const int32_t* __restrict__ win_tile0,
                                        int32_t n_windows, int32_t k, kcfx_result* __restrict__ out) {
    const int32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_windows) return;
    TileSummary acc = summary_identity();
    for (int32_t t = win_tile0[w]; t < win_tile0[w + 1]; ++t) acc = summary_combine(acc, tiles[t], k);
    if (acc.flags & F_EMPTY) {
        acc.flags = 0;   // window shorter than k: empty list => all zeros (GetVariants.java:219)
    }
    out[w] = summary_finalize(acc, k);
}

// ======================================================================================================
// launchers
// ======================================================================================================
static inline int grid_for(int64_t n, int block, int cap = 256 * 8) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

hipError_t launch_table_clear(uint8_t* lines, uint32_t n_lines, hipStream_t st) {
    const uint64_t n_vec = (uint64_t)n_lines * (LINE_BYTES / 16);
    hipLaunchKernelGGL(table_clear_kernel, dim3(grid_for((int64_t)n_vec, 256)), dim3(256), 0, st,
                       reinterpret_cast<uint4*>(lines), n_vec);
    return hipGetLastError();
}

hipError_t launch_decode_records(const uint8_t* records, int64_t first_record, int64_t n, int32_t record_size,
                                 int32_t suffix_bytes, int32_t counter_size, int32_t lut_prefix_length,
                                 const uint64_t* lut, int64_t lut_entries, uint64_t* keys_out, uint32_t* counts_out,
                                 hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(decode_records_kernel, dim3(grid_for(n, 256, 256 * 16)), dim3(256), 0, st, records, first_record,
                       n, record_size, suffix_bytes, counter_size, lut_prefix_length, lut, lut_entries, keys_out,
                       counts_out);
    return hipGetLastError();
}

hipError_t launch_table_insert(uint8_t* lines, uint32_t n_lines, int32_t wide, const uint64_t* keys,
                               const uint32_t* counts, int64_t n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 g(grid_for(n, 256, 256 * 16)), b(256);
    if (wide)
        hipLaunchKernelGGL(table_insert_kernel<true>, g, b, 0, st, lines, n_lines, keys, counts, n);
    else
        hipLaunchKernelGGL(table_insert_kernel<false>, g, b, 0, st, lines, n_lines, keys, counts, n);
    return hipGetLastError();
}

hipError_t launch_table_finalize(uint8_t* lines, uint32_t n_lines, int32_t wide, hipStream_t st) {
    const dim3 g(grid_for(n_lines, 256)), b(256);
    if (wide)
        hipLaunchKernelGGL(table_finalize_kernel<true>, g, b, 0, st, lines, n_lines);
    else
        hipLaunchKernelGGL(table_finalize_kernel<false>, g, b, 0, st, lines, n_lines);
    return hipGetLastError();
}

hipError_t launch_lookup(TableView tv, const uint64_t* keys, int64_t n, int32_t* counts_out, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 g(grid_for(n, 256, 256 * 16)), b(256);
    if (tv.wide)
        hipLaunchKernelGGL(lookup_kernel<true>, g, b, 0, st, tv, keys, n, counts_out);
    else
        hipLaunchKernelGGL(lookup_kernel<false>, g, b, 0, st, tv, keys, n, counts_out);
    return hipGetLastError();
}

hipError_t launch_screen_tiles(const uint8_t* text, const TileDesc* tiles, int32_t n_tiles, TableView tv, int32_t k,
                               int32_t both_strands, int32_t min_count, TileSummary* out, hipStream_t st) {
    if (n_tiles <= 0) return hipSuccess;
    const dim3 g(n_tiles), b(BLOCK);
#ifdef KCFX_EXPERIMENTS
    static const int var = getenv("KCFX_VARIANT") ? atoi(getenv("KCFX_VARIANT")) : 0;
#define KCFX_VARIANT_CASE(V) \
    if (var == V) { hipLaunchKernelGGL((screen_tiles_kernel<false, V>), g, b, 0, st, text, tiles, tv, k, both_strands, min_count, out); return hipGetLastError(); }
    if (!tv.wide) { KCFX_VARIANT_CASE(1) KCFX_VARIANT_CASE(2) KCFX_VARIANT_CASE(3) KCFX_VARIANT_CASE(4) }
#endif
    if (tv.wide)
        hipLaunchKernelGGL((screen_tiles_kernel<true, 0>), g, b, 0, st, text, tiles, tv, k, both_strands, min_count, out);
    else
        hipLaunchKernelGGL((screen_tiles_kernel<false, 0>), g, b, 0, st, text, tiles, tv, k, both_strands, min_count, out);
    return hipGetLastError();
}

hipError_t launch_finalize_windows(const TileSummary* tiles, const int32_t* win_tile0, int32_t n_windows, int32_t k,
                                   kcfx_result* out, hipStream_t st) {
    if (n_windows <= 0) return hipSuccess;
    hipLaunchKernelGGL(finalize_windows_kernel, dim3((n_windows + 127) / 128), dim3(128), 0, st, tiles, win_tile0,
                       n_windows, k, out);
    return hipGetLastError();
}

}  // namespace kcfx
