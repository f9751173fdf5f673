// Lane-by-lane CPU emulation of screen_tiles_kernel built from the SAME __host__ __device__ helpers
// (kcfx_device.h): base packing, k-mer extraction, ballot-style gap accounting, monoid combine.
// TEST INFRASTRUCTURE ONLY: lets the bit logic be checked against the oracle without a GPU. The table probe is
// replaced by a caller-provided count per position (the test computes it with the oracle).
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../kcftools_amd/csrc/kcfx_device.h"
#include "../../kcftools_amd/csrc/kcfx_plan.h"

using namespace kcfx;

namespace kcfx {
void set_error(const std::string&) {}
int32_t hip_fail(hipError_t, const char*) { return 4; }
}

static int g_ext = 0;      // harness_set_ext: the cell functions below act on an EXTENDED table (cell_first_ext)

static constexpr int MAX_CHUNKS = (31 + TILE_K + 31 + 31) / 32 + 1;

struct Packed {
    std::vector<uint32_t> fwx, cwx, vbx;     // [0] = the lead word bits_at may touch (as in the kernel's LDS arrays)
    int off;
    const uint32_t* fw() const { return fwx.data() + 1; }
    const uint32_t* cw() const { return cwx.data() + 1; }
    const uint32_t* vb() const { return vbx.data() + 1; }
};

static Packed pack_tile(const uint8_t* text, int64_t text_off, int n_kmers, int k) {
    Packed P;
    P.fwx.assign(2 * (MAX_CHUNKS + 2) + 1, 0xDEADBEEFu);
    P.cwx.assign(2 * (MAX_CHUNKS + 2) + 1, 0xDEADBEEFu);
    P.vbx.assign(MAX_CHUNKS + 2 + 1, 0xDEADBEEFu);
    uint32_t* fw = P.fwx.data() + 1;
    uint32_t* cw = P.cwx.data() + 1;
    uint32_t* vb = P.vbx.data() + 1;
    const int64_t a0 = text_off & ~(int64_t)31;
    P.off = (int)(text_off - a0);
    const int nbases = P.off + n_kmers + k - 1;
    const int nchunks = (nbases + 31) >> 5;
    for (int c = 0; c < nchunks + 2; ++c) {
        uint32_t f0 = 0, f1 = 0, v0 = 0, v1 = 0;
        if (c < nchunks) {
            uint32_t x[8];
            memcpy(x, text + a0 + 32 * (int64_t)c, 32);
            // as k31_pack_tile: codes + one "dirty" word per 16 bases, per-base validity only for dirty chunks
            uint32_t d0, d1;
            pack16_codes(x[0], x[1], x[2], x[3], f0, d0);
            pack16_codes(x[4], x[5], x[6], x[7], f1, d1);
            v0 = v1 = 0xFFFFu;
            uint32_t g0, g1, w0, w1;
            pack16(x[0], x[1], x[2], x[3], g0, w0);
            pack16(x[4], x[5], x[6], x[7], g1, w1);
            if (g0 != f0 || g1 != f1 || ((d0 | d1) == 0) != (w0 == 0xFFFFu && w1 == 0xFFFFu)) {   // the two forms must agree
                f0 = f1 = 0xBADBAD00u;
                v0 = v1 = 0;
            }
            if ((d0 | d1) != 0) {
                v0 = w0;
                v1 = w1;
            }
        }
        fw[2 * c] = f0;
        fw[2 * c + 1] = f1;
        cw[2 * c] = comp_swap(f0);
        cw[2 * c + 1] = comp_swap(f1);
        vb[c] = (v0 << 16) | v1;
    }
    return P;
}

extern "C" {

// per position of the tile: forward word, reverse-complement word, validity
int harness_extract(const uint8_t* text, int64_t text_off, int n_kmers, int k, uint64_t* fwd, uint64_t* rc, uint8_t* valid) {
    Packed P = pack_tile(text, text_off, n_kmers, k);
    for (int j = 0; j < n_kmers; ++j) {
        uint64_t f, r;
        valid[j] = extract_kmer(P.fw(), P.cw(), P.vb(), P.off + j, k, f, r) ? 1 : 0;
        fwd[j] = f;
        rc[j] = r;
    }
    return 0;
}

// the screen kernel -> reduce_windows_kernel data flow with counts[j] standing in for the probe result:
// per 64-position word the valid / hit ballots, then summarize_words (the function the kernel runs per tile)
int harness_screen_tile(const uint8_t* text, int64_t text_off, int n_kmers, int k, const int32_t* counts,
                        int min_count, TileSummary* out) {
    Packed P = pack_tile(text, text_off, n_kmers, k);
    const int n_words = (n_kmers + 63) >> 6;
    std::vector<uint64_t> Vw(TILE_WORDS, 0), Hw(TILE_WORDS, 0);
    int64_t sum = 0;
    for (int w = 0; w < n_words; ++w) {
        uint64_t V = 0, H = 0;
        for (int lane = 0; lane < 64; ++lane) {
            const int j = w * 64 + lane;
            uint64_t f, r;
            const int q = P.off + std::min(j, n_kmers - 1);
            const bool valid = extract_kmer(P.fw(), P.cw(), P.vb(), q, k, f, r) && j < n_kmers;
            const int32_t cnt = valid ? counts[j] : 0;
            const bool hit = valid && cnt >= min_count;
            if (valid) V |= 1ull << lane;
            if (hit) { H |= 1ull << lane; sum += cnt; }
        }
        Vw[w] = V;
        Hw[w] = H;
    }
    *out = summarize_words(Vw.data(), Hw.data(), n_kmers, k, sum);
    // the register-resident form reduce_windows_kernel runs on 4-word pieces must agree with the walking one, piece by piece
    for (int p0 = 0; p0 * 256 < n_kmers; ++p0) {
        uint64_t Vl[4] = {0, 0, 0, 0}, Hl[4] = {0, 0, 0, 0};
        const int n_local = std::min(n_kmers - p0 * 256, 256);
        for (int i = 0; i < 4 && (size_t)(p0 * 4 + i) < Vw.size(); ++i) {
            Vl[i] = Vw[(size_t)(p0 * 4 + i)];
            Hl[i] = Hw[(size_t)(p0 * 4 + i)];
        }
        const TileSummary a = summarize_piece<4>(Vl, Hl, n_local, k, 7);
        const TileSummary b = summarize_words(Vw.data() + p0 * 4, Hw.data() + p0 * 4, n_local, k, 7);
        if (memcmp(&a, &b, sizeof a) != 0) out->n = -1000000;               // poisons the comparison with the oracle
    }
    return 0;
}

void harness_combine(const TileSummary* a, const TileSummary* b, int k, TileSummary* out) { *out = summary_combine(*a, *b, k); }
void harness_identity(TileSummary* out) { *out = summary_identity(); }
void harness_finalize(const TileSummary* s, int k, kcfx_result* out) {
    TileSummary t = *s;
    if (t.flags & F_EMPTY) t.flags = 0;
    *out = summary_finalize(t, k);
}
uint32_t harness_hash32(uint64_t key) { return hash32(key); }

// both formulations of the per-word gap accounting on the same masks: lane-parallel (gap_closed_by) and the
// scalar loop the kernel uses (gap_close_scalar). out[0..1] = var, inner of the former; out[2..3] of the latter.
void harness_gap_both(uint64_t V, uint64_t H, int cg, int cany, int k, int32_t* out) {
    const uint64_t M = V & ~H;
    GapCarry c{cg, 0, cany != 0};
    int32_t v1 = 0, i1 = 0, v2 = 0, i2 = 0;
    for (int lane = 0; lane < 64; ++lane) {
        if (!((H >> lane) & 1)) continue;
        if (M) {
            const int32_t g = gap_closed_by(H, M, lane, c);
            if (g > 0) { v1 += 1; i1 += gap_distance(k, g); }
        } else if (c.cany && c.cg > 0 && lane == __builtin_ctzll(H)) {
            v1 += 1; i1 += gap_distance(k, c.cg);
        }
    }
    gap_close_scalar(V, H, M, c, k, v2, i2);
    out[0] = v1; out[1] = i1; out[2] = v2; out[3] = i2;
}

// minimizer consistency: query side (hash of the canonical M-mer per text position, min over the k-mer's window -- what
// screen_tiles_kernel does with s_h) versus build side (minimizer_of_kmer on the stored word, canonical or forward)
int harness_minimizers(const uint8_t* text, int64_t text_off, int n_kmers, int k, int M, uint32_t* query_side,
                       uint32_t* build_fwd, uint32_t* build_canon) {
    Packed P = pack_tile(text, text_off, n_kmers, k);
    const int wlen = k - M + 1;
    std::vector<uint32_t> h(n_kmers + wlen);
    for (int p = 0; p < n_kmers + wlen - 1; ++p) {
        uint64_t f, r;
        extract_pair(P.fw(), P.cw(), P.off + p, M, f, r);
        h[p] = mmer_hash(f, r);
    }
    for (int j = 0; j < n_kmers; ++j) {
        uint32_t mz = h[j];
        for (int t = 1; t < wlen; ++t) mz = std::min(mz, h[j + t]);
        query_side[j] = mz;
        uint64_t f, r;
        extract_pair(P.fw(), P.cw(), P.off + j, k, f, r);
        build_fwd[j] = minimizer_of_kmer(f, k, M);
        build_canon[j] = minimizer_of_kmer(f < r ? f : r, k, M);
    }
    return 0;
}
// FORMAT_CELLS consistency (k = 31, M = 16): the cell the query side derives for the k-mer at every position (tagged
// hashes, sliding minimum, cell_probe_of -- what screen_k31c_kernel does in registers) must be one of the cells the
// build side stores the canonical k-mer in (cells_of_key). Returns the number of positions where it is not (0 = pass);
// n_cands_out[j] = number of build-side candidates, probe_out[4 j ..] = the query-side (line, sh, w, r).
// both = 0: a forward-only database -- the k-mer is stored, and looked up, exactly as the text reads it.
int harness_cells(const uint8_t* text, int64_t text_off, int n_kmers, int k, int lbits, int both, int32_t* n_cands_out, uint32_t* probe_out) {
    const int M = cell_mlen(k), W = cell_nw(k);
    Packed P = pack_tile(text, text_off, n_kmers, k);
    // position by position, straight from the definition (kcfx_device.h): u = hash of the canonical 6-mer at every position,
    // the 16-mer's key from its eleven, tagged with strand and position
    std::vector<uint32_t> u(n_kmers + 32), h(n_kmers + 16);
    for (int p = 0; p < n_kmers + 15 + 10; ++p) {
        uint64_t f, r;
        extract_pair(P.fw(), P.cw(), P.off + p, 6, f, r);
        u[p] = smer6_hash((uint32_t)(f >> 52), (uint32_t)(r >> 52));
    }
    for (int p = 0; p < n_kmers + 15; ++p) {
        uint64_t f, r;
        extract_pair(P.fw(), P.cw(), P.off + p, M, f, r);
        h[p] = order27_tagged(cell_order27(u.data() + p), (uint32_t)(f >> 32), (uint32_t)(r >> 32), order27_label2((uint32_t)p));     // extract_pair masks to M bases
    }
    int bad = 0;
    for (int j = 0; j < n_kmers; ++j) {
        uint32_t mn = h[j];
        for (int t = 1; t < W; ++t) mn = std::min(mn, h[j + t]);
        uint64_t F, R;
        extract_pair(P.fw(), P.cw(), P.off + j, k, F, R);
        CellProbe q = cell_probe_of(F, R, mn, 2u * ((uint32_t)j & 63u), lbits, M, both != 0, W);
        if (g_ext) q = cell_first_ext(q, lbits);            // (what cell_probe_planned does for an extended table)
        CellProbe c[CELL_MAX_CANDS];
        const int n = cells_of_key((both && R < F) ? R : F, k, lbits, c, both != 0, g_ext != 0);
        bool ok = false;
        for (int i = 0; i < n; ++i) ok |= c[i].line == q.line && c[i].sh == q.sh && c[i].w == q.w && c[i].r == q.r;
        if (!ok) ++bad;
        n_cands_out[j] = n;
        probe_out[4 * j] = q.line;
        probe_out[4 * j + 1] = q.sh;
        probe_out[4 * j + 2] = q.w;
        probe_out[4 * j + 3] = q.r;
    }
    return bad;
}
// injectivity spot check: the cells of two different canonical k-mers never coincide in (line, sh, w, r)
int harness_cells_of_key(uint64_t C, int k, int lbits, int both, uint32_t* out) {
    CellProbe c[CELL_MAX_CANDS];
    const int n = cells_of_key(C, k, lbits, c, both != 0, g_ext != 0);
    for (int i = 0; i < n; ++i) {
        out[4 * i] = c[i].line;
        out[4 * i + 1] = c[i].sh;
        out[4 * i + 2] = c[i].w;
        out[4 * i + 3] = c[i].r;
    }
    return n;
}
// out[0..2] = free-cell second choice (2 * index), the r field of the entry there, the bits of r compared there; returns 0 when
// the table has no free-cell scheme (k = 31, or too few spare r bits)
int harness_cell_alt(uint32_t sh, uint32_t w, uint32_t r, int k, int lbits, int both, uint32_t* out) {
    if (!cell_alt_free(cell_nw(k), lbits, both != 0)) return 0;
    const CellAlt a = cell_alt_of(sh, w, r, cell_nw(k), lbits, both != 0);
    out[0] = a.sh;
    out[1] = a.r;
    out[2] = a.rmask;
    return 1;
}
// out[0..3] = (line, sh, w, r) of the second-line probe of a first-choice probe
void harness_cell_second_line(uint32_t line, uint32_t sh, uint32_t w, uint32_t r, int lbits, uint32_t* out) {
    CellProbe p;
    p.line = line;
    p.sh = sh;
    p.w = w;
    p.r = r;
    const CellProbe q = cell_second_line(p, lbits);
    out[0] = q.line;
    out[1] = q.sh;
    out[2] = q.w;
    out[3] = q.r;
}
// out[0..3] = (line, sh, w, r) of the third-line probe; returns 0 when the table has no third lines
int harness_cell_third_line(uint32_t line, uint32_t sh, uint32_t w, uint32_t r, int k, int lbits, int both, uint32_t* out) {
    if (!cell_has_third(lbits, both != 0)) return 0;
    CellProbe p;
    p.line = line;
    p.sh = sh;
    p.w = w;
    p.r = r;
    const CellProbe q = cell_third_line(p, lbits, both != 0, cell_nw(k));
    out[0] = q.line;
    out[1] = q.sh;
    out[2] = q.w;
    out[3] = q.r;
    return 1;
}
uint32_t harness_cell_rmask(int lbits, int both) { return cell_rmask(lbits, both != 0); }
// the physical line of n probes (line, r) in a table of `regions` x 2^lbits lines (cell_line_index)
void harness_cell_line_index(const uint32_t* line, const uint32_t* r, int64_t n, int lbits, int regions, uint32_t* out) {
    for (int64_t i = 0; i < n; ++i) out[i] = cell_line_index(line[i], r[i], lbits, regions);
}
void harness_set_ext(int on) { g_ext = on; }
int harness_alt_free_max_w(void) { return CELL_ALT_FREE_MAX_W; }
uint64_t harness_revcomp_word(uint64_t x, int len) { return revcomp_word(x, len); }
int harness_tile_k(void) { return TILE_K; }

}

// ---- the strip planner (kcfx_plan.h), emulated: `n_strips` consecutive strips of L positions from text + origin, processed from
// the last to the first so that every strip finds what its right-hand neighbour published (the device runs them in lockstep
// with the same three exchanges). Strips past the last have published nothing: zeros, as the device reads unwritten rows.
template <int K, int L>
struct HostStripIo {
    using P = StripPlanner<K, L>;
    struct Pub {
        typename P::First first{0, 0, 0};
        uint32_t u[P::UH] = {0};
        uint32_t t[16] = {0};
    };
    std::vector<Pub>* pub;
    int strip;
    const uint8_t* text;      // null: zeros
    void load_text(uint32_t (&x)[L / 4]) const {
        if (text) memcpy(x, text, L); else memset(x, 0, L);
    }
    void exchange_first(const typename P::First& mine, typename P::First (&nb)[P::VN]) const {
        (*pub)[strip].first = mine;
        for (int i = 0; i < P::VN; ++i) nb[i] = (*pub)[strip + 1 + i].first;
    }
    void exchange_hashes(const uint32_t (&u)[L + P::UH], uint32_t (&uh)[P::UH]) const {
        for (int i = 0; i < P::UH; ++i) (*pub)[strip].u[i] = u[i];
        for (int i = 0; i < P::UH; ++i) uh[i] = (*pub)[strip + 1].u[i];
    }
    void exchange_keys(const uint32_t (&t)[L + P::TH], uint32_t (&th)[P::TH]) const {
        for (int i = 0; i < P::TH; ++i) (*pub)[strip].t[i] = t[i];
        for (int i = 0; i < P::TH; ++i) th[i] = (*pub)[strip + 1].t[i];
    }
    bool all_clean(bool mine) const { return mine; }
};
// exchange 0 must be complete for strips to the right before a strip hashes: on the device every lane packs first. Here: a
// first pass that only publishes `first`, then the full pass from right to left (run() publishes the same `first` again).
template <int K, int L>
static void plan_strips(const uint8_t* text, int64_t readable, int64_t origin, int n_strips, uint32_t* codes_out, uint8_t* anchors_out) {
    using P = StripPlanner<K, L>;
    std::vector<typename HostStripIo<K, L>::Pub> pub(n_strips + 3);
    auto strip_text = [&](int s) -> const uint8_t* { return origin + (int64_t)(s + 1) * L <= readable ? text + origin + (int64_t)s * L : nullptr; };
    for (int s = 0; s < n_strips; ++s) {
        uint32_t x[L / 4];
        HostStripIo<K, L> io{&pub, s, strip_text(s)};
        io.load_text(x);
        uint32_t valid = 0, F0 = 0, G0 = 0;
        for (int i = 0; i < L / 16; ++i) {
            uint32_t c, v;
            pack16(x[4 * i], x[4 * i + 1], x[4 * i + 2], x[4 * i + 3], c, v);
            if (i == 0) {
                F0 = c;
                G0 = brev32(comp_swap(c));
            }
            valid = (valid << 16) | v;
        }
        pub[s].first = typename P::First{F0, G0, valid};
    }
    for (int s = n_strips - 1; s >= 0; --s) {
        HostStripIo<K, L> io{&pub, s, strip_text(s)};
        const typename P::Result r = P::run(io);
        for (int i = 0; i < P::NF; ++i) codes_out[(size_t)s * P::NF + i] = r.codes[i];
        memcpy(anchors_out + (size_t)s * L, r.anchors, L);
    }
}
template <int L>
static int plan_strips_k(int k, const uint8_t* text, int64_t readable, int64_t origin, int n_strips, uint32_t* codes_out, uint8_t* anchors_out) {
    switch (k) {
#define KCFX_CASE(KK) case KK: plan_strips<KK, L>(text, readable, origin, n_strips, codes_out, anchors_out); return 0;
        KCFX_CASE(19) KCFX_CASE(20) KCFX_CASE(21) KCFX_CASE(22) KCFX_CASE(23) KCFX_CASE(24) KCFX_CASE(25)
        KCFX_CASE(26) KCFX_CASE(27) KCFX_CASE(28) KCFX_CASE(29) KCFX_CASE(30) KCFX_CASE(31)
#undef KCFX_CASE
    }
    return 1;
}
extern "C" {
// codes_out: n_strips * L / 16 words, anchors_out: n_strips * L bytes. The LAST strips depend on neighbours that do not exist
// (L = 32: the last one; L = 16: the last two) exactly as the last lanes of a wave do on the device.
int harness_plan_strips(const uint8_t* text, int64_t readable, int64_t origin, int n_strips, int k, int L, uint32_t* codes_out, uint8_t* anchors_out) {
    if (L == 32) return plan_strips_k<32>(k, text, readable, origin, n_strips, codes_out, anchors_out);
    if (L == 16) return plan_strips_k<16>(k, text, readable, origin, n_strips, codes_out, anchors_out);
    return 1;
}
// what the planner must produce for the k-mer at text position pos, from the definition: bit 7 valid, bits 6..5 = the slice class
// of the minimizer (two bits of its order key), bits 4..1 = 2 * offset of the minimizer in N, bit 0 = N is the reverse complement. Returns how many anchors are acceptable (cands[0 ..]): more than one
// when the minimum key is attained at several offsets -- the planner breaks such ties by position labels relative to its
// origin, and the build side stores the k-mer at every tied offset (cells_visit).
int harness_anchor_reference(const uint8_t* text, int64_t pos, int k, int* cands) {
    const int W = cell_nw(k);
    Packed P = pack_tile(text, pos, 1, k);
    const bool valid = kmer_valid(P.vb(), P.off, k);
    uint32_t u[32], key[16], best = 0xFFFFFFFFu;
    for (int i = 0; i < W + 10; ++i) {
        uint64_t f, r;
        extract_pair(P.fw(), P.cw(), P.off + i, 6, f, r);
        u[i] = smer6_hash((uint32_t)(f >> 52), (uint32_t)(r >> 52));
    }
    for (int j = 0; j < W; ++j) {
        key[j] = cell_order27(u + j);
        if (key[j] < best) best = key[j];
    }
    int n = 0;
    for (int j = 0; j < W; ++j) {
        if (key[j] != best) continue;
        uint64_t f, r;
        extract_pair(P.fw(), P.cw(), P.off + j, 16, f, r);
        const bool flip = r < f;
        const int off = flip ? W - 1 - j : j;
        cands[n++] = (valid ? 0x80 : 0) | (int)(order_slice_class(best) << 5) | (2 * off) | (flip ? 1 : 0);
    }
    return n;
}
}
