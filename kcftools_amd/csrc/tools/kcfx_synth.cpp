// kcfx_synth -- native generator of bench-scale synthetic inputs for the getVariations hot path.
//
// There is no `kmc` binary and no real data in the build image, so this tool writes
//   <out>.ref.txt   newline-free reference text (uniform ACGT + N-runs + soft-masked blocks + rare IUPAC codes)
//   <out>.fa        the same as a 60-column FASTA (optional, --fasta)
//   <out>.kmc_pre / <out>.kmc_suf   a KMC 0x200 database (layout: KMC.java:107-189) holding
//                   all canonical k-mers of a mutated "sample" genome with depth-like counts, plus --noise
//                   uniformly random k-mers
// from fixed seeds with counter-based hashing (position -> value), so every run reproduces the same bytes.
// It is data tooling, not part of the product compute path, and shares no code with oracle/.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

static inline uint64_t mix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void parallel_for(int64_t n, int threads, const std::function<void(int64_t, int64_t, int)>& fn) {
    if (threads < 1) threads = 1;
    std::vector<std::thread> th;
    const int64_t per = (n + threads - 1) / threads;
    for (int t = 0; t < threads; ++t) {
        const int64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo >= hi) break;
        th.emplace_back(fn, lo, hi, t);
    }
    for (auto& x : th) x.join();
}

// ---- KMC signature (restated from the published KMC "mmer" rules, cf. Signature.java:23-95) -----------
static bool sig_allowed(uint32_t s, int m) {
    if ((s & 0x3f) == 0x3f || (s & 0x3f) == 0x3b || (s & 0x3c) == 0x3c) return false;
    for (int j = 0; j < m - 3; ++j) {
        if ((s & 0xf) == 0) return false;
        s >>= 2;
    }
    return !(s == 0 || s == 4 || (s & 0xf) == 0);
}
static std::vector<uint32_t> make_norm(int m) {
    const uint32_t special = 1u << (2 * m);
    std::vector<uint32_t> norm(special);
    for (uint32_t i = 0; i < special; ++i) {
        uint32_t rev = 0, t = i;
        for (int j = 0; j < m; ++j) { rev = (rev << 2) | (3 - (t & 3)); t >>= 2; }
        const uint32_t a = sig_allowed(i, m) ? i : special, b = sig_allowed(rev, m) ? rev : special;
        norm[i] = std::min(a, b);
    }
    return norm;
}

struct Opts {
    int64_t ref_len = 1000000;
    int64_t ref_offset = 0;                // --ref-offset O (a multiple of 100 000): the reference is bases [O, O + ref_len) of the seed's genome
    bool ref_only = false;                 // --ref-only: write PREFIX.ref.txt (and .fa) and stop: no sample, no database
    uint64_t seed = 1, sample_seed = 2;
    int64_t noise = 0;
    int64_t error_events = 0;              // --error-events: read-error model, see below
    int depth = 0;                         // --depth D: counters of genomic k-mers spread over D / 2 .. 3 D / 2 - 1 (0: 1 .. 30, the older spread)
    int k = 31, p = 7, cs = 2, sig_len = 9, n_bins = 512, threads = 0;
    double sub_rate = 1e-2, indel_rate = 1e-3;
    double repeats = 0.0;                  // --repeats F: fraction of the reference overwritten with repeat structure (see below)
    std::string out;
    bool fasta = false;
    std::string contig = "chr1";
    std::vector<int64_t> contig_lens;      // --contig-lens a,b,c: the reference is cut into contigs chr1..chrN
};

int main(int argc, char** argv) {
    Opts o;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--ref-len") o.ref_len = atoll(val());
        else if (a == "--ref-offset") o.ref_offset = atoll(val());
        else if (a == "--ref-only") o.ref_only = true;
        else if (a == "--seed") o.seed = strtoull(val(), nullptr, 10);
        else if (a == "--sample-seed") o.sample_seed = strtoull(val(), nullptr, 10);
        else if (a == "--noise") o.noise = atoll(val());
        else if (a == "--error-events") o.error_events = atoll(val());
        else if (a == "--depth") o.depth = atoi(val());
        else if (a == "--repeats") o.repeats = atof(val());
        else if (a == "--k") o.k = atoi(val());
        else if (a == "--p") o.p = atoi(val());
        else if (a == "--cs") o.cs = atoi(val());
        else if (a == "--sig-len") o.sig_len = atoi(val());
        else if (a == "--bins") o.n_bins = atoi(val());
        else if (a == "--threads") o.threads = atoi(val());
        else if (a == "--out") o.out = val();
        else if (a == "--fasta") o.fasta = true;
        else if (a == "--contig") o.contig = val();
        else if (a == "--contig-lens") {
            std::string v = val();
            size_t p0 = 0;
            while (p0 < v.size()) {
                size_t c = v.find(',', p0);
                o.contig_lens.push_back(atoll(v.substr(p0, c == std::string::npos ? std::string::npos : c - p0).c_str()));
                if (c == std::string::npos) break;
                p0 = c + 1;
            }
            o.ref_len = 0;
            for (int64_t l : o.contig_lens) o.ref_len += l;
        }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (o.error_events > 0 && o.k > 31) { fprintf(stderr, "--error-events needs k <= 31\n"); return 2; }
    if (o.ref_offset < 0 || o.ref_offset % 100000 != 0 || (o.ref_offset > 0 && o.repeats > 0)) {
        fprintf(stderr, "--ref-offset must be a non-negative multiple of 100000 (and excludes --repeats)\n");
        return 2;
    }
    if (o.out.empty() || o.k < 9 || o.k > 32 || (o.k - o.p) % 4 != 0 || o.cs < 1 || o.cs > 4 || o.sig_len > o.k) {
        fprintf(stderr, "usage: kcfx_synth --out PREFIX [--ref-len N --ref-offset O --ref-only --seed S --sample-seed S --noise N --error-events N --depth D --repeats F --k 31 --p 7 --cs 2 --bins 512 --threads T --fasta]\n");
        return 2;
    }
    if (o.threads <= 0) o.threads = (int)std::max(1u, std::thread::hardware_concurrency());
    const int T = o.threads;
    const int k = o.k;
    const double t0 = now_s();

    // ---- reference ------------------------------------------------------------------------------------
    const int64_t L = o.ref_len;
    if (o.contig_lens.empty()) o.contig_lens.push_back(L);
    std::vector<int64_t> contig_start;     // offsets of the contigs in the concatenated reference text
    {
        int64_t acc = 0;
        for (int64_t l : o.contig_lens) { contig_start.push_back(acc); acc += l; }
    }
    std::vector<uint8_t> ref((size_t)L);
    const int64_t O = o.ref_offset;        // every position-keyed choice below is keyed by the position in the seed's whole genome
    const uint64_t s1 = mix64(o.seed * 0x1000193ull + 11);
    parallel_for(L, T, [&](int64_t lo, int64_t hi, int) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint64_t h = mix64(s1 + (uint64_t)(O + i));
            uint8_t c = "ACGT"[h & 3];
            if ((h >> 8) % 10000 == 0) c = "RYKMSWBDHVN"[(h >> 40) % 11];     // rare IUPAC codes
            ref[(size_t)i] = c;
        }
    });
    // --repeats F: genome structure on top of the uniform text (uniform text has none, and repeats are what crowds minimizers
    // and fills lines of the cell table): a fraction F of the reference is overwritten, in this order, with
    //   * transposable-element families (60 % of F): 48 consensus sequences of 300 bp - 8 kb, copies of 80 - 95 % identity to
    //     their consensus (per-family divergence), half of them 5'-truncated, both orientations, Zipf-like family sizes;
    //   * tandem satellite arrays (20 %): units of 100 - 400 bp repeated to arrays of 10 - 400 kb, 1 - 4 % divergence per copy;
    //   * segmental duplications (15 %): 10 - 100 kb copied from elsewhere in the reference at 98 - 99.5 % identity;
    //   * microsatellites / low-complexity runs (5 %): units of 1 - 6 bp repeated over 50 - 500 bp.
    // The sample genome is derived from the result below, so sample and reference share the structure. Sequential and
    // counter-based: the same seed gives the same bytes.
    int64_t rep_bases[4] = {0, 0, 0, 0};
    if (o.repeats > 0) {
        const uint64_t s4 = mix64(o.seed * 0xC0FFEEull + 77);
        uint64_t ctr = 0;
        auto rnd = [&]() { return mix64(s4 + (ctr++)); };
        auto subst = [&](uint8_t c, uint64_t h) -> uint8_t { return (uint8_t)"ACGT"[((c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : 3) + 1 + (int)(h % 3)) & 3]; };
        auto comp = [](uint8_t c) -> uint8_t { return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : 'A'; };
        const double F = std::min(o.repeats, 0.9);
        // transposable elements
        struct Family { std::vector<uint8_t> cons; double div; };
        std::vector<Family> fam(48);
        for (auto& f : fam) {
            const uint64_t h = rnd();
            f.cons.resize(300 + (size_t)(h % 7700));
            for (auto& c : f.cons) c = (uint8_t)"ACGT"[rnd() & 3];
            f.div = 0.05 + 0.15 * (double)((h >> 32) % 1000) / 1000.0;
        }
        const int64_t te_target = (int64_t)(0.60 * F * (double)L);
        while (rep_bases[0] < te_target) {
            const uint64_t h = rnd();
            // Zipf-like: family index ~ floor(48 * u^2)
            const double u = (double)(h % 1000000) / 1e6;
            const Family& f = fam[(size_t)(48.0 * u * u)];
            size_t len = f.cons.size(), off = 0;
            if ((h >> 20) & 1) {                           // 5'-truncated copy
                off = (size_t)((h >> 24) % (len * 3 / 4));
                len -= off;
            }
            if ((int64_t)len >= L) continue;
            const int64_t at = (int64_t)(rnd() % (uint64_t)(L - (int64_t)len));
            const bool rev = (h >> 21) & 1;
            const uint64_t thr = (uint64_t)(f.div * 4294967296.0);
            for (size_t j = 0; j < len; ++j) {
                uint8_t c = rev ? comp(f.cons[off + len - 1 - j]) : f.cons[off + j];
                const uint64_t g = rnd();
                if ((g & 0xFFFFFFFFull) < thr) c = subst(c, g >> 32);
                ref[(size_t)at + j] = c;
            }
            rep_bases[0] += (int64_t)len;
        }
        // tandem satellite arrays
        const int64_t sat_target = (int64_t)(0.20 * F * (double)L);
        while (rep_bases[1] < sat_target) {
            const uint64_t h = rnd();
            const size_t unit = 100 + (size_t)(h % 300);
            std::vector<uint8_t> u(unit);
            for (auto& c : u) c = (uint8_t)"ACGT"[rnd() & 3];
            int64_t total = 10000 + (int64_t)((h >> 16) % 390000);
            total = std::min<int64_t>(total, L / 4);
            if (total >= L) continue;
            const int64_t at = (int64_t)(rnd() % (uint64_t)(L - total));
            const uint64_t thr = (uint64_t)((0.01 + 0.03 * (double)((h >> 40) % 100) / 100.0) * 4294967296.0);
            for (int64_t j = 0; j < total; ++j) {
                uint8_t c = u[(size_t)(j % (int64_t)unit)];
                const uint64_t g = rnd();
                if ((g & 0xFFFFFFFFull) < thr) c = subst(c, g >> 32);
                ref[(size_t)(at + j)] = c;
            }
            rep_bases[1] += total;
        }
        // segmental duplications
        const int64_t sd_target = (int64_t)(0.15 * F * (double)L);
        while (rep_bases[2] < sd_target && L > 400000) {
            const uint64_t h = rnd();
            const int64_t len = 10000 + (int64_t)(h % 90000);
            const int64_t src = (int64_t)(rnd() % (uint64_t)(L - len)), dst = (int64_t)(rnd() % (uint64_t)(L - len));
            if (std::llabs(src - dst) < len) continue;
            const uint64_t thr = (uint64_t)((0.005 + 0.015 * (double)((h >> 40) % 100) / 100.0) * 4294967296.0);
            for (int64_t j = 0; j < len; ++j) {
                uint8_t c = ref[(size_t)(src + j)] & 0xDF;
                const uint64_t g = rnd();
                if (c == 'A' || c == 'C' || c == 'G' || c == 'T') {
                    if ((g & 0xFFFFFFFFull) < thr) c = subst(c, g >> 32);
                } else {
                    c = (uint8_t)"ACGT"[g & 3];
                }
                ref[(size_t)(dst + j)] = c;
            }
            rep_bases[2] += len;
        }
        // microsatellites
        const int64_t ms_target = (int64_t)(0.05 * F * (double)L);
        while (rep_bases[3] < ms_target) {
            const uint64_t h = rnd();
            const int unit = 1 + (int)(h % 6);
            const int64_t len = 50 + (int64_t)((h >> 8) % 450);
            if (len >= L) break;
            uint8_t u[6];
            for (int j = 0; j < unit; ++j) u[j] = (uint8_t)"ACGT"[(h >> (20 + 2 * j)) & 3];
            const int64_t at = (int64_t)(rnd() % (uint64_t)(L - len));
            for (int64_t j = 0; j < len; ++j) ref[(size_t)(at + j)] = u[j % unit];
            rep_bases[3] += len;
        }
    }
    // (blocks of the whole genome; the one in front of the window may reach into it)
    for (int64_t blk = std::max<int64_t>(0, O / 5000 - 1); blk * 5000 < O + L; ++blk) {            // soft-masked blocks (~10 %)
        const uint64_t h = mix64(s1 ^ (0xABCDull + (uint64_t)blk * 7919));
        if (h % 10 != 0) continue;
        const int64_t s = blk * 5000 + (int64_t)((h >> 8) % 2500) - O, e = std::min(L, s + 2000 + (int64_t)((h >> 24) % 3000));
        for (int64_t i = std::max<int64_t>(s, 0); i < e; ++i) ref[(size_t)i] |= 0x20;
    }
    for (int64_t blk = std::max<int64_t>(0, O / 100000 - 1); blk * 100000 < O + L; ++blk) {        // one N-run per ~100 kb
        const uint64_t h = mix64(s1 ^ (0x5151ull + (uint64_t)blk * 104729));
        const int64_t s = blk * 100000 + (int64_t)((h >> 4) % 100000) - O, e = std::min(L, s + 1 + (int64_t)((h >> 32) % 2000));
        for (int64_t i = std::max<int64_t>(s, 0); i < e && i < L; ++i) ref[(size_t)i] = 'N';
    }
    {
        FILE* f = fopen((o.out + ".ref.txt").c_str(), "wb");
        if (!f) { perror("ref.txt"); return 1; }
        fwrite(ref.data(), 1, ref.size(), f);
        fclose(f);
        if (o.fasta) {
            f = fopen((o.out + ".fa").c_str(), "wb");
            if (!f) { perror("fa"); return 1; }
            for (size_t c = 0; c < o.contig_lens.size(); ++c) {
                if (o.contig_lens.size() == 1) fprintf(f, ">%s synthetic seed=%llu\n", o.contig.c_str(), (unsigned long long)o.seed);
                else fprintf(f, ">chr%zu synthetic seed=%llu\n", c + 1, (unsigned long long)o.seed);
                const int64_t b = contig_start[c], e = b + o.contig_lens[c];
                for (int64_t i = b; i < e; i += 60) {
                    fwrite(ref.data() + i, 1, (size_t)std::min<int64_t>(60, e - i), f);
                    fputc('\n', f);
                }
            }
            fclose(f);
        }
    }
    const double t_ref = now_s();
    if (o.ref_only) {
        printf("{\"ref_len\": %lld, \"ref_offset\": %lld, \"seed\": %llu, \"seconds\": %.3f}\n", (long long)L, (long long)O,
               (unsigned long long)o.seed, t_ref - t0);
        return 0;
    }

    // ---- sample genome: substitutions, 1-10 bp indels, 20 kb deletions (5 per Mb) ----------------------
    std::vector<uint8_t> smp;
    smp.reserve((size_t)L + (size_t)L / 100);
    const uint64_t s2 = mix64(o.sample_seed * 0x9E37ull + 5);
    const uint64_t sub_thr = (uint64_t)(o.sub_rate * 1e6), indel_thr = (uint64_t)(o.indel_rate * 1e6);
    size_t next_contig = 1;
    for (int64_t i = 0; i < L;) {
        if (next_contig < contig_start.size() && i >= contig_start[next_contig]) {
            smp.push_back('#');                 // no sample k-mer spans two contigs
            ++next_contig;
        }
        // one 20 kb deletion per 200 kb block (5 per Mb)
        const int64_t blk0 = (i / 200000) * 200000;
        const int64_t del_s = blk0 + (int64_t)(mix64(s2 ^ (0x77ull + (uint64_t)blk0)) % 180000);
        if (i >= del_s && i < del_s + 20000) { i = del_s + 20000; continue; }
        uint8_t c = ref[(size_t)i] & 0xDF;                  // upper-case; non-letters keep breaking k-mers
        const uint64_t h = mix64(s2 + (uint64_t)i);
        const uint64_t r = h % 1000000;
        const bool acgt = (c == 'A' || c == 'C' || c == 'G' || c == 'T');
        if (acgt && r < sub_thr) {
            const char* alt = (c == 'A') ? "CGT" : (c == 'C') ? "AGT" : (c == 'G') ? "ACT" : "ACG";
            c = (uint8_t)alt[(h >> 32) % 3];
            smp.push_back(c);
            ++i;
        } else if (r < sub_thr + indel_thr) {
            const int len = 1 + (int)((h >> 40) % 10);
            if ((h >> 36) & 1) {
                i += len;                                   // deletion
            } else {
                for (int j = 0; j < len; ++j) smp.push_back("ACGT"[mix64(h + j) & 3]);
                smp.push_back(c);
                ++i;
            }
        } else {
            smp.push_back(c);
            ++i;
        }
    }
    const double t_smp = now_s();

    // ---- k-mers of the sample (canonical, left-aligned words) + noise ------------------------------------
    const int shift = 64 - 2 * k;
    const uint64_t rmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    std::vector<uint64_t> km;
    km.reserve(smp.size() + (size_t)o.noise);
    {
        uint64_t fw = 0, rc = 0;
        int run = 0;
        for (size_t i = 0; i < smp.size(); ++i) {
            int code;
            switch (smp[i]) { case 'A': code = 0; break; case 'C': code = 1; break; case 'G': code = 2; break; case 'T': code = 3; break; default: code = -1; }
            if (code < 0) { run = 0; fw = rc = 0; continue; }
            fw = ((fw << 2) | (uint64_t)code) & rmask;
            rc = (rc >> 2) | ((uint64_t)(3 - code) << (2 * (k - 1)));
            if (++run >= k) km.push_back(std::min(fw, rc) << shift);
        }
    }
    const size_t n_sample_kmers = km.size();
    {
        const size_t base = km.size();
        km.resize(base + (size_t)o.noise);
        const uint64_t s3 = mix64(o.sample_seed * 31 + 0xBEEF);
        parallel_for(o.noise, T, [&](int64_t lo, int64_t hi, int) {
            for (int64_t i = lo; i < hi; ++i) {
                const uint64_t w = mix64(s3 + (uint64_t)i) & rmask;
                uint64_t r = 0, t = w;
                for (int j = 0; j < k; ++j) { r = (r << 2) | (3 - (t & 3)); t >>= 2; }
                km[base + (size_t)i] = std::min(w, r) << shift;
            }
        });
    }
    // --error-events E: what a k-mer count of sequencing reads (kmc -ci1) contains beside the genome -- E substitution
    // errors at random sample positions, each contributing the (up to k) k-mers that cover the wrong base. They come in
    // runs that share minimizers with their genomic neighbours and are seen once (counter 1), while genomic k-mers are seen
    // "depth" times. Marked by bit 0 of the word (free for k <= 31) until the counts are assigned.
    if (o.error_events > 0) {
        const uint64_t s5 = mix64(o.sample_seed * 131 + 0xE44);
        std::vector<std::vector<uint64_t>> part((size_t)T);
        parallel_for(o.error_events, T, [&](int64_t lo, int64_t hi, int t) {
            std::vector<uint64_t>& out = part[(size_t)t];
            auto code_of = [](uint8_t c) -> int { return c == 'A' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : -1; };
            for (int64_t e = lo; e < hi; ++e) {
                const uint64_t h = mix64(s5 + (uint64_t)e);
                const int64_t x = (int64_t)(h % smp.size());
                const int orig = code_of(smp[(size_t)x]);
                if (orig < 0) continue;
                const int wrong = (orig + 1 + (int)((h >> 40) % 3)) & 3;
                for (int64_t st = std::max<int64_t>(0, x - k + 1); st <= x && st + k <= (int64_t)smp.size(); ++st) {
                    uint64_t fw = 0, rc = 0;
                    bool ok = true;
                    for (int j = 0; j < k; ++j) {
                        const int c = (st + j == x) ? wrong : code_of(smp[(size_t)(st + j)]);
                        if (c < 0) { ok = false; break; }
                        fw = (fw << 2) | (uint64_t)c;
                        rc |= (uint64_t)(3 - c) << (2 * j);
                    }
                    if (ok) out.push_back((std::min(fw, rc) << shift) | 1ull);
                }
            }
        });
        for (auto& v : part) km.insert(km.end(), v.begin(), v.end());
    }
    const size_t n_error_kmers = km.size() - n_sample_kmers - (size_t)o.noise;
    const double t_km = now_s();

    // ---- sort + unique + counts (256 top-byte buckets, sorted in parallel) -------------------------------
    const size_t N0 = km.size();
    std::vector<size_t> boff(257, 0);
    for (size_t i = 0; i < N0; ++i) boff[(km[i] >> 56) + 1]++;
    for (int b = 0; b < 256; ++b) boff[b + 1] += boff[b];
    std::vector<uint64_t> sorted(N0);
    {
        std::vector<size_t> cur(boff.begin(), boff.end() - 1);
        for (size_t i = 0; i < N0; ++i) sorted[cur[km[i] >> 56]++] = km[i];
    }
    km.clear();
    km.shrink_to_fit();
    std::vector<size_t> ucount(256, 0);
    std::vector<uint32_t> occ(N0);
    {
        std::atomic<int> next(0);
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t)
            th.emplace_back([&]() {
                for (;;) {
                    const int b = next.fetch_add(1);
                    if (b >= 256) break;
                    uint64_t* lo = sorted.data() + boff[b];
                    uint64_t* hi = sorted.data() + boff[b + 1];
                    std::sort(lo, hi);
                    size_t w = boff[b];
                    for (uint64_t* p = lo; p < hi;) {
                        // (an error k-mer that equals a genomic one sorts right behind it: one group, genomic)
                        const uint64_t word = *p & ~1ull;
                        const bool only_errors = (*p & 1ull) != 0;
                        uint64_t* q = p;
                        while (q < hi && (*q & ~1ull) == word) ++q;
                        sorted[w] = word | (only_errors ? 1ull : 0ull);
                        occ[w] = (uint32_t)(q - p);
                        ++w;
                        p = q;
                    }
                    ucount[b] = w - boff[b];
                }
            });
        for (auto& x : th) x.join();
    }
    size_t N = 0;
    std::vector<size_t> uoff(257, 0);
    for (int b = 0; b < 256; ++b) { uoff[b + 1] = uoff[b] + ucount[b]; }
    N = uoff[256];
    const double t_sort = now_s();

    // ---- signature -> bin; stable distribution into bins keeps each bin sorted by k-mer --------------------
    const int m = o.sig_len;
    const std::vector<uint32_t> norm = make_norm(m);
    const uint32_t mmask = (1u << (2 * m)) - 1;
    const uint32_t cmax = (o.cs >= 4) ? 0x7FFFFFFFu : ((1u << (8 * o.cs)) - 1u);
    std::vector<uint64_t> uk(N);
    std::vector<uint32_t> uc(N);
    std::vector<uint16_t> ubin(N);
    parallel_for(256, T, [&](int64_t lo, int64_t hi, int) {
        for (int64_t b = lo; b < hi; ++b) {
            for (size_t j = 0; j < ucount[b]; ++j) {
                const bool only_errors = (sorted[boff[b] + j] & 1ull) != 0 && o.error_events > 0;
                const uint64_t w = o.error_events > 0 ? (sorted[boff[b] + j] & ~1ull) : sorted[boff[b] + j];
                const size_t d = uoff[b] + j;
                uk[d] = w;
                const uint64_t h = mix64(w ^ 0xC0FFEEull);
                // depth-like spread for -c and KD; a k-mer that only read errors produced is seen as often as they occurred.
                // (--depth D: D / 2 .. 3 D / 2 - 1 -- a count of reads that holds E error k-mers per genomic k-mer was sequenced to a
                // depth of about E / (k x error rate), 4.5 M events on 27 M k-mers at 0.5 % errors: ~30 x, where a genomic k-mer
                // seen ONCE does not occur; the older 1 .. 30 makes one in thirty of them indistinguishable from an error)
                const uint64_t spread = o.depth >= 4 ? (uint64_t)(o.depth / 2) + h % (uint64_t)o.depth : 1 + h % 30;
                uint64_t c = only_errors ? (uint64_t)occ[boff[b] + j] : (uint64_t)occ[boff[b] + j] * spread;
                uc[d] = (uint32_t)std::min<uint64_t>(c, cmax);
                uint32_t best = 1u << (2 * m);
                for (int i = 0; i + m <= k; ++i) {
                    const uint32_t mm = (uint32_t)(w >> (64 - 2 * (i + m))) & mmask;
                    best = std::min(best, norm[mm]);
                }
                ubin[d] = (uint16_t)((((uint64_t)best * 2654435761ull) >> 7) % (uint64_t)o.n_bins);
            }
        }
    });
    sorted.clear(); sorted.shrink_to_fit();
    occ.clear(); occ.shrink_to_fit();
    std::vector<size_t> binoff((size_t)o.n_bins + 1, 0);
    for (size_t i = 0; i < N; ++i) binoff[ubin[i] + 1]++;
    for (int b = 0; b < o.n_bins; ++b) binoff[b + 1] += binoff[b];
    const int sb = (k - o.p) / 4, rs = sb + o.cs;
    std::vector<uint8_t> rec(N * (size_t)rs);
    const int64_t lut_size = 1ll << (2 * o.p);
    std::vector<uint64_t> lut((size_t)o.n_bins * (size_t)lut_size, ~0ull);
    {
        std::vector<size_t> cur(binoff.begin(), binoff.end() - 1);
        for (size_t i = 0; i < N; ++i) {
            const size_t d = cur[ubin[i]]++;
            const uint64_t w = uk[i];
            const uint64_t suf = o.p ? (w << (2 * o.p)) : w;
            uint8_t* r = rec.data() + d * rs;
            for (int j = 0; j < sb; ++j) r[j] = (uint8_t)(suf >> (56 - 8 * j));
            for (int j = 0; j < o.cs; ++j) r[sb + j] = (uint8_t)(uc[i] >> (8 * j));
            const uint64_t pre = o.p ? (w >> (64 - 2 * o.p)) : 0;
            uint64_t& slot = lut[(size_t)ubin[i] * (size_t)lut_size + (size_t)pre];
            if (slot == ~0ull) slot = d;
        }
        uint64_t nxt = N;
        for (size_t i = lut.size(); i-- > 0;) {
            if (lut[i] == ~0ull) lut[i] = nxt; else nxt = lut[i];
        }
    }
    const double t_bin = now_s();

    // ---- write the database ----------------------------------------------------------------------------------
    {
        FILE* f = fopen((o.out + ".kmc_suf").c_str(), "wb");
        if (!f) { perror("kmc_suf"); return 1; }
        fwrite("KMCS", 1, 4, f);
        fwrite(rec.data(), 1, rec.size(), f);
        fwrite("KMCS", 1, 4, f);
        fclose(f);
        f = fopen((o.out + ".kmc_pre").c_str(), "wb");
        if (!f) { perror("kmc_pre"); return 1; }
        fwrite("KMCP", 1, 4, f);
        fwrite(lut.data(), 8, lut.size(), f);
        const uint64_t guard = N;
        fwrite(&guard, 8, 1, f);
        std::vector<uint32_t> smap(((size_t)1 << (2 * m)) + 1);
        for (size_t s = 0; s < smap.size(); ++s) smap[s] = (uint32_t)((((uint64_t)s * 2654435761ull) >> 7) % (uint64_t)o.n_bins);
        fwrite(smap.data(), 4, smap.size(), f);
        uint8_t hdr[68];
        memset(hdr, 0, sizeof hdr);
        const uint32_t f32[7] = {(uint32_t)k, 0u, (uint32_t)o.cs, (uint32_t)o.p, (uint32_t)m, 1u, cmax};
        memcpy(hdr, f32, 28);
        const uint64_t total = N;
        memcpy(hdr + 28, &total, 8);
        hdr[36] = 0;                                   // both strands (canonical) -- KMC.java:133
        const uint32_t ver = 0x200;
        memcpy(hdr + 64, &ver, 4);
        fwrite(hdr, 1, 68, f);
        const uint32_t hoff = 68;
        fwrite(&hoff, 4, 1, f);
        fwrite("KMCP", 1, 4, f);
        fclose(f);
    }
    const double t_end = now_s();
    printf("{\"ref_len\": %lld, \"sample_len\": %zu, \"sample_kmers\": %zu, \"noise_kmers\": %lld, \"error_kmers\": %zu, \"db_kmers\": %zu, "
           "\"repeat_bases\": {\"te\": %lld, \"satellite\": %lld, \"segdup\": %lld, \"microsat\": %lld}, \"k\": %d, \"p\": %d, \"counter_size\": %d, \"n_bins\": %d, \"seed\": %llu, \"sample_seed\": %llu, \"threads\": %d, "
           "\"seconds\": {\"ref\": %.2f, \"sample\": %.2f, \"kmers\": %.2f, \"sort\": %.2f, \"bins\": %.2f, \"write\": %.2f, \"total\": %.2f}}\n",
           (long long)L, smp.size(), n_sample_kmers, (long long)o.noise, n_error_kmers, N, (long long)rep_bases[0], (long long)rep_bases[1],
           (long long)rep_bases[2], (long long)rep_bases[3], k, o.p, o.cs, o.n_bins,
           (unsigned long long)o.seed, (unsigned long long)o.sample_seed, T, t_ref - t0, t_smp - t_ref, t_km - t_smp,
           t_sort - t_km, t_bin - t_sort, t_end - t_bin, t_end - t0);
    return 0;
}
