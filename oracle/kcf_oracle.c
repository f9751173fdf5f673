/*
 * kcf_oracle.c -- CPU oracle for the kcftools `getVariations` hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see kcf_oracle.h).  PARITY UNPINNED: no reference test,
 * golden vector or runnable reference exists for this path in this image; this file is a
 * literal restatement of the cited Java lines, checked against SURVEY.md section 8(c) KATs
 * and an independent Python model.
 *
 * Citations are relative to /root/reference/src/main/java/nl/wur/bis/kcftools/.
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no dependencies besides pthread).
 */
#define _FILE_OFFSET_BITS 64
#define _POSIX_C_SOURCE 200809L   /* fseeko / ftello under -std=c11: without the prototype they return int and a .kmc_suf over 2 GiB looks truncated */
#include "kcf_oracle.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static __thread char g_err[512];
const char* kcfo_last_error(void) { return g_err; }
static int fail(const char* msg, const char* arg) {
    snprintf(g_err, sizeof g_err, "%s%s", msg, arg ? arg : "");
    return 1;
}

/* ------------------------------------------------------------------------------------
 * Signature.java
 * ---------------------------------------------------------------------------------- */

/* Signature.java:42-76 isAllowed */
int kcfo_sig_is_allowed(int32_t signature, int m) {
    if ((signature & 0x3F) == 0x3F) return 0; /* :44 */
    if ((signature & 0x3F) == 0x3B) return 0; /* :47 */
    if ((signature & 0x3C) == 0x3C) return 0; /* :50 */
    for (int j = 0; j < m - 3; ++j) {         /* :55-62 */
        if ((signature & 0xF) == 0) return 0;
        signature >>= 2;
    }
    if (signature == 0) return 0;             /* :65 */
    if (signature == 0x04) return 0;          /* :68 */
    if ((signature & 0xF) == 0) return 0;     /* :71 */
    return 1;
}

/* Signature.java:82-95 getRev */
int32_t kcfo_sig_rev(int32_t sequence, int length) {
    int32_t rc = 0;
    for (int i = 0; i < length; i++) {
        int32_t base = sequence & 3;
        base = (~base) & 3;
        rc = (rc << 2) | base;
        sequence >>= 2;
    }
    return rc;
}

/* Signature.java:23-37 initNorm */
void kcfo_norm_table(int m, int32_t* out) {
    int32_t special = 1 << (m * 2);
    for (int32_t i = 0; i < special; ++i) {
        int32_t rev = kcfo_sig_rev(i, m);
        int32_t str_val = kcfo_sig_is_allowed(i, m) ? i : special;
        int32_t rev_val = kcfo_sig_is_allowed(rev, m) ? rev : special;
        out[i] = str_val < rev_val ? str_val : rev_val;
    }
}

/* ------------------------------------------------------------------------------------
 * Kmer.java (k <= 32: the long[] has exactly one word)
 * ---------------------------------------------------------------------------------- */

/* Kmer.java:286-294 baseToBits */
static int base_to_bits(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}

/* Kmer.java:232-252 kmerToLong: base i at bits [63-2i, 62-2i] */
int kcfo_kmer_encode(const char* s, int k, uint64_t* out) {
    uint64_t w = 0;
    for (int i = 0; i < k; i++) {
        int v = base_to_bits(s[i]);
        if (v < 0) return -1;
        w |= ((uint64_t)v) << (62 - 2 * i);
    }
    *out = w;
    return 0;
}

/* Kmer.java:300-338 getReverseComplement: rc base (k-1-i) = ~base(i) */
uint64_t kcfo_kmer_revcomp(uint64_t w, int k) {
    uint64_t rev = 0;
    for (int i = 0; i < k; i++) {
        int bits = (int)((w >> (62 - 2 * i)) & 3);
        int comp = (~bits) & 3;
        int rev_index = (k - i - 1) * 2;
        rev |= ((uint64_t)comp) << (62 - rev_index);
    }
    return rev;
}

/* Kmer.java:72-79 getCanonical; :406-414 compareLongArrays (Long.compareUnsigned) */
uint64_t kcfo_kmer_canonical(uint64_t w, int k) {
    uint64_t rc = kcfo_kmer_revcomp(w, k);
    return (w > rc) ? rc : w;
}

/* Kmer.java:208-226 extractIntFromBits (single word) */
static int32_t extract_int(uint64_t w, int start_base, int length_bases) {
    int32_t result = 0;
    for (int i = 0; i < length_bases; i++) {
        int bit_index = (start_base + i) * 2;
        int base_bits = (int)((w >> (62 - bit_index)) & 3);
        result = (result << 2) | base_bits;
    }
    return result;
}

/* Kmer.java:105-118 getSignature */
int32_t kcfo_kmer_signature(uint64_t w, int k, int m, const int32_t* norm) {
    int32_t cur = extract_int(w, 0, m);
    int32_t min_sig = norm[cur];
    for (int i = 1; i <= k - m; i++) {
        cur = ((cur << 2) & ((1 << (2 * m)) - 1)) | extract_int(w, i + m - 1, 1);
        if (norm[cur] < min_sig) min_sig = norm[cur];
    }
    return min_sig;
}

/* Kmer.java:123-128 getPrefixFwd */
int32_t kcfo_kmer_prefix(uint64_t w, int p) { return extract_int(w, 0, p); }

/* Kmer.java:143-170 extractSuffix: suffix base i at bits (3 - i%4)*2 of byte i/4 */
void kcfo_kmer_suffix(uint64_t w, int k, int p, uint8_t* out) {
    int suffix_length = k - p;
    memset(out, 0, (size_t)(suffix_length + 3) / 4);
    for (int i = 0; i < suffix_length; ++i) {
        int bit_index = 2 * p + i * 2;
        int base_bits = (int)((w >> (62 - bit_index)) & 3);
        out[i / 4] |= (uint8_t)(base_bits << ((3 - i % 4) * 2));
    }
}

/* ------------------------------------------------------------------------------------
 * KMC.java
 * ---------------------------------------------------------------------------------- */

struct kcfo_db {
    kcfo_info info;
    int32_t suffix_bytes;       /* sufixLength / 4  KMC.java:61 */
    int32_t record_size;        /* KMC.java:61 */
    int32_t lut_prefix_array_size; /* 4^p KMC.java:154 */
    uint64_t* prefix_array;     /* KMC.java:160-163 */
    int64_t prefix_array_len;
    int32_t* signature_map;     /* KMC.java:148-151 */
    int64_t signature_map_len;
    int32_t* norm;              /* Signature(signatureLength) KMC.java:60 */
    uint8_t* suf;               /* records, after the 4-byte KMCS marker KMC.java:94,179 */
    int64_t suf_bytes;
};

static uint32_t rd_u32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static uint64_t rd_u64(const uint8_t* p) { return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32); }

static uint8_t* read_file(const char* path, int64_t* size) {
    FILE* f = fopen(path, "rb");
    if (!f) return NULL;
    fseeko(f, 0, SEEK_END);
    int64_t n = ftello(f);
    fseeko(f, 0, SEEK_SET);
    uint8_t* buf = (uint8_t*)malloc(n > 0 ? (size_t)n : 1);
    if (!buf) { fclose(f); return NULL; }
    if (n > 0 && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); fclose(f); return NULL; }
    fclose(f);
    *size = n;
    return buf;
}

/* KMC.java:56-78 ctor, :107-168 readPrefixFile, :84-102 preloadSuffixBuffers */
int kcfo_db_open(const char* prefix, kcfo_db** out) {
    char path[4096];
    *out = NULL;
    snprintf(path, sizeof path, "%s.kmc_pre", prefix);
    int64_t file_size = 0;
    uint8_t* pre = read_file(path, &file_size);
    if (!pre) return fail("Error reading prefix file ", path);           /* :166 */
    if (file_size < 8 + 4) { free(pre); return fail("prefix file too small ", path); }

    kcfo_db* db = (kcfo_db*)calloc(1, sizeof *db);
    int64_t header_offset = (int32_t)rd_u32(pre + file_size - 8);         /* :117-118 */
    int64_t pos = file_size - header_offset - 8;                          /* :121 */
    if (pos < 4 || pos + 68 > file_size) { free(pre); free(db); return fail("bad header offset in ", path); }
    const uint8_t* h = pre + pos;
    db->info.kmer_length = (int32_t)rd_u32(h + 0);                        /* :124 */
    db->info.mode = (int32_t)rd_u32(h + 4);                               /* :125 */
    db->info.counter_size = (int32_t)rd_u32(h + 8);                       /* :126 */
    db->info.lut_prefix_length = (int32_t)rd_u32(h + 12);                 /* :127 */
    db->info.signature_length = (int32_t)rd_u32(h + 16);                  /* :129 */
    db->info.min_count = (int32_t)rd_u32(h + 20);                         /* :130 */
    db->info.max_count = (int32_t)rd_u32(h + 24);                         /* :131 */
    db->info.total_kmers = (int64_t)rd_u64(h + 28);                       /* :132 */
    db->info.both_strands = (h[36] == 0);                                 /* :133 */
    /* :135 skip 3, :137 skip 24 */
    db->info.version = (int32_t)rd_u32(h + 36 + 1 + 3 + 24);              /* :139 */
    if (db->info.version != 0x200) {                                      /* :140-142 */
        free(pre); free(db);
        return fail("KMC version is not 0x200", NULL);
    }
    int k = db->info.kmer_length, p = db->info.lut_prefix_length, m = db->info.signature_length;
    if (k < 1 || k > 32) { free(pre); free(db); return fail("oracle supports 1 <= k <= 32", NULL); }
    if (p < 0 || p > k || p > 15 || m < 1 || m > 12) { free(pre); free(db); return fail("bad p / signature length", NULL); }

    int64_t signature_map_size = (1LL << (2 * m)) + 1;                    /* :145 */
    int64_t signature_map_start = file_size - header_offset - 8 - signature_map_size * 4; /* :146 */
    if (signature_map_start < 4 + 8) { free(pre); free(db); return fail("prefix file too small for signature map", NULL); }
    db->signature_map = (int32_t*)malloc((size_t)signature_map_size * 4);
    db->signature_map_len = signature_map_size;
    for (int64_t i = 0; i < signature_map_size; i++)                      /* :149-151 */
        db->signature_map[i] = (int32_t)rd_u32(pre + signature_map_start + 4 * i);

    db->lut_prefix_array_size = 1 << (2 * p);                             /* :154 */
    int64_t single_lut_size = (int64_t)db->lut_prefix_array_size * 8;     /* :155 */
    int32_t num_prefix_arrays = (int32_t)((signature_map_start - 8 - 4) / (int32_t)single_lut_size); /* :156 */
    db->info.n_bins = num_prefix_arrays;
    db->prefix_array_len = (int64_t)num_prefix_arrays * db->lut_prefix_array_size; /* :160 */
    db->info.lut_entries = db->prefix_array_len;
    db->prefix_array = (uint64_t*)malloc((size_t)(db->prefix_array_len ? db->prefix_array_len : 1) * 8);
    for (int64_t i = 0; i < db->prefix_array_len; i++)                    /* :161-163, start = 4 (:153) */
        db->prefix_array[i] = rd_u64(pre + 4 + 8 * i);
    free(pre);

    db->suffix_bytes = (k - p) / 4;                                       /* sufixLength / 4 */
    db->record_size = db->info.counter_size + (k - p) / 4;                /* :61 */
    if ((k - p) % 4 != 0) {
        /* Java: compareByteArray (HelperFunctions.java:232-243) would index past the shorter array */
        kcfo_db_close(db);
        return fail("(k - lut_prefix_length) must be a multiple of 4", NULL);
    }
    db->norm = (int32_t*)malloc(((size_t)1 << (2 * m)) * 4);              /* :60 */
    kcfo_norm_table(m, db->norm);

    snprintf(path, sizeof path, "%s.kmc_suf", prefix);
    int64_t suf_size = 0;
    uint8_t* suf = read_file(path, &suf_size);
    if (!suf) { kcfo_db_close(db); return fail("Error reading suffix buffers from file ", path); } /* :187 */
    int64_t need = 4 + db->info.total_kmers * db->record_size;            /* :94 seek(4); :96 pages */
    if (suf_size < need) { free(suf); kcfo_db_close(db); return fail("suffix file shorter than total_kmers * record_size: ", path); }
    db->suf = suf;
    db->suf_bytes = suf_size;
    *out = db;
    return 0;
}

void kcfo_db_close(kcfo_db* db) {
    if (!db) return;
    free(db->prefix_array);
    free(db->signature_map);
    free(db->norm);
    free(db->suf);
    free(db);
}

void kcfo_db_info(const kcfo_db* db, kcfo_info* info) { *info = db->info; }

/* KMC.java:366-381 getEntry: the 16 MiB paging is an implementation detail; record `index` starts
 * at byte 4 + index * record_size of the .kmc_suf file. */
static const uint8_t* get_entry(const kcfo_db* db, int64_t index) { return db->suf + 4 + index * db->record_size; }

/* KMC.java:395-401 getCountFromEntry (Java int: counter_size 4 with the top bit set is negative) */
static int32_t count_from_entry(const kcfo_db* db, const uint8_t* entry) {
    uint32_t count = 0;
    for (int i = 0; i < db->info.counter_size; i++) count |= ((uint32_t)entry[db->suffix_bytes + i]) << (i * 8);
    return (int32_t)count;
}

/* HelperFunctions.java:232-243 compareByteArray */
static int compare_bytes(const uint8_t* a, const uint8_t* b, int n) {
    int i;
    for (i = 0; i < n; ++i)
        if (a[i] != b[i]) break;
    if (i == n) return 0;
    return (a[i] < b[i]) ? -1 : 1;
}

/* KMC.java:292-326 getCount */
int32_t kcfo_get_count(const kcfo_db* db, uint64_t kmer) {
    int k = db->info.kmer_length, p = db->info.lut_prefix_length, m = db->info.signature_length;
    int32_t signature = kcfo_kmer_signature(kmer, k, m, db->norm);         /* :295 */
    int32_t prefix = kcfo_kmer_prefix(kmer, p);                            /* :296 */
    uint8_t suffix[16];
    kcfo_kmer_suffix(kmer, k, p, suffix);                                  /* :297 */

    int64_t signature_index = (int64_t)db->signature_map[signature] * db->lut_prefix_array_size; /* :300 */
    int64_t start = (int64_t)db->prefix_array[signature_index + prefix];   /* :301 */
    int64_t end;
    if (signature_index + prefix + 1 >= db->prefix_array_len)              /* :302 */
        end = db->info.total_kmers - 1;
    else
        end = (int64_t)db->prefix_array[signature_index + prefix + 1] - 1; /* :306 */

    while (start <= end) {                                                 /* :310-323 */
        int64_t mid = (start + end) / 2;
        const uint8_t* entry = get_entry(db, mid);
        int cmp = compare_bytes(suffix, entry, db->suffix_bytes);
        if (cmp < 0) end = mid - 1;
        else if (cmp > 0) start = mid + 1;
        else return count_from_entry(db, entry);
    }
    return 0;
}

/* KMC.java:427-450 dumpKmerTable: k-mer = prefix bases (i mod 4^p) || suffix bases */
int64_t kcfo_dump(const kcfo_db* db, uint64_t* kmers, int32_t* counts) {
    int k = db->info.kmer_length, p = db->info.lut_prefix_length;
    int64_t n = 0;
    for (int64_t i = 0; i < db->prefix_array_len; i++) {
        int64_t start = (int64_t)db->prefix_array[i];
        int64_t end = (i == db->prefix_array_len - 1) ? db->info.total_kmers - 1 : (int64_t)db->prefix_array[i + 1] - 1;
        uint64_t prefix = (uint64_t)(i % db->lut_prefix_array_size);
        for (int64_t j = start; j <= end; j++) {
            const uint8_t* e = get_entry(db, j);
            uint64_t w = p ? (prefix << (64 - 2 * p)) : 0;
            for (int b = 0; b < k - p; b++) {
                uint64_t base = (e[b / 4] >> (6 - 2 * (b % 4))) & 3;        /* :342 */
                w |= base << (62 - 2 * (p + b));
            }
            if (kmers) kmers[n] = w;
            if (counts) counts[n] = count_from_entry(db, e);
            n++;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------
 * Fasta.java
 * ---------------------------------------------------------------------------------- */

/* Character.toUpperCase restricted to what can reach isValidBase: only ASCII letters can map onto A/C/G/T */
static char to_upper(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }
/* Fasta.java:132-134 */
static int is_valid_base(char b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }

/* Fasta.java:90-127 getKmersList(kmerLength, prefixLength, useCanonical=false) */
int32_t kcfo_kmers_list(const char* seq, int32_t len, int k, uint64_t* out) {
    char kmer_chars[33];
    int have_kmer = 0;     /* kmer != null */
    int32_t n = 0;
    int32_t valid_start = 0;
    uint64_t w = 0;
    memset(kmer_chars, 0, sizeof kmer_chars);
    for (int32_t i = 0; i < len; i++) {
        char base = to_upper(seq[i]);                   /* :98 */
        if (!is_valid_base(base)) {                     /* :99-104 */
            valid_start = i + 1;
            have_kmer = 0;
            memset(kmer_chars, 0, sizeof kmer_chars);
            continue;
        }
        int32_t offset = i - valid_start;               /* :106 */
        if (offset < k) {                               /* :107-111 */
            kmer_chars[offset] = base;
            if (offset == k - 1) {
                kcfo_kmer_encode(kmer_chars, k, &w);
                have_kmer = 1;
            }
        } else if (have_kmer) {                         /* :112-119 */
            memmove(kmer_chars, kmer_chars + 1, (size_t)(k - 1));
            kmer_chars[k - 1] = base;
            kcfo_kmer_encode(kmer_chars, k, &w);
        }
        if (have_kmer) {                                /* :121-123 */
            if (out) out[n] = w;
            n++;
        }
    }
    return n;
}

/* Fasta.java:140-167 getEffectiveATGCCount */
int32_t kcfo_eff_length(const char* seq, int32_t len, int k) {
    int32_t count = 0, stretch = 0;
    for (int32_t i = 0; i < len; i++) {
        char base = to_upper(seq[i]);
        if (base == 'A' || base == 'T' || base == 'G' || base == 'C') {
            stretch++;
        } else {
            if (stretch >= k) count += stretch;
            stretch = 0;
        }
    }
    if (stretch >= k) count += stretch;
    return count;
}

/* ------------------------------------------------------------------------------------
 * GetVariants.java
 * ---------------------------------------------------------------------------------- */

/* GetVariants.java:267-273 getDistance */
int32_t kcfo_distance(int k, int32_t gap) {
    int32_t distance = gap - (k - 1);
    if (distance <= 0) distance = abs(distance + 1);
    return distance;
}

/* GetVariants.java:203-252: the state machine, over the counts of the valid k-mers in order */
void kcfo_accumulate(const int32_t* counts, int32_t n, int k, int32_t min_count, kcfo_result* out) {
    int32_t total = 0, obs = 0, var = 0, inner = 0, gap = 0, left = 0, right = 0;
    int is_tail = 1;
    int64_t kmer_count = 0;
    for (int32_t i = 0; i < n; i++) {                  /* :220 */
        total++;                                       /* :221 */
        int32_t c = counts[i];                         /* :223 */
        if (c >= min_count) {                          /* :224 */
            kmer_count += c;                           /* :226 */
            obs++;                                     /* :227 */
            if (gap > 0) {                             /* :228 */
                var++;                                 /* :230 */
                if (is_tail) left += gap;              /* :232-234 */
                else inner += kcfo_distance(k, gap);   /* :237 */
            }
            is_tail = 0;                               /* :240 */
            gap = 0;                                   /* :241 */
        } else {
            gap++;                                     /* :243 */
        }
    }
    if (gap > 0) {                                     /* :248-251 */
        var++;
        right += gap;
    }
    out->total_kmers = total;
    out->obs = obs;
    out->var = var;
    out->inner = inner;
    out->left = left;
    out->right = right;
    out->kmer_count_sum = kmer_count;
}

/* GetVariants.java:202-261 processWindow, on scratch the caller owns (a worker thread keeps its two arrays across windows:
 * the JVM's allocator is thread-local too, and two malloc / free pairs of ~100 KB per window are not part of the algorithm) */
typedef struct {
    uint64_t* kmers;
    int32_t* counts;
    int32_t cap;
} kcfo_scratch;

static int scratch_reserve(kcfo_scratch* s, int32_t cap) {
    if (cap <= s->cap) return 0;
    free(s->kmers);
    free(s->counts);
    s->kmers = (uint64_t*)malloc((size_t)cap * 8);
    s->counts = (int32_t*)malloc((size_t)cap * 4);
    s->cap = (s->kmers && s->counts) ? cap : 0;
    return s->cap ? 0 : 1;
}

static void scratch_free(kcfo_scratch* s) {
    free(s->kmers);
    free(s->counts);
    s->kmers = NULL;
    s->counts = NULL;
    s->cap = 0;
}

static int process_window_on(const kcfo_db* db, const char* seq, int32_t len, int32_t min_count, kcfo_result* out, kcfo_scratch* s) {
    int k = db->info.kmer_length;
    memset(out, 0, sizeof *out);
    int32_t cap = len - k + 1;
    if (cap < 1) cap = 1;
    if (scratch_reserve(s, cap)) return fail("out of memory", NULL);
    int32_t n = kcfo_kmers_list(seq, len, k, s->kmers);                    /* :217 */
    for (int32_t i = 0; i < n; i++) {
        uint64_t km = db->info.both_strands ? kcfo_kmer_canonical(s->kmers[i], k) : s->kmers[i]; /* :222 */
        s->counts[i] = kcfo_get_count(db, km);                             /* :223 */
    }
    kcfo_accumulate(s->counts, n, k, min_count, out);                      /* :219-252 */
    out->eff_length = kcfo_eff_length(seq, len, k);                        /* :256 */
    return 0;
}

int kcfo_process_window(const kcfo_db* db, const char* seq, int32_t len, int32_t min_count, kcfo_result* out) {
    kcfo_scratch s = {NULL, NULL, 0};
    int rc = process_window_on(db, seq, len, min_count, out, &s);
    scratch_free(&s);
    return rc;
}

/* Data.java:87 (meanKmerCount) and :95-107 (computeScore); weights = {wi, wt, wr} (GetVariants.java:388-390) */
int kcfo_score(const kcfo_result* r, double wi, double wt, double wr, double* score, double* mean_count) {
    double weights[3] = {wi, wt, wr};
    if (mean_count) *mean_count = (r->obs > 0) ? (double)r->kmer_count_sum / r->obs : 0.00;  /* :87 */
    if (r->obs == 0 || r->total_kmers == 0 || r->eff_length == 0) {                            /* :96-98 */
        if (score) *score = 0;
        return 0;
    }
    int bad = (weights[0] + weights[1] + weights[2] != 1.0);                                  /* :101-103 */
    int32_t tail = r->left + r->right;                                                        /* :170-172 */
    double a = weights[2] * ((double)r->obs / r->total_kmers);                                /* :104 */
    double b = weights[0] * (1.0f - ((double)r->inner / r->eff_length));                      /* :105 */
    double c = weights[1] * (1.0f - ((double)tail / r->eff_length));                          /* :106 */
    double s = a + b;
    s = s + c;
    s = s * 100.0f;
    if (score) *score = s;
    return bad;
}

/* GetVariants.java:293-320 getWindows, case "window" */
int64_t kcfo_windows(int32_t seq_len, int32_t window, int32_t step, int k, int32_t* starts, int32_t* ends, int64_t cap) {
    int64_t n = 0;
    if (step > 0) {                                           /* :295-306 sliding */
        int64_t last_pos = 0;
        while (last_pos < seq_len) {
            int64_t start = last_pos;
            int64_t end = start + window < seq_len ? start + window : seq_len;
            if (end - start >= k) {
                if (n < cap) { if (starts) starts[n] = (int32_t)start; if (ends) ends[n] = (int32_t)end; }
                n++;
            }
            last_pos += step;
        }
    } else {                                                  /* :307-320 tiling */
        int32_t last_end = 0;
        while (last_end < seq_len) {
            int32_t start = last_end - k + 1 > 0 ? last_end - k + 1 : 0;
            int32_t end = start + window < seq_len ? start + window : seq_len;
            if (end - start >= k) {
                if (n < cap) { if (starts) starts[n] = start; if (ends) ends[n] = end; }
                n++;
            }
            if (end <= last_end) return -1;                   /* reference would loop forever (window <= k-1) */
            last_end = end;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------------------
 * batch driver (GetVariants.java:129-159: fixed thread pool, one task per window)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    const kcfo_db* db;
    const char* text;
    const int64_t* win_begin;
    const int32_t* win_len;
    int32_t n_windows, min_count;
    kcfo_result* out;
    volatile int32_t* next;
    int rc;
} job_t;

static void* worker(void* arg) {
    job_t* j = (job_t*)arg;
    kcfo_scratch s = {NULL, NULL, 0};
    for (;;) {
        int32_t w = __sync_fetch_and_add(j->next, 1);
        if (w >= j->n_windows) break;
        if (process_window_on(j->db, j->text + j->win_begin[w], j->win_len[w], j->min_count, &j->out[w], &s)) j->rc = 1;
    }
    scratch_free(&s);
    return NULL;
}

int kcfo_screen(const kcfo_db* db, const char* text, const int64_t* win_begin, const int32_t* win_len,
                int32_t n_windows, int32_t min_count, int32_t n_threads, kcfo_result* out) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    volatile int32_t next = 0;
    job_t jobs[256];
    pthread_t th[256];
    for (int t = 0; t < n_threads; t++) {
        job_t j = {db, text, win_begin, win_len, n_windows, min_count, out, &next, 0};
        jobs[t] = j;
    }
    if (n_threads == 1) {
        worker(&jobs[0]);
        return jobs[0].rc;
    }
    for (int t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
    int rc = 0;
    for (int t = 0; t < n_threads; t++) { pthread_join(th[t], NULL); rc |= jobs[t].rc; }
    return rc;
}
