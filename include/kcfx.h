/*
 * kcfx.h -- C ABI of the MI355X-native `getVariations` engine (libkcfx.so).
 *
 * The reference (sivasubramanics/kcftools v0.4.0, pure Java) has no FFI / plugin ABI; "plugin"
 * means a picocli sub-command class.  The seam this ABI replaces is Java-internal
 * (paths relative to src/main/java/nl/wur/bis/kcftools/):
 *
 *   new KMC(prefix, inMemory)                      Data/KMC.java:56-78         -> kcfx_db_open
 *   KMC getters used for the KCF header            Data/KMC.java:403-410, Plugins/GetVariants.java:96,103
 *                                                                               -> kcfx_db_info
 *   KMC.getCount(Kmer)                             Data/KMC.java:292-326       -> kcfx_db_lookup (batched)
 *   kmc.close()                                    Data/KMC.java:412-420       -> kcfx_db_close
 *   getWindows() "window" case                     Plugins/GetVariants.java:293-320 -> kcfx_windows
 *   thread-pool fan-out + getFasta + processWindow Plugins/GetVariants.java:129-159,188-261
 *                                                                               -> kcfx_ref_create + kcfx_screen_ref
 *                                                                                  (or the one-shot kcfx_screen)
 *   Window.addTotalKmers/setEffLength/addData args Plugins/GetVariants.java:254-258 -> kcfx_result
 *   Data.update / Data.computeScore                Data/Data.java:70-107       -> kcfx_score
 *   Logger.error -> System.exit(1)                 Utils/Logger.java:29-31     -> int32 status + kcfx_last_error
 *   one run per sample, then `cohort`              utils/run_kcftools.sh:124-140, Plugins/GetVariants.java:129-151
 *                                                                               -> kcfx_ctx_create + kcfx_screen_sharded
 *                                                                                  (every GPU of the node, one call)
 *
 * Conventions
 *   - every entry point returns 0 on success, non-zero on failure; the library never exits or aborts;
 *     kcfx_last_error() returns a thread-local message for the last failure on the calling thread.
 *   - the caller owns every input and output buffer; the library owns only the opaque handles.
 *   - a handle may be used from one thread at a time; distinct handles (one per GPU) may be used concurrently.
 *   - plain pointers and sizes only; no torch / C++ types.  "device" pointers are HIP device pointers.
 *   - k-mers cross the ABI as the reference's Kmer.kmerLong word (Kmer.java:232-252): base i in bits
 *     [63-2i, 62-2i], A=0 C=1 G=2 T=3, left-aligned; k <= 32.
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef KCFX_H
#define KCFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KCFX_OK 0
#define KCFX_ERR_ARG 1       /* bad argument */
#define KCFX_ERR_IO 2        /* file missing / truncated */
#define KCFX_ERR_FORMAT 3    /* not a KMC 0x200 database, unsupported k / prefix length */
#define KCFX_ERR_DEVICE 4    /* HIP error (no device, out of memory, launch failure) */
#define KCFX_ERR_WEIGHTS 5   /* wi + wt + wr != 1.0 (Data.java:101-103; the reference exits) */
#define KCFX_ERR_HOST 6      /* host side: out of memory, thread start failure (no C++ exception ever crosses this ABI) */

typedef struct kcfx_db kcfx_db;     /* one KMC database re-indexed into an HBM-resident table on one GPU */
typedef struct kcfx_ref kcfx_ref;   /* reference text + window list resident in HBM on one GPU */
typedef struct kcfx_ctx kcfx_ctx;   /* a set of GPUs of one node that kcfx_screen_sharded spreads a job over */

/* Per (window, sample) integers: exactly the arguments of Window.addTotalKmers(int), Window.setEffLength(int)
 * and Window.addData(sample, obs, var, inner, left, right, kmerCount, ...)  (GetVariants.java:254-258). 40 bytes. */
typedef struct {
    int32_t total_kmers;     /* valid k-mers in the window (Fasta.java:90-127)                */
    int32_t eff_length;      /* Fasta.getEffectiveATGCCount (Fasta.java:140-167)              */
    int32_t obs;             /* k-mers with count >= min_count (GetVariants.java:224-227)     */
    int32_t var;             /* maximal miss-runs (GetVariants.java:228-230,248-249)          */
    int32_t inner;           /* sum of getDistance over inner gaps (GetVariants.java:237,267) */
    int32_t left;            /* leading miss-run length, in k-mers (GetVariants.java:232-234) */
    int32_t right;           /* trailing miss-run length, in k-mers (GetVariants.java:248-251)*/
    int32_t _pad;
    int64_t kmer_count_sum;  /* sum of counts over observed k-mers (GetVariants.java:226)     */
} kcfx_result;

/* Header fields of the database (KMC.java:124-139) plus the derived sizes. */
typedef struct {
    int32_t kmer_length, mode, counter_size, lut_prefix_length, signature_length;
    int32_t min_count, max_count;
    int32_t both_strands;    /* 1 = canonical k-mers (header byte == 0, KMC.java:133) */
    int32_t version;         /* 0x200 */
    int32_t n_bins;          /* numPrefixArrays (KMC.java:156) */
    int64_t total_kmers;
    int64_t table_bytes;     /* HBM bytes of the re-indexed table */
    int64_t table_lines;     /* number of 128-byte lines */
    int32_t slots_per_line;
    int32_t device;
    double  load_seconds;    /* file read + upload + table build */
    int32_t table_mode;      /* 1 = minimizer-bucketed lines (default), 0 = per-k-mer hashing */
    int32_t minimizer_length;/* M of the minimizer bucketing (0 in mode 0) */
    int64_t heavy_lines;     /* minimizer home lines whose k-mers were re-homed by key hash (low-complexity repeats) */
    int64_t overflow_lines;  /* lines that pushed k-mers to the next line */
    int32_t table_format;    /* 0 = 12-slot lines; 1 = 16 offset-addressed cells per line, second lines, stash (the default for k = 19..31) */
    int32_t lut_suspect;     /* 1: the prefix LUT is not a run of prefix sums and the file was accepted only because
                                KCFX_LENIENT_LUT=1 is set (by default such a .kmc_pre is refused with KCFX_ERR_FORMAT) */
    int64_t stored_kmers;    /* format 1: k-mers with a non-zero counter (the others can never be observed) */
    int64_t stash_kmers;     /* format 1: k-mers that found their cell and their second line's cell taken and live in the stash */
    int64_t stash_bytes;     /* format 1: HBM bytes of the stash (included in table_bytes) */
    int64_t multi_cell_kmers;/* format 1: k-mers stored in more than one cell (tied or palindromic minimizers) */
    int64_t second_choice_kmers; /* format 1: k-mers placed outside their first-choice cell (second or third line) */
    int32_t extended;        /* format 1: 1 = the first-choice line of a k-mer depends on the flank bases next to its minimizer too
                                (databases of >= 0.5 G k-mers whose plain table left > 15 % of them outside their cell) */
    int32_t regions;         /* format 1: table_lines = regions x 2^n (1, 3, 5 or 7: table sizes between the powers of two for large databases) */
    int32_t table_slices;    /* 1, or the n of kcfx_db_open_slice: the handle holds one slice of the database's table */
    int32_t table_slice;     /* ... this one (stored_kmers, stash_kmers, table_* are the slice's) */
} kcfx_db_info_t;

/* ---- database ------------------------------------------------------------------------------------ */

/* Parse <kmc_prefix>.kmc_pre / .kmc_suf (layout: KMC.java:107-189) and build the HBM table on `device`.
 * Fails with KCFX_ERR_FORMAT unless version == 0x200, 1 <= k <= 32, (k - lut_prefix_length) % 4 == 0 and the LUT is the
 * non-decreasing run of record indices from 0 to at most total_kmers that a KMC writer produces (the reference reads the
 * LUT unchecked, KMC.java:153-163; KCFX_LENIENT_LUT=1 in the environment does the same and sets lut_suspect). */
int32_t kcfx_db_open(const char* kmc_prefix, int32_t device, kcfx_db** out);

/* kcfx_db_open with options (NULL = the defaults of kcfx_db_open).
 *   expected_queries  k-mer look-ups this handle is expected to serve over its lifetime = reference k-mers x passes (the
 *                     reference runs ONE pass per database: GetVariants.java:129-159); 0 = unknown. Sizes the table: memory (and
 *                     the time the driver takes to hand it out on a device other processes have used: 28 - 42 GB/s) against the
 *                     speed of a pass. The size minimises  bytes / 35 GB/s + look-ups x seconds per look-up at that density  over
 *                     the sizes from ~0.3 k-mers per cell (150 M k-mers: 2^25 lines, 4.4 GB, 0.081 ms per pass of a 30 Mb reference)
 *                     to ~0.075 per cell (2^27 lines, 17 GB, 0.075 ms): a one-pass job gets the first, a handle that will serve
 *                     more than ~10^12 look-ups -- or an unknown number: 0 -- the second (profiles/r06_notes.md section 1; rounds
 *                     4 - 5 gave every handle the sparse table, and a one-pass job on a busy device paid 0.5 - 2.5 s for its 13 GB).
 *                     A handle that serves fewer than 10^10 look-ups also builds with the cheaper crowding counters (no 4 GiB scratch).
 *   no_fallback       1: a cell table that does not fit the device fails with KCFX_ERR_DEVICE instead of being rebuilt as
 *                     a 12-slot line table (callers that can free memory and retry, like kcfx_screen_sharded's loader). */
typedef struct {
    int64_t expected_queries;
    int32_t no_fallback;
    int32_t reserved;
} kcfx_db_opts;
int32_t kcfx_db_open_opts(const char* kmc_prefix, int32_t device, const kcfx_db_opts* opts, kcfx_db** out);

/* ONE SLICE of the database's table (round 5; KMC.java:173-189 maps a .kmc_suf of any size -- an HBM-resident table has to fit):
 * the k-mers are partitioned over n_slices handles BY MINIMIZER (a hash of the 32 bits that name a k-mer's first-choice line and
 * cell, so a k-mer's second / third lines and stash entry live in the slice of its first-choice cell and the k-mers of a minimizer
 * run stay together); this handle builds the cells of slice `slice` in a table of 1 / n_slices of the size -- n_slices devices hold
 * a database n_slices times as large at the density one device gives a small one. Every handle still streams the whole file.
 * k = 19 .. 31 (cell tables) only, never rebuilt as a line table. Screening: kcfx_screen_ref_partial_async on every slice (each
 * against a reference of the same text and windows on its device: a planned reference walks the positions whose minimizer can be
 * this slice's, a text-only one every position with the other slices' lanes idle), the
 * tiles' hit bitmaps and count sums merged (kcfx_ref_merge_from inside a process, kcfx_ref_export_partial_async / an RCCL gather /
 * kcfx_ref_merge_partial_async between processes) and ONE kcfx_ref_finish_async -- or all of it in one call:
 * kcfx_screen_sharded(KCFX_SHARD_TABLE). kcfx_screen_ref* on a slice handle fail with KCFX_ERR_ARG; kcfx_db_lookup returns a
 * k-mer's counter on the slice(s) that hold it and 0 elsewhere. n_slices = 1 is kcfx_db_open_opts. */
#define KCFX_SLICES_MAX 64
int32_t kcfx_db_open_slice(const char* kmc_prefix, int32_t device, const kcfx_db_opts* opts, int32_t n_slices, int32_t slice, kcfx_db** out);
/* which slice holds the cells of a minimizer: slice_class = two hash bits of the minimizer's order key (0 .. 3; the anchor byte of
 * every planned reference position carries them, so a pass against a slice walks only the positions that can be its own -- with
 * 2 or 4 slices the class alone is the slice), g32 = the 32 bits (line : r) of a first-choice cell in a table of 2^lbits lines
 * per region, which decide among the slices that share a class; host arithmetic (the kernels' cell_slice_of), -1 on bad arguments */
int32_t kcfx_table_slice_of(uint32_t g32, int32_t lbits, int32_t n_slices, int32_t slice_class);

/* Build the same table from host arrays of distinct k-mer words and counts (no files). `counter_size` in
 * 0..4 selects the stored counter width exactly like the KMC header field. */
int32_t kcfx_db_build(const uint64_t* kmers, const uint32_t* counts, int64_t n, int32_t kmer_length,
                      int32_t counter_size, int32_t both_strands, int32_t device, kcfx_db** out);

int32_t kcfx_db_info(const kcfx_db* db, kcfx_db_info_t* info);

/* The header fields alone (KMC.java:107-168 readPrefixFile; what GetVariants.java:96,103 needs for the KCF header before
 * any window is screened): parses <kmc_prefix>.kmc_pre on the host, touches no GPU. The table_* fields stay zero. */
int32_t kcfx_db_header(const char* kmc_prefix, kcfx_db_info_t* info);

/* Batched KMC.getCount (KMC.java:292-326): counts_out[i] = stored counter of kmers[i] or 0.  The words are looked
 * up as given (canonicalise first for a both-strands database, as Kmer(Kmer, bothStrands) does, Kmer.java:57-65).
 * Counts are Java ints: a 4-byte counter with the top bit set is negative. Host pointers. */
int32_t kcfx_db_lookup(kcfx_db* db, const uint64_t* kmers, int64_t n, int32_t* counts_out);

void kcfx_db_close(kcfx_db* db);
/* Large device buffers of closed databases and destroyed references (tables, stashes, reference text) are kept, per
 * device, for the next handle of the same size: a run over many samples opens one database after the other, and on this
 * platform hipMalloc of a table that follows the hipFree of the previous one costs up to 2 s (KMC.java:412-420 has no
 * counterpart: the JVM unmaps). At most 16 blocks and half of the device (kcfx_set_cache_fraction; kcfx_screen_sharded raises it to
 * 0.9 for the length of its call: a table of 200 GB must survive until the next database takes it); given back by themselves when an allocation of THIS
 * library fails, by this call (device < 0: all devices), never kept with KCFX_NO_POOL=1 in the environment. A process that
 * shares the device with another allocator (PyTorch, ...) should call this after closing its handles, or set KCFX_NO_POOL:
 * the other allocator cannot see the kept blocks and may run out of memory while they sit here. */
int32_t kcfx_release_cached(int32_t device);
/* The share of a device's memory that released buffers may occupy before the oldest go back to the driver (0 .. 1; default 0.5,
 * or KCFX_POOL_FRACTION in the environment). A process that opens one large database after the other (tables of 100 - 200 GB:
 * the next open gets the previous table's memory instead of waiting for the driver to clear it, 7 s against 12.8 s) raises it;
 * a process that shares the device lowers it or calls kcfx_release_cached. Returns the previous value; out-of-range: unchanged. */
double kcfx_set_cache_fraction(double fraction);

/* ---- reference text + windows ---------------------------------------------------------------------- */

/* Upload newline-free, case-preserved text and a window list: window w is text[win_begin[w], win_begin[w]+win_len[w]).
 * Windows may overlap (tiling windows share k-1 bases) and gene-mode windows are pre-spliced by the caller.
 * `kmer_length` fixes the tile plan; use the database's. Host pointers; they are not retained.
 * For kmer_length 19 .. 31 the call also works out, once, what a look-up in a cell table needs of the reference and no sample
 * changes: 2-bit codes of the text and one byte per k-mer start (is it valid, which of its 16-mers is its minimizer, on which
 * strand) -- 1.25 bytes per base beside the text, one kernel pass. Every screening launch against this reference then starts
 * from those instead of from the ASCII (KCFX_REF_PLAN=0 in the environment: text only, the kernels derive it per launch). */
int32_t kcfx_ref_create(int32_t device, int32_t kmer_length, const uint8_t* text, int64_t text_len,
                        const int64_t* win_begin, const int32_t* win_len, int32_t n_windows, kcfx_ref** out);
/* kcfx_ref_create with options (NULL = the defaults of kcfx_ref_create).
 *   expected_screens  how many databases will be screened against this reference: 0 = unknown / several -- the reference is
 *                     PLANNED once (2-bit codes + one anchor byte per k-mer beside the text) and every pass starts from the plan;
 *                     1 = exactly one (the reference program's own usage: one getVariations run per sample,
 *                     run_kcftools.sh:124-140) -- the reference stays text only and the one screening launch plans its tiles
 *                     itself: K1 - K3 in one kernel, faster than plan + screen as two launches and 1.25 bytes per base lighter.
 *                     (KCFX_REF_PLAN=0 / 1 in the environment forces text-only / planned whatever this says.) */
typedef struct {
    int32_t expected_screens;
    int32_t reserved;
} kcfx_ref_opts;
int32_t kcfx_ref_create_opts(int32_t device, int32_t kmer_length, const uint8_t* text, int64_t text_len,
                             const int64_t* win_begin, const int32_t* win_len, int32_t n_windows, const kcfx_ref_opts* opts,
                             kcfx_ref** out);
int32_t kcfx_ref_n_windows(const kcfx_ref* ref);
int64_t kcfx_ref_n_kmer_slots(const kcfx_ref* ref);   /* sum over windows of max(0, len - k + 1) */
void kcfx_ref_destroy(kcfx_ref* ref);

/* ---- the hot path ----------------------------------------------------------------------------------- */

/* Screen every window of `ref` against `db`; out[w] (host, n_windows entries) receives the integers.
 * Replaces GetVariants.java:129-159. Synchronous. db and ref must live on the same device. */
int32_t kcfx_screen_ref(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, kcfx_result* out);

/* Same, asynchronous on `hip_stream` (a hipStream_t, NULL = default stream) with the results left in device
 * memory `out_device` (n_windows * sizeof(kcfx_result) bytes). No host synchronisation, no allocation.
 * ONE screen in flight per kcfx_ref: the hit bitmaps between the two kernels of a call are scratch owned by the
 * reference, so calls on the same reference must be ordered (same stream, or an event between them); use one
 * kcfx_ref per stream to overlap. */
int32_t kcfx_screen_ref_async(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, void* out_device, void* hip_stream);

/* Several samples, ONE pass over the reference (SURVEY f-4): what the reference does as S getVariations runs followed by
 * `cohort` (run_kcftools.sh:124-140, Plugins/Cohort.java:71-119). out[s * n_windows + w] = sample s (input order),
 * window w; each slice equals kcfx_screen_ref(dbs[s], ...). Databases must live on ref's device and share its k;
 * cell tables (the default for k = 19 .. 31) are screened one launch per sample, the reference staying resident, and reduced up to
 * eight samples per launch (round 6);
 * 12-slot line tables of k = 31 with counters of at most 2 bytes are screened up to 8 at a time by one kernel (same
 * results either way). */
int32_t kcfx_screen_ref_multi(kcfx_db* const* dbs, int32_t n_db, kcfx_ref* ref, int32_t min_kmer_count, kcfx_result* out);
/* Same, asynchronous on `hip_stream`, results left in device memory (n_db * n_windows records). Databases that the
 * multi-sample kernel takes (12-slot line tables) need scratch of 512 B per tile and sample on the reference: it is allocated
 * by the first call that needs it, or beforehand by kcfx_ref_reserve_multi -- do that before capturing the call into a HIP
 * graph (an allocation is not capturable). Two or more cell tables (table_format 1) use the same scratch for their batched window
 * reduce while 520 B per tile and sample stay under 2 GiB (a 14.5 Gb reference keeps the launch pair per sample and needs none). */
int32_t kcfx_screen_ref_multi_async(kcfx_db* const* dbs, int32_t n_db, kcfx_ref* ref, int32_t min_kmer_count,
                                    void* out_device, void* hip_stream);

/* Reserve the multi-sample scratch of `ref` for batches of up to `n_samples` line-table databases (clamped to 8). */
int32_t kcfx_ref_reserve_multi(kcfx_ref* ref, int32_t n_samples);

/* The table kcfx_db_open would build for a database of `total_kmers` k-mers of length `kmer_length` (canonical: both_strands != 0)
 * on a device with `free_bytes` of free memory -- host arithmetic only, no GPU: what a caller budgets memory with before it opens
 * anything (new KMC(prefix, loadMemory) holds the whole database too: KMC.java:56-78). *lines = regions x 2^n lines of 128 bytes
 * (16 cells each) for k = 19 .. 31; the stash (the k-mers that found their cells taken: 0.03 - 4 % of the database, 16 - 64 bytes
 * each) comes on top. *lines = 0 (and KCFX_OK): k outside 19 .. 31 or more k-mers than cells -- such a database gets the 12-slot
 * line table, whose size depends on its minimizer statistics. lines / regions may be NULL. */
int32_t kcfx_table_plan(int64_t total_kmers, int32_t kmer_length, int32_t both_strands, int64_t free_bytes,
                        int64_t* lines, int32_t* regions);
/* ... for a handle opened with kcfx_db_opts.expected_queries = `expected_queries` (0 = unknown: kcfx_table_plan). */
int32_t kcfx_table_plan_for(int64_t total_kmers, int32_t kmer_length, int32_t both_strands, int64_t free_bytes, int64_t expected_queries,
                            int64_t* lines, int32_t* regions);

/* One-shot convenience: kcfx_ref_create + kcfx_screen_ref + kcfx_ref_destroy. */
int32_t kcfx_screen(kcfx_db* db, const uint8_t* text, int64_t text_len, const int64_t* win_begin,
                    const int32_t* win_len, int32_t n_windows, int32_t min_kmer_count, kcfx_result* out);

/* Data.update + Data.computeScore (Data.java:87,95-107), strict IEEE double, no FMA; runs on the host.
 * score/mean_count may be NULL. Returns KCFX_ERR_WEIGHTS if wi + wt + wr != 1.0 (values are still written). */
int32_t kcfx_score(const kcfx_result* res, int64_t n, double wi, double wt, double wr, double* score, double* mean_count);

/* GetVariants.getWindows "window" case (GetVariants.java:293-320). step <= 0: tiling. Writes up to `cap` windows,
 * *n_out = number of windows. window <= k-1 in tiling mode (the reference loops forever) is KCFX_ERR_ARG. */
int32_t kcfx_windows(int32_t seq_len, int32_t window, int32_t step, int32_t kmer_length,
                     int32_t* starts, int32_t* ends, int64_t cap, int64_t* n_out);

/* ---- cohort table: S samples in ONE table ---------------------------------------------------------------------------- */

/* What the reference does as S getVariations runs and a `cohort` merge (utils/run_kcftools.sh:124-160, Plugins/Cohort.java:71-119),
 * taken into the table: the k-mers of up to KCFX_COHORT_MAX samples are re-indexed into ONE cell table whose cells carry one
 * counter per sample, so a reference k-mer is located once for all samples (one line request per minimizer run plus the
 * lines of the counters, instead of one per sample) and only the `count >= -c` test, the hit ballot and the count sum are
 * per sample: algorithmic bytes per sample-k-mer = (1 + 8) / S + counter_size. Needs k = 19 .. 31 (the k-mer lengths a cell
 * table exists for; one k across the samples), counters of at most 2 bytes and
 * one strandedness across the samples (KCFX_ERR_ARG otherwise: screen such samples through kcfx_screen_ref_multi).
 * Memory: 8 + KCFX_COHORT_MAX bytes per cell of the union of the samples' k-mers (a cell holds 8 bits per sample; a k-mer that some
 * sample counted more than 255 times lives in the overflow stash with 16-bit counters -- the results are exact either way). */
#define KCFX_COHORT_MAX 8
typedef struct kcfx_cohort kcfx_cohort;
typedef struct {
    int32_t n_samples, kmer_length, both_strands, device;
    int64_t total_records;     /* k-mer records of all samples together (a k-mer that several samples hold counts once per sample) */
    int64_t table_lines;       /* 128-byte key lines (16 cells each) */
    int64_t table_bytes;       /* HBM bytes: keys + counters + stash + stash counters */
    int64_t stash_records;     /* records whose k-mer lives in the stash */
    double  load_seconds;
} kcfx_cohort_info_t;
/* opts may be NULL; opts->expected_queries sizes the key table as in kcfx_db_open_opts. */
int32_t kcfx_cohort_open(const char* const* kmc_prefixes, int32_t n_samples, int32_t device, const kcfx_db_opts* opts,
                         kcfx_cohort** out);
int32_t kcfx_cohort_info(const kcfx_cohort* cohort, kcfx_cohort_info_t* info);
void kcfx_cohort_close(kcfx_cohort* cohort);
/* Every window of `ref` against every sample: out[s * n_windows + w] (host) = sample s (input order), window w -- each slice
 * equals kcfx_screen_ref on that sample's own database. Synchronous. */
int32_t kcfx_cohort_screen_ref(kcfx_cohort* cohort, kcfx_ref* ref, int32_t min_kmer_count, kcfx_result* out);
/* Same, asynchronous on `hip_stream`, results left in device memory (n_samples * n_windows records). The reference's
 * multi-sample scratch is allocated by the first call (kcfx_ref_reserve_multi beforehand makes the call capturable). */
int32_t kcfx_cohort_screen_ref_async(kcfx_cohort* cohort, kcfx_ref* ref, int32_t min_kmer_count, void* out_device, void* hip_stream);

/* ---- all GPUs of a node ---------------------------------------------------------------------------------- */

/* The units are independent -- a (window, sample) result needs that window's text and that sample's database only
 * (GetVariants.java:138-150) -- so a job shards with no exchange step: one worker thread per listed device inside the
 * library, each with its own resident reference. A device index may be listed more than once (shards share the GPU). */
int32_t kcfx_ctx_create(const int32_t* devices, int32_t n_devices, kcfx_ctx** out);
int32_t kcfx_ctx_n_devices(const kcfx_ctx* ctx);
void kcfx_ctx_destroy(kcfx_ctx* ctx);

#define KCFX_SHARD_AUTO 0      /* samples when n_samples >= devices; windows for one sample; else grid */
#define KCFX_SHARD_SAMPLES 1   /* device g screens samples g, g + G, ... against the whole window list */
#define KCFX_SHARD_WINDOWS 2   /* device g screens a contiguous window range (equal k-mer start positions) of every sample */
#define KCFX_SHARD_GRID 3      /* samples x windows: the sample-major sequence of (sample, window) units cut into equal contiguous
                                  parts, one per device -- a device opens at most ceil(S / G) + 1 databases, not all S */
#define KCFX_SHARD_COHORT 4    /* cohort tables across the devices (never chosen by AUTO): the samples in groups of KCFX_COHORT_MAX
                                  consecutive ones, each group ONE cohort table (kcfx_cohort_open: one look-up per reference k-mer
                                  for the whole group -- what run_kcftools.sh:124-160 + Cohort.java:71-119 produce from S runs); the
                                  group-major sequence of (group, window) units is cut into equal contiguous parts like the grid's.
                                  8 samples on 8 devices: every device builds the same table and screens an eighth of the windows;
                                  32 samples: four tables, two devices each. A group whose table cannot be built (3- / 4-byte
                                  counters, mixed strandedness, k outside 19 .. 31, no room: key table + 8 B of counters per cell,
                                  ~0.55 GiB per million k-mers of the union at the one-pass density) is screened sample by sample.
                                  kcfx_shard_plan reports a group as seg_sample = its first sample; it spans
                                  min(KCFX_COHORT_MAX, n_samples - seg_sample) samples */

#define KCFX_SHARD_TABLE 5     /* the TABLE of every sample partitioned over the devices (never chosen by AUTO; k = 19 .. 31): device g
                                  opens slice g of G of the sample's database (kcfx_db_open_slice: 1 / G of the lines; the G loaders
                                  share one stream of the file: kcfx_shard_slice_io), screens the WHOLE window list against it, the tiles' hit bitmaps are OR-ed on
                                  the first device and reduced there; sample after sample. For a database that one device holds
                                  only at a density that costs it half its speed (a wheat-scale count: BASELINE configs[4]).
                                  kcfx_shard_plan reports one segment per (device, sample), all windows each */

typedef struct {
    int32_t shard_mode_used, n_devices;
    double seconds_total;         /* wall time of the call */
    double seconds_ref;           /* slowest device: reference upload + tile plan */
    double seconds_db_open_sum;   /* summed over (device, sample): kcfx_db_open */
    double seconds_screen_sum;    /* summed over (device, sample): kcfx_screen_ref */
    double seconds_exposed_load;  /* slowest device: time its screening waited for a database (the rest of the loading
                                     ran under the previous sample's screening) */
} kcfx_shard_stats;

/* The partitioning kcfx_screen_sharded would use, without touching a GPU: segment i = device seg_device[i] (index into the
 * context's device list) screens sample seg_sample[i] against windows [seg_w_lo[i], seg_w_hi[i]); segments of one device are in
 * execution order. Up to `cap` segments are written, *n_segments = how many there are; *mode_used = what KCFX_SHARD_AUTO chose.
 * Every (sample, window) pair lies in exactly one segment (KCFX_SHARD_TABLE: in one segment PER DEVICE -- every device screens
 * every pair against its slice of the sample's table). Output arrays may be NULL. */
int32_t kcfx_shard_plan(int32_t shard_mode, int32_t n_devices, int32_t n_samples, const int32_t* win_len, int32_t n_windows,
                        int32_t kmer_length, int32_t* seg_device, int32_t* seg_sample, int32_t* seg_w_lo, int32_t* seg_w_hi,
                        int64_t cap, int64_t* n_segments, int32_t* mode_used);

/* getVariations for n_samples databases against one reference, on every device of `ctx`: what the reference does as one
 * process per sample (run_kcftools.sh:124-140), each fanning its windows over a thread pool (GetVariants.java:129-151).
 * out[s * n_windows + w] (host) = sample s (input order), window w -- each slice equals kcfx_screen_ref on that sample.
 * On a device the next sample's database is opened while the current one is screened (at most two resident). Every
 * database must have k-mer length `kmer_length`. `stats` may be NULL. Synchronous; the caller's buffers are not retained. */
int32_t kcfx_screen_sharded(kcfx_ctx* ctx, const char* const* kmc_prefixes, int32_t n_samples, int32_t kmer_length,
                            const uint8_t* text, int64_t text_len, const int64_t* win_begin, const int32_t* win_len,
                            int32_t n_windows, int32_t min_kmer_count, int32_t shard_mode, kcfx_result* out,
                            kcfx_shard_stats* stats);

/* What the slice loaders of this process's most recent kcfx_screen_sharded(KCFX_SHARD_TABLE) call read from the host side: the G
 * loaders of a sample share ONE stream of its .kmc_suf (every chunk is read into pinned memory once and copied to each device over
 * that device's own link; KMC.java:84-102,173-189 maps a file once too) -- bytes_read is therefore ~ passes x the file, not G x
 * that; chunks_shared counts the chunk requests a sibling loader's read served. Outputs may be NULL. */
int32_t kcfx_shard_slice_io(int64_t* bytes_read, int64_t* chunks_read, int64_t* chunks_shared);

/* ---- the pieces of a pass over a partitioned table (kcfx_db_open_slice) -------------------------------------------------- */

/* The screening kernel alone on the caller's stream: the tiles' bitmaps (validity, and the hits this slice found) and count sums
 * stay in the reference's scratch (one screen in flight per reference, as for kcfx_screen_ref_async). Also takes a whole table. */
int32_t kcfx_screen_ref_partial_async(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count, void* hip_stream);
int32_t kcfx_screen_ref_partial(kcfx_db* db, kcfx_ref* ref, int32_t min_kmer_count);     /* synchronous, on the reference's own stream */
/* bytes of a reference's partial results: 512 per tile of bitmaps + 8 per tile of count sums (returns their sum) */
int64_t kcfx_ref_partial_bytes(const kcfx_ref* ref, int64_t* bitmap_bytes, int64_t* sum_bytes);
/* the partial results copied into caller-owned DEVICE buffers (tensors handed to an RCCL gather), on the caller's stream */
int32_t kcfx_ref_export_partial_async(kcfx_ref* ref, void* bitmaps_device, void* sums_device, void* hip_stream);
/* ... and back: caller-owned device buffers of that layout INTO the reference's scratch. With kcfx_ref_tile_range (windows
 * [w_lo, w_hi) of a reference are its tiles [*t_lo, *t_hi): 64 bitmap words and one count sum per tile, in tile order) this is what a
 * job needs to let the slices' results meet RANGE BY RANGE instead of on one root: rank q owns a window range, receives that range's
 * tiles from every rank (1 / G of the bitmaps each), imports its own into a reference of just those windows, merges the others and
 * finishes -- no device carries (G - 1) x the bitmaps of the whole window list (kcftools_amd/shard.py: reduce_scatter_table_partials).
 * Both references must be of the same form (kcfx_ref_opts.expected_screens): a text-only one cuts its tiles at 2 001 k-mers, a planned one at 2 048. */
int32_t kcfx_ref_import_partial_async(kcfx_ref* ref, const void* bitmaps_device, const void* sums_device, void* hip_stream);
int32_t kcfx_ref_tile_range(const kcfx_ref* ref, int32_t w_lo, int32_t w_hi, int32_t* t_lo, int32_t* t_hi);
/* another slice's partial results -- device buffers on the reference's device, same text and windows -- into this reference's:
 * hit words OR-ed, count sums added (every k-mer is looked up on exactly one slice) */
int32_t kcfx_ref_merge_partial_async(kcfx_ref* ref, const void* bitmaps_device, const void* sums_device, void* hip_stream);
/* the same from another reference handle of this process, on this device or another (peer copy); synchronous */
int32_t kcfx_ref_merge_from(kcfx_ref* dst, kcfx_ref* src);
/* reduce_windows over what the reference's scratch holds now: 40 bytes per window to out_device / to host memory */
int32_t kcfx_ref_finish_async(kcfx_ref* ref, void* out_device, void* hip_stream);
int32_t kcfx_ref_finish(kcfx_ref* ref, kcfx_result* out);

/* ---- measurement ------------------------------------------------------------------------------------ */

/* When enabled, kcfx_screen_ref_async brackets its dominant kernel (screen_tiles) with HIP events on the launch
 * stream; kcfx_prof_read synchronises those events and returns the summed kernel time and launch count since the
 * last reset. Used by bench.py for roofline.achieved. */
int32_t kcfx_prof_enable(kcfx_ref* ref, int32_t enable);
int32_t kcfx_prof_read(kcfx_ref* ref, double* kernel_ms_sum, int64_t* launches, int32_t reset);

/* The sample-independent half of a pass -- SURVEY's K1: 2-bit pack (Kmer.java:232-252), which positions start a valid k-mer
 * (Fasta.java:90-127), and which 16-mer of each k-mer is its minimizer -- is worked out by kcfx_ref_create once per reference
 * and shared by every sample screened against it. kcfx_ref_plan_async runs it again over the resident text on the caller's
 * stream (idempotent; no allocation, no synchronisation): the cost a one-sample-per-reference job pays per pass, which
 * bench.py reports beside the screening kernel (ref_plan_ms, frac_one_pass). KCFX_ERR_ARG for a text-only reference.
 * With profiling enabled the launch is bracketed by events of its own, read by kcfx_prof_read_plan. */
int32_t kcfx_ref_plan_async(kcfx_ref* ref, void* hip_stream);
int32_t kcfx_ref_is_planned(const kcfx_ref* ref);     /* 1: codes + anchors resident; 0: text only (planned per launch) */
int32_t kcfx_prof_read_plan(kcfx_ref* ref, double* kernel_ms_sum, int64_t* launches, int32_t reset);

/* ---- misc --------------------------------------------------------------------------------------------- */
const char* kcfx_last_error(void);
int32_t kcfx_device_count(int32_t* n);
const char* kcfx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KCFX_H */
