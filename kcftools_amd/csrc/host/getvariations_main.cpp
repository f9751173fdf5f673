// kcftools-amd getVariations -- command-line driver that keeps kcftools' `getVariations` option set
// (Plugins/GetVariants.java:21-61) and flow (:92-183) around the C ABI of include/kcfx.h:
//   KMC open -> header -> FASTA index -> (GTF) -> windows -> ONE kcfx_screen call (replaces the thread pool :129-159)
//   -> Window.addTotalKmers/setEffLength/addData semantics (kcfx_score) -> sort -> KCF text.
// Host side only; all k-mer work happens in libkcfx.so on the GPU.
#include <sys/time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/kcfx.h"
#include "fasta_index.h"
#include "gtf.h"
#include "kcf_writer.h"

using namespace kcfhost;

namespace {
const char* CLASS_NAME = "GetVariants";

// Utils/Logger.java:14-32: "yyyy-MM-dd HH:mm:ss:SSS - LEVEL    - <class padded to 20> - msg"; ERROR exits with status 1
void log_line(const char* level, const std::string& msg) {
    timeval tv;
    gettimeofday(&tv, nullptr);
    std::tm tm;
    localtime_r(&tv.tv_sec, &tm);
    char ts[64];
    std::strftime(ts, sizeof ts, "%Y-%m-%d %H:%M:%S", &tm);
    std::printf("%s:%03d - %-8s - %-20s - %s\n", ts, (int)(tv.tv_usec / 1000), level, CLASS_NAME, msg.c_str());
    std::fflush(stdout);
}
void info(const std::string& m) { log_line("INFO", m); }
void warning(const std::string& m) { log_line("WARNING", m); }
std::thread* g_background = nullptr;       // the GPU runtime coming up (main): joined before the process exits on an error
[[noreturn]] void error(const std::string& m) {
    log_line("ERROR", m);
    if (g_background && g_background->joinable()) g_background->join();
    std::exit(1);                           // Logger.java:29-31
}

struct Options {
    std::string ref, kmc, out, sample, feature, gtf;
    int threads = 2, window = 0, min_count = 1, step = 0;
    std::vector<int> devices{0};          // --device N | --devices a,b,...: the job is sharded over them (kcfx_screen_sharded)
    int shard = KCFX_SHARD_AUTO;          // --shard auto|samples|windows|grid|cohort|table
    bool cohort_table = false;            // --cohort-table (= --shard cohort): the samples in cohort tables of up to 8, tables x window
                                          // ranges over the devices (KCFX_SHARD_COHORT)
    bool memory = false;
    double wi = 0.3, wt = 0.3, wr = 0.4;
};

void usage() {
    std::puts("Usage: kcftools-amd getVariations -r <ref.fa> -k <kmc prefix> -o <out.kcf> -s <sample> -f window|gene|transcript\n"
              "       [-w window] [-p step] [-g gtf] [-c min-k-count] [--wi 0.3] [--wt 0.3] [--wr 0.4] [-t threads] [-m] [--device 0 | --devices 0,1,...] [--shard auto|samples|windows|grid|cohort|table] [--cohort-table]");
}

std::vector<std::string> split_commas(const std::string& v) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t c = v.find(',', p);
        out.push_back(v.substr(p, c == std::string::npos ? std::string::npos : c - p));
        if (c == std::string::npos) break;
        p = c + 1;
    }
    return out;
}

struct Win {
    std::string id, seq;
    int32_t start, end;      // as printed (0-based half-open in window mode, raw 1-based GTF coordinates otherwise)
    int64_t text_begin;
    int32_t text_len;
    size_t order;
};

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2 || std::strcmp(argv[1], "getVariations") != 0) {
        usage();
        return 2;
    }
    // KCFX_CLI_TRACE=1: wall time of every phase of the run on stderr
    const bool trace_on = std::getenv("KCFX_CLI_TRACE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!trace_on) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[kcftools-amd] %-34s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(t - t_prev).count());
        t_prev = t;
    };
    auto epoch = [&](const char* what) {      // (KCFX_CLI_TRACE: wall-clock stamps a parent can hold against its own spawn / wait times)
        if (!trace_on) return;
        timeval tv;
        gettimeofday(&tv, nullptr);
        std::fprintf(stderr, "[kcftools-amd] epoch %-28s %ld.%06ld\n", what, (long)tv.tv_sec, (long)tv.tv_usec);
    };
    epoch("main entered");
    Options o;
    std::string cmdline = "kcftools-amd";
    for (int i = 1; i < argc; ++i) cmdline += std::string(" ") + argv[i];
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string {
            if (i + 1 >= argc) error("Missing value for option " + a);
            return argv[++i];
        };
        if (a == "-r" || a == "--reference") o.ref = val();
        else if (a == "-k" || a == "--kmc") o.kmc = val();
        else if (a == "-o" || a == "--output") o.out = val();
        else if (a == "-s" || a == "--sample") o.sample = val();
        else if (a == "-f" || a == "--feature") o.feature = val();
        else if (a == "-t" || a == "--threads") o.threads = std::atoi(val().c_str());
        else if (a == "-m" || a == "--memory") o.memory = true;
        else if (a == "--wi") o.wi = std::atof(val().c_str());
        else if (a == "--wt") o.wt = std::atof(val().c_str());
        else if (a == "--wr") o.wr = std::atof(val().c_str());
        else if (a == "-w" || a == "--window") o.window = std::atoi(val().c_str());
        else if (a == "-g" || a == "--gtf") o.gtf = val();
        else if (a == "-c" || a == "--min-k-count") o.min_count = std::atoi(val().c_str());
        else if (a == "-p" || a == "--step") o.step = std::atoi(val().c_str());
        else if (a == "--device" || a == "--devices") {
            o.devices.clear();
            for (const auto& d : split_commas(val())) o.devices.push_back(std::atoi(d.c_str()));
            if (o.devices.empty()) o.devices.push_back(0);
        }
        else if (a == "--shard") {
            const std::string v = val();
            if (v == "auto") o.shard = KCFX_SHARD_AUTO;
            else if (v == "samples") o.shard = KCFX_SHARD_SAMPLES;
            else if (v == "windows") o.shard = KCFX_SHARD_WINDOWS;
            else if (v == "grid") o.shard = KCFX_SHARD_GRID;
            else if (v == "cohort") o.shard = KCFX_SHARD_COHORT;
            else if (v == "table") o.shard = KCFX_SHARD_TABLE;          // every sample's table in slices over the devices (a database too large for one)
            else error("Invalid --shard value: " + v + " (auto, samples, windows, grid, cohort or table)");
        }
        else if (a == "--cohort-table") o.cohort_table = true;
        else if (a == "-h" || a == "--help") { usage(); return 0; }
        else error("Unknown option: " + a);
    }
    if (o.ref.empty() || o.kmc.empty() || o.out.empty() || o.sample.empty() || o.feature.empty()) {
        usage();
        error("Missing required option (-r, -k, -o, -s, -f)");
    }
    // validateCMD (GetVariants.java:357-386)
    if (o.feature == "window") {
        if (o.window <= 0) error("Window size is required for window model");
        if (!o.gtf.empty()) error("GTF file is not valid for window model");
    } else if (o.feature == "gene" || o.feature == "transcript") {
        if (o.gtf.empty()) error("GTF file is required for targeted model");
        if (o.window > 0) error("Window size is not valid for targeted model");
    } else {
        error("Invalid model type: " + o.feature + ". Supported models are 'window' or 'gene' or 'transcript'");
    }
    if (o.threads <= 0) error("Number of threads should be greater than 0");
    if (o.min_count < 1) error("Minimum kmer count should be at least 1");
    // Extension: -k and -s may be comma-separated lists of equal length. The reference is uploaded once
    // and screened against every sample; the output is the multi-sample KCF that `cohort` (Plugins/Cohort.java) would
    // produce from the S single-sample files.
    std::vector<std::string> kmc_list = split_commas(o.kmc), sample_list = split_commas(o.sample);
    if (kmc_list.size() != sample_list.size()) error("The numbers of KMC databases (-k) and sample names (-s) differ");
    // cleanSampleName (:392-401)
    for (auto& name : sample_list) {
        std::string clean = name;
        for (char& c : clean)
            if (std::strchr("\\/:*?\"<>|", c)) c = '_';
        if (clean != name) warning("Sample name contains invalid characters, changed to: " + clean);
        name = clean;
    }
    o.sample = sample_list[0];
    for (size_t i = 1; i < sample_list.size(); ++i) o.sample += "\t" + sample_list[i];    // #CHROM line columns

    // The GPU runtime takes ~0.1 s to come up (kcfx_ctx_create is the process's first HIP call): it does so on its own thread while
    // this one reads the .kmc_pre header, the FASTA index and the windows' text
    std::vector<int32_t> devices(o.devices.begin(), o.devices.end());
    kcfx_ctx* ctx = nullptr;
    int32_t ctx_rc = 0;
    std::string ctx_err;
    std::thread ctx_thread([&] {
        ctx_rc = kcfx_ctx_create(devices.data(), (int32_t)devices.size(), &ctx);
        if (ctx_rc) ctx_err = kcfx_last_error();             // (the message is thread-local)
    });
    g_background = &ctx_thread;                              // (error() exits the process: not with this thread inside HIP)

    // new KMC(prefix, loadMemory) (:95): the header now (k for the windows and the KCF header, :96,:103); the records are
    // loaded onto the GPUs by kcfx_screen_sharded, each database while the previous one is being screened
    kcfx_db_info_t dbi;
    if (kcfx_db_header(kmc_list[0].c_str(), &dbi)) error(kcfx_last_error());
    const int k = dbi.kmer_length;
    info("KMC database: k=" + std::to_string(k) + ", " + std::to_string(dbi.total_kmers) + " k-mers");

    mark("options + .kmc_pre header");
    FastaIndex index;
    std::string err;
    if (!index.open(o.ref, err)) error(err);
    mark("FASTA index");
    Gtf gtf;
    const bool targeted = o.feature != "window";
    if (targeted && !gtf.parse(o.gtf, err)) error(err);

    info("Generating windows...");
    KcfHeaderInfo hdr;
    hdr.reference = o.ref;
    hdr.sample = o.sample;
    hdr.command_line = cmdline;
    hdr.window = o.window;
    hdr.step = o.step;
    hdr.kmer = k;
    hdr.wi = o.wi;
    hdr.wt = o.wt;
    hdr.wr = o.wr;

    std::string text;
    if (!targeted) {                                      // (every contig once: no regrowth while it is assembled)
        size_t total = 0;
        for (const auto& e : index.entries()) total += (size_t)e.length;
        text.reserve(total);
    }
    std::vector<Win> wins;
    std::vector<std::pair<size_t, size_t>> per_contig;    // [first, last) into wins, in .faidx order
    for (const auto& e : index.entries()) {
        hdr.contigs.emplace_back(e.name, e.length);
        const size_t first = wins.size();
        if (!targeted) {
            // getWindows "window" (:293-320) through the ABI, text = the contig once
            int64_t n = 0;
            if (kcfx_windows(e.length, o.window, o.step, k, nullptr, nullptr, 0, &n)) error(kcfx_last_error());
            std::vector<int32_t> s((size_t)n), en((size_t)n);
            if (n && kcfx_windows(e.length, o.window, o.step, k, s.data(), en.data(), n, &n)) error(kcfx_last_error());
            const int64_t base = (int64_t)text.size();
            if (e.length > 0 && !index.append_sequence(e, 0, e.length, text, err)) error(err);
            for (int64_t i = 0; i < n; ++i)
                wins.push_back(Win{e.name + "_" + std::to_string(s[(size_t)i]), e.name, s[(size_t)i], en[(size_t)i], base + s[(size_t)i],
                                   en[(size_t)i] - s[(size_t)i], wins.size()});
        } else {
            const bool is_gene = o.feature == "gene";
            const auto genes = gtf.children(e.name);                       // gtf.getGenes(sequenceName)
            if (!is_gene && genes.empty()) warning("No genes found in GTF file for sequence: " + e.name);   // :333-336
            for (const auto& gene : genes) {
                std::vector<std::string> ids;
                if (is_gene) {
                    ids.push_back(gene);
                } else {
                    ids = gtf.children(gene);
                    if (ids.empty()) error("No transcripts found for gene: " + gene + " in GTF file for sequence: " + e.name);   // :339-342
                }
                for (const auto& id : ids) {
                    const GtfFeature* f = gtf.feature(id);
                    if (!f) error("Feature ID not found: " + id);                         // GTF.java:300-303
                    std::vector<GtfLoci> loci;
                    if (!gtf.merged_exon_loci(id, is_gene, loci)) error("Fasta object is null for window: " + id);   // :213-216
                    const int64_t begin = (int64_t)text.size();
                    for (const auto& l : loci) {
                        const FaidxEntry* ce = index.find(l.chrom);
                        if (!ce) error("Sequence not found in index: " + l.chrom);
                        std::string seq;
                        if (!index.get_sequence(*ce, (int64_t)l.start - 1, (int64_t)l.end - l.start + 1, seq, err)) error(err);   // GTF.java:242
                        text += seq;
                    }
                    wins.push_back(Win{id, f->chrom, f->start, f->end, begin, (int32_t)((int64_t)text.size() - begin), wins.size()});
                }
            }
        }
        per_contig.emplace_back(first, wins.size());
    }
    info("Number of windows: " + std::to_string(wins.size()));
    mark("windows + their text");

    // the hot path: one call instead of the thread pool (:129-159)
    const int32_t n = (int32_t)wins.size();
    std::vector<int64_t> wb((size_t)n);
    std::vector<int32_t> wl((size_t)n);
    for (int32_t i = 0; i < n; ++i) {
        wb[(size_t)i] = wins[(size_t)i].text_begin;
        wl[(size_t)i] = wins[(size_t)i].text_len;
    }
    // One call for every sample on every device (replaces the thread pool :129-159 and the per-sample runs of
    // run_kcftools.sh:124-140): sample- or window-sharded, databases opened while their predecessors are screened. Errors
    // come back as a status -- nothing exits from a worker thread.
    const size_t S = kmc_list.size();
    std::vector<kcfx_result> all(S * (size_t)n);
    std::vector<std::vector<double>> score(S, std::vector<double>((size_t)n)), depth(S, std::vector<double>((size_t)n));
    // --cohort-table: the samples in ONE table per group of eight, a counter per sample and cell, one look-up per reference k-mer for
    // the whole group (kcfx_cohort_open) -- tables x window ranges over the listed devices. A table opens more slowly than its
    // samples' own tables and screens them S x faster: worth it for a large reference. A group that cannot have one (3- / 4-byte
    // counters, mixed strandedness, no room) is screened sample by sample inside the same call.
    if (o.cohort_table && S >= 2) o.shard = KCFX_SHARD_COHORT;
    {
        std::vector<const char*> prefixes;
        for (const auto& p : kmc_list) prefixes.push_back(p.c_str());
        ctx_thread.join();
        g_background = nullptr;
        if (ctx_rc) error(ctx_err);
        mark("kcfx_ctx_create (rest of)");
        kcfx_shard_stats st;
        const int32_t rc = kcfx_screen_sharded(ctx, prefixes.data(), (int32_t)S, k, reinterpret_cast<const uint8_t*>(text.data()),
                                               (int64_t)text.size(), wb.data(), wl.data(), n, o.min_count, o.shard, all.data(), &st);
        const std::string msg = rc ? kcfx_last_error() : "";
        mark("kcfx_screen_sharded");
        kcfx_ctx_destroy(ctx);
        if (rc) error(msg);
        char line[256];
        std::snprintf(line, sizeof line, "Screened %zu sample(s) on %d device(s), %s-sharded: %.3f s (reference %.3f s, database loading %.3f s of which %.3f s not hidden behind screening, screening %.3f s)",
                      S, st.n_devices, st.shard_mode_used == KCFX_SHARD_SAMPLES ? "sample" : st.shard_mode_used == KCFX_SHARD_GRID ? "sample x window" :
                      st.shard_mode_used == KCFX_SHARD_COHORT ? "cohort-table x window" : st.shard_mode_used == KCFX_SHARD_TABLE ? "table" : "window", st.seconds_total, st.seconds_ref,
                      st.seconds_db_open_sum, st.seconds_exposed_load, st.seconds_screen_sum);
        info(line);
    }
    std::vector<const kcfx_result*> res(S);
    for (size_t s = 0; s < S; ++s) {
        res[s] = all.data() + s * (size_t)n;
        if (kcfx_score(res[s], n, o.wi, o.wt, o.wr, score[s].data(), depth[s].data())) error(kcfx_last_error());   // Data.java:101-103
    }

    hdr.n_windows = n;
    std::ofstream out(o.out, std::ios::binary);
    if (!out) error("Error writing KCF file: " + o.out);
    info("Writing KCF file: " + o.out);
    out << kcf_header_text(hdr);
    // the rows in file order: each contig's windows by start (:169-171; ties keep generation order) ...
    std::vector<size_t> order;
    order.reserve(wins.size());
    for (const auto& pc : per_contig) {
        const size_t at = order.size();
        for (size_t i = pc.first; i < pc.second; ++i) order.push_back(i);
        std::stable_sort(order.begin() + (std::ptrdiff_t)at, order.end(), [&](size_t a, size_t b) { return wins[a].start < wins[b].start; });
    }
    // ... formatted by a few threads, a contiguous range of rows each (Java's %.2f and Float.toString per sample column are most of
    // what a multi-sample run does after its last launch), written in order
    const int T = (int)std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)std::max(1u, std::thread::hardware_concurrency()), order.size() / 512 + 1}));
    std::vector<std::string> part((size_t)T);
    auto format_rows = [&](int t) {
        const size_t lo = order.size() * (size_t)t / (size_t)T, hi = order.size() * (size_t)(t + 1) / (size_t)T;
        std::string& dst = part[(size_t)t];
        dst.reserve((hi - lo) * (64 + 40 * S));
        for (size_t r = lo; r < hi; ++r) {
            const size_t i = order[r];
            if (S == 1) {
                dst += kcf_row_text(wins[i].seq, wins[i].start, wins[i].end, wins[i].id, res[0][i], score[0][i], depth[0][i]);
            } else {
                std::vector<SampleCol> cols;
                for (size_t s = 0; s < S; ++s) cols.push_back(cohort_sample(res[s][i], score[s][i], depth[s][i]));
                dst += kcf_row_text_multi(wins[i].seq, wins[i].start, wins[i].end, wins[i].id, res[0][i].total_kmers, res[0][i].eff_length, cols);
            }
            dst += '\n';
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < T; ++t) th.emplace_back(format_rows, t);
        format_rows(0);
        for (auto& x : th) x.join();
    }
    for (const auto& p : part) out.write(p.data(), (std::streamsize)p.size());
    out.close();
    if (!out) error("Error writing KCF file: " + o.out);
    mark("scores + KCF written");
    epoch("leaving");
    return 0;
}
