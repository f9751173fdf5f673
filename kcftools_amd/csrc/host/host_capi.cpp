// host_capi.cpp -- C surface over the host-side modules (FASTA index, GTF, KCF text, Java number formats) so that the
// CPU tests can drive them through ctypes. Not part of include/kcfx.h (these rows sit beside the hot path).
#include <cstring>
#include <string>
#include <vector>

#include "fasta_index.h"
#include "gtf.h"
#include "java_format.h"
#include "kcf_writer.h"

using namespace kcfhost;

namespace {
thread_local std::string g_out;
const char* ret(const std::string& s) { g_out = s; return g_out.c_str(); }
}  // namespace

extern "C" {
const char* kcfh_fmt_2f(double x) { return ret(java_fmt_2f(x)); }
const char* kcfh_double_to_string(double x) { return ret(java_double_to_string(x)); }
const char* kcfh_float_to_string(float x) { return ret(java_float_to_string(x)); }
const char* kcfh_jdk17_fmt_2f(double x) { return ret(jdk17_fmt_2f(x)); }
const char* kcfh_jdk17_double_to_string(double x) { return ret(jdk17_double_to_string(x)); }
const char* kcfh_jdk17_float_to_string(float x) { return ret(jdk17_float_to_string(x)); }
// bulk form for the enumeration tests: how many of xs[0 .. n) the two digit generators format differently as %.2f; the first
// such value in *first_bad
long kcfh_fmt_2f_disagreements(const double* xs, long n, double* first_bad) {
    long bad = 0;
    for (long i = 0; i < n; ++i)
        if (java_fmt_2f(xs[i]) != jdk17_fmt_2f(xs[i])) {
            if (bad == 0 && first_bad) *first_bad = xs[i];
            ++bad;
        }
    return bad;
}
// ... and as Double.toString (1) / how often the JDK 17 digits are NOT the shortest (2: a longer string, or another last digit)
long kcfh_double_to_string_disagreements(const double* xs, long n, double* first_bad) {
    long bad = 0;
    for (long i = 0; i < n; ++i)
        if (java_double_to_string(xs[i]) != jdk17_double_to_string(xs[i])) {
            if (bad == 0 && first_bad) *first_bad = xs[i];
            ++bad;
        }
    return bad;
}

// "" on success else the error text; *out = .faidx text
const char* kcfh_faidx(const char* fasta, const char** out) {
    static thread_local std::string text;
    std::vector<FaidxEntry> e;
    std::string err;
    if (!FastaIndex::generate(fasta, e, err)) return ret(err);
    text = FastaIndex::to_text(e);
    *out = text.c_str();
    return ret("");
}
// whole sequence [start, start+len) of `name`
const char* kcfh_get_sequence(const char* fasta, const char* name, long start, long len, const char** out) {
    static thread_local std::string seq;
    FastaIndex idx;
    std::string err;
    if (!idx.open(fasta, err)) return ret(err);
    const FaidxEntry* e = idx.find(name);
    if (!e) return ret(std::string("Sequence not found in index: ") + name);
    if (!idx.get_sequence(*e, start, len, seq, err)) return ret(err);
    *out = seq.c_str();
    return ret("");
}
// gene / transcript windows of one chromosome: lines "id\tchrom\tstart\tend\tlocus;locus;..." (locus = chrom:start-end[strand])
const char* kcfh_gtf_windows(const char* gtf_path, const char* chrom, int is_gene, const char** out) {
    static thread_local std::string text;
    Gtf g;
    std::string err;
    if (!g.parse(gtf_path, err)) return ret(err);
    text.clear();
    for (const auto& gene : g.children(chrom)) {
        std::vector<std::string> ids = is_gene ? std::vector<std::string>{gene} : g.children(gene);
        for (const auto& id : ids) {
            const GtfFeature* f = g.feature(id);
            if (!f) return ret("Feature ID not found: " + id);
            std::vector<GtfLoci> loci;
            text += id + "\t" + f->chrom + "\t" + std::to_string(f->start) + "\t" + std::to_string(f->end) + "\t";
            if (g.merged_exon_loci(id, is_gene != 0, loci))
                for (size_t i = 0; i < loci.size(); ++i)
                    text += (i ? ";" : "") + loci[i].chrom + ":" + std::to_string(loci[i].start) + "-" + std::to_string(loci[i].end) + "[" + loci[i].strand + "]";
            text += "\n";
        }
    }
    *out = text.c_str();
    return ret("");
}
const char* kcfh_header(const char* reference, const char* sample, const char* cmd, const char* date, int n_contigs,
                        const char** names, const int* lengths, int window, int step, int kmer, int n_windows, double wi, double wt,
                        double wr) {
    KcfHeaderInfo h;
    h.reference = reference; h.sample = sample; h.command_line = cmd; h.date = date;
    for (int i = 0; i < n_contigs; ++i) h.contigs.emplace_back(names[i], lengths[i]);
    h.window = window; h.step = step; h.kmer = kmer; h.n_windows = n_windows; h.wi = wi; h.wt = wt; h.wr = wr;
    return ret(kcf_header_text(h));
}
const char* kcfh_row(const char* seq, int start, int end, const char* id, const kcfx_result* r, double score, double mean) {
    return ret(kcf_row_text(seq, start, end, id, *r, score, mean));
}
}
