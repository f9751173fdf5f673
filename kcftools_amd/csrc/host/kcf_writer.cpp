#include "kcf_writer.h"

#include <climits>
#include <cmath>
#include <ctime>

#include "java_format.h"

namespace kcfhost {
namespace {
// Utils/Configs.java:14-37 (text blocks: indentation stripped)
const char* INFO_LINES[] = {
    "<ID=EFFLEN,Type=Integer,Description=\"Effective length of the window\">",
    "<ID=IS,Type=Float,Description=\"Minimum score for the window\">",
    "<ID=XS,Type=Float,Description=\"Maximum score for the window\">",
    "<ID=MS,Type=Float,Description=\"Mean score for the window\">",
    "<ID=IO,Type=Integer,Description=\"Minimum observed kmers in the window\">",
    "<ID=XO,Type=Integer,Description=\"Maximum observed kmers in the window\">",
    "<ID=MO,Type=Integer,Description=\"Mean observed kmers in the window\">",
    "<ID=IV,Type=Integer,Description=\"Minimum variations in the window\">",
    "<ID=XV,Type=Integer,Description=\"Maximum variations in the window\">",
    "<ID=MV,Type=Integer,Description=\"Mean variations in the window\">"};
const char* FORMAT_LINES[] = {
    "<ID=IB,Type=Integer,Description=\"IBS number\">",
    "<ID=VA,Type=Integer,Description=\"Variations\">",
    "<ID=OB,Type=Integer,Description=\"Observed kmers\">",
    "<ID=ID,Type=Integer,Description=\"Inner Distance\">",
    "<ID=LD,Type=Integer,Description=\"Kmer Variation Distance at the leftTail\">",
    "<ID=RD,Type=Integer,Description=\"Kmer Variation Distance at the rightTail\">",
    "<ID=KD,Type=Float,Description=\"Mean Kmer Depth\">",
    "<ID=SC,Type=Float,Description=\"Score\">"};
const char* KCF_VERSION = "0.4.0";      // pom.xml:8 via version.properties (Configs.getVersion)
}  // namespace

std::string today() {
    std::time_t t = std::time(nullptr);
    std::tm tm;
    localtime_r(&t, &tm);
    char buf[32];
    std::strftime(buf, sizeof buf, "%Y-%m-%d", &tm);
    return buf;
}

// KCFHeader.toString (:291-330); PARAM order = params[0..7] (:28): window, step, kmer, IBS, nwindow, wti, wtt, wtk
std::string kcf_header_text(const KcfHeaderInfo& h) {
    std::string s;
    s += std::string("##format=KCF") + KCF_VERSION + "\n";
    s += "##date=" + (h.date.empty() ? today() : h.date) + "\n";
    s += "##source=kcftools\n";
    s += "##reference=" + h.reference + "\n";
    for (const auto& c : h.contigs) s += "##contig=<ID=" + c.first + ",length=" + std::to_string(c.second) + ">\n";
    for (const char* l : INFO_LINES) s += std::string("##INFO=") + l + "\n";
    for (const char* l : FORMAT_LINES) s += std::string("##FORMAT=") + l + "\n";
    auto param = [&](const char* k, const std::string& v) { s += std::string("##PARAM=<ID=") + k + ",value=" + v + ">\n"; };
    param("window", std::to_string(h.window));
    param("step", std::to_string(h.step));
    param("kmer", std::to_string(h.kmer));
    param("IBS", "false");                                   // GetVariants.java:104
    param("nwindow", std::to_string(h.n_windows));
    param("wti", java_double_to_string(h.wi));               // String.valueOf(double) KCFHeader.java:165
    param("wtt", java_double_to_string(h.wt));
    param("wtk", java_double_to_string(h.wr));
    s += "##CMD=" + h.command_line + "\n";
    s += "#CHROM\tSTART\tEND\tID\tTOTAL_KMERS\tINFO\tFORMAT\t" + h.sample + "\n";
    return s;
}

// Window.toString (:125-138) + getInfoField (:140-152) + calculateStats for ONE sample (:177-214) + Data.toString (:120-132)
std::string kcf_row_text(const std::string& seq, int32_t start, int32_t end, const std::string& id, const kcfx_result& r,
                         double score, double mean_count) {
    const double FLOAT_MIN_VALUE = 1.401298464324817e-45;    // maxScore starts at Float.MIN_VALUE (:185): XS prints 0.00 for scores <= 0
    const double xs = score > FLOAT_MIN_VALUE ? score : FLOAT_MIN_VALUE;
    std::string s = seq + "\t" + std::to_string(start) + "\t" + std::to_string(end) + "\t" + id + "\t" + std::to_string(r.total_kmers) + "\t";
    s += "EFFLEN=" + std::to_string(r.eff_length) + ";IS=" + java_fmt_2f(score) + ";XS=" + java_fmt_2f(xs) + ";MS=" + java_fmt_2f(score) +
         ";IO=" + std::to_string(r.obs) + ";XO=" + std::to_string(r.obs) + ";MO=" + java_fmt_2f((double)(float)r.obs) +
         ";IV=" + std::to_string(r.var) + ";XV=" + std::to_string(r.var) + ";MV=" + java_float_to_string((float)r.var);
    s += "\tGT:VA:OB:ID:LD:RD:KD:SC\t";
    s += "N:" + std::to_string(r.var) + ":" + std::to_string(r.obs) + ":" + std::to_string(r.inner) + ":" + std::to_string(r.left) + ":" +
         std::to_string(r.right) + ":" + java_fmt_2f(mean_count) + ":" + java_fmt_2f(score);
    return s;
}

SampleCol cohort_sample(const kcfx_result& r, double score, double mean_count) {
    SampleCol c;
    c.var = r.var; c.obs = r.obs; c.inner = r.inner; c.left = r.left; c.right = r.right;
    const double kd = std::stod(java_fmt_2f(mean_count));                       // Double.parseDouble(sampleData[6])
    const long long kmer_count = (long long)std::floor(kd * r.obs + 0.5);       // Math.round (Window.java:70)
    c.mean = kmer_count > 0 ? (double)kmer_count / r.obs : 0.0;                 // Data.java:57-59
    c.score = score;                                                            // same integers, same weights (Data.java:61)
    return c;
}

std::string kcf_row_text_multi(const std::string& seq, int32_t start, int32_t end, const std::string& id, int32_t total,
                               int32_t eff, const std::vector<SampleCol>& samples) {
    double min_s = 3.4028234663852886e38, max_s = 1.401298464324817e-45, mean_s = 0;   // Float.MAX_VALUE / MIN_VALUE (:184-185)
    int32_t min_o = INT32_MAX, max_o = INT32_MIN, min_v = INT32_MAX, max_v = INT32_MIN;
    float mo = 0, mv = 0;                                                       // float accumulators (:24,27)
    for (const auto& d : samples) {
        if (d.obs < min_o) min_o = d.obs;
        if (d.obs > max_o) max_o = d.obs;
        mo += (float)d.obs;
        if (d.var < min_v) min_v = d.var;
        if (d.var > max_v) max_v = d.var;
        mv += (float)d.var;
        if (d.score < min_s) min_s = d.score;
        if (d.score > max_s) max_s = d.score;
        mean_s += d.score;
    }
    const int n = (int)samples.size();
    mo /= (float)n;
    mv /= (float)n;
    mean_s /= n;
    std::string s = seq + "\t" + std::to_string(start) + "\t" + std::to_string(end) + "\t" + id + "\t" + std::to_string(total) + "\t";
    s += "EFFLEN=" + std::to_string(eff) + ";IS=" + java_fmt_2f(min_s) + ";XS=" + java_fmt_2f(max_s) + ";MS=" + java_fmt_2f(mean_s) +
         ";IO=" + std::to_string(min_o) + ";XO=" + std::to_string(max_o) + ";MO=" + java_fmt_2f((double)mo) + ";IV=" +
         std::to_string(min_v) + ";XV=" + std::to_string(max_v) + ";MV=" + java_float_to_string(mv);
    s += "\tGT:VA:OB:ID:LD:RD:KD:SC";
    for (const auto& d : samples)
        s += "\tN:" + std::to_string(d.var) + ":" + std::to_string(d.obs) + ":" + std::to_string(d.inner) + ":" + std::to_string(d.left) + ":" +
             std::to_string(d.right) + ":" + java_fmt_2f(d.mean) + ":" + java_fmt_2f(d.score);
    return s;
}

}  // namespace kcfhost
