#include "kcf_writer.h"

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

}  // namespace kcfhost
