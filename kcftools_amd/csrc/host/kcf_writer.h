// kcf_writer.h -- KCF text as Data/KCFHeader.java:291-330, Data/Window.java:125-152,177-214 and Data/Data.java:120-132
// render it for a single-sample getVariations run.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/kcfx.h"

namespace kcfhost {

struct KcfHeaderInfo {
    std::string reference, sample, command_line, date;   // date: yyyy-MM-dd (HelperFunctions.getTodayDate)
    std::vector<std::pair<std::string, int32_t>> contigs;
    int32_t window = 0, step = 0, kmer = 0, n_windows = 0;
    double wi = 0.3, wt = 0.3, wr = 0.4;
};

std::string kcf_header_text(const KcfHeaderInfo& h);
std::string kcf_row_text(const std::string& seq, int32_t start, int32_t end, const std::string& id, const kcfx_result& r,
                         double score, double mean_count);
// One sample column of a multi-sample (cohort) row.
struct SampleCol {
    int32_t var = 0, obs = 0, inner = 0, left = 0, right = 0;
    double mean = 0, score = 0;
};
// What Plugins/Cohort.java keeps of a single-sample row when it re-parses the KCF text (Window.java:57-83,
// Data.java:40-68): the count sum is re-derived from the 2-decimal KD text as Math.round(KD * obs).
SampleCol cohort_sample(const kcfx_result& r, double score, double mean_count);
// Window.toString for several samples (Window.java:125-152, calculateStats :177-214) -- the row `cohort` writes.
std::string kcf_row_text_multi(const std::string& seq, int32_t start, int32_t end, const std::string& id, int32_t total,
                               int32_t eff, const std::vector<SampleCol>& samples);
std::string today();

}  // namespace kcfhost
