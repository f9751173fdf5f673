// gtf.h -- GTF gene / transcript model as Data/GTF.java builds it: chrom -> gene -> transcript -> exon,
// children in file order; a window's text is the concatenation of the merged exon loci.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace kcfhost {

struct GtfFeature {
    std::string chrom;
    int32_t start = 0, end = 0;   // 1-based inclusive, as in the file
    char strand = '.';
    std::string type;
};
struct GtfLoci {
    std::string chrom;
    int32_t start = 0, end = 0;
    char strand = '.';
};

class Gtf {
  public:
    bool parse(const std::string& path, std::string& err);                       // GTF.java:26-100
    std::vector<std::string> children(const std::string& id) const;              // :207-217
    bool has_vertex(const std::string& id) const { return children_.count(id) != 0; }
    const GtfFeature* feature(const std::string& id) const;                      // getLoci :299-306
    // getFasta (:223-239): de-duplicated exon loci, sorted by (chrom, start), merged when overlapping on one strand
    bool merged_exon_loci(const std::string& id, bool is_gene, std::vector<GtfLoci>& out) const;

  private:
    void vertex(const std::string& v);
    void edge(const std::string& a, const std::string& b);
    std::map<std::string, std::vector<std::string>> children_;
    std::map<std::string, GtfFeature> features_;
};

}  // namespace kcfhost
