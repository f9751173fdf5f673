#include "gtf.h"

#include <algorithm>
#include <fstream>

namespace kcfhost {
namespace {
std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    size_t p = 0;
    for (;;) {
        size_t t = s.find(sep, p);
        out.push_back(s.substr(p, t == std::string::npos ? std::string::npos : t - p));
        if (t == std::string::npos) break;
        p = t + 1;
    }
    return out;
}
std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (unsigned char)s[a] <= ' ') ++a;
    while (b > a && (unsigned char)s[b - 1] <= ' ') --b;
    return s.substr(a, b - a);
}
bool is_gene_type(const std::string& t) { return t == "gene" || t == "pseudogene"; }
bool is_transcript_type(const std::string& t) {
    return t == "transcript" || t == "mRNA" || t == "RNA" || t == "lnc_RNA" || t == "rRNA" || t == "tRNA" || t == "snRNA" || t == "snoRNA";
}
}  // namespace

void Gtf::vertex(const std::string& v) { children_[v]; }
void Gtf::edge(const std::string& a, const std::string& b) {
    vertex(b);
    auto& c = children_[a];
    if (std::find(c.begin(), c.end(), b) == c.end()) c.push_back(b);
}

bool Gtf::parse(const std::string& path, std::string& err) {
    std::ifstream in(path);
    if (!in) { err = "Error parsing GTF file: " + path; return false; }
    std::map<std::string, int> exon_counts;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if ((!line.empty() && line[0] == '#') || trim(line).empty()) continue;                  // :35
        const auto f = split(line, '\t');
        if (f.size() < 9) { err = "Malformed line: " + line; return false; }                   // :37-40
        std::map<std::string, std::string> attrs;                                               // :156-163
        for (const auto& a : split(f[8], ';')) {
            std::string t = trim(a);
            t.erase(std::remove(t.begin(), t.end(), '"'), t.end());
            const auto pair = split(t, ' ');
            if (pair.size() == 2) attrs[pair[0]] = pair[1];
        }
        const std::string &type = f[2], &chrom = f[0];
        vertex(chrom);                                                                          // :46
        std::string fid, parent;
        bool has_parent = true;
        int32_t s, e;
        try { s = std::stoi(f[3]); e = std::stoi(f[4]); } catch (...) { err = "Malformed line: " + line; return false; }
        const char strand = f[6].empty() ? '.' : f[6][0];
        if (is_gene_type(type)) {                                                               // :49-52
            fid = attrs["gene_id"];
            parent = chrom;
        } else if (is_transcript_type(type)) {                                                  // :53-76
            fid = attrs["transcript_id"];
            parent = attrs["gene_id"];
            if (fid == parent) { err = "Transcript ID is the same as Gene ID: " + fid + ". Fix the GTF file using AGAT."; return false; }
            if (!has_vertex(parent)) {
                vertex(parent);
                features_[parent] = GtfFeature{chrom, s, e, strand, "gene"};
                edge(chrom, parent);
            }
            auto g = features_.find(parent);
            if (g != features_.end()) {
                g->second.start = std::min(g->second.start, s);
                g->second.end = std::max(g->second.end, e);
            }
        } else if (type == "exon") {                                                            // :77-82
            parent = attrs["transcript_id"];
            fid = parent + "-e-" + std::to_string(++exon_counts[parent]);
        } else {
            continue;
        }
        features_[fid] = GtfFeature{chrom, s, e, strand, type};                                 // :87-90
        vertex(fid);
        if (has_parent) { vertex(parent); edge(parent, fid); }                                  // :92-95
    }
    return true;
}

std::vector<std::string> Gtf::children(const std::string& id) const {
    std::vector<std::string> out;
    auto it = children_.find(id);
    if (it == children_.end()) return out;
    for (const auto& c : it->second)
        if (c != id) out.push_back(c);
    return out;
}

const GtfFeature* Gtf::feature(const std::string& id) const {
    auto it = features_.find(id);
    return it == features_.end() ? nullptr : &it->second;
}

bool Gtf::merged_exon_loci(const std::string& id, bool is_gene, std::vector<GtfLoci>& out) const {
    out.clear();
    if (!has_vertex(id)) return false;
    std::vector<GtfLoci> loci;
    for (const auto& t : children(id)) {
        const std::vector<std::string> exons = is_gene ? children(t) : std::vector<std::string>{t};
        for (const auto& ex : exons) {
            const GtfFeature* f = feature(ex);
            if (!f) continue;
            bool dup = false;                                                                   // HashSet<Loci> :225
            for (const auto& l : loci) dup = dup || (l.chrom == f->chrom && l.start == f->start && l.end == f->end && l.strand == f->strand);
            if (!dup) loci.push_back(GtfLoci{f->chrom, f->start, f->end, f->strand});
        }
    }
    if (loci.empty()) return false;                                                             // :236
    std::stable_sort(loci.begin(), loci.end(), [](const GtfLoci& a, const GtfLoci& b) {        // Loci.compareTo :413-419
        return a.chrom != b.chrom ? a.chrom < b.chrom : a.start < b.start;
    });
    for (const auto& cur : loci) {                                                              // :278-293
        if (!out.empty()) {
            GtfLoci& last = out.back();
            if (last.chrom == cur.chrom && last.strand == cur.strand && last.start <= cur.end && cur.start <= last.end) {
                last.start = std::min(last.start, cur.start);
                last.end = std::max(last.end, cur.end);
                continue;
            }
        }
        out.push_back(cur);
    }
    return true;
}

}  // namespace kcfhost
