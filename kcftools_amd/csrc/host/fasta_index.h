// fasta_index.h -- <fasta>.faidx index + sequence access compatible with Data/FastaIndex.java.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace kcfhost {

struct FaidxEntry {            // FastaIndexEntry.java: name, length, offset, lineBases, lineWidth
    std::string name;
    int32_t length = 0;
    int64_t offset = 0;
    int32_t line_bases = 0;
    int32_t line_width = 0;
};

class FastaIndex {
  public:
    // FastaIndex.java:26-77: use <fasta>.faidx if it exists and is not older than the FASTA, else (re)generate it.
    // Returns false and sets `err` where the reference calls Logger.error (which exits).
    bool open(const std::string& fasta_path, std::string& err);
    const std::vector<FaidxEntry>& entries() const { return entries_; }
    const FaidxEntry* find(const std::string& name) const;
    // FastaIndex.getSequence (:122-182): bases [start, start+length) by (offset, lineBases, lineWidth) arithmetic
    bool get_sequence(const FaidxEntry& e, int64_t start, int64_t length, std::string& out, std::string& err) const;
    // the same bases appended to `out` (the driver builds one text of all windows: no second copy of a chromosome)
    bool append_sequence(const FaidxEntry& e, int64_t start, int64_t length, std::string& out, std::string& err) const;
    static bool generate(const std::string& fasta_path, std::vector<FaidxEntry>& out, std::string& err);   // :239-299
    static std::string to_text(const std::vector<FaidxEntry>& entries);

  private:
    std::vector<FaidxEntry> entries_;
    std::string data_;         // the whole FASTA file
};

}  // namespace kcfhost
