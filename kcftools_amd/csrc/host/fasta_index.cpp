#include "fasta_index.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

namespace kcfhost {
namespace {
// the whole file into `out` (FastaIndex.java maps it; here: pread by a few threads straight into the string -- a 119 Mb genome
// through ifstream -> ostringstream -> string was 90 ms of the run, on the host path that the GPU runtime's start no longer hides)
bool read_file(const std::string& path, std::string& out) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        ::close(fd);
        return false;
    }
    const int64_t size = (int64_t)st.st_size;
    out.resize((size_t)size);
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(4, size >> 24));      // >= 16 MiB per thread
    std::atomic<bool> ok{true};
    auto work = [&](int t) {
        int64_t got = size * t / T;
        const int64_t hi = size * (t + 1) / T;
        while (got < hi) {
            const ssize_t r = pread(fd, &out[(size_t)got], (size_t)(hi - got), got);
            if (r <= 0) {
                ok = false;
                return;
            }
            got += r;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    ::close(fd);
    return ok;
}
bool valid_base(char c) {       // FastaIndex.java:21-24
    return std::strchr("ACGTYRWSMKHBVDNacgtyrwsmkhbvdn", c) != nullptr && c != 0;
}
}  // namespace

bool FastaIndex::generate(const std::string& fasta_path, std::vector<FaidxEntry>& out, std::string& err) {
    std::string data;
    if (!read_file(fasta_path, data)) { err = "Error memory-mapping fasta file: " + fasta_path; return false; }
    if (data.size() >= 2 && (unsigned char)data[0] == 0x1f && (unsigned char)data[1] == 0x8b) {   // :240-243
        err = "Fasta file is compressed. Please decompress before indexing: " + fasta_path;
        return false;
    }
    out.clear();
    int64_t offset = 0;
    bool in_seq = false;
    FaidxEntry cur;
    size_t pos = 0;
    int64_t line_no = 0;
    while (pos < data.size()) {                      // BufferedReader.readLine: \n, \r\n or \r end a line
        size_t eol = data.find('\n', pos);
        size_t next = (eol == std::string::npos) ? data.size() : eol + 1;
        std::string line = data.substr(pos, (eol == std::string::npos ? data.size() : eol) - pos);
        if (!line.empty() && line.back() == '\r') line.pop_back();
        pos = next;
        ++line_no;
        if (offset == 0 && (line.empty() || line[0] != '>')) { err = "Invalid fasta file: " + fasta_path; return false; }   // :261-263
        if (!line.empty() && line[0] == '>') {
            if (in_seq) out.push_back(cur);
            cur = FaidxEntry();
            const size_t sp = line.find(' ', 1);
            cur.name = line.substr(1, sp == std::string::npos ? std::string::npos : sp - 1);       // :268
            offset += (int64_t)line.size() + 1;                                                     // :269 (1-byte separator)
            cur.offset = offset;
            in_seq = true;
            for (const auto& e : out)
                if (e.name == cur.name) { err = "Duplicate sequence name in fasta file: " + cur.name + " at line " + std::to_string(line_no); return false; }
        } else {
            for (char c : line)
                if (!valid_base(c)) { err = std::string("Invalid character '") + c + "' in fasta file: " + fasta_path + " at line " + std::to_string(line_no); return false; }
            if (cur.length == 0) {                                                                  // :287-290
                cur.line_bases = (int32_t)line.size();
                cur.line_width = (int32_t)line.size() + 1;
            }
            cur.length += (int32_t)line.size();
            offset += (int64_t)line.size() + 1;
        }
    }
    if (in_seq) out.push_back(cur);
    return true;
}

std::string FastaIndex::to_text(const std::vector<FaidxEntry>& entries) {
    std::string s;
    for (const auto& e : entries)
        s += e.name + "\t" + std::to_string(e.length) + "\t" + std::to_string(e.offset) + "\t" + std::to_string(e.line_bases) +
             "\t" + std::to_string(e.line_width) + "\n";
    return s;
}

bool FastaIndex::open(const std::string& fasta_path, std::string& err) {
    const std::string fai = fasta_path + ".faidx";
    struct stat sf, si;
    if (stat(fasta_path.c_str(), &sf) != 0) { err = "Error memory-mapping fasta file: " + fasta_path; return false; }
    const bool have = stat(fai.c_str(), &si) == 0;
    const auto ms = [](const struct stat& s) { return (int64_t)s.st_mtim.tv_sec * 1000 + s.st_mtim.tv_nsec / 1000000; };
    if (!have || ms(si) < ms(sf)) {                                                                 // :31-36
        if (!generate(fasta_path, entries_, err)) return false;
        std::ofstream o(fai, std::ios::binary);
        o << to_text(entries_);
    } else {                                                                                        // loadIndex :82-103
        entries_.clear();
        std::ifstream in(fai);
        std::string line;
        while (std::getline(in, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();
            std::vector<std::string> f;
            size_t p = 0;
            for (;;) {
                size_t t = line.find('\t', p);
                f.push_back(line.substr(p, t == std::string::npos ? std::string::npos : t - p));
                if (t == std::string::npos) break;
                p = t + 1;
            }
            if (f.size() < 5) { err = "Malformed index line in " + fai; return false; }
            FaidxEntry e;
            e.name = f[0];
            try {
                e.length = std::stoi(f[1]);
                e.offset = std::stoll(f[2]);
                e.line_bases = std::stoi(f[3]);
                e.line_width = std::stoi(f[4]);
            } catch (...) { err = "Malformed index line in " + fai; return false; }
            for (const auto& x : entries_)
                if (x.name == e.name) { err = "Duplicate sequence name in index: " + e.name; return false; }
            entries_.push_back(e);
        }
    }
    if (!read_file(fasta_path, data_)) { err = "Error memory-mapping fasta file: " + fasta_path; return false; }
    return true;
}

const FaidxEntry* FastaIndex::find(const std::string& name) const {
    for (const auto& e : entries_)
        if (e.name == name) return &e;
    return nullptr;
}

bool FastaIndex::get_sequence(const FaidxEntry& e, int64_t start, int64_t length, std::string& out, std::string& err) const {
    out.clear();
    return append_sequence(e, start, length, out, err);
}

bool FastaIndex::append_sequence(const FaidxEntry& e, int64_t start, int64_t length, std::string& out, std::string& err) const {
    const int64_t end = start + length;
    if (start < 0 || end > e.length || start >= end) {                                              // :131-134
        err = "Invalid range: " + std::to_string(start) + "-" + std::to_string(end) + " for sequence: " + e.name;
        return false;
    }
    const size_t before = out.size();
    out.resize(before + (size_t)length);
    char* dst = &out[before];
    int64_t pos = e.offset + (start / e.line_bases) * e.line_width + start % e.line_bases;           // :148-154
    int64_t in_line = start % e.line_bases;
    int64_t todo = length;
    while (todo > 0) {
        const int64_t n = std::min<int64_t>(todo, e.line_bases - in_line);                          // :160-161
        if (pos < 0 || pos + n > (int64_t)data_.size()) {                                          // :175-178
            out.resize(before);
            err = "Error reading sequence: position out of file for " + e.name;
            return false;
        }
        std::memcpy(dst, data_.data() + pos, (size_t)n);
        dst += n;
        pos += n + (e.line_width - e.line_bases);                                                   // :169
        todo -= n;
        in_line = 0;
    }
    return true;
}

}  // namespace kcfhost
