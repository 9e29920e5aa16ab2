// bank.cpp -- see bank.hpp.  zlib's gzread transparently reads plain files.
#include "bank.hpp"
#include "tool.hpp"

#include <sys/stat.h>
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <fstream>

namespace dsk {

void IBank::estimate(uint64_t& number, uint64_t& totalSize, uint64_t& maxSize) {
    uint64_t bases = 0, longest = 0;
    uint64_t n = stream(1 << 24, [&](const char* d, size_t nb) {
        size_t run = 0;
        for (size_t i = 0; i < nb; ++i) {
            if (d[i] == '\n') { longest = std::max<uint64_t>(longest, run); run = 0; }
            else { ++run; ++bases; }
        }
    });
    number = n; totalSize = bases; maxSize = longest;
}

namespace {

uint64_t file_size(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0; }
bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// One FASTA/FASTQ file, line-driven state machine over the inflated bytes.
class BankFasta : public IBank {
public:
    explicit BankFasta(const std::string& path) : path_(path) {
        if (!file_exists(path)) throw Exception("unable to open file '%s'", path.c_str());
    }
    std::string getId() const override { return path_; }
    uint64_t getSize() const override { return file_size(path_); }
    std::vector<std::string> files() const override { return {path_}; }

    uint64_t stream(size_t chunkBytes, const Sink& sink) override {
        gzFile f = gzopen(path_.c_str(), "rb");
        if (!f) throw Exception("unable to open file '%s'", path_.c_str());
        gzbuffer(f, 1 << 20);
        std::vector<char> raw(1 << 22);
        std::string out; out.reserve(chunkBytes + (1 << 20));
        std::string line;                       // carry of an incomplete line
        enum { HEADER, SEQ_FA, SEQ_FQ, QUAL } st = HEADER;
        uint64_t nseq = 0; size_t seqlen = 0, qleft = 0; bool open_rec = false;

        auto end_record = [&]() {
            if (!open_rec) return;
            out.push_back('\n'); ++nseq; open_rec = false; seqlen = 0;
            if (out.size() >= chunkBytes) { sink(out.data(), out.size()); out.clear(); }
        };
        auto append_seq = [&](const char* p, size_t n) {
            for (size_t i = 0; i < n; ++i) { char c = p[i]; if (c != '\r' && c != ' ' && c != '\t') { out.push_back(c); ++seqlen; } }
        };
        auto on_line = [&](const char* p, size_t n) {
            switch (st) {
                case HEADER:
                    if (n == 0) return;
                    if (p[0] == '>') { st = SEQ_FA; open_rec = true; seqlen = 0; }
                    else if (p[0] == '@') { st = SEQ_FQ; open_rec = true; seqlen = 0; }
                    return;
                case SEQ_FA:
                    if (n && p[0] == '>') { end_record(); st = SEQ_FA; open_rec = true; seqlen = 0; return; }
                    append_seq(p, n); return;
                case SEQ_FQ:
                    if (n && p[0] == '+') { st = QUAL; qleft = seqlen; if (qleft == 0) { end_record(); st = HEADER; } return; }
                    append_seq(p, n); return;
                case QUAL: {
                    size_t q = 0; for (size_t i = 0; i < n; ++i) if (p[i] != '\r') ++q;
                    if (q >= qleft) { end_record(); st = HEADER; } else qleft -= q;
                    return;
                }
            }
        };
        for (;;) {
            int got = gzread(f, raw.data(), (unsigned)raw.size());
            if (got < 0) { gzclose(f); throw Exception("read error in file '%s'", path_.c_str()); }
            if (got == 0) break;
            const char* p = raw.data(); const char* e = p + got;
            while (p < e) {
                const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
                if (!nl) { line.append(p, (size_t)(e - p)); break; }
                if (line.empty()) on_line(p, (size_t)(nl - p));
                else { line.append(p, (size_t)(nl - p)); on_line(line.data(), line.size()); line.clear(); }
                p = nl + 1;
            }
        }
        if (!line.empty()) { on_line(line.data(), line.size()); line.clear(); }
        end_record();
        gzclose(f);
        if (!out.empty()) sink(out.data(), out.size());
        return nseq;
    }
private:
    std::string path_;
};

// Concatenation of banks (comma list or album): one summed count
// (scripts/simple_test.sh:52 vs :36 share the same golden).
class BankComposite : public IBank {
public:
    BankComposite(const std::string& id, std::vector<IBank*> banks) : id_(id), banks_(std::move(banks)) {}
    ~BankComposite() override { for (auto* b : banks_) delete b; }
    std::string getId() const override { return id_; }
    uint64_t getSize() const override { uint64_t s = 0; for (auto* b : banks_) s += b->getSize(); return s; }
    std::vector<std::string> files() const override {
        std::vector<std::string> v; for (auto* b : banks_) { auto f = b->files(); v.insert(v.end(), f.begin(), f.end()); } return v;
    }
    uint64_t stream(size_t chunkBytes, const Sink& sink) override {
        uint64_t n = 0; for (auto* b : banks_) n += b->stream(chunkBytes, sink); return n;
    }
private:
    std::string id_; std::vector<IBank*> banks_;
};

std::string dirname_of(const std::string& p) { size_t s = p.find_last_of('/'); return s == std::string::npos ? "" : p.substr(0, s + 1); }

// An album is a text file whose non-empty lines name existing files
// (relative to the album's directory or to the cwd).
bool try_album(const std::string& path, std::vector<std::string>& members) {
    gzFile f = gzopen(path.c_str(), "rb");
    if (!f) return false;
    char buf[4096]; int got = gzread(f, buf, sizeof(buf) - 1); gzclose(f);
    if (got <= 0) return false;
    buf[got] = 0;
    if (buf[0] == '>' || buf[0] == '@') return false;
    for (int i = 0; i < got; ++i) if ((unsigned char)buf[i] < 9) return false;   // binary
    std::ifstream in(path);
    std::string line; std::vector<std::string> found;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
        if (line.empty()) continue;
        std::string cand = line;
        if (!file_exists(cand)) cand = dirname_of(path) + line;
        if (!file_exists(cand)) return false;
        found.push_back(cand);
    }
    if (found.empty()) return false;
    members = found;
    return true;
}

IBank* open_one(const std::string& path) {
    if (!file_exists(path)) throw Exception("unable to open file '%s'", path.c_str());
    std::vector<std::string> members;
    if (try_album(path, members)) {
        std::vector<IBank*> banks;
        for (auto& m : members) banks.push_back(new BankFasta(m));
        return new BankComposite(path, banks);
    }
    return new BankFasta(path);
}

}  // namespace

IBank* Bank::open(const std::string& uri) {
    std::vector<std::string> parts;
    size_t b = 0;
    while (b <= uri.size()) {
        size_t c = uri.find(',', b);
        if (c == std::string::npos) c = uri.size();
        if (c > b) parts.push_back(uri.substr(b, c - b));
        b = c + 1;
    }
    if (parts.empty()) throw Exception("empty bank uri");
    if (parts.size() == 1) return open_one(parts[0]);
    std::vector<IBank*> banks;
    try { for (auto& p : parts) banks.push_back(open_one(p)); }
    catch (...) { for (auto* x : banks) delete x; throw; }
    return new BankComposite(uri, banks);
}

}  // namespace dsk
