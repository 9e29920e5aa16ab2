// bank.cpp -- see bank.hpp.  zlib's gzread transparently reads plain files.
#include "bank.hpp"
#include "tool.hpp"

#include <sys/stat.h>
#include <zlib.h>
#include "pgzip.hpp"
#include <stdexcept>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <exception>
#include <cstring>
#include <fstream>
#include <mutex>
#include <thread>
#include <atomic>
#include <condition_variable>

namespace dsk {

void IBank::estimate(uint64_t& number, uint64_t& totalSize, uint64_t& maxSize) {
    uint64_t bases = 0, longest = 0;
    const Sink sink = [&](const char* d, size_t nb) {
        size_t run = 0;
        for (size_t i = 0; i < nb; ++i) {
            if (d[i] == '\n') { longest = std::max<uint64_t>(longest, run); run = 0; }
            else { ++run; ++bases; }
        }
    };
    bool exact = true;
    uint64_t n = stream(1 << 24, sink, &exact);
    if (!exact) { bases = longest = 0; n = streamSerial(1 << 24, sink); }      // (a damaged file: what ONE thread reads, see IBank::stream)
    number = n; totalSize = bases; maxSize = longest;
}

namespace {

uint64_t file_size(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0; }
bool file_exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode); }

// Line-driven FASTA/FASTQ state machine; emits "<sequence>\n" per record into `out`.
struct RecordParser {
    enum { HEADER, SEQ_FA, SEQ_FQ, QUAL } st = HEADER;
    std::string out, line;
    uint64_t nseq = 0; size_t seqlen = 0, qleft = 0; bool open_rec = false;
    size_t chunkBytes; const IBank::Sink* sink; std::mutex* mu;
    RecordParser(size_t chunk, const IBank::Sink* s, std::mutex* m) : chunkBytes(chunk), sink(s), mu(m) { out.reserve(chunk + (1 << 20)); }
    void flush() {
        if (out.empty()) return;
        if (mu) { std::lock_guard<std::mutex> g(*mu); (*sink)(out.data(), out.size()); }
        else (*sink)(out.data(), out.size());
        out.clear();
    }
    void end_record() {
        if (!open_rec) return;
        out.push_back('\n'); ++nseq; open_rec = false; seqlen = 0;
        if (out.size() >= chunkBytes) flush();
    }
    void append_seq(const char* p, size_t n) {
        // the rule: a sequence line holds no blank and no CR and is copied as it is (one vectorised scan + memcpy: the per-character
        // loop below kept 32 parser threads at 12 GB/s on a 3 GB FASTQ file -- longer than the device runtime takes to start)
        unsigned char any = 0;
        for (size_t i = 0; i < n; ++i) any |= (unsigned char)((unsigned char)p[i] <= (unsigned char)' ');
        if (!any) { out.append(p, n); seqlen += n; return; }
        for (size_t i = 0; i < n; ++i) { char c = p[i]; if (c != '\r' && c != ' ' && c != '\t') { out.push_back(c); ++seqlen; } }
    }
    void on_line(const char* p, size_t n) {
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
                size_t cr = 0; for (size_t i = 0; i < n; ++i) cr += (size_t)(p[i] == '\r');      // (a counting loop the compiler vectorises)
                const size_t q = n - cr;
                if (q >= qleft) { end_record(); st = HEADER; } else qleft -= q;
                return;
            }
        }
    }
    void feed(const char* p, size_t n) {
        const char* e = p + n;
        while (p < e) {
            const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
            if (!nl) { line.append(p, (size_t)(e - p)); break; }
            if (line.empty()) on_line(p, (size_t)(nl - p));
            else { line.append(p, (size_t)(nl - p)); on_line(line.data(), line.size()); line.clear(); }
            p = nl + 1;
        }
    }
    void finish() {
        if (!line.empty()) { on_line(line.data(), line.size()); line.clear(); }
        end_record();
        flush();
    }
    // Is the parser between two records?  A range of a file cut at record starts ends here exactly when the parser was in step with the
    // cutter: then the next range, parsed from scratch, continues as this parser would have (so all ranges together = one serial parse).
    // (kind = what the cutter cuts at: a FASTA record also ends where the next '>' line begins, and only there)
    bool at_record_border(char kind) const { return line.empty() && ((st == HEADER && !open_rec) || (kind == '>' && st == SEQ_FA)); }
};

// First record start at or after `from` in an uncompressed buffer (used to cut a file into
// independently parsable ranges).  FASTA: a line starting with '>'.  FASTQ (4-line records): a line
// starting with '@' whose second following line starts with '+' -- a quality line that happens to
// start with '@' fails that test because two lines below it comes a sequence line.
const char* next_record_start(const char* base, const char* from, const char* end, char kind) {
    const char* p = from;
    if (p > base) {                               // move to the next line start
        const char* nl = (const char*)memchr(p - 1, '\n', (size_t)(end - (p - 1)));
        if (!nl) return end;
        p = nl + 1;
    }
    while (p < end) {
        if (*p == kind) {
            if (kind == '>') return p;
            const char* l1 = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!l1) return end;
            const char* l2 = (const char*)memchr(l1 + 1, '\n', (size_t)(end - (l1 + 1)));
            if (!l2) return end;
            if (l2 + 1 < end && l2[1] == '+') return p;
        }
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!nl) return end;
        p = nl + 1;
    }
    return end;
}

// One FASTA/FASTQ file.  gzip'ed: inflate overlapped with parsing, BGZF members inflated by a thread pool.  Plain and large: memory-mapped
// and parsed by several threads on record-aligned ranges (the sink is serialised by a mutex;
// counting does not depend on record order).
// threads of the parallel inflate: pure CPU work on independent chunks -- it takes up to 64 where the parser is capped at 32
static unsigned pgz_threads(unsigned nthreads, unsigned hw) {
    const unsigned cap = 64;      // (r06, 81 MB E. coli .fastq.gz on 2 x 64 cores: 96 / 128 threads inflate in 26 / 13 ms instead of 42 and lose it again in the byte pass -- and the ingest waits for the device runtime either way)
    return std::min(std::max(nthreads, std::min(hw, cap)), cap);
}

// gzread returns 0 at a premature end of a gzip stream exactly as at its real end; only gzerror tells them apart (Z_BUF_ERROR: the
// file ended in the middle of a stream, Z_DATA_ERROR: corrupt).  A truncated download must not count as a smaller input.
static bool gz_ended_badly(gzFile f) { int err = Z_OK; (void)gzerror(f, &err); return err == Z_BUF_ERROR || err == Z_DATA_ERROR; }

class BankFasta : public IBank {
public:
    explicit BankFasta(const std::string& path) : path_(path) {
        if (!file_exists(path)) throw Exception("unable to open file '%s'", path.c_str());
    }
    std::string getId() const override { return path_; }
    uint64_t getSize() const override { return file_size(path_); }
    std::vector<std::string> files() const override { return {path_}; }

    uint64_t streamSerial(size_t chunkBytes, const Sink& sink) override { return stream_impl(chunkBytes, sink, 1, nullptr); }
    uint64_t stream(size_t chunkBytes, const Sink& sink, bool* exact = nullptr) override { return stream_impl(chunkBytes, sink, Bank::parseThreads(), exact); }
    uint64_t stream_impl(size_t chunkBytes, const Sink& sink, unsigned nthreads, bool* exact) {
        if (exact) *exact = true;
        const uint64_t size = file_size(path_);
        unsigned char magic[2] = {0, 0};
        { FILE* f = fopen(path_.c_str(), "rb"); if (f) { size_t got = fread(magic, 1, 2, f); (void)got; fclose(f); } }
        const bool gz = magic[0] == 0x1f && magic[1] == 0x8b;
        uint64_t min_bytes = 16u << 20;                       // below this a single thread is as fast
        if (const char* e = getenv("DSK_PARSE_MIN_BYTES")) min_bytes = (uint64_t)atoll(e);
        if (!gz && size >= min_bytes && nthreads > 1) {
            uint64_t n = 0;
            if (stream_parallel(chunkBytes, sink, size, nthreads, n, exact)) return n;
        }
        if (gz && (size >= (1u << 20) || getenv("DSK_PGZIP_CHUNK_BYTES"))) {      // (the switch: tests run the parallel gzip path on small files)
            uint64_t n = 0;
            if (nthreads > 1 && stream_bgzf(chunkBytes, sink, size, nthreads, n, exact)) return n;
            if (nthreads > 1 && !getenv("DSK_NO_PGZIP") && stream_pgz(chunkBytes, sink, size, nthreads, n, exact)) return n;
            return stream_gz_pipelined(chunkBytes, sink);
        }
        return stream_serial(chunkBytes, sink);
    }
    bool streamRaw(const RawSink& sink) override {
        const uint64_t size = file_size(path_);
        if (size < 2) return false;
        int fd = open(path_.c_str(), O_RDONLY);
        if (fd < 0) return false;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        const unsigned char* zb = (const unsigned char*)m;
        const bool gz = zb[0] == 0x1f && zb[1] == 0x8b;
        struct NotRecords {};
        bool first = true; int fmt = 0;
        // every piece of text goes on as it is; the first one decides the format (and whether this is text the device parser takes)
        auto hand_on = [&](const char* data, size_t len) {
            if (first) {
                const char* p = data;
                const char kind = record_kind(p, data + len);
                if (!kind) throw NotRecords{};
                fmt = kind == '>' ? 1 : 2;
                len -= (size_t)(p - data); data = p;
            }
            const size_t PIECE = (size_t)64 << 20;
            for (size_t off = 0; off < len || first; off += PIECE) {
                sink(data + off, std::min(PIECE, len - off), fmt, first);
                first = false;
            }
        };
        bool ok = false;
        try {
            if (!gz) { hand_on((const char*)m, size); ok = true; }
            else {
                uint32_t cs = 0;
                if (bgzf_block_at(zb, size, &cs)) { munmap(m, size); return false; }      // (BGZF: the member pool of stream() already inflates it in parallel)
                const unsigned nthreads = Bank::parseThreads();
                if (nthreads > 1 && !getenv("DSK_NO_PGZIP") && (size >= (1u << 20) || getenv("DSK_PGZIP_CHUNK_BYTES"))) {
                    size_t pgz_chunk = 0;
                    if (const char* e = getenv("DSK_PGZIP_CHUNK_BYTES")) pgz_chunk = (size_t)atoll(e);
                    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
                    ok = pgz_inflate(zb, size, pgz_threads(nthreads, hw), pgz_chunk, 0,
                                     [&](char* data, size_t len, bool) { hand_on(data, len); });
                }
                if (!ok) {                                  // one zlib stream
                    gzFile f = gzopen(path_.c_str(), "rb");
                    if (f) {
                        gzbuffer(f, 1 << 20);
                        std::vector<char> raw((size_t)8 << 20);
                        try {
                            for (;;) {
                                const int got = gzread(f, raw.data(), (unsigned)raw.size());
                                if (got < 0) throw Exception("read error in file '%s'", path_.c_str());
                                if (got == 0) break;
                                hand_on(raw.data(), (size_t)got);
                            }
                            if (gz_ended_badly(f)) throw Exception("truncated or corrupt gzip file '%s'", path_.c_str());
                        } catch (...) { gzclose(f); throw; }
                        gzclose(f);
                        ok = !first;
                    }
                }
            }
        } catch (const NotRecords&) {
            munmap(m, size);
            return false;
        } catch (const std::runtime_error& e) {
            munmap(m, size);
            throw Exception("%s: file '%s'", e.what(), path_.c_str());
        } catch (...) { munmap(m, size); throw; }
        munmap(m, size);
        return ok;
    }
private:
    uint64_t stream_serial(size_t chunkBytes, const Sink& sink) {
        gzFile f = gzopen(path_.c_str(), "rb");
        if (!f) throw Exception("unable to open file '%s'", path_.c_str());
        gzbuffer(f, 1 << 20);
        std::vector<char> raw(1 << 22);
        RecordParser ps(chunkBytes, &sink, nullptr);
        for (;;) {
            int got = gzread(f, raw.data(), (unsigned)raw.size());
            if (got < 0) { gzclose(f); throw Exception("read error in file '%s'", path_.c_str()); }
            if (got == 0) break;
            ps.feed(raw.data(), (size_t)got);
        }
        const bool bad_end = gz_ended_badly(f);
        gzclose(f);
        if (bad_end) throw Exception("truncated or corrupt gzip file '%s'", path_.c_str());
        ps.finish();
        return ps.nseq;
    }
    // Parse [p, end) of an uncompressed buffer with several threads on record-aligned ranges.
    // more: the text goes on behind `end` (a slab of a longer file): the LAST range must end between two records as well.
    // *exact &= every range that has a successor ended between two records (RecordParser::at_record_border)
    static uint64_t parse_parallel(const char* base, const char* p, const char* end, char kind, size_t chunkBytes, const Sink& sink, unsigned nthreads,
                                   bool more = false, bool* exact = nullptr) {
        const uint64_t size = (uint64_t)(end - p);
        nthreads = (unsigned)std::min<uint64_t>(nthreads, std::max<uint64_t>(1, size / std::max<uint64_t>(1, std::min<uint64_t>(8u << 20, size / 4 + 1))));
        std::vector<const char*> cut(nthreads + 1);
        cut[0] = p; cut[nthreads] = end;
        for (unsigned t = 1; t < nthreads; ++t) cut[t] = next_record_start(base, p + size * t / nthreads, end, kind);
        std::mutex mu; std::vector<uint64_t> counts(nthreads, 0); std::vector<std::thread> th;
        std::vector<char> border(nthreads, 1);
        // An exception of the sink (e.g. the engine running out of HBM) must not leave a worker thread: the first one
        // is kept, the other workers' chunks are dropped from then on, and it is thrown again after the join, so the
        // tool reports `EXCEPTION: <msg>` and exits with a failure code (src/main.cpp:42-46) instead of aborting.
        std::exception_ptr failure; std::atomic<bool> stop(false);
        const Sink guarded = [&](const char* d, size_t n) { if (!stop.load()) sink(d, n); };      // (called under `mu`)
        for (unsigned t = 0; t < nthreads; ++t)
            th.emplace_back([&, t]() {
                if (cut[t] >= cut[t + 1]) return;
                try {
                    RecordParser ps(chunkBytes, &guarded, &mu);
                    ps.feed(cut[t], (size_t)(cut[t + 1] - cut[t]));
                    if (t + 1 < nthreads || more) border[t] = ps.at_record_border(kind) ? 1 : 0;
                    ps.finish();
                    counts[t] = ps.nseq;
                } catch (...) {
                    stop = true;                       // (the throwing sink call held `mu`; it is released by now)
                    std::lock_guard<std::mutex> g(mu);
                    if (!failure) failure = std::current_exception();
                }
            });
        for (auto& x : th) x.join();
        if (failure) std::rethrow_exception(failure);
        uint64_t nseq = 0; for (auto c : counts) nseq += c;
        if (exact) for (char b : border) if (!b) *exact = false;
        return nseq;
    }
    // '>' / '@' if the buffer starts with a FASTA record / a 4-line FASTQ record (what the range cutter assumes), else 0
    static char record_kind(const char*& p, const char* end) {
        while (p < end && (*p == '\n' || *p == '\r' || *p == ' ')) ++p;
        const char kind = p < end ? *p : 0;
        if (kind != '>' && kind != '@') return 0;
        if (kind == '@') {
            const char* l1 = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* l2 = l1 ? (const char*)memchr(l1 + 1, '\n', (size_t)(end - (l1 + 1))) : nullptr;
            if (!l2 || l2 + 1 >= end || l2[1] != '+') return 0;
        }
        return kind;
    }
    bool stream_parallel(size_t chunkBytes, const Sink& sink, uint64_t size, unsigned nthreads, uint64_t& nseq, bool* exact) {
        int fd = open(path_.c_str(), O_RDONLY);
        if (fd < 0) return false;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        const char* base = (const char*)m; const char* end = base + size;
        const char* p = base;
        const char kind = record_kind(p, end);
        if (!kind) { munmap(m, size); return false; }
        nseq = parse_parallel(base, p, end, kind, chunkBytes, sink, nthreads, false, exact);
        munmap(m, size);
        return true;
    }

    // ---- gzip input
    // BGZF (bgzip, the blocked gzip variant of htslib): every member is an independent deflate stream of
    // <= 64 KB whose compressed size sits in the 'BC' extra field and whose inflated size in its trailer, so
    // the members are inflated by a thread pool straight to their final offsets of a slab, and the slab is
    // parsed like a memory-mapped plain file.  Returns false if the file is not BGZF.
    struct BgzfBlock { uint64_t off; uint32_t csize, isize; };
    static bool bgzf_block_at(const unsigned char* b, uint64_t left, uint32_t* csize) {
        if (left < 18 || b[0] != 0x1f || b[1] != 0x8b || b[2] != 8 || !(b[3] & 4)) return false;
        const uint32_t xlen = b[10] | (b[11] << 8);
        if (left < 12 + xlen) return false;
        for (uint32_t x = 0; x + 4 <= xlen;) {
            const unsigned char* f = b + 12 + x;
            const uint32_t slen = f[2] | (f[3] << 8);
            if (f[0] == 'B' && f[1] == 'C' && slen == 2 && x + 6 <= xlen) { *csize = (uint32_t)(f[4] | (f[5] << 8)) + 1; return *csize <= left; }
            x += 4 + slen;
        }
        return false;
    }
    bool stream_bgzf(size_t chunkBytes, const Sink& sink, uint64_t size, unsigned nthreads, uint64_t& nseq, bool* exact) {
        int fd = open(path_.c_str(), O_RDONLY);
        if (fd < 0) return false;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        const unsigned char* zb = (const unsigned char*)m;
        std::vector<BgzfBlock> blocks;
        for (uint64_t off = 0; off < size;) {
            uint32_t cs = 0;
            if (!bgzf_block_at(zb + off, size - off, &cs) || cs < 26) { munmap(m, size); return false; }
            const unsigned char* t = zb + off + cs - 4;
            blocks.push_back({off, cs, (uint32_t)(t[0] | (t[1] << 8) | (t[2] << 16) | ((uint32_t)t[3] << 24))});
            off += cs;
        }
        uint64_t SLAB = 1ull << 30;                             // inflated bytes handled at once
        if (const char* e = getenv("DSK_BGZF_SLAB_BYTES")) SLAB = std::max<uint64_t>(1, (uint64_t)atoll(e));
        std::vector<char> slab; std::string carry;
        nseq = 0; char kind = 0; bool first = true; bool ok = true;
        for (size_t b0 = 0; b0 < blocks.size() && ok;) {
            size_t b1 = b0; uint64_t bytes = 0;
            while (b1 < blocks.size() && (bytes == 0 || bytes + blocks[b1].isize <= SLAB)) bytes += blocks[b1++].isize;
            slab.resize(carry.size() + bytes);
            std::memcpy(slab.data(), carry.data(), carry.size());
            std::vector<uint64_t> dst(b1 - b0 + 1); dst[0] = carry.size();
            for (size_t i = b0; i < b1; ++i) dst[i - b0 + 1] = dst[i - b0] + blocks[i].isize;
            std::atomic<size_t> next(b0); std::atomic<bool> bad(false);
            std::vector<std::thread> th;
            for (unsigned t = 0; t < std::max(1u, nthreads); ++t)
                th.emplace_back([&]() {
                    z_stream zs; std::memset(&zs, 0, sizeof(zs));
                    if (inflateInit2(&zs, -15) != Z_OK) { bad = true; return; }
                    for (;;) {
                        const size_t i = next.fetch_add(1);
                        if (i >= b1) break;
                        const BgzfBlock& bk = blocks[i];
                        const unsigned char* src = zb + bk.off;
                        const uint32_t xlen = src[10] | (src[11] << 8);
                        inflateReset(&zs);
                        zs.next_in = const_cast<unsigned char*>(src + 12 + xlen); zs.avail_in = bk.csize - 12 - xlen - 8;
                        zs.next_out = (unsigned char*)slab.data() + dst[i - b0]; zs.avail_out = bk.isize;
                        const int rc = inflate(&zs, Z_FINISH);
                        if ((rc != Z_STREAM_END && !(rc == Z_OK && bk.isize == 0)) || zs.avail_out != 0) { bad = true; break; }
                    }
                    inflateEnd(&zs);
                });
            for (auto& x : th) x.join();
            if (bad) { ok = false; break; }
            const char* base = slab.data(); const char* end = base + slab.size(); const char* p = base;
            if (first) { kind = record_kind(p, end); first = false; if (!kind) { ok = false; break; } }
            const char* stop = end;
            if (b1 < blocks.size()) {                           // keep the (possibly cut) last record for the next slab
                uint64_t back = 1u << 16;
                const char* q = end;
                for (;;) {
                    const char* from = (uint64_t)(end - p) > back ? end - back : p;
                    q = next_record_start(base, from, end, kind);
                    if (q < end || from == p) break;
                    back *= 16;
                }
                if (q < end) for (;;) { const char* r = next_record_start(base, q + 1, end, kind); if (r >= end) break; q = r; }
                stop = q < end ? q : p;
            }
            if (stop > p) nseq += parse_parallel(base, p, stop, kind, chunkBytes, sink, nthreads, b1 < blocks.size(), exact);
            carry.assign(stop, (size_t)(end - stop));
            b0 = b1;
        }
        munmap(m, size);
        if (!ok) throw Exception("corrupt BGZF file '%s'", path_.c_str());
        return true;
    }
    // Ordinary gzip, ONE member (what `gzip reads.fastq` writes): inflated on the thread pool by the two-pass scheme of pgzip.hpp --
    // block starts searched inside the stream, chunks inflated with a symbolic window, windows resolved in order --, slab by slab;
    // every slab is parsed like a memory-mapped plain file, the cut last record carried into the next one.  Returns false when the
    // file is not such a gzip (several members, no dynamic blocks, a first slab that does not pass): nothing was consumed then.
    bool stream_pgz(size_t chunkBytes, const Sink& sink, uint64_t size, unsigned nthreads, uint64_t& nseq, bool* exact) {
        int fd = open(path_.c_str(), O_RDONLY);
        if (fd < 0) return false;
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return false;
        std::vector<char> slab; std::string carry;
        nseq = 0; char kind = 0; bool first = true;
        struct NotRecords {};                                  // the first slab does not start like a FASTA / FASTQ file: the lenient serial parser takes the file
        size_t pgz_chunk = 0;
        if (const char* e = getenv("DSK_PGZIP_CHUNK_BYTES")) pgz_chunk = (size_t)atoll(e);
        bool ok = false;
        try {
            // (the inflate is pure CPU work on independent chunks: it takes up to 64 threads where the parser is capped at 32)
            const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
            const size_t HEAD = 4u << 20;                        // room in front of every slab for the previous slab's cut-off record
            ok = pgz_inflate((const uint8_t*)m, size, pgz_threads(nthreads, hw), pgz_chunk, HEAD, [&](char* data, size_t len, bool last) {
                const char* base; const char* end;
                if (carry.size() <= HEAD) { std::memcpy(data - carry.size(), carry.data(), carry.size()); base = data - carry.size(); end = data + len; }
                else {                                           // (a record longer than the headroom: the slab is copied behind it)
                    slab.resize(carry.size() + len);
                    std::memcpy(slab.data(), carry.data(), carry.size());
                    std::memcpy(slab.data() + carry.size(), data, len);
                    base = slab.data(); end = base + slab.size();
                }
                const char* p = base;
                if (first) { kind = record_kind(p, end); first = false; if (!kind) throw NotRecords{}; }
                const char* stop = end;
                if (!last) {                                     // keep the (possibly cut) last record for the next slab
                    uint64_t back = 1u << 16;
                    const char* q = end;
                    for (;;) {
                        const char* from = (uint64_t)(end - p) > back ? end - back : p;
                        q = next_record_start(base, from, end, kind);
                        if (q < end || from == p) break;
                        back *= 16;
                    }
                    if (q < end) for (;;) { const char* r = next_record_start(base, q + 1, end, kind); if (r >= end) break; q = r; }
                    stop = q < end ? q : p;
                }
                if (stop > p) nseq += parse_parallel(base, p, stop, kind, chunkBytes, sink, nthreads, !last, exact);
                carry.assign(stop, (size_t)(end - stop));
            });
        } catch (const NotRecords&) {
            munmap(m, size);
            return false;
        } catch (const std::runtime_error& e) {
            munmap(m, size);
            throw Exception("%s: file '%s'", e.what(), path_.c_str());
        }
        munmap(m, size);
        return ok;
    }
    // Ordinary gzip otherwise: one deflate stream inflated by zlib, but inflating and parsing overlap -- a producer
    // thread inflates 4 MB buffers into a small queue while the caller parses and hands chunks to the sink.
    uint64_t stream_gz_pipelined(size_t chunkBytes, const Sink& sink) {
        gzFile f = gzopen(path_.c_str(), "rb");
        if (!f) throw Exception("unable to open file '%s'", path_.c_str());
        gzbuffer(f, 1 << 20);
        const size_t NB = 4, BUF = 4u << 20;
        std::vector<std::vector<char>> bufs(NB, std::vector<char>(BUF));
        std::vector<int> len(NB, 0);
        std::mutex mu; std::condition_variable cv;
        size_t produced = 0, consumed = 0; bool done = false, failed = false, cancel = false;
        std::thread producer([&]() {
            for (;;) {
                { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return produced - consumed < NB || cancel; }); if (cancel) return; }
                const size_t slot = produced % NB;
                const int got = gzread(f, bufs[slot].data(), (unsigned)BUF);
                std::lock_guard<std::mutex> g(mu);
                if (got <= 0) { failed = got < 0 || gz_ended_badly(f); done = true; cv.notify_all(); return; }
                len[slot] = got; ++produced; cv.notify_all();
            }
        });
        RecordParser ps(chunkBytes, &sink, nullptr);
        try {
            for (;;) {
                { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return consumed < produced || done; }); if (consumed == produced && done) break; }
                const size_t slot = consumed % NB;
                ps.feed(bufs[slot].data(), (size_t)len[slot]);
                { std::lock_guard<std::mutex> g(mu); ++consumed; } cv.notify_all();
            }
            producer.join();
            gzclose(f);
            if (failed) throw Exception("read error, truncated or corrupt gzip file '%s'", path_.c_str());
            ps.finish();
        } catch (...) {                                // the sink threw: stop the inflate thread before this frame goes away
            if (producer.joinable()) { { std::lock_guard<std::mutex> g(mu); cancel = true; } cv.notify_all(); producer.join(); gzclose(f); }
            throw;
        }
        return ps.nseq;
    }
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
    uint64_t stream(size_t chunkBytes, const Sink& sink, bool* exact = nullptr) override {
        if (exact) *exact = true;
        uint64_t n = 0;
        for (auto* b : banks_) { bool e = true; n += b->stream(chunkBytes, sink, exact ? &e : nullptr); if (exact && !e) *exact = false; }
        return n;
    }
    uint64_t streamSerial(size_t chunkBytes, const Sink& sink) override { uint64_t n = 0; for (auto* b : banks_) n += b->streamSerial(chunkBytes, sink); return n; }
    std::vector<IBank*> banks() override { return banks_; }
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

namespace { unsigned g_parse_threads = 0; }
void Bank::setParseThreads(unsigned n) { g_parse_threads = n; }
unsigned Bank::parseThreads() {
    unsigned hw = std::thread::hardware_concurrency(); if (hw == 0) hw = 1;
    unsigned n = g_parse_threads ? g_parse_threads : hw;
    return std::min(n, 32u);
}

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
