// pgzip.cpp -- see pgzip.hpp.  The deflate decoder below is written from RFC 1951; zlib is used for crc32 / crc32_combine and for the
// serial continuation (inflatePrime + inflateSetDictionary) when the speculative scheme gives up on a slab.
#include "pgzip.hpp"

#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <new>

#include <sys/mman.h>

namespace dsk {
namespace {

typedef uint16_t sym_t;                       // < 256: a byte; 0x8000 | j: byte j of the 32 KB window before the chunk
constexpr uint32_t WIN = 32768;

struct Bits {                                 // LSB-first bit reader over [base, base + n)
    const uint8_t* base; size_t n; size_t byte; uint64_t buf; int cnt;
    void seek(uint64_t bitpos) { byte = (size_t)(bitpos >> 3); buf = 0; cnt = 0; refill(); const int sk = (int)(bitpos & 7); buf >>= sk; cnt -= sk; }
    inline void refill() {
        while (cnt <= 56) { buf |= (uint64_t)(byte < n ? base[byte] : 0) << cnt; ++byte; cnt += 8; }      // (past the end: zeros; `over()` tells)
    }
    inline uint32_t peek(int k) const { return (uint32_t)(buf & ((1ull << k) - 1)); }
    inline void drop(int k) { buf >>= k; cnt -= k; }
    inline uint32_t get(int k) { if (cnt < k) refill(); const uint32_t v = peek(k); drop(k); return v; }
    uint64_t pos() const { return (uint64_t)byte * 8 - (uint64_t)cnt; }
    bool over() const { return pos() > (uint64_t)n * 8; }
};

// canonical Huffman decoding table: PB primary bits, longer codes through second-level tables
// entry: low 4 bits = bits to drop (0 = invalid), bit 4 = link to a second-level table, high 16 bits = symbol / table offset, bits 8..11 = second-level width
struct Huff {
    std::vector<uint32_t> t; int pb = 0;
    // lens[0..n): code lengths (0 = unused).  Returns false for an over-subscribed or (unless allow_incomplete) incomplete code.
    bool build(const uint8_t* lens, int n, int primary_bits, bool allow_incomplete) {
        int count[16] = {0}; for (int i = 0; i < n; ++i) ++count[lens[i]];
        count[0] = 0;
        int maxlen = 0; for (int l = 1; l <= 15; ++l) if (count[l]) maxlen = l;
        if (!maxlen) return false;
        long left = 1;
        for (int l = 1; l <= 15; ++l) { left <<= 1; left -= count[l]; if (left < 0) return false; }
        if (left > 0 && !(allow_incomplete && maxlen == 1 && count[1] == 1)) return false;
        uint32_t next[16]; uint32_t code = 0;
        for (int l = 1; l <= 15; ++l) { code = (code + (uint32_t)count[l - 1]) << 1; next[l] = code; }
        pb = std::min(primary_bits, maxlen);
        t.assign((size_t)1 << pb, 0u);
        // second-level tables: one per distinct pb-bit prefix of a long code, each 2^(maxlen - pb) entries (simple, at most a few KB)
        const int sb = maxlen - pb;
        std::vector<int> link((size_t)1 << pb, -1);
        for (int i = 0; i < n; ++i) {
            const int l = lens[i]; if (!l) continue;
            const uint32_t c = next[l]++;
            uint32_t r = 0; for (int b = 0; b < l; ++b) r |= ((c >> (l - 1 - b)) & 1u) << b;      // bit-reversed: the stream is LSB first
            if (l <= pb) {
                for (uint32_t x = r; x < ((uint32_t)1 << pb); x += (uint32_t)1 << l) t[x] = ((uint32_t)i << 16) | (uint32_t)l;
            } else {
                const uint32_t pre = r & (((uint32_t)1 << pb) - 1);
                if (link[pre] < 0) { link[pre] = (int)t.size(); t.resize(t.size() + ((size_t)1 << sb), 0u); t[pre] = ((uint32_t)link[pre] << 16) | ((uint32_t)sb << 8) | 0x10u | (uint32_t)pb; }
                const uint32_t hi = r >> pb; const int hl = l - pb;
                for (uint32_t x = hi; x < ((uint32_t)1 << sb); x += (uint32_t)1 << hl) t[(size_t)link[pre] + x] = ((uint32_t)i << 16) | (uint32_t)hl;
            }
        }
        return true;
    }
    // -> symbol, or -1 (invalid code)
    inline int decode(Bits& br) const {
        if (br.cnt < 15) br.refill();
        uint32_t e = t[br.peek(pb)];
        if (e & 0x10u) { br.drop(pb); e = t[(e >> 16) + br.peek((int)((e >> 8) & 15u))]; }
        const int l = (int)(e & 15u);
        if (!l) return -1;
        br.drop(l);
        return (int)(e >> 16);
    }
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t CL_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct BlockCodes { Huff lit, dist; bool has_dist = false; };

// dynamic-block header at the reader's position (behind BFINAL / BTYPE) -> codes.  strict: what the SEARCH accepts (complete codes, an end-of-block code)
bool read_dynamic(Bits& br, BlockCodes& bc, bool strict) {
    const int hlit = (int)br.get(5) + 257, hdist = (int)br.get(5) + 1, hclen = (int)br.get(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t cl[19] = {0};
    for (int i = 0; i < hclen; ++i) cl[CL_ORDER[i]] = (uint8_t)br.get(3);
    Huff clh;
    if (!clh.build(cl, 19, 7, !strict)) return false;
    uint8_t lens[286 + 30]; int i = 0; const int tot = hlit + hdist;
    while (i < tot) {
        const int s = clh.decode(br);
        if (s < 0) return false;
        if (s < 16) lens[i++] = (uint8_t)s;
        else {
            int rep; uint8_t v = 0;
            if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + (int)br.get(2); }
            else if (s == 17) rep = 3 + (int)br.get(3);
            else rep = 11 + (int)br.get(7);
            if (i + rep > tot) return false;
            while (rep--) lens[i++] = v;
        }
    }
    if (br.over() || lens[256] == 0) return false;
    if (!bc.lit.build(lens, hlit, 11, false)) return false;
    int nd = 0; for (int d = 0; d < hdist; ++d) nd += lens[hlit + d] != 0;
    bc.has_dist = nd > 0;
    if (bc.has_dist && !bc.dist.build(lens + hlit, hdist, 9, true)) return false;
    return true;
}

void fixed_codes(BlockCodes& bc) {
    uint8_t l[288]; for (int i = 0; i < 144; ++i) l[i] = 8; for (int i = 144; i < 256; ++i) l[i] = 9; for (int i = 256; i < 280; ++i) l[i] = 7; for (int i = 280; i < 288; ++i) l[i] = 8;
    bc.lit.build(l, 288, 11, false);
    uint8_t d[32]; for (int i = 0; i < 32; ++i) d[i] = 5;      // RFC 1951 3.2.6: distance codes 0-31 in 5 bits (30 and 31 never occur: the decoder rejects them)
    bc.dist.build(d, 32, 9, false); bc.has_dist = true;          // (as 30 codes the set is incomplete and build() refused it: a fixed block then decoded its distances with the previous block's table)
}

inline bool texty(int c) { return c == '\n' || c == '\r' || c == '\t' || (c >= 32 && c < 127); }

// One chunk's output: WIN marker slots, then the symbols
// (raw storage, never zero-filled: a chunk's symbols are twice its inflated bytes, and value-initialising them -- plus the page faults of a
//  fresh std::vector per chunk -- cost more than the inflate itself on a 64-thread host)
struct RawBuf {
    void* p = nullptr; size_t cap = 0;
    ~RawBuf() { free(p); }
    RawBuf() = default; RawBuf(const RawBuf&) = delete; RawBuf& operator=(const RawBuf&) = delete;
    void reserve(size_t bytes, size_t keep) {              // grow to >= bytes, keeping the first `keep` bytes
        if (bytes <= cap) return;
        void* q = nullptr;
        const size_t want = (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
        if (posix_memalign(&q, 2u << 20, want) != 0) throw std::bad_alloc();
        madvise(q, want, MADV_HUGEPAGE);
        if (keep && p) std::memcpy(q, p, keep);
        free(p); p = q; cap = want;
    }
    void release() { free(p); p = nullptr; cap = 0; }
};
struct SymBuf {
    RawBuf b; sym_t* v = nullptr; size_t n = 0, cap = 0;    // symbols written behind the WIN prefix; cap = symbols of room (prefix included)
    void init(size_t reserve) { b.reserve((WIN + reserve) * sizeof(sym_t), 0); v = (sym_t*)b.p; cap = b.cap / sizeof(sym_t); for (uint32_t j = 0; j < WIN; ++j) v[j] = (sym_t)(0x8000u | j); n = 0; }
    inline void need(size_t more) {
        if (WIN + n + more > cap) { b.reserve(std::max(cap * 3 / 2, WIN + n + more + (1u << 20)) * sizeof(sym_t), (WIN + n) * sizeof(sym_t)); v = (sym_t*)b.p; cap = b.cap / sizeof(sym_t); }
    }
};

enum Stop { ST_OK = 0, ST_BAD = 1, ST_FINAL = 2 };
// Inflate whole blocks from the reader's position until a block ends at or behind `until_bit` (ST_OK; *end_bit = where), the final block
// has been decoded (ST_FINAL) or the stream is invalid (ST_BAD).  probe: the search's validation -- literals must be text, stop
// after `probe_out` symbols at the next block end.
Stop inflate_blocks(Bits& br, SymBuf& out, uint64_t until_bit, uint64_t* end_bit, bool probe, size_t probe_out) {
    BlockCodes bc;
    for (;;) {
        const uint32_t bfinal = br.get(1), btype = br.get(2);
        if (btype == 3) return ST_BAD;
        if (btype == 0) {
            br.drop(br.cnt & 7);
            if (br.cnt < 32) br.refill();
            const uint32_t len = br.get(16), nlen = br.get(16);
            if ((len ^ nlen) != 0xFFFFu || br.over()) return ST_BAD;
            if (probe) return ST_BAD;                       // (the search does not start on stored blocks; inside a validation run they are too rare to bother)
            out.need(len);
            for (uint32_t i = 0; i < len; ++i) out.v[WIN + out.n++] = (sym_t)br.get(8);
            if (br.over()) return ST_BAD;
        } else {
            if (btype == 1) fixed_codes(bc);
            else if (!read_dynamic(br, bc, probe)) return ST_BAD;
            for (;;) {
                out.need(258 + 8);
                sym_t* o = out.v + WIN + out.n;
                const int s = bc.lit.decode(br);
                if (s < 0) return ST_BAD;
                if (s < 256) { if (probe && !texty(s)) return ST_BAD; *o = (sym_t)s; ++out.n; continue; }
                if (s == 256) break;
                if (s > 285) return ST_BAD;
                if (br.cnt < 32) br.refill();
                const uint32_t len = LEN_BASE[s - 257] + br.get(LEN_EXTRA[s - 257]);
                if (!bc.has_dist) return ST_BAD;
                const int ds = bc.dist.decode(br);
                if (ds < 0 || ds > 29) return ST_BAD;
                if (br.cnt < 16) br.refill();
                const uint32_t dist = DIST_BASE[ds] + br.get(DIST_EXTRA[ds]);
                if (dist > WIN + out.n) return ST_BAD;       // (before the window: only possible at the very start of a member, where it is an error)
                const sym_t* from = o - dist;
                for (uint32_t i = 0; i < len; ++i) o[i] = from[i];
                out.n += len;
                if (br.over()) return ST_BAD;
            }
        }
        if (br.over()) return ST_BAD;
        if (bfinal) { *end_bit = br.pos(); return ST_FINAL; }
        const uint64_t p = br.pos();
        if (p >= until_bit || (probe && out.n >= probe_out)) { *end_bit = p; return ST_OK; }
    }
}

// first bit position >= from_bit (and < limit_bit) where a non-final dynamic block starts that survives validation; ~0 = none
uint64_t find_block(const uint8_t* data, size_t n, uint64_t from_bit, uint64_t limit_bit) {
    Bits br{data, n, 0, 0, 0};
    SymBuf probe; probe.init(1u << 18);
    for (uint64_t p = from_bit; p < limit_bit; ++p) {
        br.seek(p);
        if ((br.peek(3) & 7u) != 4u) continue;              // BFINAL = 0, BTYPE = 10 (LSB first: bits 0, then 01 -> value 0b100)
        // cheap header screen before building tables: HLIT <= 29, HDIST <= 29
        const uint32_t h = br.peek(17);
        if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) continue;
        probe.n = 0;
        uint64_t e = 0;
        const Stop st = inflate_blocks(br, probe, ~0ull, &e, true, 48u << 10);
        if (st == ST_OK && probe.n >= (8u << 10)) return p;  // >= 8 K symbols of text through >= 1 complete block with valid successors' headers
    }
    return ~0ull;
}

struct Chunk {
    uint64_t start_bit = 0, end_bit = 0; bool found = false, ok = false, final_block = false;
    SymBuf out; std::vector<uint8_t> win_after;             // resolved window behind this chunk
    uLong crc = 0; size_t out_off = 0;
};

void run_pool(unsigned nthreads, size_t njobs, const std::function<void(size_t)>& job) {
    std::atomic<size_t> next(0);
    std::vector<std::thread> th;
    const unsigned nt = (unsigned)std::min<size_t>(std::max(1u, nthreads), std::max<size_t>(1, njobs));
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([&]() { for (;;) { const size_t i = next.fetch_add(1); if (i >= njobs) break; job(i); } });
    for (auto& x : th) x.join();
}

}  // namespace

// gzip member header at file[off..] (RFC 1952) -> offset of its deflate stream, 0 when there is none
static size_t member_header(const uint8_t* file, size_t n, size_t off) {
    if (off + 18 > n || file[off] != 0x1f || file[off + 1] != 0x8b || file[off + 2] != 8) return 0;
    const uint8_t flg = file[off + 3];
    size_t h = off + 10;
    if (flg & 4) { if (h + 2 > n) return 0; h += 2 + (size_t)(file[h] | (file[h + 1] << 8)); }
    if (flg & 8) { while (h < n && file[h]) ++h; ++h; }
    if (flg & 16) { while (h < n && file[h]) ++h; ++h; }
    if (flg & 2) h += 2;
    return h + 8 <= n ? h : 0;
}

bool pgz_inflate(const uint8_t* file, size_t n, unsigned nthreads, size_t chunk_bytes, size_t headroom,
                 const std::function<void(char*, size_t, bool)>& consume) {
    const size_t h0 = member_header(file, n, 0);
    if (!h0) return false;
    // chunk = the unit of parallel work.  2 MB of compressed text inflate to ~12 MB at ~160 MB/s per thread: a file of 80 MB cut that
    // way keeps 39 threads busy for 80 ms each.  Files that do not fill the pool with 2 MB chunks are cut finer (not below 512 KB: a
    // chunk's block start is searched in its first 256 KB).
    if (!chunk_bytes) {
        const size_t per_thread = ((n + nthreads - 1) / nthreads + 0xFFFF) & ~(size_t)0xFFFF;
        chunk_bytes = std::min<size_t>(2u << 20, std::max<size_t>(512u << 10, per_thread));
    }
    if (nthreads < 2 || n < h0 + 2 * chunk_bytes) return false;
    const size_t per_slab = std::min<size_t>(std::max<size_t>(2, (size_t)nthreads * 2), 128);          // chunks per slab
    const uint8_t* data = file; const size_t dn = n;         // bit positions are positions in the file
    const bool trace = getenv("DSK_PGZIP_TRACE") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

    uint64_t at_bit = (uint64_t)h0 * 8;                      // exact position reached so far (inside the current member's deflate stream)
    std::vector<uint8_t> window(WIN, 0);                    // the 32 KB before it (resolved)
    uLong crc = crc32(0L, Z_NULL, 0); uint64_t total_out = 0;      // of the current member
    RawBuf bytes;                                           // [headroom | the slab's inflated bytes]
    bool first_slab = true, file_done = false;

    // the member's deflate stream ended at byte `end_byte`: its trailer must match; -> true when another member follows (at_bit set)
    auto next_member = [&](size_t end_byte) -> bool {
        if (end_byte + 8 > n) throw std::runtime_error("truncated gzip file");
        const uint32_t want_crc = (uint32_t)file[end_byte] | ((uint32_t)file[end_byte + 1] << 8) | ((uint32_t)file[end_byte + 2] << 16) | ((uint32_t)file[end_byte + 3] << 24);
        const uint32_t want_isize = (uint32_t)file[end_byte + 4] | ((uint32_t)file[end_byte + 5] << 8) | ((uint32_t)file[end_byte + 6] << 16) | ((uint32_t)file[end_byte + 7] << 24);
        if ((uint32_t)crc != want_crc || (uint32_t)total_out != want_isize) throw std::runtime_error("gzip CRC / size mismatch (corrupt file)");
        const size_t h = member_header(file, n, end_byte + 8);      // (anything that is not a gzip member behind the last one is ignored, as zlib does)
        if (!h) return false;
        at_bit = (uint64_t)h * 8; std::fill(window.begin(), window.end(), 0); crc = crc32(0L, Z_NULL, 0); total_out = 0;
        return true;
    };

    while (!file_done) {
        // ---- chunks of this slab: [at_bit, ...) cut at multiples of chunk_bytes
        const size_t b0 = (size_t)(at_bit >> 3);
        const size_t nch = std::min(per_slab, std::max<size_t>(1, (dn - b0 + chunk_bytes - 1) / chunk_bytes));
        std::vector<Chunk> ch(nch);
        std::vector<uint64_t> cut(nch + 1);
        for (size_t i = 0; i <= nch; ++i) cut[i] = (uint64_t)std::min(dn, b0 + i * chunk_bytes) * 8;
        ch[0].start_bit = at_bit; ch[0].found = true;
        const double tt0 = now();
        // 1. search (parallel): the first block start inside every later chunk
        run_pool(nthreads, nch - 1, [&](size_t j) {
            Chunk& c = ch[j + 1];
            // (zlib closes a block every 16 K symbols -- some tens of KB of text --; a chunk without a block start in its first 256 KB is
            //  stored or otherwise unusual data: it is left to its predecessor instead of being searched bit by bit to its end)
            const uint64_t p = find_block(data, dn, cut[j + 1], std::min<uint64_t>(cut[j + 2], cut[j + 1] + (256u << 13)));
            c.found = p != ~0ull; c.start_bit = p;
        });
        const double tt1 = now();
        // chunks without a block start are merged into their predecessor (a block longer than a chunk: stored data, long runs)
        std::vector<size_t> live; for (size_t i = 0; i < nch; ++i) if (ch[i].found) live.push_back(i);
        // 2. inflate (parallel): from the chunk's start to exactly the next live chunk's start (the slab's last one: to the first block end
        // behind the slab); a chunk that meets the member's final block ends the slab there
        run_pool(nthreads, live.size(), [&](size_t li) {
            Chunk& c = ch[live[li]];
            const bool last = li + 1 == live.size();
            const uint64_t until = last ? cut[nch] : ch[live[li + 1]].start_bit;
            Bits br{data, dn, 0, 0, 0}; br.seek(c.start_bit);
            c.out.init((size_t)((until > c.start_bit ? until - c.start_bit : 0) / 8) * 4 + (1u << 20));
            uint64_t e = 0;
            const Stop st = inflate_blocks(br, c.out, until, &e, false, 0);
            c.end_bit = e; c.final_block = st == ST_FINAL;
            c.ok = st == ST_FINAL || (st == ST_OK && (last || e == until));
        });
        const double tt2 = now();
        if (trace) for (size_t li = 0; li < live.size(); ++li) { const Chunk& c = ch[live[li]]; fprintf(stderr, "[pgzip]     chunk %zu: start %llu end %llu out %zu ok %d final %d\n", live[li], (unsigned long long)c.start_bit, (unsigned long long)c.end_bit, c.out.n, (int)c.ok, (int)c.final_block); }
        size_t good = 0;
        while (good < live.size() && ch[live[good]].ok) { ++good; if (ch[live[good - 1]].final_block) break; }
        const bool ended = good > 0 && ch[live[good - 1]].final_block;
        if (trace) fprintf(stderr, "[pgzip] slab at bit %llu: %zu chunks, %zu with a block start, %zu good%s; search %.1f ms, inflate %.1f ms\n", (unsigned long long)at_bit, nch, live.size(), good,
                           ended ? " (the member ends)" : "", tt1 - tt0, tt2 - tt1);
        if (good == 0 || (first_slab && good < live.size() && !ended)) {
            if (first_slab) return false;                   // nothing consumed yet: the caller's zlib path takes the whole file
            // ---- the rest of this member with zlib, from the exact state (position + window): the scheme gave up, the file may still be fine
            z_stream zs; std::memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib: inflateInit2 failed");
            inflateSetDictionary(&zs, window.data(), WIN);
            const int pre = (int)(at_bit & 7);
            size_t ib = (size_t)(at_bit >> 3);
            if (pre) { inflatePrime(&zs, 8 - pre, data[ib] >> pre); ++ib; }
            zs.next_in = const_cast<Bytef*>(data + ib); zs.avail_in = (uInt)std::min<size_t>(dn - ib, 1u << 30);
            const size_t zbuf = 64u << 20;
            bytes.reserve(headroom + zbuf, 0);
            char* const zout = (char*)bytes.p + headroom;
            size_t end_byte = 0; bool more = false;
            for (;;) {
                zs.next_out = (Bytef*)zout; zs.avail_out = (uInt)zbuf;
                if (zs.avail_in == 0) { const size_t used = (size_t)(zs.next_in - data); zs.avail_in = (uInt)std::min<size_t>(dn - used, 1u << 30); }
                const int rc = inflate(&zs, Z_NO_FLUSH);
                const size_t got = zbuf - zs.avail_out;
                if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); throw std::runtime_error("corrupt gzip stream"); }
                crc = crc32(crc, (const Bytef*)zout, (uInt)got); total_out += got;
                const bool end = rc == Z_STREAM_END;
                if (end) { end_byte = (size_t)(zs.next_in - data); inflateEnd(&zs); more = next_member(end_byte); }
                if (got || (end && !more)) consume(zout, got, end && !more);
                if (end) break;
                if (got == 0 && zs.avail_in == 0 && (size_t)(zs.next_in - data) >= dn) { inflateEnd(&zs); throw std::runtime_error("truncated gzip stream"); }
            }
            if (!more) file_done = true;
            continue;
        }
        // (a chunk that failed and everything behind it is left to the next slab, which starts where the last good chunk ended)
        // 3. windows, chunk after chunk (serial, 32 KB each); output offsets
        size_t out_total = 0;
        for (size_t li = 0; li < good; ++li) {
            Chunk& c = ch[live[li]];
            const std::vector<uint8_t>& wprev = li == 0 ? window : ch[live[li - 1]].win_after;
            c.out_off = out_total; out_total += c.out.n;
            c.win_after.resize(WIN);
            const size_t take = std::min<size_t>(WIN, c.out.n);
            if (take < WIN) std::memcpy(c.win_after.data(), wprev.data() + take, WIN - take);
            const sym_t* sy = c.out.v + WIN + c.out.n - take;
            for (size_t i = 0; i < take; ++i) c.win_after[WIN - take + i] = sy[i] < 256 ? (uint8_t)sy[i] : wprev[sy[i] & 0x7FFFu];
        }
        // 4. symbols -> bytes + CRC per chunk (parallel)
        bytes.reserve(headroom + out_total + 64, 0);
        char* const out_base = (char*)bytes.p + headroom;
        run_pool(nthreads, good, [&](size_t li) {
            Chunk& c = ch[live[li]];
            const std::vector<uint8_t>& wprev = li == 0 ? window : ch[live[li - 1]].win_after;
            const sym_t* sy = c.out.v + WIN; char* o = out_base + c.out_off;
            size_t i = 0;
            const size_t n8 = c.out.n & ~(size_t)7;
            for (; i < n8; i += 8) {                          // eight symbols at a time when none of them is a marker (nearly always, past a chunk's first KBs)
                uint64_t a, b2; std::memcpy(&a, sy + i, 8); std::memcpy(&b2, sy + i + 4, 8);
                if (((a | b2) & 0xFF00FF00FF00FF00ull) == 0) {
                    a = (a | (a >> 8)) & 0x0000FFFF0000FFFFull; a = (a | (a >> 16)) & 0xFFFFFFFFull;
                    b2 = (b2 | (b2 >> 8)) & 0x0000FFFF0000FFFFull; b2 = (b2 | (b2 >> 16)) & 0xFFFFFFFFull;
                    const uint64_t w = a | (b2 << 32);
                    std::memcpy(o + i, &w, 8);
                } else for (size_t x = i; x < i + 8; ++x) o[x] = (char)(sy[x] < 256 ? (uint8_t)sy[x] : wprev[sy[x] & 0x7FFFu]);
            }
            for (; i < c.out.n; ++i) o[i] = (char)(sy[i] < 256 ? (uint8_t)sy[i] : wprev[sy[i] & 0x7FFFu]);
            c.crc = crc32(crc32(0L, Z_NULL, 0), (const Bytef*)o, (uInt)c.out.n);
            // (the symbol buffers are given back by the slab's destructor, after the bytes have been handed on: 25 MB munmaps from
            //  64 threads at once serialise in the kernel)
        });
        for (size_t li = 0; li < good; ++li) { crc = crc32_combine(crc, ch[live[li]].crc, (z_off_t)ch[live[li]].out.n); total_out += ch[live[li]].out.n; }
        const Chunk& lastc = ch[live[good - 1]];
        at_bit = lastc.end_bit; window = lastc.win_after;
        if (trace) fprintf(stderr, "[pgzip]   resolve + bytes + crc %.1f ms, %zu bytes out\n", now() - tt2, out_total);
        if (ended) file_done = !next_member((size_t)((lastc.end_bit + 7) >> 3));      // (checks this member's CRC-32 and size before its last bytes are handed on)
        consume(out_base, out_total, file_done);
        first_slab = false;
    }
    return true;
}

}  // namespace dsk
