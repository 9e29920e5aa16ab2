// rawparse.h -- FASTA / FASTQ text -> the engine's read stream, on the device (gfx950, wave64).
//
// BankFasta ingest (src/DSK.cpp:51 Bank::open; README.md:52-61) is one of the subsystems BASELINE.json's north_star lists as
// replaced.  Until round 5 the host parser (host/bank.cpp) stripped headers and quality lines and pushed clean bases; with
// dskgpu_push_raw the file's bytes go to HBM as they are, chunk by chunk, and three small kernels per chunk leave exactly what the host
// parser would have left: the bases of every record, one '\n' between records (the read stream's separator: k-mers never span
// records, any byte outside ACGTacgt ends a window -- test/readN.fasta).
//   FASTQ  four lines per record: a byte is kept iff it lies on a line with index % 4 == 1 (the sequence line, its '\n' included);
//          which line a byte lies on is a prefix count of '\n'.  Line 0 of a record must start with '@', line 2 with '+': anything
//          else (a FASTQ whose sequences are wrapped over several lines) raises RawState::bad and the host falls back to its parser.
//          (A quality line that starts with '@' or '>' is no problem here: lines are classified by their NUMBER, not their first byte.)
//          A host parser reads as many quality characters as the record has bases (host/bank.cpp RecordParser, kseq): a file whose
//          quality line is not as long as its sequence line would be read differently there -- it is given back as well (rp_file_ok).
//   FASTA  a line that starts with '>' is a header: dropped but for its '\n', which separates the records (a line that starts with
//          '@' or '+' -- FASTQ records in the same file -- raises RawState::bad); the bytes of all
//          other lines are kept except '\n', '\r', ' ' and '\t', so that a sequence wrapped over lines is one run of bases (test/longread.fasta).
// The state between chunks -- lines so far, "inside a header line", "the next byte starts a line", bytes written -- lives on the device
// (RawState): no round trip per chunk.
//   k_rp_count  per 16 KB block: newlines, kept bytes for every state the block could start in, the state it ends in
//   k_rp_scan   one block: the blocks' start states and output offsets (<= 2048 blocks per chunk), the state after the chunk
//               (prefix scans over the summaries: line counts, "the last line start before me", kept bytes)
//   k_rp_write  per block: the kept bytes, staged in LDS, stored coalesced
#pragma once
#include "kmer_device.h"

#define RP_NT 256
#define RP_BPT 64
#define RP_BLOCK (RP_NT * RP_BPT)             // 16 KB of text per block
#define RP_FASTA 1
#define RP_FASTQ 2

struct RawState { unsigned long long lines, out_len, recs; u32 chk, hdr, bad, prev_nl, fresh, fq; };      // chk: FASTQ, sum over the records of w(record) * (sequence characters - quality characters): 0 iff every record has as many of one as of the other      // lines: of the current file; recs: records (header lines) since the raw pushes began      // fresh: a new file starts with the next chunk
// what a block tells: newlines; FASTQ: kept bytes by (start line % 4); FASTA: kept[start in header ? 1 : 0], has a line start, header state at its end
struct RpBlock { u32 nl; u32 kept[4]; u32 has_ls, end_hdr, pad; };

__host__ __device__ __forceinline__ bool rp_hdr_char(unsigned char c) { return c == '>'; }
// a FASTQ file at its end: every record has as many quality characters as bases.  Checked without pairing lines: every character of
// a sequence line adds w(record), every character of a quality line subtracts it, w = a 32-bit mix of the record's number -- the sum
// over the file (wrapping in 32 bits) is 0 iff the records balance one by one (two records that are off by +1 and -1 do not cancel)
__host__ __device__ __forceinline__ u32 rp_weight(u32 rec) { u32 h = rec + 0x9e3779b9u; h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h | 1u; }      // (a 32-bit mix: the sums below wrap in 32 bits -- a damaged file passes with probability 2^-32)
__host__ __device__ __forceinline__ bool rp_file_ok(const RawState& s) { return !s.fq || s.chk == 0u; }
__device__ __forceinline__ bool rp_blank(unsigned char c) { return c == '\n' || c == '\r' || c == ' ' || c == '\t'; }      // what the host parser drops from a sequence line (host/bank.cpp append_seq)

// the thread's 64 bytes as 16 words (bytes past the end of the chunk read as '\n'); -> how many of them exist
__device__ __forceinline__ u32 rp_load(const unsigned char* __restrict__ in, u32 b0, u32 n, u32 (&w)[16]) {
    if (b0 + RP_BPT <= n) {
        const uint4* p = reinterpret_cast<const uint4*>(in + b0);          // (chunk buffers are 256-byte aligned, b0 is a multiple of 64)
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint4 v = p[q]; w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w; }
        return RP_BPT;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = 0x0A0A0A0Au;
    const u32 m = b0 < n ? n - b0 : 0u;
#pragma unroll
    for (int i = 0; i < RP_BPT; ++i)          // (unrolled: w[] stays in registers)
        if ((u32)i < m) w[i >> 2] = (w[i >> 2] & ~(0xFFu << (8 * (i & 3)))) | ((u32)in[b0 + i] << (8 * (i & 3)));
    return m;
}
#define RP_BYTE(w, i) ((unsigned char)(((w)[(i) >> 2] >> (8 * ((i) & 3))) & 0xFFu))

// ---- block-wide prefix helpers (RP_NT = 256 threads = 4 waves; every thread of the block calls them)
// exclusive prefix sum of v over the block's threads; -> total
__device__ __forceinline__ u32 rp_excl_sum(u32 v, u32* s_w, u32& total) {
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const u32 inc = wave_incl_scan(v);
    __syncthreads();                                   // (s_w may still be read from the previous call)
    if (lane == 63u) s_w[wave] = inc;
    __syncthreads();
    u32 base = 0; total = 0;
#pragma unroll
    for (u32 x = 0; x < RP_NT / 64; ++x) { const u32 t = s_w[x]; if (x < wave) base += t; total += t; }
    return base + inc - v;
}
// "the last thread BEFORE me whose flag is set": -> its val (0 / 1), or 2 when there is none in the block.  last: the same seen from
// behind the block's last thread
__device__ __forceinline__ u32 rp_last_flagged(bool flag, u32 val, u32* s_w, u32& last) {
    const u32 lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(flag), vals = __ballot(flag && val);
    __syncthreads();
    if (lane == 0) s_w[wave] = mask ? (2u | (u32)((vals >> (63 - __clzll((long long)mask))) & 1ull)) : 0u;      // bit 1: the wave has one, bit 0: val of its last
    __syncthreads();
    u32 inh = 2u; last = 2u;
#pragma unroll
    for (u32 x = 0; x < RP_NT / 64; ++x) { const u32 t = s_w[x]; if (t & 2u) { if (x < wave) inh = t & 1u; last = t & 1u; } }
    const unsigned long long prior = mask & ((1ull << lane) - 1ull);
    return prior ? (u32)((vals >> (63 - __clzll((long long)prior))) & 1ull) : inh;
}

template <int FMT>
__global__ __launch_bounds__(RP_NT) void k_rp_count(const unsigned char* __restrict__ in, u32 n, RpBlock* __restrict__ blk) {
    __shared__ u32 s_w[RP_NT / 64];
    __shared__ u32 s_sum[4];
    const u32 tid = threadIdx.x;
    const u32 b0 = blockIdx.x * RP_BLOCK + tid * RP_BPT;
    u32 w[16];
    const u32 m = rp_load(in, b0, n, w);
    unsigned char prev = (b0 > 0 && b0 <= n) ? in[b0 - 1] : (unsigned char)0;      // (the chunk's first byte: the scan knows whether it starts a line)
    u32 nl = 0, kq[4] = {0, 0, 0, 0}, has_ls = 0, hdr = 0;      // FASTQ: kq[r] = bytes on the thread's relative line r % 4; FASTA: kq[0 / 1] = kept before the thread's first line start when it starts outside / inside a header, kq[2] = kept from there on
#pragma unroll
    for (int i = 0; i < RP_BPT; ++i) {
        if ((u32)i < m) {
            const unsigned char c = RP_BYTE(w, i);
            if (FMT == RP_FASTQ) {
                const u32 one = c != '\r' ? 1u : 0u;
                kq[0] += (nl & 3u) == 0u ? one : 0u; kq[1] += (nl & 3u) == 1u ? one : 0u; kq[2] += (nl & 3u) == 2u ? one : 0u; kq[3] += (nl & 3u) == 3u ? one : 0u;
            } else {
                if (prev == '\n') { has_ls = 1; hdr = rp_hdr_char(c) ? 1u : 0u; }
                const u32 keep_seq = rp_blank(c) ? 0u : 1u, keep_hdr = c == '\n' ? 1u : 0u;
                if (has_ls) kq[2] += hdr ? keep_hdr : keep_seq;
                else { kq[0] += keep_seq; kq[1] += keep_hdr; }
            }
            if (c == '\n') ++nl;
            prev = c;
        }
    }
    if (tid < 4) s_sum[tid] = 0;
    // the threads' summaries composed in order: a prefix over the line counts (FASTQ) / "the last line start before me" (FASTA)
    u32 nl_total = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, last = 2u;
    if (FMT == RP_FASTQ) {
        const u32 L = rp_excl_sum(nl, s_w, nl_total);                  // the thread's relative line x is the block's relative line L + x
        const u32 r = L & 3u;
        c0 = r == 0 ? kq[0] : r == 1 ? kq[3] : r == 2 ? kq[2] : kq[1];      // c[y] = kq[(y - r) & 3]
        c1 = r == 0 ? kq[1] : r == 1 ? kq[0] : r == 2 ? kq[3] : kq[2];
        c2 = r == 0 ? kq[2] : r == 1 ? kq[1] : r == 2 ? kq[0] : kq[3];
        c3 = r == 0 ? kq[3] : r == 1 ? kq[2] : r == 2 ? kq[1] : kq[0];
    } else {
        (void)rp_excl_sum(nl, s_w, nl_total);
        const u32 inh = rp_last_flagged(has_ls != 0, hdr, s_w, last);      // the state the thread's bytes before its own first line start continue
        if (inh == 2u) { c0 = kq[0]; c1 = kq[1]; }                       // ... the block's own start state: kept for either
        else c2 = inh ? kq[1] : kq[0];
        c2 += kq[2];
    }
    c0 = wave_incl_scan(c0); c1 = wave_incl_scan(c1); c2 = wave_incl_scan(c2); c3 = wave_incl_scan(c3);      // (lane 63: the wave's sums)
    __syncthreads();
    if ((tid & 63u) == 63u) { atomicAdd(&s_sum[0], c0); atomicAdd(&s_sum[1], c1); atomicAdd(&s_sum[2], c2); atomicAdd(&s_sum[3], c3); }
    __syncthreads();
    if (tid == 0) {
        RpBlock r; r.nl = nl_total; r.kept[0] = s_sum[0]; r.kept[1] = s_sum[1]; r.kept[2] = s_sum[2]; r.kept[3] = s_sum[3];
        r.has_ls = last != 2u ? 1u : 0u; r.end_hdr = last == 1u ? 1u : 0u; r.pad = 0;
        blk[blockIdx.x] = r;
    }
}

// one block: start state and output offset of every block, the state after the chunk.  Thread t takes the RP_SCAN_PER consecutive
// summaries from t * RP_SCAN_PER on (<= 2048 per chunk); three block-wide prefixes.  bstate[nblocks] = "the chunk's first byte starts a line"
#define RP_SCAN_PER 8
template <int FMT>
__global__ __launch_bounds__(RP_NT) void k_rp_scan(const unsigned char* __restrict__ in, u32 n, u32 nblocks, const RpBlock* __restrict__ blk, RawState* __restrict__ st,
                                                   unsigned long long* __restrict__ boff, u32* __restrict__ bstate, unsigned char* __restrict__ out) {
    __shared__ u32 s_w[RP_NT / 64];
    __shared__ RawState s;
    const u32 tid = threadIdx.x;
    if (tid == 0) {
        s = *st;
        if (s.fresh) {                                   // a new file: its first byte starts a line; a separator behind what came before
            if (!rp_file_ok(s)) s.bad = 1;               // (the file before it, now that it is complete)
            if (s.out_len) out[s.out_len++] = '\n';
            s.lines = 0; s.hdr = 0; s.prev_nl = 1; s.fresh = 0; s.chk = 0; s.fq = FMT == RP_FASTQ ? 1u : 0u;
        }
        bstate[nblocks] = s.prev_nl;
        if (FMT == RP_FASTA && n && s.prev_nl) s.hdr = rp_hdr_char(in[0]) ? 1u : 0u;      // (the chunk's first byte starts a line: k_rp_count could not see that)
    }
    __syncthreads();
    const u32 line0 = (u32)s.lines, hdr0 = s.hdr;      // (line numbers travel as their low 32 bits: the record weights repeat after 2^34 lines)
    const unsigned long long out0 = s.out_len;
    RpBlock r[RP_SCAN_PER];
    u32 mine = 0;                                        // how many of the summaries exist
#pragma unroll
    for (int x = 0; x < RP_SCAN_PER; ++x) {
        const u32 b = tid * RP_SCAN_PER + x;
        if (b < nblocks) { r[x] = blk[b]; mine = x + 1; }
        else { r[x].nl = 0; r[x].kept[0] = r[x].kept[1] = r[x].kept[2] = r[x].kept[3] = 0; r[x].has_ls = 0; r[x].end_hdr = 0; }
    }
    u32 nls = 0, any = 0, endh = 0;
#pragma unroll
    for (int x = 0; x < RP_SCAN_PER; ++x) { nls += r[x].nl; if (r[x].has_ls) { any = 1; endh = r[x].end_hdr; } }
    u32 nl_total = 0, lastf = 2u;
    const u32 lbase = rp_excl_sum(nls, s_w, nl_total);                  // newlines before the thread's first block
    u32 hstate = hdr0;
    if (FMT == RP_FASTA) { const u32 inh = rp_last_flagged(any != 0, endh, s_w, lastf); if (inh != 2u) hstate = inh; }
    // the blocks' start states and what each keeps
    u32 stt[RP_SCAN_PER], kept[RP_SCAN_PER], ksum = 0;
    {
        u32 L = line0 + lbase, h = hstate;
#pragma unroll
        for (int x = 0; x < RP_SCAN_PER; ++x) {
            if (FMT == RP_FASTQ) {
                stt[x] = L;                                                 // (FASTQ: the line number at the block's first byte)
                const u32 sel = (1u - L) & 3u;                              // relative line x is a sequence line iff (start + x) % 4 == 1
                kept[x] = sel == 0 ? r[x].kept[0] : sel == 1 ? r[x].kept[1] : sel == 2 ? r[x].kept[2] : r[x].kept[3];
                L += r[x].nl;
            } else {
                stt[x] = h;
                kept[x] = (h ? r[x].kept[1] : r[x].kept[0]) + r[x].kept[2];
                if (r[x].has_ls) h = r[x].end_hdr;
            }
            ksum += kept[x];
        }
    }
    u32 ktotal = 0;
    u32 kbase = rp_excl_sum(ksum, s_w, ktotal);                          // (a chunk is <= 32 MB: 32 bits)
#pragma unroll
    for (int x = 0; x < RP_SCAN_PER; ++x) {
        const u32 b = tid * RP_SCAN_PER + x;
        if ((u32)x < mine) { boff[b] = out0 + kbase; bstate[b] = stt[x]; }
        kbase += kept[x];
    }
    if (tid == 0) {
        s.out_len = out0 + ktotal;
        s.lines += nl_total;
        if (FMT == RP_FASTA && lastf != 2u) s.hdr = lastf;
        if (n) s.prev_nl = in[n - 1] == '\n' ? 1u : 0u;
        *st = s;
    }
}

// (a context's first raw push, or one behind dskgpu_push_reads: the stream so far is `out_len` bytes)
__global__ void k_rp_init(RawState* st, unsigned long long out_len) {
    RawState s; s.lines = 0; s.out_len = out_len; s.recs = 0; s.chk = 0; s.hdr = 0; s.bad = 0; s.prev_nl = 1; s.fresh = 1; s.fq = 0;
    *st = s;
}
__global__ void k_rp_fresh(RawState* st) { st->fresh = 1; }

template <int FMT>
__global__ __launch_bounds__(RP_NT) void k_rp_write(const unsigned char* __restrict__ in, u32 n, const unsigned long long* __restrict__ boff,
                                                    const u32* __restrict__ bstate, RawState* __restrict__ st, unsigned char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[RP_BLOCK + 16];
    __shared__ u32 s_w[RP_NT / 64];
    const u32 tid = threadIdx.x;
    const u32 b0 = blockIdx.x * RP_BLOCK + tid * RP_BPT;
    u32 w[16];
    const u32 m = rp_load(in, b0, n, w);
    const unsigned char prev0 = b0 == 0 ? (unsigned char)(bstate[gridDim.x] ? '\n' : 0) : ((b0 <= n) ? in[b0 - 1] : (unsigned char)0);
    // what every thread needs to know about the ones before it: newlines (FASTQ), line starts and the header state behind them (FASTA)
    u32 nl = 0, has_ls = 0, eh = 0;
    {
        unsigned char prev = b0 == 0 ? (unsigned char)0 : prev0;          // (the chunk's first byte as a line start is already in bstate[0])
#pragma unroll
        for (int i = 0; i < RP_BPT; ++i) {
            if ((u32)i < m) {
                const unsigned char c = RP_BYTE(w, i);
                if (FMT == RP_FASTA && prev == '\n') { has_ls = 1; eh = rp_hdr_char(c) ? 1u : 0u; }
                if (c == '\n') ++nl;
                prev = c;
            }
        }
    }
    const u32 bst = bstate[blockIdx.x];                  // FASTQ: the line number at the block's first byte; FASTA: inside a header line
    u32 state, tot, lastf, lineno = 0;
    if (FMT == RP_FASTQ) { lineno = bst + rp_excl_sum(nl, s_w, tot); state = lineno & 3u; }
    else { const u32 inh = rp_last_flagged(has_ls != 0, eh, s_w, lastf); state = inh == 2u ? bst : inh; }
    // count what the thread keeps, then place it
    u32 keepm[2] = {0u, 0u}, cnt = 0, bad = 0, recs = 0;           // bit i: byte i is kept
    u32 wgt = FMT == RP_FASTQ ? rp_weight(lineno >> 2) : 0u, chk = 0u;      // (FASTQ) this record's weight; sequence minus quality characters, weighted
    u32 cl = 0;
    {
        unsigned char prev = prev0;
#pragma unroll
        for (int i = 0; i < RP_BPT; ++i) {
            if ((u32)i < m) {
                const unsigned char c = RP_BYTE(w, i);
                const bool ls = prev == '\n';
                bool k;
                if (FMT == RP_FASTQ) {
                    if (ls && state == 2u && c != '+') bad = 1;
                    if (ls && state == 0u && c != '@' && c != '\n') {      // where a header would be: only a BLANK line is let through (the end of a file; a host parser skips it) --
                        // empty, or a lone '\r' in front of its '\n'.  (What follows a '\r' at the very end of a piece is not known here: given back.)
                        const u32 nxi = b0 + (u32)i + 1u;
                        const unsigned char nx = (i + 1 < RP_BPT && (u32)(i + 1) < m) ? RP_BYTE(w, (i + 1) & (RP_BPT - 1)) : (nxi < n ? in[nxi] : (unsigned char)0);
                        if (c != '\r' || nx != '\n') bad = 1;
                    }
                    if (ls && state == 0u && c == '@') ++recs;
                    k = state == 1u && c != '\r';
                    if (state == 1u && (c == ' ' || c == '\t')) bad = 1;      // (the host parser drops blanks inside a sequence line: leave such a file to it)
                    cl += (c != '\n' && c != '\r') ? 1u : 0u;              // characters of the current line seen by this thread
                    if (c == '\n') {
                        if (state == 1u) chk += wgt * cl; else if (state == 3u) chk -= wgt * cl;
                        cl = 0; ++lineno; state = lineno & 3u;
                        if (state == 0u) wgt = rp_weight(lineno >> 2);
                    }
                } else {
                    if (ls) { state = rp_hdr_char(c) ? 1u : 0u; recs += state; if (c == '@' || c == '+') bad = 1; }      // (FASTQ records inside a FASTA file: a host parser switches per record, this one is told one format per file)
                    k = state ? c == '\n' : !rp_blank(c);
                }
                if (k) { keepm[i >> 5] |= 1u << (i & 31); ++cnt; }
                prev = c;
            }
        }
    }
    if (FMT == RP_FASTQ) { if (state == 1u) chk += wgt * cl; else if (state == 3u) chk -= wgt * cl; }      // (the line the thread ends in)
    // staged at the offset the block's bytes have inside their 16-byte group of the output: whole groups then leave as 16-byte stores
    const unsigned long long gb = boff[blockIdx.x];
    const u32 a = (u32)(gb & 15ull);
    u32 total = 0;
    {
        u32 o = a + rp_excl_sum(cnt, s_w, total);
#pragma unroll
        for (int i = 0; i < RP_BPT; ++i) if ((keepm[i >> 5] >> (i & 31)) & 1u) stage[o++] = RP_BYTE(w, i);
    }
    __syncthreads();
    unsigned char* base = out + (gb - a);                     // 16-byte aligned (the stream's buffer is)
    const u32 end = a + total, g0 = (a + 15u) >> 4, g1 = end >> 4;
    for (u32 g = g0 + tid; g < g1; g += RP_NT) reinterpret_cast<uint4*>(base)[g] = reinterpret_cast<const uint4*>(stage)[g];
    const u32 h_end = end < g0 * 16u ? end : g0 * 16u;          // the bytes in front of the first whole group ...
    for (u32 i = a + tid; i < h_end; i += RP_NT) base[i] = stage[i];
    const u32 t_beg = g1 * 16u > h_end ? g1 * 16u : h_end;      // ... and behind the last
    for (u32 i = t_beg + tid; i < end; i += RP_NT) base[i] = stage[i];
    // one atomic per block (8192 wave-level ones on one address were most of this kernel's time)
    u32 rtotal = 0;
    (void)rp_excl_sum(recs, s_w, rtotal);
    const int any_bad = __syncthreads_or((int)bad);
    if (tid == 0) {
        if (rtotal) atomicAdd(&st->recs, (unsigned long long)rtotal);
        if (any_bad) atomicOr(&st->bad, 1u);
    }
    if (FMT == RP_FASTQ) {                               // the block's share of the file's checksum (wrapping 64-bit sums: order does not matter)
        u32 ctotal = 0;
        (void)rp_excl_sum(chk, s_w, ctotal);
        if (tid == 0 && ctotal) atomicAdd(&st->chk, ctotal);
    }
}
