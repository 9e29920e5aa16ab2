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
//   FASTA  a line that starts with '>' is a header: dropped but for its '\n', which separates the records; the bytes of all
//          other lines are kept except '\n', '\r', ' ' and '\t', so that a sequence wrapped over lines is one run of bases (test/longread.fasta).
// The state between chunks -- lines so far, "inside a header line", "the next byte starts a line", bytes written -- lives on the device
// (RawState): no round trip per chunk.
//   k_rp_count  per 16 KB block: newlines, kept bytes for every state the block could start in, the state it ends in
//   k_rp_scan   one block: the blocks' start states and output offsets (<= 2048 blocks per chunk), the state after the chunk
//   k_rp_write  per block: the kept bytes, staged in LDS, stored coalesced
#pragma once
#include "kmer_device.h"

#define RP_NT 256
#define RP_BPT 64
#define RP_BLOCK (RP_NT * RP_BPT)             // 16 KB of text per block
#define RP_FASTA 1
#define RP_FASTQ 2

struct RawState { unsigned long long lines, out_len, recs; u32 hdr, bad, prev_nl, fresh; };      // lines: of the current file; recs: records (header lines) since the raw pushes began      // fresh: a new file starts with the next chunk
// what a block tells: newlines; FASTQ: kept bytes by (start line % 4); FASTA: kept[start in header ? 1 : 0], has a line start, header state at its end
struct RpBlock { u32 nl; u32 kept[4]; u32 has_ls, end_hdr, pad; };

__device__ __forceinline__ bool rp_hdr_char(unsigned char c) { return c == '>'; }
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

template <int FMT>
__global__ __launch_bounds__(RP_NT) void k_rp_count(const unsigned char* __restrict__ in, u32 n, RpBlock* __restrict__ blk) {
    __shared__ u32 s_nl[RP_NT], s_k[RP_NT][4], s_ls[RP_NT], s_eh[RP_NT];
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
    s_nl[tid] = nl; s_ls[tid] = has_ls; s_eh[tid] = hdr;
#pragma unroll
    for (int x = 0; x < 4; ++x) s_k[tid][x] = kq[x];
    __syncthreads();
    if (tid == 0) {          // compose the threads' summaries in order (256 short steps per 16 KB of text)
        RpBlock r; r.nl = 0; r.kept[0] = r.kept[1] = r.kept[2] = r.kept[3] = 0; r.has_ls = 0; r.end_hdr = 0; r.pad = 0;
        if (FMT == RP_FASTQ) {
            for (u32 t = 0; t < RP_NT; ++t) {
#pragma unroll
                for (u32 x = 0; x < 4; ++x) r.kept[(r.nl + x) & 3u] += s_k[t][x];      // the thread's relative line x is the block's relative line nl + x
                r.nl += s_nl[t];
            }
        } else {
            u32 k0 = 0, k1 = 0, kd = 0;            // kept if the block starts outside / inside a header (until the block's first line start), kept after it
            for (u32 t = 0; t < RP_NT; ++t) {
                if (!r.has_ls) { k0 += s_k[t][0]; k1 += s_k[t][1]; }
                else kd += r.end_hdr ? s_k[t][1] : s_k[t][0];      // (the part of a thread before its own first line start continues the state it inherits)
                if (s_ls[t]) { kd += s_k[t][2]; r.has_ls = 1; r.end_hdr = s_eh[t]; }
                r.nl += s_nl[t];
            }
            r.kept[0] = k0; r.kept[1] = k1; r.kept[2] = kd;
        }
        blk[blockIdx.x] = r;
    }
}

// one block: start state and output offset of every block (walked by one thread over LDS tiles the whole block loads); the state after
// the chunk.  bstate[nblocks] = "the chunk's first byte starts a line"
#define RP_SCAN_TILE 512
template <int FMT>
__global__ __launch_bounds__(RP_NT) void k_rp_scan(const unsigned char* __restrict__ in, u32 n, u32 nblocks, const RpBlock* __restrict__ blk, RawState* __restrict__ st,
                                                   unsigned long long* __restrict__ boff, u32* __restrict__ bstate, unsigned char* __restrict__ out) {
    __shared__ RpBlock s_b[RP_SCAN_TILE];
    __shared__ unsigned long long s_o[RP_SCAN_TILE];
    __shared__ u32 s_s[RP_SCAN_TILE];
    __shared__ RawState s;
    const u32 tid = threadIdx.x;
    if (tid == 0) {
        s = *st;
        if (s.fresh) {                                   // a new file: its first byte starts a line; a separator behind what came before
            if (s.out_len) out[s.out_len++] = '\n';
            s.lines = 0; s.hdr = 0; s.prev_nl = 1; s.fresh = 0;
        }
        bstate[nblocks] = s.prev_nl;
        if (FMT == RP_FASTA && n && s.prev_nl) s.hdr = rp_hdr_char(in[0]) ? 1u : 0u;      // (the chunk's first byte starts a line: k_rp_count could not see that)
    }
    for (u32 base = 0; base < nblocks; base += RP_SCAN_TILE) {
        const u32 cnt = nblocks - base < RP_SCAN_TILE ? nblocks - base : RP_SCAN_TILE;
        __syncthreads();
        for (u32 i = tid; i < cnt; i += RP_NT) s_b[i] = blk[base + i];
        __syncthreads();
        if (tid == 0) {
            for (u32 b = 0; b < cnt; ++b) {
                const RpBlock r = s_b[b];
                s_o[b] = s.out_len;
                if (FMT == RP_FASTQ) {
                    s_s[b] = (u32)(s.lines & 3ull);
                    s.out_len += r.kept[(1u - (u32)(s.lines & 3ull)) & 3u];      // relative line x is a sequence line iff (start + x) % 4 == 1
                } else {
                    s_s[b] = s.hdr;
                    s.out_len += (s.hdr ? r.kept[1] : r.kept[0]) + r.kept[2];
                    if (r.has_ls) s.hdr = r.end_hdr;
                }
                s.lines += r.nl;
            }
        }
        __syncthreads();
        for (u32 i = tid; i < cnt; i += RP_NT) { boff[base + i] = s_o[i]; bstate[base + i] = s_s[i]; }
    }
    if (tid == 0) {
        if (n) s.prev_nl = in[n - 1] == '\n' ? 1u : 0u;
        *st = s;
    }
}

// (a context's first raw push, or one behind dskgpu_push_reads: the stream so far is `out_len` bytes)
__global__ void k_rp_init(RawState* st, unsigned long long out_len) {
    RawState s; s.lines = 0; s.out_len = out_len; s.recs = 0; s.hdr = 0; s.bad = 0; s.prev_nl = 1; s.fresh = 1;
    *st = s;
}
__global__ void k_rp_fresh(RawState* st) { st->fresh = 1; }

template <int FMT>
__global__ __launch_bounds__(RP_NT) void k_rp_write(const unsigned char* __restrict__ in, u32 n, const unsigned long long* __restrict__ boff,
                                                    const u32* __restrict__ bstate, RawState* __restrict__ st, unsigned char* __restrict__ out) {
    __shared__ unsigned char stage[RP_BLOCK];
    __shared__ u32 s_cnt[RP_NT], s_nl[RP_NT], s_ls[RP_NT], s_eh[RP_NT], s_off[RP_NT], s_start[RP_NT];
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
    s_nl[tid] = nl; s_ls[tid] = has_ls; s_eh[tid] = eh;
    __syncthreads();
    if (tid == 0) {
        u32 stt = bstate[blockIdx.x];            // FASTQ: line % 4 at the block's first byte; FASTA: inside a header line
        for (u32 t = 0; t < RP_NT; ++t) {
            s_start[t] = stt;
            if (FMT == RP_FASTQ) stt = (stt + s_nl[t]) & 3u;
            else if (s_ls[t]) stt = s_eh[t];
        }
    }
    __syncthreads();
    // count what the thread keeps, then place it
    u32 keepm[2] = {0u, 0u}, cnt = 0, bad = 0, recs = 0;           // bit i: byte i is kept
    {
        u32 state = s_start[tid];
        unsigned char prev = prev0;
#pragma unroll
        for (int i = 0; i < RP_BPT; ++i) {
            if ((u32)i < m) {
                const unsigned char c = RP_BYTE(w, i);
                const bool ls = prev == '\n';
                bool k;
                if (FMT == RP_FASTQ) {
                    if (ls && ((state == 0u && c != '@') || (state == 2u && c != '+')) && c != '\n' && c != '\r') bad = 1;      // (blank lines at the end of a file are let through)
                    if (ls && state == 0u && c == '@') ++recs;
                    k = state == 1u && c != '\r';
                    if (state == 1u && (c == ' ' || c == '\t')) bad = 1;      // (the host parser drops blanks inside a sequence line: leave such a file to it)
                    if (c == '\n') state = (state + 1u) & 3u;
                } else {
                    if (ls) { state = rp_hdr_char(c) ? 1u : 0u; recs += state; }
                    k = state ? c == '\n' : !rp_blank(c);
                }
                if (k) { keepm[i >> 5] |= 1u << (i & 31); ++cnt; }
                prev = c;
            }
        }
    }
    s_cnt[tid] = cnt;
    __syncthreads();
    if (tid == 0) { u32 run = 0; for (u32 t = 0; t < RP_NT; ++t) { s_off[t] = run; run += s_cnt[t]; } s_cnt[0] = run; }
    __syncthreads();
    const u32 total = s_cnt[0];
    {
        u32 o = s_off[tid];
#pragma unroll
        for (int i = 0; i < RP_BPT; ++i) if ((keepm[i >> 5] >> (i & 31)) & 1u) stage[o++] = RP_BYTE(w, i);
    }
    __syncthreads();
    unsigned char* dst = out + boff[blockIdx.x];
    for (u32 i = tid; i < total; i += RP_NT) dst[i] = stage[i];
    if (bad) atomicOr(&st->bad, 1u);
    recs = wave_incl_scan(recs);            // (lane 63: the wave's sum)
    if ((tid & 63u) == 63u && recs) atomicAdd(&st->recs, (unsigned long long)recs);
}
