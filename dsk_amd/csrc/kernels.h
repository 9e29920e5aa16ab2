// kernels.h -- HIP kernels of the count path (gfx950 / MI355X, wave64).
//
// Pipeline (all buffers resident in HBM; replaces the disk partitions of
// doc/paper.tex:60-97 and gatb-core's fillPartitions / fillSolidKmers stages
// named at scripts/quick-build.sh:52-57):
//
//   k_encode        ASCII read stream -> 2-bit packed + invalid mask          (K1)
//   k_hist_*        per-chunk digit histogram  (LDS histogram, no global atomics) (K3, "PartiInfo")
//   k_scan_*        exclusive scan of the (bin-major) chunk x bin matrix
//   k_scatter_*     LDS-staged radix scatter into partition-contiguous HBM    (K4)
//   k_plan_*        chunk descriptors of the next level from the scanned matrix
//   k_count         per-sub-partition open-addressing hash aggregate in LDS,
//                   histogram + solidity filter fused into the table sweep    (K5+K6)
//   k_compact       gather solid rows to a dense array, un-mix the keys
//
// Keys in the partition arrays are h = kmix(canonical k-mer) (bijective), so a
// radix digit is (h >> shift) & mask and the table slot another bit field of h.
// Work is split into static "chunks"; chunk c of segment s owns matrix entries
// flat_base + bin*stride, laid out so that ONE linear exclusive scan yields
// every (segment, bin, chunk) destination offset: no global atomics, and the
// output order of each level is deterministic.
#pragma once
#include "kmer_device.h"

#ifndef SC_NT
#define SC_NT 1024           // threads of hist/scatter blocks: one block per CU stages a 128 KB tile in LDS
#endif
// Tile geometry: 128 KB of keys staged in LDS per tile whatever the key width
// (longer runs per bin and fewer open write streams per XCD L2 than 2 x 64 KB tiles: measured -2.9 ms).
template <int W> struct Tile {
    static constexpr int KPT = 16 / W;            // keys (or read positions) per thread per tile
    static constexpr int KEYS = SC_NT * KPT;      // 16384 one-word / 8192 two-word keys = 128 KB
    static constexpr int WORDS = KEYS / 32;       // packed read words per tile
};
#define MAX_BINS 2048

// Workgroup barrier that orders LDS traffic only.  HIP's __syncthreads() also
// drains every outstanding global load (s_waitcnt vmcnt(0)), which would kill
// the register prefetch of the next tile / sub-partition; this one waits for
// LDS (lgkmcnt) and leaves HBM reads in flight across the barrier
// (cdna_hip_programming.md "Pipelining across barriers").  Global data is never
// exchanged between threads inside these kernels, so no vmcnt wait is needed.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct ChunkDesc {
    u64 begin, end;          // source range: packed words (reads) or keys
    u32 flat_base;           // matrix entry of bin 0
    u32 stride;              // matrix stride between bins (= chunks in the segment)
};

// ------------------------------------------------------------------ K1
// One thread encodes 32 bases (two 16-byte loads) into one packed word.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_encode(const uint8_t* __restrict__ s, u64 n,
                                                u64* __restrict__ packed, u32* __restrict__ inval, u64 nwords) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x; w < nwords; w += stride) {
        const u64 base = w * 32;
        u32 x[8];
        if (ALIGNED && base + 32 <= n) {
            const uint4 a = *reinterpret_cast<const uint4*>(s + base);
            const uint4 b = *reinterpret_cast<const uint4*>(s + base + 16);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32 v = 0;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const u64 p = base + q * 4 + b;
                    const u32 c = p < n ? s[p] : (u32)'\n';
                    v |= c << (8 * b);
                }
                x[q] = v;
            }
        }
        // Four bytes at a time.  Codes: (c >> 1) & 3 of every byte, the four 2-bit fields gathered MSB first by one multiply (byte 0 ->
        // bits 31..30, byte 1 -> 29..28, ...: no two partial products meet in the top byte).  Validity: a byte is a base iff
        // (c & 0xC0) == 0x40 and its low five bits are one of A 00001, C 00011, G 00111, T 10100 -- bit (c & 31) of a 32-bit mask
        // (the shift takes c's low five bits by itself); the per-byte flags are gathered by a second multiply.  (r05: ~6 VALU
        // instructions per base instead of ~9 -- the kernel is not as memory-bound as its 4.7 TB/s suggest.)
        u64 pk = 0; u32 iv = 0;
        constexpr u32 BASES = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32 xx = x[q];
            const u32 codes = ((((xx >> 1) & 0x03030303u) * 0x40100401u) >> 24);                       // 8 bits: bases 4q .. 4q + 3
            const u32 hi = (xx & 0xC0C0C0C0u) ^ 0x40404040u;                                            // zero byte <=> 0x40 <= c < 0x80
            u32 ok = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) ok |= ((BASES >> ((xx >> (8 * b)) & 31u)) & 1u) << (8 * b);     // bit 8b: the low five bits name a base
            const u32 hz = (((hi & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | hi) >> 7;                              // bit 8b: byte b of hi is NOT zero
            const u32 good = ok & ~hz & 0x01010101u;
            const u32 bad4 = (((good ^ 0x01010101u) * 0x08040201u) >> 24) & 0xFu;                       // bit 3: byte 0 ... bit 0: byte 3
            pk |= (u64)codes << (56 - 8 * q);
            iv |= bad4 << (28 - 4 * q);
        }
        packed[w] = pk;
        inval[w] = iv;
    }
}

// ------------------------------------------------------------------ key sources
// READS: chunk range is in packed words; a tile is Tile<W>::WORDS words; each
// thread generates Tile<W>::KPT consecutive window end positions of one word.
template <int W> struct KeyT { typedef KN<W> T; };
template <> struct KeyT<1> { typedef u64 T; };

__device__ __forceinline__ u64 digit_word(u64 h) { return h; }
template <int W> __device__ __forceinline__ u64 digit_word(const KN<W>& h) { return h.w[W - 1]; }

// Radix digits are bit fields of the mixed key, mapped onto an ARBITRARY number
// of bins with the multiply-shift range reduction (one v_mul_hi_u32):
//   level 1 : d1 = mulhi(h[63:32], P1)
//   level 2 : d2 = mulhi(lo32(h[63:32] * P1), P2)        (the fraction left over by d1)
//   owner   : g  = (h[31:12] * G) >> 20                  (multi-GPU owner of the k-mer)
//   slot    : h[11:0]                                    (home slot of the LDS table)
// so final sub-partition q = d1 * P2 + d2 and the fields are independent.
//   pass    : the fraction left over by the owner in h[31:12], mapped onto npass passes: inputs with more
//             k-mers than one pass may hold (2^32 offsets / the HBM budget) are counted in several
//             passes over the encoded reads, each keeping only its share of the key space -- the
//             in-HBM counterpart of DSK's disk passes (doc/paper.tex:65-67, README.md:126-130).
struct DigitSpec { u32 mode, pa, pb, world, npass, pass; };   // mode 0: owner(pa=G)  1: level1(pa=P1)  2: level2(pa=P1,pb=P2)
// MODE (compile time): 0 owner, 1 level 1, 2 level 2, 3 level 1 of a multi-pass count (adds the pass filter;
// a separate instantiation so that the single-pass tile body stays branch-free)
// pass of a key (of ds.npass): the fraction left over by the owner in h[31:12]
__device__ __forceinline__ u32 key_pass(u64 w, const DigitSpec& ds) {
    const u32 frac = (((u32)(w >> 12) & 0xFFFFFu) * ds.world) & 0xFFFFFu;
    return (frac * ds.npass) >> 20;
}
// MODE 4 ("level 0" of a multi-pass count): the digit is the PASS of the key, counted from ds.pass, for the ds.pa passes
// materialised together -- one sweep over the reads writes the keys of a group of passes grouped by pass, and the passes then
// run from key arrays instead of re-generating every k-mer once per pass.
template <int MODE>
__device__ __forceinline__ u32 key_digit(u64 w, const DigitSpec& ds) {
    if (MODE == 1 || MODE == 3) return __umulhi((u32)(w >> 32), ds.pa);
    if (MODE == 2) return __umulhi((u32)(w >> 32) * ds.pa, ds.pb);
    if (MODE == 4) return key_pass(w, ds) - ds.pass;
    return (((u32)(w >> 12) & 0xFFFFFu) * ds.pa) >> 20;
}
// does this key belong to the pass (MODE 3) / the group of passes (MODE 4) being handled?
template <int MODE>
__device__ __forceinline__ bool key_in_pass(u64 w, const DigitSpec& ds) {
    if (MODE == 3) return key_pass(w, ds) == ds.pass;
    if (MODE == 4) return key_pass(w, ds) - ds.pass < ds.pa;
    return true;
}

// multi-word windows: the tuned two-word generator for W = 2, the general one otherwise
template <int NP>
__device__ __forceinline__ u32 gen_kmers_multi(const u64* __restrict__ packed, const u32* __restrict__ inval, u64 wi, int t0, int k, K2 (&c)[NP]) {
    return gen_kmers2<NP>(packed, inval, wi, t0, k, c);
}
template <int W, int NP>
__device__ __forceinline__ u32 gen_kmers_multi(const u64* __restrict__ packed, const u32* __restrict__ inval, u64 wi, int t0, int k, KN<W> (&c)[NP]) {
    return gen_kmersN<W, NP>(packed, inval, wi, t0, k, c);
}

template <int W>
__device__ __forceinline__ u32 tile_keys_reads(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                               u64 w0, u64 wend, int k, KN<W> (&h)[Tile<W>::KPT]) {
    constexpr int KPT = Tile<W>::KPT, TPW = 32 / KPT;   // threads per packed word, KPT windows each
    const u64 wi = w0 + (threadIdx.x / TPW);
    if (wi >= wend) return 0u;
    const u32 vm = gen_kmers_multi(packed, inval, wi, (int)(threadIdx.x % TPW) * KPT, k, h);
#pragma unroll
    for (int j = 0; j < KPT; ++j) kmixN(h[j]);
    return vm;
}
__device__ __forceinline__ u32 tile_keys_reads(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                               u64 w0, u64 wend, int k, u64 (&h)[16]) {
    const u64 wi = w0 + (threadIdx.x >> 1);             // 2 threads per packed word, 16 windows each
    if (wi >= wend) return 0u;
    const u32 vm = gen_kmers1<16>(packed, inval, wi, (threadIdx.x & 1) * 16, k, h);
#pragma unroll
    for (int j = 0; j < 16; ++j) h[j] = kmix(h[j]);
    return vm;
}

__device__ __forceinline__ bool is_pad_key(u64 h) { return h == DSK_EMPTY; }
template <int W> __device__ __forceinline__ bool is_pad_key(const KN<W>& h) {      // (multi-word key arrays with pads: the sampled records of the receive side)
    bool e = true;
#pragma unroll
    for (int x = 0; x < W; ++x) e = e && (h.w[x] == DSK_EMPTY);
    return e;
}
__device__ __forceinline__ bool keys_same(u64 a, u64 b) { return a == b; }
template <int W> __device__ __forceinline__ bool keys_same(const KN<W>& a, const KN<W>& b) { return key_eq(a, b); }
// KEYS: chunk range is in keys; tile t covers keys [begin + t*Tile<W>::KEYS, ...);
// thread loads keys tid + j*SC_NT (coalesced).
template <int W>
__device__ __forceinline__ u32 tile_keys_array(const typename KeyT<W>::T* __restrict__ in, u64 k0, u64 kend,
                                               typename KeyT<W>::T (&h)[Tile<W>::KPT]) {
    // branch-free: out-of-range lanes re-read the tile's last key (masked out by the
    // returned bits), so the loads stay one straight-line burst the compiler can count
    const typename KeyT<W>::T* base = in + k0;                       // wave-uniform
    const u64 left = kend - k0;                                      // >= 1
    const u32 n = left < (u64)Tile<W>::KEYS ? (u32)left : (u32)Tile<W>::KEYS;
    u32 vm = 0;
#pragma unroll
    for (int j = 0; j < Tile<W>::KPT; ++j) {
        const u32 o = threadIdx.x + (u32)j * SC_NT;
        const bool ok = o < n;
        h[j] = base[ok ? o : n - 1];
        vm |= ((ok && !is_pad_key(h[j])) ? 1u : 0u) << j;      // (the all-ones sentinel pads the slices a level-0 scatter leaves: never a key)
    }
    return vm;
}

// ------------------------------------------------------------------ K3: histogram
// SRC 0 = reads, 1 = key array; MODE = which digit (compile time, so the tile
// body is straight-line code).  Persistent blocks walk chunks; per chunk an LDS
// histogram over P bins (+1 dummy bin for invalid windows) is written to the
// matrix row of that chunk.
template <int W, int SRC, int MODE>
__global__ __launch_bounds__(SC_NT) void k_hist(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                const typename KeyT<W>::T* __restrict__ keys,
                                                const ChunkDesc* __restrict__ descs, const u32* __restrict__ d_nchunks,
                                                u32* __restrict__ matrix, int k, DigitSpec ds, u32 P) {
    __shared__ u32 lh[MAX_BINS + 1];
    constexpr int KPT = Tile<W>::KPT;
    const u32 nchunks = *d_nchunks;
    for (u32 g = blockIdx.x; g < nchunks; g += gridDim.x) {
        const ChunkDesc d = descs[g];
        for (u32 b = threadIdx.x; b <= P; b += SC_NT) lh[b] = 0;
        __syncthreads();
        const u64 step = SRC == 0 ? Tile<W>::WORDS : Tile<W>::KEYS;
        typename KeyT<W>::T ha[KPT], hb[KPT]; u32 vma = 0, vmb = 0;
        auto process = [&](typename KeyT<W>::T (&h)[KPT], u32 vm) {
            u32 dg[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j)
                dg[j] = ((vm & (1u << j)) && key_in_pass<MODE>(digit_word(h[j]), ds)) ? key_digit<MODE>(digit_word(h[j]), ds) : P;
#pragma unroll
            for (int j = 0; j < KPT; ++j) atomicAdd(&lh[dg[j]], 1u);
        };
        if (SRC == 0) {
            for (u64 t0 = d.begin; t0 < d.end; t0 += step) {
                vma = tile_keys_reads(packed, inval, t0, d.end, k, ha);
                process(ha, vma);
            }
        } else {
            if (d.begin < d.end) vma = tile_keys_array<W>(keys, d.begin, d.end, ha);
            for (u64 t0 = d.begin; t0 < d.end; t0 += 2 * step) {   // two tiles per trip: register ping-pong, no copies
                if (t0 + step < d.end) vmb = tile_keys_array<W>(keys, t0 + step, d.end, hb);
                process(ha, vma);
                const u64 t1 = t0 + step;
                if (t1 >= d.end) break;
                if (t1 + step < d.end) vma = tile_keys_array<W>(keys, t1 + step, d.end, ha);
                process(hb, vmb);
            }
        }
        __syncthreads();
        for (u32 b = threadIdx.x; b < P; b += SC_NT) matrix[d.flat_base + (u64)b * d.stride] = lh[b];
        __syncthreads();
    }
}

// ------------------------------------------------------------------ scan (3 kernels)
#define SCAN_NT 256
#define SCAN_IPT 16
#define SCAN_BLK (SCAN_NT * SCAN_IPT)

__global__ __launch_bounds__(SCAN_NT) void k_scan_reduce(const u32* __restrict__ a, const u32* __restrict__ d_len,
                                                         u32* __restrict__ sums) {
    __shared__ u32 ws[SCAN_NT / 64];
    const u64 len = *d_len;
    const u64 b0 = (u64)blockIdx.x * SCAN_BLK;
    if (b0 >= len) return;
    u32 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) {
        const u64 i = b0 + threadIdx.x + (u64)j * SCAN_NT;
        if (i < len) s += a[i];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < SCAN_NT / 64; ++i) t += ws[i]; sums[blockIdx.x] = t; }
}

// single block: exclusive scan of the block sums; total -> a[len]
__global__ __launch_bounds__(1024) void k_scan_sums(u32* __restrict__ sums, const u32* __restrict__ d_len,
                                                    u32* __restrict__ a) {
    __shared__ u32 ws[16]; __shared__ u32 carry_s;
    const u64 len = *d_len;
    const u32 nb = (u32)((len + SCAN_BLK - 1) / SCAN_BLK);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 c0 = 0; c0 < nb; c0 += 1024) {
        const u32 i = c0 + threadIdx.x;
        const u32 x = i < nb ? sums[i] : 0u;
        u32 y = x;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(y, d); if (lane >= d) y += t; }
        if (lane == 63) ws[wave] = y;
        __syncthreads();
        if (wave == 0) {
            const u32 wx = lane < 16 ? ws[lane] : 0u; u32 wy = wx;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const u32 t = __shfl_up(wy, d); if (lane >= d) wy += t; }
            if (lane < 16) ws[lane] = wy - wx;
        }
        __syncthreads();
        const u32 carry = carry_s;
        const u32 excl = carry + ws[wave] + y - x;
        if (i < nb) sums[i] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = excl + x;
        __syncthreads();
    }
    if (threadIdx.x == 0) a[len] = carry_s;
}

__global__ __launch_bounds__(SCAN_NT) void k_scan_apply(u32* __restrict__ a, const u32* __restrict__ d_len,
                                                        const u32* __restrict__ sums) {
    __shared__ u32 ws[SCAN_NT / 64];
    const u64 len = *d_len;
    const u64 b0 = (u64)blockIdx.x * SCAN_BLK;
    if (b0 >= len) return;
    // thread owns SCAN_IPT consecutive entries
    const u64 i0 = b0 + (u64)threadIdx.x * SCAN_IPT;
    u32 v[SCAN_IPT]; u32 s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) { v[j] = (i0 + j < len) ? a[i0 + j] : 0u; s += v[j]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 y = s;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(y, d); if (lane >= d) y += t; }
    if (lane == 63) ws[wave] = y;
    __syncthreads();
    u32 wpre = 0;
    for (int w = 0; w < wave; ++w) wpre += ws[w];
    u32 run = sums[blockIdx.x] + wpre + y - s;
#pragma unroll
    for (int j = 0; j < SCAN_IPT; ++j) { if (i0 + j < len) a[i0 + j] = run; run += v[j]; }
}

// ------------------------------------------------------------------ K4: scatter
// Per tile: rank keys inside their bin with one LDS atomic each, scan the tile
// histogram, stage the tile bin-sorted in LDS, then write runs to HBM so that
// consecutive lanes hit consecutive addresses.  Per-chunk cursors live in LDS.
// The next tile's keys are loaded (key array) before the current tile enters
// its LDS phases, so HBM reads stay in flight across the barriers.

// Exclusive scan of cnt[0..P) fused with the cursor bookkeeping of the tile:
//   off[b]   = start of bin b inside the staged tile
//   delta[b] = cur[b] - off[b]   (HBM index of staged element i of bin b is delta[b] + i)
//   cur[b]  += cnt[b];  cnt[b] = 0
// sg.lim (block-owned slices): bin b may only be written below lim[b], the end of the block's slice of that bin; a bin whose keys
// of this tile would not fit is redirected, for this tile, to the dump zone [dump, dump + tile) behind the last slice (never read) -- the
// check costs a few instructions per BIN and tile instead of per key, and nothing is ever written outside the block's own
// slices or the dump zone.  The cursor of such a bin is parked at end + 1, so the overflow shows at the end of the launch.
struct SliceGuard { const u32* lim; u32 dump; u32 uslice, first; };      // lim[b] (LDS): end of the block's slice of bin b; or uniform slices of uslice keys from `first` (no array); neither = no guard
template <int NT>
__device__ __forceinline__ void tile_scan(u32* cnt, u32* off, u32* delta, u32* cur, int P, u32* wsum, u32* tot, SliceGuard sg = SliceGuard{nullptr, 0u, 0u, 0u}) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ipt = (P + NT - 1) / NT;
    const int base = tid * ipt;
    u32 v[4]; u32 s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = base + j;
        v[j] = (j < ipt && idx < P) ? cnt[idx] : 0u;
        s += v[j];
    }
    const u32 inc = wave_incl_scan(s);
    if (lane == 63) wsum[wave] = inc;
    lds_barrier();
    if (wave == 0) {
        const u32 x = lane < NT / 64 ? wsum[lane] : 0u;
        const u32 y = wave_incl_scan(x);
        if (lane < NT / 64) wsum[lane] = y - x;
        if (lane == NT / 64 - 1) { *tot = y; off[P] = y; }        // off[P]: the dummy bin (invalid windows) is staged behind the keys
    }
    lds_barrier();
    u32 run = wsum[wave] + inc - s;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = base + j;
        if (j < ipt && idx < P) {
            const u32 c = cur[idx];
            // (a cursor that left its slice stays at end + 1: it marks the overflow for the end of the launch and cannot wrap 2^32
            //  however many keys the bin still receives; below the end, c + v <= 0xFFFF0000 + a tile)
            const u32 end = sg.lim ? sg.lim[idx] : sg.uslice ? sg.first + (u32)(idx + 1) * sg.uslice : 0xFFFFFFFFu;
            const bool fits = (sg.lim || sg.uslice) ? c + v[j] <= end : true;
            off[idx] = run; delta[idx] = fits ? c - run : sg.dump; cur[idx] = fits ? c + v[j] : end + 1u; cnt[idx] = 0;
            run += v[j];
        }
    }
}

struct OptSpec { u32 cap; u32* subcnt; u32* ovf; const u32* fill; u32 slice, nsl; u64 sstride;     // fill/slice/nsl/sstride: SLICED input (below)
                 u32 F, max_ext; u32* next; u32* ext_cursor; u32* chain_list; u32* chain_cnt;     // region chains (below)
                 u32* work;                                                                         // work counter of the segment hand-out
                 unsigned long long* dbg; };                                                       // DSKGPU_VERBOSE: per-segment (start, end) clock, 100 MHz
// Region chains of the segment-owned level-2 scatter.  Regions 0 .. F-1 are the home regions of the sub-partitions, regions
// F .. F + max_ext - 1 a pool of extension regions of the same size behind them (region r starts at key r * cap of the output
// buffer).  A sub-partition that outgrows the region it is writing -- a k-mer with thousands of occurrences: every repeat family
// of a real genome -- takes the next free extension region(s) (one global atomic per switch) and goes on there; subcnt[r] holds
// the real keys of region r, its top bit says that the list goes on at region next[r]; a sub-partition that leaves its home region
// is appended to chain_list (at most max_ext entries: each takes a pool region), and k_count_chained walks those lists.  Only
// when the pool is used up does *ovf go up (the host then repeats the level with the exact histogram + scan path).
#define CHAIN_BIT 0x80000000u
#define HV_KEYS 4                  // k-mers the level-1 scatter can count apart (HEAVY)
__device__ __forceinline__ bool is_empty_key(u64 h) { return h == DSK_EMPTY; }
template <int W> __device__ __forceinline__ bool is_empty_key(const KN<W>& h) {
    bool e = true;
#pragma unroll
    for (int x = 0; x < W; ++x) e = e && (h.w[x] == DSK_EMPTY);
    return e;
}
template <int W> __device__ __forceinline__ typename KeyT<W>::T empty_key();
template <> __device__ __forceinline__ u64 empty_key<1>() { return DSK_EMPTY; }
template <> __device__ __forceinline__ K2 empty_key<2>() { K2 k; k.w[0] = k.w[1] = DSK_EMPTY; return k; }
template <> __device__ __forceinline__ KN<4> empty_key<4>() { KN<4> k; k.w[0] = k.w[1] = k.w[2] = k.w[3] = DSK_EMPTY; return k; }


// SRC 2: super-k-mer records (superkmer.h) as the key source of the level-1 scatter on the multi-GPU receive side:
// records -> mixed keys straight into the tile registers, no expanded key array in HBM.  A tile takes as many of
// the next Tile<W>::KEYS / 8 records as fit Tile<W>::KEYS keys (a record holds <= SK_MAXN = 32), stages them in the (not yet
// used) key staging area together with a slot map (slot -> record, k-mer index), and every thread then builds its
// KPT keys with a funnel shift + rev_pairs.  Returns the validity mask; *taken = records consumed.
__device__ __forceinline__ u64 sk_key1(const u64* r, int j, int k);
__device__ __forceinline__ K2 sk_key2(const u64* r, int j, int k);
__device__ __forceinline__ void sk_key(const u64* r, int j, int k, u64& out) { out = sk_key1(r, j, k); }
__device__ __forceinline__ void sk_key(const u64* r, int j, int k, K2& out) { out = sk_key2(r, j, k); }
template <int W> __device__ __forceinline__ void sk_key(const u64*, int, int, KN<W>&) {}      // (records carry k <= 64 only)

// candidate records per tile: 3072 for 16384 one-word key slots (more than fit on average -- the tile takes the prefix that
// fits and is ~100 % full; 2048 candidates: 7.4 -> 6.5 ms), 1024 for 8192 two-word slots.
// Thread t looks at candidates t * RPT .. t * RPT + RPT - 1 (requested from HBM a tile ahead: RecPre) and builds the keys of the
// KPT CONSECUTIVE slots t * KPT ..: the candidates that fit are staged densely (zero-length pad records dropped), every record
// that covers a slot q * KPT notes (record, k-mer index) in first[q], and thread q walks on from there -- no per-slot map.
template <int W> struct RecTile { static constexpr int NR = W == 1 ? 3 * SC_NT : SC_NT; static constexpr int RPT = NR / SC_NT;
                                  static constexpr int RS = W == 1 ? 2 : 3; };          // staged words per record (W = 1: k <= 32, two words)
template <int W> struct RecPre { u64 w[RecTile<W>::RPT * RecTile<W>::RS]; };
template <int W>
__device__ __forceinline__ void rec_prefetch(RecPre<W>& pre, const u64* __restrict__ rec, u32 R, u64 r0, u64 rend) {
    constexpr int RPT = RecTile<W>::RPT, RS = RecTile<W>::RS;
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const u64 r = r0 + (u64)threadIdx.x * RPT + u;
        const u64* p = rec + (r < rend ? r : r0) * R;                          // (clamped: unconditional loads; candidates past the end are ignored)
        pre.w[u * RS] = p[0]; pre.w[u * RS + 1] = p[1];
        if (RS == 3) pre.w[u * RS + 2] = (R == 3) ? p[2] : 0ull;
    }
}
template <int W>
__device__ __forceinline__ u32 tile_keys_records(const RecPre<W>& pre, u32 R, u64 r0, u64 rend, int k,
                                                 typename KeyT<W>::T (&h)[Tile<W>::KPT], char* scratch, u32* wsum, u32* taken) {
    constexpr int KPT = Tile<W>::KPT, KEYS = Tile<W>::KEYS, NR = RecTile<W>::NR, RPT = RecTile<W>::RPT, RS = RecTile<W>::RS;
    u64* srec = reinterpret_cast<u64*>(scratch);                             // NR * RS words
    u32* first = reinterpret_cast<u32*>(srec + (size_t)NR * RS);             // KEYS / KPT = SC_NT entries: (staged record << 5) | k-mer index (a record holds <= SK_MAXN = 32)
    u32* info = first + SC_NT;                                               // (records taken << 16) | keys of the tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 left = rend - r0;                                              // candidates that exist
    u32 n[RPT], s = 0;                                                       // s = (records with keys << 18) | keys, over this thread's candidates
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const u32 l = (u32)tid * RPT + u;
        n[u] = l < left ? (u32)(pre.w[u * RS + ((RS == 3 && R == 3) ? 2 : 1)] & 0xFFu) : 0u;
        s += n[u] + (n[u] ? 0x40000u : 0u);
    }
    // (sums: keys <= 32 * NR = 98304 < 2^18, records <= NR = 3072 < 2^14)
    const u32 inc = wave_incl_scan(s);
    if (lane == 63) wsum[wave] = inc;
    lds_barrier();                                                           // ... and the previous tile's write-out is done with the staging area
    u32 run = inc - s;
#pragma unroll
    for (int x = 0; x < SC_NT / 64; ++x) { const u32 v = wsum[x]; if (x < wave) run += v; }
    // records are taken in order while their keys still fit the tile.  Exactly one candidate position is "the first that is not
    // taken" (or none: all NR are), and its thread publishes (taken, keys)
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const u32 l = (u32)tid * RPT + u;
        const u32 start = run & 0x3FFFFu, dense = run >> 18;
        const bool exists = l < left;
        const bool fit = exists && start + n[u] <= (u32)KEYS;
        if (!fit && start <= (u32)KEYS && (exists || l == left)) *info = (l << 16) | start;
        if (fit && n[u]) {
#pragma unroll
            for (int x = 0; x < RS; ++x) srec[dense * RS + x] = pre.w[u * RS + x];
            const u32 bnd = (start + KPT - 1) & ~(u32)(KPT - 1);              // slot groups that begin inside this record
            for (u32 q = bnd; q < start + n[u]; q += KPT) first[q / KPT] = (dense << 5) | (q - start);
        }
        run += n[u] + (n[u] ? 0x40000u : 0u);
        if (u == RPT - 1 && tid == SC_NT - 1 && fit) *info = ((u32)NR << 16) | (run & 0x3FFFFu);
    }
    lds_barrier();
    const u32 inf = *info;
    *taken = inf >> 16;
    const u32 nkeys = inf & 0xFFFFu;
    u32 vm = 0;
    const u32 slot0 = (u32)tid * KPT;
    if (slot0 < nkeys) {
        const u32 e = first[tid];
        u32 rl = e >> 5, jj = e & 31u;
        u64 r[3]; r[2] = 0ull;
#pragma unroll
        for (int x = 0; x < RS; ++x) r[x] = srec[rl * RS + x];
        u32 nn = (u32)(r[(RS == 3 && R == 3) ? 2 : 1] & 0xFFu);
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            if (slot0 + j < nkeys) {
                if (jj == nn) {
                    ++rl; jj = 0;
#pragma unroll
                    for (int x = 0; x < RS; ++x) r[x] = srec[rl * RS + x];
                    nn = (u32)(r[(RS == 3 && R == 3) ? 2 : 1] & 0xFFu);
                }
                sk_key(r, (int)jj, k, h[j]);
                ++jj;
                vm |= 1u << j;
            }
        }
    }
    return vm;
}

// OPT (level 1): "block-owned slices" -- no histogram pass.  Every (block, bin) pair owns one slice of `out`, block-major: the
// slice of bin b of block g is [g * area + boff[b], g * area + boff[b + 1]), so the P write fronts of a block lie within `area`
// keys (44 MB on the bench workload).  A block appends its keys of bin b to its own slice, the write cursors live in LDS for the
// whole launch.  The slices are sized per bin from a positional sample of the level-1 loads scaled to the exact number of valid
// k-mers (k_count_valid), plus 6 % + 160 keys; how much of each slice holds keys goes to fill[b*grid + block], and the level-2
// scatter (SLICED) gathers the grid slices of its bin (stride `area`) and reads exactly that much of each.  A slice that would
// overflow raises *ovf (the host repeats the pass with the exact histogram + scan path).  dump = grid * area = the dump zone.
// boff[b] .. boff[b + 1]: the slice of bin b inside a block's area (P + 1 offsets, sized per bin from the sampled level-1 loads:
// a bin that holds a repeat family simply gets longer slices); area = boff[P] keys per block; dump = grid * area.
// HEAVY: hv_keys[HV_KEYS] (DSK_EMPTY = unused) are counted apart (hv_cnt) instead of being partitioned: a k-mer that alone is a large
// share of a level-1 bin -- poly-A reads: millions of occurrences -- would make its bin's level-2 segment twice the work of the others
// and put a third of that segment's keys on one rank counter and one sub-partition.  A window that holds such a k-mer simply loses
// its validity bit (windows without a key are neither ranked nor staged); the compares are VALU work, which this kernel has to spare.
// A separate instantiation: the plain one keeps its instruction schedule.
// slice_len != 0 (level 0, MODE 4): BIN-major slices -- bin b owns the region of `out` that starts at key obase[b] (64-bit: a group
// of passes holds more than 2^32 keys; positions inside a region stay 32-bit; obase is a device array, copied to LDS), its slices
// are boff[b] keys long (per bin: sized from the sampled load of the pass, so a pass that holds a k-mer with 10^8 occurrences gets
// longer slices), slice (block g, bin b) at g * boff[b] in the region, the dump zone behind the last slice of bin 0 -- and the
// unused tail of every slice is filled with the all-ones sentinel at the end: a bin's region is then ONE key array (with pads
// that tile_keys_array masks), which the pass reads as its input.
struct Opt1Spec { const u32* boff; u32 area, dump; u32* ovf; u32* fill; u32 R; u64* nkeys;      // R: words per super-k-mer record (SRC 2)
                  const u64* hv_keys; unsigned long long* hv_cnt; u32 slice_len; const u64* obase; u64 obase0;      // obase: MODE 4, key offset of every bin's region in `out` (device array; obase0 = obase[0])
                  // anchor[]: never read on a live path.  ONE run-time-dead access with a dynamic index (the sentinel fill at the end of k_scatter) keeps the
                  // compiler from scalarising this whole kernel-argument struct into SGPRs up front: its fields are then loaded from the kernarg segment
                  // where they are used, and the level-1 kernels -- all at the 128-VGPR limit -- keep 8 bytes per lane out of scratch (records source:
                  // 5.15 -> 4.81 ms on the emulated rank of 8; found by bisecting a regression that came with nothing but a change of this struct)
                  u64 anchor[4];
                  u32 uslice;         // != 0: UNIFORM slices of that many keys (bin b at b * uslice, boff unused) -- no slice-end array in LDS: plans above 1634 bins
                  // records (SRC 2) arriving in slices (a multi-GPU step whose exchange overlaps this kernel): one launch per slice over
                  // the chunks [g0, g0 + gn) (cur_state == nullptr: one launch over all chunks), the blocks' write cursors parked in cur_state[block * P + bin] in between;
                  // `resume` picks them up, only the `last` launch reports fill / overflow / keys placed
                  u32* cur_state; u32 g0, gn, resume, last;
                  };
#define L0_MAX_PASSES 16           // passes a level-0 sweep materialises together (bins of the MODE 4 scatter)

template <int W, int SRC, int MODE, bool OPT = false, bool HEAVY = false>
__global__ __launch_bounds__(SC_NT, 4) void k_scatter(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                   const typename KeyT<W>::T* __restrict__ keys,
                                                   const ChunkDesc* __restrict__ descs, const u32* __restrict__ d_nchunks,
                                                   const u32* __restrict__ scanned,
                                                   typename KeyT<W>::T* __restrict__ out, int k, DigitSpec ds, u32 P, Opt1Spec o1) {
    typedef typename KeyT<W>::T Key;
    constexpr int KPT = Tile<W>::KPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Key* stage = reinterpret_cast<Key*>(smem);                       // Tile<W>::KEYS keys
    u32* cnt = reinterpret_cast<u32*>(smem + sizeof(Key) * Tile<W>::KEYS); // P + 1 (dummy bin P: invalid windows)
    u32* off = cnt + (P + 1);                                        // P + 1 (off[P] = valid keys of the tile = where the dummy bin is staged)
    u32* cur = off + (P + 1);                                        // P
    u32* delta = cur + P;                                            // P
    u32* wsum = delta + P;                                           // 16 (+1 total)
    u32* tot = wsum + 16;
    u32* lim = tot + 1;                                              // P (OPT): end of this block's slice of every bin
    u64* lob = MODE == 4 ? reinterpret_cast<u64*>(smem + ((reinterpret_cast<char*>(lim + P) - smem + 7) & ~size_t(7))) : nullptr;      // MODE 4: region base of every bin
    const u32 nchunks = *d_nchunks;
    // OPT: this block's slices are CONTIGUOUS in `out` -- slice of bin b at blockIdx * area + boff[b] -- so its P write fronts
    // stay inside a few 2 MB pages (bin-major, the fronts of one block were P regions of grid * slice keys apart: P pages to
    // cycle through on every tile, far more than the CU's translation cache holds)
    const u32 first = OPT ? blockIdx.x * o1.area : 0u;
    if (OPT && o1.uslice) for (u32 b = threadIdx.x; b < P; b += SC_NT) cur[b] = first + b * o1.uslice;
    else if (OPT && MODE == 4) for (u32 b = threadIdx.x; b < P; b += SC_NT) { const u32 sl = o1.boff[b]; cur[b] = blockIdx.x * sl; lim[b] = cur[b] + sl; lob[b] = o1.obase[b]; }
    else if (OPT) for (u32 b = threadIdx.x; b < P; b += SC_NT) { cur[b] = first + o1.boff[b]; lim[b] = first + o1.boff[b + 1]; }
    u32 gbeg = blockIdx.x, gend = nchunks;
    if constexpr (SRC == 2 && OPT) {
        if (o1.cur_state) { gbeg += o1.g0; gend = o1.g0 + o1.gn; }
        if (o1.resume) for (u32 b = threadIdx.x; b < P; b += SC_NT) cur[b] = o1.cur_state[(u64)blockIdx.x * P + b];
    }
    Key hvk[HV_KEYS]; u32 hc[HV_KEYS]; int nhk = 0;       // (nhk: how many are in use -- the list is dense, a wave-uniform count)
    if constexpr (HEAVY) {
#pragma unroll
        for (int x = 0; x < HV_KEYS; ++x) { hvk[x] = reinterpret_cast<const Key*>(o1.hv_keys)[x]; hc[x] = 0; if (!is_empty_key(hvk[x])) nhk = x + 1; }
    }
    for (u32 g = gbeg; g < gend; g += gridDim.x) {
        const ChunkDesc d = descs[g];
        lds_barrier();   // previous chunk's write-out reads delta/off/stage
        for (u32 b = threadIdx.x; b < P; b += SC_NT) { if (!OPT) cur[b] = scanned[d.flat_base + (u64)b * d.stride]; cnt[b] = 0; }
        if (threadIdx.x == 0) cnt[P] = 0;
        const u64 step = SRC == 0 ? Tile<W>::WORDS : Tile<W>::KEYS;
        Key ha[KPT], hb[KPT]; u32 vma = 0, vmb = 0;
        // a tile in two parts, so that the reads-source loop can put the NEXT tile's loads between them: vmcnt counts loads and
        // stores in issue order, so loads issued after the write-out's stores can only be waited for together with every one
        // of those stores (a full store round trip per tile); issued before them, they are older and the wait leaves the
        // stores in flight -- which needs their number to be known: the write-out is a fixed KPT / 4 trips of 4 predicated stores
        auto rank_and_stage = [&](Key (&h)[KPT], u32 vm) {
            if constexpr (HEAVY && W <= 2) {      // the k-mers counted apart leave the tile here (one- and two-word keys)
#pragma unroll
                for (int x = 0; x < HV_KEYS; ++x) {
                    if (x < nhk) {                   // (uniform: one k-mer counted apart costs one compare per key, not HV_KEYS)
                        // compare, select, or per key -- no exec masking per key (as `valid && equal` with a conditional count it was 7
                        // VALU + 3 SALU instructions per key: +0.26 ms on a kernel that is bound by its instruction count)
                        u32 hits = 0;
#pragma unroll
                        for (int j = 0; j < KPT; ++j) hits |= keys_same(h[j], hvk[x]) ? (1u << j) : 0u;
                        hits &= vm;
                        hc[x] += (u32)__popc(hits);
                        vm &= ~hits;
                    }
                }
            }
            u32 rk[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                u32 dj = key_digit<MODE>(digit_word(h[j]), ds);      // for every window, then a select (as a branch around the multiply it cost exec-mask traffic per key)
                asm volatile("" : "+v"(dj));
                dj = ((vm & (1u << j)) && key_in_pass<MODE>(digit_word(h[j]), ds)) ? dj : P;
                rk[j] = dj << 16;                                    // (digit, rank) packed: rank < 16384, digit <= 2048
            }
            // Windows without a key of this launch (read borders, N, keys of other passes) are neither ranked nor staged: as members
            // of a dummy bin they would all add to ONE counter -- a fifth of a wave's lanes on 150 bp reads, 60 of 64 in a pass of
            // sixteen, serialised on it in every rank instruction (multi-pass level 1: 137 -> 65 ms per pass) -- and masked LDS
            // operations cost no more than full ones
#pragma unroll
            for (int j = 0; j < KPT; ++j) if ((rk[j] >> 16) < P) rk[j] |= atomicAdd(&cnt[rk[j] >> 16], 1u);
            lds_barrier();
            tile_scan<SC_NT>(cnt, off, delta, cur, (int)P, wsum, tot, OPT ? SliceGuard{o1.uslice ? nullptr : lim, o1.dump, o1.uslice, first} : SliceGuard{nullptr, 0u, 0u, 0u});
            lds_barrier();
#pragma unroll
            for (int j = 0; j < KPT; ++j) if ((rk[j] >> 16) < P) stage[off[rk[j] >> 16] + (rk[j] & 0xFFFFu)] = h[j];
            lds_barrier();
        };
        auto write_out = [&]() {
            const u32 ntile = *tot;
#pragma unroll
            for (int it = 0; it < (KPT + 3) / 4; ++it) {
                const u32 i0 = (u32)it * 4 * SC_NT;
                if constexpr (W >= 4) {      // four-word keys: one key at a time (four 32-byte keys in flight spilled 96 bytes per lane to scratch: 22.7 -> see NOTEBOOK.md section 6)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const u32 i = i0 + u * SC_NT + threadIdx.x;
                        const Key kv = stage[i];
                        const u32 dg = key_digit<MODE>(digit_word(kv), ds);
                        const u32 d1 = delta[MODE == 4 ? (dg < P ? dg : 0u) : dg];
                        if (OPT) out[i < ntile ? (u64)(d1 + i) : (u64)(o1.dump + i)] = kv;
                        else if (i < ntile) out[(u64)(d1 + i)] = kv;
                    }
                    continue;
                }
                Key hk[4]; u32 dd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) hk[u] = stage[i0 + u * SC_NT + threadIdx.x];      // (slots past the tile's keys hold stale keys: readable, never stored as keys)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32 dg = key_digit<MODE>(digit_word(hk[u]), ds);
                    dd[u] = delta[MODE == 4 ? (dg < P ? dg : 0u) : dg];      // (MODE 4: the stale keys past the tile's keys may carry any pass)
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const u32 i = i0 + u * SC_NT + threadIdx.x;
                    // (OPT: tile_scan keeps a bin that outgrew its slice out of the other slices.)  With slices there is a dump zone
                    // behind the last bin: a lane past the tile's keys stores there instead of being masked off, so that every trip
                    // issues exactly 4 stores -- the compiler can then count them (see rank_and_stage)
                    if (OPT && MODE == 4) out[i < ntile ? lob[key_digit<MODE>(digit_word(hk[u]), ds) & (L0_MAX_PASSES - 1)] + (u64)(dd[u] + i) : o1.obase0 + (u64)(o1.dump + i)] = hk[u];
                    else if (OPT) out[i < ntile ? (u64)(dd[u] + i) : (u64)(o1.dump + i)] = hk[u];
                    else if (i < ntile) out[(u64)(dd[u] + i)] = hk[u];
                }
            }
            // no barrier here: the next tile's rank phase only touches cnt (zeroed
            // by tile_scan); its first barrier orders this write-out before the
            // next tile_scan / stage writes.
        };
        auto process = [&](Key (&h)[KPT], u32 vm) { rank_and_stage(h, vm); write_out(); };
        if (SRC == 1 && d.begin < d.end) vma = tile_keys_array<W>(keys, d.begin, d.end, ha);
        lds_barrier();
        if constexpr (SRC == 2) {              // records: `packed` is the record array, the chunk range is in records
            if constexpr (W <= 2) {
                RecPre<W> pre;
                if (d.begin < d.end) rec_prefetch<W>(pre, packed, o1.R, d.begin, d.end);
                for (u64 r0 = d.begin; r0 < d.end;) {
                    u32 taken = 0;
                    vma = tile_keys_records<W>(pre, o1.R, r0, d.end, k, ha, smem, wsum, &taken);
                    rank_and_stage(ha, vma);
                    r0 += taken ? taken : 1u;
                    if (r0 < d.end) rec_prefetch<W>(pre, packed, o1.R, r0, d.end);      // the next tile's candidates, requested before this tile's stores
                    write_out();
                }
            }
        } else if constexpr (SRC == 0 && W == 1) {
            // one-word keys from the reads.  Thread -> window mapping: lane l of wave w takes the 16 windows ending in half (w >> 3) of
            // word (w & 7) * 64 + l of the tile, so which half (t0 = 0 or 16) is wave-uniform and, inside either branch below, a
            // compile-time constant: every base is one v_bfe_u32 at a fixed offset instead of a 64-bit shift by a per-lane amount.
            // The four words a thread needs for the NEXT tile are requested before the stores of this one.
            struct Raw { u64 cur, prev; u32 ic, ip; };
            const u32 wlane = threadIdx.x & (SC_NT / 2 - 1);
            const bool upper = threadIdx.x >= SC_NT / 2;                      // wave-uniform
            auto load_raw = [&](u64 t0) {
                const u64 wi = t0 + wlane;
                const u64 wc = wi < d.end ? wi : d.end - 1;                   // clamped: the loads stay unconditional (and countable)
                Raw r; r.cur = packed[wc]; r.prev = packed[wc ? wc - 1 : 0]; r.ic = inval[wc]; r.ip = inval[wc ? wc - 1 : 0];
                if (wc == 0) { r.prev = 0ull; r.ip = 0xFFFFFFFFu; }
                return r;
            };
            if (d.begin < d.end) {
                Raw raw = load_raw(d.begin);
                if (OPT) {          // as many (dump-zone) stores behind the first loads as a tile's write-out issues behind the prefetched ones:
                    // the loop is then entered with the same in-flight picture on both edges and the wait at its top is exact.
                    // (The base is made opaque per chunk: left to itself the compiler hoisted all 16 store addresses out of the CHUNK loop, kept
                    //  them in 32 VGPRs across the whole launch and spilled three of them -- profiles/r05_resource_usage.md.)
                    u64* dz = out + (MODE == 4 ? o1.obase0 : 0ull) + (u64)(o1.dump + threadIdx.x);
                    asm volatile("" : "+v"(dz));
#pragma unroll
                    for (int u = 0; u < 4 * ((KPT + 3) / 4); ++u) dz[u * SC_NT] = (u64)threadIdx.x;
                }
                for (u64 t0 = d.begin; t0 < d.end; t0 += step) {
                    const bool live = t0 + wlane < d.end;
                    if (upper) vma = gen_kmers1_words<16>(raw.cur, raw.prev, raw.ic, raw.ip, 16, k, ha);
                    else vma = gen_kmers1_words<16>(raw.cur, raw.prev, raw.ic, raw.ip, 0, k, ha);
                    if (!live) vma = 0u;
#pragma unroll
                    for (int j = 0; j < 16; ++j) ha[j] = kmix(ha[j]);
                    rank_and_stage(ha, vma);
                    const u64 tn = t0 + step < d.end ? t0 + step : t0;      // (the last tile re-reads its own words: same number of loads every trip)
                    raw = load_raw(tn);
                    write_out();
                }
            }
        } else if constexpr (SRC == 0 && W == 2) {
            // two-word keys from the reads, the same way: lane l of wave w takes the 8 windows ending in quarter (w >> 2) of word
            // (w & 3) * 64 + l of the tile (256 words = 8192 windows), so the quarter -- the window offset t0 = 0 / 8 / 16 / 24 -- is
            // wave-uniform and a compile-time constant inside each branch below (every shift of the generator by a fixed amount, the
            // validity masks scalar), and the six words a thread needs for the NEXT tile are requested before the stores of this one.
            struct Raw2 { u64 cur, p1, p2; u32 ic, i1, i2; };
            const u32 wlane = threadIdx.x & (SC_NT / 4 - 1);
            const u32 quarter = threadIdx.x / (SC_NT / 4);                    // wave-uniform
            auto load_raw = [&](u64 t0) {
                const u64 wi = t0 + wlane;
                const u64 wc = wi < d.end ? wi : d.end - 1;                   // clamped: the loads stay unconditional (and countable)
                Raw2 r; r.cur = packed[wc]; r.p1 = packed[wc >= 1 ? wc - 1 : 0]; r.p2 = packed[wc >= 2 ? wc - 2 : 0];
                r.ic = inval[wc]; r.i1 = inval[wc >= 1 ? wc - 1 : 0]; r.i2 = inval[wc >= 2 ? wc - 2 : 0];
                if (wc < 1) { r.p1 = 0ull; r.i1 = 0xFFFFFFFFu; }
                if (wc < 2) { r.p2 = 0ull; r.i2 = 0xFFFFFFFFu; }
                return r;
            };
            if (d.begin < d.end) {
                Raw2 raw = load_raw(d.begin);
                if (OPT) {          // (see the one-word loop: the same in-flight picture on both edges of the loop)
                    Key fillk; fillk.w[0] = fillk.w[1] = (u64)threadIdx.x;
#pragma unroll
                    for (int u = 0; u < 4 * ((KPT + 3) / 4); ++u) out[(u64)(o1.dump + u * SC_NT + threadIdx.x)] = fillk;
                }
                for (u64 t0 = d.begin; t0 < d.end; t0 += step) {
                    const bool live = t0 + wlane < d.end;
                    if (quarter == 0) vma = gen_kmers2_words<8>(raw.cur, raw.p1, raw.p2, raw.ic, raw.i1, raw.i2, 0, k, ha);
                    else if (quarter == 1) vma = gen_kmers2_words<8>(raw.cur, raw.p1, raw.p2, raw.ic, raw.i1, raw.i2, 8, k, ha);
                    else if (quarter == 2) vma = gen_kmers2_words<8>(raw.cur, raw.p1, raw.p2, raw.ic, raw.i1, raw.i2, 16, k, ha);
                    else vma = gen_kmers2_words<8>(raw.cur, raw.p1, raw.p2, raw.ic, raw.i1, raw.i2, 24, k, ha);
                    if (!live) vma = 0u;
#pragma unroll
                    for (int j = 0; j < 8; ++j) kmixN(ha[j]);
                    rank_and_stage(ha, vma);
                    const u64 tn = t0 + step < d.end ? t0 + step : t0;      // (the last tile re-reads its own words: same number of loads every trip)
                    raw = load_raw(tn);
                    write_out();
                }
            }
        } else if constexpr (SRC == 0) {
            for (u64 t0 = d.begin; t0 < d.end; t0 += step) {
                vma = tile_keys_reads(packed, inval, t0, d.end, k, ha);
                process(ha, vma);
            }
        } else {
            for (u64 t0 = d.begin; t0 < d.end; t0 += 2 * step) {   // two tiles per trip: register ping-pong, no copies
                if (t0 + step < d.end) vmb = tile_keys_array<W>(keys, t0 + step, d.end, hb);   // prefetch under the LDS phases
                process(ha, vma);
                const u64 t1 = t0 + step;
                if (t1 >= d.end) break;
                if (t1 + step < d.end) vma = tile_keys_array<W>(keys, t1 + step, d.end, ha);
                process(hb, vmb);
            }
        }
    }
    if (OPT) {      // how much of each of its slices this block filled; report a slice that was outgrown
        lds_barrier();
        if constexpr (HEAVY) {      // the occurrences of the k-mers counted apart (every launch of a sliced receive reports its own; k_heavy_rows adds
#pragma unroll                      //  them to the keys the step placed)
            for (int x = 0; x < HV_KEYS; ++x) {
                u32 v = hc[x];
#pragma unroll
                for (int dd = 32; dd >= 1; dd >>= 1) v += __shfl_down(v, dd);
                if ((threadIdx.x & 63) == 0 && v) atomicAdd(&o1.hv_cnt[x], (unsigned long long)v);
            }
        }
        if constexpr (SRC == 2) {
            if (o1.cur_state) {
                for (u32 b = threadIdx.x; b < P; b += SC_NT) o1.cur_state[(u64)blockIdx.x * P + b] = cur[b];
                if (!o1.last) return;
            }
        }
        bool ovf = false;
        u32 mine = 0;
        for (u32 b = threadIdx.x; b < P; b += SC_NT) {
            const u32 beg = MODE == 4 ? blockIdx.x * o1.boff[b] : first + (o1.uslice ? b * o1.uslice : o1.boff[b]), c = cur[b], end = o1.uslice ? beg + o1.uslice : lim[b];
            if (c > end) ovf = true;
            const u32 f = c > end ? end - beg : c - beg;
            o1.fill[(u64)b * gridDim.x + blockIdx.x] = f;
            mine += f;
        }
        if constexpr (W == 1) {
            if (o1.slice_len) {                  // level 0: the unused tail of every slice of this block becomes sentinel keys
                lds_barrier();
                for (u32 b = 0; b < P; ++b) {
                    const u32 c = cur[b] < lim[b] ? cur[b] : lim[b];
                    for (u32 i = c + threadIdx.x; i < lim[b]; i += SC_NT) out[(MODE == 4 ? lob[b & (L0_MAX_PASSES - 1)] : o1.anchor[b & 3]) + i] = DSK_EMPTY;      // (slice_len != 0 only with MODE 4: see Opt1Spec::anchor)
                }
            }
        }
        if (ovf) *o1.ovf = 1u;
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) mine += __shfl_down(mine, dd);
        if ((threadIdx.x & 63) == 0 && mine) atomicAdd(o1.nkeys, (u64)mine);      // keys this launch placed (all of them unless a slice overflowed)
    }
}

// ------------------------------------------------------------------ level 0 of a multi-pass count (one-word keys)
// One sweep over the 2-bit reads: every k-mer whose pass lies in [ds.pass, ds.pass + G) is appended to the slice that this block
// owns inside that pass's region of `out` (the region starts at key obase[pass - ds.pass], block g's slice at g * slen[..] in it).
// With <= L0_MAX_PASSES bins and one contiguous slice per (block, bin) there is nothing to stage: a key takes its position from
// an LDS cursor (one returning atomic; the lanes of a wave that hit the same bin get consecutive positions) and leaves straight
// from the register it was built in -- no tile in LDS, no tile scan, no barrier inside the loop.  A sweep keeps G of npass
// passes, a sixth of the windows on a 90 Gbp input: staged through a 16384-slot tile (k_scatter<1, 0, 4>, the first version) it
// paid for every slot of every tile -- 2.8 ps per window, as much as a full level 1; this way it is bound by the k-mer
// generation itself.  The stores of a (block, bin) pair fall into a window of a few KB that moves on tile by tile: L2 merges them
// into whole lines.  The unused tail of every slice is filled with the all-ones sentinel at the end, so a region is ONE key array
// (tile_keys_array masks the pads).  A slice that would overflow raises *ovf (nothing is written past a slice): the passes of
// the group then re-generate their keys from the reads.  Key order inside a slice follows the atomics, i.e. it is not the same
// from run to run -- counts and (sorted) rows are.
__global__ __launch_bounds__(SC_NT) void k_level0(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                   const ChunkDesc* __restrict__ descs, const u32* __restrict__ d_nchunks,
                                                   u64* __restrict__ out, int k, DigitSpec ds, u32 G, const u32* __restrict__ slen, const u64* __restrict__ obase,
                                                   u32* __restrict__ ovf) {
    __shared__ u32 cur[L0_MAX_PASSES], lim[L0_MAX_PASSES];
    __shared__ u64 lob[L0_MAX_PASSES];
    if (threadIdx.x < L0_MAX_PASSES) {
        const u32 b = threadIdx.x < G ? threadIdx.x : G - 1;
        const u32 sl = slen[b];
        cur[threadIdx.x] = blockIdx.x * sl; lim[threadIdx.x] = blockIdx.x * sl + sl; lob[threadIdx.x] = obase[b];
    }
    __syncthreads();
    const u32 nchunks = *d_nchunks;
    struct Raw { u64 cur, prev; u32 ic, ip; };
    const u32 wlane = threadIdx.x & (SC_NT / 2 - 1);
    const bool upper = threadIdx.x >= SC_NT / 2;                      // wave-uniform: which half of its word a thread's 16 windows end in
    bool over = false;
    for (u32 g = blockIdx.x; g < nchunks; g += gridDim.x) {
        const ChunkDesc d = descs[g];
        if (d.begin >= d.end) continue;
        auto load_raw = [&](u64 t0) {
            const u64 wi = t0 + wlane;
            const u64 wc = wi < d.end ? wi : d.end - 1;                   // clamped: the loads stay unconditional
            Raw r; r.cur = packed[wc]; r.prev = packed[wc ? wc - 1 : 0]; r.ic = inval[wc]; r.ip = inval[wc ? wc - 1 : 0];
            if (wc == 0) { r.prev = 0ull; r.ip = 0xFFFFFFFFu; }
            return r;
        };
        Raw raw = load_raw(d.begin);
        for (u64 t0 = d.begin; t0 < d.end; t0 += Tile<1>::WORDS) {
            const bool live = t0 + wlane < d.end;
            u64 h[16]; u32 vm;
            if (upper) vm = gen_kmers1_words<16>(raw.cur, raw.prev, raw.ic, raw.ip, 16, k, h);
            else vm = gen_kmers1_words<16>(raw.cur, raw.prev, raw.ic, raw.ip, 0, k, h);
            if (!live) vm = 0u;
            const u64 tn = t0 + Tile<1>::WORDS < d.end ? t0 + Tile<1>::WORDS : t0;
            raw = load_raw(tn);                                          // the next tile's words fly under this tile's work
            u32 bin[16], pos[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                h[j] = kmix(h[j]);
                const u32 p = key_pass(h[j], ds) - ds.pass;
                bin[j] = ((vm >> j) & 1u) && p < G ? p : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) pos[j] = bin[j] != 0xFFFFFFFFu ? atomicAdd(&cur[bin[j]], 1u) : 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (bin[j] != 0xFFFFFFFFu) {
                    if (pos[j] < lim[bin[j]]) out[lob[bin[j]] + pos[j]] = h[j]; else over = true;
                }
        }
    }
    __syncthreads();
    if (over) *ovf = 1u;
    for (u32 b = 0; b < G; ++b) {                                      // sentinel keys behind what this block wrote into its slices
        const u32 c = cur[b] < lim[b] ? cur[b] : lim[b];
        for (u32 i = c + threadIdx.x; i < lim[b]; i += SC_NT) out[lob[b] + i] = DSK_EMPTY;
    }
}

// number of valid k-mer windows of the encoded stream (any k <= 128): sizes the slices of the OPT level-1 scatter
__global__ __launch_bounds__(256) void k_count_valid(const u32* __restrict__ inval, u64 nwords, int k, u64* __restrict__ total) {
    __shared__ u32 ws[4];
    const u64 stride = (u64)gridDim.x * 256;
    u32 c = 0;
    u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k <= 32) {
        // four words per trip, their eight loads in flight together (one word per trip left the kernel waiting for memory: 0.081 ms
        // for 0.19 GB)
        for (; w + 3 * stride < nwords; w += 4 * stride) {
            u32 ic4[4], ip4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const u64 x = w + u * stride; ic4[u] = inval[x]; ip4[u] = x ? inval[x - 1] : 0xFFFFFFFFu; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                u64 bad = ((u64)ip4[u] << 32) | ic4[u];
                int rem = k - 1;
#pragma unroll
                for (int st = 1; st <= 16; st <<= 1) { const int sh = rem < st ? rem : st; bad |= bad >> sh; rem -= sh; }
                c += 32u - (u32)__popc((u32)bad);
            }
        }
    }
    if (k > 32 && k <= 64) {
        // the same smear over a frame of 96 bases (word w - 2 : word w - 1 : word w; as superkmer.h's sk_tile), four words per trip, their
        // twelve loads independent (r05: the walk-back below waits for one load after the other -- 0.13 ms at k = 63 against 0.06)
        for (; w + 3 * stride < nwords; w += 4 * stride) {
            u32 ic4[4], i14[4], i24[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u64 x = w + u * stride;
                ic4[u] = inval[x]; i14[u] = x >= 1 ? inval[x - 1] : 0xFFFFFFFFu; i24[u] = x >= 2 ? inval[x - 2] : 0xFFFFFFFFu;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                u64 bad_lo = ((u64)i14[u] << 32) | ic4[u], bad_hi = i24[u];
                int rem = k - 1;
#pragma unroll
                for (int st = 1; st <= 32; st <<= 1) {
                    const int sh = rem < st ? rem : st;
                    if (sh) { bad_lo |= (bad_lo >> sh) | (bad_hi << (64 - sh)); bad_hi |= bad_hi >> sh; }
                    rem -= sh;
                }
                c += 32u - (u32)__popc((u32)bad_lo);
            }
        }
    }
    for (; w < nwords; w += stride) {
        const u32 ic = inval[w];
        if (k <= 32) {
            // frame of 64 bases (previous word : this word), bit (63 - i) <-> base i.  A window ending at base e is bad if
            // an invalid base lies in [e-k+1, e]: smear every invalid bit over the k-1 following bases (log-step ORs)
            u64 bad = ((u64)(w ? inval[w - 1] : 0xFFFFFFFFu) << 32) | ic;
            int rem = k - 1;                                // bases still to cover: steps of 1, 2, 4, 8, 16 (or what is left), as in gen_kmers1
#pragma unroll
            for (int st = 1; st <= 16; st <<= 1) { const int sh = rem < st ? rem : st; bad |= bad >> sh; rem -= sh; }
            c += 32u - (u32)__popc((u32)bad);
        } else {
            // longer windows: run = valid bases that end just before this word (walk back over whole valid words), then roll
            int run = 0;
            for (int q = 1; q <= 4 && run < k; ++q) {
                if (w < (u64)q) break;
                const u32 iv = inval[w - q];
                if (iv == 0) run += 32; else { run += __builtin_ctz(iv); break; }
            }
            // k > 32: behind an invalid base of THIS word no window of the word can be whole again, so only the bases before the first
            // invalid one (f of them) can end a valid window -- base t does if run + t + 1 >= k
            const int f = ic ? __clz((int)ic) : 32;
            const int need = k - 1 - run;
            const int good = f - (need > 0 ? need : 0);
            c += good > 0 ? (u32)good : 0u;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_down(c, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(total, (u64)ws[0] + ws[1] + ws[2] + ws[3]);
}

// ------------------------------------------------------------------ K4b: scatter with aligned write-out
// Key-array source only.  Same ranking / staging as k_scatter, but the write-out emits whole
// 64-byte-aligned groups of G keys: per bin a carry of up to G-1 keys waits in LDS until the group
// it belongs to is complete, so HBM/L2 see aligned 64-byte (often 128-byte) writes instead of
// unaligned ~170-byte runs (measured on the level-2 scatter: 6.6 ms -> 3.8 ms for 128-byte lines,
// 4.7 ms for 64-byte groups, emulated; the level-1 scatter from reads is ALU-bound and does not gain).
//   pos[b]  global index (in `out`) of bin b's first not-yet-written key
//   rn[b]   keys of bin b waiting in carry[b][0..rn)          (next key goes to pos + rn)
// per tile and bin: t = rn + c new keys; everything below the last group boundary is emitted:
//   bound = (pos + t) & ~(G-1);  e = bound > pos ? bound - pos : 0;  rn' = t - e;  pos' = pos + e
// Write-out is bin-centric: a lane group of G lanes owns a bin for the tile, writes its complete
// groups and then refreshes its carry, so no barrier separates the two.
#ifndef AL_G2
#define AL_G2 4              // two-word keys: keys per aligned group of the level-2 write-out (4 = 64 bytes; 8 = 128 bytes: experiments)
#define AL_KPT2 6            //                keys per thread and tile
#endif
#ifndef AL_G1
#define AL_G1 8              // one-word keys: the same (8 = 64 bytes; 16 = 128 bytes: experiments)
#define AL_KPT1 12
#endif
template <int W> struct ATile {
    static constexpr int G = W == 2 ? AL_G2 : W == 1 ? AL_G1 : 8 / W;      // keys per aligned group (64 bytes)
    static constexpr int CARRY = G - 1;
    static constexpr int KPT = W == 2 ? AL_KPT2 : W == 1 ? AL_KPT1 : 12 / W; // 12288 one-word / 6144 two-word keys per tile = 96 KB
    static constexpr int KEYS = SC_NT * KPT;
};
// A bin that receives more than AL_BIG_KEYS keys in ONE tile (a k-mer with tens of thousands of occurrences: 2 % of a tile) is
// written out by the whole block instead of by its lane group alone -- 300 keys are 38 trips of one group of 8 lanes while the
// other 1016 threads wait at the next barrier (0.7 ms of 4.4 on a repeat-rich genome; with poly-A reads, 20 ms).
#define AL_BIG 64                   // most such bins per tile (a tile holds 12288 keys: at most 191 could exist, the rest stay with their groups)
#define AL_BIG_KEYS 64u
#ifndef SLICED_MAX
#define SLICED_MAX 320      // most level-1 slices per bin (= blocks of the level-1 launch: 256 CUs x 1) the level-2 loader can walk
#endif
__host__ __device__ inline size_t ascatter_lds(int W, u32 P) {
    const size_t key = 8 * (size_t)W, keys = (size_t)SC_NT * (W == 2 ? AL_KPT2 : W == 1 ? AL_KPT1 : 12 / W), G = W == 2 ? AL_G2 : W == 1 ? AL_G1 : 8 / W;
    return keys * key + (size_t)P * (G - 1) * key + (size_t)(P + 1) * 4 + (size_t)P * 4 + (size_t)(P + 1) * 12 + (size_t)P * 2 + 20 * 4 + 32
           + SLICED_MAX * 4 + 16      // + prefix sums of the slice fills (SLICED input)
           + 2 * AL_BIG * 2 + 16;     // + the lists (one per tile parity) of bins with a long run in the tile (written by the whole block)
}

// OPT (one-word keys, level 2): "segment-owned" variant that needs NO histogram pass.  A chunk is a whole
// segment (level-1 bin), processed by one block from start to end, and every sub-bin q = s*P + b owns the
// fixed region [q*cap, (q+1)*cap) of `out`; pos[] starts at the region bases and simply advances, the carry
// lives through the whole segment, and at its end the partial groups are padded with the DSK_EMPTY sentinel
// (never read: subcnt[q] = number of real keys of the region).  A sub-bin that would
// outgrow its region raises *ovf (writes wrap to the region start: the result is discarded and the host
// repeats the level with the exact histogram + scan path).  flat_base of a chunk = s*P.
// SLICED (with OPT): the input segment is a level-1 bin written as block-owned slices -- nsl slices of `slice` keys,
// os.sstride keys apart (slice i at keys + d.begin + i * sstride), of which the first fill[s*nsl + i] hold keys.  The loader walks the slices in order and skips their
// unused tails: a thread's keys of consecutive tiles are monotone in the logical stream, so it only keeps the bounds
// of its current slice in registers and touches the LDS prefix array when it crosses into the next slice.
template <int W, int MODE, bool OPT = false, bool SLICED = false>
__global__ __launch_bounds__(SC_NT, 4) void k_scatter_al(const typename KeyT<W>::T* __restrict__ keys,
                                                         const ChunkDesc* __restrict__ descs, const u32* __restrict__ d_nchunks,
                                                         const u32* __restrict__ scanned,
                                                         typename KeyT<W>::T* __restrict__ out_all, DigitSpec ds, u32 P, OptSpec os) {
    typedef typename KeyT<W>::T Key;
    typedef unsigned short u16;
    constexpr int KPT = ATile<W>::KPT, G = ATile<W>::G, CARRY = ATile<W>::CARRY, TKEYS = ATile<W>::KEYS;
    constexpr int NGRP = SC_NT / G;                                   // lane groups per block
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Key* stage = reinterpret_cast<Key*>(smem);
    Key* carry = stage + TKEYS;                                       // [P][CARRY]
    u32* cnt = reinterpret_cast<u32*>(carry + (size_t)P * CARRY);     // P + 1 (dummy bin P)
    u32* pos = cnt + (P + 1);                                         // P   (persistent across tiles)
    // per-bin record of the current tile, read with one LDS access by the write-out:
    //   x = pos before this tile   y = emitted (lo16) | carry fill before (hi16)   z = offset of the bin in the staged tile
    uint3* rec = reinterpret_cast<uint3*>(pos + P);                   // P + 1 (rec[P].z only)
    u16* rn = reinterpret_cast<u16*>(rec + (P + 1));                  // P   carry fill after this tile (persistent)
    u32* wsum = reinterpret_cast<u32*>(smem + ((reinterpret_cast<char*>(rn + P) - smem + 3) & ~size_t(3)));
    u32* pre = wsum + 20;                                             // SLICED: nsl + 1 prefix sums of the slice fills
    u16* big = reinterpret_cast<u16*>(pre + SLICED_MAX + 4);          // bins with a long run in this tile; nbig[parity] = how many (wsum[17], wsum[18])
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 gi = tid / G, gl = tid % G;
    const u32 nchunks = *d_nchunks;
    const int ipt = (int)((P + SC_NT - 1) / SC_NT);
    // OPT: the segments are handed out by a work counter, in the order of the descriptors (the host lists the heaviest first): a
    // segment that holds a repeat family takes its block longer, and the block then simply takes fewer -- a static round-robin left
    // the block of a poly-A segment 0.7 ms behind the rest.  The first round needs no atomic (block b takes descriptor b).
    u32 g = blockIdx.x;
    for (;; ) {
        if (g >= nchunks) break;
        const ChunkDesc d = descs[g];
        if (OPT && os.dbg && tid == 0) { os.dbg[3 * g] = wall_clock64(); os.dbg[3 * g + 2] = blockIdx.x; }
        // OPT: positions are relative to the segment's first region (keeps them 32-bit whatever the total)
        Key* out = OPT ? out_all + (u64)d.flat_base * os.cap : out_all;
        Key* extb = OPT ? out_all + (u64)os.F * os.cap : out_all;         // first extension region
        bool ovf = false;
        lds_barrier();
        for (u32 b = tid; b < P; b += SC_NT) { pos[b] = OPT ? b * os.cap : scanned[d.flat_base + (u64)b * d.stride]; cnt[b] = 0; rn[b] = 0; }
        if (tid == 0) { cnt[P] = 0; wsum[17] = 0; wsum[18] = 0; }
        u32 par = 0;                                                  // tile parity: which of the two list counters this tile fills
        u64 lbeg = d.begin, lend = d.end;                             // range of the (logical) key stream of this chunk
        u32 sg = 0, slo = 0, shi = 0;                                 // SLICED: this thread's current slice and its logical bounds
        if (SLICED) {
            if (wave == 0) {                                          // exclusive prefix of the slice fills (one wave, nsl <= SLICED_MAX)
                const u32* f = os.fill + (u64)(d.flat_base / P) * os.nsl;
                u32 run = 0;
                for (u32 i0 = 0; i0 < os.nsl; i0 += 64) {
                    const u32 i = i0 + lane;
                    const u32 v = i < os.nsl ? f[i] : 0u;
                    u32 inc = v;
#pragma unroll
                    for (int dd = 1; dd < 64; dd <<= 1) { const u32 t = __shfl_up(inc, dd); if (lane >= dd) inc += t; }
                    if (i < os.nsl) pre[i] = run + inc - v;
                    run += __shfl(inc, 63);
                }
                if (lane == 0) pre[os.nsl] = run;
            }
            lds_barrier();
            lbeg = 0; lend = pre[os.nsl];
            shi = pre[1];
        }
        Key h[KPT]; u32 vm = 0;       // one register set: the next tile is loaded as soon as the stage writes have consumed this one
        auto load = [&](u64 k0, Key (&hh)[KPT]) -> u32 {
            const Key* base = keys + (SLICED ? d.begin : k0);
            const u64 left = lend - k0;
            const u32 n = left < (u64)TKEYS ? (u32)left : (u32)TKEYS;
            u32 m = 0;
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const u32 o = tid + (u32)j * SC_NT;
                const bool ok = o < n;
                if (!SLICED) hh[j] = base[ok ? o : n - 1];
                else {
                    const u32 i = (u32)k0 + (ok ? o : n - 1);                 // logical index, monotone over j and over tiles for ok lanes
                    while (i >= shi && sg + 1 < os.nsl) { ++sg; slo = shi; shi = pre[sg + 1]; }
                    const u32 off = i >= slo ? i - slo : 0u;                  // (a clamped, not-ok lane may point below its slice: any valid address will do)
                    hh[j] = base[(u64)sg * os.sstride + off];
                }
                m |= (ok ? 1u : 0u) << j;
            }
            return m;
        };
        auto process = [&](u64 tnext) {
            u32 rk[KPT];
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                const bool pad = (OPT && is_empty_key(h[j])) || is_pad_key(h[j]);      // sentinel of a level-1 slice tail / of a level-0 key array
                u32 dj = key_digit<MODE>(digit_word(h[j]), ds);             // unconditional + select: no exec-mask traffic around the multiply
                asm volatile("" : "+v"(dj));
                dj = ((vm & (1u << j)) && !pad && key_in_pass<MODE>(digit_word(h[j]), ds)) ? dj : P;
                rk[j] = dj << 16;
            }
#pragma unroll
            for (int j = 0; j < KPT; ++j) rk[j] |= atomicAdd(&cnt[rk[j] >> 16], 1u);
            lds_barrier();
            // ---- scan of the tile histogram fused with the carry bookkeeping
            if (tid == 0) {
                cnt[P] = 0;               // the dummy bin's ranks are in registers now: its counter starts every tile at 0 (masked slots are
                                          // staged at rec[P].z + rank, so a rank that grew over the tiles of a chunk would leave the staging area)
                wsum[17 + (par ^ 1u)] = 0;  // the previous tile's list of long runs has been read by everyone (they are past this tile's first barrier)
            }
            {
                const int base = tid * ipt;
                u32 c[4], pe[4], er[4], s = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int b = base + j;
                    c[j] = 0; pe[j] = 0; er[j] = 0;
                    if (j < ipt && b < (int)P) {
                        c[j] = cnt[b];
                        const u32 r = rn[b]; u32 p = pos[b];
                        // OPT: bit 31 of a position = "inside the extension pool" (offset from its start), clear = offset from the segment's
                        // first home region.  A position always lies strictly inside its region: a region is left as soon as the next
                        // groups would reach its end, so the region of a position is its offset / cap.
                        const u32 pl = OPT ? p & ~CHAIN_BIT : p;
                        const u32 bound = (pl + r + c[j]) & ~(u32)(G - 1);
                        const u32 e = bound > pl ? bound - pl : 0u;
                        if (OPT) {
                            const bool ext = (p >> 31) != 0u;
                            const u32 rs = ext ? pl / os.cap * os.cap : (u32)b * os.cap;           // start of the region being written
                            if (pl + e >= rs + os.cap) {                                            // the groups of this tile would reach its end
                                const u32 need = e / os.cap + 1u;                                   // regions for e keys, at least one key of room left
                                const u32 x = atomicAdd(os.ext_cursor, need);
                                if (x + need > os.max_ext) { ovf = true; p = (u32)b * os.cap; }     // pool used up: wrap (the result is discarded)
                                else {
                                    const u32 cur = ext ? os.F + pl / os.cap : d.flat_base + (u32)b;
                                    os.subcnt[cur] = CHAIN_BIT | (pl - rs); os.next[cur] = os.F + x;
                                    if (!ext) os.chain_list[atomicAdd(os.chain_cnt, 1u)] = cur;      // (< max_ext entries: every one holds a pool region)
                                    for (u32 i = 0; i + 1 < need; ++i) { os.subcnt[os.F + x + i] = CHAIN_BIT | os.cap; os.next[os.F + x + i] = os.F + x + i + 1u; }
                                    p = CHAIN_BIT | (x * os.cap);
                                }
                            }
                        }
                        pe[j] = p; er[j] = e | (r << 16);
                        rn[b] = (u16)(r + c[j] - e);
                        pos[b] = p + e; cnt[b] = 0;
                        if (e > AL_BIG_KEYS) { const u32 x = atomicAdd(&wsum[17 + par], 1u); if (x < AL_BIG) big[par * AL_BIG + x] = (u16)b; else er[j] |= 0x8000u; }   // (list full: bit 15 of e = "the group writes it all")
                    }
                    s += c[j];
                }
                const u32 inc = wave_incl_scan(s);
                if (lane == 63) wsum[wave] = inc;
                lds_barrier();
                if (wave == 0) {
                    const u32 x = lane < SC_NT / 64 ? wsum[lane] : 0u;
                    const u32 y = wave_incl_scan(x);
                    if (lane < SC_NT / 64) wsum[lane] = y - x;
                    if (lane == SC_NT / 64 - 1) rec[P].z = y;              // the dummy bin (masked slots) is staged behind the keys
                }
                lds_barrier();
                u32 run = wsum[wave] + inc - s;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int b = base + j;
                    if (j < ipt && b < (int)P) { rec[b] = make_uint3(pe[j], er[j], run); run += c[j]; }
                }
            }
            lds_barrier();
            {   // all bin offsets first, then all stage writes (no dependent LDS read -> write round trip per key, no branches); masked
                // slots (dummy bin P) are staged like any bin, behind the keys (rec[P].z = their number), and never written out
                u32 so[KPT];
#pragma unroll
                for (int j = 0; j < KPT; ++j) so[j] = rec[rk[j] >> 16].z;
#pragma unroll
                for (int j = 0; j < KPT; ++j) stage[so[j] + (rk[j] & 0xFFFFu)] = h[j];
            }
            if (tnext < lend) vm = load(tnext, h);        // HBM reads of the next tile fly under the write-out phase
            lds_barrier();
            // ---- long runs first, by the whole block: consecutive threads, consecutive keys (the first group of such a bin -- it holds
            // the carried keys -- and the carry refresh stay with the bin's lane group below)
            {
                const u32 nb = wsum[17 + par] < (u32)AL_BIG ? wsum[17 + par] : (u32)AL_BIG;
                for (u32 x = 0; x < nb; ++x) {
                    const uint3 rbig = rec[big[par * AL_BIG + x]];
                    const u32 e = rbig.y & 0x7FFFu, r = rbig.y >> 16, o = rbig.z;
                    const u32 pl = OPT ? rbig.x & ~CHAIN_BIT : rbig.x;
                    Key* ob = (OPT && (rbig.x >> 31)) ? extb : out;
                    for (u32 idx = G + tid; idx < e; idx += SC_NT) ob[pl + idx] = stage[o + idx - r];
                }
            }
            // ---- write-out + carry refresh, one lane group per bin
            for (u32 b0 = 0; b0 < P; b0 += 2 * NGRP) {
                uint3 rb[2]; u32 rnew[2]; bool act[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const u32 b = b0 + u * NGRP + gi;
                    act[u] = b < P;
                    rb[u] = rec[act[u] ? b : 0]; rnew[u] = rn[act[u] ? b : 0];
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!act[u]) continue;
                    const u32 b = b0 + u * NGRP + gi;
                    const u32 e = rb[u].y & 0x7FFFu, r = rb[u].y >> 16, o = rb[u].z;
                    const u32 mine = (e > AL_BIG_KEYS && !(rb[u].y & 0x8000u)) ? (u32)G : e;   // a listed long run: only its first group (the block wrote the rest)
                    const u32 pold = OPT ? rb[u].x & ~CHAIN_BIT : rb[u].x;
                    Key* ob = (OPT && (rb[u].x >> 31)) ? extb : out;                           // home regions of the segment or the extension pool
                    Key* cb = carry + (size_t)b * CARRY;
                    for (u32 a = (pold & ~(u32)(G - 1)) + gl; a < pold + mine; a += G) {   // complete groups of this bin
                        if (a >= pold) {
                            const u32 idx = a - pold;
                            const Key kv = idx < r ? cb[idx] : stage[o + idx - r];
                            ob[a] = kv;
                        }
                    }
                    const u32 c = rnew[u] + e - r;                        // keys this tile gave the bin
                    if (e) { if (gl < rnew[u]) cb[gl] = stage[o + c - rnew[u] + gl]; }
                    else if (gl < c) cb[r + gl] = stage[o + gl];
                }
            }
            // no barrier: the next tile's rank phase only touches cnt; its barriers order the rest
            par ^= 1u;
        };
        if (lbeg < lend) vm = load(lbeg, h);
        lds_barrier();
        for (u64 t0 = lbeg; t0 < lend; t0 += (u64)TKEYS) process(t0 + TKEYS);
        // ---- end of chunk: flush what is left in the carries (one partial group per bin)
        lds_barrier();
        for (u32 b = gi; b < P; b += NGRP) {
            const u32 r = rn[b];
            if (!OPT) { if (gl < r) out[pos[b] + gl] = carry[(size_t)b * CARRY + gl]; }
            else {
                // (a position lies strictly inside its region and is a multiple of G: the last partial group always has room)
                const u32 p = pos[b], pl = p & ~CHAIN_BIT;
                const bool ext = (p >> 31) != 0u;
                const u32 rs = ext ? pl / os.cap * os.cap : b * os.cap;
                Key* ob = ext ? extb : out;
                if (r) ob[pl + gl] = gl < r ? carry[(size_t)b * CARRY + gl] : empty_key<W>();       // pad the last group
                if (gl == 0) os.subcnt[ext ? os.F + pl / os.cap : d.flat_base + b] = pl - rs + r;    // real keys of the list's last region: the pads are never read
            }
        }
        if (OPT && ovf) *os.ovf = 1u;
        if (OPT && os.dbg && tid == 0) os.dbg[3 * g + 1] = wall_clock64();
        if (OPT && os.work) {          // next segment: from the work counter (starts at the grid size)
            lds_barrier();
            if (tid == 0) wsum[19] = atomicAdd(os.work, 1u);
            lds_barrier();
            g = wsum[19];
        } else g += gridDim.x;
    }
}

// ------------------------------------------------------------------ plan of the next level
// Segments of the next level = bins of this level.  seg s spans
// [src[s*sstride], src[(s+1)*sstride]) of the key array.  One block.
struct SegInfo { u32 start, nch, chunk_base, pad; };

__global__ __launch_bounds__(1024) void k_plan(const u32* __restrict__ src, u32 sstride, u32 S, u32 CH, u32 P,
                                               SegInfo* __restrict__ seg, ChunkDesc* __restrict__ descs,
                                               u32* __restrict__ d_nchunks, u32* __restrict__ d_mlen, u32 opt) {
    __shared__ u32 ws[16]; __shared__ u32 carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (u32 s0 = 0; s0 < S; s0 += 1024) {
        const u32 s = s0 + threadIdx.x;
        u32 start = 0, size = 0, nch = 0;
        if (s < S) { start = src[(u64)s * sstride]; size = src[(u64)(s + 1) * sstride] - start; nch = (size + CH - 1) / CH; }
        u32 y = nch;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 t = __shfl_up(y, d); if (lane >= d) y += t; }
        if (lane == 63) ws[wave] = y;
        __syncthreads();
        if (wave == 0) {
            const u32 wx = lane < 16 ? ws[lane] : 0u; u32 wy = wx;
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) { const u32 t = __shfl_up(wy, d); if (lane >= d) wy += t; }
            if (lane < 16) ws[lane] = wy - wx;
        }
        __syncthreads();
        const u32 cb = carry_s + ws[wave] + y - nch;
        if (s < S) {
            SegInfo si; si.start = start; si.nch = nch; si.chunk_base = cb; si.pad = 0;
            seg[s] = si;
            for (u32 c = 0; c < nch; ++c) {
                ChunkDesc d;
                d.begin = (u64)start + (u64)c * CH;
                const u64 e = d.begin + CH, lim = (u64)start + size;
                d.end = e < lim ? e : lim;
                d.flat_base = opt ? s * P : cb * P + c;       // segment-owned scatter: first sub-bin of the segment
                d.stride = nch;
                descs[cb + c] = d;
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = cb + nch;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *d_nchunks = carry_s; *d_mlen = carry_s * P; }
}

// start offset of every final sub-partition q (F+1 entries)
// one level  (seg == nullptr): fstart[q] = scanned[q * stride1]            (q == F -> total)
// two levels: q = s*P + b -> scanned[chunk_base[s]*P + b*nch[s]]  (empty segment: its start)
__global__ void k_final_offsets(const u32* __restrict__ scanned, const SegInfo* __restrict__ seg,
                                u32 P, u32 stride1, const u32* __restrict__ d_mlen,
                                u32* __restrict__ fstart, u32 F) {
    const u32 q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q > F) return;
    if (seg == nullptr) { fstart[q] = scanned[(u64)q * stride1]; return; }
    if (q == F) { fstart[q] = scanned[*d_mlen]; return; }
    const u32 s = q / P, b = q % P;
    const SegInfo si = seg[s];
    fstart[q] = si.nch ? scanned[(u64)si.chunk_base * P + (u64)b * si.nch] : si.start;
}

// ------------------------------------------------------------------ K5+K6: hash aggregate
#ifndef CNT_NT
#define CNT_NT 1024          // 2 blocks x 16 waves per CU: the table kernels are LDS-latency bound, occupancy pays (5.5 -> 4.6 ms)
#endif
#define CNT_SLOTS 4096
#define CNT_MAXLOAD 3584          // distinct keys allowed per table (0.875)
#define CNT_LH 512                // histogram bins kept in LDS
#ifndef CNT_KPT
#define CNT_KPT 3                  // keys per thread prefetched for the next sub-partition: covers 3072 keys (mean <= 2560); the
#endif                             // rest of a larger one is read in the insert loop.  4 -> 3: 4.66 -> 4.47 ms (fewer idle rounds)

struct CountParams {
    u32 F;
    u32 amin, amax, histo_max;
    u32 maxload;              // distinct keys a table may hold (CNT_MAXLOAD / C2_MAXLOAD; tests lower it to force the finer-partition retry)
    u32 cap;                  // != 0: fixed-capacity sub-partition regions (segment-owned level-2 scatter):
    const u32* subcnt;        //       sub-partition q = keys [q*cap, q*cap + subcnt[q])
};
// key range of sub-partition q
template <bool REG>
__device__ __forceinline__ void sub_range(const CountParams& cp, const u32* __restrict__ fstart, u32 q, u64* begin, u32* n) {
    if (REG) { *begin = (u64)q * cp.cap; *n = cp.subcnt[q]; }
    else { const u32 b = fstart[q]; *begin = b; *n = fstart[q + 1] - b; }
}

// One persistent block per sub-partition in turn.  Insert = 64-bit LDS CAS on
// the key + LDS add on the count; the slot of every newly claimed key is
// appended to a list, so the sweep (histogram every distinct key, keep the
// solid ones via wave ballot + prefix, reset the slot) costs O(distinct), not
// O(table).  Solid rows are written IN PLACE over the sub-partition's own key
// range (abundance to `abund` at the same index).  The keys of the block's
// next sub-partition are loaded into registers before the sweep, so HBM reads
// overlap the LDS work.
// keys      : mixed keys grouped by sub-partition (fstart)
// solid_keys: where the solid rows' (still mixed) keys go, at index begin+pos.
//             one-word keys: == keys (in place); two-word keys: the free ping-pong buffer
// abund     : abundance of the solid rows at the same index
__device__ __forceinline__ void table_insert1(u64* tk, u32* tc, unsigned short* lst, u32* ndist, u32* ovf, u64 h, u32 inc = 1u) {
    u32 slot = (u32)h & (CNT_SLOTS - 1);
    int probe = 0;
    for (; probe < CNT_SLOTS; ++probe) {
        u64 old = tk[slot];
        if (old == DSK_EMPTY) {
            old = atomicCAS(&tk[slot], DSK_EMPTY, h);
            if (old == DSK_EMPTY) { lst[atomicAdd(ndist, 1u)] = (unsigned short)slot; old = h; }
        }
        if (old == h) { atomicAdd(&tc[slot], inc); return; }
        slot = (slot + 1) & (CNT_SLOTS - 1);
    }
    *ovf = 1;
}

// Memory latency is kept off the critical path: the RANGE of sub-partition q + 2 * grid (subcnt[q] / fstart[q], a scalar
// load) is requested an iteration early, so the key loads of q + grid go out right after the inserts of q without
// waiting for a dependent load first, and every load is unconditional (clamped indices), which lets the compiler count
// the outstanding ones instead of draining them (4.46 -> 4.10 ms; a read-only skeleton of this loop streams the keys
// at 6 TB/s, so the rest of the time is the LDS phases themselves).
// REG: fixed-capacity regions (q * cap, subcnt[q]) instead of exact offsets (compile-time: see k_count_mw).
template <bool REG>
__global__ __launch_bounds__(CNT_NT) void k_count1(u64* keys, u64* solid_keys, const u32* __restrict__ fstart,
                                                    u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                    u64* __restrict__ ghist, u64* __restrict__ gstats,
                                                    u32* __restrict__ overflow, CountParams cp, const u32* __restrict__ subcnt) {
    __shared__ u64 tk[CNT_SLOTS];
    __shared__ u32 tc[CNT_SLOTS];
    __shared__ unsigned short lst[CNT_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 s_ctr[2][4];                 // [parity][ndist, out, ovf]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < CNT_SLOTS; s += CNT_NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += CNT_NT) lh[b] = 0;
    if (tid < 8) s_ctr[tid >> 2][tid & 3] = 0;
    u32 ones = 0;          // lane 0 of each wave: abundance-1 keys seen (flushed at the end)
    u64 ndist_acc = 0;
    // range words of sub-partition qq (clamped to a valid index: the caller ignores them when qq >= F)
    auto range_lo = [&](u32 qq) { const u32 c = qq < cp.F ? qq : cp.F - 1; return REG ? subcnt[c] : fstart[c]; };
    auto range_hi = [&](u32 qq) { const u32 c = qq < cp.F ? qq : cp.F - 1; return REG ? 0u : fstart[c + 1]; };
    auto begin_of = [&](u32 qq, u32 lo) { return REG ? (u64)qq * cp.cap : (u64)lo; };
    // (REG: the top bit of subcnt says "this sub-partition goes on in extension regions": it reads as empty here and is counted by
    //  k_count_chained -- a branch for it inside this loop cost 1.6 of 4.0 ms: the kernel's speed follows its instruction schedule)
    auto count_of = [&](u32 qq, u32 lo, u32 hi) { return qq < cp.F ? (REG ? ((int)lo < 0 ? 0u : lo) : hi - lo) : 0u; };
    // two register sets of keys, used in turn: the keys of sub-partition q + 2 * grid are requested when q's inserts are done
    // and consumed two iterations later, so by then neither the loads nor the solid-row stores issued in between (vmcnt counts
    // both, in issue order) hold the wave up
    struct Sub { u32 q; u64 begin; u32 n; };
    auto load_keys = [&](const Sub& sb, u64 (&pk)[CNT_KPT]) {      // branch-free: an index past the keys re-reads the last one (or key 0 of an empty range)
        const u32 last = sb.n ? sb.n - 1 : 0u;
#pragma unroll
        for (int j = 0; j < CNT_KPT; ++j) { const u32 i = tid + j * CNT_NT; pk[j] = keys[sb.begin + (i < sb.n ? i : last)]; }
    };
    auto sub_of = [&](u32 qq, u32 lo, u32 hi) { Sub sb; sb.q = qq; sb.begin = qq < cp.F ? begin_of(qq, lo) : 0ull; sb.n = count_of(qq, lo, hi); return sb; };
    const u32 G = gridDim.x;
    u64 pa[CNT_KPT], pb[CNT_KPT];
    Sub sa = sub_of(blockIdx.x, range_lo(blockIdx.x), range_hi(blockIdx.x));
    Sub sb = sub_of(blockIdx.x + G, range_lo(blockIdx.x + G), range_hi(blockIdx.x + G));
    u32 rq = blockIdx.x + 2 * G, rlo = range_lo(rq), rhi = range_hi(rq);        // range of the sub-partition whose keys are loaded next
    load_keys(sa, pa);
    load_keys(sb, pb);
    lds_barrier();
    int par = 0;
    // inserts of `cur` from its register set, then the set is refilled with the keys of sub-partition rq; then cur's sweep
    auto one = [&](Sub& cur, u64 (&pk)[CNT_KPT]) {
        u32* ctr = s_ctr[par];
        const u32 q = cur.q, n = cur.n; const u64 begin = cur.begin;
#pragma unroll
        for (int j = 0; j < CNT_KPT; ++j)
            if ((u32)(tid + j * CNT_NT) < n) table_insert1(tk, tc, lst, &ctr[0], &ctr[2], pk[j]);
        for (u32 i = CNT_KPT * CNT_NT + tid; i < n; i += CNT_NT)                 // oversized sub-partition
            table_insert1(tk, tc, lst, &ctr[0], &ctr[2], keys[begin + i]);
        cur = sub_of(rq, rlo, rhi);
        load_keys(cur, pk);
        rq += G; rlo = range_lo(rq); rhi = range_hi(rq);
        lds_barrier();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;            // block-uniform
        if (bad) {
            for (int s = tid; s < CNT_SLOTS; s += CNT_NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
            if (tid == 0) *overflow = 1;
        } else {
            for (u32 i0 = 0; i0 < nd; i0 += CNT_NT) {
                const u32 i = i0 + tid;
                const bool act = i < nd;
                u64 key = 0; u32 c = 0;
                if (act) {
                    const u32 slot = lst[i];
                    key = tk[slot]; c = tc[slot];
                    tk[slot] = DSK_EMPTY; tc[slot] = 0;
                }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) {
                    const u32 bin = c < cp.histo_max ? c : cp.histo_max;
                    if (bin < CNT_LH) atomicAdd(&lh[bin], 1u);
                    else atomicAdd(&ghist[bin], 1ull);
                }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) {
                        const u32 pos = base + __popcll(ms & ((1ull << lane) - 1));
                        solid_keys[begin + pos] = key;
                        abund[begin + pos] = c;
                    }
                }
            }
        }
        lds_barrier();
        if (tid == 0) {
            nsolid[q] = bad ? 0u : ctr[1];
            ndist_acc += bad ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;     // this parity is next used two barriers from now
        }
        par ^= 1;
    };
    while (sa.q < cp.F) {
        one(sa, pa);
        if (sb.q >= cp.F) break;
        one(sb, pb);
    }
    lds_barrier();
    // flush block-local histogram
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    lds_barrier();
    for (int b = tid; b < CNT_LH; b += CNT_NT) {
        const u32 v = lh[b];
        if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v);
    }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

// ---- k_count1v3 (fixed-capacity regions only: the kernel of the histogram-free path): the same table WITHOUT the slot list.  The
// lane whose CAS claims a slot remembers the slot in a register and sweeps it itself after the barrier: the insert chain loses the
// list index (a returning LDS add and its wait) and the list write, the sweep loses the list read (by itself: 4.01 -> 3.98 ms --
// the kernel is bound by dependent LDS round trips and its barriers at 44 % LDS busy, not by the LDS instruction count).  A region
// holds at most cap <= CNT_V3_KEYS * CNT_NT keys, so a lane has at most CNT_V3_KEYS claims.  (Tried on the way, same results, both
// slower: a round that handles a lane's three keys together -- three reads, then three CASes in flight, a wave-private slot list,
// every step behind a wave-uniform ballot -- 6.2 ms; the same round one key at a time 5.0 ms; three blocks of 640 threads per CU
// -- this kernel needs 50 KB of LDS -- 4.9 ms: 64 VGPRs are not enough for five keys per lane in two register sets.)
#define CNT_V3_KEYS 5
#define CNT_NONE 0xFFFFFFFFu
// The probe loop is written for the WAVE -- one exit test per round (a ballot), no per-lane loop state: ~30 instructions per round
// against ~65 of the per-lane loop of table_insert1 (3.98 -> 3.8 ms).  Called under the lanes' own condition: the ballot sees
// the active ones.  -> the slot if this lane claimed it, else CNT_NONE
__device__ __forceinline__ u32 table_insert3(u64* tk, u32* tc, u32* ovf, u64 h) {
    u32 slot = (u32)h & (CNT_SLOTS - 1), res = CNT_NONE;
    bool pend = true;
    for (int probe = 0; probe < CNT_SLOTS; ++probe) {
        u64 old = 0ull;
        if (pend) old = tk[slot];
        const bool e = pend && old == DSK_EMPTY;
        if (e) { old = atomicCAS(&tk[slot], DSK_EMPTY, h); if (old == DSK_EMPTY) { res = slot; old = h; } }
        const bool m = pend && old == h;
        if (m) atomicAdd(&tc[slot], 1u);
        pend = pend && !m;
        slot = (slot + 1) & (CNT_SLOTS - 1);
        if (!__ballot(pend)) return res;
    }
    *ovf = 1;
    return res;
}

template <int NT, int KPT, int NKEYS>
__global__ __launch_bounds__(NT) void k_count1v3(u64* keys, u64* solid_keys, u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                      u64* __restrict__ ghist, u64* __restrict__ gstats,
                                                      u32* __restrict__ overflow, CountParams cp, const u32* __restrict__ subcnt) {
    __shared__ u64 tk[CNT_SLOTS];
    __shared__ u32 tc[CNT_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 s_ctr[2][4];                 // [parity][ndist, out, ovf]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < CNT_SLOTS; s += NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += NT) lh[b] = 0;
    if (tid < 8) s_ctr[tid >> 2][tid & 3] = 0;
    u32 ones = 0;          // lane 0 of each wave: abundance-1 keys seen (flushed at the end)
    u64 ndist_acc = 0;
    auto range_lo = [&](u32 qq) { const u32 c = qq < cp.F ? qq : cp.F - 1; return subcnt[c]; };
    auto count_of = [&](u32 qq, u32 lo) { return qq < cp.F ? ((int)lo < 0 ? 0u : lo) : 0u; };      // (chained: counted by k_count_chained)
    struct Sub { u32 q; u64 begin; u32 n; };
    auto load_keys = [&](const Sub& sb, u64 (&pk)[KPT]) {
        const u32 last = sb.n ? sb.n - 1 : 0u;
#pragma unroll
        for (int j = 0; j < KPT; ++j) { const u32 i = tid + j * NT; pk[j] = keys[sb.begin + (i < sb.n ? i : last)]; }
    };
    auto sub_of = [&](u32 qq, u32 lo) { Sub sb; sb.q = qq; sb.begin = qq < cp.F ? (u64)qq * cp.cap : 0ull; sb.n = count_of(qq, lo); return sb; };
    const u32 G = gridDim.x;
    u64 pa[KPT], pb[KPT];
    Sub sa = sub_of(blockIdx.x, range_lo(blockIdx.x));
    Sub sb = sub_of(blockIdx.x + G, range_lo(blockIdx.x + G));
    u32 rq = blockIdx.x + 2 * G, rlo = range_lo(rq);
    load_keys(sa, pa);
    load_keys(sb, pb);
    lds_barrier();
    int par = 0;
    auto one = [&](Sub& cur, u64 (&pk)[KPT]) {
        u32* ctr = s_ctr[par];
        const u32 q = cur.q, n = cur.n; const u64 begin = cur.begin;
        u32 cl[NKEYS];                                       // slots this lane claimed
#pragma unroll
        for (int j = 0; j < NKEYS; ++j) cl[j] = CNT_NONE;
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((u32)(tid + j * NT) < n) cl[j] = table_insert3(tk, tc, &ctr[2], pk[j]);
#pragma unroll
        for (int j = KPT; j < NKEYS; ++j)                // keys past the prefetched ones (n <= cap <= NKEYS * NT)
            if ((u32)(tid + j * NT) < n) cl[j] = table_insert3(tk, tc, &ctr[2], keys[begin + tid + j * NT]);
        {
            u32 mine = 0;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) mine += (u32)__popcll(__ballot(cl[j] != CNT_NONE));
            if (lane == 0 && mine) atomicAdd(&ctr[0], mine);
        }
        cur = sub_of(rq, rlo);
        load_keys(cur, pk);
        rq += G; rlo = range_lo(rq);
        lds_barrier();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;            // block-uniform
        if (bad) {
            for (int s = tid; s < CNT_SLOTS; s += NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
            if (tid == 0) *overflow = 1;
        } else {
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) {
                const bool act = cl[j] != CNT_NONE;
                if (!__ballot(act)) continue;                      // (wave-uniform)
                u64 key = 0; u32 c = 0;
                if (act) {
                    const u32 slot = cl[j];
                    key = tk[slot]; c = tc[slot];
                    tk[slot] = DSK_EMPTY; tc[slot] = 0;
                }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) {
                    const u32 bin = c < cp.histo_max ? c : cp.histo_max;
                    if (bin < CNT_LH) atomicAdd(&lh[bin], 1u);
                    else atomicAdd(&ghist[bin], 1ull);
                }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) {
                        const u32 pos = base + __popcll(ms & ((1ull << lane) - 1));
                        solid_keys[begin + pos] = key;
                        abund[begin + pos] = c;
                    }
                }
            }
        }
        lds_barrier();
        if (tid == 0) {
            nsolid[q] = bad ? 0u : ctr[1];
            ndist_acc += bad ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;     // this parity is next used two barriers from now
        }
        par ^= 1;
    };
    while (sa.q < cp.F) {
        one(sa, pa);
        if (sb.q >= cp.F) break;
        one(sb, pb);
    }
    lds_barrier();
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    lds_barrier();
    for (int b = tid; b < CNT_LH; b += NT) {
        const u32 v = lh[b];
        if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v);
    }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

// Sub-partitions that go on in extension regions (k_scatter_al, "region chains"; listed in chain_list by the scatter): the same
// table, sweep and in-place solid rows as k_count1, one block per listed sub-partition in turn, keys read region by region along
// the chain.  Such keys arrive in runs (a k-mer with thousands of occurrences), so a wave first adds up the lanes that hold the
// same key as its first lane -- one insert for all of them instead of a 64-way same-address atomic -- and the other lanes insert
// their own.  Rare by construction (no sub-partition of repeat-free reads is chained): written for clarity, not for speed.
__global__ __launch_bounds__(CNT_NT) void k_count_chained(u64* keys, u64* solid_keys, u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                          u64* __restrict__ ghist, u64* __restrict__ gstats, u32* __restrict__ overflow,
                                                          CountParams cp, const u32* __restrict__ subcnt, const u32* __restrict__ chain_next,
                                                          const u32* __restrict__ chain_list, const u32* __restrict__ d_nchained, u32 list_cap) {
    __shared__ u64 tk[CNT_SLOTS];
    __shared__ u32 tc[CNT_SLOTS];
    __shared__ unsigned short lst[CNT_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 ctr[4];                      // ndist, out, ovf
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 nchained = *d_nchained < list_cap ? *d_nchained : list_cap;
    if (blockIdx.x >= nchained) return;
    for (int s = tid; s < CNT_SLOTS; s += CNT_NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += CNT_NT) lh[b] = 0;
    if (tid < 4) ctr[tid] = 0;
    u32 ones = 0;          // lane 0 of each wave: abundance-1 keys seen (flushed at the end)
    u64 ndist_acc = 0;
    __syncthreads();
    for (u32 li = blockIdx.x; li < nchained; li += gridDim.x) {
        const u32 q = chain_list[li];
        const u64 begin = (u64)q * cp.cap;
        u32 rg = q;
        while (true) {
            const u32 f = subcnt[rg], nn = f & ~CHAIN_BIT;
            const u64 rb = (u64)rg * cp.cap;
            for (u32 i0 = 0; i0 < nn; i0 += CNT_NT) {
                const bool act = i0 + tid < nn;
                const u64 kv = keys[rb + (act ? i0 + tid : 0u)];
                const u64 first = ((u64)__builtin_amdgcn_readfirstlane((u32)(kv >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)kv);
                const bool same = act && kv == first;
                const u64 m = __ballot(same);
                if (same && lane == __ffsll((unsigned long long)m) - 1) table_insert1(tk, tc, lst, &ctr[0], &ctr[2], first, (u32)__popcll(m));
                if (act && !same) table_insert1(tk, tc, lst, &ctr[0], &ctr[2], kv);
            }
            if (!(f >> 31)) break;
            rg = chain_next[rg];
        }
        __syncthreads();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;            // block-uniform
        if (bad) {
            for (int s = tid; s < CNT_SLOTS; s += CNT_NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
            if (tid == 0) *overflow = 1;
        } else {
            for (u32 i0 = 0; i0 < nd; i0 += CNT_NT) {
                const u32 i = i0 + tid;
                const bool act = i < nd;
                u64 key = 0; u32 c = 0;
                if (act) {
                    const u32 slot = lst[i];
                    key = tk[slot]; c = tc[slot];
                    tk[slot] = DSK_EMPTY; tc[slot] = 0;
                }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) {
                    const u32 bin = c < cp.histo_max ? c : cp.histo_max;
                    if (bin < CNT_LH) atomicAdd(&lh[bin], 1u);
                    else atomicAdd(&ghist[bin], 1ull);
                }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) {
                        const u32 pos = base + __popcll(ms & ((1ull << lane) - 1));
                        solid_keys[begin + pos] = key;      // (distinct keys <= maxload < cap: the rows fit the home region)
                        abund[begin + pos] = c;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            nsolid[q] = bad ? 0u : ctr[1];
            ndist_acc += bad ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;
        }
        __syncthreads();
    }
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    __syncthreads();
    for (int b = tid; b < CNT_LH; b += CNT_NT) {
        const u32 v = lh[b];
        if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v);
    }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

// ---- two-word keys (k in 33..64): no 128-bit LDS CAS exists, so the table holds
// 32-bit INDICES: slot value v = 1 + (index of the representative key inside the
// sub-partition).  The sub-partition's keys are staged immutably in LDS (the
// first C2_STAGE of them; later ones are re-read from HBM), a slot is claimed
// with a 32-bit CAS on the index, and equality is checked against the staged
// representative -- no thread ever waits on another.
#define C2_SLOTS 4096
#define C2_MAXLOAD 3584
// staged keys: 32 KB of LDS whatever the key width (2048 two-word, 1024 four-word keys)
template <int W> struct CStage { static constexpr int N = 4096 / W; static constexpr int KPT = N / CNT_NT; };

template <int W>
__device__ __forceinline__ void table_insert2(const KN<W>* sk, const KN<W>* __restrict__ gkeys, u32* slots, u32* tc,
                                              unsigned short* lst, u32* ndist, u32* ovf, const KN<W>& key, u32 idx) {
    constexpr u32 C2_STAGE = CStage<W>::N;
    u32 slot = (u32)key.w[W - 1] & (C2_SLOTS - 1);
    for (int probe = 0; probe < C2_SLOTS; ++probe) {
        u32 v = slots[slot];
        if (v == 0) {
            v = atomicCAS(&slots[slot], 0u, idx + 1);
            if (v == 0) { lst[atomicAdd(ndist, 1u)] = (unsigned short)slot; atomicAdd(&tc[slot], 1u); return; }
        }
        const u32 r = v - 1;
        bool same = true;
        // W = 4: word by word (a struct temporary is spilled to scratch: 20.3 -> 12.4 ms at k = 101).  W = 2: the struct
        // temporary is the faster form (7.6 vs 8.3 ms at k = 63) -- measured, not reasoned: see the note at k_count_mw
        if (W == 2) {
            KN<W> rep;
            if (r < C2_STAGE) rep = sk[r]; else rep = gkeys[r];
            same = key_eq(rep, key);
        } else if (r < C2_STAGE) {
#pragma unroll
            for (int x = 0; x < W; ++x) same = same && (sk[r].w[x] == key.w[x]);
        } else {
#pragma unroll
            for (int x = 0; x < W; ++x) same = same && (gkeys[r].w[x] == key.w[x]);
        }
        if (same) { atomicAdd(&tc[slot], 1u); return; }
        slot = (slot + 1) & (C2_SLOTS - 1);
    }
    *ovf = 1;
}

// REG: fixed-capacity regions (q*cap, subcnt[q]) instead of exact offsets -- a compile-time switch, because this
// kernel's speed depends on its exact instruction schedule (a run-time branch here cost 45 % at k = 63)
template <int W, bool REG>
__global__ __launch_bounds__(CNT_NT) void k_count_mw(KN<W>* keys, KN<W>* solid_keys, const u32* __restrict__ fstart,
                                                     u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                     u64* __restrict__ ghist, u64* __restrict__ gstats,
                                                     u32* __restrict__ overflow, CountParams cp) {
    typedef KN<W> K2;                            // (multi-word key of this instantiation)
    constexpr int C2_STAGE = CStage<W>::N, C2_KPT = CStage<W>::KPT;
    __shared__ K2 sk[C2_STAGE];
    __shared__ u32 slots[C2_SLOTS];
    __shared__ u32 tc[C2_SLOTS];
    __shared__ unsigned short lst[C2_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 ctr[4];                      // ndist, out, ovf
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < C2_SLOTS; s += CNT_NT) { slots[s] = 0; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += CNT_NT) lh[b] = 0;
    if (tid < 4) ctr[tid] = 0;
    u32 ones = 0; u64 ndist_acc = 0;
    // (32-bit offsets on purpose: this kernel is sensitive to its register / address arithmetic shape; the host keeps
    //  F * cap below 2^32 for multi-word keys)
    u32 q = blockIdx.x, begin = 0, end = 0;
    K2 pk[C2_KPT];
    if (q < cp.F) {
        // (REG: the top bit of subcnt = "this sub-partition goes on in extension regions": it reads as empty here, k_count_chained_mw counts it)
        if (REG) { const u32 c = cp.subcnt[q]; begin = q * cp.cap; end = begin + ((int)c < 0 ? 0u : c); } else { begin = fstart[q]; end = fstart[q + 1]; }
#pragma unroll
        for (int j = 0; j < C2_KPT; ++j) { const u32 i = begin + tid + j * CNT_NT; if (i < end) pk[j] = keys[i]; }
    }
    lds_barrier();
    while (q < cp.F) {
        const u32 n = end - begin;
        const K2* gk = keys + begin;
#pragma unroll
        for (int j = 0; j < C2_KPT; ++j) { const u32 i = tid + j * CNT_NT; if (i < n) sk[i] = pk[j]; }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < C2_KPT; ++j) {
            const u32 i = tid + j * CNT_NT;
            if (i < n) table_insert2(sk, gk, slots, tc, lst, &ctr[0], &ctr[2], pk[j], i);
        }
        for (u32 i = C2_STAGE + tid; i < n; i += CNT_NT) {                 // oversized sub-partition
            const K2 kx = gk[i];
            table_insert2(sk, gk, slots, tc, lst, &ctr[0], &ctr[2], kx, i);
        }
        const u32 qn = q + gridDim.x;
        u32 nbeg = 0, nend = 0;
        if (qn < cp.F) {
            if (REG) { const u32 c = cp.subcnt[qn]; nbeg = qn * cp.cap; nend = nbeg + ((int)c < 0 ? 0u : c); } else { nbeg = fstart[qn]; nend = fstart[qn + 1]; }
#pragma unroll
            for (int j = 0; j < C2_KPT; ++j) { const u32 i = nbeg + tid + j * CNT_NT; if (i < nend) pk[j] = keys[i]; }
        }
        lds_barrier();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;
        if (bad) {
            for (int s = tid; s < C2_SLOTS; s += CNT_NT) { slots[s] = 0; tc[s] = 0; }
            if (tid == 0) *overflow = 1;
        } else {
            for (u32 i0 = 0; i0 < nd; i0 += CNT_NT) {
                const u32 i = i0 + tid;
                const bool act = i < nd;
                u64 kw[W]; u32 c = 0;
#pragma unroll
                for (int x = 0; x < W; ++x) kw[x] = 0;
                if (act) {
                    const u32 slot = lst[i];
                    const u32 r = slots[slot] - 1;
                    if (W == 2) {
                        K2 key; if (r < C2_STAGE) key = sk[r]; else key = gk[r];
#pragma unroll
                        for (int x = 0; x < W; ++x) kw[x] = key.w[x];
                    } else if (r < C2_STAGE) {
#pragma unroll
                        for (int x = 0; x < W; ++x) kw[x] = sk[r].w[x];
                    } else {
#pragma unroll
                        for (int x = 0; x < W; ++x) kw[x] = gk[r].w[x];
                    }
                    c = tc[slot];
                    slots[slot] = 0; tc[slot] = 0;
                }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) {
                    const u32 bin = c < cp.histo_max ? c : cp.histo_max;
                    if (bin < CNT_LH) atomicAdd(&lh[bin], 1u);
                    else atomicAdd(&ghist[bin], 1ull);
                }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) {
                        const u32 pos = base + __popcll(ms & ((1ull << lane) - 1));
#pragma unroll
                        for (int x = 0; x < W; ++x) solid_keys[begin + pos].w[x] = kw[x];
                        abund[begin + pos] = c;
                    }
                }
            }
        }
        lds_barrier();
        if (tid == 0) {
            nsolid[q] = bad ? 0u : ctr[1];
            ndist_acc += bad ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;
        }
        q = qn; begin = nbeg; end = nend;
    }
    lds_barrier();
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    lds_barrier();
    for (int b = tid; b < CNT_LH; b += CNT_NT) {
        const u32 v = lh[b];
        if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v);
    }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

// ---- k_count2v3 (two-word keys, fixed-capacity regions): the table of k_count1v3 keyed by the MIXED TOP WORD alone.  kmixN folds
// the low word into the top one before the 64-bit finalizer, so the mixed top word is a 64-bit hash of the whole k-mer: two different
// keys of one sub-partition (~ 1300 keys) share it with probability ~ 4 * 10^-14.  The insert is then the one-word insert (read,
// 64-bit CAS when the slot is empty, add) -- no keys staged in LDS, no index table with a second dependent 16-byte read per probe, no
// slot list -- plus ONE store by the claiming lane (the low word, tl[slot]) and ONE read per key after the barrier: every key checks
// the low word of the slot it was counted on.  A key that disagrees, or a key whose mixed top word IS the empty-slot value, ORs
// bit 1 into *overflow: the host then repeats the attempt with k_count_mw (index table, full compares) -- exact by construction,
// never expected (tests force it with crafted k-mers).  7.0 -> see NOTEBOOK.md section 6 "k = 63".
#define C2V_SLOTS 3072              // (not a power of two: 20 bytes per slot, two blocks per CU; home slot by multiply-shift)
#define C2V_MAXLOAD 2688            // 0.875
#define C2V_NKEYS 4                 // a region holds at most cap <= C2V_NKEYS * CNT_NT keys
#define C2V_KPT 2                   // keys per thread prefetched two sub-partitions ahead (2048 of a mean of <= 2560; the third is requested first thing)
#define CNT_OVF_VERIFY 2u           // bits of *overflow: "the top-word table cannot be trusted on this input" -- two k-mers with one top word,
#define CNT_OVF_SENTINEL 4u         //  a k-mer whose top word is the empty-slot value
__device__ __forceinline__ u32 table_insert3w(u64* tk, u64* tl, u32* tc, u32* ovf, u64 top, u64 low) {      // -> slot | claimed << 31, CNT_NONE when not placed
    u32 slot = (u32)(((u64)(u32)top * C2V_SLOTS) >> 32), res = CNT_NONE;
    bool pend = true;
    for (int probe = 0; probe < C2V_SLOTS; ++probe) {
        u64 old = 0ull;
        if (pend) old = tk[slot];
        const bool e = pend && old == DSK_EMPTY;
        u32 mine = 0u;
        if (e) { old = atomicCAS(&tk[slot], DSK_EMPTY, top); if (old == DSK_EMPTY) { mine = 0x80000000u; old = top; tl[slot] = low; } }
        const bool m = pend && old == top;
        if (m) { atomicAdd(&tc[slot], 1u); res = slot | mine; }
        pend = pend && !m;
        slot = slot + 1 == C2V_SLOTS ? 0u : slot + 1;
        if (!__ballot(pend)) return res;
    }
    *ovf = 1;
    return res;
}

template <int NT, int KPT, int NKEYS>
__global__ __launch_bounds__(NT, 8) void k_count2v3(const K2* __restrict__ keys, K2* __restrict__ solid_keys, u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                 u64* __restrict__ ghist, u64* __restrict__ gstats,
                                                 u32* __restrict__ overflow, CountParams cp, const u32* __restrict__ subcnt) {
    __shared__ u64 tk[C2V_SLOTS];
    __shared__ u64 tl[C2V_SLOTS];
    __shared__ u32 tc[C2V_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 s_ctr[2][4];                 // [parity][ndist, out, ovf]
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < C2V_SLOTS; s += NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += NT) lh[b] = 0;
    if (tid < 8) s_ctr[tid >> 2][tid & 3] = 0;
    u32 ones = 0;
    u64 ndist_acc = 0;
    auto range_lo = [&](u32 qq) { const u32 c = qq < cp.F ? qq : cp.F - 1; return subcnt[c]; };
    auto count_of = [&](u32 qq, u32 lo) { return qq < cp.F ? ((int)lo < 0 ? 0u : lo) : 0u; };      // (chained: counted by k_count_chained_mw)
    struct Sub { u32 q; u32 begin; u32 n; };                                                       // (32-bit offsets: the host keeps F * cap below 2^32 for two-word keys)
    auto load_keys = [&](const Sub& sb, K2 (&pk)[KPT]) {
        const u32 last = sb.n ? sb.n - 1 : 0u;
#pragma unroll
        for (int j = 0; j < KPT; ++j) { const u32 i = tid + j * NT; pk[j] = keys[sb.begin + (i < sb.n ? i : last)]; }
    };
    auto sub_of = [&](u32 qq, u32 lo) { Sub sb; sb.q = qq; sb.begin = qq < cp.F ? qq * cp.cap : 0u; sb.n = count_of(qq, lo); return sb; };
    const u32 G = gridDim.x;
    K2 pa[KPT], pb[KPT];
    Sub sa = sub_of(blockIdx.x, range_lo(blockIdx.x));
    Sub sb = sub_of(blockIdx.x + G, range_lo(blockIdx.x + G));
    u32 rq = blockIdx.x + 2 * G, rlo = range_lo(rq);
    load_keys(sa, pa);
    load_keys(sb, pb);
    lds_barrier();
    int par = 0;
    auto one = [&](Sub& cur, K2 (&pk)[KPT]) {
        u32* ctr = s_ctr[par];
        const u32 q = cur.q, n = cur.n, begin = cur.begin;
        u32 at[NKEYS];                                       // where every key of this lane was counted (bit 31: this lane claimed the slot)
        u64 lw[NKEYS > KPT ? NKEYS - KPT : 1];               // low words of the keys past the prefetched ones
        bool sentinel = false;
#pragma unroll
        for (int j = 0; j < NKEYS; ++j) at[j] = CNT_NONE;
        // the first key past the prefetched ones is requested now and flies under the inserts of those (a mean sub-partition is 2560 keys:
        // half the lanes have one; three prefetched keys per lane in two register sets do not fit 64 VGPRs)
        K2 late; late.w[0] = late.w[1] = 0ull;
        if (NKEYS > KPT && (u32)(tid + KPT * NT) < n) late = keys[begin + tid + KPT * NT];
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((u32)(tid + j * NT) < n) { at[j] = table_insert3w(tk, tl, tc, &ctr[2], pk[j].w[1], pk[j].w[0]); sentinel = sentinel || pk[j].w[1] == DSK_EMPTY; }
#pragma unroll
        for (int j = KPT; j < NKEYS; ++j) {
            lw[j - KPT] = 0ull;
            if ((u32)(tid + j * NT) < n) {
                K2 kx;
                if (j == KPT) kx = late; else kx = keys[begin + tid + j * NT];
                lw[j - KPT] = kx.w[0];
                at[j] = table_insert3w(tk, tl, tc, &ctr[2], kx.w[1], kx.w[0]);
                sentinel = sentinel || kx.w[1] == DSK_EMPTY;
            }
        }
        {
            u32 mine = 0;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) mine += (u32)__popcll(__ballot(at[j] != CNT_NONE && (at[j] >> 31)));
            if (lane == 0 && mine) atomicAdd(&ctr[0], mine);
        }
        lds_barrier();
        {   // every key against the low word of its slot (the claiming lanes' stores are visible now)
            bool wrong = false;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) {
                const u64 low = j < KPT ? pk[j < KPT ? j : 0].w[0] : lw[j < KPT ? 0 : j - KPT];
                if (at[j] != CNT_NONE) wrong = wrong || tl[at[j] & 0x7FFFFFFFu] != low;
            }
            if (wrong) atomicOr(overflow, CNT_OVF_VERIFY);
            if (sentinel) atomicOr(overflow, CNT_OVF_SENTINEL);
        }
        cur = sub_of(rq, rlo);
        load_keys(cur, pk);
        rq += G; rlo = range_lo(rq);
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;            // block-uniform
        if (bad) {
            lds_barrier();                                     // (the other waves' checks read tl; the reset below touches tk / tc only, but keep the phases apart)
            for (int s = tid; s < C2V_SLOTS; s += NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
            if (tid == 0) atomicOr(overflow, 1u);
        } else {
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) {
                const bool act = at[j] != CNT_NONE && (at[j] >> 31);
                if (!__ballot(act)) continue;                      // (wave-uniform)
                u64 top = 0, low = 0; u32 c = 0;
                if (act) {
                    const u32 slot = at[j] & 0x7FFFFFFFu;
                    top = tk[slot]; low = tl[slot]; c = tc[slot];
                    tk[slot] = DSK_EMPTY; tc[slot] = 0;
                }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) {
                    const u32 bin = c < cp.histo_max ? c : cp.histo_max;
                    if (bin < CNT_LH) atomicAdd(&lh[bin], 1u);
                    else atomicAdd(&ghist[bin], 1ull);
                }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) {
                        const u32 pos = base + __popcll(ms & ((1ull << lane) - 1));
                        K2 row; row.w[0] = low; row.w[1] = top;
                        solid_keys[begin + pos] = row;
                        abund[begin + pos] = c;
                    }
                }
            }
        }
        lds_barrier();
        if (tid == 0) {
            nsolid[q] = bad ? 0u : ctr[1];
            ndist_acc += bad ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;     // this parity is next used two barriers from now
        }
        par ^= 1;
    };
    while (sa.q < cp.F) {
        one(sa, pa);
        if (sb.q >= cp.F) break;
        one(sb, pb);
    }
    lds_barrier();
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    lds_barrier();
    for (int b = tid; b < CNT_LH; b += NT) {
        const u32 v = lh[b];
        if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v);
    }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

// Multi-word sub-partitions that went on in extension regions (k_scatter_al's region chains; listed in chain_list): the index table
// of k_count_mw addresses ONE contiguous key range, a chain is several.  Here a slot holds 1 + the GLOBAL index of the
// representative key (the host keeps (F + pool) * cap below 2^32 for multi-word keys) and equality is checked against keys[] in
// HBM -- L2 serves it: a chain is a few MB read once, and its keys come in runs of the same k-mer, so a wave first adds up the
// lanes that hold its first lane's key (one insert for all of them).  Rare by construction, written for clarity: one block per
// chained sub-partition.  The chained sub-partitions read as empty in k_count_mw (top bit of subcnt).
template <int W>
__device__ __forceinline__ void table_insert2g(const KN<W>* __restrict__ gkeys, u32* slots, u32* tc, unsigned short* lst, u32* ndist, u32* ovf,
                                               const KN<W>& key, u32 gidx, u32 inc) {
    u32 slot = (u32)key.w[W - 1] & (C2_SLOTS - 1);
    for (int probe = 0; probe < C2_SLOTS; ++probe) {
        u32 v = slots[slot];
        if (v == 0) {
            v = atomicCAS(&slots[slot], 0u, gidx + 1u);
            if (v == 0) { lst[atomicAdd(ndist, 1u)] = (unsigned short)slot; atomicAdd(&tc[slot], inc); return; }
        }
        bool same = true;
#pragma unroll
        for (int x = 0; x < W; ++x) same = same && (gkeys[v - 1u].w[x] == key.w[x]);
        if (same) { atomicAdd(&tc[slot], inc); return; }
        slot = (slot + 1) & (C2_SLOTS - 1);
    }
    *ovf = 1;
}
template <int W>
__global__ __launch_bounds__(CNT_NT) void k_count_chained_mw(const KN<W>* __restrict__ keys, KN<W>* __restrict__ solid_keys, u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                             u64* __restrict__ ghist, u64* __restrict__ gstats, u32* __restrict__ overflow,
                                                             CountParams cp, const u32* __restrict__ subcnt, const u32* __restrict__ chain_next,
                                                             const u32* __restrict__ chain_list, const u32* __restrict__ d_nchained, u32 list_cap) {
    __shared__ u32 slots[C2_SLOTS];
    __shared__ u32 tc[C2_SLOTS];
    __shared__ unsigned short lst[C2_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 ctr[4];                      // ndist, out, ovf
    const int tid = threadIdx.x, lane = tid & 63;
    const u32 nchained = *d_nchained < list_cap ? *d_nchained : list_cap;
    if (blockIdx.x >= nchained) return;
    for (int s = tid; s < C2_SLOTS; s += CNT_NT) { slots[s] = 0; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += CNT_NT) lh[b] = 0;
    if (tid < 4) ctr[tid] = 0;
    u32 ones = 0; u64 ndist_acc = 0;
    __syncthreads();
    for (u32 li = blockIdx.x; li < nchained; li += gridDim.x) {
        const u32 q = chain_list[li];
        const u32 begin = q * cp.cap;
        u32 rg = q;
        while (true) {
            const u32 f = subcnt[rg], nn = f & ~CHAIN_BIT;
            const u32 rb = rg * cp.cap;
            for (u32 i0 = 0; i0 < nn; i0 += CNT_NT) {
                const bool act = i0 + tid < nn;
                const u32 gi = rb + (act ? i0 + tid : 0u);
                const KN<W> kv = keys[gi];
                bool same = act;
#pragma unroll
                for (int x = 0; x < W; ++x) {
                    const u64 f0 = ((u64)__builtin_amdgcn_readfirstlane((u32)(kv.w[x] >> 32)) << 32) | __builtin_amdgcn_readfirstlane((u32)kv.w[x]);
                    same = same && kv.w[x] == f0;
                }
                const u64 m = __ballot(same);
                if (same && lane == __ffsll((unsigned long long)m) - 1) table_insert2g<W>(keys, slots, tc, lst, &ctr[0], &ctr[2], kv, gi, (u32)__popcll(m));
                if (act && !same) table_insert2g<W>(keys, slots, tc, lst, &ctr[0], &ctr[2], kv, gi, 1u);
            }
            if (!(f >> 31)) break;
            rg = chain_next[rg];
        }
        __syncthreads();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;
        if (bad) {
            for (int s = tid; s < C2_SLOTS; s += CNT_NT) { slots[s] = 0; tc[s] = 0; }
            if (tid == 0) atomicOr(overflow, 1u);      // (bit 1 of the word belongs to k_count2v3)
        } else {
            for (u32 i0 = 0; i0 < nd; i0 += CNT_NT) {
                const u32 i = i0 + tid;
                const bool act = i < nd;
                KN<W> key; u32 c = 0;
#pragma unroll
                for (int x = 0; x < W; ++x) key.w[x] = 0;
                if (act) {
                    const u32 slot = lst[i];
                    key = keys[slots[slot] - 1u];
                    c = tc[slot];
                    slots[slot] = 0; tc[slot] = 0;
                }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) {
                    const u32 bin = c < cp.histo_max ? c : cp.histo_max;
                    if (bin < CNT_LH) atomicAdd(&lh[bin], 1u);
                    else atomicAdd(&ghist[bin], 1ull);
                }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) {
                        const u32 pos = base + __popcll(ms & ((1ull << lane) - 1));
                        solid_keys[begin + pos] = key;      // (distinct keys <= maxload < cap: the rows fit the home region)
                        abund[begin + pos] = c;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            nsolid[q] = bad ? 0u : ctr[1];
            ndist_acc += bad ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;
        }
        __syncthreads();
    }
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    __syncthreads();
    for (int b = tid; b < CNT_LH; b += CNT_NT) {
        const u32 v = lh[b];
        if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v);
    }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

// ------------------------------------------------------------------ compaction
// One wave per sub-partition: copy its solid rows to the dense output and
// restore the k-mer from the mixed key.  soff = exclusive scan of the per-
// sub-partition solid counts (F+1 entries).
struct RowsOut { u64* w[4]; };                    // struct-of-arrays rows: word i of row r at w[i][r]
struct RowsIn { const u64* w[4]; };

template <int W>
__global__ __launch_bounds__(256) void k_compact(const typename KeyT<W>::T* __restrict__ keys, const u32* __restrict__ abund,
                                                 const u32* __restrict__ fstart, const u32* __restrict__ soff, u32 F,
                                                 RowsOut out, u32* __restrict__ out_ab, u32 cap) {
    const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= F) return;
    const u32 o = soff[q], ns = soff[q + 1] - o;
    const u64 b = cap ? (u64)q * cap : (u64)fstart[q];        // fixed-capacity regions or exact offsets
    for (u32 i = lane; i < ns; i += 64) {
        KN<W> kx = keys[b + i];
        kunmixN(kx);
#pragma unroll
        for (int x = 0; x < W; ++x) out.w[x][o + i] = kx.w[x];
        out_ab[o + i] = abund[b + i];
    }
}
template <>
__global__ __launch_bounds__(256) void k_compact<1>(const u64* __restrict__ keys, const u32* __restrict__ abund,
                                                    const u32* __restrict__ fstart, const u32* __restrict__ soff, u32 F,
                                                    RowsOut out, u32* __restrict__ out_ab, u32 cap) {
    const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (q >= F) return;
    const u32 o = soff[q], ns = soff[q + 1] - o;
    const u64 b = cap ? (u64)q * cap : (u64)fstart[q];        // fixed-capacity regions or exact offsets
    for (u32 i = lane; i < ns; i += 64) {
        out.w[0][o + i] = kunmix(keys[b + i]);
        out_ab[o + i] = abund[b + i];
    }
}

// helpers of the two-word row sort (two stable 64-bit radix passes over an index permutation)
__global__ void k_iota(u32* __restrict__ idx, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) idx[i] = (u32)i;
}
template <class T>
__global__ void k_gather(T* __restrict__ dst, const T* __restrict__ src, const u32* __restrict__ idx, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// Gathering the rows of a multi-word sort by index costs W + 1 random reads per row from the per-word arrays -- more than
// the sort itself.  k_top_key_aos therefore also packs every row into ONE record (W value words + the abundance, padded to a
// multiple of 16 bytes: 32 bytes for two words, 48 for four) while it reads the words anyway, and k_gather_aos fetches a row
// with one random access of that record.
template <int W> struct AosRow { static constexpr int WORDS = (W + 1 + 1) & ~1; };      // u64 words per record
template <int W>
__global__ __launch_bounds__(256) void k_top_key_aos(RowsIn rows, const u32* __restrict__ ab, u64 n, int bits, u64* __restrict__ key,
                                                     u32* __restrict__ idx, u64* __restrict__ aos) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sh = bits - 63, ws = sh >> 6, b = sh & 63;     // bits > 64 for multi-word values
    u64 w[W], lo = 0, hi = 0;
#pragma unroll
    for (int x = 0; x < W; ++x) { w[x] = rows.w[x][i]; if (x == ws) lo = w[x]; if (x == ws + 1) hi = w[x]; }
    key[i] = b ? (lo >> b) | (hi << (64 - b)) : lo;
    idx[i] = (u32)i;
    ulonglong2* rec = reinterpret_cast<ulonglong2*>(aos + i * AosRow<W>::WORDS);
#pragma unroll
    for (int x = 0; x < AosRow<W>::WORDS; x += 2)
        rec[x / 2] = make_ulonglong2(x < W ? w[x] : (x == W ? (u64)ab[i] : 0ull), x + 1 < W ? w[x + 1] : (x + 1 == W ? (u64)ab[i] : 0ull));
}
template <int W>
__global__ __launch_bounds__(256) void k_gather_aos(RowsOut dst, u32* __restrict__ dab, const u64* __restrict__ aos, const u32* __restrict__ idx, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const ulonglong2* rec = reinterpret_cast<const ulonglong2*>(aos + (u64)idx[i] * AosRow<W>::WORDS);
    u64 w[AosRow<W>::WORDS];
#pragma unroll
    for (int x = 0; x < AosRow<W>::WORDS; x += 2) { const ulonglong2 v = rec[x / 2]; w[x] = v.x; w[x + 1] = v.y; }
#pragma unroll
    for (int x = 0; x < W; ++x) dst.w[x][i] = w[x];
    dab[i] = (u32)w[W];
}

// whole rows in one pass: every thread takes four rows, reads their indices once and has all its W + 1 loads per row in flight
template <int W>
__global__ __launch_bounds__(256) void k_gather_rows(RowsOut dst, u32* __restrict__ dab, RowsIn src, const u32* __restrict__ sab,
                                                     const u32* __restrict__ idx, u64 n) {
    const u64 b0 = (u64)blockIdx.x * 1024 + threadIdx.x;          // rows b0, b0 + 256, ..: consecutive lanes, consecutive rows
    u32 id[4]; u64 v[4][W]; u32 a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) id[r] = b0 + r * 256 < n ? idx[b0 + r * 256] : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int x = 0; x < W; ++x) v[r][x] = src.w[x][id[r]];
        a[r] = sab[id[r]];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (b0 + r * 256 < n) {
#pragma unroll
            for (int x = 0; x < W; ++x) dst.w[x][b0 + r * 256] = v[r][x];
            dab[b0 + r * 256] = a[r];
        }
}

// ------------------------------------------------------------------ test kernels
// canonical k-mer + validity for the window ending at every byte
template <int W>
__global__ __launch_bounds__(256) void k_enumerate(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                   u64 nwords, u64 nbytes, int k, u64* __restrict__ out, uint8_t* __restrict__ valid, int ow) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 wi = t >> 1;
    if (wi >= nwords) return;
    const int t0 = (int)(t & 1) * 16;
    u32 vm;
    if (W == 1) {
        u64 c[16];
        vm = gen_kmers1<16>(packed, inval, wi, t0, k, c);
        for (int j = 0; j < 16; ++j) {
            const u64 p = wi * 32 + t0 + j;
            if (p < nbytes) { const bool v = vm & (1u << j); out[p] = v ? c[j] : 0ull; valid[p] = v; }
        }
    } else {
        constexpr int WW = W > 1 ? W : 2;
        KN<WW> c[16];
        vm = gen_kmers_multi(packed, inval, wi, t0, k, c);
        for (int j = 0; j < 16; ++j) {
            const u64 p = wi * 32 + t0 + j;
            if (p < nbytes) {
                const bool v = vm & (1u << j);
                for (int x = 0; x < WW; ++x) if (x < ow) out[(u64)ow * p + x] = v ? c[j].w[x] : 0ull;
                valid[p] = v;
            }
        }
    }
}

// Minimizer (m <= 16) of the k-mer window ending at every byte: the smallest canonical
// m-mer (A<C<T<G numeric order) inside the window.  Stands in for gatb-core's
// ModelMinimizer on the path (named in BASELINE.json; call site src/DSK.cpp:63 getConfig);
// used by the parity tests today and by the super-k-mer exchange planned next.
// One thread = 16 consecutive end positions; rolling m-mers over the 2-bit stream.
__global__ __launch_bounds__(256) void k_minimizers(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                    u64 nwords, u64 nbytes, int k, int m,
                                                    u32* __restrict__ minim, uint8_t* __restrict__ valid) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 p0 = t * 16;
    if (p0 >= nbytes) return;
    const u32 mmask = (m == 16) ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1);
    const int halo = k - 1;                                 // bases needed before p0
    const long long s0 = (long long)p0 - halo;
    u32 cm[16 + 127];                                       // canonical m-mer ending at s0 + i
    u32 f = 0, r = 0; int run = 0;
    int runs[16];
    for (int i = 0; i < halo + 16; ++i) {
        const long long pos = s0 + i;
        bool ok = false; u32 c = 0;
        if (pos >= 0 && (u64)pos < nbytes) {
            const u64 w = (u64)pos >> 5; const int j = (int)(pos & 31);
            ok = !((inval[w] >> (31 - j)) & 1u);
            c = (u32)(packed[w] >> (62 - 2 * j)) & 3u;
        }
        if (!ok) { run = 0; f = 0; r = 0; }
        else { f = ((f << 2) | c) & mmask; r = (r >> 2) | ((c ^ 2u) << (2 * (m - 1))); ++run; }
        cm[i] = f < r ? f : r;
        if (i >= halo) runs[i - halo] = run;
    }
    for (int j = 0; j < 16; ++j) {
        const u64 p = p0 + j;
        if (p >= nbytes) break;
        const bool v = runs[j] >= k;
        u32 best = 0xFFFFFFFFu;
        if (v) for (int e = halo + j - (k - m); e <= halo + j; ++e) best = cm[e] < best ? cm[e] : best;
        minim[p] = v ? best : 0u;
        valid[p] = v ? 1 : 0;
    }
}

#define FIX_CAP 32
// Multi-word rows: the same two-step order.  k_top_key builds the top 63 bits of every value (bit 63 stays clear:
// rocPRIM's partial-range sort misbehaves when end_bit == 64) next to the identity permutation; after a radix sort
// of (key, index) on the key's top 32 bits and a gather of the rows, k_fix_runs_multi orders the runs of equal
// prefix by full multi-word comparison.
template <int W>
__global__ __launch_bounds__(256) void k_top_key(RowsIn rows, u64 n, int bits, u64* __restrict__ key, u32* __restrict__ idx) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sh = bits - 63, ws = sh >> 6, b = sh & 63;     // bits > 64 for multi-word values
    u64 lo = 0, hi = 0;
#pragma unroll
    for (int x = 0; x < W; ++x) { if (x == ws) lo = rows.w[x][i]; if (x == ws + 1) hi = rows.w[x][i]; }
    key[i] = b ? (lo >> b) | (hi << (64 - b)) : lo;
    idx[i] = (u32)i;
}
template <int W>
__device__ __forceinline__ bool row_less(const u64 (&a)[W], const RowsOut& r, u64 j) {     // a < row j ?
#pragma unroll
    for (int x = W - 1; x >= 0; --x) { const u64 v = r.w[x][j]; if (a[x] != v) return a[x] < v; }
    return false;
}
// Runs of 33 .. FIX_BLOCK_ROWS rows (16384: the one- and two-error variants of a 63-mer are 4300; the error variants of a k-mer with 10^5 and more occurrences share their first 63 bits) are
// LISTED -- list[0] = how many, then (first row, rows) pairs -- and ordered by k_fix_long_runs, one block per run; beyond that, or
// when the list is full, *flag (the full-width fallback).
#define FIX_LIST_CAP 4096
#define FIX_BLOCK_ROWS 16384
template <int W>
__global__ __launch_bounds__(256) void k_fix_runs_multi(RowsOut rows, u32* __restrict__ ab, const u64* __restrict__ key, u64 n, int sh, u32* __restrict__ flag,
                                                        const u32* __restrict__ ties, u32* __restrict__ list) {
    if (ties && *ties == 0u) return;                         // the sort saw no two equal keys: nothing to order
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) {      // (grid-stride: the launch may be capped)
        const u64 p = key[i] >> sh;
        if (i > 0 && (key[i - 1] >> sh) == p) continue;         // not a run head
        u64 e = i + 1;
        while (e < n && e - i <= FIX_BLOCK_ROWS && (key[e] >> sh) == p) ++e;
        const u64 L = e - i;
        if (L == 1) continue;
        if (L > FIX_CAP) {
            if (L > FIX_BLOCK_ROWS || i + L >= 0xFFFFFFFFull || list == nullptr) { *flag = 1; continue; }
            const u32 at = atomicAdd(&list[0], 1u);
            if (at < FIX_LIST_CAP) { list[1 + 2 * at] = (u32)i; list[2 + 2 * at] = (u32)L; } else *flag = 1;
            continue;
        }
        for (u64 a = i + 1; a < e; ++a) {
            u64 kv[W]; const u32 av = ab[a];
#pragma unroll
            for (int x = 0; x < W; ++x) kv[x] = rows.w[x][a];
            u64 b = a;
            while (b > i && row_less<W>(kv, rows, b - 1)) {
#pragma unroll
                for (int x = 0; x < W; ++x) rows.w[x][b] = rows.w[x][b - 1];
                ab[b] = ab[b - 1]; --b;
            }
#pragma unroll
            for (int x = 0; x < W; ++x) rows.w[x][b] = kv[x];
            ab[b] = av;
        }
    }
}
// one block per listed run: a bitonic network over the run's row numbers in LDS (rows compared word by word in HBM / L2 -- the run
// is a few hundred KB), then every thread fetches the rows of its final positions, and after a barrier writes them there
template <int W>
__global__ __launch_bounds__(1024) void k_fix_long_runs(RowsOut rows, u32* __restrict__ ab, const u32* __restrict__ list, const u32* __restrict__ ties) {
    if (ties && *ties == 0u) return;
    __shared__ unsigned short idx[FIX_BLOCK_ROWS];
    const u32 nl = list[0] < FIX_LIST_CAP ? list[0] : FIX_LIST_CAP;
    for (u32 r = blockIdx.x; r < nl; r += gridDim.x) {
        const u64 base = list[1 + 2 * r]; const u32 L = list[2 + 2 * r];
        u32 np2 = 2; while (np2 < L) np2 <<= 1;
        __syncthreads();
        for (u32 j = threadIdx.x; j < np2; j += 1024) idx[j] = (unsigned short)(j < L ? j : 0xFFFFu);      // (pads order behind every row)
        __syncthreads();
        auto less = [&](u32 a, u32 b) {          // row a < row b (row numbers inside the run; a pad is larger than any row)
            if (a == 0xFFFFu || b == 0xFFFFu) return a != 0xFFFFu && b == 0xFFFFu;
#pragma unroll
            for (int x = W - 1; x >= 0; --x) { const u64 va = rows.w[x][base + a], vb = rows.w[x][base + b]; if (va != vb) return va < vb; }
            return false;
        };
        for (u32 kk = 2; kk <= np2; kk <<= 1)
            for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                for (u32 x = threadIdx.x; x < np2; x += 1024) {
                    const u32 y = x ^ jj;
                    if (y > x) {
                        const u32 a0 = idx[x], a1 = idx[y];
                        const bool up = (x & kk) == 0;
                        if (less(a1, a0) == up) { idx[x] = (unsigned short)a1; idx[y] = (unsigned short)a0; }
                    }
                }
                __syncthreads();
            }
        constexpr int RPT = FIX_BLOCK_ROWS / 1024;
        u64 kv[RPT][W]; u32 av[RPT];
#pragma unroll
        for (int t = 0; t < RPT; ++t) {
            const u32 j = threadIdx.x + 1024 * t;
            if (j < L) {
                const u32 src = idx[j];
#pragma unroll
                for (int x = 0; x < W; ++x) kv[t][x] = rows.w[x][base + src];
                av[t] = ab[base + src];
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RPT; ++t) {
            const u32 j = threadIdx.x + 1024 * t;
            if (j < L) {
#pragma unroll
                for (int x = 0; x < W; ++x) rows.w[x][base + j] = kv[t][x];
                ab[base + j] = av[t];
            }
        }
    }
}

// ------------------------------------------------------------------ multi-bank merge (solidity kinds, 2-D histogram)
// Input: the union of the per-bank rows, sorted by k-mer, value = (bank << 32) | abundance.  One thread per
// row; the first row of every k-mer walks its run (<= number of banks) and decides solidity:
//   kind 0 sum  : amin <= sum <= amax            3 one : some bank has amin <= c <= amax
//   kind 1 min  : amin <= min over banks <= amax 4 all : every bank has amin <= c <= amax
//   kind 2 max  : amin <= max over banks <= amax 5 custom: banks in `mask` have c >= amin, the others c == 0
// (README.md:12 documents the default; the other kinds are named by the -solidity-kind option of gatb-core and
// are NOT pinned by any reference test -- see DESIGN.md).  The abundance written is the sum over banks.
// hist[min(sum, hmax)]++ for every distinct k-mer; h2d[min(reads, hmax)][min(genome, 10)]++ with genome = bank 0
// and reads = the other banks (README.md:98-102, utils/plot-histo2D.R:22-30).
struct MergeParams { u32 nbanks, kind, mask, amin, amax, hmax, want2d; };
#define MB_LH 256
#define MB_L2 32
template <int W>
__device__ __forceinline__ bool rows_same(const RowsIn& r, u64 a, u64 b) {
    bool e = true;
#pragma unroll
    for (int x = 0; x < W; ++x) e = e && (r.w[x][a] == r.w[x][b]);
    return e;
}
template <int W>
__global__ __launch_bounds__(256) void k_merge_banks(RowsIn rows, const u64* __restrict__ val,
                                                     u64 n, MergeParams mp, u32* __restrict__ flag, u32* __restrict__ sumv,
                                                     u64* __restrict__ ghist, u64* __restrict__ gh2d, u64* __restrict__ gstats) {
    __shared__ u32 lh[MB_LH];
    __shared__ u32 l2[MB_L2 * 11];
    for (int b = threadIdx.x; b < MB_LH; b += 256) lh[b] = 0;
    for (int b = threadIdx.x; b < MB_L2 * 11; b += 256) l2[b] = 0;
    __syncthreads();
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    u32 ndist = 0;
    if (i < n) {
        const bool head = i == 0 || !rows_same<W>(rows, i - 1, i);
        u32 f = 0, sv = 0;
        if (head) {
            u64 sum = 0; u32 mx = 0, genome = 0, present = 0, inwin = 0, nonzero = 0, nb = 0; u32 mn = 0xFFFFFFFFu;
            for (u64 j = i; j < n && rows_same<W>(rows, i, j) && nb < mp.nbanks; ++j, ++nb) {
                const u64 v = val[j]; const u32 bank = (u32)(v >> 32), c = (u32)v;
                sum += c; mx = c > mx ? c : mx; mn = c < mn ? c : mn;
                if (bank == 0) genome = c;
                nonzero |= 1u << bank;
                if (c >= mp.amin) present |= 1u << bank;
                if (c >= mp.amin && c <= mp.amax) inwin |= 1u << bank;
            }
            if (nb < mp.nbanks) mn = 0;                             // absent from some bank
            const u32 all = mp.nbanks >= 32 ? 0xFFFFFFFFu : ((1u << mp.nbanks) - 1);
            const u32 s32 = sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)sum;
            bool solid;
            switch (mp.kind) {
                case 1: solid = mn >= mp.amin && mn <= mp.amax; break;
                case 2: solid = mx >= mp.amin && mx <= mp.amax; break;
                case 3: solid = inwin != 0; break;
                case 4: solid = inwin == all; break;
                case 5: solid = (present & mp.mask) == (mp.mask & all) && (nonzero & ~mp.mask) == 0; break;
                default: solid = s32 >= mp.amin && s32 <= mp.amax;
            }
            f = solid ? 1u : 0u; sv = s32; ndist = 1;
            const u32 hb = s32 < mp.hmax ? s32 : mp.hmax;
            if (hb < MB_LH) atomicAdd(&lh[hb], 1u); else atomicAdd(&ghist[hb], 1ull);
            if (mp.want2d) {
                const u64 reads = sum - genome;
                const u32 r = reads < mp.hmax ? (u32)reads : mp.hmax, gcol = genome < 10 ? genome : 10u;
                if (r < MB_L2) atomicAdd(&l2[r * 11 + gcol], 1u); else atomicAdd(&gh2d[(u64)r * 11 + gcol], 1ull);
            }
        }
        flag[i] = f; sumv[i] = sv;
    }
    // block totals
    for (int d = 32; d >= 1; d >>= 1) ndist += __shfl_down(ndist, d);
    if ((threadIdx.x & 63) == 0 && ndist) atomicAdd(&gstats[0], (u64)ndist);
    __syncthreads();
    for (int b = threadIdx.x; b < MB_LH; b += 256) if (lh[b]) atomicAdd(&ghist[b < (int)mp.hmax ? b : (int)mp.hmax], (u64)lh[b]);
    if (mp.want2d) for (int b = threadIdx.x; b < MB_L2 * 11; b += 256) if (l2[b]) atomicAdd(&gh2d[b], (u64)l2[b]);
}

// keep the flagged rows (order preserved): pos = exclusive scan of flag
template <int W>
__global__ __launch_bounds__(256) void k_pick_rows(RowsIn rows, const u32* __restrict__ sumv,
                                                   const u32* __restrict__ flag_in, const u32* __restrict__ posx, u64 n,
                                                   RowsOut out, u32* __restrict__ out_ab) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !flag_in[i]) return;
    const u32 p = posx[i];
#pragma unroll
    for (int x = 0; x < W; ++x) out.w[x][p] = rows.w[x][i];
    out_ab[p] = sumv[i];
}
__global__ void k_pack_bank(u64* __restrict__ dst, const u32* __restrict__ ab, u64 n, u32 bank) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ((u64)bank << 32) | ab[i];
}
// Sum and sum of squares over the `nch` chunks of every bin of a (bin-major) chunk x bin matrix: one wave per bin.  The level-1
// loads of a positional sample of tiles: mom[2 b] = keys of bin b in the sample, mom[2 b + 1] = sum of the squared per-tile
// counts -- the spread tells how clumped a bin's keys arrive (a poly-A read is 120 keys of one bin in a row).
__global__ __launch_bounds__(256) void k_bin_moments(const u32* __restrict__ matrix, u32 nch, u32 P, u64* __restrict__ mom) {
    const u32 b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= P) return;
    u64 s = 0, q = 0;
    for (u32 c = lane; c < nch; c += 64) { const u64 x = matrix[(u64)b * nch + c]; s += x; q += x * x; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_down(s, d); q += __shfl_down(q, d); }
    if (lane == 0) { mom[2 * b] = s; mom[2 * b + 1] = q; }
}
// Keys of a few level-1 bins, from the sample tiles: the bins whose sampled load stands far above the others hold a k-mer that
// alone is a large share of a whole level-1 bin (poly-A reads, a satellite).  lut[d] = slot of bin d (0xFF: not wanted).  Every
// block keeps every keep_step[slot]-th key of a slot that it meets -- a systematic sample of the arrival sequence, so a key's
// share of the kept keys is its share of the bin (the FIRST arrivals are not: a thread delivers a poly-A read's keys one by
// one) -- in its own HV_BLOCK_KEYS entries of out[slot][block][..], arrival numbers from LDS counters (a single global
// counter per slot took 1 ms: 40 K returning atomics on one address); kept[slot][block] = entries written.  The host then finds
// the dominant key(s) of every slot (dskgpu.hip: find_heavy) and the level-1 scatter counts them apart (k_scatter<.., HEAVY>).
#define HV_SLOTS 16
#define HV_COLLECT 2048            // keys aimed at per slot, over all blocks
#define HV_BLOCK_KEYS 32           // entries of a block per slot
template <int W, int SRC, int MODE>
__global__ __launch_bounds__(SC_NT) void k_collect_heavy(const u64* __restrict__ packed, const u32* __restrict__ inval, const typename KeyT<W>::T* __restrict__ keys,
                                                         const ChunkDesc* __restrict__ descs, const u32* __restrict__ d_nchunks, int k, DigitSpec ds, u32 P,
                                                         const unsigned char* __restrict__ lut, u32* __restrict__ kept, typename KeyT<W>::T* __restrict__ out, const u32* __restrict__ keep_step) {
    typedef typename KeyT<W>::T Key;
    constexpr int KPT = Tile<W>::KPT;
    __shared__ unsigned char slut[MAX_BINS + 8];
    __shared__ u32 lcnt[HV_SLOTS], lstep[HV_SLOTS];
    const int lane = threadIdx.x & 63;
    for (u32 b = threadIdx.x; b < P; b += SC_NT) slut[b] = lut[b];
    // (arrival numbers start at a per-block phase: starting every block at 0 would keep every block's FIRST key -- a quarter of all kept keys)
    if (threadIdx.x < HV_SLOTS) { const u32 st = keep_step[threadIdx.x]; lstep[threadIdx.x] = st; lcnt[threadIdx.x] = (blockIdx.x * 7919u) % st; }
    __syncthreads();
    const u32 nchunks = *d_nchunks;
    for (u32 g = blockIdx.x; g < nchunks; g += gridDim.x) {
        const ChunkDesc d = descs[g];
        const u64 step = SRC == 0 ? Tile<W>::WORDS : Tile<W>::KEYS;
        for (u64 t0 = d.begin; t0 < d.end; t0 += step) {
            Key h[KPT];
            u32 vm;
            if constexpr (SRC == 0) vm = tile_keys_reads(packed, inval, t0, d.end, k, h); else vm = tile_keys_array<W>(keys, t0, d.end, h);
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                u32 f = 0xFFu;
                if ((vm & (1u << j)) && key_in_pass<MODE>(digit_word(h[j]), ds)) f = slut[key_digit<MODE>(digit_word(h[j]), ds)];
                u64 todo = __ballot(f != 0xFFu);
                while (todo) {                   // one LDS atomic per wave and slot: the lanes of a slot take consecutive arrival numbers
                    const int lead = __ffsll((unsigned long long)todo) - 1;
                    const u32 fl = (u32)__shfl((int)f, lead);
                    const u64 grp = __ballot(f == fl);
                    u32 base = 0;
                    if (lane == lead) base = atomicAdd(&lcnt[fl], (u32)__popcll(grp));
                    base = (u32)__shfl((int)base, lead);
                    if (f == fl) {
                        const u32 idx = base + (u32)__popcll(grp & ((1ull << lane) - 1)), st = lstep[fl];
                        const u32 p0 = ((blockIdx.x * 7919u) % st + st - 1) / st;        // kept entries "before" the phase
                        if (idx % st == 0u && idx / st - p0 < HV_BLOCK_KEYS) out[((u64)fl * gridDim.x + blockIdx.x) * HV_BLOCK_KEYS + idx / st - p0] = h[j];
                    }
                    todo &= ~grp;
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < HV_SLOTS) {
        const u32 st = lstep[threadIdx.x];
        const u32 n = (lcnt[threadIdx.x] + st - 1) / st - ((blockIdx.x * 7919u) % st + st - 1) / st;
        kept[threadIdx.x * gridDim.x + blockIdx.x] = n < HV_BLOCK_KEYS ? n : HV_BLOCK_KEYS;
    }
}
// The k-mers counted apart by the level-1 scatter (HEAVY) join the result here: histogram, distinct count and, when solid, a row
// (value restored from the mixed key) in rows_k / rows_ab; gstats[1] = rows written.  One thread per slot.
template <int W>
__global__ void k_heavy_rows(const typename KeyT<W>::T* __restrict__ hv_keys, const unsigned long long* __restrict__ hv_cnt, u32 nslots, u32 amin, u32 amax, u32 histo_max,
                             u64* __restrict__ ghist, u64* __restrict__ gstats, u64* __restrict__ rows_k /* word x of row r at [x * nslots + r] */, u32* __restrict__ rows_ab) {
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nslots) return;
    const typename KeyT<W>::T key = hv_keys[i];
    const unsigned long long c64 = hv_cnt[i];
    if (is_empty_key(key) || c64 == 0ull) return;
    const u32 c = c64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)c64;
    atomicAdd(&ghist[c < histo_max ? c : histo_max], 1ull);
    atomicAdd(&gstats[0], 1ull);
    atomicAdd(&gstats[2], c64);                   // they count as keys the level-1 scatter placed
    if (c >= amin && c <= amax) {
        const u64 at = atomicAdd(&gstats[1], 1ull);
        if constexpr (W == 1) rows_k[at] = kunmix(key);
        else {
            KN<W> kx = key; kunmixN(kx);
#pragma unroll
            for (int x = 0; x < W; ++x) rows_k[(u64)x * nslots + at] = kx.w[x];
        }
        rows_ab[at] = c;
    }
}
__global__ void k_copy_u32(u32* __restrict__ dst, const u32* __restrict__ src, u64 n) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
