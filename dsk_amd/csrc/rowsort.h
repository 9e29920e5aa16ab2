// rowsort.h -- ascending order of the solid rows (one-word k-mers), hand-written for gfx950.
//
// The rows leave the count kernels grouped by hash sub-partition; the result contract is "sorted by k-mer value"
// (Partition<Count> rows as dsk2ascii walks them, utils/dsk2ascii.cpp:77-104; output partitions are slices of that order).
// 43 M rows of (u64 value, u32 abundance) on the bench workload: an MSD radix sort in three steps, five launches + a scan:
//
//   A  k_rs_hist + scan + k_rs_scatter : 1024 buckets on the top 10 value bits.  Exact chunk x bin offsets (the same
//      one-linear-scan matrix layout as the key scatters), 8192-row tiles staged in LDS, runs of ~8 rows per bucket and tile.
//   B  k_rs_split, 512-thread blocks taking one bucket at a time from a work counter: histogram of the next 8 bits in LDS
//      (9 / 10 bits above 96 M / 192 M rows), then the same LDS-staged scatter inside the bucket (a bucket is ~42 K rows =
//      500 KB: its second read comes from L2); the sub-bucket starts go to `sub`
//   C  k_rs_cells, one wave per sub-bucket (~160 rows, at most 512), in place: the rows are placed by the next 8 bits with LDS
//      atomics (< 1 row per cell on average); a row that shares its cell counts the smaller values of the cell and moves to
//      its final slot.  Sub-buckets of 513..4096 rows are listed and done by whole blocks (k_rs_big).
//
// Skew: buckets are exact (any distribution of values works); what the kernels do not order themselves -- a first-digit
// bucket above 64 x the mean + 256 K rows (one block would walk it alone) or a sub-bucket above 4096 rows -- raises *flag, and the host orders
// the rows with the full-width library sort instead (stats.sort_fallback): never seen on reads, provoked in tests.  Cells of any
// size are ordered here: above 64 rows by a block (k_rs_big), there above 16 with a bitonic network over the sub-bucket (the error
// variants of a poly-A k-mer share 13 and more leading bases).
#pragma once
#include "kernels.h"

#define RS_NT 1024                        // step A: one 1024-thread block per chunk, 8192-row tiles
#define RS_RPT 8
#define RS_TILE (RS_NT * RS_RPT)
#ifndef RS_ABITS
#define RS_ABITS 10                       // step A: bits of the first digit (experiments: -DRS_ABITS=11)
#endif
#define RS_ABINS (1 << RS_ABITS)
#define RS_BNT 512                        // step B: 512-thread blocks, 4096-row tiles (52 KB of LDS: three blocks per CU hide each
                                          //   other's memory latency; a heavy bucket is ~20 tiles)
#define RS_BTILE (RS_BNT * RS_RPT)
#define RS_CELLS 256                      // step C: cells of a sub-bucket (third digit)
#define RS_WAVE_ROWS 512                  // step C: a wave orders sub-buckets up to this size alone (8 rows per lane)
#define RS_MAX_ROWS (384ull << 20)        // most rows this sort takes: 1024 x 1024 sub-buckets of ~380 rows on average (the densest twice that)
#define RS_BLOCK_ROWS 4096                // larger ones: a whole block (k_rs_big); beyond this: flag
#define RS_WAVE_CELL_CAP 64               // rows sharing all three digits that are still ordered here (wave path / block path)
#define RS_OVS_CAP 16384                  // sub-buckets above RS_BLOCK_ROWS rows that a sort may list for another round on their remaining bits
#define RS_BLOCK_CELL_CAP 16              // block path: cells up to here are ordered by one thread each (insertion), a sub-bucket with a larger one by the bitonic network

// digit X of value v = (v >> shX) & mX; A = top 10 significant bits, B and C the next 8 + 8 (fewer for very small k)
struct RsSpec { int shA, shB, shC; u32 mA, mB, mC; };
__device__ __forceinline__ u32 rs_dig(u64 v, int sh, u32 m) { return (u32)(v >> sh) & m; }

// LDS of a scatter block: TILE rows (value + abundance), P + 1 counters and tile offsets, P cursors and deltas, scan scratch
template <int P, int TILE>
struct RsLds {
    u64* skey; u32* sab; u32* cnt; u32* off; u32* cur; u32* delta; u32* wsum; u32* tot;
    static constexpr size_t bytes = (size_t)TILE * 12 + (size_t)(P + 1) * 8 + (size_t)P * 8 + 17 * 4 + 16;
    __device__ __forceinline__ explicit RsLds(char* smem) {
        skey = reinterpret_cast<u64*>(smem);
        sab = reinterpret_cast<u32*>(smem + (size_t)TILE * 8);
        cnt = sab + TILE;                        // bins + 1 (dummy bin: slots past the end of the range)
        off = cnt + (P + 1);                     // bins + 1
        cur = off + (P + 1);
        delta = cur + P;
        wsum = delta + P;                        // 16
        tot = wsum + 16;
    }
};

// per-chunk histogram of the first digit -> matrix[bin * nch + chunk]
// (c0: first matrix column of this launch -- the rows counted here are chunks c0, c0 + 1, .. of a row set whose first c0 chunks come from
//  somewhere else: k_rs_hist_sp below)
__global__ __launch_bounds__(RS_NT) void k_rs_hist(const u64* __restrict__ v, u64 n, u32 chunk, u32 nch, u32* __restrict__ matrix, RsSpec sp, u32 c0 = 0) {
    __shared__ u32 lh[RS_ABINS];
    const u32 c = blockIdx.x;
    for (u32 b = threadIdx.x; b < RS_ABINS; b += RS_NT) lh[b] = 0;
    __syncthreads();
    const u64 beg = (u64)c * chunk;
    const u64 end = beg + chunk < n ? beg + chunk : n;
    for (u64 i0 = beg; i0 < end; i0 += 8 * RS_NT) {                   // eight loads in flight per thread
        u64 x[8]; bool ok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u64 i = i0 + threadIdx.x + (u64)j * RS_NT; ok[j] = i < end; x[j] = v[ok[j] ? i : end - 1]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (ok[j]) atomicAdd(&lh[rs_dig(x[j], sp.shA, sp.mA)], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < RS_ABINS; b += RS_NT) matrix[(u64)b * nch + c0 + c] = lh[b];
}

// ---- step A straight from the count kernels' output (one-word rows, a single pass): the solid rows of sub-partition q lie where
// the count kernel left them -- ns_q = soff[q + 1] - soff[q] of them from key index base(q) on (q * cap for fixed-capacity regions,
// fstart[q] with exact offsets), keys still MIXED, abundances at the same index of `ab`.  Until round 5 k_compact gathered them into a
// dense array first (0.5 GB read + 0.5 GB written, 0.30 ms) that step A then read again; here step A reads the regions itself:
// chunk c = the sub-partitions [c * qpc, (c + 1) * qpc), i.e. the logical rows [soff[c * qpc], soff[(c + 1) * qpc)).
struct RsSparse { const u64* keys; const u32* ab; const u32* soff; const u32* fstart; u32 cap, F, qpc; };
__device__ __forceinline__ u64 rs_sp_base(const RsSparse& s, u32 q) { return s.cap ? (u64)q * s.cap : (u64)s.fstart[q]; }

// histogram of the first digit, one wave per sub-partition in turn -> matrix[bin * nch + chunk]
__global__ __launch_bounds__(RS_NT) void k_rs_hist_sp(RsSparse s, u32 nch, u32* __restrict__ matrix, RsSpec sp) {
    __shared__ u32 lh[RS_ABINS];
    const u32 c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 b = threadIdx.x; b < RS_ABINS; b += RS_NT) lh[b] = 0;
    __syncthreads();
    const u32 q0 = c * s.qpc, q1 = q0 + s.qpc < s.F ? q0 + s.qpc : s.F;
    for (u32 q = q0 + wave; q < q1; q += RS_NT / 64) {
        const u32 o = s.soff[q], ns = s.soff[q + 1] - o;
        const u64 b = rs_sp_base(s, q);
        for (u32 i = lane; i < ns; i += 64) atomicAdd(&lh[rs_dig(kunmix(s.keys[b + i]), sp.shA, sp.mA)], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < RS_ABINS; b += RS_NT) matrix[(u64)b * nch + c] = lh[b];
}

// rows [beg, end) -> their bins, through LDS-staged tiles of NT * RS_RPT rows; L.cur[] = next free output index per bin,
// L.cnt[0..P] zero on entry.  The next tile's rows are loaded before the current tile enters its LDS phases.
// gdel (HUGE: row sets of 2^32 rows and more, step A slab by slab): a 64-bit offset per bin that is added to the (32-bit, slab-local)
// output index -- the slab's rows of bin b then land behind the rows the earlier slabs put there (dskgpu.hip: sort_rows_huge).
// where a tile's rows come from: a dense (value, abundance) array, or the sparse regions of the count kernels (RsSparse; lsoff = the chunk's
// slice of soff in LDS: the sub-partition of logical row r is found there by bisection, its key is un-mixed on the way in)
struct RsDenseSrc {
    const u64* v; const u32* ab;
    __device__ __forceinline__ void get(u64 r, u64& k, u32& a) const { k = v[r]; a = ab[r]; }
};
struct RsSparseSrc {
    RsSparse s; const u32* lsoff; u32 q0, nq;            // lsoff[x] = soff[q0 + x], x = 0 .. nq
    __device__ __forceinline__ void get(u64 r, u64& k, u32& a) const {
        u32 lo = 0, hi = nq;                              // largest x with lsoff[x] <= r (behind a run of empty sub-partitions: the one that holds r)
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((u64)lsoff[mid] <= r) lo = mid; else hi = mid; }
        const u64 src = rs_sp_base(s, q0 + lo) + (r - (u64)lsoff[lo]);
        k = kunmix(s.keys[src]); a = s.ab[src];
    }
};
template <int P, int NT, bool HUGE = false, class Src = RsDenseSrc>
__device__ __forceinline__ void rs_scatter_range(const Src src_rows, u64 beg, u64 end,
                                                 u64* __restrict__ ov, u32* __restrict__ oab, int sh, u32 m, const RsLds<P, NT * RS_RPT>& L,
                                                 const u64* __restrict__ gdel = nullptr) {
    constexpr u32 TILE = NT * RS_RPT;
    const u32 tid = threadIdx.x;
    u64 kk[RS_RPT], kn[RS_RPT]; u32 aa[RS_RPT], an[RS_RPT];
    auto load = [&](u64 t0, u64 (&k)[RS_RPT], u32 (&a)[RS_RPT]) {
        const u64 left = end - t0;
        const u32 n = left < (u64)TILE ? (u32)left : TILE;
#pragma unroll
        for (int j = 0; j < RS_RPT; ++j) { const u32 i = tid + (u32)j * NT; src_rows.get(t0 + (i < n ? i : n - 1), k[j], a[j]); }
    };
    if (beg < end) load(beg, kk, aa);
    for (u64 t0 = beg; t0 < end; t0 += TILE) {
        const u64 left = end - t0;
        const u32 n = left < (u64)TILE ? (u32)left : TILE;
        const bool more = t0 + TILE < end;
        if (more) load(t0 + TILE, kn, an);
        u32 rk[RS_RPT];
#pragma unroll
        for (int j = 0; j < RS_RPT; ++j) rk[j] = (tid + (u32)j * NT < n ? rs_dig(kk[j], sh, m) : (u32)P) << 16;
#pragma unroll
        for (int j = 0; j < RS_RPT; ++j) rk[j] |= atomicAdd(&L.cnt[rk[j] >> 16], 1u);
        lds_barrier();
        tile_scan<NT>(L.cnt, L.off, L.delta, L.cur, P, L.wsum, L.tot);
        lds_barrier();
#pragma unroll
        for (int j = 0; j < RS_RPT; ++j) {
            const u32 pos = L.off[rk[j] >> 16] + (rk[j] & 0xFFFFu);
            L.skey[pos] = kk[j]; L.sab[pos] = aa[j];
        }
        if (tid == 0) L.cnt[P] = 0;
        lds_barrier();
        const u32 ntile = *L.tot;
#pragma unroll
        for (int j = 0; j < RS_RPT; ++j) {
            const u32 i = tid + (u32)j * NT;
            if (i < ntile) {
                const u64 k = L.skey[i];
                const u32 dg = rs_dig(k, sh, m);
                const u32 dst = L.delta[dg] + i;
                if constexpr (HUGE) { const u64 d64 = (u64)dst + gdel[dg]; ov[d64] = k; oab[d64] = L.sab[i]; }
                else { ov[dst] = k; oab[dst] = L.sab[i]; }
            }
        }
        // no barrier: the next tile's rank phase only touches cnt; its first barrier orders this write-out before the next stage writes
        if (more) {
#pragma unroll
            for (int j = 0; j < RS_RPT; ++j) { kk[j] = kn[j]; aa[j] = an[j]; }
        }
    }
    lds_barrier();
}

// step A: chunk c scatters its rows to the 1024 buckets (offsets from the scanned matrix)
template <bool HUGE = false>
__global__ __launch_bounds__(RS_NT) void k_rs_scatter(const u64* __restrict__ v, const u32* __restrict__ ab, u64 n, u32 chunk, u32 nch,
                                                      const u32* __restrict__ scanned, u64* __restrict__ ov, u32* __restrict__ oab, RsSpec sp,
                                                      const u64* __restrict__ gdel, u32 c0 = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RsLds<RS_ABINS, RS_TILE> L(smem);
    const u32 c = blockIdx.x;
    for (u32 b = threadIdx.x; b < RS_ABINS; b += RS_NT) { L.cur[b] = scanned[(u64)b * nch + c0 + c]; L.cnt[b] = 0; }
    if (threadIdx.x == 0) L.cnt[RS_ABINS] = 0;
    lds_barrier();
    const u64 beg = (u64)c * chunk;
    const u64 end = beg + chunk < n ? beg + chunk : n;
    rs_scatter_range<RS_ABINS, RS_NT, HUGE>(RsDenseSrc{v, ab}, beg, end, ov, oab, sp.shA, sp.mA, L, gdel);
}
// the same from the sparse regions: chunk c = sub-partitions [c * qpc, ..); its slice of soff sits behind the scatter's LDS
#define RS_SP_MAXQ 1024                      // most sub-partitions per chunk (the LDS slice of soff; the bisection is <= 10 steps)
__global__ __launch_bounds__(RS_NT) void k_rs_scatter_sp(RsSparse s, u32 nch, const u32* __restrict__ scanned, u64* __restrict__ ov, u32* __restrict__ oab, RsSpec sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RsLds<RS_ABINS, RS_TILE> L(smem);
    u32* lsoff = reinterpret_cast<u32*>(smem + RsLds<RS_ABINS, RS_TILE>::bytes);
    const u32 c = blockIdx.x;
    const u32 q0 = c * s.qpc, q1 = q0 + s.qpc < s.F ? q0 + s.qpc : s.F, nq = q1 > q0 ? q1 - q0 : 0u;
    for (u32 b = threadIdx.x; b < RS_ABINS; b += RS_NT) { L.cur[b] = scanned[(u64)b * nch + c]; L.cnt[b] = 0; }
    for (u32 x = threadIdx.x; x <= nq; x += RS_NT) lsoff[x] = s.soff[q0 + x];
    if (threadIdx.x == 0) L.cnt[RS_ABINS] = 0;
    lds_barrier();
    if (nq == 0) return;
    rs_scatter_range<RS_ABINS, RS_NT, false, RsSparseSrc>(RsSparseSrc{s, lsoff, q0, nq}, (u64)lsoff[0], (u64)lsoff[nq], ov, oab, sp.shA, sp.mA, L);
}

// step B: a block splits one bucket at a time (rows [scanned[b * nch], scanned[(b + 1) * nch]) of v / ab) into 256 sub-buckets on
// the second digit and leaves their starts (absolute row indices) in sub[b * 257 ..].  Buckets are handed out by a work
// counter in ascending order: canonical k-mers are densest at small values (up to twice the mean), so the heavy buckets go first.
template <int BB>
__global__ __launch_bounds__(RS_BNT) void k_rs_split(const u64* __restrict__ v, const u32* __restrict__ ab, u32 nch, const u32* __restrict__ scanned,
                                                     u64* __restrict__ ov, u32* __restrict__ oab, u32* __restrict__ sub, RsSpec sp, u32* __restrict__ work,
                                                     u32 heavy, u32* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RsLds<BB, RS_BTILE> L(smem);
    u32& s_b = L.tot[1];                                 // (the spare word behind the scan scratch: no static LDS next to the dynamic block)
    const u32 tid = threadIdx.x, lane = tid & 63;
    constexpr int CPL = BB / 64;                         // counters per lane in the scan of a bucket's histogram
    for (;;) {
        __syncthreads();
        if (tid == 0) s_b = atomicAdd(work, 1u);
        __syncthreads();
        const u32 b = s_b;
        if (b >= RS_ABINS) break;
        const u32 beg = scanned[(u64)b * nch], end = scanned[(u64)(b + 1) * nch];     // (the scan leaves the total behind the last entry)
        if (end - beg > heavy) {                         // one bucket with a large share of all rows (values far from any k-mer spectrum):
            if (tid == 0) *flag = 1u;                    // a single block would take it alone -> leave the rows to the full-width fallback
            for (u32 d = tid; d <= BB; d += RS_BNT) sub[(u64)b * (BB + 1) + d] = d < BB ? beg : end;
            // (the rows of this bucket still have to reach the output array: a fallback that sorts FROM it -- the per-group library
            //  sort of sort_rows_big / sort_rows_huge -- wants a complete permutation there, as k2_split leaves one)
            for (u32 i = beg + tid; i < end; i += RS_BNT) { ov[i] = v[i]; oab[i] = ab[i]; }
            continue;
        }
        for (u32 d = tid; d <= BB; d += RS_BNT) L.cnt[d] = 0;
        __syncthreads();
        for (u32 i0 = beg; i0 < end; i0 += 8 * RS_BNT) {
            u64 x[8]; bool ok[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const u32 i = i0 + tid + (u32)j * RS_BNT; ok[j] = i < end; x[j] = v[ok[j] ? i : end - 1]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (ok[j]) atomicAdd(&L.cnt[rs_dig(x[j], sp.shB, sp.mB)], 1u);
        }
        __syncthreads();
        if (tid < 64) {                              // exclusive scan over the BB counters: CPL per lane
            u32 c[CPL], s = 0;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { c[q] = L.cnt[CPL * lane + q]; s += c[q]; }
            u32 run = beg + wave_incl_scan(s) - s;
#pragma unroll
            for (int q = 0; q < CPL; ++q) { L.cur[CPL * lane + q] = run; sub[(u64)b * (BB + 1) + CPL * lane + q] = run; run += c[q]; }
            if (lane == 63) sub[(u64)b * (BB + 1) + BB] = run;
        }
        __syncthreads();
        for (u32 d = tid; d <= BB; d += RS_BNT) L.cnt[d] = 0;
        __syncthreads();
        rs_scatter_range<BB, RS_BNT>(RsDenseSrc{v, ab}, beg, end, ov, oab, sp.shB, sp.mB, L);
    }
}

__device__ __forceinline__ void rs_wave_sync() { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// insertion order of rows [o, o + m) of an LDS row area by value (one lane; m is small)
__device__ __forceinline__ bool rs_insertion(u64* rk, u32* ra, u32 o, u32 m) {      // -> two equal values met
    bool tie = false;
    for (u32 x = o + 1; x < o + m; ++x) {
        const u64 kv = rk[x]; const u32 av = ra[x];
        u32 y = x;
        while (y > o && rk[y - 1] > kv) { rk[y] = rk[y - 1]; ra[y] = ra[y - 1]; --y; }
        if (y > o && rk[y - 1] == kv) tie = true;
        rk[y] = kv; ra[y] = av;
    }
    return tie;
}

// step C: one wave per sub-bucket (2 <= nd <= RS_WAVE_ROWS rows at gk / ga, ordered in place).  The rows are placed by the third
// digit with LDS atomics (rk / ra = the wave's LDS row area, wc = its 256 + 1 cell counters); a row that shares its cell then
// counts the smaller values of the cell (all rows of the wave at once, one LDS read per step and row) and moves to its final slot.
// -> false (rows untouched) when a cell holds more than RS_WAVE_CELL_CAP rows: the caller hands the sub-bucket to k_rs_big
__device__ __forceinline__ bool rs_wave_sort(u64* gk, u32* ga, u32 nd, u64* rk, u32* ra, u32* wc, int sh, u32 m, u32* ties) {
    const u32 lane = threadIdx.x & 63;
    constexpr int RPL = RS_WAVE_ROWS / 64;
    u64 k[RPL]; u32 a[RPL], r[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t) { const u32 i = lane + 64 * t; if (i < nd) { k[t] = gk[i]; a[t] = ga[i]; } }
#pragma unroll
    for (int t = 0; t < 4; ++t) wc[lane + 64 * t] = 0;
    rs_wave_sync();
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const u32 i = lane + 64 * t;
        if (i < nd) { const u32 d = rs_dig(k[t], sh, m); r[t] = (d << 16) | atomicAdd(&wc[d], 1u); }
    }
    rs_wave_sync();
    {   // exclusive scan of the 256 cell counts (four per lane); wc[256] = nd
        u32 c4[4], s = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { c4[q] = wc[4 * lane + q]; s += c4[q]; }
        u32 run = wave_incl_scan(s) - s;
        rs_wave_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) { wc[4 * lane + q] = run; run += c4[q]; }
        if (lane == 63) wc[RS_CELLS] = run;
    }
    rs_wave_sync();
    u32 o[RPL], c[RPL], fin[RPL], at[RPL], cmax = 0;
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const u32 i = lane + 64 * t;
        o[t] = 0; c[t] = 0; fin[t] = 0; at[t] = 0;
        if (i < nd) {
            const u32 d = r[t] >> 16;
            o[t] = wc[d]; c[t] = wc[d + 1] - o[t];
            at[t] = o[t] + (r[t] & 0xFFFFu);
            rk[at[t]] = k[t]; ra[at[t]] = a[t];
            fin[t] = o[t];
            if (c[t] > 1) cmax = c[t] > cmax ? c[t] : cmax;
        }
    }
    rs_wave_sync();
    if (__ballot(cmax > RS_WAVE_CELL_CAP)) return false;      // (wave-uniform) k-mers sharing all three digits -- low-complexity sequence: a block orders them
    for (u32 j = 0; __ballot(j < cmax); ++j) {       // (equal keys -- the 63-bit prefixes of multi-word rows can tie -- keep their placement order)
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            if (c[t] > 1 && j < c[t]) {
                const u64 kk = rk[o[t] + j];
                fin[t] += (kk < k[t] || (kk == k[t] && o[t] + j < at[t])) ? 1u : 0u;
                if (kk == k[t] && o[t] + j != at[t]) *ties = 1u;          // equal keys exist (63-bit prefixes of multi-word rows): the caller's tie pass has work
            }
    }
    rs_wave_sync();
#pragma unroll
    for (int t = 0; t < RPL; ++t) if (c[t] > 1 && cmax) { rk[fin[t]] = k[t]; ra[fin[t]] = a[t]; }
    rs_wave_sync();
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const u32 i = lane + 64 * t;
        if (i < nd) { gk[i] = rk[i]; ga[i] = ra[i]; }
    }
    return true;
}

// every sub-bucket of every bucket: wave w of the grid takes sub-bucket w; the large ones go to a list for k_rs_big
#define RS_CNT 256
__global__ __launch_bounds__(RS_CNT) void k_rs_cells(u64* kv, u32* av, const u32* __restrict__ sub, u32 nsub, u32 bb, RsSpec sp,
                                                     u32* __restrict__ biglist, u32* __restrict__ nbig, u32* __restrict__ flag, u32* __restrict__ ties) {
    __shared__ u64 rk[RS_CNT / 64][RS_WAVE_ROWS];
    __shared__ u32 ra[RS_CNT / 64][RS_WAVE_ROWS];
    __shared__ u32 wc[RS_CNT / 64][RS_CELLS + 1];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 id = blockIdx.x * (RS_CNT / 64) + wave;
    if (id >= nsub) return;
    const u32 i = id + id / bb;                                    // sub[] holds bb + 1 starts per bucket
    const u32 o = sub[i], nd = sub[i + 1] - o;
    if (nd < 2) return;
    bool done = false;
    if (nd <= RS_WAVE_ROWS) done = rs_wave_sort(kv + o, av + o, nd, rk[wave], ra[wave], wc[wave], sp.shC, sp.mC, ties);
    if (!done && lane == 0) biglist[atomicAdd(nbig, 1u)] = i;
}

// sub-buckets of 257 .. block_rows rows, one block each (the same placement by the third digit, block-wide)
// A sub-bucket above block_rows rows (thousands of k-mers that share their first 13 and more bases: the error variants of a k-mer
// with 10^8 occurrences) is LISTED -- (offset in the whole row array, rows, value bits not yet used) in ovs[1 + 3 i ..], ovs[0] = how
// many -- and the host sorts each listed range once more on its remaining bits (dskgpu.hip: sort_oversize); only a full list, or
// rows with no bits left to tell them apart, raise *flag (the full-width fallback).
__global__ __launch_bounds__(RS_NT) void k_rs_big(u64* kv, u32* av, const u32* __restrict__ sub, RsSpec sp, const u32* __restrict__ biglist,
                                                  const u32* __restrict__ nbig, u32* __restrict__ flag, u32 block_rows, u32* __restrict__ ties,
                                                  u32 base, u32* __restrict__ ovs) {
    __shared__ u64 rk[RS_BLOCK_ROWS];
    __shared__ u32 ra[RS_BLOCK_ROWS];
    __shared__ u32 wc[2 * RS_CELLS];
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 nb = *nbig;
    for (u32 x = blockIdx.x; x < nb; x += gridDim.x) {
        const u32 i = biglist[x];
        const u32 o = sub[i], nd = sub[i + 1] - o;
        if (nd > block_rows) {
            if (tid == 0) {
                if (sp.shB <= 0) *ties = 1u;          // no value bits below the two digits: the rows of this sub-bucket are EQUAL keys (63-bit prefixes of multi-word
                else {                                //  rows) -- in order as far as this sort goes, the caller's tie pass does the rest
                    const u32 at = atomicAdd(&ovs[0], 1u);
                    if (at < RS_OVS_CAP) { ovs[1 + 3 * at] = base + o; ovs[2 + 3 * at] = nd; ovs[3 + 3 * at] = (u32)sp.shB; } else *flag = 1u;
                }
            }
            continue;
        }
        u64* gk = kv + o; u32* ga = av + o;
        u64 k[4]; u32 a[4], r[4];
        __syncthreads();
        if (tid < RS_CELLS) wc[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 j = tid + RS_NT * t;
            if (j < nd) { k[t] = gk[j]; a[t] = ga[j]; const u32 d = rs_dig(k[t], sp.shC, sp.mC); r[t] = (d << 16) | atomicAdd(&wc[d], 1u); }
        }
        __syncthreads();
        if (tid < 64) {
            u32 c4[4], s = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c4[q] = wc[4 * lane + q]; s += c4[q]; }
            u32 run = wave_incl_scan(s) - s;
            rs_wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) { wc[RS_CELLS + 4 * lane + q] = c4[q]; wc[4 * lane + q] = run; run += c4[q]; }      // counts kept behind the offsets
        }
        __syncthreads();
        u32 cmine = 0, omine = 0;
        if (tid < RS_CELLS) { omine = wc[tid]; cmine = wc[RS_CELLS + tid]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 j = tid + RS_NT * t;
            if (j < nd) { const u32 pos = wc[r[t] >> 16] + (r[t] & 0xFFFFu); rk[pos] = k[t]; ra[pos] = a[t]; }
        }
        // cells of a few rows: one thread orders each by insertion.  A cell above RS_BLOCK_CELL_CAP rows (k-mers that share all three
        // digits: low-complexity sequence, e.g. the error variants of poly-A) would take one thread quadratic time (100 rows: 0.1 ms): then the whole
        // sub-bucket is ordered by a bitonic network over the LDS area instead (any distribution, <= 4096 rows padded with all-ones
        // keys, which no value equals: 78 compare-exchange phases)
        if (__syncthreads_or(cmine > RS_BLOCK_CELL_CAP)) {
            u32 np2 = 2; while (np2 < nd) np2 <<= 1;
            for (u32 j = nd + tid; j < np2; j += RS_NT) { rk[j] = ~0ull; ra[j] = 0u; }
            __syncthreads();
            for (u32 kk = 2; kk <= np2; kk <<= 1)
                for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (u32 x = tid; x < np2; x += RS_NT) {
                        const u32 y = x ^ jj;
                        if (y > x) {
                            const u64 a0 = rk[x], a1 = rk[y];
                            const bool up = (x & kk) == 0;
                            if ((a0 > a1) == up) { rk[x] = a1; rk[y] = a0; const u32 t0 = ra[x]; ra[x] = ra[y]; ra[y] = t0; }
                        }
                    }
                    __syncthreads();
                }
            bool tie = false;
            for (u32 j = tid; j + 1 < nd; j += RS_NT) tie = tie || rk[j] == rk[j + 1];
            if (tie) *ties = 1u;
        } else if (cmine >= 2) { if (rs_insertion(rk, ra, omine, cmine)) *ties = 1u; }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 j = tid + RS_NT * t;
            if (j < nd) { gk[j] = rk[j]; ga[j] = ra[j]; }
        }
    }
}

// ---- another round for the listed sub-buckets (k_rs_big): the rows of ALL listed ranges are gathered into one array under the
// composite key (range number << maxbits) | (the value bits the range has not used yet, LEFT-aligned in maxbits bits: the digits right
// below the range number are then the range's own leading bits, whatever it has left), that array is ordered by ONE MSD sort, and
// the rows go back to their ranges in the new order (the ranges are disjoint and keep their places).
#define RS_OVS_SHIFT 48                   // most value bits a listed range may have left
__global__ __launch_bounds__(256) void k_ovs_gather(const u64* __restrict__ k, const u32* __restrict__ v, const u32* __restrict__ list, const u32* __restrict__ starts, u32 nr,
                                                    u64* __restrict__ gk, u32* __restrict__ gv, u64* __restrict__ prefix, u32 maxbits) {
    for (u32 r = blockIdx.x; r < nr; r += gridDim.x) {
        const u32 off = list[3 * r], len = list[3 * r + 1], bits = list[3 * r + 2], st = starts[r];
        const u64 mask = (1ull << bits) - 1ull;
        for (u32 i = threadIdx.x; i < len; i += 256) {
            const u64 kk = k[(u64)off + i];
            gk[(u64)st + i] = ((u64)r << maxbits) | ((kk & mask) << (maxbits - bits));
            gv[(u64)st + i] = v[(u64)off + i];
            if (i == 0) prefix[r] = kk & ~mask;
        }
    }
}
__global__ __launch_bounds__(256) void k_ovs_scatter(u64* __restrict__ k, u32* __restrict__ v, const u32* __restrict__ list, const u32* __restrict__ starts, u32 nr,
                                                     const u64* __restrict__ gk, const u32* __restrict__ gv, const u64* __restrict__ prefix, u32 maxbits) {
    for (u32 r = blockIdx.x; r < nr; r += gridDim.x) {
        const u32 off = list[3 * r], len = list[3 * r + 1], bits = list[3 * r + 2], st = starts[r];
        const u64 pre = prefix[r];
        for (u32 i = threadIdx.x; i < len; i += 256) {
            k[(u64)off + i] = pre | ((gk[(u64)st + i] & ((1ull << maxbits) - 1ull)) >> (maxbits - bits));
            v[(u64)off + i] = gv[(u64)st + i];
        }
    }
}
