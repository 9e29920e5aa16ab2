// rowsort2.h -- ascending order of two-word solid rows (k = 33..64: value = hi:lo, 2k bits; abundance), hand-written for gfx950.
//
// The same three MSD steps as rowsort.h (A: 1024 buckets on the top 10 value bits through exact chunk x bin offsets; B: every bucket
// split on the next 8-10 bits by one block at a time; C: one wave per sub-bucket, rows placed by the next 8 bits and ordered by
// comparison inside a cell; sub-buckets of 513..4096 rows by whole blocks) -- but the ROWS THEMSELVES move through them, as three
// arrays (hi, lo, abundance), and every comparison is on the whole 128-bit value.  Before (r02-r04a) two-word rows were sorted as
// (top 63 value bits, row index) pairs, gathered by index (32-byte records: 128-byte fetches, 4 x the bytes) and the rows that share
// those 63 bits -- millions at 50 x coverage: a solid error variant and its parent -- ordered by a separate tie pass.  Here there are
// no indices, no gather and no ties (rows are distinct k-mers).  20 bytes per row: tiles hold half the rows of the one-word sort.
// What the kernels do not order themselves: a first-digit bucket above `heavy` rows raises *flag (full-width library fallback in
// run_pipeline); a sub-bucket above RS_BLOCK_ROWS rows is LISTED (offset, rows, value bits not used yet) and the host runs the same
// sort again on that range with the remaining bits (dskgpu.hip: msd_sort_rows2, a handful of ranges at most on real reads: the
// error variants of poly-A share 9 and more leading bases).
#pragma once
#include "rowsort.h"

#ifndef RS2_ABITS
#define RS2_ABITS 11                      // step A: 2048 buckets (r05: 1.91-1.97 ms against 2.42-2.53 with the one-word sort's 1024 on the k = 63 bench rows --
#endif                                    //   steps B and C work on 2048- / 256-row units here, half the one-word sort's: half-size buckets suit them)
#define RS2_ABINS (1 << RS2_ABITS)
#define RS2_RPT 4                         // rows per thread and tile (20 bytes per row: 4096-row tiles in step A, 2048 in step B)
#define RS2_TILE (RS_NT * RS2_RPT)
#define RS2_BTILE (RS_BNT * RS2_RPT)
#define RS2_OVS_CAP 64                    // listed sub-buckets per sort call
#ifndef RS2_WAVE_ROWS
#define RS2_WAVE_ROWS 256                  // step C: a wave orders sub-buckets up to this size alone (4 rows per lane: half the registers and LDS of 512 --
#endif                                    //  sub-buckets average 140 rows; larger ones go to the block path)

// digit of the 128-bit value hi:lo at bit `sh` (wave-uniform)
__device__ __forceinline__ u32 rs2_dig(u64 hi, u64 lo, int sh, u32 m) {
    const u64 v = sh >= 64 ? hi >> (sh - 64) : sh == 0 ? lo : (lo >> sh) | (hi << (64 - sh));
    return (u32)v & m;
}
__device__ __forceinline__ bool rs2_less(u64 ah, u64 al, u64 bh, u64 bl) { return ah < bh || (ah == bh && al < bl); }

template <int P, int TILE>
struct Rs2Lds {
    u64* shi; u64* slo; u32* sab; u32* cnt; u32* off; u32* cur; u32* delta; u32* wsum; u32* tot;
    static constexpr size_t bytes = (size_t)TILE * 20 + (size_t)(P + 1) * 8 + (size_t)P * 8 + 17 * 4 + 16;
    __device__ __forceinline__ explicit Rs2Lds(char* smem) {
        shi = reinterpret_cast<u64*>(smem);
        slo = shi + TILE;
        sab = reinterpret_cast<u32*>(slo + TILE);
        cnt = sab + TILE;                        // bins + 1 (dummy bin: slots past the end of the range)
        off = cnt + (P + 1);
        cur = off + (P + 1);
        delta = cur + P;
        wsum = delta + P;                        // 16
        tot = wsum + 16;
    }
};

struct Rows2 { u64* hi; u64* lo; u32* ab; };
struct Rows2C { const u64* hi; const u64* lo; const u32* ab; };

// per-chunk histogram of the first digit -> matrix[bin * nch + chunk]
// (c0: first matrix column of this launch, as in k_rs_hist)
__global__ __launch_bounds__(RS_NT) void k2_hist(Rows2C v, u64 n, u32 chunk, u32 nch, u32* __restrict__ matrix, RsSpec sp, u32 c0 = 0) {
    __shared__ u32 lh[RS2_ABINS];
    const u32 c = blockIdx.x;
    for (u32 b = threadIdx.x; b < RS2_ABINS; b += RS_NT) lh[b] = 0;
    __syncthreads();
    const u64 beg = (u64)c * chunk;
    const u64 end = beg + chunk < n ? beg + chunk : n;
    const bool need_lo = sp.shA < 64;                                 // (k = 63: the first digit lies in hi alone)
    for (u64 i0 = beg; i0 < end; i0 += 8 * RS_NT) {
        u64 xh[8], xl[8]; bool ok[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const u64 i = i0 + threadIdx.x + (u64)j * RS_NT; ok[j] = i < end; const u64 s = ok[j] ? i : end - 1; xh[j] = v.hi[s]; xl[j] = need_lo ? v.lo[s] : 0ull; }
#pragma unroll
        for (int j = 0; j < 8; ++j) if (ok[j]) atomicAdd(&lh[rs2_dig(xh[j], xl[j], sp.shA, sp.mA)], 1u);
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < RS2_ABINS; b += RS_NT) matrix[(u64)b * nch + c0 + c] = lh[b];
}

// ---- step A straight from the count kernel's output (two-word rows of a single pass; rowsort.h has the one-word twin): the solid rows
// of sub-partition q lie where k_count2v3 / k_count_mw left them -- ns_q = soff[q + 1] - soff[q] two-word keys (still MIXED) from key
// index base(q) on, abundances at the same index of `ab`.  k_compact<2> gathered them into three dense arrays first (0.26 ms at k = 63).
struct Rs2Sparse { const K2* keys; const u32* ab; const u32* soff; const u32* fstart; u32 cap, F, qpc; };
__device__ __forceinline__ u64 rs2_sp_base(const Rs2Sparse& s, u32 q) { return s.cap ? (u64)q * s.cap : (u64)s.fstart[q]; }
__global__ __launch_bounds__(RS_NT) void k2_hist_sp(Rs2Sparse s, u32 nch, u32* __restrict__ matrix, RsSpec sp) {
    __shared__ u32 lh[RS2_ABINS];
    const u32 c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u32 b = threadIdx.x; b < RS2_ABINS; b += RS_NT) lh[b] = 0;
    __syncthreads();
    const u32 q0 = c * s.qpc, q1 = q0 + s.qpc < s.F ? q0 + s.qpc : s.F;
    for (u32 q = q0 + wave; q < q1; q += RS_NT / 64) {
        const u32 o = s.soff[q], ns = s.soff[q + 1] - o;
        const u64 b = rs2_sp_base(s, q);
        for (u32 i = lane; i < ns; i += 64) { K2 kx = s.keys[b + i]; kunmixN(kx); atomicAdd(&lh[rs2_dig(kx.w[1], kx.w[0], sp.shA, sp.mA)], 1u); }
    }
    __syncthreads();
    for (u32 b = threadIdx.x; b < RS2_ABINS; b += RS_NT) matrix[(u64)b * nch + c] = lh[b];
}
// where a tile's rows come from: three dense arrays, or the sparse regions (lsoff = the chunk's slice of soff in LDS, as in RsSparseSrc)
struct Rs2DenseSrc {
    Rows2C v;
    __device__ __forceinline__ void get(u64 r, u64& h, u64& l, u32& a) const { h = v.hi[r]; l = v.lo[r]; a = v.ab[r]; }
};
struct Rs2SparseSrc {
    Rs2Sparse s; const u32* lsoff; u32 q0, nq;
    __device__ __forceinline__ void get(u64 r, u64& h, u64& l, u32& a) const {
        u32 lo = 0, hi = nq;
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((u64)lsoff[mid] <= r) lo = mid; else hi = mid; }
        const u64 src = rs2_sp_base(s, q0 + lo) + (r - (u64)lsoff[lo]);
        K2 kx = s.keys[src]; kunmixN(kx);
        h = kx.w[1]; l = kx.w[0]; a = s.ab[src];
    }
};

// rows [beg, end) -> their bins, through LDS-staged tiles of NT * RS2_RPT rows (the structure of rs_scatter_range)
template <int P, int NT, bool HUGE = false, class Src = Rs2DenseSrc>
__device__ __forceinline__ void rs2_scatter_range(const Src v, u64 beg, u64 end, Rows2 o, int sh, u32 m, const Rs2Lds<P, NT * RS2_RPT>& L,
                                                  const u64* __restrict__ gdel = nullptr) {      // (gdel: see rs_scatter_range)
    constexpr u32 TILE = NT * RS2_RPT;
    const u32 tid = threadIdx.x;
    u64 kh[RS2_RPT], kl[RS2_RPT], nh[RS2_RPT], nl[RS2_RPT]; u32 aa[RS2_RPT], an[RS2_RPT];
    auto load = [&](u64 t0, u64 (&h)[RS2_RPT], u64 (&l)[RS2_RPT], u32 (&a)[RS2_RPT]) {
        const u64 left = end - t0;
        const u32 n = left < (u64)TILE ? (u32)left : TILE;
#pragma unroll
        for (int j = 0; j < RS2_RPT; ++j) { const u32 i = tid + (u32)j * NT; v.get(t0 + (i < n ? i : n - 1), h[j], l[j], a[j]); }
    };
    if (beg < end) load(beg, kh, kl, aa);
    for (u64 t0 = beg; t0 < end; t0 += TILE) {
        const u64 left = end - t0;
        const u32 n = left < (u64)TILE ? (u32)left : TILE;
        const bool more = t0 + TILE < end;
        if (more) load(t0 + TILE, nh, nl, an);
        u32 rk[RS2_RPT];
#pragma unroll
        for (int j = 0; j < RS2_RPT; ++j) rk[j] = (tid + (u32)j * NT < n ? rs2_dig(kh[j], kl[j], sh, m) : (u32)P) << 16;
#pragma unroll
        for (int j = 0; j < RS2_RPT; ++j) rk[j] |= atomicAdd(&L.cnt[rk[j] >> 16], 1u);
        lds_barrier();
        tile_scan<NT>(L.cnt, L.off, L.delta, L.cur, P, L.wsum, L.tot);
        lds_barrier();
#pragma unroll
        for (int j = 0; j < RS2_RPT; ++j) {
            const u32 pos = L.off[rk[j] >> 16] + (rk[j] & 0xFFFFu);
            L.shi[pos] = kh[j]; L.slo[pos] = kl[j]; L.sab[pos] = aa[j];
        }
        if (tid == 0) L.cnt[P] = 0;
        lds_barrier();
        const u32 ntile = *L.tot;
#pragma unroll
        for (int j = 0; j < RS2_RPT; ++j) {
            const u32 i = tid + (u32)j * NT;
            if (i < ntile) {
                const u64 h = L.shi[i], l = L.slo[i];
                const u32 dg = rs2_dig(h, l, sh, m);
                const u32 dst = L.delta[dg] + i;
                if constexpr (HUGE) { const u64 d64 = (u64)dst + gdel[dg]; o.hi[d64] = h; o.lo[d64] = l; o.ab[d64] = L.sab[i]; }
                else { o.hi[dst] = h; o.lo[dst] = l; o.ab[dst] = L.sab[i]; }
            }
        }
        if (more) {
#pragma unroll
            for (int j = 0; j < RS2_RPT; ++j) { kh[j] = nh[j]; kl[j] = nl[j]; aa[j] = an[j]; }
        }
    }
    lds_barrier();
}

// step A: chunk c scatters its rows to the 1024 buckets (offsets from the scanned matrix)
template <bool HUGE = false>
__global__ __launch_bounds__(RS_NT) void k2_scatter(Rows2C v, u64 n, u32 chunk, u32 nch, const u32* __restrict__ scanned, Rows2 o, RsSpec sp,
                                                    const u64* __restrict__ gdel, u32 c0 = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Rs2Lds<RS2_ABINS, RS2_TILE> L(smem);
    const u32 c = blockIdx.x;
    for (u32 b = threadIdx.x; b < RS2_ABINS; b += RS_NT) { L.cur[b] = scanned[(u64)b * nch + c0 + c]; L.cnt[b] = 0; }
    if (threadIdx.x == 0) L.cnt[RS2_ABINS] = 0;
    lds_barrier();
    const u64 beg = (u64)c * chunk;
    const u64 end = beg + chunk < n ? beg + chunk : n;
    rs2_scatter_range<RS2_ABINS, RS_NT, HUGE>(Rs2DenseSrc{v}, beg, end, o, sp.shA, sp.mA, L, gdel);
}
// the same from the sparse regions: chunk c = sub-partitions [c * qpc, ..); its slice of soff sits behind the scatter's LDS
__global__ __launch_bounds__(RS_NT) void k2_scatter_sp(Rs2Sparse s, u32 nch, const u32* __restrict__ scanned, Rows2 o, RsSpec sp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Rs2Lds<RS2_ABINS, RS2_TILE> L(smem);
    u32* lsoff = reinterpret_cast<u32*>(smem + Rs2Lds<RS2_ABINS, RS2_TILE>::bytes);
    const u32 c = blockIdx.x;
    const u32 q0 = c * s.qpc, q1 = q0 + s.qpc < s.F ? q0 + s.qpc : s.F, nq = q1 > q0 ? q1 - q0 : 0u;
    for (u32 b = threadIdx.x; b < RS2_ABINS; b += RS_NT) { L.cur[b] = scanned[(u64)b * nch + c]; L.cnt[b] = 0; }
    for (u32 x = threadIdx.x; x <= nq; x += RS_NT) lsoff[x] = s.soff[q0 + x];
    if (threadIdx.x == 0) L.cnt[RS2_ABINS] = 0;
    lds_barrier();
    if (nq == 0) return;
    rs2_scatter_range<RS2_ABINS, RS_NT, false, Rs2SparseSrc>(Rs2SparseSrc{s, lsoff, q0, nq}, (u64)lsoff[0], (u64)lsoff[nq], o, sp.shA, sp.mA, L);
}

// step B: a block splits one bucket at a time into BB sub-buckets on the second digit; starts (row indices) to sub[b * (BB + 1) ..]
template <int BB>
__global__ __launch_bounds__(RS_BNT) void k2_split(Rows2C v, u32 nch, const u32* __restrict__ scanned, Rows2 o, u32* __restrict__ sub, RsSpec sp,
                                                   u32* __restrict__ work, u32 heavy, u32* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const Rs2Lds<BB, RS2_BTILE> L(smem);
    u32& s_b = L.tot[1];
    const u32 tid = threadIdx.x, lane = tid & 63;
    constexpr int CPL = BB / 64;
    const bool need_lo = sp.shB < 64;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_b = atomicAdd(work, 1u);
        __syncthreads();
        const u32 b = s_b;
        if (b >= RS2_ABINS) break;
        const u32 beg = scanned[(u64)b * nch], end = scanned[(u64)(b + 1) * nch];
        if (end - beg > heavy) {
            if (tid == 0) *flag = 1u;
            for (u32 d = tid; d <= BB; d += RS_BNT) sub[(u64)b * (BB + 1) + d] = d < BB ? beg : end;
            // (the rows of this bucket still have to reach the output array: the fallback wants a complete permutation)
            for (u32 i = beg + tid; i < end; i += RS_BNT) { o.hi[i] = v.hi[i]; o.lo[i] = v.lo[i]; o.ab[i] = v.ab[i]; }
            continue;
        }
        for (u32 d = tid; d <= BB; d += RS_BNT) L.cnt[d] = 0;
        __syncthreads();
        for (u32 i0 = beg; i0 < end; i0 += 8 * RS_BNT) {
            u64 xh[8], xl[8]; bool ok[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const u32 i = i0 + tid + (u32)j * RS_BNT; ok[j] = i < end; const u32 s = ok[j] ? i : end - 1; xh[j] = v.hi[s]; xl[j] = need_lo ? v.lo[s] : 0ull; }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (ok[j]) atomicAdd(&L.cnt[rs2_dig(xh[j], xl[j], sp.shB, sp.mB)], 1u);
        }
        __syncthreads();
        if (tid < 64) {
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
        rs2_scatter_range<BB, RS_BNT>(Rs2DenseSrc{v}, beg, end, o, sp.shB, sp.mB, L);
    }
}

// step C: one wave per sub-bucket (2 <= nd <= RS2_WAVE_ROWS rows, ordered in place): placed by the third digit, a row that shares its
// cell counts the smaller rows of the cell (whole value) and moves to its final slot.  -> false when a cell holds more than
// RS_WAVE_CELL_CAP rows (the caller hands the sub-bucket to k2_big)
__device__ __forceinline__ bool rs2_wave_sort(u64* gh, u64* gl, u32* ga, u32 nd, u64* rh, u64* rl, u32* ra, u32* wc, int sh, u32 m) {
    const u32 lane = threadIdx.x & 63;
    constexpr int RPL = RS2_WAVE_ROWS / 64;
    u64 h[RPL], l[RPL]; u32 a[RPL], r[RPL];
#pragma unroll
    for (int t = 0; t < RPL; ++t) { const u32 i = lane + 64 * t; if (i < nd) { h[t] = gh[i]; l[t] = gl[i]; a[t] = ga[i]; } }
#pragma unroll
    for (int t = 0; t < 4; ++t) wc[lane + 64 * t] = 0;
    rs_wave_sync();
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const u32 i = lane + 64 * t;
        if (i < nd) { const u32 d = rs2_dig(h[t], l[t], sh, m); r[t] = (d << 16) | atomicAdd(&wc[d], 1u); }
    }
    rs_wave_sync();
    {
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
    u32 o[RPL], c[RPL], fin[RPL], cmax = 0;
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const u32 i = lane + 64 * t;
        o[t] = 0; c[t] = 0; fin[t] = 0;
        if (i < nd) {
            const u32 d = r[t] >> 16;
            o[t] = wc[d]; c[t] = wc[d + 1] - o[t];
            const u32 at = o[t] + (r[t] & 0xFFFFu);
            rh[at] = h[t]; rl[at] = l[t]; ra[at] = a[t];
            fin[t] = o[t];
            if (c[t] > 1) cmax = c[t] > cmax ? c[t] : cmax;
        }
    }
    rs_wave_sync();
    if (__ballot(cmax > RS_WAVE_CELL_CAP)) return false;
    for (u32 j = 0; __ballot(j < cmax); ++j) {
#pragma unroll
        for (int t = 0; t < RPL; ++t)
            if (c[t] > 1 && j < c[t]) fin[t] += rs2_less(rh[o[t] + j], rl[o[t] + j], h[t], l[t]) ? 1u : 0u;      // (rows are distinct: no ties)
    }
    rs_wave_sync();
#pragma unroll
    for (int t = 0; t < RPL; ++t) if (c[t] > 1 && cmax) { rh[fin[t]] = h[t]; rl[fin[t]] = l[t]; ra[fin[t]] = a[t]; }
    rs_wave_sync();
#pragma unroll
    for (int t = 0; t < RPL; ++t) {
        const u32 i = lane + 64 * t;
        if (i < nd) { gh[i] = rh[i]; gl[i] = rl[i]; ga[i] = ra[i]; }
    }
    return true;
}

__global__ __launch_bounds__(RS_CNT) void k2_cells(Rows2 v, const u32* __restrict__ sub, u32 nsub, u32 bb, RsSpec sp, u32* __restrict__ biglist, u32* __restrict__ nbig) {
    __shared__ u64 rh[RS_CNT / 64][RS2_WAVE_ROWS];
    __shared__ u64 rl[RS_CNT / 64][RS2_WAVE_ROWS];
    __shared__ u32 ra[RS_CNT / 64][RS2_WAVE_ROWS];
    __shared__ u32 wc[RS_CNT / 64][RS_CELLS + 1];
    const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32 id = blockIdx.x * (RS_CNT / 64) + wave;
    if (id >= nsub) return;
    const u32 i = id + id / bb;
    const u32 o = sub[i], nd = sub[i + 1] - o;
    if (nd < 2) return;
    bool done = false;
    if (nd <= RS2_WAVE_ROWS) done = rs2_wave_sort(v.hi + o, v.lo + o, v.ab + o, nd, rh[wave], rl[wave], ra[wave], wc[wave], sp.shC, sp.mC);
    if (!done && lane == 0) biglist[atomicAdd(nbig, 1u)] = i;
}

// sub-buckets of up to block_rows rows, one block each; larger ones are listed in ovs (ovs[0] = how many; offset + base, rows, bits
// below the second digit) for another round of the whole sort on their remaining bits
__global__ __launch_bounds__(RS_NT) void k2_big(Rows2 v, const u32* __restrict__ sub, RsSpec sp, const u32* __restrict__ biglist, const u32* __restrict__ nbig,
                                                u32* __restrict__ flag, u32 block_rows, u32 base, u32* __restrict__ ovs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u64* rh = reinterpret_cast<u64*>(smem);
    u64* rl = rh + RS_BLOCK_ROWS;
    u32* ra = reinterpret_cast<u32*>(rl + RS_BLOCK_ROWS);
    u32* wc = ra + RS_BLOCK_ROWS;                          // 2 * RS_CELLS
    const u32 tid = threadIdx.x, lane = tid & 63;
    const u32 nb = *nbig;
    for (u32 x = blockIdx.x; x < nb; x += gridDim.x) {
        const u32 i = biglist[x];
        const u32 o = sub[i], nd = sub[i + 1] - o;
        if (nd > block_rows) {
            if (tid == 0) {
                // (distinct rows with no value bits left below the two digits cannot exist; shB <= 0 here would mean a broken digit plan)
                const u32 at = sp.shB > 0 ? atomicAdd(&ovs[0], 1u) : RS2_OVS_CAP;
                if (at < RS2_OVS_CAP) { ovs[1 + 3 * at] = base + o; ovs[2 + 3 * at] = nd; ovs[3 + 3 * at] = (u32)sp.shB; } else *flag = 1u;
            }
            continue;
        }
        u64* gh = v.hi + o; u64* gl = v.lo + o; u32* ga = v.ab + o;
        u64 h[4], l[4]; u32 a[4], r[4];
        __syncthreads();
        if (tid < RS_CELLS) wc[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 j = tid + RS_NT * t;
            if (j < nd) { h[t] = gh[j]; l[t] = gl[j]; a[t] = ga[j]; const u32 d = rs2_dig(h[t], l[t], sp.shC, sp.mC); r[t] = (d << 16) | atomicAdd(&wc[d], 1u); }
        }
        __syncthreads();
        if (tid < 64) {
            u32 c4[4], s = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) { c4[q] = wc[4 * lane + q]; s += c4[q]; }
            u32 run = wave_incl_scan(s) - s;
            rs_wave_sync();
#pragma unroll
            for (int q = 0; q < 4; ++q) { wc[RS_CELLS + 4 * lane + q] = c4[q]; wc[4 * lane + q] = run; run += c4[q]; }
        }
        __syncthreads();
        u32 cmine = 0, omine = 0;
        if (tid < RS_CELLS) { omine = wc[tid]; cmine = wc[RS_CELLS + tid]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 j = tid + RS_NT * t;
            if (j < nd) { const u32 pos = wc[r[t] >> 16] + (r[t] & 0xFFFFu); rh[pos] = h[t]; rl[pos] = l[t]; ra[pos] = a[t]; }
        }
        if (__syncthreads_or(cmine > RS_BLOCK_CELL_CAP)) {          // a crowded cell: bitonic network over the whole sub-bucket (pads: all-ones, above every value)
            u32 np2 = 2; while (np2 < nd) np2 <<= 1;
            for (u32 j = nd + tid; j < np2; j += RS_NT) { rh[j] = ~0ull; rl[j] = ~0ull; ra[j] = 0u; }
            __syncthreads();
            for (u32 kk = 2; kk <= np2; kk <<= 1)
                for (u32 jj = kk >> 1; jj > 0; jj >>= 1) {
                    for (u32 xx = tid; xx < np2; xx += RS_NT) {
                        const u32 y = xx ^ jj;
                        if (y > xx) {
                            const u64 h0 = rh[xx], l0 = rl[xx], h1 = rh[y], l1 = rl[y];
                            const bool up = (xx & kk) == 0;
                            if (rs2_less(h1, l1, h0, l0) == up) { rh[xx] = h1; rl[xx] = l1; rh[y] = h0; rl[y] = l0; const u32 t0 = ra[xx]; ra[xx] = ra[y]; ra[y] = t0; }
                        }
                    }
                    __syncthreads();
                }
        } else if (cmine >= 2) {                               // one thread per small cell: insertion
            for (u32 xx = omine + 1; xx < omine + cmine; ++xx) {
                const u64 hv = rh[xx], lv = rl[xx]; const u32 av = ra[xx];
                u32 y = xx;
                while (y > omine && rs2_less(hv, lv, rh[y - 1], rl[y - 1])) { rh[y] = rh[y - 1]; rl[y] = rl[y - 1]; ra[y] = ra[y - 1]; --y; }
                rh[y] = hv; rl[y] = lv; ra[y] = av;
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 j = tid + RS_NT * t;
            if (j < nd) { gh[j] = rh[j]; gl[j] = rl[j]; ga[j] = ra[j]; }
        }
    }
}
