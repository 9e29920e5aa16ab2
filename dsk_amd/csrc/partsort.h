// partsort.h -- the solid rows in the REFERENCE's order: ascending inside an output partition, partitions = classes of the hash
// (gfx950, wave64; one-word and two-word rows of a single pass).
//
// What the reference's readers see is `Partition<Count> "solid"`: dsk2ascii walks the partitions one after the other
// (utils/dsk2ascii.cpp:61,77) and prints the rows of each as they come (:85-104); gatb-core's partitions are classes of the
// minimizer hash, and only inside a partition are the rows ascending.  The global order rowsort.h produces is more than that
// contract asks for and costs three passes over the rows (4.3 x their bytes, 1.3 ms of a 13.7 ms step: VERDICT r05).  With
// DSKGPU_F_PARTITION_ORDER an output partition is a run of `qpp` consecutive hash sub-partitions of the count kernel -- at most
// PS_CAP rows --, and ONE block orders it in LDS: the rows are read once where the count kernel left them (RsSparse: the regions,
// keys still mixed) and written once, dense, partition after partition.
//
//   load   every thread takes PS_RPT (4) rows of the partition into registers (the sub-partition of a row by bisection over the LDS
//          slice of the scanned counts, as rowsort.h's RsSparseSrc), un-mixes the key
//   bins   bin = the top 12 bits of the value; rank inside the bin from an LDS atomic; one block scan turns counts into offsets
//   place  row -> LDS at offset[bin] + rank; a row that shares its bin (0.7 rows per bin on average) then counts the smaller values
//          of the bin -- the rows are distinct k-mers, so there are no ties -- and moves to its final slot
//   store  LDS -> the dense result arrays, coalesced
//
// What the block does not order itself raises *flag and the host orders ALL rows with the global sort instead (exact for any
// input): a partition above PS_CAP rows (consecutive sub-partitions far above the mean: -abundance-min 1 on a repeat family), or -- a
// test switch only, DSKGPU_PS_MAXC -- a bin above pp.maxc rows.
// Either way the block still writes ALL its rows to their dense place (unordered / in bin order): the output is a complete
// permutation of the rows whatever the flag says -- the passes of a multi-pass count lay their rows behind each other this way
// (one launch per pass, one flag for the job) and a raised flag only means "sort these dense rows globally after all".
// (the tail block of a pass may read and write the same addresses: every row is in registers or copied 1:1 before it is stored)
#pragma once
#include "rowsort.h"
#include "rowsort2.h"

#ifndef PS_NT
#define PS_NT 1024                        // (two blocks of 16 waves per CU; with 512 threads x 8 rows the same partition took 0.53 instead of 0.37 ms on the bench rows)
#define PS_RPT 4
#endif
#define PS_CAP (PS_NT * PS_RPT)           // rows a block orders (4096: 48 KB of one-word rows; two blocks per CU)
#define PS_BINS 4096
#define PS_MAXC 4096u                     // rows sharing a bin that are still ordered here: all a block holds -- a bin of c rows costs every one of them c LDS
                                          // reads (the poly-A variants of a partition share their first six bases: some hundred rows, a few microseconds)
#define PS_MAXQ 512                       // most sub-partitions per output partition (the LDS slice of the scanned counts)

struct PsParams { u32 qpp, nparts_sparse; int sh; u32 n_tail, maxc; };      // sh: value >> sh = bin (the top 12 of the 2k value bits); maxc <= PS_MAXC: rows of a bin ordered here

// One block per output partition p: sub-partitions [p * qpp, (p + 1) * qpp) of the sparse rows; block nparts_sparse (when there is a
// tail: the rows of the k-mers counted apart, dense and already un-mixed) orders the tail.  part_off[p] = first row of partition p
// in the result (part_off[nparts] = all rows).
__global__ __launch_bounds__(PS_NT) void k_part_sort(RsSparse s, const u64* __restrict__ tail_k, const u32* __restrict__ tail_v, PsParams pp,
                                                     u64* __restrict__ ov, u32* __restrict__ oab, u32* __restrict__ part_off, u32* __restrict__ flag) {
    __shared__ u64 lk[PS_CAP];
    __shared__ u32 la[PS_CAP];
    __shared__ u32 cnt[PS_BINS];                      // counts, then (offset << 16) | count
    __shared__ u32 lsoff[PS_MAXQ + 1];
    __shared__ u32 wsum[PS_NT / 64 + 1];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 p = blockIdx.x;
    const bool is_tail = p >= pp.nparts_sparse;
    const u32 q0 = is_tail ? s.F : p * pp.qpp, q1 = is_tail ? s.F : (q0 + pp.qpp < s.F ? q0 + pp.qpp : s.F), nq = q1 - q0;
    for (u32 x = tid; x <= nq; x += PS_NT) lsoff[x] = s.soff[q0 + x];
    for (u32 b = tid; b < PS_BINS; b += PS_NT) cnt[b] = 0;
    __syncthreads();
    const u32 r0 = is_tail ? s.soff[s.F] : lsoff[0];
    const u32 n = is_tail ? pp.n_tail : lsoff[nq] - r0;
    if (tid == 0) {
        part_off[p] = r0;
        if (p + 1 == gridDim.x) part_off[p + 1] = r0 + n;
    }
    if (n > PS_CAP) {                                               // (block-uniform) more rows than a block orders: hand them on as they are, dense, and say so
        if (tid == 0) *flag = 1u;
        for (u32 i = tid; i < n; i += PS_NT) {
            const u32 r = r0 + i;
            if (is_tail) { ov[(u64)r] = tail_k[i]; oab[(u64)r] = tail_v[i]; }
            else {
                u32 lo = 0, hi = nq;
                while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (lsoff[mid] <= r) lo = mid; else hi = mid; }
                const u64 src = rs_sp_base(s, q0 + lo) + (u64)(r - lsoff[lo]);
                ov[(u64)r] = kunmix(s.keys[src]); oab[(u64)r] = s.ab[src];
            }
        }
        return;
    }
    if (n == 0) return;
    u64 k[PS_RPT]; u32 a[PS_RPT], rb[PS_RPT];                       // rb = bin << 16 | rank inside the bin
#pragma unroll
    for (int j = 0; j < PS_RPT; ++j) {
        const u32 i = tid + (u32)j * PS_NT;
        const u32 r = r0 + (i < n ? i : n - 1);                     // (clamped: unconditional loads)
        if (is_tail) { k[j] = tail_k[r - r0]; a[j] = tail_v[r - r0]; }
        else {
            u32 lo = 0, hi = nq;                                    // largest x with lsoff[x] <= r
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (lsoff[mid] <= r) lo = mid; else hi = mid; }
            const u64 src = rs_sp_base(s, q0 + lo) + (u64)(r - lsoff[lo]);
            k[j] = kunmix(s.keys[src]); a[j] = s.ab[src];
        }
    }
#pragma unroll
    for (int j = 0; j < PS_RPT; ++j) {
        const u32 i = tid + (u32)j * PS_NT;
        u32 bin = (u32)(k[j] >> pp.sh); bin = bin < PS_BINS ? bin : PS_BINS - 1;
        rb[j] = i < n ? (bin << 16) | atomicAdd(&cnt[bin], 1u) : 0xFFFFFFFFu;
    }
    __syncthreads();
    {   // exclusive scan over the PS_BINS counts, PS_BINS / PS_NT per thread
        constexpr int CPT = PS_BINS / PS_NT;
        u32 c[CPT], sum = 0;
#pragma unroll
        for (int x = 0; x < CPT; ++x) { c[x] = cnt[tid * CPT + x]; sum += c[x]; }
        const u32 inc = wave_incl_scan(sum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 run = inc - sum;
#pragma unroll
        for (int x = 0; x < PS_NT / 64; ++x) { const u32 v = wsum[x]; if ((u32)x < wave) run += v; }
        bool heavy = false;
#pragma unroll
        for (int x = 0; x < CPT; ++x) { cnt[tid * CPT + x] = (run << 16) | c[x]; run += c[x]; heavy = heavy || c[x] > pp.maxc; }
        if (heavy) *flag = 1u;                                       // (the rows are still all written, a bin above PS_MAXC in the order its rows arrived)
    }
    __syncthreads();
    u32 pos[PS_RPT], cb[PS_RPT];
#pragma unroll
    for (int j = 0; j < PS_RPT; ++j) {
        pos[j] = 0; cb[j] = 0;
        if (rb[j] != 0xFFFFFFFFu) {
            cb[j] = cnt[rb[j] >> 16];
            pos[j] = (cb[j] >> 16) + (rb[j] & 0xFFFFu);
            lk[pos[j]] = k[j]; la[pos[j]] = a[j];
        }
    }
    __syncthreads();
    // rows that share their bin: final slot = first slot of the bin + number of smaller values in it (distinct k-mers: no ties)
    bool moved = false;
#pragma unroll
    for (int j = 0; j < PS_RPT; ++j) {
        const u32 c = cb[j] & 0xFFFFu;
        if (rb[j] != 0xFFFFFFFFu && c > 1u && c <= PS_MAXC) {       // (a bin above PS_MAXC rows -- flagged -- keeps the order it was placed in: the rows stay a complete permutation)
            const u32 o = cb[j] >> 16;
            u32 less = 0;
            for (u32 x = 0; x < c; ++x) less += lk[o + x] < k[j] ? 1u : 0u;
            pos[j] = o + less; moved = true;
        }
    }
    __syncthreads();
    if (moved) {
#pragma unroll
        for (int j = 0; j < PS_RPT; ++j)
            if (rb[j] != 0xFFFFFFFFu && (cb[j] & 0xFFFFu) > 1u && (cb[j] & 0xFFFFu) <= PS_MAXC) { lk[pos[j]] = k[j]; la[pos[j]] = a[j]; }
    }
    __syncthreads();
    for (u32 i = tid; i < n; i += PS_NT) { ov[(u64)r0 + i] = lk[i]; oab[(u64)r0 + i] = la[i]; }
}

// ---- the same for two-word rows (33 <= k <= 64): (hi, lo, abundance) in three arrays, PS2_CAP rows per block
#ifndef PS2_RPT
#define PS2_RPT 2
#endif
#define PS2_CAP (PS_NT * PS2_RPT)         // 2048 rows: 40 KB of rows + 16 KB of bins, two blocks per CU
__device__ __forceinline__ u32 ps2_bin(u64 hi, u64 lo, int sh) {      // the top 12 of the 2k value bits: (hi : lo) >> sh
    const u64 v = sh >= 64 ? hi >> (sh - 64) : sh == 0 ? lo : ((hi << (64 - sh)) | (lo >> sh));
    return v < PS_BINS ? (u32)v : PS_BINS - 1;
}
__global__ __launch_bounds__(PS_NT) void k_part_sort2(Rs2Sparse s, Rows2C tail, PsParams pp, Rows2 o, u32* __restrict__ part_off, u32* __restrict__ flag) {
    __shared__ u64 lh[PS2_CAP];
    __shared__ u64 ll[PS2_CAP];
    __shared__ u32 la[PS2_CAP];
    __shared__ u32 cnt[PS_BINS];
    __shared__ u32 lsoff[PS_MAXQ + 1];
    __shared__ u32 wsum[PS_NT / 64 + 1];
    const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 p = blockIdx.x;
    const bool is_tail = p >= pp.nparts_sparse;
    const u32 q0 = is_tail ? s.F : p * pp.qpp, q1 = is_tail ? s.F : (q0 + pp.qpp < s.F ? q0 + pp.qpp : s.F), nq = q1 - q0;
    for (u32 x = tid; x <= nq; x += PS_NT) lsoff[x] = s.soff[q0 + x];
    for (u32 b = tid; b < PS_BINS; b += PS_NT) cnt[b] = 0;
    __syncthreads();
    const u32 r0 = is_tail ? s.soff[s.F] : lsoff[0];
    const u32 n = is_tail ? pp.n_tail : lsoff[nq] - r0;
    if (tid == 0) {
        part_off[p] = r0;
        if (p + 1 == gridDim.x) part_off[p + 1] = r0 + n;
    }
    if (n > PS2_CAP) {
        if (tid == 0) *flag = 1u;
        for (u32 i = tid; i < n; i += PS_NT) {
            const u32 r = r0 + i;
            if (is_tail) { const u64 th = tail.hi[i], tl = tail.lo[i]; const u32 ta = tail.ab[i]; o.hi[(u64)r] = th; o.lo[(u64)r] = tl; o.ab[(u64)r] = ta; }
            else {
                u32 lo = 0, hi = nq;
                while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (lsoff[mid] <= r) lo = mid; else hi = mid; }
                const u64 src = rs2_sp_base(s, q0 + lo) + (u64)(r - lsoff[lo]);
                K2 kx = s.keys[src]; kunmixN(kx);
                o.hi[(u64)r] = kx.w[1]; o.lo[(u64)r] = kx.w[0]; o.ab[(u64)r] = s.ab[src];
            }
        }
        return;
    }
    if (n == 0) return;
    u64 kh[PS2_RPT], kl[PS2_RPT]; u32 a[PS2_RPT], rb[PS2_RPT];
#pragma unroll
    for (int j = 0; j < PS2_RPT; ++j) {
        const u32 i = tid + (u32)j * PS_NT;
        const u32 r = r0 + (i < n ? i : n - 1);
        if (is_tail) { kh[j] = tail.hi[r - r0]; kl[j] = tail.lo[r - r0]; a[j] = tail.ab[r - r0]; }
        else {
            u32 lo = 0, hi = nq;
            while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (lsoff[mid] <= r) lo = mid; else hi = mid; }
            const u64 src = rs2_sp_base(s, q0 + lo) + (u64)(r - lsoff[lo]);
            K2 kx = s.keys[src]; kunmixN(kx);
            kh[j] = kx.w[1]; kl[j] = kx.w[0]; a[j] = s.ab[src];
        }
    }
#pragma unroll
    for (int j = 0; j < PS2_RPT; ++j) {
        const u32 i = tid + (u32)j * PS_NT;
        const u32 bin = ps2_bin(kh[j], kl[j], pp.sh);
        rb[j] = i < n ? (bin << 16) | atomicAdd(&cnt[bin], 1u) : 0xFFFFFFFFu;
    }
    __syncthreads();
    {
        constexpr int CPT = PS_BINS / PS_NT;
        u32 c[CPT], sum = 0;
#pragma unroll
        for (int x = 0; x < CPT; ++x) { c[x] = cnt[tid * CPT + x]; sum += c[x]; }
        const u32 inc = wave_incl_scan(sum);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 run = inc - sum;
#pragma unroll
        for (int x = 0; x < PS_NT / 64; ++x) { const u32 v = wsum[x]; if ((u32)x < wave) run += v; }
        bool heavy = false;
#pragma unroll
        for (int x = 0; x < CPT; ++x) { cnt[tid * CPT + x] = (run << 16) | c[x]; run += c[x]; heavy = heavy || c[x] > pp.maxc; }
        if (heavy) *flag = 1u;
    }
    __syncthreads();
    u32 pos[PS2_RPT], cb[PS2_RPT];
#pragma unroll
    for (int j = 0; j < PS2_RPT; ++j) {
        pos[j] = 0; cb[j] = 0;
        if (rb[j] != 0xFFFFFFFFu) {
            cb[j] = cnt[rb[j] >> 16];
            pos[j] = (cb[j] >> 16) + (rb[j] & 0xFFFFu);
            lh[pos[j]] = kh[j]; ll[pos[j]] = kl[j]; la[pos[j]] = a[j];
        }
    }
    __syncthreads();
    bool moved = false;
#pragma unroll
    for (int j = 0; j < PS2_RPT; ++j) {
        const u32 c = cb[j] & 0xFFFFu;
        if (rb[j] != 0xFFFFFFFFu && c > 1u && c <= PS_MAXC) {
            const u32 ob = cb[j] >> 16;
            u32 less = 0;
            for (u32 x = 0; x < c; ++x) { const u64 h = lh[ob + x], l = ll[ob + x]; less += (h < kh[j] || (h == kh[j] && l < kl[j])) ? 1u : 0u; }
            pos[j] = ob + less; moved = true;
        }
    }
    __syncthreads();
    if (moved) {
#pragma unroll
        for (int j = 0; j < PS2_RPT; ++j)
            if (rb[j] != 0xFFFFFFFFu && (cb[j] & 0xFFFFu) > 1u && (cb[j] & 0xFFFFu) <= PS_MAXC) { lh[pos[j]] = kh[j]; ll[pos[j]] = kl[j]; la[pos[j]] = a[j]; }
    }
    __syncthreads();
    for (u32 i = tid; i < n; i += PS_NT) { o.hi[(u64)r0 + i] = lh[i]; o.lo[(u64)r0 + i] = ll[i]; o.ab[(u64)r0 + i] = la[i]; }
}
