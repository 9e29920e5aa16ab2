// would a table keyed by a 32-bit TAG (ds_cmpst_rtn_b32 instead of the 64-bit CAS; the other 32 bits of the key stored
// by the claiming lane and verified after the barrier, as k_count2v3 does for two-word keys) take the one-word count kernel from 3.8 to
// 3.2 ms?  Measured before building the per-sub-partition re-count it would need (~30 of 414 K sub-partitions hold two k-mers with one
// tag).  Same synthetic regions as count_sort.hip (414 000 x 2900 keys: 73 k-mers x 30 copies + singletons).
//   A   the product kernel (k_count1v3: 64-bit keys in the table)
//   T1  UPPER BOUND of the idea: the same kernel on a u32 tag table, nothing stored besides the tag, nothing verified
//       (rows carry a made-up high word: timing only)
//   T2  T1 + the claiming lane stores the high word (th[slot]) + every key reads it back after the barrier and raises a flag on a
//       mismatch -- everything the real kernel needs except the list of sub-partitions to count again
// hipcc -O3 --offload-arch=gfx950 -o count_tag count_tag.hip
#include "../../dsk_amd/csrc/kernels.h"
#include <cstdio>
#include <vector>

__global__ void k_fill(u64* keys, u32* subcnt, u32 F, u32 cap, u32 n, u32 ngen, u32 copies) {
    const u32 q = blockIdx.x;
    if (threadIdx.x == 0) subcnt[q] = n;
    for (u32 i = threadIdx.x; i < n; i += blockDim.x) {
        const u32 j = (u32)(((u64)i * 2654435761u) % n);                  // a fixed permutation of the positions
        const u32 id = j < ngen * copies ? j % ngen : 100000u + j;
        u64 x = ((u64)q << 20) | id;
        x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
        keys[(u64)q * cap + i] = x == DSK_EMPTY ? 1 : x;
    }
}

#define TAG_EMPTY 0xFFFFFFFFu
template <bool VERIFY>
__device__ __forceinline__ u32 tag_insert(u32* tg, u32* th, u32* tc, u32* ovf, u32 tag, u32 hi) {      // -> slot | claimed << 31
    u32 slot = tag & (CNT_SLOTS - 1), res = CNT_NONE;
    bool pend = true;
    for (int probe = 0; probe < CNT_SLOTS; ++probe) {
        u32 old = 0u;
        if (pend) old = tg[slot];
        const bool e = pend && old == TAG_EMPTY;
        u32 mine = 0u;
        if (e) { old = atomicCAS(&tg[slot], TAG_EMPTY, tag); if (old == TAG_EMPTY) { mine = 0x80000000u; old = tag; if (VERIFY) th[slot] = hi; } }
        const bool m = pend && old == tag;
        if (m) { atomicAdd(&tc[slot], 1u); res = slot | mine; }
        pend = pend && !m;
        slot = (slot + 1) & (CNT_SLOTS - 1);
        if (!__ballot(pend)) return res;
    }
    *ovf = 1;
    return res;
}

template <int NT, int KPT, int NKEYS, bool VERIFY>
__global__ __launch_bounds__(NT) void k_count_tag(u64* keys, u64* solid_keys, u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                  u64* __restrict__ ghist, u64* __restrict__ gstats,
                                                  u32* __restrict__ overflow, CountParams cp, const u32* __restrict__ subcnt) {
    __shared__ u32 tg[CNT_SLOTS];
    __shared__ u32 th[VERIFY ? CNT_SLOTS : 1];
    __shared__ u32 tc[CNT_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 s_ctr[2][4];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < CNT_SLOTS; s += NT) { tg[s] = TAG_EMPTY; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += NT) lh[b] = 0;
    if (tid < 8) s_ctr[tid >> 2][tid & 3] = 0;
    u32 ones = 0;
    u64 ndist_acc = 0;
    auto range_lo = [&](u32 qq) { const u32 c = qq < cp.F ? qq : cp.F - 1; return subcnt[c]; };
    auto count_of = [&](u32 qq, u32 lo) { return qq < cp.F ? ((int)lo < 0 ? 0u : lo) : 0u; };
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
        u32 at[NKEYS], hiw[NKEYS];
#pragma unroll
        for (int j = 0; j < NKEYS; ++j) { at[j] = CNT_NONE; hiw[j] = 0; }
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((u32)(tid + j * NT) < n) { hiw[j] = (u32)(pk[j] >> 32); at[j] = tag_insert<VERIFY>(tg, th, tc, &ctr[2], (u32)pk[j], hiw[j]); }
#pragma unroll
        for (int j = KPT; j < NKEYS; ++j)
            if ((u32)(tid + j * NT) < n) { const u64 kx = keys[begin + tid + j * NT]; hiw[j] = (u32)(kx >> 32); at[j] = tag_insert<VERIFY>(tg, th, tc, &ctr[2], (u32)kx, hiw[j]); }
        {
            u32 mine = 0;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) mine += (u32)__popcll(__ballot(at[j] != CNT_NONE && (at[j] >> 31)));
            if (lane == 0 && mine) atomicAdd(&ctr[0], mine);
        }
        cur = sub_of(rq, rlo);
        load_keys(cur, pk);
        rq += G; rlo = range_lo(rq);
        lds_barrier();
        if (VERIFY) {
            bool wrong = false;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) if (at[j] != CNT_NONE) wrong = wrong || th[at[j] & 0x7FFFFFFFu] != hiw[j];
            if (wrong) atomicOr(overflow, 2u);
        }
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;
        if (bad) {
            for (int s = tid; s < CNT_SLOTS; s += NT) { tg[s] = TAG_EMPTY; tc[s] = 0; }
            if (tid == 0) atomicOr(overflow, 1u);
        } else {
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) {
                const bool act = at[j] != CNT_NONE && (at[j] >> 31);
                if (!__ballot(act)) continue;
                u64 key = 0; u32 c = 0;
                if (act) {
                    const u32 slot = at[j] & 0x7FFFFFFFu;
                    key = ((u64)(VERIFY ? th[slot] : q) << 32) | tg[slot]; c = tc[slot];
                    tg[slot] = TAG_EMPTY; tc[slot] = 0;
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
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0;
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

// T3: no stored high word, no per-key verification read.  Every key adds (hi << 32 | 1) to ONE 64-bit word of its slot: low half = count, high half
// = sum of the high words of everything counted there (mod 2^32).  The lane that claimed the slot still holds ITS high word in a register at sweep
// time and checks sum == count * hi: two k-mers with one tag in a slot make the sum differ (c_B * (hi_B - hi_A) != 0 mod 2^32 as long as c_B < 2^19:
// inside a sub-partition the high words differ by less than 2^14).  Exact, one LDS atomic per key as in the product kernel -- but a 64-bit one.
__device__ __forceinline__ u32 tag_insert3(u32* tg, unsigned long long* tcs, u32* ovf, u32 tag, u32 hi) {
    u32 slot = tag & (CNT_SLOTS - 1), res = CNT_NONE;
    bool pend = true;
    const unsigned long long inc = ((unsigned long long)hi << 32) | 1ull;
    for (int probe = 0; probe < CNT_SLOTS; ++probe) {
        u32 old = 0u;
        if (pend) old = tg[slot];
        const bool e = pend && old == TAG_EMPTY;
        u32 mine = 0u;
        if (e) { old = atomicCAS(&tg[slot], TAG_EMPTY, tag); if (old == TAG_EMPTY) { mine = 0x80000000u; old = tag; } }
        const bool m = pend && old == tag;
        if (m) { atomicAdd(&tcs[slot], inc); res = slot | mine; }
        pend = pend && !m;
        slot = (slot + 1) & (CNT_SLOTS - 1);
        if (!__ballot(pend)) return res;
    }
    *ovf = 1;
    return res;
}
template <int NT, int KPT, int NKEYS>
__global__ __launch_bounds__(NT) void k_count_tag3(u64* keys, u64* solid_keys, u32* __restrict__ abund, u32* __restrict__ nsolid,
                                                   u64* __restrict__ ghist, u64* __restrict__ gstats,
                                                   u32* __restrict__ overflow, CountParams cp, const u32* __restrict__ subcnt) {
    __shared__ u32 tg[CNT_SLOTS];
    __shared__ unsigned long long tcs[CNT_SLOTS];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 s_ctr[2][4];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int s = tid; s < CNT_SLOTS; s += NT) { tg[s] = TAG_EMPTY; tcs[s] = 0ull; }
    for (int b = tid; b < CNT_LH; b += NT) lh[b] = 0;
    if (tid < 8) s_ctr[tid >> 2][tid & 3] = 0;
    u32 ones = 0;
    u64 ndist_acc = 0;
    auto range_lo = [&](u32 qq) { const u32 c = qq < cp.F ? qq : cp.F - 1; return subcnt[c]; };
    auto count_of = [&](u32 qq, u32 lo) { return qq < cp.F ? ((int)lo < 0 ? 0u : lo) : 0u; };
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
        u32 at[NKEYS], hiw[NKEYS];
#pragma unroll
        for (int j = 0; j < NKEYS; ++j) { at[j] = CNT_NONE; hiw[j] = 0; }
#pragma unroll
        for (int j = 0; j < KPT; ++j)
            if ((u32)(tid + j * NT) < n) { hiw[j] = (u32)(pk[j] >> 32); at[j] = tag_insert3(tg, tcs, &ctr[2], (u32)pk[j], hiw[j]); }
#pragma unroll
        for (int j = KPT; j < NKEYS; ++j)
            if ((u32)(tid + j * NT) < n) { const u64 kx = keys[begin + tid + j * NT]; hiw[j] = (u32)(kx >> 32); at[j] = tag_insert3(tg, tcs, &ctr[2], (u32)kx, hiw[j]); }
        {
            u32 mine = 0;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) mine += (u32)__popcll(__ballot(at[j] != CNT_NONE && (at[j] >> 31)));
            if (lane == 0 && mine) atomicAdd(&ctr[0], mine);
        }
        cur = sub_of(rq, rlo);
        load_keys(cur, pk);
        rq += G; rlo = range_lo(rq);
        lds_barrier();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;
        if (bad) {
            for (int s = tid; s < CNT_SLOTS; s += NT) { tg[s] = TAG_EMPTY; tcs[s] = 0ull; }
            if (tid == 0) atomicOr(overflow, 1u);
        } else {
            bool wrong = false;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) {
                const bool act = at[j] != CNT_NONE && (at[j] >> 31);
                if (!__ballot(act)) continue;
                u64 key = 0; u32 c = 0;
                if (act) {
                    const u32 slot = at[j] & 0x7FFFFFFFu;
                    const unsigned long long cs = tcs[slot];
                    c = (u32)cs;
                    wrong = wrong || (u32)(cs >> 32) != c * hiw[j];
                    key = ((u64)hiw[j] << 32) | tg[slot];
                    tg[slot] = TAG_EMPTY; tcs[slot] = 0ull;
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
                        solid_keys[begin + cp.cap - 1 - pos] = key;      // rows from the END of the region: the keys stay intact for a re-count
                        abund[begin + cp.cap - 1 - pos] = c;
                    }
                }
            }
            if (wrong) ctr[3] = 1u;
        }
        lds_barrier();
        if (tid == 0) {
            const bool redo = ctr[3] != 0;
            if (redo) atomicOr(overflow, 2u);
            nsolid[q] = (bad || redo) ? 0u : ctr[1];
            ndist_acc += (bad || redo) ? 0u : nd;
            ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; ctr[3] = 0;
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

int main() {
    const u32 F = 414000, cap = 4360, n = 2900, ngen = 73, copies = 30;
    u64* keys; u32* subcnt; u64* ghist; u64* gstats; u32* nsolid; u32* abund; u32* ovf;
    hipMalloc(&keys, (size_t)F * cap * 8); hipMalloc(&subcnt, (size_t)F * 4 + 64); hipMalloc(&ghist, 10001 * 8); hipMalloc(&gstats, 64);
    hipMalloc(&nsolid, (size_t)F * 4 + 64); hipMalloc(&abund, (size_t)F * cap * 4); hipMalloc(&ovf, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CountParams cp; cp.F = F; cp.amin = 2; cp.amax = 0x7fffffff; cp.histo_max = 10000; cp.maxload = CNT_MAXLOAD; cp.cap = cap; cp.subcnt = subcnt;
    auto run = [&](int which, const char* name) {
        hipLaunchKernelGGL(k_fill, dim3(F), dim3(256), 0, 0, keys, subcnt, F, cap, n, ngen, copies);
        hipMemset(ghist, 0, 10001 * 8); hipMemset(gstats, 0, 64); hipMemset(ovf, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        if (which == 0) hipLaunchKernelGGL((k_count1v3<CNT_NT, CNT_KPT, CNT_V3_KEYS>), dim3(512), dim3(CNT_NT), 0, 0, keys, keys, abund, nsolid, ghist, gstats, ovf, cp, (const u32*)subcnt);
        else if (which == 1) hipLaunchKernelGGL((k_count_tag<CNT_NT, CNT_KPT, CNT_V3_KEYS, false>), dim3(512), dim3(CNT_NT), 0, 0, keys, keys, abund, nsolid, ghist, gstats, ovf, cp, (const u32*)subcnt);
        else if (which == 2) hipLaunchKernelGGL((k_count_tag<CNT_NT, CNT_KPT, CNT_V3_KEYS, true>), dim3(512), dim3(CNT_NT), 0, 0, keys, keys, abund, nsolid, ghist, gstats, ovf, cp, (const u32*)subcnt);
        else hipLaunchKernelGGL((k_count_tag3<CNT_NT, CNT_KPT, CNT_V3_KEYS>), dim3(512), dim3(CNT_NT), 0, 0, keys, keys, abund, nsolid, ghist, gstats, ovf, cp, (const u32*)subcnt);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        u64 st[4]; u32 o; hipMemcpy(st, gstats, 32, hipMemcpyDeviceToHost); hipMemcpy(&o, ovf, 4, hipMemcpyDeviceToHost);
        printf("%-58s %7.3f ms  distinct %llu  flags %u\n", name, ms, (unsigned long long)st[0], o);
    };
    for (int rep = 0; rep < 4; ++rep) {
        run(0, "A   k_count1v3 (64-bit keys in the table)");
        run(1, "T1  u32 tag table, no verification (upper bound)");
        run(2, "T2  u32 tag table + high word stored and verified");
        run(3, "T3  u32 tag + u64 (count | sum of high words), checked in the sweep");
    }
    return 0;
}
