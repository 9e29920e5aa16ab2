// Is an LDS counting sort + run-length sweep a faster way to count a sub-partition than the hash table of k_count1v3?
// ("measure an LDS sort-and-run-length on one real sub-partition trace before calling it quadratic".)
// Same data for both: F regions of `cap` keys, n used, shaped like the bench workload's sub-partitions (50x coverage, 1 % errors:
// ~73 genomic k-mers x ~30 copies + ~700 singletons per 2900 keys).  Kernel A = the product kernel (k_count1v3 from kernels.h).
// Kernel B = a LOWER BOUND of the sort-based form: keys ranked by their 12 slot bits with one returning LDS atomic each, a scan of
// the 4096 counters, keys scattered into sorted order in LDS, then ONE linear sweep in which every key reads the first key of its
// cell and equal neighbours are counted with wave ballots -- cells that hold two distinct keys (10 % of the occupied cells) are
// simply miscounted here, i.e. the exact form can only be slower.   hipcc -O3 --offload-arch=gfx950 -o count_sort count_sort.hip
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

#define B_NT 1024
#define B_CAP 5120
__global__ __launch_bounds__(B_NT) void k_sort_count(const u64* __restrict__ keys, const u32* __restrict__ subcnt, u32 F, u32 cap, u64* __restrict__ gstats, u64* __restrict__ ghist) {
    __shared__ u32 cnt[4096 + 4];
    __shared__ u64 skey[B_CAP];
    __shared__ u32 wsum[20];
    __shared__ u32 lh[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int s = tid; s < 4096; s += B_NT) cnt[s] = 0;
    if (tid < 64) lh[tid] = 0;
    u64 ndist = 0;
    constexpr int KPT = 5;
    u64 pk[KPT];
    u32 q = blockIdx.x;
    u32 n = q < F ? subcnt[q] : 0;
#pragma unroll
    for (int j = 0; j < KPT; ++j) { const u32 i = tid + j * B_NT; pk[j] = keys[(u64)(q < F ? q : 0) * cap + (i < n ? i : (n ? n - 1 : 0))]; }
    lds_barrier();
    while (q < F) {
        u32 rk[KPT];
#pragma unroll
        for (int j = 0; j < KPT; ++j) if ((u32)(tid + j * B_NT) < n) rk[j] = atomicAdd(&cnt[(u32)pk[j] & 4095u], 1u);
        lds_barrier();
        // exclusive scan of the 4096 counters, four per thread
        u32 c4[4], s = 0;
#pragma unroll
        for (int x = 0; x < 4; ++x) { c4[x] = cnt[4 * tid + x]; s += c4[x]; }
        const u32 inc = wave_incl_scan(s);
        if (lane == 63) wsum[wave] = inc;
        lds_barrier();
        if (wave == 0) { const u32 x = lane < 16 ? wsum[lane] : 0u; const u32 y = wave_incl_scan(x); if (lane < 16) wsum[lane] = y - x; }
        lds_barrier();
        u32 run = wsum[wave] + inc - s;
#pragma unroll
        for (int x = 0; x < 4; ++x) { cnt[4 * tid + x] = run; run += c4[x]; }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < KPT; ++j) if ((u32)(tid + j * B_NT) < n) skey[cnt[(u32)pk[j] & 4095u] + rk[j]] = pk[j];
        // next sub-partition's keys fly under the sweep
        const u32 qn = q + gridDim.x;
        const u32 nn = qn < F ? subcnt[qn] : 0;
#pragma unroll
        for (int j = 0; j < KPT; ++j) { const u32 i = tid + j * B_NT; pk[j] = keys[(u64)(qn < F ? qn : 0) * cap + (i < nn ? i : (nn ? nn - 1 : 0))]; }
        lds_barrier();
        // linear sweep: position i is the head of a run when its key differs from its left neighbour's; the run's length = distance to the next head
        u32 heads = 0;
        for (u32 i0 = 0; i0 < n; i0 += B_NT) {
            const u32 i = i0 + tid;
            const bool act = i < n;
            const u64 k = act ? skey[i] : 0ull;
            const u64 left = (act && i) ? skey[i - 1] : ~0ull;
            const bool head = act && k != left;
            const u64 hm = __ballot(head);
            // run length inside the wave: distance to the next head lane (or to the end of the wave: the tail is added by the next wave's first lanes -- ignored here)
            const u64 above = hm & ~((2ull << lane) - 1ull);
            const u32 len = head ? (above ? (u32)__ffsll((unsigned long long)above) - 1u - lane : 64u - lane) : 0u;
            if (head) { atomicAdd(&lh[len < 63 ? len : 63], 1u); }
            heads += (u32)__popcll(hm);
        }
        if (lane == 0) ndist += heads;
        lds_barrier();
        for (int s2 = tid; s2 < 4096; s2 += B_NT) cnt[s2] = 0;
        lds_barrier();
        q = qn; n = nn;
    }
    if (lane == 0 && ndist) atomicAdd(&gstats[0], ndist);
    if (tid < 64 && lh[tid]) atomicAdd(&ghist[tid], (u64)lh[tid]);
}

int main() {
    const u32 F = 414000, cap = 4360, n = 2900, ngen = 73, copies = 30;
    u64* keys; u32* subcnt; u64* ghist; u64* gstats; u32* nsolid; u32* abund; u32* ovf;
    hipMalloc(&keys, (size_t)F * cap * 8); hipMalloc(&subcnt, (size_t)F * 4 + 64); hipMalloc(&ghist, 10001 * 8); hipMalloc(&gstats, 64);
    hipMalloc(&nsolid, (size_t)F * 4 + 64); hipMalloc(&abund, (size_t)F * cap * 4); hipMalloc(&ovf, 4);
    hipLaunchKernelGGL(k_fill, dim3(F), dim3(256), 0, 0, keys, subcnt, F, cap, n, ngen, copies);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CountParams cp; cp.F = F; cp.amin = 2; cp.amax = 0x7fffffff; cp.histo_max = 10000; cp.maxload = CNT_MAXLOAD; cp.cap = cap; cp.subcnt = subcnt;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(ghist, 0, 10001 * 8); hipMemset(gstats, 0, 64); hipMemset(ovf, 0, 4);
        hipEventRecord(a);
        hipLaunchKernelGGL((k_count1v3<CNT_NT, CNT_KPT, CNT_V3_KEYS>), dim3(512), dim3(CNT_NT), 0, 0, keys, keys, abund, nsolid, ghist, gstats, ovf, cp, (const u32*)subcnt);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        u64 st[4]; hipMemcpy(st, gstats, 32, hipMemcpyDeviceToHost);
        printf("A  k_count1v3 (hash table)            %7.3f ms  distinct %llu  (%.2f TB/s of keys)\n", ms, (unsigned long long)st[0], (double)F * n * 8 / ms / 1e9);
        // (k_count1v3 writes its solid rows over the keys: refill)
        hipLaunchKernelGGL(k_fill, dim3(F), dim3(256), 0, 0, keys, subcnt, F, cap, n, ngen, copies);
        hipMemset(ghist, 0, 10001 * 8); hipMemset(gstats, 0, 64);
        hipDeviceSynchronize();
        for (int bpc = 1; bpc <= 2; ++bpc) {
            hipMemset(gstats, 0, 64);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_sort_count, dim3(256 * bpc), dim3(B_NT), 0, 0, (const u64*)keys, (const u32*)subcnt, F, cap, gstats, ghist);
            hipEventRecord(b); hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
            hipMemcpy(st, gstats, 32, hipMemcpyDeviceToHost);
            printf("B  counting sort + run-length (lower bound), %d block(s)/CU  %7.3f ms  run heads %llu\n", bpc, ms, (unsigned long long)st[0]);
        }
    }
    return 0;
}
