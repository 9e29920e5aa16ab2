// What would the count stage cost if a sub-partition arrived as super-k-mer RECORDS (16 bytes per ~10 k-mers) instead of 8-byte keys?
// The largest unknown of the "count straight from records" pipeline sketched in DESIGN.md section 9: level A writes records by coarse
// minimizer bin (3 GB instead of 9.6 GB of keys), level B scatters the 16-byte records by fine minimizer bucket, and THIS kernel expands a
// sub-partition's records into k-mers in registers and counts them in the LDS table of k_count1v3.
//   A  the product kernel on 8-byte keys (414 000 regions x 2810 keys: 7 genomic super-k-mers x 10 k-mers x 30 copies + 71 x 10 singletons)
//   R  the same table fed from records: 281 records per region (4.5 KB instead of 22.5 KB), staged in LDS with a slot map (slot group ->
//      record, k-mer index), every thread builds its 3 keys with a funnel shift + rev_pairs + kmix (sk_key1), then inserts as A does.
//      One more barrier per sub-partition than A (the staging), ~30 more VALU instructions per key.
// hipcc -O3 --offload-arch=gfx950 -o count_rec count_rec.hip
#include "../../dsk_amd/csrc/kernels.h"
#include "../../dsk_amd/csrc/superkmer.h"
#include <cstdio>

#define K 31
#define NREC 281
#define RCAP 320                 // records per region (capacity)
#define NPR 10                   // k-mers per record

__device__ __forceinline__ u64 mixr(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// region q: records [q * RCAP, q * RCAP + NREC): 7 segments x 30 copies + 71 singletons, in a fixed pseudo-random order
__global__ void k_fill_rec(u64* rec, u32* nrec, u32 F) {
    const u32 q = blockIdx.x;
    if (threadIdx.x == 0) nrec[q] = NREC;
    for (u32 i = threadIdx.x; i < NREC; i += blockDim.x) {
        const u32 j = (u32)(((u64)i * 2654435761u) % NREC);
        const u32 id = j < 210 ? j % 7 : 1000u + j;
        const u64 a = mixr(((u64)q << 20) | id), b = mixr(a ^ 0x9E3779B97F4A7C15ull);
        // 40 bases = 80 bits: w0 = 32 bases, top 16 bits of w1 = 8 bases; low byte of w1 = number of k-mers
        rec[((u64)q * RCAP + i) * 2] = a;
        rec[((u64)q * RCAP + i) * 2 + 1] = (b & 0xFFFF000000000000ull) | NPR;
    }
}
// the same k-mers as an 8-byte key array (for kernel A): region q: keys [q * cap, q * cap + NREC * NPR)
__global__ void k_expand_keys(const u64* rec, u64* keys, u32* subcnt, u32 cap) {
    const u32 q = blockIdx.x;
    if (threadIdx.x == 0) subcnt[q] = NREC * NPR;
    for (u32 i = threadIdx.x; i < NREC * NPR; i += blockDim.x) {
        const u64* r = rec + ((u64)q * RCAP + i / NPR) * 2;
        const u64 rr[3] = {r[0], r[1], 0ull};
        keys[(u64)q * cap + i] = sk_key1(rr, (int)(i % NPR), K);
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_count_rec(const u64* __restrict__ rec, const u32* __restrict__ nrec, u32 F, u64* __restrict__ solid_keys, u32* __restrict__ abund,
                                                  u32* __restrict__ nsolid, u64* __restrict__ ghist, u64* __restrict__ gstats, u32* __restrict__ overflow, CountParams cp) {
    constexpr int KPT = 3, NKEYS = 3;                    // 3 key slots per thread: 3072 >= 2810 (a real kernel would loop for larger sub-partitions)
    __shared__ u64 tk[CNT_SLOTS];
    __shared__ u32 tc[CNT_SLOTS];
    __shared__ u64 srec[2 * RCAP];
    __shared__ unsigned short first[NT];
    __shared__ u32 wsum[NT / 64];
    __shared__ u32 lh[CNT_LH];
    __shared__ u32 s_ctr[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int s = tid; s < CNT_SLOTS; s += NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
    for (int b = tid; b < CNT_LH; b += NT) lh[b] = 0;
    if (tid < 8) s_ctr[tid >> 2][tid & 3] = 0;
    u32 ones = 0; u64 ndist_acc = 0;
    const u32 G = gridDim.x;
    auto load_rec = [&](u32 q, u64& a, u64& b, u32& n) {           // thread t < nrec[q] holds record t of region q (prefetched a sub-partition ahead)
        const u32 nr = q < F ? nrec[q] : 0u;
        const u32 t = (u32)tid < nr ? (u32)tid : 0u;
        const u64* p = rec + ((u64)(q < F ? q : 0) * RCAP + t) * 2;
        a = p[0]; b = p[1]; n = (u32)tid < nr ? (u32)(b & 0xFFu) : 0u;
    };
    u64 ra, rb; u32 rn;
    u32 q = blockIdx.x;
    load_rec(q, ra, rb, rn);
    lds_barrier();
    int par = 0;
    while (q < F) {
        u32* ctr = s_ctr[par];
        // ---- stage the records and the slot map
        const u32 inc = wave_incl_scan(rn);
        if (lane == 63) wsum[wave] = inc;
        lds_barrier();
        u32 start = inc - rn, total = 0;
#pragma unroll
        for (int x = 0; x < NT / 64; ++x) { const u32 v = wsum[x]; if (x < wave) start += v; total += v; }
        if (rn) {
            srec[2 * tid] = ra; srec[2 * tid + 1] = rb;
            const u32 bnd = (start + KPT - 1) / KPT * KPT;
            for (u32 s = bnd; s < start + rn; s += KPT) first[s / KPT] = (unsigned short)((tid << 4) | (s - start));
        }
        // the next sub-partition's records fly under the inserts
        const u32 qn = q + G;
        load_rec(qn, ra, rb, rn);
        lds_barrier();
        // ---- every thread builds its 3 keys and inserts them
        u32 cl[NKEYS];
#pragma unroll
        for (int j = 0; j < NKEYS; ++j) cl[j] = CNT_NONE;
        const u32 slot0 = (u32)tid * KPT;
        if (slot0 < total) {
            const u32 e = first[tid];
            u32 rl = e >> 4, jj = e & 15u;
            u64 r[3] = {srec[2 * rl], srec[2 * rl + 1], 0ull};
            u32 nn = (u32)(r[1] & 0xFFu);
#pragma unroll
            for (int j = 0; j < KPT; ++j) {
                if (slot0 + j < total) {
                    if (jj == nn) { ++rl; jj = 0; r[0] = srec[2 * rl]; r[1] = srec[2 * rl + 1]; nn = (u32)(r[1] & 0xFFu); }
                    const u64 h = sk_key1(r, (int)jj, K);
                    ++jj;
                    cl[j] = table_insert3(tk, tc, &ctr[2], h);
                }
            }
        }
        {
            u32 mine = 0;
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) mine += (u32)__popcll(__ballot(cl[j] != CNT_NONE));
            if (lane == 0 && mine) atomicAdd(&ctr[0], mine);
        }
        lds_barrier();
        const u32 nd = ctr[0];
        const bool bad = ctr[2] || nd > cp.maxload;
        const u64 begin = (u64)q * cp.cap;
        if (bad) {
            for (int s = tid; s < CNT_SLOTS; s += NT) { tk[s] = DSK_EMPTY; tc[s] = 0; }
            if (tid == 0) *overflow = 1;
        } else {
#pragma unroll
            for (int j = 0; j < NKEYS; ++j) {
                const bool act = cl[j] != CNT_NONE;
                if (!__ballot(act)) continue;
                u64 key = 0; u32 c = 0;
                if (act) { const u32 slot = cl[j]; key = tk[slot]; c = tc[slot]; tk[slot] = DSK_EMPTY; tc[slot] = 0; }
                const u64 m1 = __ballot(act && c == 1);
                if (lane == 0) ones += __popcll(m1);
                if (act && c > 1) { const u32 bin = c < cp.histo_max ? c : cp.histo_max; if (bin < CNT_LH) atomicAdd(&lh[bin], 1u); else atomicAdd(&ghist[bin], 1ull); }
                const bool solid = act && c >= cp.amin && c <= cp.amax;
                const u64 ms = __ballot(solid);
                if (ms) {
                    u32 base = 0;
                    if (lane == 0) base = atomicAdd(&ctr[1], (u32)__popcll(ms));
                    base = __shfl(base, 0);
                    if (solid) { const u32 pos = base + __popcll(ms & ((1ull << lane) - 1)); solid_keys[begin + pos] = key; abund[begin + pos] = c; }
                }
            }
        }
        lds_barrier();
        if (tid == 0) { nsolid[q] = bad ? 0u : ctr[1]; ndist_acc += bad ? 0u : nd; ctr[0] = 0; ctr[1] = 0; ctr[2] = 0; }
        par ^= 1;
        q = qn;
    }
    lds_barrier();
    if (lane == 0 && ones) atomicAdd(&lh[1], ones);
    lds_barrier();
    for (int b = tid; b < CNT_LH; b += NT) { const u32 v = lh[b]; if (v) atomicAdd(&ghist[b < (int)cp.histo_max ? b : (int)cp.histo_max], (u64)v); }
    if (tid == 0 && ndist_acc) atomicAdd(&gstats[0], ndist_acc);
}

int main() {
    const u32 F = 414000, cap = 4360;
    u64* rec; u32* nrec; u64* keys; u32* subcnt; u64* ghist; u64* gstats; u32* nsolid; u32* abund; u32* ovf; u64* rows;
    hipMalloc(&rec, (size_t)F * RCAP * 16); hipMalloc(&nrec, (size_t)F * 4 + 64);
    hipMalloc(&keys, (size_t)F * cap * 8); hipMalloc(&subcnt, (size_t)F * 4 + 64); hipMalloc(&ghist, 10001 * 8); hipMalloc(&gstats, 64);
    hipMalloc(&nsolid, (size_t)F * 4 + 64); hipMalloc(&abund, (size_t)F * cap * 4); hipMalloc(&ovf, 4); hipMalloc(&rows, (size_t)F * cap * 8);
    hipLaunchKernelGGL(k_fill_rec, dim3(F), dim3(256), 0, 0, rec, nrec, F);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CountParams cp; cp.F = F; cp.amin = 2; cp.amax = 0x7fffffff; cp.histo_max = 10000; cp.maxload = CNT_MAXLOAD; cp.cap = cap; cp.subcnt = subcnt;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL(k_expand_keys, dim3(F), dim3(256), 0, 0, (const u64*)rec, keys, subcnt, cap);
        hipMemset(ghist, 0, 10001 * 8); hipMemset(gstats, 0, 64); hipMemset(ovf, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k_count1v3<CNT_NT, CNT_KPT, CNT_V3_KEYS>), dim3(512), dim3(CNT_NT), 0, 0, keys, keys, abund, nsolid, ghist, gstats, ovf, cp, (const u32*)subcnt);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        u64 st[4]; u64 h2[4]; hipMemcpy(st, gstats, 32, hipMemcpyDeviceToHost); hipMemcpy(h2, ghist + 28, 32, hipMemcpyDeviceToHost);
        printf("A  k_count1v3 on 8-byte keys (9.3 GB read)        %7.3f ms  distinct %llu  hist[30] %llu\n", ms, (unsigned long long)st[0], (unsigned long long)h2[2]);
        hipMemset(ghist, 0, 10001 * 8); hipMemset(gstats, 0, 64); hipMemset(ovf, 0, 4);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL((k_count_rec<1024>), dim3(512), dim3(1024), 0, 0, (const u64*)rec, (const u32*)nrec, F, rows, abund, nsolid, ghist, gstats, ovf, cp);
        hipEventRecord(b); hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        u32 o; hipMemcpy(st, gstats, 32, hipMemcpyDeviceToHost); hipMemcpy(h2, ghist + 28, 32, hipMemcpyDeviceToHost); hipMemcpy(&o, ovf, 4, hipMemcpyDeviceToHost);
        printf("R  the same table fed from records (1.9 GB read)  %7.3f ms  distinct %llu  hist[30] %llu  flags %u\n", ms, (unsigned long long)st[0], (unsigned long long)h2[2], o);
    }
    return 0;
}
