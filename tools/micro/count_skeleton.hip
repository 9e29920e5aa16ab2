// Read-only skeleton of the count kernels: persistent blocks walk fixed-capacity regions (cap keys each, n used), load
// the keys the way k_count1 does and do next to nothing with them.  Tells what the memory side alone allows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// MODE 0: loads one region ahead, two barriers per region (k_count1's shape); 1: same without barriers; 2: 16-byte loads, barriers
template <int NT, int KPT, int MODE>
__global__ __launch_bounds__(NT) void k(const u64* keys, u32 F, u32 cap, u32 n, u64* out) {
    __shared__ u64 sink[64];
    const int tid = threadIdx.x;
    u64 acc = 0;
    u64 pk[KPT];
    u32 q = blockIdx.x;
    auto load = [&](u32 qq) {
        const u64* base = keys + (u64)qq * cap;
        if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < KPT / 2; ++u) { const u32 i = tid + u * NT; const ulonglong2 v = reinterpret_cast<const ulonglong2*>(base)[i < n / 2 ? i : n / 2 - 1]; pk[2 * u] = v.x; pk[2 * u + 1] = v.y; }
        } else {
#pragma unroll
            for (int j = 0; j < KPT; ++j) { const u32 i = tid + j * NT; pk[j] = base[i < n ? i : n - 1]; }
        }
    };
    if (q < F) load(q);
    while (q < F) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) acc ^= pk[j] * 0x9e3779b97f4a7c15ull;
        const u32 qn = q + gridDim.x;
        if (qn < F) load(qn);
        if (MODE != 1) { lds_barrier(); if (tid < 64) sink[tid] = acc; lds_barrier(); }
        q = qn;
    }
    if (acc == 0x123456789ull) out[0] = acc + sink[tid & 63];
}
template <int NT, int KPT, int MODE>
void run(const char* name, const u64* d, u32 F, u32 cap, u32 n, int bpc, u64* out) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NT, KPT, MODE>), dim3(256 * bpc), dim3(NT), 0, 0, d, F / 8, cap, n, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NT, KPT, MODE>), dim3(256 * bpc), dim3(NT), 0, 0, d, F, cap, n, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s NT=%4d KPT=%2d x%d/CU  %7.3f ms  %6.2f TB/s (useful bytes)\n", name, NT, KPT, bpc, ms, (double)F * n * 8 / ms / 1e9);
}
int main() {
    const u32 F = 590000, cap = 4360, n = 2034;
    u64* d; hipMalloc(&d, (size_t)F * cap * 8); hipMemset(d, 1, (size_t)F * cap * 8);
    u64* out; hipMalloc(&out, 8);
    run<1024, 3, 0>("8-B loads, 1 ahead, barriers", d, F, cap, n, 2, out);
    run<1024, 3, 1>("8-B loads, 1 ahead, no barriers", d, F, cap, n, 2, out);
    run<1024, 2, 2>("16-B loads, 1 ahead, barriers", d, F, cap, n, 2, out);
    run<1024, 2, 0>("8-B loads KPT 2", d, F, cap, n, 2, out);
    run<512, 4, 0>("8-B loads", d, F, cap, n, 4, out);
    run<512, 4, 2>("16-B loads", d, F, cap, n, 4, out);
    run<256, 8, 0>("8-B loads", d, F, cap, n, 8, out);
    run<256, 8, 2>("16-B loads", d, F, cap, n, 8, out);
    run<256, 8, 0>("8-B loads", d, F, cap, n, 2, out);
    run<1024, 3, 0>("8-B loads, 1 block/CU", d, F, cap, n, 1, out);
    return 0;
}
