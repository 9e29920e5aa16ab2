// write_vmm.hip -- the scatter-shaped store pattern on buffers built with the virtual-memory API (hipMemAddressReserve /
// hipMemCreate / hipMemMap) from physical chunks of a chosen size, against hipMalloc: does chunk size / alignment change the rate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
__global__ __launch_bounds__(1024) void k_fill(u64* __restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x * 2;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * 2; i + 2 <= n; i += stride) { out[i] = i; out[i + 1] = i; }
}
__global__ __launch_bounds__(1024) void k_runs(u64* __restrict__ out, u64 n, int P, int run) {
    const u64 per_block = n / gridDim.x, per_stream = per_block / P;
    u64* base = out + (u64)blockIdx.x * per_block;
    const int runs_per_trip = 1024 / run;
    const int r = threadIdx.x / run, l = threadIdx.x % run;
    for (u64 off = 0; off + run <= per_stream; off += run)
        for (int p = r; p < P; p += runs_per_trip) base[(u64)p * per_stream + off + l] = off + l;
}
static float run_ms(u64* out, u64 n, bool fill) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int i = 0; i < 4; ++i) {
        hipEventRecord(a);
        if (fill) hipLaunchKernelGGL(k_fill, dim3(512), dim3(1024), 0, 0, out, n);
        else hipLaunchKernelGGL(k_runs, dim3(256), dim3(1024), 0, 0, out, n, 512, 32);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (i && ms < best) best = ms;
    }
    return best;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const size_t total = 12ull << 30;
    const u64 n = total / 8;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    u64* m; CK(hipMalloc(&m, total));
    printf("hipMalloc at %p: runs %.3f ms, fill %.3f ms\n", (void*)m, run_ms(m, n, false), run_ms(m, n, true));
    for (size_t chunk : {size_t(2) << 20, size_t(64) << 20, size_t(1) << 30, size_t(4) << 30, total}) {
        for (size_t align : {size_t(2) << 20, size_t(1) << 30}) {
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, total, align, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> hs;
            bool ok = true;
            for (size_t off = 0; off < total && ok; off += chunk) {
                hipMemGenericAllocationHandle_t h;
                if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { ok = false; break; }
                hs.push_back(h);
                if (hipMemMap((char*)va + off, chunk, 0, h, 0) != hipSuccess) { ok = false; break; }
            }
            hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            if (ok && hipMemSetAccess(va, total, &acc, 1) != hipSuccess) ok = false;
            if (ok) printf("chunks of %6zu MB, VA aligned %4zu MB (%p): runs %.3f ms, fill %.3f ms\n", chunk >> 20, align >> 20, va, run_ms((u64*)va, n, false), run_ms((u64*)va, n, true));
            else printf("chunks of %zu MB: failed (%s)\n", chunk >> 20, hipGetErrorString(hipGetLastError()));
            (void)hipMemUnmap(va, total);
            for (auto h : hs) (void)hipMemRelease(h);
            (void)hipMemAddressFree(va, total);
            if (chunk == (size_t(2) << 20)) break;             // (6144 chunks: once is enough)
        }
    }
    return 0;
}
