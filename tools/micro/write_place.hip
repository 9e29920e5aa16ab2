// write_place.hip -- is the rate of the scatter-shaped store pattern a property of WHERE the buffer lies?  Allocates several 9.6 GB
// buffers in one process and times the 256 blocks x 512 streams x 256-byte-runs pattern (and a streaming fill) on each, twice.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 8;
    const u64 n = 1200000000ull;
    u64* buf[32];
    for (int round = 0; round < 2; ++round) {
        for (int i = 0; i < nb; ++i) if (hipMalloc(&buf[i], n * 8) != hipSuccess) { printf("alloc %d failed\n", i); return 1; }
        for (int i = 0; i < nb; ++i) printf("round %d buffer %d at %p: runs %.3f ms, fill %.3f ms\n", round, i, (void*)buf[i], run_ms(buf[i], n, false), run_ms(buf[i], n, true));
        for (int i = 0; i < nb; ++i) hipFree(buf[i]);
    }
    return 0;
}
