// write_age.hip -- does the write rate of a scatter-shaped store pattern change with time under load?  Runs the 256 blocks x 512 streams
// x 256-byte-runs pattern of write_bw.hip (and a streaming fill) back to back for `seconds`, printing both every ~5 s.
#include <hip/hip_runtime.h>
#include <chrono>
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
int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 200.0;
    const u64 n = 1200000000ull;
    u64* out;
    if (hipMalloc(&out, n * 8) != hipSuccess) return 1;
    hipEvent_t a, b, c; hipEventCreate(&a); hipEventCreate(&b); hipEventCreate(&c);
    const auto t0 = std::chrono::steady_clock::now();
    double next = 0;
    for (;;) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_runs, dim3(256), dim3(1024), 0, 0, out, n, 512, 32);
        hipEventRecord(b);
        hipLaunchKernelGGL(k_fill, dim3(512), dim3(1024), 0, 0, out, n);
        hipEventRecord(c);
        hipEventSynchronize(c);
        const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (t >= next) {
            float m1, m2; hipEventElapsedTime(&m1, a, b); hipEventElapsedTime(&m2, b, c);
            printf("t %6.1f s: 512 streams x 256 B runs %.3f ms, streaming fill %.3f ms\n", t, m1, m2); fflush(stdout);
            next += 5.0;
        }
        if (t > seconds) break;
    }
    return 0;
}
