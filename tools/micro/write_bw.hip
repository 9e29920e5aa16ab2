// write_bw.hip -- how fast does MI355X take a WRITE-ONLY stream?  (The level-1 scatter writes 9.6 GB and reads 0.6 GB: its bound is
// the write rate of the memory system, not the 8 TB/s read+write figure.)   hipcc -O3 --offload-arch=gfx950 -o write_bw write_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
// every lane stores W consecutive 8-byte words per trip, a wave covers 64 * W * 8 contiguous bytes
template <int W> __global__ __launch_bounds__(1024) void k_fill(u64* __restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x * W;
    for (u64 i = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * W; i + W <= n; i += stride) {
#pragma unroll
        for (int x = 0; x < W; ++x) out[i + x] = i + x;
    }
}
// the same bytes as P interleaved streams per block (block b writes runs of `run` words to P regions in turn): the shape of a scatter
__global__ __launch_bounds__(1024) void k_runs(u64* __restrict__ out, u64 n, int P, int run) {
    const u64 per_block = n / gridDim.x, per_stream = per_block / P;
    u64* base = out + (u64)blockIdx.x * per_block;
    const int lanes_per_run = run;                         // one lane per word of a run
    const int runs_per_trip = 1024 / lanes_per_run;
    const int r = threadIdx.x / lanes_per_run, l = threadIdx.x % lanes_per_run;
    for (u64 off = 0; off + run <= per_stream; off += run)
        for (int p = r; p < P; p += runs_per_trip) base[(u64)p * per_stream + off + l] = off + l;
}
// read + write (copy) for reference
__global__ __launch_bounds__(1024) void k_copy(const u64* __restrict__ in, u64* __restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> float timeit(F f, int reps = 5) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main() {
    const u64 n = 1200000000ull;                           // 9.6 GB
    u64 *out, *in;
    CK(hipMalloc(&out, n * 8)); CK(hipMalloc(&in, n * 8));
    CK(hipMemset(in, 1, n * 8));
    float ms;
    ms = timeit([&] { hipMemsetAsync(out, 0, n * 8, 0); });              printf("hipMemsetAsync          : %.3f ms  %.2f TB/s written\n", ms, n * 8e-9 / ms);
    for (int grid : {256, 512, 1024, 2048}) {
        ms = timeit([&] { hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(1024), 0, 0, out, n); });  printf("fill 8 B/lane grid %4d  : %.3f ms  %.2f TB/s written\n", grid, ms, n * 8e-9 / ms);
        ms = timeit([&] { hipLaunchKernelGGL(k_fill<2>, dim3(grid), dim3(1024), 0, 0, out, n); });  printf("fill 16 B/lane grid %4d : %.3f ms  %.2f TB/s written\n", grid, ms, n * 8e-9 / ms);
    }
    ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(1024), 0, 0, in, out, n); });     printf("copy (read + write)     : %.3f ms  %.2f TB/s moved (%.2f written)\n", ms, 2 * n * 8e-9 / ms, n * 8e-9 / ms);
    for (int P : {64, 256, 512, 1024})
        for (int run : {16, 32, 64, 128}) {
            ms = timeit([&] { hipLaunchKernelGGL(k_runs, dim3(256), dim3(1024), 0, 0, out, n, P, run); });
            printf("256 blocks x %4d streams, runs of %3d words (%4d B): %.3f ms  %.2f TB/s written\n", P, run, run * 8, ms, n * 8e-9 / ms);
        }
    return 0;
}
