// write_align.hip -- the scatter-shaped store pattern (256 blocks x P streams x runs) with the stream bases aligned to A bytes and
// every run starting `shift` bytes past an A-boundary: what does run ALIGNMENT cost?  9.6 GB written per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
// stream p of block b starts at word (b * per_block + p * per_stream) rounded down to `aw` words, + shiftw words
__global__ __launch_bounds__(1024) void k_runs(u64* __restrict__ out, u64 n, int P, int run, u64 aw, u64 shiftw) {
    const u64 per_block = n / gridDim.x, per_stream = per_block / P;
    const int runs_per_trip = 1024 / run;
    const int r = threadIdx.x / run, l = threadIdx.x % run;
    for (u64 off = 0; off + run + aw <= per_stream; off += run)
        for (int p = r; p < P; p += runs_per_trip) {
            const u64 base = ((u64)blockIdx.x * per_block + (u64)p * per_stream) / aw * aw + shiftw;
            out[base + off + l] = off + l;
        }
}
static float run_ms(u64* out, u64 n, int P, int run, u64 aw, u64 shiftw) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int i = 0; i < 4; ++i) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k_runs, dim3(256), dim3(1024), 0, 0, out, n, P, run, aw, shiftw);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (i && ms < best) best = ms;
    }
    return best;
}
int main() {
    const u64 n = 1200000000ull;
    u64* out; if (hipMalloc(&out, n * 8 + (1 << 20)) != hipSuccess) return 1;
    for (int P : {512, 256}) for (int run : {32, 16, 64}) {
        printf("P %d, runs of %d B:", P, run * 8);
        for (u64 a : {1ull, 8ull, 16ull, 32ull, 512ull}) printf("  base %% %llu B: %.3f ms", a * 8, run_ms(out, n, P, run, a, 0));
        printf("  | 256 B-aligned + 64 B: %.3f ms, + 128 B: %.3f ms\n", run_ms(out, n, P, run, 32, 8), run_ms(out, n, P, run, 32, 16));
    }
    return 0;
}
