// write_map.hip -- WHERE inside a buffer is the scatter-shaped store pattern slow?  Of several 12 GB allocations the fastest and the
// slowest (timed as write_place.hip does) are mapped in 32 MB chunks: 256 blocks write 256 different chunks at a time (each chunk
// with the level-1 pattern: 512 streams of 256-byte runs), every block stamps the wall clock around its chunk.  A slow buffer whose
// chunks are ALL a little slower points at an interleave property; one with a few very slow chunks at particular physical ranges.
//   hipcc -O3 --offload-arch=gfx950 -o write_map write_map.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
// pad: words added to the distance between a block's streams; skew: words added to the distance between the blocks' areas
__global__ __launch_bounds__(1024) void k_runs(u64* __restrict__ out, u64 n, int P, int run, u64 pad = 0, u64 skew = 0) {
    const u64 per_block = n / gridDim.x - skew * 0, per_stream0 = (n / gridDim.x - 4096 * 0) / P;
    const u64 per_stream = per_stream0 - 600 + pad;          // (600 words of room for the pads tried below)
    u64* base = out + (u64)blockIdx.x * (per_block - 70000 + skew);
    const int runs_per_trip = 1024 / run;
    const int r = threadIdx.x / run, l = threadIdx.x % run;
    for (u64 off = 0; off + run <= per_stream; off += run)
        for (int p = r; p < P; p += runs_per_trip) base[(u64)p * per_stream + off + l] = off + l;
}
// chunk c = words [c * cw, (c + 1) * cw); block b of round r takes chunk r * gridDim.x + b
__global__ __launch_bounds__(1024) void k_map(u64* __restrict__ out, u64 cw, u64 nchunks, u64 round, u64* __restrict__ t) {
    const u64 c = round * gridDim.x + blockIdx.x;
    if (c >= nchunks) return;
    u64* base = out + c * cw;
    const u64 per_stream = cw / 512;
    const int r = threadIdx.x / 32, l = threadIdx.x % 32;
    __syncthreads();
    const u64 t0 = wall_clock64();
    for (u64 off = 0; off + 32 <= per_stream; off += 32)
        for (int p = r; p < 512; p += 32) base[(u64)p * per_stream + off + l] = off + l;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) t[c] = wall_clock64() - t0;
}
static float runs_ms(u64* out, u64 n, u64 pad = 600, u64 skew = 70000) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9;
    for (int i = 0; i < 4; ++i) {
        hipEventRecord(a); hipLaunchKernelGGL(k_runs, dim3(256), dim3(1024), 0, 0, out, n, 512, 32, pad, skew); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (i && ms < best) best = ms;
    }
    return best;
}
// dependent loads that hop over the buffer, one line per `step` bytes, in a scrambled order: the time per hop is the cost of an address
// translation that the TLBs do not hold -- it depends on how large the physically contiguous fragments behind the buffer are
__global__ void k_chain_init(u64* __restrict__ buf, u64 step_words, u64 npages) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npages) return;
    const u64 next = (i * 2654435761ull + 12345ull) % npages;           // (a permutation when npages is coprime with the multiplier: npages is a prime-ish odd number below)
    buf[i * step_words] = next;
}
__global__ void k_chase(const u64* __restrict__ buf, u64 step_words, u64 hops, u64* __restrict__ out) {
    u64 p = 0;
    const u64 t0 = wall_clock64();
    for (u64 h = 0; h < hops; ++h) p = buf[p * step_words];
    out[0] = wall_clock64() - t0; out[1] = p;
}
static double chase_ns(u64* buf, u64 n_words, u64 step_bytes) {
    const u64 step_words = step_bytes / 8; u64 npages = n_words / step_words; if (npages % 2 == 0) --npages;
    u64* d; hipMalloc(&d, 16);
    hipLaunchKernelGGL(k_chain_init, dim3((unsigned)((npages + 255) / 256)), dim3(256), 0, 0, buf, step_words, npages);
    const u64 hops = 20000;
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, (const u64*)buf, step_words, hops, d);
    hipLaunchKernelGGL(k_chase, dim3(1), dim3(1), 0, 0, (const u64*)buf, step_words, hops, d);
    u64 h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); hipFree(d);
    return (double)h[0] * 10.0 / (double)hops;      // 100 MHz ticks -> ns
}
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 8;
    const u64 n = 1500000000ull;                    // 12 GB
    const u64 cw = (32ull << 20) / 8, nchunks = n / cw;
    std::vector<u64*> buf(nb); std::vector<float> ms(nb);
    for (int i = 0; i < nb; ++i) if (hipMalloc(&buf[i], n * 8) != hipSuccess) { printf("alloc %d failed\n", i); return 1; }
    for (int i = 0; i < nb; ++i) { ms[i] = runs_ms(buf[i], n); printf("buffer %d at %p: pattern %.3f ms\n", i, (void*)buf[i], ms[i]); }
    for (int i = 0; i < nb; ++i)
        printf("buffer %d (%.3f ms): dependent-load hop over 64 KB %.0f ns, 2 MB %.0f ns, 32 MB %.0f ns, 1 GB %.0f ns\n", i, ms[i],
               chase_ns(buf[i], n, 64ull << 10), chase_ns(buf[i], n, 2ull << 20), chase_ns(buf[i], n, 32ull << 20), chase_ns(buf[i], n, 1ull << 30));
    const int fast = (int)(std::min_element(ms.begin(), ms.end()) - ms.begin()), slow = (int)(std::max_element(ms.begin(), ms.end()) - ms.begin());
    // the same pattern with the streams / the blocks' areas a little closer together: does the class of a buffer follow the strides?
    for (int which : {fast, slow}) {
        printf("%s buffer %d: stream pad (words) ->", which == fast ? "FAST" : "SLOW", which);
        for (u64 pad : {600ull, 599ull, 592ull, 584ull, 568ull, 536ull, 472ull, 344ull, 88ull}) printf("  %llu: %.3f", (unsigned long long)(600 - pad), runs_ms(buf[which], n, pad, 70000));
        printf("\n%s buffer %d: block skew (words)  ->", which == fast ? "FAST" : "SLOW", which);
        for (u64 skew : {70000ull, 69984ull, 69968ull, 69744ull, 69488ull, 65904ull, 61808ull, 37232ull, 4464ull}) printf("  %llu: %.3f", (unsigned long long)(70000 - skew), runs_ms(buf[which], n, 600, skew));
        printf("\n");
    }
    u64* d_t; hipMalloc(&d_t, nchunks * 8);
    for (int which : {fast, slow}) {
        std::vector<u64> acc(nchunks, ~0ull), t(nchunks);
        for (int rep = 0; rep < 3; ++rep) {
            for (u64 r = 0; r * 256 < nchunks; ++r) hipLaunchKernelGGL(k_map, dim3(256), dim3(1024), 0, 0, buf[which], cw, nchunks, r, d_t);
            hipMemcpy(t.data(), d_t, nchunks * 8, hipMemcpyDeviceToHost);
            for (u64 c = 0; c < nchunks; ++c) acc[c] = std::min(acc[c], t[c]);
        }
        std::vector<u64> s(acc); std::sort(s.begin(), s.end());
        printf("%s buffer %d (%.3f ms): chunk time (100 MHz ticks) min %llu  p10 %llu  median %llu  p90 %llu  max %llu\n", which == fast ? "FAST" : "SLOW", which, ms[which],
               (unsigned long long)s[0], (unsigned long long)s[nchunks / 10], (unsigned long long)s[nchunks / 2], (unsigned long long)s[nchunks * 9 / 10], (unsigned long long)s[nchunks - 1]);
        // where the slow chunks are: chunks above 1.15 x the median, as ranges
        const u64 thr = s[nchunks / 2] * 115 / 100; int shown = 0; u64 cnt = 0;
        for (u64 c = 0; c < nchunks; ++c) if (acc[c] > thr) { ++cnt; if (shown < 24) { printf("  slow chunk %llu (%llu)", (unsigned long long)c, (unsigned long long)acc[c]); if (++shown % 6 == 0) printf("\n"); } }
        printf("\n  %llu of %llu chunks above 1.15 x median\n", (unsigned long long)cnt, (unsigned long long)nchunks);
    }
    return 0;
}
