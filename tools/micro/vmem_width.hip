// Wave-instruction cost of loads / stores by width, gfx950: is a 16-byte-per-lane access cheaper per byte than an 8-byte one
// on the CU's vector-memory path?   hipcc --offload-arch=gfx950 -O3 -o vmem_width vmem_width.hip ; ./vmem_width
// 256 blocks x 1024 threads; loads stream a 6 GB buffer (HBM), stores go to per-block 128 KB windows (L2-resident) or stream.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
// MODE 0: load dwordx2, lanes consecutive   1: load dwordx4, lanes consecutive
//      2: store 8 groups of 8 lanes x 8 B on scattered aligned 64-B lines   3: store 16 groups of 4 lanes x 16 B on scattered aligned 64-B lines
//      4: streaming store dwordx2 consecutive   5: streaming store dwordx4 consecutive
template <int MODE>
__global__ __launch_bounds__(1024) void k(u64* buf, u64 nkeys, int iters, u64* sink) {
    const u32 tid = threadIdx.x, lane = tid & 63;
    u64 acc = 0;
    u32 x = (tid >> 6) * 2654435761u + blockIdx.x * 40503u + 12345u;
    u64* win = buf + (size_t)blockIdx.x * 16384;
    const u64 per_block = nkeys / gridDim.x;
    const u64 base = (u64)blockIdx.x * per_block;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const u64 tile = ((u64)it * 8 + u) * (MODE == 1 || MODE == 5 ? 2048 : 1024);
            if (MODE == 0) acc += buf[base + (tile + tid) % per_block];
            if (MODE == 1) { const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(buf + base + (tile + 2 * tid) % per_block); acc += v.x + v.y; }
            x = x * 1664525u + 1013904223u;
            const u32 r = x >> 8;
            if (MODE == 2) win[(((r >> (lane >> 3)) * 2654435761u >> 8) % 1900) * 8 + (lane & 7)] = (u64)x + lane;
            if (MODE == 3) *reinterpret_cast<ulonglong2*>(win + (((r >> (lane >> 2)) * 2654435761u >> 8) % 1900) * 8 + (lane & 3) * 2) = make_ulonglong2((u64)x, (u64)lane);
            if (MODE == 4) buf[base + (tile + tid) % per_block] = (u64)x + lane;
            if (MODE == 5) *reinterpret_cast<ulonglong2*>(buf + base + (tile + 2 * tid) % per_block) = make_ulonglong2((u64)x, (u64)lane);
        }
    }
    if (acc == 0x1234567812345678ull) sink[0] = acc;
}
template <int MODE> void run(const char* name, u64* d, u64 nkeys, u64* sink) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 300;
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, d, nkeys, 5, sink);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(1024), 0, 0, d, nkeys, iters, sink);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)iters * 8 * 16;
    const double bytes = (double)iters * 8 * 1024 * 256 * (MODE == 1 || MODE == 3 || MODE == 5 ? 16 : 8);
    printf("%-72s %8.3f ms  %6.1f cycles / wave-instruction / CU (2.1 GHz)  %6.0f GB/s\n", name, ms, ms * 1e-3 * 2.1e9 / instr_per_cu, bytes / ms / 1e6);
}
int main() {
    const u64 nkeys = 768ull << 20;                    // 6 GB
    u64 *d, *sink; (void)hipMalloc(&d, nkeys * 8 + 4096); (void)hipMalloc(&sink, 64);
    (void)hipMemset(d, 1, nkeys * 8);
    run<0>("load  dwordx2, 64 consecutive lanes (512 B), streaming from HBM", d, nkeys, sink);
    run<1>("load  dwordx4, 64 consecutive lanes (1 KB), streaming from HBM", d, nkeys, sink);
    run<2>("store 8 groups of 8 lanes x 8 B, each on its own aligned 64-B line", d, nkeys, sink);
    run<3>("store 16 groups of 4 lanes x 16 B, each on its own aligned 64-B line", d, nkeys, sink);
    run<4>("store dwordx2, 64 consecutive lanes (512 B), streaming to HBM", d, nkeys, sink);
    run<5>("store dwordx4, 64 consecutive lanes (1 KB), streaming to HBM", d, nkeys, sink);
    return 0;
}
