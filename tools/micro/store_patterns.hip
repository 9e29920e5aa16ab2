// Cost of a wave's global store instruction as a function of how its 64 lanes' addresses fall on cache lines, gfx950.
// hipcc --offload-arch=gfx950 -O3 -o store_patterns store_patterns.hip ; ./store_patterns
// 256 blocks x 1024 threads (one per CU), every wave issues ITER x 16 stores into its block's own 128 KB window (L2-resident:
// what is measured is the CU's store path -- address coalescing and the L1 -> L2 write requests -- not HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;

// PAT 0: 64 consecutive 8-byte slots, 512-byte aligned          1: the same shifted by one slot (8 bytes)
//     2: three runs (21, 21, 22 slots) at unrelated 8-byte-aligned places   3: eight 8-lane groups, each on its own aligned 64-byte line
//     4: eight 8-lane groups, each at an unaligned place        5: sixteen 4-lane groups on own aligned 32-byte sectors
//     6: dwordx4: 32 consecutive 16-byte slots per half wave ... (lanes store 16 B), aligned   7: dwordx4 shifted by 8 bytes
//     8: every lane its own line (worst case)
template <int PAT>
__global__ __launch_bounds__(1024) void k(u64* out, int iters) {
    u64* win = out + (size_t)blockIdx.x * 16384 + 64;         // the block's 128 KB window (+ room for shifts)
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 x = (threadIdx.x >> 6) * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            x = x * 1664525u + 1013904223u;                       // wave-uniform pseudo-random
            const u32 r = x >> 8;
            u32 slot;
            if (PAT == 0) slot = ((r % 240) * 64 + lane);
            if (PAT == 1) slot = ((r % 240) * 64 + lane + 1);
            if (PAT == 2) { const u32 g = lane < 21 ? 0 : lane < 42 ? 1 : 2; slot = ((r >> (5 * g)) % 15000) + g * 7 + lane; }
            if (PAT == 3) slot = (((r >> (lane >> 3)) * 2654435761u >> 8) % 1900) * 8 + (lane & 7);
            if (PAT == 4) slot = (((r >> (lane >> 3)) * 2654435761u >> 8) % 15000) + (lane & 7);
            if (PAT == 5) slot = (((r >> (lane >> 2)) * 2654435761u >> 8) % 3800) * 4 + (lane & 3);
            if (PAT == 8) slot = (((r + lane) * 2654435761u >> 8) % 1900) * 8;
            if (PAT == 6 || PAT == 7) {
                const u32 s2 = ((r % 120) * 128 + lane * 2 + (PAT == 7 ? 1 : 0));
                *reinterpret_cast<ulonglong2*>(win + s2) = make_ulonglong2((u64)x, (u64)lane);
            } else win[slot] = (u64)x + lane;
        }
    }
    if (wave == 99) out[0] = x;
}
template <int PAT> void run(const char* name, u64* d) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int iters = 300;
    hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(1024), 0, 0, d, 5);
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<PAT>), dim3(256), dim3(1024), 0, 0, d, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    const double instr_per_cu = (double)iters * 16 * 16;          // wave store instructions per CU
    const double bytes = (double)iters * 16 * 1024 * 256 * (PAT == 6 || PAT == 7 ? 16 : 8);
    printf("%-66s %8.3f ms  %6.1f cycles / wave-store / CU (2.1 GHz)  %6.0f GB/s\n", name, ms, ms * 1e-3 * 2.1e9 / instr_per_cu, bytes / ms / 1e6);
}
int main() {
    u64* d; (void)hipMalloc(&d, (size_t)257 * 16384 * 8 + 4096);
    run<0>("64 consecutive 8-B slots, 512-B aligned", d);
    run<1>("64 consecutive 8-B slots, shifted by 8 B", d);
    run<2>("three runs of 21-22 slots at unrelated 8-B-aligned places", d);
    run<3>("eight 8-lane groups, each on its own aligned 64-B line", d);
    run<4>("eight 8-lane groups, each at an unaligned place", d);
    run<5>("sixteen 4-lane groups, each on its own aligned 32-B sector", d);
    run<8>("every lane on its own line", d);
    run<6>("dwordx4: 64 consecutive 16-B slots, aligned", d);
    run<7>("dwordx4: 64 consecutive 16-B slots, shifted by 8 B", d);
    return 0;
}
