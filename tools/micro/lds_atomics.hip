// Throughput of LDS operations on random slots of a 4096-entry table (what the count kernels do), gfx950.
// hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip ; ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;
#define SLOTS 4096
template <int OP, int NT, int HOT>
__global__ __launch_bounds__(NT) void k(u64* out, int iters) {
    __shared__ u64 t64[SLOTS];
    u32* t32 = reinterpret_cast<u32*>(t64);
    for (int i = threadIdx.x; i < SLOTS; i += NT) t64[i] = 0;
    __syncthreads();
    u32 x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    u64 acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = x * 1664525u + 1013904223u;
            // HOT: the slot distribution of a real sub-partition: 3/4 of the keys are copies of ~51 hot k-mers, the rest singletons
            const u32 s = (HOT && ((x >> 4) & 3u) != 0u) ? (((x >> 8) % 51u) * 77u + 13u) & (SLOTS - 1) : (x >> 12) & (SLOTS - 1);
            if (OP == 0) acc += t64[s];
            if (OP == 1) atomicAdd(&t32[s], 1u);
            if (OP == 2) acc += atomicAdd(&t32[s], 1u);
            if (OP == 3) atomicAdd(&t64[s], 1ull);
            if (OP == 4) acc += atomicCAS(&t32[s], x & 7u, x);
            if (OP == 5) acc += atomicCAS(&t64[s], (u64)(x & 7u), (u64)x);
            if (OP == 6) t64[s] = x;
            if (OP == 7) acc += atomicAdd(&t64[s], 1ull);
            if (OP == 8) acc += t32[s];
            if (OP == 9) atomicMax(&t64[s], (u64)x);
        }
    }
    __syncthreads();
    acc += t64[threadIdx.x & (SLOTS - 1)];
    if (acc == 0x1234567812345678ull) out[0] = acc;
}
template <int OP, int NT, int HOT = 0>
void run(const char* name, int blocks_per_cu) {
    u64* d; hipMalloc(&d, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((k<OP, NT, HOT>), dim3(grid), dim3(NT), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<OP, NT, HOT>), dim3(grid), dim3(NT), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)grid * NT * iters * 4;
    // cycles per wave-instruction per CU at 2.4 GHz nominal
    const double wave_instr_per_cu = ops / 64 / 256;
    printf("%-28s %s NT=%4d x%d/CU  %8.3f ms  %7.1f Gops/s  %6.1f cyc/wave-instr/CU (at 2.1 GHz)\n", name, HOT ? "hot " : "unif", NT, blocks_per_cu, ms, ops / ms / 1e6, ms * 1e-3 * 2.1e9 / wave_instr_per_cu);
    hipFree(d);
}
int main() {
    run<0, 1024>("ds_read_b64", 2);
    run<8, 1024>("ds_read_b32", 2);
    run<6, 1024>("ds_write_b64", 2);
    run<1, 1024>("ds_add_u32 (no return)", 2);
    run<2, 1024>("ds_add_rtn_u32", 2);
    run<3, 1024>("ds_add_u64 (no return)", 2);
    run<7, 1024>("ds_add_rtn_u64", 2);
    run<9, 1024>("ds_max_u64 (no return)", 2);
    run<4, 1024>("ds_cmpst_rtn_b32", 2);
    run<5, 1024>("ds_cmpst_rtn_b64", 2);
    run<5, 512>("ds_cmpst_rtn_b64", 2);
    run<2, 512>("ds_add_rtn_u32", 2);
    run<0, 512>("ds_read_b64", 2);
    run<0, 1024, 1>("ds_read_b64", 2);
    run<8, 1024, 1>("ds_read_b32", 2);
    run<6, 1024, 1>("ds_write_b64", 2);
    run<1, 1024, 1>("ds_add_u32 (no return)", 2);
    run<2, 1024, 1>("ds_add_rtn_u32", 2);
    run<3, 1024, 1>("ds_add_u64 (no return)", 2);
    run<5, 1024, 1>("ds_cmpst_rtn_b64", 2);
    return 0;
}
