// Issue cost of the integer VALU instructions the key generation / mixer use, gfx950 (MI355X), wave64.
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; ./valu_rates
// 16 waves per CU (4 per SIMD), every wave runs ITER x 32 instructions of one kind on 8 independent register
// chains (so neither latency nor dependencies bound the rate): prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64; typedef unsigned int u32;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(1024) void k(u64* out, int iters, u32 seed, u32 cs) {
    u32 a[8], b[8]; u64 q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = a[i] ^ 0x9e3779b9u; q[i] = ((u64)a[i] << 32) | b[i]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 3) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == 4) asm volatile("v_lshlrev_b64 %0, 2, %0" : "+v"(q[i]));
                if (OP == 5) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(q[i]) : "v"(b[i]));
                if (OP == 6) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 8) asm volatile("v_alignbit_b32 %0, %0, %1, 30" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 9) asm volatile("v_cmp_lt_u64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(q[i]), "v"(q[(i + 1) & 7]), "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == 10) asm volatile("v_bfe_u32 %0, %1, 6, 2" : "=v"(a[i]) : "v"(b[i]));
                if (OP == 11) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 12) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 13) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (OP == 14) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "s"(cs));
                if (OP == 15) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 16) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
                if (OP == 17) asm volatile("v_mad_u32_u16 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 18) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 19) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b[i]));
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += a[i] + b[i] + q[i];
    if (acc == 0x1234567812345678ull) out[0] = acc;
}

// candidate mixers, whole: cost per key inside a rolling-generation-like loop
__device__ __forceinline__ u64 mix_murmur(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }
__device__ __forceinline__ u64 mix_one(u64 x) { x *= 0x9e3779b97f4a7c15ULL; x ^= x >> 32; return x; }
__device__ __forceinline__ u64 mix_fold_one(u64 x) { x ^= x >> 32; x *= 0x9e3779b97f4a7c15ULL; x ^= x >> 32; return x; }
__device__ __forceinline__ u64 mix_feistel(u64 x) {
    u32 l = (u32)(x >> 32), r = (u32)x;
    l += r * 0x9e3779b1u; r += l * 0x85ebca6bu; l ^= r >> 15; return ((u64)l << 32) | r;
}
template <int M>
__global__ __launch_bounds__(1024) void km(u64* out, int iters, u64 seed) {
    u64 x = seed + threadIdx.x * 0x9e3779b97f4a7c15ULL + blockIdx.x, acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            x = (x << 2) | (r & 3);
            u64 h = M == 0 ? mix_murmur(x) : M == 1 ? mix_one(x) : M == 2 ? mix_fold_one(x) : M == 3 ? mix_feistel(x) : x;
            acc += __umulhi((u32)(h >> 32), 768u) + (u32)h;
        }
    }
    if (acc == 0x1234567812345678ull) out[0] = acc;
}

template <class F>
float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(10); hipEventRecord(a); f(0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms;
}
template <int OP> void run(const char* name, u64* d) {
    const int iters = 4000;
    const float ms = timeit([&](int warm) { hipLaunchKernelGGL((k<OP>), dim3(256), dim3(1024), 0, 0, d, warm ? warm : iters, 12345u, 0x9e3779b1u); });
    const double wi = (double)iters * 32 * 4;                     // wave-instructions per SIMD (4 waves per SIMD)
    printf("%-34s %8.3f ms  %6.2f cycles / wave-instruction / SIMD (at 2.1 GHz; includes loop overhead ~3 %%)\n", name, ms, ms * 1e-3 * 2.1e9 / wi);
}
template <int M> void runm(const char* name, u64* d) {
    const int iters = 2000;
    const float ms = timeit([&](int warm) { hipLaunchKernelGGL((km<M>), dim3(256), dim3(1024), 0, 0, d, warm ? warm : iters, 777ull); });
    const double keys = (double)iters * 16 * 4;                    // wave-keys per SIMD
    printf("%-34s %8.3f ms  %6.1f cycles / wave-key / SIMD\n", name, ms, ms * 1e-3 * 2.1e9 / keys);
}
int main() {
    u64* d; hipMalloc(&d, 8);
    run<0>("v_add_u32", d); run<11>("v_xor_b32", d); run<12>("v_add3_u32", d); run<19>("v_lshl_or_b32", d);
    run<1>("v_mul_lo_u32", d); run<14>("v_mul_lo_u32 (sgpr operand)", d); run<2>("v_mul_hi_u32", d); run<3>("v_mad_u64_u32", d);
    run<6>("v_mul_u32_u24", d); run<7>("v_mad_u32_u24", d); run<18>("v_mad_i32_i24", d); run<17>("v_mad_u32_u16", d);
    run<4>("v_lshlrev_b64 (const 2)", d); run<5>("v_lshrrev_b64 (vgpr shift)", d); run<13>("v_lshl_add_u64", d);
    run<8>("v_alignbit_b32", d); run<10>("v_bfe_u32", d); run<15>("v_perm_b32", d); run<16>("v_bfrev_b32", d);
    run<9>("v_cmp_lt_u64 + v_cndmask_b32", d);
    runm<4>("roll + digit only (no mixer)", d);
    runm<0>("murmur3 finalizer", d); runm<1>("x * C; x ^= x >> 32", d); runm<2>("fold; x * C; x ^= x >> 32", d); runm<3>("32-bit add-multiply Feistel", d);
    return 0;
}
