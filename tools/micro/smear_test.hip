// gen_kmers1<16>'s validity mask (one smear) against the per-window test, on random invalid masks
#include "../../dsk_amd/csrc/kmer_device.h"
#include <cstdio>
#include <vector>
#include <random>
__global__ void k(const u64* packed, const u32* inval, u64 nwords, int kk, unsigned long long* bad, u32* vm_out) {
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 wi = t >> 1;
    if (wi >= nwords) return;
    const int t0 = (int)(t & 1) * 16;
    u64 c[16];
    const u32 vm = gen_kmers1<16>(packed, inval, wi, t0, kk, c);
    const u32 ic = inval[wi], ip = wi ? inval[wi - 1] : 0xFFFFFFFFu;
    const u64 invwin = ((u64)ip << 32) | ic, kbits = (1ull << kk) - 1;
    u32 ref = 0;
    for (int j = 0; j < 16; ++j) if ((invwin & (kbits << (31 - (t0 + j)))) == 0) ref |= 1u << j;
    vm_out[t] = vm;
    if (vm != ref) atomicAdd(bad, 1ull);
}
int main() {
    const u64 n = 1 << 22;
    std::vector<u64> p(n); std::vector<u32> iv(n);
    std::mt19937_64 g(5);
    for (u64 i = 0; i < n; ++i) { p[i] = g(); const u64 r = g(); iv[i] = (r & 3) == 0 ? (1u << (r >> 8 & 31)) : (r & 12) == 0 ? (u32)(r >> 16) : 0u; }
    u64* dp; u32* di; unsigned long long* db; u32* dv;
    hipMalloc(&dp, n * 8); hipMalloc(&di, n * 4); hipMalloc(&db, 8); hipMalloc(&dv, n * 2 * 4);
    hipMemcpy(dp, p.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(di, iv.data(), n * 4, hipMemcpyHostToDevice);
    for (int kk : {1, 2, 15, 16, 17, 20, 27, 31, 32}) {
        hipMemset(db, 0, 8);
        hipLaunchKernelGGL(k, dim3((unsigned)(n * 2 / 256)), dim3(256), 0, 0, dp, di, n, kk, db, dv);
        unsigned long long b = 0; hipMemcpy(&b, db, 8, hipMemcpyDeviceToHost);
        printf("k=%d mismatches %llu\n", kk, b);
    }
    return 0;
}
