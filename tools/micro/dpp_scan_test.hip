#include "../../dsk_amd/csrc/kmer_device.h"
#include <cstdio>
__global__ void k(u32* out) { u32 v = threadIdx.x * 7u + 1u; out[threadIdx.x] = wave_incl_scan(v); }
int main() { u32* d; hipMalloc(&d, 64*4*4); hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, d); u32 h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int bad = 0; for (int w = 0; w < 4; ++w) { u32 run = 0; for (int l = 0; l < 64; ++l) { run += (w*64+l)*7u+1u; if (h[w*64+l] != run) ++bad; } } printf("dpp scan bad=%d\n", bad); return bad; }
