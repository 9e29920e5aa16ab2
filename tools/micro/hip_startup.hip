// What a process pays before its first kernel runs: each HIP start-up call timed on its own (a binary with ONE trivial kernel, so
// that code-object loading is not in the picture), then the same through libdskgpu.so (dlopen + dskgpu_create + a 1 KB count).
//   hipcc -O2 --offload-arch=gfx950 -o hip_startup hip_startup.hip -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <sys/time.h>
#include <cstdio>
#include <cstring>
#include "../../include/dskgpu.h"
static double now() { timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }
__global__ void k_nop(int* p) { if (p) *p = 1; }
int main(int argc, char** argv) {
    double t0 = now(), t;
    if (argc > 1 && !strcmp(argv[1], "lib")) {
        void* h = dlopen(argv[2], RTLD_NOW); t = now(); printf("dlopen(libdskgpu.so)        %.3f s\n", t - t0); t0 = t;
        if (!h) { printf("%s\n", dlerror()); return 1; }
        auto create = (int (*)(const dskgpu_config*, dskgpu_ctx**))dlsym(h, "dskgpu_create");
        auto push = (int (*)(dskgpu_ctx*, const char*, uint64_t))dlsym(h, "dskgpu_push_reads");
        auto count = (int (*)(dskgpu_ctx*))dlsym(h, "dskgpu_count");
        dskgpu_config c{}; c.kmer_size = 31; c.abundance_min = 2; c.abundance_max = 0x7fffffff; c.histo_max = 10000; c.world_size = 1;
        dskgpu_ctx* ctx = nullptr;
        int rc = create(&c, &ctx); t = now(); printf("dskgpu_create               %.3f s (rc %d)\n", t - t0, rc); t0 = t;
        char reads[1024]; memset(reads, 'A', sizeof reads); for (int i = 0; i < 1024; i += 7) reads[i] = "ACGT"[i & 3];
        rc = push(ctx, reads, sizeof reads); t = now(); printf("dskgpu_push_reads (1 KB)    %.3f s (rc %d)\n", t - t0, rc); t0 = t;
        rc = count(ctx); t = now(); printf("dskgpu_count (first launch) %.3f s (rc %d)\n", t - t0, rc); t0 = t;
        rc = count(ctx); t = now(); printf("dskgpu_count (again)        %.3f s (rc %d)\n", t - t0, rc); t0 = t;
        return 0;
    }
    hipInit(0); t = now(); printf("hipInit                     %.3f s\n", t - t0); t0 = t;
    int n = 0; hipGetDeviceCount(&n); t = now(); printf("hipGetDeviceCount (%d)       %.3f s\n", n, t - t0); t0 = t;
    hipSetDevice(0); t = now(); printf("hipSetDevice                %.3f s\n", t - t0); t0 = t;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); t = now(); printf("hipGetDeviceProperties      %.3f s\n", t - t0); t0 = t;
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); t = now(); printf("hipStreamCreate             %.3f s\n", t - t0); t0 = t;
    int* d; hipMalloc(&d, 1 << 20); t = now(); printf("hipMalloc(1 MB)             %.3f s\n", t - t0); t0 = t;
    hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, s, d); hipStreamSynchronize(s); t = now(); printf("first kernel + sync         %.3f s\n", t - t0); t0 = t;
    void* big; hipMalloc(&big, (size_t)8 << 30); t = now(); printf("hipMalloc(8 GB)             %.3f s\n", t - t0); t0 = t;
    return 0;
}
