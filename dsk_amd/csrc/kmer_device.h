// kmer_device.h -- device-side k-mer arithmetic for gfx950 (wave64).
//
// MI355X-native restatement of what gatb-core's Kmer<span>::ModelCanonical does
// per base (call sites: utils/dsk2ascii.cpp:65,91; semantics README.md:104-112):
// 2-bit code A=0 C=1 T=2 G=3, first base most significant, canonical =
// min(forward, reverse-complement), complement = code ^ 2.
//
// Layout of the encoded read stream in HBM (kernel k_encode):
//   packed[w]  : u64, bases 32w .. 32w+31, base j in bits (63-2j, 62-2j)   (MSB first)
//   inval[w]   : u32, bit (31-j) set  <=>  base 32w+j is not one of ACGTacgt
// so a k-mer is a funnel shift over (packed[w-1] : packed[w]) and its validity a
// mask test over (inval[w-1] : inval[w]).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

// Empty-slot sentinel of the one-word LDS table.  Tables hold MIXED keys, so what matters is the
// pre-image kunmix(~0) = 0x89a5850e63c5f8aa: it has bit 63 set (not a k-mer for k <= 31) and as a
// 32-mer it is larger than its reverse complement, i.e. never canonical -- no real key collides.
#define DSK_EMPTY 0xFFFFFFFFFFFFFFFFull

// ---------------------------------------------------------------- hashing
// Bijective 64-bit mixer (murmur3 finalizer).  Partition arrays hold
// h = kmix(canonical) so radix digits and table slots are plain bit fields of
// the stored word; kunmix restores the k-mer for the emitted rows only.
__host__ __device__ __forceinline__ u64 kmix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33; return x;
}
__host__ __device__ __forceinline__ u64 kunmix(u64 x) {
    x ^= x >> 33; x *= 0x9cb4b2f8129337dbULL;
    x ^= x >> 33; x *= 0x4f74430c22a54005ULL;
    x ^= x >> 33; return x;
}

// 128-bit keys (k in 33..64): two Feistel rounds over the 64-bit mixer, still a
// bijection on (hi, lo); digits/slots are taken from the mixed hi word.
__host__ __device__ __forceinline__ void kmix2(u64& hi, u64& lo) {
    lo ^= kmix(hi);            // round 1
    hi ^= kmix(lo + 0x9e3779b97f4a7c15ULL);   // round 2
    hi = kmix(hi);             // spread within the digit word
}
__host__ __device__ __forceinline__ void kunmix2(u64& hi, u64& lo) {
    hi = kunmix(hi);
    hi ^= kmix(lo + 0x9e3779b97f4a7c15ULL);
    lo ^= kmix(hi);
}

// reverse the order of the 32 2-bit groups of x
__device__ __forceinline__ u64 rev_pairs(u64 x) {
    u64 r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
}

// ---------------------------------------------------------------- one-word k-mers (k <= 32)
// Generate the canonical k-mers of the NP windows ending at bases
// 32*wi + t0 .. 32*wi + t0 + NP-1   (t0 + NP <= 32).
// canon[j] valid iff bit j of the returned mask is set.
template <int NP>
__device__ __forceinline__ u32 gen_kmers1(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                          u64 wi, int t0, int k, u64 (&canon)[NP]) {
    const u64 cur = packed[wi];
    const u64 prev = wi ? packed[wi - 1] : 0ull;
    const u32 ic = inval[wi];
    const u32 ip = wi ? inval[wi - 1] : 0xFFFFFFFFu;
    const u64 invwin = ((u64)ip << 32) | ic;                 // bit (63-i) <-> base i of (prev:cur)
    const u64 kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const u64 kbits = (1ull << k) - 1;                       // k <= 32
    // forward value of the window ending just before t0
    u64 fwd = (t0 == 0) ? prev : ((prev << (2 * t0)) | (cur >> (64 - 2 * t0)));
    fwd &= kmask;
    u64 rc = (rev_pairs(fwd) >> (64 - 2 * k)) ^ (0xAAAAAAAAAAAAAAAAULL & kmask);
    const int rcs = 2 * k - 2;
    u32 vmask = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int t = t0 + j;
        const u64 c = (cur >> (62 - 2 * t)) & 3ull;
        fwd = ((fwd << 2) | c) & kmask;
        rc = (rc >> 2) | ((c ^ 2ull) << rcs);
        canon[j] = fwd < rc ? fwd : rc;
        if ((invwin & (kbits << (31 - t))) == 0) vmask |= (1u << j);
    }
    return vmask;
}

// ---------------------------------------------------------------- two-word k-mers (33 <= k <= 64)
struct K2 { u64 hi, lo; };

// Windows ending at bases 32*wi + t0 + j, j < NP.  Needs packed[wi-2..wi].
template <int NP>
__device__ __forceinline__ u32 gen_kmers2(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                          u64 wi, int t0, int k, K2 (&canon)[NP]) {
    const u64 cur = packed[wi];
    const u64 p1 = wi >= 1 ? packed[wi - 1] : 0ull;
    const u64 p2 = wi >= 2 ? packed[wi - 2] : 0ull;
    const u32 ic = inval[wi];
    const u32 i1 = wi >= 1 ? inval[wi - 1] : 0xFFFFFFFFu;
    const u32 i2 = wi >= 2 ? inval[wi - 2] : 0xFFFFFFFFu;
    // 96-bit invalid window: bit (95-i) <-> base i of (p2:p1:cur); keep as hi32:lo64
    const u64 inv_lo = ((u64)i1 << 32) | ic;
    const u32 inv_hi = i2;
    const int kh = 2 * k - 64;                                // bits of the k-mer living in hi (2..64)
    const u64 hmask = (kh == 64) ? ~0ull : ((1ull << kh) - 1);
    // forward value (128 bit) of the window ending just before t0: the last 64
    // bases of (p2:p1:cur) up to base 64+t0-1, i.e. bits of (p2:p1:cur) >> 2*(32-t0)
    u64 fhi, flo;
    if (t0 == 0) { fhi = p2; flo = p1; }
    else { fhi = (p2 << (2 * t0)) | (p1 >> (64 - 2 * t0)); flo = (p1 << (2 * t0)) | (cur >> (64 - 2 * t0)); }
    fhi &= hmask;
    // reverse complement of the k-base window: reverse all 64 pairs of (fhi:flo)
    // -> (rev(flo):rev(fhi)), shift right by 128-2k, complement k pairs
    u64 rhi = rev_pairs(flo), rlo = rev_pairs(fhi);
    {
        const int sh = 128 - 2 * k;                           // 0..62
        if (sh) { rlo = (rlo >> sh) | (rhi << (64 - sh)); rhi >>= sh; }
        rlo ^= 0xAAAAAAAAAAAAAAAAULL;
        rhi ^= (0xAAAAAAAAAAAAAAAAULL & hmask);
    }
    const int rcs = kh - 2;                                   // position of the newest complement in hi
    u32 vmask = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int t = t0 + j;
        const u64 c = (cur >> (62 - 2 * t)) & 3ull;
        fhi = ((fhi << 2) | (flo >> 62)) & hmask;
        flo = (flo << 2) | c;
        rlo = (rlo >> 2) | (rhi << 62);
        rhi = (rhi >> 2) | ((c ^ 2ull) << rcs);
        const bool fl = fhi < rhi || (fhi == rhi && flo < rlo);
        canon[j].hi = fl ? fhi : rhi;
        canon[j].lo = fl ? flo : rlo;
        // window = bases [64+t-k+1, 64+t] of the 96-base frame -> bits (31-t) .. (31-t+k-1)
        const int b0 = 31 - t;                                // 0..31
        const u64 lo_bits = (k + b0 >= 64) ? (~0ull << b0) : (((1ull << k) - 1) << b0);
        const int over = k + b0 - 64;                         // bits spilling into inv_hi
        const u32 hi_bits = over > 0 ? ((over >= 32) ? 0xFFFFFFFFu : ((1u << over) - 1)) : 0u;
        if (((inv_lo & lo_bits) == 0) && ((inv_hi & hi_bits) == 0)) vmask |= (1u << j);
    }
    return vmask;
}
