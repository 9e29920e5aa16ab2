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
// pre-image kunmix(~0): it must not be a canonical k-mer of the k being counted (sentinel_is_a_kmer() in dskgpu.hip checks
// that when a context is created) -- then no real key collides.  tests/host/test_mixer.cpp runs the mixers on the host.
#define DSK_EMPTY 0xFFFFFFFFFFFFFFFFull

// ---------------------------------------------------------------- hashing
// Bijective 64-bit mixer: fold the high half into the low one, ONE 64-bit multiply by an odd constant, fold again.
// Partition arrays hold h = kmix(canonical) so radix digits and table slots are plain bit fields of the stored word;
// kunmix restores the k-mer for the emitted rows only.  The level-1 / level-2 scatters and the count are bound by the
// number of VALU instructions they issue (rocprofv3: 74-83 % of the SIMD issue cycles busy), and the murmur3 finalizer
// used before (two multiplies, three folds: 14 instructions, 43 cycles per wave and key) was a fifth of them; this one
// is 6 instructions / 19 cycles (tools/micro/valu_rates.hip) and partitions the same inputs as evenly: the product's
// top bits (the radix digits) depend on every input bit, the first fold brings the k-mer's leading bases into the low
// half before the multiply and the second fold the well-mixed high half into the slot / owner bits.
#define DSK_MIX_C 0xff51afd7ed558ccdULL
#define DSK_MIX_CINV 0x4f74430c22a54005ULL        // DSK_MIX_C * DSK_MIX_CINV == 1 (mod 2^64)
__host__ __device__ __forceinline__ u64 kmix(u64 x) {
    x ^= x >> 32; x *= DSK_MIX_C; x ^= x >> 32; return x;
}
__host__ __device__ __forceinline__ u64 kunmix(u64 x) {
    x ^= x >> 32; x *= DSK_MIX_CINV; x ^= x >> 32; return x;
}

// Inclusive prefix sum over the 64 lanes of a wave with DPP adds (row shifts inside the rows of 16, then the two row
// broadcasts): six v_add_u32_dpp, no LDS traffic -- __shfl_up() goes through ds_bpermute_b32 and costs ~5 instructions and an
// LDS round trip per step.
__device__ __forceinline__ u32 wave_incl_scan(u32 x) {
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);     // row_shr:1
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);     // row_shr:2
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);     // row_shr:4
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);     // row_shr:8
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);     // row_bcast:15 -> rows 1, 3
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);     // row_bcast:31 -> rows 2, 3
    return x;
}

// Multi-word keys (k in 33..128): KN<W>, word 0 least significant; digits/slots are taken from the mixed top word.
template <int W> struct KN { u64 w[W]; };
typedef KN<2> K2;
#define DSK_GOLD 0x9e3779b97f4a7c15ULL

// Only the top word has to be mixed: radix digits, table slot and owner are bit fields of it, the other words are
// just compared for equality.  top' = kmix(top ^ t), t = XOR over the lower words of kmum(word, odd constant): a bijection on the
// W words (for fixed lower words it is one in the top word).  kmum = low half XOR high half of the full 128-bit product: a
// difference in ANY bit p of a word moves bits p .. p + 63 of the product, so it reaches all 64 bits of t whatever p is -- which
// makes top' a 64-bit hash of the WHOLE key (two different keys share it with probability 2^-64, related or not), and that is
// what k_count2v3 needs: its table is keyed by top' alone.  Two weaker folds were in use before and both were caught by that
// kernel's verification on real-looking reads: `low * odd constant` (a multiply only carries upward: two k-mers that differ in
// base 0 and base 32 alone share top'), and kmix(low) (its first step x ^= x >> 32 cancels differences 32 bits = 16 bases apart,
// after which the multiply again sees a single high bit: 8 colliding pairs among the 1.1 * 10^7 distinct 63-mers of `small_repeats`).
// Both were fine for the partition; neither is a hash of the low word's high bits.
__host__ __device__ __forceinline__ u64 kfold_mult(int i) {
    return i == 0 ? 0x9e3779b97f4a7c15ULL : i == 1 ? 0xc2b2ae3d27d4eb4fULL : 0x165667b19e3779f9ULL;
}
__host__ __device__ __forceinline__ u64 kmum(u64 a, u64 c) {
    const unsigned __int128 p = (unsigned __int128)a * c;
    return (u64)p ^ (u64)(p >> 64);
}
template <int W>
__host__ __device__ __forceinline__ u64 kfold_low(const KN<W>& x) {
    u64 t = 0;
#pragma unroll
    for (int i = 0; i < W - 1; ++i) t ^= kmum(x.w[i], kfold_mult(i));
    return t;
}
template <int W>
__host__ __device__ __forceinline__ void kmixN(KN<W>& x) { x.w[W - 1] = kmix(x.w[W - 1] ^ kfold_low(x)); }
template <int W>
__host__ __device__ __forceinline__ void kunmixN(KN<W>& x) { x.w[W - 1] = kunmix(x.w[W - 1]) ^ kfold_low(x); }
template <int W>
__host__ __device__ __forceinline__ bool key_eq(const KN<W>& a, const KN<W>& b) {
    bool e = true;
#pragma unroll
    for (int i = 0; i < W; ++i) e = e && (a.w[i] == b.w[i]);
    return e;
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
// (the words of the frame already in registers: cur / prev = packed[wi], packed[wi-1]; ic / ip = their invalid masks)
template <int NP>
__device__ __forceinline__ u32 gen_kmers1_words(u64 cur, u64 prev, u32 ic, u32 ip, int t0, int k, u64 (&canon)[NP]) {
    const u64 invwin = ((u64)ip << 32) | ic;                 // bit (63-i) <-> base i of (prev:cur)
    const u64 kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const u64 kbits = (1ull << k) - 1;                       // k <= 32
    // forward value of the window ending just before t0
    u64 fwd = (t0 == 0) ? prev : ((prev << (2 * t0)) | (cur >> (64 - 2 * t0)));
    fwd &= kmask;
    u64 rc = (rev_pairs(fwd) >> (64 - 2 * k)) ^ (0xAAAAAAAAAAAAAAAAULL & kmask);
    const int rcs = 2 * k - 2;
    u32 vmask = 0;
    if (NP == 16) {
        // validity of all 16 windows at once: smear every invalid bit over the k - 1 following bases (log-step ORs), then the
        // window ending at base t of `cur` is bad iff bit (31 - t) is set; bit j of the result <-> t = t0 + j (t0 is 0 or 16).
        // (Straight-line on purpose: the same smear written as `while (2 * done <= k)` with the shift count in a scalar register
        //  lost a few windows in 10^4, differently from run to run, inside k_hist / k_scatter on gfx950 / ROCm 7.2 -- and not in
        //  a stand-alone kernel; the five fixed steps below are exact everywhere, see tools/micro/smear_test.hip.)
        u64 bad = invwin;
        int rem = k - 1;                                      // bases still to cover: steps of 1, 2, 4, 8, 16 (or what is left)
#pragma unroll
        for (int st = 1; st <= 16; st <<= 1) { const int sh = rem < st ? rem : st; bad |= bad >> sh; rem -= sh; }
        vmask = __brev(~(u32)(bad >> (16 - t0)) & 0xFFFFu) >> 16;
    }
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int t = t0 + j;
        const u64 c = (cur >> (62 - 2 * t)) & 3ull;
        fwd = ((fwd << 2) | c) & kmask;
        rc = (rc >> 2) | ((c ^ 2ull) << rcs);
        canon[j] = fwd < rc ? fwd : rc;
        if (NP != 16 && (invwin & (kbits << (31 - t))) == 0) vmask |= (1u << j);
    }
    return vmask;
}

template <int NP>
__device__ __forceinline__ u32 gen_kmers1(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                          u64 wi, int t0, int k, u64 (&canon)[NP]) {
    return gen_kmers1_words<NP>(packed[wi], wi ? packed[wi - 1] : 0ull, inval[wi], wi ? inval[wi - 1] : 0xFFFFFFFFu, t0, k, canon);
}

// ---------------------------------------------------------------- two-word k-mers (33 <= k <= 64)

// Windows ending at bases 32*wi + t0 + j, j < NP.  Needs packed[wi-2..wi].
// (the words of the frame already in registers: cur / p1 / p2 = packed[wi], [wi-1], [wi-2]; ic / i1 / i2 = their invalid masks)
template <int NP>
__device__ __forceinline__ u32 gen_kmers2_words(u64 cur, u64 p1, u64 p2, u32 ic, u32 i1, u32 i2, int t0, int k, K2 (&canon)[NP]) {
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
        canon[j].w[1] = fl ? fhi : rhi;
        canon[j].w[0] = fl ? flo : rlo;
        // window = bases [64+t-k+1, 64+t] of the 96-base frame -> bits (31-t) .. (31-t+k-1)
        const int b0 = 31 - t;                                // 0..31
        const u64 lo_bits = (k + b0 >= 64) ? (~0ull << b0) : (((1ull << k) - 1) << b0);
        const int over = k + b0 - 64;                         // bits spilling into inv_hi
        const u32 hi_bits = over > 0 ? ((over >= 32) ? 0xFFFFFFFFu : ((1u << over) - 1)) : 0u;
        if (((inv_lo & lo_bits) == 0) && ((inv_hi & hi_bits) == 0)) vmask |= (1u << j);
    }
    return vmask;
}
template <int NP>
__device__ __forceinline__ u32 gen_kmers2(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                          u64 wi, int t0, int k, K2 (&canon)[NP]) {
    return gen_kmers2_words<NP>(packed[wi], wi >= 1 ? packed[wi - 1] : 0ull, wi >= 2 ? packed[wi - 2] : 0ull,
                                inval[wi], wi >= 1 ? inval[wi - 1] : 0xFFFFFFFFu, wi >= 2 ? inval[wi - 2] : 0xFFFFFFFFu, t0, k, canon);
}

// ---------------------------------------------------------------- W-word k-mers (general; used for 65 <= k <= 128 with W = 4)
// Windows ending at bases 32*wi + t0 + j, j < NP.  Frame = packed[wi-W .. wi].  Same scheme as gen_kmers2,
// written over word arrays (all loops unroll; k is wave-uniform so the word-shift branches are uniform).
template <int W, int NP>
__device__ __forceinline__ u32 gen_kmersN(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                          u64 wi, int t0, int k, KN<W> (&canon)[NP]) {
    u64 p[W + 1]; u32 iv[W + 1];
#pragma unroll
    for (int q = 0; q <= W; ++q) {
        const bool in = wi >= (u64)q;
        p[q] = in ? packed[wi - q] : 0ull;
        iv[q] = in ? inval[wi - q] : 0xFFFFFFFFu;
    }
    u64 msk[W];
#pragma unroll
    for (int i = 0; i < W; ++i) { const int bits = 2 * k - 64 * i; msk[i] = bits >= 64 ? ~0ull : bits <= 0 ? 0ull : ((1ull << bits) - 1); }
    // forward value of the 32W bases ending just before t0, cut to k bases
    u64 f[W];
#pragma unroll
    for (int i = 0; i < W; ++i) f[i] = (t0 ? ((p[i + 1] << (2 * t0)) | (p[i] >> (64 - 2 * t0))) : p[i + 1]) & msk[i];
    // reverse complement: reverse all 32W pairs, shift right by 64W - 2k bits, complement k pairs
    u64 r[W];
#pragma unroll
    for (int i = 0; i < W; ++i) r[i] = rev_pairs(f[W - 1 - i]);
    {
        const int sh = 64 * W - 2 * k;
        const int ws = sh >> 6, bs = sh & 63;
#pragma unroll
        for (int s = 0; s < W - 1; ++s)
            if (ws > s) {
#pragma unroll
                for (int i = 0; i < W - 1; ++i) r[i] = r[i + 1];
                r[W - 1] = 0ull;
            }
        if (bs) {
#pragma unroll
            for (int i = 0; i < W - 1; ++i) r[i] = (r[i] >> bs) | (r[i + 1] << (64 - bs));
            r[W - 1] >>= bs;
        }
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] ^= (0xAAAAAAAAAAAAAAAAULL & msk[i]);
    }
    const int rcs = 2 * k - 2, rw = rcs >> 6, rb = rcs & 63;   // where the newest complement enters
    // valid bases that end just before base t0 of this word: the bases of this word below t0, then whole words back
    int run = 0;
    {
        const u32 head = t0 ? (iv[0] >> (32 - t0)) : 0u;         // invalid bits of bases 0..t0-1 (bit i <-> base t0-1-i)
        if (t0 && head) run = __builtin_ctz(head);
        else {
            run = t0;
#pragma unroll
            for (int q = 1; q <= W; ++q) {
                if (run >= k) break;
                if (iv[q] == 0u) run += 32; else { run += __builtin_ctz(iv[q]); break; }
            }
        }
    }
    u32 vmask = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int t = t0 + j;
        const u64 c = (p[0] >> (62 - 2 * t)) & 3ull;
#pragma unroll
        for (int i = W - 1; i >= 1; --i) f[i] = ((f[i] << 2) | (f[i - 1] >> 62)) & msk[i];
        f[0] = ((f[0] << 2) | c) & msk[0];
#pragma unroll
        for (int i = 0; i < W - 1; ++i) r[i] = (r[i] >> 2) | (r[i + 1] << 62);
        r[W - 1] >>= 2;
#pragma unroll
        for (int i = 0; i < W; ++i) r[i] |= (i == rw) ? ((c ^ 2ull) << rb) : 0ull;
        bool lt = false, decided = false;                        // f < r, most significant word first
#pragma unroll
        for (int i = W - 1; i >= 0; --i) { if (!decided && f[i] != r[i]) { lt = f[i] < r[i]; decided = true; } }
#pragma unroll
        for (int i = 0; i < W; ++i) canon[j].w[i] = lt ? f[i] : r[i];
        // validity: `run` = valid bases ending at this one (capped walk-back at the start, then one update per base)
        run = ((iv[0] >> (31 - t)) & 1u) ? 0 : run + 1;
        const bool bad = run < k;
        if (!bad) vmask |= (1u << j);
    }
    return vmask;
}
