// superkmer.h -- super-k-mer records for the multi-GPU exchange (gfx950, wave64).
//
// DSK v2 partitions k-mers by MINIMIZER and spills runs of consecutive k-mers that share one
// ("super-k-mers") 2-bit packed (CHANGELOG.md:13; gatb-core's Sequence2SuperKmer, named at
// scripts/quick-build.sh:53-57).  Here the same idea is the wire format between GPUs: the owner of
// a k-mer is a function of the minimizer of its window, so consecutive windows mostly share an
// owner and travel as one record of packed bases instead of one 8/16-byte key per k-mer.
//
//   sender   : k_sk_hist -> scan -> k_sk_scatter      (reads only the 2-bit stream; never forms k-mers)
//   receiver : k_sk_count -> prefix -> k_sk_expand    (records -> dense array of mixed keys)
//
// Minimizer order: the m-mers of a window are compared by a 32-bit hash of their canonical value
// (a random order balances the owners better than the lexicographic one, which favours poly-A).
// Owner map = a REPARTITION TABLE over SK_BUCKETS minimizer buckets (bucket = bits 15..4 of the winning hash),
// gatb-core's minimizer repartition (src/DSK.cpp:63 getConfig; the `minimRepart` table) on the GPU:
//   table[bucket] = owner in [0, G)   or   SK_SPLIT: the bucket is too heavy for any single owner (poly-A, microsatellites):
//                                          each of its windows goes to the owner of its own K-MER (hash of the canonical
//                                          k-mer, as in the explicit-key exchange) and travels as a one-k-mer record.
// The default table scales the bucket to [0, G); a balanced one is built from sampled bucket loads summed over all ranks
// (k_sk_sample -> dskgpu_mg_make_table: heavy buckets split, the others placed largest first on the least loaded owner).
// Results never depend on the table: any function of the window that is the same on every rank is a valid owner map.
//
// Record = R 64-bit words (R = 2 for k <= 45, 3 for k <= 64):
//   bases  : n + k - 1 bases, 2 bits each, first base in bits 63..62 of word 0, continuing MSB first (what follows the last k-mer's
//            last base, up to the 64 R - 8 bits a record holds, is unspecified: no receiver looks there)
//   header : low 8 bits of word R-1 = n, the number of k-mers (1..sk_record_nmax(k) <= SK_MAXN)
// One thread owns 16 consecutive window end positions; the two threads of a packed word (32 window ends) join their runs where the
// run goes on across the middle of the word (round 6: sk_join_pairs), so a record never leaves its word's frame of 96 bases and
// holds up to min(32, (64 R - 8) / 2 - k + 1) k-mers -- 30 at k = 31 and k = 63 -- instead of 16: 8.5 k-mers per record on average
// instead of 6.9, a fifth fewer bytes to write, to send over xGMI and to read back in level 1.
#pragma once
#include "kmer_device.h"

#define SK_NT 512
#define SK_HALO 4                         // leading groups of a tile that only contribute m-mer hashes (64 positions >= k - m)
#define SK_GROUPS (SK_NT - SK_HALO)       // groups (16 window ends each) a tile emits records for
#define SK_MAX_OWNERS 64
#define SK_BUCKETS 4096                   // minimizer buckets of the repartition table
#define SK_DESC 256                       // records a wave deals out per round (k_sk_scatter)
#define SK_SPLIT 255u                     // table entry: route the window by its k-mer, not by its minimizer
#define SK_MAXN 32                        // most k-mers of a record (slots per record of the receivers' maps)

struct SkParams {
    u64 ngroups;          // 2 * packed words
    u64 ntiles;
    u32 tiles_per_chunk, nchunks;
    u32 k, m, G, R;
    u32 sample_step;      // k_sk_hist: look at every sample_step-th tile only (1 = exact count)
    u32 slice;            // k_sk_scatter<true>: records per (owner, chunk) slice
    // k_sk_scatter<true> writes the chunks [c0, c0 + gridDim) of a layout GROUP of clen chunks that starts at chunk c0g and at record
    // rbase of the send buffer: owner o of the group starts at rbase + o * clen * slice, its chunk c at + (c - c0g) * slice (all 64-bit).
    // One group = all chunks (c0 = c0g = 0, clen = nchunks, rbase = 0): the layout of a whole step; S groups: a step sent in S
    // slices, each complete -- and on its way -- before the next is written (dskgpu_mg_scatter_slice).
    u32 c0, c0g, clen;
    u64 rbase;                    // (64-bit: a rank's shard of a 90 Gbp job holds more than 2^32 records' worth of slices)
    const unsigned char* table;   // SK_BUCKETS owners (device memory)
    u32 has_split;                // the table holds SK_SPLIT entries (set with the table: the kernels skip the split bookkeeping otherwise)
    // k_sk_scatter<true> for the passes of a multi-pass count on ONE GPU ("virtual owners": owner = pass; dskgpu.hip: rec_l0_*): only
    // the records of owners [olo, ohi) are written (a sweep materialises as many passes as HBM holds), every owner has its own slice
    // length oslice[o] (a pass that holds a k-mer with 10^8 occurrences gets longer slices, the others do not pay for it) and its
    // region starts at record obase[o] of the buffer (64-bit: a sweep holds more than 2^32 records).  oslice == nullptr: the
    // uniform layout above, all owners.
    u32 olo, ohi;
    const u32* oslice; const unsigned long long* obase;
};

__host__ __device__ __forceinline__ u32 sk_record_words(u32 k) { return (2u * (k + 15u) + 8u + 63u) / 64u; }
// most k-mers a record of R words holds at this k: n + k - 1 bases + the 8 header bits in 64 R bits, and never more than one word's 32 windows
__host__ __device__ __forceinline__ u32 sk_record_nmax(u32 k, u32 R) { const u32 n = (64u * R - 8u) / 2u + 1u - k; return n < (u32)SK_MAXN ? n : (u32)SK_MAXN; }

__device__ __forceinline__ u32 fmix32(u32 h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}

__device__ __forceinline__ void sk_lds_barrier() { __asm__ volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// What one thread knows about its 16 windows after the minimizer phase.
struct SkThread {
    u32 vm;               // bit i: window i is a valid k-mer
    u32 bm;               // bit i: a record starts at window i
    u64 ow_lo, ow_hi;     // owner of window i in byte i (lo: 0..7, hi: 8..15)
    u32 sp;               // bit i: window i is routed by its own k-mer (a split bucket): always a record of its own
};

__device__ __forceinline__ u32 sk_owner(const SkThread& s, int i) {
    return (u32)((i < 8 ? s.ow_lo >> (8 * i) : s.ow_hi >> (8 * (i - 8))) & 0xFFu);
}

__device__ __forceinline__ u64 sk_key1(const u64* r, int j, int k);
__device__ __forceinline__ K2 sk_key2(const u64* r, int j, int k);
// bases [bs, bs + nb) of the 96-base frame (w2 : w1 : w0), shifted to the top of o[0..2], the rest cleared
__device__ __forceinline__ void sk_extract(u64 w2, u64 w1, u64 w0, int bs, int nb, u64 (&o)[3]) {
    const int sh = 2 * bs, ws = sh >> 6, b = sh & 63;
    const u64 a = ws == 0 ? w2 : ws == 1 ? w1 : w0;
    const u64 bb = ws == 0 ? w1 : ws == 1 ? w0 : 0ull;
    const u64 cc = ws == 0 ? w0 : 0ull;
    o[0] = b ? (a << b) | (bb >> (64 - b)) : a;
    o[1] = b ? (bb << b) | (cc >> (64 - b)) : bb;
    o[2] = b ? (cc << b) : cc;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int keep = 2 * nb - 64 * j;
        o[j] = keep <= 0 ? 0ull : keep >= 64 ? o[j] : (o[j] & ~(~0ull >> keep));
    }
}
// owner of ONE k-mer (the window ending at base t0 + i of word wi): the same bit field of the mixed canonical k-mer as the
// explicit-key exchange uses (kernels.h key_digit<0>), so a split bucket spreads over all owners
__device__ __forceinline__ u32 sk_kmer_owner(const u64* __restrict__ packed, u64 wi, int t0, int i, int k, u32 G) {
    const u64 w0 = packed[wi];
    const u64 w1 = wi >= 1 ? packed[wi - 1] : 0ull;
    const u64 w2 = wi >= 2 ? packed[wi - 2] : 0ull;
    u64 o[3];
    sk_extract(w2, w1, w0, 64 + t0 + i - k + 1, k, o);
    const u64 h = k <= 32 ? sk_key1(o, 0, k) : sk_key2(o, 0, k).w[1];
    return (((u32)(h >> 12) & 0xFFFFFu) * G) >> 20;
}

// Minimizer phase of one tile.  Thread t handles group gfirst + t.  H = SK_NT * 17 words of LDS; tab = the repartition
// table in LDS.  SAMPLE: no records -- every valid window adds 1 to load[bucket] (LDS), for the table builder.
template <bool SAMPLE = false>
__device__ __forceinline__ SkThread sk_tile(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                            const SkParams& sp, long long gfirst, u32* H, const unsigned char* tab, u32* load = nullptr) {
    const int t = threadIdx.x;
    const long long g = gfirst + t;
    const bool live = g >= 0 && (u64)g < sp.ngroups;
    const int k = (int)sp.k, m = (int)sp.m;
    u32 h[16];
    u64 wi = 0; int t0 = 0;
    if (live) {
        wi = (u64)g >> 1; t0 = (int)(g & 1) << 4;
        const u64 cur = packed[wi];
        const u64 prev = wi ? packed[wi - 1] : 0ull;
        // x = the 32 bases ending at the group's last position, first base most significant
        const u64 x = t0 ? cur : ((prev << 32) | (cur >> 32));
        const u64 rcx = rev_pairs(x) ^ 0xAAAAAAAAAAAAAAAAull;
        const u32 mmask = (m == 16) ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 fw = (u32)(x >> (2 * (15 - j))) & mmask;          // m-mer ending at position j of the group
            const u32 rv = (u32)(rcx >> (2 * (17 + j - m))) & mmask;    // its reverse complement
            h[j] = fmix32(fw < rv ? fw : rv);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) h[j] = 0xFFFFFFFFu;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) H[17 * t + j] = h[j];                   // 1 pad word per 16: lane stride 17, conflict free
    sk_lds_barrier();
    SkThread r; r.vm = 0; r.bm = 0; r.ow_lo = 0; r.ow_hi = 0; r.sp = 0;
    if (live && t >= SK_HALO) {
        const int w = k - m + 1;                                         // m-mers per window, 16 <= w <= 64
        const int q0 = 16 * t;                                           // tile-local position of window 0
        // window i covers hash positions [q0 + i - w + 1, q0 + i]:
        //   L = [q0 - w + 1, q0 - w + 15] (suffix from i), common = [q0 - w + 16, q0], R = own h[1..i]
        u32 sl[15];
#pragma unroll
        for (int j = 0; j < 15; ++j) { const int i = q0 - w + 1 + j; sl[j] = H[i + (i >> 4)]; }
#pragma unroll
        for (int j = 13; j >= 0; --j) sl[j] = sl[j] < sl[j + 1] ? sl[j] : sl[j + 1];
        u32 cm = h[0];
        for (int c = q0 - w + 16; c < q0; ++c) { const u32 v = H[c + (c >> 4)]; cm = v < cm ? v : cm; }
        // validity of the 16 windows: no invalid base among the last k (frame of 96 bases: words wi-2 .. wi)
        const u32 ic = inval[wi];
        const u32 i1 = wi >= 1 ? inval[wi - 1] : 0xFFFFFFFFu;
        const u32 i2 = wi >= 2 ? inval[wi - 2] : 0xFFFFFFFFu;
        // a window is invalid if an invalid base lies among its k: smear every invalid bit of the frame (i2 : i1 : ic, bit 31 - p
        // of ic <-> base p of word wi) over the k - 1 following bases in log steps, then window i is bit 31 - (t0 + i)
        u64 bad_lo = ((u64)i1 << 32) | ic, bad_hi = i2;
        {
            int rem = k - 1;
#pragma unroll
            for (int st = 1; st <= 32; st <<= 1) {
                const int sh = rem < st ? rem : st;
                if (sh) { bad_lo |= (bad_lo >> sh) | (bad_hi << (64 - sh)); bad_hi |= bad_hi >> sh; }
                rem -= sh;
            }
        }
        const u32 bad16 = (u32)(bad_lo >> (16 - t0)) & 0xFFFFu;              // bit 15 - i: window i holds an invalid base
        u32 pr = 0xFFFFFFFFu;
        if constexpr (SAMPLE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i > 0) pr = h[i] < pr ? h[i] : pr;
                u32 mn = cm < pr ? cm : pr;
                if (i < 15) mn = sl[i] < mn ? sl[i] : mn;
                if (!((bad16 >> (15 - i)) & 1u)) atomicAdd(&load[(mn & 0xFFFFu) >> 4], 1u);
            }
        } else {
            // The loop only looks the owners up and packs them, four to a word; validity, "this window is routed by its own k-mer" and
            // "the owner changes here" then come from a few word-wide operations instead of a chain of conditions per window (the
            // kernel is bound by its instruction count: 73 % VALU busy).
            u32 ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i > 0) pr = h[i] < pr ? h[i] : pr;
                u32 mn = cm < pr ? cm : pr;
                if (i < 15) mn = sl[i] < mn ? sl[i] : mn;
                ow[i >> 2] |= (u32)tab[(mn & 0xFFFFu) >> 4] << (8 * (i & 3));
            }
            r.vm = __brev(~bad16 & 0xFFFFu) >> 16;                             // bit i: window i is a valid k-mer (bad16 runs the other way)
            // bit j of nib(w) = byte j of w has its low bit set (bytes hold 0 or 1): one multiply gathers the four bits
            auto nib = [](u32 w) { return (w * 0x10204080u) >> 28; };
            u32 splitm = 0;                                                    // table entry SK_SPLIT (255): the only bytes with bit 7 set
#pragma unroll
            for (int q = 0; q < 4; ++q) splitm |= nib((ow[q] >> 7) & 0x01010101u) << (4 * q);
            splitm &= r.vm;
            u64 lo = ((u64)ow[1] << 32) | ow[0], hi = ((u64)ow[3] << 32) | ow[2];
            for (u32 sm = splitm; sm; sm &= sm - 1) {                          // (rare: heavy buckets only) the owner of the window's own k-mer
                const int i = __builtin_ctz(sm);
                const u64 o = sk_kmer_owner(packed, wi, t0, i, k, sp.G);
                if (i < 8) lo = (lo & ~(0xFFull << (8 * i))) | (o << (8 * i));
                else hi = (hi & ~(0xFFull << (8 * (i - 8)))) | (o << (8 * (i - 8)));
            }
            lo &= 0x3F3F3F3F3F3F3F3Full; hi &= 0x3F3F3F3F3F3F3F3Full;         // owners are < 64; what is left of a 255 belongs to a window without a k-mer
            r.ow_lo = lo; r.ow_hi = hi;
            // d bit i (i >= 1): owner of window i differs from the owner of window i - 1
            const u32 w0 = (u32)lo, w1 = (u32)(lo >> 32), w2 = (u32)hi, w3 = (u32)(hi >> 32);
            auto diff = [&](u32 w, u32 prev_top) { const u32 x = w ^ ((w << 8) | prev_top); return nib(((x + 0x3F3F3F3Fu) >> 6) & 0x01010101u); };
            const u32 d = diff(w0, 0u) | (diff(w1, w0 >> 24) << 4) | (diff(w2, w1 >> 24) << 8) | (diff(w3, w2 >> 24) << 12);
            // a record starts at a valid window whose predecessor is not valid, has another owner, or when either is routed by its k-mer
            r.bm = r.vm & (~(r.vm << 1) | d | splitm | (splitm << 1)) & 0xFFFFu;
            r.sp = splitm;
        }
    }
    return r;
}

// ---------------------------------------------------------------- the same tile with k and m known at compile time
// Round 6 (profiles/r06_records.md): the sender is paced by its instruction count -- 64 VALU instructions per window, the vector
// ALUs busy 76 % of the kernel with every issue port taken together above 100 % -- and writes exactly its records (WRITE_SIZE =
// 1.00 x), so what pays is fewer instructions.  With k and m fixed (the BASELINE configs: k = 31 and k = 63, m = 10) every shift is
// an immediate, the hashes of the neighbouring groups come out of LDS as 128-bit reads (lane stride 20 words: 16-byte aligned and
// conflict-free for the eight lanes of a 128-bit access group) and the sliding-window minimum is one running minimum over the
// w - 1 positions to the left, nearest first: pm[x] = min over the x nearest, window i = min(own prefix, pm[w - 1 - i]).
// Same SkThread, bit for bit, as sk_tile (tests: the records of both paths are compared byte for byte).
#define SK_HSTRIDE 20                     // u32 words per thread in the hash exchange area of sk_tile_fx
template <int K, int M, bool SAMPLE = false>
__device__ __forceinline__ SkThread sk_tile_fx(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                               const SkParams& sp, long long gfirst, u32* H, const unsigned char* tab, u32* load = nullptr) {
    constexpr int W = K - M + 1;              // m-mers per window
    constexpr int NL = W - 1;                 // positions left of the group's first one that its windows reach
    static_assert(W >= 16 && NL <= 16 * SK_HALO && M >= 1 && M <= 16, "window geometry");
    const int t = threadIdx.x;
    const long long g = gfirst + t;
    const bool live = g >= 0 && (u64)g < sp.ngroups;
    u32 h[16];
    u64 wi = 0; int t0 = 0;
    if (live) {
        wi = (u64)g >> 1; t0 = (int)(g & 1) << 4;
        const u64 cur = packed[wi];
        const u64 prev = wi ? packed[wi - 1] : 0ull;
        const u64 x = t0 ? cur : ((prev << 32) | (cur >> 32));
        const u64 rcx = rev_pairs(x) ^ 0xAAAAAAAAAAAAAAAAull;
        constexpr u32 mmask = (M == 16) ? 0xFFFFFFFFu : ((1u << (2 * M)) - 1u);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const u32 fw = (u32)(x >> (2 * (15 - j))) & mmask;
            const u32 rv = (u32)(rcx >> (2 * (17 + j - M))) & mmask;
            h[j] = fmix32(fw < rv ? fw : rv);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j) h[j] = 0xFFFFFFFFu;
    }
    uint4* Hv = reinterpret_cast<uint4*>(H);
#pragma unroll
    for (int q = 0; q < 4; ++q) Hv[5 * t + q] = make_uint4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
    sk_lds_barrier();
    SkThread r; r.vm = 0; r.bm = 0; r.ow_lo = 0; r.ow_hi = 0; r.sp = 0;
    if (live && t >= SK_HALO) {
        // pm[x] = min over the NL - 15 + x nearest left positions (x = 0 with NL == 15: none)
        u32 pm[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) pm[x] = 0xFFFFFFFFu;
        {
            u32 run = 0xFFFFFFFFu;
            constexpr int NG = (NL + 15) / 16;
#pragma unroll
            for (int gq = 1; gq <= NG; ++gq) {
#pragma unroll
                for (int q = 3; q >= 0; --q) {
                    if (16 * gq - (4 * q + 3) > NL) continue;                   // (constant: the whole quad lies beyond the windows' reach)
                    const uint4 v = Hv[5 * (t - gq) + q];
                    const u32 e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 3; c >= 0; --c) {
                        const int d = 16 * gq - (4 * q + c);                    // distance of this position from the group's first one
                        if (d > NL) continue;
                        run = e[c] < run ? e[c] : run;
                        if (d >= NL - 15 && d >= 1) pm[d - (NL - 15)] = run;
                    }
                }
            }
        }
        u32 mn[16];
        {
            u32 pr = h[0];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (i > 0) pr = h[i] < pr ? h[i] : pr;
                mn[i] = pm[15 - i] < pr ? pm[15 - i] : pr;
            }
        }
        const u32 ic = inval[wi];
        const u32 i1 = wi >= 1 ? inval[wi - 1] : 0xFFFFFFFFu;
        const u32 i2 = wi >= 2 ? inval[wi - 2] : 0xFFFFFFFFu;
        u64 bad_lo = ((u64)i1 << 32) | ic, bad_hi = i2;
        {
            // (only bits 16 - t0 .. 31 - t0 of the smeared mask are looked at, and a bit travels k - 1 places down: for k <= 33 nothing
            //  of the frame's third word can reach them)
            int rem = K - 1;
#pragma unroll
            for (int st = 1; st <= 32; st <<= 1) {
                const int sh = rem < st ? rem : st;
                if (sh) {
                    if constexpr (K <= 33) bad_lo |= bad_lo >> sh;
                    else { bad_lo |= (bad_lo >> sh) | (bad_hi << (64 - sh)); bad_hi |= bad_hi >> sh; }
                }
                rem -= sh;
            }
        }
        const u32 bad16 = (u32)(bad_lo >> (16 - t0)) & 0xFFFFu;
        if constexpr (SAMPLE) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (!((bad16 >> (15 - i)) & 1u)) atomicAdd(&load[(mn[i] & 0xFFFFu) >> 4], 1u);
        } else {
            u32 ow[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int i = 0; i < 16; ++i) ow[i >> 2] |= (u32)tab[(mn[i] & 0xFFFFu) >> 4] << (8 * (i & 3));
            r.vm = __brev(~bad16 & 0xFFFFu) >> 16;
            auto nib = [](u32 w) { return (w * 0x10204080u) >> 28; };
            u32 splitm = 0;
            u64 lo = ((u64)ow[1] << 32) | ow[0], hi = ((u64)ow[3] << 32) | ow[2];
            if (sp.has_split) {                                               // (uniform: most tables route every bucket by its minimizer)
#pragma unroll
                for (int q = 0; q < 4; ++q) splitm |= nib((ow[q] >> 7) & 0x01010101u) << (4 * q);
                splitm &= r.vm;
                for (u32 sm = splitm; sm; sm &= sm - 1) {
                    const int i = __builtin_ctz(sm);
                    const u64 o = sk_kmer_owner(packed, wi, t0, i, K, sp.G);
                    if (i < 8) lo = (lo & ~(0xFFull << (8 * i))) | (o << (8 * i));
                    else hi = (hi & ~(0xFFull << (8 * (i - 8)))) | (o << (8 * (i - 8)));
                }
                lo &= 0x3F3F3F3F3F3F3F3Full; hi &= 0x3F3F3F3F3F3F3F3Full;  // what is left of a 255 belongs to a window without a k-mer
            }
            r.ow_lo = lo; r.ow_hi = hi;
            const u32 w0 = (u32)lo, w1 = (u32)(lo >> 32), w2 = (u32)hi, w3 = (u32)(hi >> 32);
            auto diff = [&](u32 w, u32 prev_top) { const u32 x = w ^ ((w << 8) | prev_top); return nib(((x + 0x3F3F3F3Fu) >> 6) & 0x01010101u); };
            const u32 d = diff(w0, 0u) | (diff(w1, w0 >> 24) << 4) | (diff(w2, w1 >> 24) << 8) | (diff(w3, w2 >> 24) << 12);
            r.bm = r.vm & (~(r.vm << 1) | d | splitm | (splitm << 1)) & 0xFFFFu;
            r.sp = splitm;
        }
    }
    return r;
}
// K == 0: k and m at run time (sk_tile); H holds SK_NT * SK_HSTRIDE words either way
template <int K, int M, bool SAMPLE = false>
__device__ __forceinline__ SkThread sk_tile_any(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                const SkParams& sp, long long gfirst, u32* H, const unsigned char* tab, u32* load = nullptr) {
    if constexpr (K != 0) return sk_tile_fx<K, M, SAMPLE>(packed, inval, sp, gfirst, H, tab, load);
    else return sk_tile<SAMPLE>(packed, inval, sp, gfirst, H, tab, load);
}

// The two threads of a packed word (lanes 2j and 2j + 1: window ends 0..15 and 16..31 of the word) join their runs: when the even
// lane's last run reaches its window 15, the odd lane's window 0 goes on with the same owner (neither routed by its k-mer) and both
// together stay within nmax k-mers, the odd lane's first record becomes part of the even lane's last one -- its start bit is
// cleared, and the even lane learns how many k-mers its last record gains (-> ext; 0 for odd lanes and where nothing is joined).
// Called by ALL lanes of a wave (two DPP exchanges inside the lane pair: no LDS, no divergence).
__device__ __forceinline__ u32 sk_join_pairs(SkThread& s, u32 nmax) {
    const bool odd = threadIdx.x & 1u;
    // what the even lane tells: trailing run length (its last start bit .. window 15), owner of window 15, valid / split there
    const u32 bm16 = s.bm & 0xFFFFu;
    const u32 le = bm16 ? (u32)(__builtin_clz(bm16) - 15) : 0u;                      // 16 - index of the highest start bit (1..16); 0: no record at all
    const u32 tell = le | ((u32)(s.ow_hi >> 56) & 0x3Fu) << 8 | ((s.vm >> 15) & 1u) << 16 | ((s.sp >> 15) & 1u) << 17;
    const u32 heard = (u32)__builtin_amdgcn_mov_dpp((int)tell, 0xB1, 0xF, 0xF, true);            // quad_perm [1, 0, 3, 2]: the pair partner's word
    // the odd lane decides
    const u32 stop = ((((~s.vm) | s.bm) & 0xFFFFu) | 0x10000u) >> 1;                 // first run of the odd lane: up to its next start / window without a k-mer
    const u32 lo = (u32)__builtin_ctz(stop) + 1u;
    const bool join = odd && (s.vm & 1u) && !(s.sp & 1u) && ((heard >> 16) & 1u) && !((heard >> 17) & 1u)
                      && ((heard >> 8) & 0x3Fu) == ((u32)s.ow_lo & 0x3Fu) && (heard & 0xFFu) != 0u && (heard & 0xFFu) + lo <= nmax;
    if (join) s.bm &= ~1u;
    const u32 gain = join ? lo : 0u;
    const u32 ext = (u32)__builtin_amdgcn_mov_dpp((int)gain, 0xB1, 0xF, 0xF, true);               // even lane: what its partner handed over
    return odd ? 0u : ext;
}

// number of k-mers of the record starting at window i
__device__ __forceinline__ u32 sk_run_length(const SkThread& s, int i) {
    const u32 stop = ((~s.vm | s.bm) & 0xFFFFu) >> (i + 1);
    return stop ? (u32)__builtin_ctz(stop) + 1u : (u32)(16 - i);
}

// ---------------------------------------------------------------- sender: records per (owner, chunk)
// kmers[o] += k-mers inside the records counted for owner o (with sample_step > 1: of the sampled tiles -- the estimate that
// sizes the receivers of a sliced step before any record exists)
// <K, M>: k and m at compile time (sk_tile_fx); <0, 0>: at run time
template <int K = 0, int M = 0>
__global__ __launch_bounds__(SK_NT) void k_sk_hist(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                   SkParams sp, u32* __restrict__ mat, unsigned long long* __restrict__ kmers) {
    __shared__ __attribute__((aligned(16))) u32 H[SK_NT * SK_HSTRIDE];
    __shared__ u32 cnt[SK_MAX_OWNERS];
    __shared__ u32 kcn[SK_MAX_OWNERS];
    __shared__ unsigned char tab[SK_BUCKETS];
    const u32 c = blockIdx.x;
    if (threadIdx.x < SK_MAX_OWNERS) { cnt[threadIdx.x] = 0; kcn[threadIdx.x] = 0; }
    for (int i = threadIdx.x; i < SK_BUCKETS / 8; i += SK_NT) reinterpret_cast<u64*>(tab)[i] = reinterpret_cast<const u64*>(sp.table)[i];
    __syncthreads();
    const u64 tbeg = (u64)c * sp.tiles_per_chunk;
    const u64 tend = tbeg + sp.tiles_per_chunk < sp.ntiles ? tbeg + sp.tiles_per_chunk : sp.ntiles;
    const u32 nmax = sk_record_nmax(K ? (u32)K : sp.k, sp.R);
    for (u64 tile = tbeg; tile < tend; tile += sp.sample_step) {
        SkThread s = sk_tile_any<K, M>(packed, inval, sp, (long long)(tile * SK_GROUPS) - SK_HALO, H, tab);
        const u32 ext = sk_join_pairs(s, nmax);
        u32 bm = s.bm;
        while (bm) {
            const int i = __builtin_ctz(bm); bm &= bm - 1;
            const u32 len = sk_run_length(s, i);
            atomicAdd(&cnt[sk_owner(s, i)], 1u);
            atomicAdd(&kcn[sk_owner(s, i)], len + ((u32)i + len == 16u ? ext : 0u));
        }
        sk_lds_barrier();
    }
    __syncthreads();
    if (threadIdx.x < sp.G) {
        mat[(u64)threadIdx.x * sp.nchunks + c] = cnt[threadIdx.x];
        if (kcn[threadIdx.x]) atomicAdd(&kmers[threadIdx.x], (unsigned long long)kcn[threadIdx.x]);
    }
}

// ---------------------------------------------------------------- repartition: sampled k-mer load per minimizer bucket
// Every sample_step-th tile; load[] += the number of valid windows whose minimizer falls into the bucket.
template <int K = 0, int M = 0>
__global__ __launch_bounds__(SK_NT) void k_sk_sample(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                     SkParams sp, unsigned long long* __restrict__ gload) {
    __shared__ __attribute__((aligned(16))) u32 H[SK_NT * SK_HSTRIDE];
    __shared__ u32 load[SK_BUCKETS];
    for (int i = threadIdx.x; i < SK_BUCKETS; i += SK_NT) load[i] = 0;
    __syncthreads();
    const u32 c = blockIdx.x;
    const u64 tbeg = (u64)c * sp.tiles_per_chunk;
    const u64 tend = tbeg + sp.tiles_per_chunk < sp.ntiles ? tbeg + sp.tiles_per_chunk : sp.ntiles;
    for (u64 tile = tbeg; tile < tend; tile += sp.sample_step) {
        (void)sk_tile_any<K, M, true>(packed, inval, sp, (long long)(tile * SK_GROUPS) - SK_HALO, H, nullptr, load);
        sk_lds_barrier();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < SK_BUCKETS; i += SK_NT) if (load[i]) atomicAdd(&gload[i], (unsigned long long)load[i]);
}

// ---------------------------------------------------------------- sender: write the records
// `cbase` (exact layout, !SLICES) holds the exclusive scan of the (owner-major) count matrix: the 64-bit record index of every
// (owner, chunk) pair.  Every position is 64-bit: a block keeps, per owner, the WORD index of its first slot (obw) and a 32-bit
// cursor relative to it -- a shard of any size goes through (30x human on 8 GPUs: 11.3 GB of reads per rank; ~1.5 * 10^9 records).
// SLICES: no exact counts -- every (owner, chunk) pair owns a slice of sp.slice records (sized from a sampled
// estimate); what a block leaves unused is filled with zero-length records (n = 0: the receiver skips them), a slice
// that would overflow raises *ovf (nothing is written past a slice) and the host repeats with exact counts.
// The bases behind a record's last k-mer (up to the 64 R - 8 bits a record holds) are whatever followed in the read stream: no
// receiver looks past k-mer n - 1, and clearing them cost 15 instructions per record in a kernel paced by its instruction count.
template <bool SLICES, int K = 0, int M = 0>
__global__ __launch_bounds__(SK_NT) void k_sk_scatter(const u64* __restrict__ packed, const u32* __restrict__ inval,
                                                      SkParams sp, const unsigned long long* __restrict__ cbase, u64* __restrict__ send, u32* __restrict__ ovf,
                                                      unsigned long long* __restrict__ kmers) {      // kmers[o] += k-mers inside the records written for owner o
    __shared__ __attribute__((aligned(16))) u32 H[SK_NT * SK_HSTRIDE];
    __shared__ u32 cur[SK_MAX_OWNERS];
    __shared__ u32 kc[SK_MAX_OWNERS];
    __shared__ u32 lim[SK_MAX_OWNERS];                    // SLICES: end of this block's slice of every owner (same units as cur)
    __shared__ unsigned long long obw[SK_MAX_OWNERS];     // WORD index of the record where cur[o] == 0 lies (this block's first slot of owner o)
    __shared__ unsigned short desc[SK_NT / 64][SK_DESC];  // lane | first window << 6 | owner << 10
    __shared__ unsigned char tab[SK_BUCKETS];
    const u32 c = blockIdx.x + (SLICES ? sp.c0 : 0u);
    for (int i = threadIdx.x; i < SK_BUCKETS / 8; i += SK_NT) reinterpret_cast<u64*>(tab)[i] = reinterpret_cast<const u64*>(sp.table)[i];
    const u32 R = sp.R;
    if (threadIdx.x < sp.G) {
        const u32 o = threadIdx.x;
        unsigned long long ob = 0ull;
        kc[o] = 0; cur[o] = 0u; lim[o] = 0u;
        if (!SLICES) ob = cbase[(u64)o * sp.nchunks + c];
        else if (sp.oslice) {          // per-owner slices inside per-owner regions: positions relative to the block's own slice
            const u32 sl = sp.oslice[o];
            lim[o] = (o >= sp.olo && o < sp.ohi) ? sl : 0u;
            ob = sp.obase[o] + (unsigned long long)(c - sp.c0g) * sl;
        } else { lim[o] = sp.slice; ob = sp.rbase + ((u64)o * sp.clen + (u64)(c - sp.c0g)) * sp.slice; }
        obw[o] = ob * R;
    }
    __syncthreads();
    bool over = false;
    const u64 tbeg = (u64)c * sp.tiles_per_chunk;
    const u64 tend = tbeg + sp.tiles_per_chunk < sp.ntiles ? tbeg + sp.tiles_per_chunk : sp.ntiles;
    const int k = K ? K : (int)sp.k;
    const u32 nmax = sk_record_nmax((u32)k, R);
    const bool sub = SLICES && sp.oslice != nullptr;      // only some owners are written
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 tile = tbeg; tile < tend; ++tile) {
        const long long gfirst = (long long)(tile * SK_GROUPS) - SK_HALO;
        SkThread s = sk_tile_any<K, M>(packed, inval, sp, gfirst, H, tab);
        const u32 ext = sk_join_pairs(s, nmax);               // (even lanes: k-mers their last record gains from the other half of the word)
        // The records of a WAVE are dealt out to its lanes, one record per lane and trip: a thread holds 0..16 records (2.3 on
        // average), and a loop over a thread's own records runs as often as the busiest of 64 lanes needs (5-6 trips).  Every
        // thread notes (lane, first window, owner) of its records in the wave's list -- a short loop body --; lane e then builds
        // record e, e + 64, ..: the run length comes from the noting lane's window masks (one cross-lane read), the frame of its
        // group from the 2-bit stream.
        u32 mybm = s.bm;
        if (sub) {      // (virtual owners: the records of owners outside [olo, ohi) are not even noted) -- byte-parallel range test, owners < 64
            auto nib = [](u32 w) { return (w * 0x10204080u) >> 28; };
            const u32 lo4 = sp.olo * 0x01010101u, hi4 = ((sp.ohi - 1u) | 0x80u) * 0x01010101u;
            auto inr = [&](u32 w) { return nib(((((w | 0x80808080u) - lo4) & (hi4 - w)) >> 7) & 0x01010101u); };
            const u32 w0 = (u32)s.ow_lo, w1 = (u32)(s.ow_lo >> 32), w2 = (u32)s.ow_hi, w3 = (u32)(s.ow_hi >> 32);
            mybm &= inr(w0) | (inr(w1) << 4) | (inr(w2) << 8) | (inr(w3) << 12);
        }
        const u32 vmbm = s.vm | (s.bm << 16);
        const u32 ow0 = (u32)s.ow_lo, ow1 = (u32)(s.ow_lo >> 32), ow2 = (u32)s.ow_hi, ow3 = (u32)(s.ow_hi >> 32);
        const u32 cnt = (u32)__popc(mybm);
        const u32 inc = wave_incl_scan(cnt);
        const u32 total = (u32)__shfl((int)inc, 63);
        for (u32 B = 0; B < total; B += SK_DESC) {                          // (wave-uniform; more than SK_DESC records in a wave: several rounds)
            u32 bm = mybm, id = inc - cnt - B;
            while (bm) {
                const u32 i = (u32)__builtin_ctz(bm); bm &= bm - 1;
                // owner of window i = byte i of (ow3 : ow2 : ow1 : ow0): a byte permute over the half it lies in
                const u32 own = (i < 8u ? __builtin_amdgcn_perm(ow1, ow0, i) : __builtin_amdgcn_perm(ow3, ow2, i - 8u)) & 0x3Fu;
                if (id < SK_DESC) desc[wave][id] = (unsigned short)((u32)lane | (i << 6) | (own << 10));
                ++id;
            }
            __builtin_amdgcn_wave_barrier();          // (the LDS executes a wave's accesses in order: no wait needed between its lanes)
            const u32 m = total - B < SK_DESC ? total - B : SK_DESC;
            for (u32 e0 = 0; e0 < m; e0 += 64) {                              // (wave-uniform trips: the cross-lane read below needs its SOURCE lane active)
                const u32 e = e0 + (u32)lane;
                const u32 dsc = desc[wave][e < m ? e : 0u];
                const u32 L = dsc & 63u, i = (dsc >> 6) & 15u, own = dsc >> 10;
                const u32 vb = (u32)__shfl((int)vmbm, (int)L);
                const u32 xt = (u32)__shfl((int)ext, (int)L);
                if (e >= m) continue;
                // k-mers of the record that starts at window i: up to the next start, the next window without a k-mer, or the group's end
                const u32 stop = ((((~vb) | (vb >> 16)) & 0xFFFFu) | 0x10000u) >> (i + 1);
                u32 n = (u32)__builtin_ctz(stop) + 1u;
                if (i + n == 16u) n += xt;                        // the run goes on in the other half of the word (sk_join_pairs)
                const u64 g = (u64)(gfirst + (long long)((wave << 6) + (int)L));
                const u64 wi = g >> 1; const int t0 = (int)(g & 1) << 4;
                const u64 w0 = packed[wi];
                const u64 w1 = wi >= 1 ? packed[wi - 1] : 0ull;
                const u64 w2 = wi >= 2 ? packed[wi - 2] : 0ull;
                const u32 slot = atomicAdd(&cur[own], 1u);
                if (SLICES && slot >= lim[own]) { over = true; continue; }
                atomicAdd(&kc[own], n);
                // the record's bases start at base bs of the 96-base frame (w2 : w1 : w0): the frame shifted left by 2 * bs bits
                const int sh = 2 * (64 + t0 + (int)i - k + 1), ws = sh >> 6, b = sh & 63;
                const u64 fa = ws == 0 ? w2 : ws == 1 ? w1 : w0;
                const u64 fb = ws == 0 ? w1 : ws == 1 ? w0 : 0ull;
                const u64 fc = ws == 0 ? w0 : 0ull;
                const u64 o0 = (fa << b) | ((fb >> 1) >> (63 - b));
                const u64 o1 = (fb << b) | ((fc >> 1) >> (63 - b));
                u64* dst = send + obw[own] + (u64)slot * R;
                if (R == 2) { dst[0] = o0; dst[1] = (o1 & ~0xFFull) | n; }
                else { dst[0] = o0; dst[1] = o1; dst[2] = ((fc << b) & ~0xFFull) | n; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        sk_lds_barrier();
    }
    __syncthreads();
    if (threadIdx.x < sp.G && kc[threadIdx.x]) atomicAdd(&kmers[threadIdx.x], (unsigned long long)kc[threadIdx.x]);
    if (SLICES) {
        if (over) *ovf = 1u;
        for (u32 o = 0; o < sp.G; ++o) {                      // zero-length records up to the end of each of this block's slices
            const u64 beg = obw[o] + (u64)(cur[o] < lim[o] ? cur[o] : lim[o]) * R, end = obw[o] + (u64)lim[o] * R;
            for (u64 w = beg + threadIdx.x; w < end; w += SK_NT) send[w] = 0ull;
        }
    }
}

// ---------------------------------------------------------------- receiver: k-mers per chunk of records
#define SKX_NT 256
__global__ __launch_bounds__(SKX_NT) void k_sk_count(const u64* __restrict__ rec, u64 nrec, u32 R, u32 rpc, u32* __restrict__ sums) {
    __shared__ u32 ws[SKX_NT / 64];
    const u64 rbeg = (u64)blockIdx.x * rpc;
    const u64 rend = rbeg + rpc < nrec ? rbeg + rpc : nrec;
    u32 s = 0;
    for (u64 r = rbeg + threadIdx.x; r < rend; r += SKX_NT) s += (u32)(rec[r * R + R - 1] & 0xFFu);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { u32 t = 0; for (int i = 0; i < SKX_NT / 64; ++i) t += ws[i]; sums[blockIdx.x] = t; }
}

// k-mer `j` of a staged record -> mixed key (same key as tile_keys_reads would give for that window)
__device__ __forceinline__ u64 sk_key1(const u64* r, int j, int k) {
    const u64 x = j ? (r[0] << (2 * j)) | (r[1] >> (64 - 2 * j)) : r[0];
    const u64 kmask = (k == 32) ? ~0ull : ((1ull << (2 * k)) - 1);
    const u64 fwd = x >> (64 - 2 * k);
    const u64 rc = (rev_pairs(fwd) >> (64 - 2 * k)) ^ (0xAAAAAAAAAAAAAAAAull & kmask);
    return kmix(fwd < rc ? fwd : rc);
}
__device__ __forceinline__ K2 sk_key2(const u64* r, int j, int k) {
    const u64 y0 = j ? (r[0] << (2 * j)) | (r[1] >> (64 - 2 * j)) : r[0];
    const u64 y1 = j ? (r[1] << (2 * j)) | (r[2] >> (64 - 2 * j)) : r[1];
    const int sh = 128 - 2 * k;                                // 0..62
    const int kh = 2 * k - 64;
    const u64 hmask = (kh == 64) ? ~0ull : ((1ull << kh) - 1);
    const u64 flo = sh ? (y1 >> sh) | (y0 << (64 - sh)) : y1;
    const u64 fhi = sh ? (y0 >> sh) : y0;
    u64 rhi = rev_pairs(flo), rlo = rev_pairs(fhi);
    if (sh) { rlo = (rlo >> sh) | (rhi << (64 - sh)); rhi >>= sh; }
    rlo ^= 0xAAAAAAAAAAAAAAAAull;
    rhi ^= (0xAAAAAAAAAAAAAAAAull & hmask);
    const bool fl = fhi < rhi || (fhi == rhi && flo < rlo);
    K2 o; o.w[1] = fl ? fhi : rhi; o.w[0] = fl ? flo : rlo;
    kmixN(o);
    return o;
}

// ---------------------------------------------------------------- receiver: records -> dense mixed keys
// One block per chunk of records.  A tile of SKX_NT records is staged in LDS together with a slot map
// (output slot -> record, k-mer index), then every thread builds ONE k-mer per trip straight from the
// staged bases (a funnel shift + rev_pairs; no rolling, no idle lanes) and stores it coalesced.
template <int W>
__global__ __launch_bounds__(SKX_NT) void k_sk_expand(const u64* __restrict__ rec, u64 nrec, u32 R, int k, u32 rpc,
                                                      const u64* __restrict__ chunk_base, typename KeyT<W>::T* __restrict__ out) {
    __shared__ u64 srec[SKX_NT * 3];
    __shared__ unsigned short smap[SKX_NT * SK_MAXN];
    __shared__ u32 wsum[SKX_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u64 rbeg = (u64)blockIdx.x * rpc;
    const u64 rend = rbeg + rpc < nrec ? rbeg + rpc : nrec;
    u64 obase = chunk_base[blockIdx.x];
    for (u64 r0 = rbeg; r0 < rend; r0 += SKX_NT) {
        const u64 r = r0 + tid;
        u32 n = 0;
        if (r < rend) {
            const u64* p = rec + r * R;
            const u64 a = p[0], b = p[1], c = (R == 3) ? p[2] : 0ull;
            srec[tid * 3] = a; srec[tid * 3 + 1] = b; srec[tid * 3 + 2] = c;
            n = (u32)((R == 3 ? c : b) & 0xFFu);
        }
        u32 inc = n;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const u32 v = __shfl_up(inc, d); if (lane >= d) inc += v; }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        u32 off = inc - n, total = 0;
#pragma unroll
        for (int x = 0; x < SKX_NT / 64; ++x) { const u32 v = wsum[x]; if (x < wave) off += v; total += v; }
        for (u32 j = 0; j < n; ++j) smap[off + j] = (unsigned short)((tid << 5) | j);
        __syncthreads();
        for (u32 i = tid; i < total; i += SKX_NT) {
            const u32 e = smap[i];
            const u64* rr = srec + (e >> 5) * 3;
            if (W == 1) reinterpret_cast<u64*>(out)[obase + i] = sk_key1(rr, (int)(e & 31u), k);
            else reinterpret_cast<K2*>(out)[obase + i] = sk_key2(rr, (int)(e & 31u), k);
        }
        obase += total;
        __syncthreads();
    }
}

// ---------------------------------------------------------------- receiver: a positional SAMPLE of the records as a key array
// Sample chunk c = the records [cbeg[c], cbeg[c] + nr) (nr candidates, as a level-1 tile takes them); every candidate gets 16 key
// slots in out[(c * nr + i) * SK_MAXN ..], filled with its k-mers' mixed keys and, behind them (and for candidates past the end of the
// records), the all-ones sentinel -- a key array with pads, which the histogram / heavy-k-mer kernels of the key-array source read
// as it is (tile_keys_array masks the pads).  The level-1 slices of the receive side are sized from it, per bin.
template <int W>
__global__ __launch_bounds__(SKX_NT) void k_sk_sample_keys(const u64* __restrict__ rec, u64 nrec, u32 R, int k, const u64* __restrict__ cbeg, u32 nr,
                                                           typename KeyT<W>::T* __restrict__ out) {
    typedef typename KeyT<W>::T Key;
    const u64 r0 = cbeg[blockIdx.x];
    Key* o = out + (u64)blockIdx.x * nr * SK_MAXN;
    for (u32 i = threadIdx.x; i < nr; i += SKX_NT) {
        const u64 r = r0 + i;
        u64 w[3] = {0ull, 0ull, 0ull};
        u32 n = 0;
        if (r < nrec) {
            const u64* p = rec + r * R;
            w[0] = p[0]; w[1] = p[1]; if (R == 3) w[2] = p[2];
            n = (u32)(w[R - 1] & 0xFFu);
            if (n > SK_MAXN) n = SK_MAXN;
        }
        for (u32 j = 0; j < SK_MAXN; ++j) {
            Key key = empty_key<W>();
            if (j < n) sk_key(w, (int)j, k, key);
            o[(u64)i * SK_MAXN + j] = key;
        }
    }
}
