// test_mixer.cpp -- unit test (test infrastructure) of the key mixers in dsk_amd/csrc/kmer_device.h, compiled for the HOST (the
// functions are __host__ __device__; no kernel, no device call): bijectivity of kmix / kmixN, and the property k_count2v3 rests on --
// the mixed top word of a two-word key is a hash of the WHOLE key.  The two folds used before failed it on related k-mers, and the
// pairs that exposed them are pinned here: (a) keys that differ in base 0 and base 32 alone (`low * odd`: a multiply only carries
// upward), (b) pairs found among the 63-mers of `small_repeats` with differences 16 bases apart in the low word (kmix(low): its first
// fold cancels them).
#include <cstdio>
#include <cstdlib>
#include <set>
#include "../../dsk_amd/csrc/kmer_device.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

static u64 rnd() { u64 x = 0; for (int i = 0; i < 5; ++i) x = (x << 15) ^ (u64)rand(); return x; }
static u64 top_of(u64 top, u64 low) { K2 x; x.w[0] = low; x.w[1] = top; kmixN(x); return x.w[1]; }

int main() {
    srand(11);
    for (int i = 0; i < 100000; ++i) {
        const u64 x = rnd();
        CHECK(kunmix(kmix(x)) == x);
        K2 a; a.w[0] = rnd(); a.w[1] = rnd() >> 2;
        K2 b = a; kmixN(b); CHECK(b.w[0] == a.w[0]); kunmixN(b); CHECK(b.w[1] == a.w[1]);
        KN<4> c; for (int q = 0; q < 4; ++q) c.w[q] = rnd();
        KN<4> d = c; kmixN(d); kunmixN(d); for (int q = 0; q < 4; ++q) CHECK(d.w[q] == c.w[q]);
    }
    // (b) the colliding pairs of the kmix(low) fold, (top, low)
    const u64 pairs[][4] = {{0x026b0e79eee7560fULL, 0xed4c9ac39e77b9d5ULL, 0x326b0e79dee7560fULL, 0xfd4c9ac38e77b9d5ULL},
                            {0x16f4568fdc4771c6ULL, 0xa0c9bd15a3f711dcULL, 0x26f4568fec4771c6ULL, 0x90c9bd1593f711dcULL},
                            {0x03d9d0f38113f4adULL, 0x561cf6743cec44fdULL, 0x33d9d0f3b113f4adULL, 0x661cf6740cec44fdULL}};
    for (const auto& p : pairs) CHECK(top_of(p[0], p[1]) != top_of(p[2], p[3]));
    // (a) and its relatives: every pair of keys that differ by substitutions at one base of the top word and one base of the low word
    // (all 31 x 32 position pairs, all 3 x 3 code changes) must get different mixed top words; so must low-word pairs 16 bases apart
    long checked = 0;
    for (int rep = 0; rep < 8; ++rep) {
        const u64 top = rnd() >> 2, low = rnd();
        const u64 t0 = top_of(top, low);
        for (int pt = 0; pt < 31; ++pt) for (int pl = 0; pl < 32; ++pl) for (u64 ct = 1; ct < 4; ++ct) for (u64 cl = 1; cl < 4; ++cl) {
            CHECK(top_of(top ^ (ct << (2 * pt)), low ^ (cl << (2 * pl))) != t0); ++checked;
        }
        for (int pl = 0; pl < 16; ++pl) for (u64 c1 = 1; c1 < 4; ++c1) for (u64 c2 = 1; c2 < 4; ++c2) for (u64 dt = 0; dt < 64; ++dt) {
            CHECK(top_of(top ^ (dt << 56), low ^ (c1 << (2 * pl)) ^ (c2 << (2 * (pl + 16)))) != t0); ++checked;
        }
    }
    // no two of a million related keys (one random key, its single- and double-substitution variants) share a mixed top word
    {
        std::set<u64> seen; const u64 top = rnd() >> 2, low = rnd(); long n = 0;
        for (int p1 = 0; p1 < 63; ++p1) for (int p2 = p1; p2 < 63; ++p2) for (u64 c1 = 1; c1 < 4; ++c1) for (u64 c2 = 1; c2 < 4; ++c2) {
            u64 t = top, l = low;
            if (p1 < 31) t ^= c1 << (2 * p1); else l ^= c1 << (2 * (p1 - 31));
            if (p2 != p1) { if (p2 < 31) t ^= c2 << (2 * p2); else l ^= c2 << (2 * (p2 - 31)); }
            else if (c2 != 1) continue;                     // (single substitutions once)
            seen.insert(top_of(t, l)); ++n;
        }
        CHECK((long)seen.size() == n);
        checked += n;
    }
    printf("%ld related pairs checked\n", checked);
    printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
