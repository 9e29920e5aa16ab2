// test_kmer.cpp -- unit test (test infrastructure) of dsk_amd/host/kmer.hpp against the CPU oracle:
// Kmer<span>::ModelCanonical {codeSeed, reverse, canonical, toString, iterate} and Integer::apply,
// the surface used at utils/dsk2ascii.cpp:58-104 and src/DSK.cpp:103.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../dsk_amd/host/kmer.hpp"
#include "../../oracle/dsk_oracle.h"

using namespace dsk;

static int failures = 0;
#define CHECK(c) do { if (!(c)) { printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++failures; } } while (0)

template <size_t span>
static void check_span(const std::string& seq, size_t k) {
    typedef typename Kmer<span>::Type Type;
    typename Kmer<span>::ModelCanonical model(k);
    const size_t n = seq.size();
    std::vector<uint64_t> lo(n), hi(n), w4(4 * n); std::vector<uint8_t> valid(n);
    if (k <= 64) dsko_enumerate(reinterpret_cast<const uint8_t*>(seq.data()), n, (int)k, lo.data(), hi.data(), valid.data());
    else if (dsko_enumerate4(reinterpret_cast<const uint8_t*>(seq.data()), n, (int)k, w4.data(), valid.data()) != 0) { printf("oracle built without 256-bit keys\n"); ++failures; return; }
    std::vector<uint8_t> seen(n, 0);
    size_t count = 0;
    model.iterate(seq.data(), n, [&](const Type& canon, size_t pos) {
        seen[pos] = 1; ++count;
        CHECK(valid[pos]);
        if (k <= 64) {
            CHECK(canon.w[0] == lo[pos]);
            if (Kmer<span>::WORDS > 1) CHECK(canon.w[Kmer<span>::WORDS > 1 ? 1 : 0] == hi[pos]);
            for (size_t x = 2; x < Kmer<span>::WORDS; ++x) CHECK(canon.w[x] == 0);
            char buf[80]; dsko_kmer_to_string(lo[pos], hi[pos], (int)k, buf);
            CHECK(model.toString(canon) == std::string(buf));
        } else {
            for (size_t x = 0; x < Kmer<span>::WORDS; ++x) CHECK(canon.w[x] == w4[4 * pos + x]);
            CHECK(model.toString(canon).size() == k);
        }
        // forward window -> codeSeed -> canonical must agree; reverse is an involution
        Type fwd = model.codeSeed(seq.data() + pos + 1 - k);
        CHECK(model.canonical(fwd) == canon);
        CHECK(model.reverse(model.reverse(fwd)) == fwd);
        CHECK(!(model.reverse(canon) < canon));
    });
    for (size_t i = 0; i < n; ++i) CHECK(seen[i] == valid[i]);
    printf("span %zu k %zu: %zu windows ok\n", span, k, count);
}

template <size_t span> struct PickSpan { void operator()(size_t* out) { *out = span; } };

int main() {
    srand(7);
    std::string seq;
    for (int i = 0; i < 20000; ++i) seq.push_back(rand() % 100 < 98 ? "ACGTacgt"[rand() % 8] : "NR\n"[rand() % 3]);   // ~2 % window breakers
    for (size_t k : {1, 2, 15, 16, 27, 31}) check_span<32>(seq, k);
    for (size_t k : {32, 33, 47, 63}) check_span<64>(seq, k);
    for (size_t k : {64, 65, 95}) check_span<96>(seq, k);
    for (size_t k : {96, 97, 127}) check_span<128>(seq, k);
    // README.md:111-112: GTA / TAC -> TAC
    Kmer<32>::ModelCanonical m3(3);
    CHECK(m3.toString(m3.canonical(m3.codeSeed("GTA"))) == "TAC");
    // test/short.parse_results:1
    Kmer<32>::ModelCanonical m15(15);
    CHECK(m15.toString(m15.canonical(m15.codeSeed("ACTGTACGTATAAGA"))) == "ACTGTACGTATAAGA");
    // Integer::apply picks the smallest span with k < span; k >= 128 is refused
    size_t s = 0;
    Integer::apply<PickSpan, size_t*>(31, &s); CHECK(s == 32);
    Integer::apply<PickSpan, size_t*>(32, &s); CHECK(s == 64);
    Integer::apply<PickSpan, size_t*>(63, &s); CHECK(s == 64);
    bool threw = false;
    Integer::apply<PickSpan, size_t*>(64, &s); CHECK(s == 96);
    Integer::apply<PickSpan, size_t*>(95, &s); CHECK(s == 96);
    Integer::apply<PickSpan, size_t*>(96, &s); CHECK(s == 128);
    Integer::apply<PickSpan, size_t*>(127, &s); CHECK(s == 128);
    try { Integer::apply<PickSpan, size_t*>(128, &s); } catch (std::runtime_error&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { m3.codeSeed("GNA"); } catch (std::runtime_error&) { threw = true; }
    CHECK(threw);
    printf(failures ? "FAILED (%d)\n" : "ALL OK\n", failures);
    return failures ? 1 : 0;
}
