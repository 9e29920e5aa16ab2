// kmer.hpp -- host-side k-mer types with the gatb-core names the DSK sources use.
//
// Mirrors (by name and meaning, not by code -- gatb-core is absent from the
// reference tree) the surface used at:
//   utils/dsk2ascii.cpp:58   typedef typename Kmer<span>::Count Count;
//   utils/dsk2ascii.cpp:65   typename Kmer<span>::ModelCanonical model (kmerSize);
//   utils/dsk2ascii.cpp:91   model.toString (count.value)
//   src/DSK.cpp:103          Integer::apply<Functor,Parameter> (kmerSize, ...)
// Semantics: README.md:104-112 (A=0,C=1,T=2,G=3; canonical = min(fwd, revcomp)),
// spans are multiples of 32 and serve k < span (README.md:115-122, CMakeLists.txt:42).
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace dsk {

// LargeInt<N>: unsigned integer of N 64-bit words, word 0 least significant.
template <size_t N>
struct LargeInt {
    uint64_t w[N];
    LargeInt() { for (size_t i = 0; i < N; ++i) w[i] = 0; }
    LargeInt(uint64_t v) { for (size_t i = 0; i < N; ++i) w[i] = 0; w[0] = v; }
    bool operator==(const LargeInt& o) const { for (size_t i = 0; i < N; ++i) if (w[i] != o.w[i]) return false; return true; }
    bool operator!=(const LargeInt& o) const { return !(*this == o); }
    bool operator<(const LargeInt& o) const {
        for (size_t i = N; i-- > 0;) { if (w[i] != o.w[i]) return w[i] < o.w[i]; }
        return false;
    }
    // (this >> shift) & 3
    unsigned base_at(size_t shift) const { return (unsigned)((w[shift / 64] >> (shift % 64)) & 3u); }
    void shl2_or(unsigned c) {   // this = (this << 2) | c
        for (size_t i = N; i-- > 1;) w[i] = (w[i] << 2) | (w[i - 1] >> 62);
        w[0] = (w[0] << 2) | c;
    }
    void shr2() {
        for (size_t i = 0; i + 1 < N; ++i) w[i] = (w[i] >> 2) | (w[i + 1] << 62);
        w[N - 1] >>= 2;
    }
    void or_at(unsigned c, size_t shift) { w[shift / 64] |= (uint64_t)c << (shift % 64); }
    void mask_bits(size_t nbits) {
        for (size_t i = 0; i < N; ++i) {
            if (nbits >= 64 * (i + 1)) continue;
            if (nbits <= 64 * i) w[i] = 0;
            else w[i] &= ((uint64_t)1 << (nbits - 64 * i)) - 1;
        }
    }
};

inline int nt2code(unsigned char c) {   // -1 = not a nucleotide (breaks the window, test/readN.fasta)
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'T': case 't': return 2;
        case 'G': case 'g': return 3;
        default: return -1;
    }
}

template <size_t span>
struct Kmer {
    static_assert(span % 32 == 0 && span >= 32, "span is a multiple of 32 (README.md:117)");
    // A k-mer with k < span needs 2k < 2*span bits: span 32 -> one 64-bit word, span 64 -> two ...
    static const size_t WORDS = span / 32;
    typedef LargeInt<span / 32> Type;

    // {value, abundance}: the row type of the "solid" partition (utils/dsk2ascii.cpp:58,87,104)
    struct Count {
        static const size_t SPAN = span;      // lets generic code (Group::getPartition<Count>) find the row's HDF5 type
        Type value;
        int32_t abundance;
        Count() : abundance(0) {}
        Count(const Type& v, int32_t a) : value(v), abundance(a) {}
    };

    class ModelCanonical {
    public:
        explicit ModelCanonical(size_t kmerSize) : k_(kmerSize) {
            if (kmerSize < 1 || kmerSize > span)   // Integer::apply picks span > k; k == span still fits the words
                throw std::runtime_error("kmer size out of range for this span");
        }
        size_t getKmerSize() const { return k_; }

        // MSB-first letters, as printed by dsk2ascii (test/short.parse_results:1)
        std::string toString(const Type& v) const {
            static const char L[4] = {'A', 'C', 'T', 'G'};
            std::string s(k_, 'A');
            for (size_t i = 0; i < k_; ++i) s[i] = L[v.base_at(2 * (k_ - 1 - i))];
            return s;
        }
        Type reverse(const Type& v) const {   // reverse complement
            Type r;
            for (size_t i = 0; i < k_; ++i) r.or_at(v.base_at(2 * i) ^ 2u, 2 * (k_ - 1 - i));
            return r;
        }
        // forward value of an exactly-k-long ACGT string
        Type codeSeed(const char* seq) const {
            Type f;
            for (size_t i = 0; i < k_; ++i) {
                int c = nt2code((unsigned char)seq[i]);
                if (c < 0) throw std::runtime_error("codeSeed: non-ACGT base");
                f.shl2_or((unsigned)c);
            }
            return f;
        }
        Type canonical(const Type& fwd) const { Type r = reverse(fwd); return r < fwd ? r : fwd; }

        // Call fct(canonical, position_of_last_base) for every valid window of the sequence.
        template <class F>
        void iterate(const char* seq, size_t len, F fct) const {
            Type fwd, rc;
            size_t run = 0;
            for (size_t i = 0; i < len; ++i) {
                int c = nt2code((unsigned char)seq[i]);
                if (c < 0) { run = 0; fwd = Type(); rc = Type(); continue; }
                fwd.shl2_or((unsigned)c); fwd.mask_bits(2 * k_);
                rc.shr2(); rc.or_at((unsigned)c ^ 2u, 2 * (k_ - 1));
                if (++run >= k_) fct(rc < fwd ? rc : fwd, i);
            }
        }
    private:
        size_t k_;
    };
};

// Integer::apply<Functor,Parameter>(kmerSize, param): run Functor<span>()(param)
// for the smallest compiled span that holds kmerSize (src/DSK.cpp:103;
// KSIZE_LIST "32 64 96 128" as in CMakeLists.txt:42).
struct Integer {
    template <template <size_t> class Functor, class Parameter>
    static void apply(size_t kmerSize, Parameter p) {
        if (kmerSize < 32) Functor<32>()(p);
        else if (kmerSize < 64) Functor<64>()(p);
        else if (kmerSize < 96) Functor<96>()(p);
        else if (kmerSize < 128) Functor<128>()(p);
        else throw std::runtime_error("kmer size too large for the compiled spans (KSIZE_LIST=32 64 96 128): k must be < 128");
    }
    // Same dispatch for C++17 callers: fct(std::integral_constant<size_t, span>()).
    template <class F>
    static void dispatch(size_t kmerSize, F&& fct) {
        if (kmerSize < 32) fct(std::integral_constant<size_t, 32>());
        else if (kmerSize < 64) fct(std::integral_constant<size_t, 64>());
        else if (kmerSize < 96) fct(std::integral_constant<size_t, 96>());
        else if (kmerSize < 128) fct(std::integral_constant<size_t, 128>());
        else throw std::runtime_error("kmer size too large for the compiled spans (KSIZE_LIST=32 64 96 128): k must be < 128");
    }
};

}  // namespace dsk
