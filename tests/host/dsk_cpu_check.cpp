// dsk_cpu_check.cpp -- TEST INFRASTRUCTURE (lives under tests/, never shipped).
// The dsk tool (dsk_amd/host) with the CPU ORACLE plugged in as the counting
// backend, so that the host plumbing (bank parsing, option handling, HDF5
// layout, dsk2ascii) can be checked against the reference's golden files on a
// machine without a GPU.  The product `dsk` binary registers the GPU engine
// only and has no such fallback.
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <map>

#include "../../dsk_amd/host/dsk.hpp"
#include "../../oracle/dsk_oracle.h"

namespace {
class OracleBackend : public dsk::ICountBackend {
public:
    ~OracleBackend() override { if (r_) dsko_free(r_); }
    std::string name() const override { return "cpu-oracle (test only)"; }
    void configure(const dsk::CountConfig& c) override { cfg_ = c; }
    void push(const char* d, size_t n) override {
        // test hook: behave like an engine that runs out of memory after so many bytes (error path of the bank threads)
        if (const char* e = getenv("DSK_TEST_FAIL_AFTER_BYTES")) if (stream_.size() + n > (size_t)atoll(e)) throw dsk::Exception("test backend: out of memory after %zu bytes", stream_.size());
        stream_.insert(stream_.end(), d, d + n); stream_.push_back('\n');
    }
    void markBank() override { mark_ = stream_.size(); }
    bool rewindBank() override { stream_.resize(mark_); return true; }
    void nextBank() override { ends_.push_back(stream_.size()); }          // (an empty bank is a bank)
    void finish() override {
        words_ = (cfg_.kmer_size + 31) / 32;
        const bool banked = ends_.size() > 1 && (cfg_.solidity_kind != 0 || cfg_.histo2d);
        if (!banked) {
            r_ = dsko_count(reinterpret_cast<const uint8_t*>(stream_.data()), stream_.size(), (int)cfg_.kmer_size, 4);
            if (!r_) throw dsk::Exception("oracle: bad kmer size");
            uint64_t d = dsko_num_distinct(r_);
            std::vector<uint64_t> w[4]; for (auto& v : w) v.resize(d + 1);
            std::vector<uint32_t> ab(d + 1);
            dsko_rows4(r_, w[0].data(), w[1].data(), w[2].data(), w[3].data(), ab.data());
            total_ = dsko_total_kmers(r_); distinct_ = d;
            hist_.assign(cfg_.histo_max + 1, 0); dsko_histogram(r_, hist_.data(), cfg_.histo_max);
            for (uint64_t i = 0; i < d; ++i) if (ab[i] >= cfg_.abundance_min && ab[i] <= cfg_.abundance_max) {
                for (size_t x = 0; x < words_; ++x) k_.push_back(w[x][i]);
                a_.push_back(ab[i]);
            }
            return;
        }
        // several banks: per-bank oracle counts merged on the k-mer (restatement of include/dskgpu.h DSKGPU_SOLIDITY_*)
        typedef std::array<uint64_t, 4> K;                // most significant word first: map order = k-mer order
        std::map<K, std::vector<uint32_t>> m;
        const size_t B = ends_.size();
        for (size_t b = 0; b < B; ++b) {
            const size_t beg = b ? ends_[b - 1] : 0;
            dsko_result* r = dsko_count(reinterpret_cast<const uint8_t*>(stream_.data()) + beg, ends_[b] - beg, (int)cfg_.kmer_size, 4);
            if (!r) throw dsk::Exception("oracle: bad kmer size");
            uint64_t d = dsko_num_distinct(r); total_ += dsko_total_kmers(r);
            std::vector<uint64_t> w[4]; for (auto& v : w) v.resize(d + 1);
            std::vector<uint32_t> ab(d + 1);
            dsko_rows4(r, w[0].data(), w[1].data(), w[2].data(), w[3].data(), ab.data());
            for (uint64_t i = 0; i < d; ++i) { auto& v = m[K{w[3][i], w[2][i], w[1][i], w[0][i]}]; v.resize(B, 0); v[b] = ab[i]; }
            dsko_free(r);
        }
        distinct_ = m.size();
        hist_.assign(cfg_.histo_max + 1, 0); h2_.assign((size_t)(cfg_.histo_max + 1) * 11, 0);
        for (auto& kv : m) {
            const std::vector<uint32_t>& c = kv.second;
            uint64_t sum = 0; uint32_t mn = 0xFFFFFFFFu, mx = 0; bool one = false, all = true, custom = true;
            for (size_t b = 0; b < B; ++b) {
                sum += c[b]; mn = std::min(mn, c[b]); mx = std::max(mx, c[b]);
                const bool in = c[b] >= cfg_.abundance_min && c[b] <= cfg_.abundance_max;
                one |= in; all &= in;
                if ((cfg_.solidity_custom >> b) & 1) custom &= c[b] >= cfg_.abundance_min; else custom &= c[b] == 0;
            }
            hist_[std::min<uint64_t>(sum, cfg_.histo_max)]++;
            h2_[std::min<uint64_t>(sum - c[0], cfg_.histo_max) * 11 + std::min<uint32_t>(c[0], 10)]++;
            bool solid;
            switch (cfg_.solidity_kind) {
                case 1: solid = mn >= cfg_.abundance_min && mn <= cfg_.abundance_max; break;
                case 2: solid = mx >= cfg_.abundance_min && mx <= cfg_.abundance_max; break;
                case 3: solid = one; break;
                case 4: solid = all; break;
                case 5: solid = custom; break;
                default: solid = sum >= cfg_.abundance_min && sum <= cfg_.abundance_max;
            }
            if (solid) { for (size_t x = 0; x < words_; ++x) k_.push_back(kv.first[3 - x]); a_.push_back((uint32_t)std::min<uint64_t>(sum, 0xFFFFFFFFull)); }
        }
    }
    void histogram(std::vector<uint64_t>& h) override { h = hist_; }
    void histogram2d(std::vector<uint64_t>& h) override { h = cfg_.histo2d ? h2_ : std::vector<uint64_t>(); }
    uint32_t numPartitions() override { return cfg_.nb_partitions ? cfg_.nb_partitions : 4; }
    uint64_t partitionSize(uint32_t p) override { uint64_t n = a_.size(), P = numPartitions(); return n * (p + 1) / P - n * p / P; }
    void partitionCopy(uint32_t p, uint64_t* kmers, uint32_t* ab) override {
        uint64_t n = a_.size(), P = numPartitions(), b = n * p / P, e = n * (p + 1) / P;
        std::memcpy(kmers, k_.data() + b * words_, (e - b) * words_ * 8);
        std::memcpy(ab, a_.data() + b, (e - b) * 4);
    }
    void stats(dsk::IProperties& info, size_t d) override {
        info.add(d, "engine", name());
        info.add(d, "kmers_nb_valid", "%llu", (unsigned long long)total_);
        info.add(d, "kmers_nb_distinct", "%llu", (unsigned long long)distinct_);
        info.add(d, "kmers_nb_solid", "%llu", (unsigned long long)a_.size());
    }
private:
    size_t mark_ = 0;
    dsk::CountConfig cfg_; std::vector<char> stream_; dsko_result* r_ = nullptr;
    std::vector<uint64_t> k_; std::vector<uint32_t> a_; size_t words_ = 1;
    std::vector<size_t> ends_; std::vector<uint64_t> hist_, h2_; uint64_t total_ = 0, distinct_ = 0;
};
dsk::ICountBackend* make() { return new OracleBackend(); }
}  // namespace

int main(int argc, char* argv[]) {
    dsk::setBackendFactory(make);
    try { dsk::DSK().run(argc, argv); }
    catch (dsk::OptionFailure& e) { return e.displayErrors(std::cout); }
    catch (dsk::Exception& e) { std::cerr << "EXCEPTION: " << e.getMessage() << std::endl; return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}
