// dsk_cpu_check.cpp -- TEST INFRASTRUCTURE (lives under tests/, never shipped).
// The dsk tool (dsk_amd/host) with the CPU ORACLE plugged in as the counting
// backend, so that the host plumbing (bank parsing, option handling, HDF5
// layout, dsk2ascii) can be checked against the reference's golden files on a
// machine without a GPU.  The product `dsk` binary registers the GPU engine
// only and has no such fallback.
#include <algorithm>
#include <cstring>

#include "../../dsk_amd/host/dsk.hpp"
#include "../../oracle/dsk_oracle.h"

namespace {
class OracleBackend : public dsk::ICountBackend {
public:
    ~OracleBackend() override { if (r_) dsko_free(r_); }
    std::string name() const override { return "cpu-oracle (test only)"; }
    void configure(const dsk::CountConfig& c) override { cfg_ = c; }
    void push(const char* d, size_t n) override { stream_.insert(stream_.end(), d, d + n); stream_.push_back('\n'); }
    void finish() override {
        r_ = dsko_count(reinterpret_cast<const uint8_t*>(stream_.data()), stream_.size(), (int)cfg_.kmer_size, 4);
        if (!r_) throw dsk::Exception("oracle: bad kmer size");
        uint64_t d = dsko_num_distinct(r_);
        std::vector<uint64_t> lo(d + 1), hi(d + 1); std::vector<uint32_t> ab(d + 1);
        dsko_rows(r_, lo.data(), hi.data(), ab.data());
        words_ = cfg_.kmer_size <= 32 ? 1 : 2;
        for (uint64_t i = 0; i < d; ++i) if (ab[i] >= cfg_.abundance_min && ab[i] <= cfg_.abundance_max) {
            k_.push_back(lo[i]); if (words_ == 2) k_.push_back(hi[i]); a_.push_back(ab[i]);
        }
    }
    void histogram(std::vector<uint64_t>& h) override { h.assign(cfg_.histo_max + 1, 0); dsko_histogram(r_, h.data(), cfg_.histo_max); }
    uint32_t numPartitions() override { return cfg_.nb_partitions ? cfg_.nb_partitions : 4; }
    uint64_t partitionSize(uint32_t p) override { uint64_t n = a_.size(), P = numPartitions(); return n * (p + 1) / P - n * p / P; }
    void partitionCopy(uint32_t p, uint64_t* kmers, uint32_t* ab) override {
        uint64_t n = a_.size(), P = numPartitions(), b = n * p / P, e = n * (p + 1) / P;
        std::memcpy(kmers, k_.data() + b * words_, (e - b) * words_ * 8);
        std::memcpy(ab, a_.data() + b, (e - b) * 4);
    }
    void stats(dsk::IProperties& info, size_t d) override {
        info.add(d, "engine", name());
        info.add(d, "kmers_nb_valid", "%llu", (unsigned long long)dsko_total_kmers(r_));
        info.add(d, "kmers_nb_distinct", "%llu", (unsigned long long)dsko_num_distinct(r_));
        info.add(d, "kmers_nb_solid", "%llu", (unsigned long long)a_.size());
    }
private:
    dsk::CountConfig cfg_; std::vector<char> stream_; dsko_result* r_ = nullptr;
    std::vector<uint64_t> k_; std::vector<uint32_t> a_; size_t words_ = 1;
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
