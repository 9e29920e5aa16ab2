// gpu_backend.cpp -- the product backend: the C-ABI of include/dskgpu.h.
// This is the call that stands where SortingCountAlgorithm<span>::execute()
// does its work in the reference (src/DSK.cpp:60).  C error codes become
// dsk::Exception (src/main.cpp:42-46 prints "EXCEPTION: <msg>").
#include "../../include/dskgpu.h"
#include "count_backend.hpp"

namespace dsk {

namespace {
class GpuBackend : public ICountBackend {
public:
    GpuBackend() : ctx_(nullptr) {}
    ~GpuBackend() override { if (ctx_) dskgpu_destroy(ctx_); }
    std::string name() const override { return dskgpu_version(); }
    void configure(const CountConfig& c) override {
        if (ctx_) { dskgpu_destroy(ctx_); ctx_ = nullptr; }
        dskgpu_config g{};
        g.kmer_size = c.kmer_size; g.abundance_min = c.abundance_min; g.abundance_max = c.abundance_max;
        g.histo_max = c.histo_max; g.device = c.device; g.nb_partitions = c.nb_partitions;
        g.flags = DSKGPU_F_TIMING | (c.histo2d ? DSKGPU_F_HISTO2D : 0u); g.world_size = 1; g.rank = 0;
        g.solidity_kind = c.solidity_kind; g.solidity_custom = c.solidity_custom;
        cfg_ = c;
        int rc = dskgpu_create(&g, &ctx_);
        if (rc != DSKGPU_OK) { std::string m = dskgpu_last_error(nullptr); ctx_ = nullptr; throw Exception("GPU engine: %s (code %d)", m.c_str(), rc); }
    }
    void reserve(uint64_t n) override { ck(dskgpu_reserve_reads(ctx_, n)); }
    void push(const char* data, size_t n) override { ck(dskgpu_push_reads(ctx_, data, n)); }
    void nextBank() override { ck(dskgpu_next_bank(ctx_)); }
    void finish() override { ck(dskgpu_count(ctx_)); }
    void histogram(std::vector<uint64_t>& h) override { h.assign(cfg_.histo_max + 1, 0); ck(dskgpu_histogram(ctx_, h.data(), cfg_.histo_max + 1)); }
    void histogram2d(std::vector<uint64_t>& h) override {
        h.clear();
        if (!cfg_.histo2d) return;
        h.assign((size_t)(cfg_.histo_max + 1) * 11, 0);
        if (dskgpu_histogram2d(ctx_, h.data(), cfg_.histo_max + 1) != DSKGPU_OK) h.clear();   // single bank: nothing to cross
    }
    uint32_t numPartitions() override { return dskgpu_num_partitions(ctx_); }
    uint64_t partitionSize(uint32_t p) override { return dskgpu_partition_size(ctx_, p); }
    void partitionCopy(uint32_t p, uint64_t* kmers, uint32_t* ab) override { ck(dskgpu_partition_copy(ctx_, p, kmers, ab)); }
    void stats(IProperties& info, size_t d) override {
        dskgpu_stats s{};
        if (dskgpu_get_stats(ctx_, &s) != DSKGPU_OK) return;
        info.add(d, "engine", "%s", dskgpu_version());
        info.add(d, "bytes_read_stream", "%llu", (unsigned long long)s.n_bytes);
        info.add(d, "kmers_nb_valid", "%llu", (unsigned long long)s.n_kmers);
        info.add(d, "kmers_nb_distinct", "%llu", (unsigned long long)s.n_distinct);
        info.add(d, "kmers_nb_solid", "%llu", (unsigned long long)s.n_solid);
        info.add(d, "nb_partitions", "%u", s.n_partitions);
        info.add(d, "partition_levels", "%u", s.n_levels);
        info.add(d, "hash_sub_partitions", "%u", s.n_final_bins);
        info.add(d, "overflow_retries", "%u", s.n_retries);
        const char* names[64]; float ms[64];
        int n = dskgpu_stage_times(ctx_, names, ms, 64);
        if (n > 0) {
            info.add(d, "gpu_stage_ms");
            for (int i = 0; i < n && i < 64; ++i) info.add(d + 1, names[i], "%.3f", ms[i]);
        }
    }
private:
    void ck(int rc) { if (rc != DSKGPU_OK) throw Exception("GPU engine: %s (code %d)", dskgpu_last_error(ctx_), rc); }
    dskgpu_ctx* ctx_; CountConfig cfg_;
};
}  // namespace

ICountBackend* createGpuBackend() { return new GpuBackend(); }

}  // namespace dsk
