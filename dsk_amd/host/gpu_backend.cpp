// gpu_backend.cpp -- the product backend: the C-ABI of include/dskgpu.h.
// This is the call that stands where SortingCountAlgorithm<span>::execute()
// does its work in the reference (src/DSK.cpp:60).  C error codes become
// dsk::Exception (src/main.cpp:42-46 prints "EXCEPTION: <msg>").
#include "../../include/dskgpu.h"
#include <sys/time.h>
#include <algorithm>
#include <thread>
#include <functional>
#include <cstdio>
#include <cstdlib>

#include "count_backend.hpp"

namespace dsk {

namespace {
// One GPU: a dskgpu_ctx.  -nb-gpus N: a dskgpu_group of N ranks inside this process (include/dskgpu.h) -- every push is
// cut into N pieces at record borders (ingest data-parallel), the engine exchanges super-k-mer records between the ranks
// (RCCL over xGMI) and every rank counts the k-mers it owns; the storage still gets ONE flat list of partitions
// (global id = local id * N + rank) and one histogram (the sum), as after the reference's single execute() call.
class GpuBackend : public ICountBackend {
public:
    GpuBackend() : ctx_(nullptr), grp_(nullptr) {}
    // (r05: leaving WITHOUT giving the buffers back -- the process ends with _exit a moment later -- was measured and dropped: the teardown is
    //  0.013-0.034 s, and a process that exits holding 45 GB of HBM leaves the driver to reclaim them while the NEXT dsk run starts: its
    //  engine start-up went from 0.15-0.3 to 1.2 s in five of six back-to-back runs.  DSK_NO_TEARDOWN=1 keeps the experiment reachable.)
    ~GpuBackend() override {
        if (processExitsAfterRun() && getenv("DSK_NO_TEARDOWN")) { ctx_ = nullptr; grp_ = nullptr; return; }
        drop();
    }
    std::string name() const override { return dskgpu_version(); }
    void configure(const CountConfig& c) override {
        drop();
        dskgpu_config g{};
        g.kmer_size = c.kmer_size; g.abundance_min = c.abundance_min; g.abundance_max = c.abundance_max;
        g.histo_max = c.histo_max; g.device = c.device; g.nb_partitions = c.nb_partitions;
        g.flags = DSKGPU_F_TIMING | (c.histo2d ? DSKGPU_F_HISTO2D : 0u); g.world_size = 1; g.rank = 0;
        g.solidity_kind = c.solidity_kind; g.solidity_custom = c.solidity_custom; g.minimizer_size = c.minimizer_size;
        cfg_ = c;
        if (c.nb_gpus > 1) {
            const int ndev = dskgpu_device_count();
            if (ndev < 1) throw Exception("GPU engine: no HIP device");
            std::vector<int32_t> devs(c.nb_gpus);
            for (unsigned r = 0; r < c.nb_gpus; ++r) devs[r] = (int32_t)((c.device + (int)r) % ndev);   // fewer devices than ranks: ranks share
            int rc = dskgpu_group_create(&g, devs.data(), c.nb_gpus, &grp_);
            if (rc != DSKGPU_OK) { std::string m = dskgpu_group_last_error(nullptr); grp_ = nullptr; throw Exception("GPU engine: %s (code %d)", m.c_str(), rc); }
            pushed_.assign(c.nb_gpus, 0);
            return;
        }
        // (`dsk -device N` on a node with several GPUs: main() made only that device visible before any thread started, so it is
        //  ordinal 0 here; an embedding host keeps its own device numbering and nothing is remapped)
        if (const char* m = getenv("DSK_DEVICE_REMAPPED")) { if (atoi(m) == c.device) g.device = 0; }
        int rc = dskgpu_create(&g, &ctx_);
        if (rc != DSKGPU_OK) { std::string m = dskgpu_last_error(nullptr); ctx_ = nullptr; throw Exception("GPU engine: %s (code %d)", m.c_str(), rc); }
    }
    void reserve(uint64_t n) override {
        if (!grp_) { ck(dskgpu_reserve_reads(ctx_, n)); return; }
        const uint32_t N = dskgpu_group_size(grp_);
        for (uint32_t r = 0; r < N; ++r) ckr(r, dskgpu_reserve_reads(dskgpu_group_ctx(grp_, r), n / N + n / (8 * N) + 4096));
    }
    void prepare(uint64_t n) override {
        // optional by contract: a reservation that does not fit (its sizes are upper bounds from file-size hints) is not an error --
        // dskgpu_count sizes its buffers from the real k-mer count, in several passes if need be
        auto soft = [&](int rc) { return rc == DSKGPU_NOT_RESERVED ? DSKGPU_OK : rc; };      // (a genuine out-of-memory error of an allocation stays an error)
        if (!grp_) { ck(soft(dskgpu_reserve_work(ctx_, n))); return; }
        const uint32_t N = dskgpu_group_size(grp_);
        for (uint32_t r = 0; r < N; ++r) ckr(r, soft(dskgpu_reserve_work(dskgpu_group_ctx(grp_, r), n / N + n / (8 * N) + 4096)));
    }
    void push(const char* data, size_t n) override {
        if (mark_pending_) take_mark();
        if (!grp_) { ck(dskgpu_push_reads(ctx_, data, n)); return; }
        // N pieces cut at record separators; the rank that has received least so far gets the first (largest) one
        const uint32_t N = dskgpu_group_size(grp_);
        uint32_t r = 0;
        for (uint32_t i = 1; i < N; ++i) if (pushed_[i] < pushed_[r]) r = i;
        // one feeder thread per rank: the pieces go to their GPUs at the same time (every rank has its own pinned staging
        // buffers, stream and PCIe link); fed in turn from one thread, N GPUs took N times one GPU's push
        struct Piece { uint32_t rank; size_t beg, end; int rc; };
        std::vector<Piece> pieces;
        size_t beg = 0;
        for (uint32_t i = 0; i < N && beg < n; ++i, r = (r + 1) % N) {
            size_t end = i + 1 == N ? n : std::max(beg, n * (i + 1) / N);
            while (end < n && (end == 0 || data[end - 1] != '\n')) ++end;          // a piece ends after a separator: k-mers never span pieces
            if (end == beg) continue;
            pieces.push_back({r, beg, end, DSKGPU_OK});
            pushed_[r] += end - beg;
            beg = end;
        }
        std::vector<std::thread> th;
        auto feed = [&](Piece& pc) { pc.rc = dskgpu_push_reads(dskgpu_group_ctx(grp_, pc.rank), data + pc.beg, pc.end - pc.beg); };
        for (size_t i = 1; i < pieces.size(); ++i) th.emplace_back(feed, std::ref(pieces[i]));
        if (!pieces.empty()) feed(pieces[0]);
        for (auto& t : th) t.join();
        for (Piece& pc : pieces) ckr(pc.rank, pc.rc);
    }
    // one GPU: the file's text goes to the device as it is (dskgpu_push_raw); a group cuts its pushes at record separators, which
    // only parsed reads have
    // (the mark is taken at the first push behind it: the engine may still be starting up when the bank begins to parse)
    void markBank() override { mark_pending_ = true; }
    void take_mark() {
        mark_pending_ = false;
        if (!grp_) { ck(dskgpu_stream_bytes(ctx_, &mark_)); return; }
        const uint32_t N = dskgpu_group_size(grp_);
        marks_.assign(N, 0); pushed_mark_ = pushed_;
        for (uint32_t r = 0; r < N; ++r) ckr(r, dskgpu_stream_bytes(dskgpu_group_ctx(grp_, r), &marks_[r]));
    }
    bool rewindBank() override {
        if (mark_pending_) { mark_pending_ = false; return true; }      // nothing was pushed since the mark
        if (!grp_) { ck(dskgpu_rewind_reads(ctx_, mark_)); return true; }
        for (uint32_t r = 0; r < dskgpu_group_size(grp_); ++r) ckr(r, dskgpu_rewind_reads(dskgpu_group_ctx(grp_, r), marks_[r]));
        pushed_ = pushed_mark_;
        return true;
    }
    bool parsesOnDevice() const override { return true; }       // (one GPU; the caller does not ask with -nb-gpus > 1)
    bool pushRaw(const char* text, size_t n, int format, bool new_file) override {
        if (grp_) return false;
        if (mark_pending_) take_mark();
        ck(dskgpu_push_raw(ctx_, text, n, format, new_file ? 1 : 0));
        return true;
    }
    bool rawFinish(uint64_t& records, uint64_t& stream_bytes) override {
        records = 0; stream_bytes = 0;
        if (grp_) return false;
        const int rc = dskgpu_raw_finish(ctx_, &stream_bytes, &records);
        if (rc == DSKGPU_E_FORMAT) return false;
        ck(rc);
        return true;
    }
    void nextBank() override {
        if (!grp_) { ck(dskgpu_next_bank(ctx_)); return; }
        for (uint32_t r = 0; r < dskgpu_group_size(grp_); ++r) ckr(r, dskgpu_next_bank(dskgpu_group_ctx(grp_, r)));      // every rank's share of the bank ends here
    }
    void finish() override {
        if (!grp_) { ck(dskgpu_count(ctx_)); return; }
        const int rc = dskgpu_group_count(grp_);
        if (rc != DSKGPU_OK) throw Exception("GPU engine: %s (code %d)", dskgpu_group_last_error(grp_), rc);
    }
    void histogram(std::vector<uint64_t>& h) override {
        h.assign(cfg_.histo_max + 1, 0);
        if (grp_) { const int rc = dskgpu_group_histogram(grp_, h.data(), cfg_.histo_max + 1); if (rc) throw Exception("GPU engine: histogram (code %d)", rc); }
        else ck(dskgpu_histogram(ctx_, h.data(), cfg_.histo_max + 1));
    }
    void histogram2d(std::vector<uint64_t>& h) override {
        h.clear();
        if (!cfg_.histo2d) return;
        h.assign((size_t)(cfg_.histo_max + 1) * 11, 0);
        const int rc = grp_ ? dskgpu_group_histogram2d(grp_, h.data(), cfg_.histo_max + 1) : dskgpu_histogram2d(ctx_, h.data(), cfg_.histo_max + 1);
        if (rc != DSKGPU_OK) h.clear();   // single bank: nothing to cross
    }
    uint32_t numPartitions() override { return grp_ ? dskgpu_group_num_partitions(grp_) : dskgpu_num_partitions(ctx_); }
    uint64_t partitionSize(uint32_t p) override { return grp_ ? dskgpu_group_partition_size(grp_, p) : dskgpu_partition_size(ctx_, p); }
    void partitionCopy(uint32_t p, uint64_t* kmers, uint32_t* ab) override {
        if (grp_) { const int rc = dskgpu_group_partition_copy(grp_, p, kmers, ab); if (rc) throw Exception("GPU engine: partition %u (code %d)", p, rc); }
        else ck(dskgpu_partition_copy(ctx_, p, kmers, ab));
    }
    void stats(IProperties& info, size_t d) override {
        dskgpu_stats s{};
        if ((grp_ ? dskgpu_group_get_stats(grp_, &s) : dskgpu_get_stats(ctx_, &s)) != DSKGPU_OK) return;
        info.add(d, "engine", "%s", dskgpu_version());
        if (grp_) {
            info.add(d, "nb_gpus", "%u", dskgpu_group_size(grp_));
            info.add(d, "exchange_transport", "%s", dskgpu_group_transport(grp_));
            info.add(d, "exchange_bytes", "%llu", (unsigned long long)dskgpu_group_exchanged_words(grp_) * 8ull);
        }
        info.add(d, "bytes_read_stream", "%llu", (unsigned long long)s.n_bytes);
        info.add(d, "kmers_nb_valid", "%llu", (unsigned long long)s.n_kmers);
        info.add(d, "kmers_nb_distinct", "%llu", (unsigned long long)s.n_distinct);
        info.add(d, "kmers_nb_solid", "%llu", (unsigned long long)s.n_solid);
        info.add(d, "nb_partitions", "%u", s.n_partitions);
        info.add(d, "partition_levels", "%u", s.n_levels);
        info.add(d, "hash_sub_partitions", "%u", s.n_final_bins);
        info.add(d, "overflow_retries", "%u", s.n_retries);
        info.add(d, "extension_regions", "%llu", (unsigned long long)s.n_ext_regions);
        info.add(d, "heavy_kmers", "%llu", (unsigned long long)s.n_heavy);
        const char* names[64]; float ms[64];
        int n = dskgpu_stage_times(grp_ ? dskgpu_group_ctx(grp_, 0) : ctx_, names, ms, 64);
        if (n > 0) {
            info.add(d, grp_ ? "gpu_stage_ms_rank0" : "gpu_stage_ms");
            for (int i = 0; i < n && i < 64; ++i) info.add(d + 1, names[i], "%.3f", ms[i]);
        }
    }
private:
    void drop() {
        const bool trace = getenv("DSK_PHASE_TIMES") != nullptr && (ctx_ || grp_);
        struct timeval a, b; gettimeofday(&a, nullptr);
        if (ctx_) dskgpu_destroy(ctx_);
        if (grp_) dskgpu_group_destroy(grp_);
        ctx_ = nullptr; grp_ = nullptr;
        gettimeofday(&b, nullptr);
        if (trace) fprintf(stderr, "[dsk] engine teardown %.3f s\n", (b.tv_sec - a.tv_sec) + 1e-6 * (b.tv_usec - a.tv_usec));
    }
    void ck(int rc) { if (rc != DSKGPU_OK) throw Exception("GPU engine: %s (code %d)", dskgpu_last_error(ctx_), rc); }
    void ckr(uint32_t r, int rc) { if (rc != DSKGPU_OK) throw Exception("GPU engine, rank %u: %s (code %d)", r, dskgpu_last_error(dskgpu_group_ctx(grp_, r)), rc); }
    dskgpu_ctx* ctx_; dskgpu_group* grp_; CountConfig cfg_; std::vector<uint64_t> pushed_;
    uint64_t mark_ = 0; std::vector<uint64_t> marks_, pushed_mark_; bool mark_pending_ = false;      // markBank / rewindBank
};
}  // namespace

ICountBackend* createGpuBackend() { return new GpuBackend(); }

namespace { bool g_process_exits = false; }
void setProcessExitsAfterRun(bool yes) { g_process_exits = yes; }
bool processExitsAfterRun() { return g_process_exits; }

}  // namespace dsk
