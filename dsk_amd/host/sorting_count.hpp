// sorting_count.hpp -- SortingCountAlgorithm<span>: the object the DSK wrapper
// drives (src/DSK.cpp:55-68):
//     SortingCountAlgorithm<span> sortingCount (bank, props);
//     sortingCount.getInput()->add (0, STR_VERBOSE, ...);
//     sortingCount.execute();
//     sortingCount.getConfig().getProperties();  sortingCount.getInfo();
//     sortingCount.getStorage()->getGroup(sortingCount.getName()).setProperty("xml", ...);
// and `SortingCountAlgorithm<>::getOptionsParser()` (src/DSK.cpp:83).
// Here execute() streams the bank into the counting engine (GPU backend), then
// plays the CountProcessor chain's outputs into HDF5: histogram -> solidity ->
// dump (README.md:12,70-78).
#pragma once
#include <cstring>
#include <new>
#include <thread>
#include <memory>
#include <string>
#include <vector>

#include "bank.hpp"
#include "count_backend.hpp"
#include "kmer.hpp"
#include "storage.hpp"
#include "tool.hpp"

namespace dsk {

class Configuration {
public:
    IProperties getProperties() const { return props_; }
    IProperties props_;
};

// span-independent part (the engine works on `words` 64-bit words per k-mer)
class SortingCountBase {
public:
    SortingCountBase(IBank* bank, IProperties* params, size_t words, size_t span);
    virtual ~SortingCountBase();
    static IOptionsParser* makeOptionsParser();
    IProperties* getInput() { return &input_; }
    IProperties* getInfo() { return &info_; }
    const Configuration& getConfig() const { return config_; }
    Storage* getStorage() { return storage_.get(); }
    std::string getName() const { return "dsk"; }
    void execute();
    // results kept for callers that want them without re-reading the file
    const std::vector<uint64_t>& histogram() const { return histo_; }
    uint64_t nbSolid() const { return nb_solid_; }
    static std::string outputName(const IProperties& in, const std::vector<std::string>& files);
    // histogram cutoff used by "-abundance-min auto": first local minimum, then
    // the following maximum (the genomic peak); returns (cutoff, first_peak)
    static void autoCutoff(const std::vector<uint64_t>& h, unsigned& cutoff, unsigned& firstPeak);
protected:
    virtual void writePartition(size_t p, const uint64_t* kmers, const uint32_t* ab, uint64_t n, unsigned amin, int compress) = 0;
    virtual void openPartitions(size_t nb) = 0;
    IBank* bank_; size_t words_, span_;
    IProperties input_, info_;
    Configuration config_;
    std::unique_ptr<Storage> storage_;
    std::vector<uint64_t> histo_;
    uint64_t nb_solid_ = 0;
};

template <size_t span = 32>
class SortingCountAlgorithm : public SortingCountBase {
public:
    typedef typename Kmer<span>::Type Type;
    typedef typename Kmer<span>::Count Count;
    SortingCountAlgorithm(IBank* bank, IProperties* params) : SortingCountBase(bank, params, Kmer<span>::WORDS, span) {}
    static IOptionsParser* getOptionsParser() { return makeOptionsParser(); }
protected:
    void openPartitions(size_t nb) override { part_.reset(getStorage()->template solidPartition<span>(nb)); }
    void writePartition(size_t p, const uint64_t* kmers, const uint32_t* ab, uint64_t n, unsigned amin, int compress) override {
        // rows below amin exist only when -abundance-min auto raised the bar above the engine's: per slice of the partition,
        // count the keepers, then fill the row array at the slices' offsets (both passes on several threads)
        const unsigned nt = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(16, n >> 13));
        std::vector<uint64_t> keep(nt + 1, 0);
        auto slice = [&](unsigned t, uint64_t* b, uint64_t* e) { *b = n * t / nt; *e = n * (t + 1) / nt; };
        auto run = [&](auto fn) {
            std::vector<std::thread> th;
            for (unsigned t = 1; t < nt; ++t) th.emplace_back(fn, t);
            fn(0u);
            for (auto& x : th) x.join();
        };
        run([&](unsigned t) { uint64_t b, e, c = 0; slice(t, &b, &e); for (uint64_t i = b; i < e; ++i) c += ab[i] >= amin; keep[t + 1] = c; });
        for (unsigned t = 0; t < nt; ++t) keep[t + 1] += keep[t];
        const uint64_t m = keep[nt];
        // A big uncompressed partition: its rows get their place in the file first (a contiguous dataset allocated at once), and
        // every thread writes the rows it builds straight there (positional writes, 4 MB at a time) -- H5Dwrite would copy the whole
        // partition into the page cache from ONE thread (0.25 s of a 1.0 s run on 10 M reads).
        const long long off = (compress == 0 && m >= (1u << 14)) ? part_->reserve(p, m) : -1;
        if (off >= 0) {
            Storage* st = part_->storage();
            run([&](unsigned t) {
                uint64_t b, e, o = keep[t]; slice(t, &b, &e);
                const size_t BUF = (4u << 20) / sizeof(Count);
                std::unique_ptr<char[]> raw(new char[BUF * sizeof(Count)]);
                Count* rows = reinterpret_cast<Count*>(raw.get());
                size_t fill = 0;
                for (uint64_t i = b; i < e; ++i) {
                    if (ab[i] < amin) continue;
                    std::memset(static_cast<void*>(&rows[fill]), 0, sizeof(Count));           // (padding bytes too: the file is deterministic)
                    Count* c = new (&rows[fill]) Count();
                    for (size_t w = 0; w < Kmer<span>::WORDS; ++w) c->value.w[w] = kmers[i * words_ + w];
                    c->abundance = (int32_t)std::min<uint32_t>(ab[i], 0x7FFFFFFFu);
                    if (++fill == BUF) { st->rawWrite(off + (long long)(o * sizeof(Count)), rows, fill * sizeof(Count)); o += fill; fill = 0; }
                }
                if (fill) st->rawWrite(off + (long long)(o * sizeof(Count)), rows, fill * sizeof(Count));
            });
            nb_solid_ += m;
            return;
        }
        std::unique_ptr<char[]> raw(new char[(m + 1) * sizeof(Count)]);      // (raw storage: the rows are constructed by the filling threads)
        Count* rows = reinterpret_cast<Count*>(raw.get());
        run([&](unsigned t) {
            uint64_t b, e, o = keep[t]; slice(t, &b, &e);
            for (uint64_t i = b; i < e; ++i) {
                if (ab[i] < amin) continue;
                Count* c = new (&rows[o++]) Count();
                for (size_t w = 0; w < Kmer<span>::WORDS; ++w) c->value.w[w] = kmers[i * words_ + w];
                c->abundance = (int32_t)std::min<uint32_t>(ab[i], 0x7FFFFFFFu);
            }
        });
        part_->insert(p, rows, m, compress);
        nb_solid_ += m;
    }
private:
    std::unique_ptr<Partition<Count>> part_;
};

}  // namespace dsk
