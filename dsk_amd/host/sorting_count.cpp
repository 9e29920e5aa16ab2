// sorting_count.cpp -- see sorting_count.hpp.
#include "sorting_count.hpp"

#include <sys/time.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <fstream>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace dsk {

namespace {
BackendFactory g_factory = nullptr;
double now_s() { struct timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }
}

void setBackendFactory(BackendFactory f) { g_factory = f; }
ICountBackend* createBackend() {
    if (!g_factory) throw Exception("no counting backend registered (the dsk binary registers the GPU engine)");
    return g_factory();
}

// Options of the count path (src/DSK.cpp:83; names evidenced at README.md:12,56,92,98,127,130,
// scripts/simple_test.sh:36,88, CHANGELOG.md:22; defaults per gatb-core 1.4.x).
IOptionsParser* SortingCountBase::makeOptionsParser() {
    OptionsParser* p = new OptionsParser("kmer count");
    p->push_back(new OptionOneParam(STR_URI_INPUT, "reads file", true));
    p->push_back(new OptionOneParam(STR_KMER_SIZE, "size of a kmer", false, "31"));
    p->push_back(new OptionOneParam(STR_KMER_ABUNDANCE_MIN, "min abundance threshold for solid kmers (or 'auto')", false, "2"));
    p->push_back(new OptionOneParam(STR_KMER_ABUNDANCE_MAX, "max abundance threshold for solid kmers", false, "2147483647"));
    p->push_back(new OptionOneParam("-abundance-min-threshold", "min abundance hard threshold (only used when min abundance is 'auto')", false, "2"));
    p->push_back(new OptionOneParam(STR_HISTOGRAM_MAX, "max number of values in kmers histogram", false, "10000"));
    p->push_back(new OptionOneParam("-solidity-kind", "way to consider a solid kmer with several input files (sum, min, max, one, all, custom)", false, "sum"));
    p->push_back(new OptionOneParam("-solidity-custom", "when solidity-kind is custom: one 0/1 per input file, e.g. 101 = present in files 1 and 3, absent from file 2", false, ""));
    p->push_back(new OptionOneParam(STR_MAX_MEMORY, "max memory (in MBytes); accepted, the engine sizes itself to HBM", false, "5000"));
    p->push_back(new OptionOneParam(STR_MAX_DISK, "max disk (in MBytes); accepted and ignored: partitions live in HBM", false, "0"));
    p->push_back(new OptionOneParam(STR_URI_OUTPUT, "output file for solid kmers", false, ""));
    p->push_back(new OptionOneParam(STR_URI_OUTPUT_DIR, "output directory", false, "."));
    p->push_back(new OptionOneParam(STR_URI_OUTPUT_TMP, "output directory for temporary files; accepted and ignored", false, "."));
    p->push_back(new OptionOneParam("-out-compress", "h5 compression level (0:none, 9:best)", false, "0"));
    p->push_back(new OptionOneParam("-storage-type", "storage type of kmer counts (hdf5 only)", false, "hdf5"));
    p->push_back(new OptionOneParam("-histo2D", "compute the 2D histogram (with first file = genome, remaining files = reads)", false, "0"));
    p->push_back(new OptionOneParam("-histo", "output the kmer abundance histogram as <out>.histo", false, "0"));
    p->push_back(new OptionOneParam("-minimizer-type", "minimizer type; accepted, does not change results", false, "0", false));
    p->push_back(new OptionOneParam("-minimizer-size", "size of a minimizer", false, "10", false));
    p->push_back(new OptionOneParam("-repartition-type", "minimizer repartition; accepted, does not change results", false, "0", false));
    p->push_back(new OptionOneParam("-nb-partitions", "number of output partitions under dsk/solid (0 = default)", false, "0", false));
    p->push_back(new OptionOneParam("-device", "GPU ordinal (first one with -nb-gpus)", false, "0", false));
    p->push_back(new OptionOneParam("-device-parse", "1: the files' text goes to the GPU as it is and is parsed there (one GPU; falls back to the host parser for text the device parser does not take)", false, "0", false));
    p->push_back(new OptionOneParam("-nb-gpus", "number of GPUs sharing the k-mer space (power of two; ranks share devices when the node has fewer)", false, "1", false));
    return p;
}

SortingCountBase::SortingCountBase(IBank* bank, IProperties* params, size_t words, size_t span)
    : bank_(bank), words_(words), span_(span) {
    if (params) input_.add(0, params);
}
SortingCountBase::~SortingCountBase() {}

std::string SortingCountBase::outputName(const IProperties& in, const std::vector<std::string>& files) {
    std::string out = in.has(STR_URI_OUTPUT) ? in.getStr(STR_URI_OUTPUT) : "";
    if (out.empty()) {   // basename of the first input without its extensions (test/test_ERR039477.sh:11-12)
        std::string f = files.empty() ? "output" : files[0];
        size_t s = f.find_last_of('/'); if (s != std::string::npos) f = f.substr(s + 1);
        for (const char* ext : {".gz", ".fasta", ".fastq", ".fa", ".fq", ".fna", ".txt"}) {
            std::string e = ext;
            if (f.size() > e.size() && f.compare(f.size() - e.size(), e.size(), e) == 0) f.resize(f.size() - e.size());
        }
        out = f;
    }
    std::string dir = in.has(STR_URI_OUTPUT_DIR) ? in.getStr(STR_URI_OUTPUT_DIR) : ".";
    if (out.find('/') == std::string::npos && dir != "." && !dir.empty()) out = dir + "/" + out;
    return out;
}

void SortingCountBase::autoCutoff(const std::vector<uint64_t>& h, unsigned& cutoff, unsigned& firstPeak) {
    cutoff = 0; firstPeak = 0;
    size_t n = h.size();
    if (n < 4) return;
    size_t i = 1;
    while (i + 1 < n && h[i + 1] < h[i]) ++i;           // descend the error tail
    size_t valley = i;
    size_t peak = valley;
    for (size_t j = valley; j < n; ++j) if (h[j] > h[peak]) peak = j;
    if (peak == valley || h[peak] == 0) return;          // no genomic peak: leave cutoff 0
    cutoff = (unsigned)valley; firstPeak = (unsigned)peak;
}

void SortingCountBase::execute() {
    const double t0 = now_s();
    const size_t k = (size_t)input_.getInt(STR_KMER_SIZE);
    if (k < 1 || k > span_) throw Exception("bad kmer size %zu", k);
    const std::string aminStr = input_.has(STR_KMER_ABUNDANCE_MIN) ? input_.getStr(STR_KMER_ABUNDANCE_MIN) : "2";
    const bool autoMin = (aminStr == "auto");
    const unsigned thresh = input_.has("-abundance-min-threshold") ? (unsigned)input_.getInt("-abundance-min-threshold") : 2u;
    CountConfig cfg;
    cfg.kmer_size = (unsigned)k;
    cfg.abundance_min = autoMin ? std::max(1u, thresh) : (unsigned)std::max<long long>(0, atoll(aminStr.c_str()));
    cfg.abundance_max = input_.has(STR_KMER_ABUNDANCE_MAX) ? (unsigned)std::min<long long>(input_.getInt(STR_KMER_ABUNDANCE_MAX), 4294967295LL) : 2147483647u;
    cfg.histo_max = input_.has(STR_HISTOGRAM_MAX) ? (unsigned)input_.getInt(STR_HISTOGRAM_MAX) : 10000u;
    cfg.nb_partitions = input_.has("-nb-partitions") ? (unsigned)input_.getInt("-nb-partitions") : 0u;
    cfg.device = input_.has("-device") ? (int)input_.getInt("-device") : 0;
    cfg.nb_gpus = input_.has("-nb-gpus") ? (unsigned)std::max<long long>(1, input_.getInt("-nb-gpus")) : 1u;
    cfg.minimizer_size = input_.has("-minimizer-size") ? (unsigned)std::max<long long>(0, input_.getInt("-minimizer-size")) : 0u;
    {   // -solidity-kind / -solidity-custom / -histo2D: per-bank counts (banks = the comma-separated inputs)
        const std::string kind = input_.has("-solidity-kind") ? input_.getStr("-solidity-kind") : "sum";
        static const char* names[] = {"sum", "min", "max", "one", "all", "custom"};
        int found = -1;
        for (int i = 0; i < 6; ++i) if (kind == names[i]) found = i;
        if (found < 0) throw Exception("unknown -solidity-kind '%s' (sum|min|max|one|all|custom)", kind.c_str());
        cfg.solidity_kind = (unsigned)found;
        if (found == 5) {
            const std::string m = input_.has("-solidity-custom") ? input_.getStr("-solidity-custom") : "";
            unsigned bit = 0;
            for (char ch : m) { if (ch == '1') cfg.solidity_custom |= 1u << bit; if (ch == '0' || ch == '1') ++bit; }
            if (bit == 0) throw Exception("-solidity-kind custom needs -solidity-custom <0/1 per input file, e.g. 101>");
        }
        cfg.histo2d = input_.has("-histo2D") && input_.getInt("-histo2D") != 0;
    }
    if (input_.has("-storage-type") && input_.getStr("-storage-type") != "hdf5")
        throw Exception("-storage-type '%s' is not supported (only 'hdf5')", input_.getStr("-storage-type").c_str());
    const int compress = input_.has("-out-compress") ? (int)input_.getInt("-out-compress") : 0;

    std::string firstUri = bank_->getId();
    if (firstUri.find(',') != std::string::npos) firstUri = firstUri.substr(0, firstUri.find(','));
    const std::string out = outputName(input_, {firstUri});
    storage_.reset(StorageFactory(STORAGE_HDF5).create(out, true, false));

    if (input_.has(STR_NB_CORES)) Bank::setParseThreads((unsigned)std::max<long long>(0, input_.getInt(STR_NB_CORES)));
    std::unique_ptr<ICountBackend> be(createBackend());
    uint64_t nbytes = 0;
    const double t1 = now_s();
    // The engine comes up on a helper thread while this one already parses the input (the parser threads fill their first chunks
    // without the device): device runtime start-up (0.2 s), the device read buffer sized once (plain files hold at most their size
    // in sequence bytes, gzip ~4x), then the partition buffers of the count (tens of GB of HBM: 0.2 s).  The first chunk handed to
    // push() waits for all of it.
    uint64_t hint = 0, seq_hint = 0;                     // file bytes (inflated); sequence bytes among them (FASTQ: half is quality)
    for (const std::string& f : bank_->files()) {
        const bool gz = f.size() > 3 && f.compare(f.size() - 3, 3, ".gz") == 0;
        const std::string stem = gz ? f.substr(0, f.size() - 3) : f;
        const bool fq = (stem.size() > 6 && stem.compare(stem.size() - 6, 6, ".fastq") == 0) || (stem.size() > 3 && stem.compare(stem.size() - 3, 3, ".fq") == 0);
        std::unique_ptr<IBank> one(Bank::open(f));
        uint64_t b = one->getSize() * (gz ? 4 : 1);
        if (gz) {      // a gzip member's trailer holds its inflated size (mod 2^32): exact for an ordinary one-member file below 4 GB, a lower bound with several members
            if (FILE* fp = fopen(f.c_str(), "rb")) {
                unsigned char t[4];
                if (fseek(fp, -4, SEEK_END) == 0 && fread(t, 1, 4, fp) == 4) {
                    const uint64_t isize = (uint64_t)t[0] | ((uint64_t)t[1] << 8) | ((uint64_t)t[2] << 16) | ((uint64_t)t[3] << 24);
                    if (isize > b && isize < one->getSize() * 1100) b = isize + (isize >> 6);      // (deflate does not exceed ~1030 : 1)
                }
                fclose(fp);
            }
        }
        hint += b; seq_hint += fq ? b / 2 : b;
    }
    struct Startup {
        std::mutex mu; std::condition_variable cv; bool can_push = false, done = false; std::exception_ptr err; std::thread th; double t_cfg = 0, t_res = 0, t_prep = 0;
        void wait_push() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return can_push || err; }); if (err) std::rethrow_exception(err); }
        void wait_done() { if (th.joinable()) th.join(); if (err) std::rethrow_exception(err); }
        ~Startup() { if (th.joinable()) th.join(); }
    } startup;
    startup.th = std::thread([&]() {
        try {
            const double a = now_s();
            be->configure(cfg);
            const double b = now_s();
            be->reserve(hint + 4096);
            const double c = now_s();
            be->prepare(seq_hint + 4096);      // (before the first push: concurrent with the host-to-device copies these allocations took 1.0 s instead of 0.2)
            const double d = now_s();
            { std::lock_guard<std::mutex> lk(startup.mu); startup.can_push = true; }
            startup.cv.notify_all();
            startup.t_cfg = b - a; startup.t_res = c - b; startup.t_prep = d - c;
        } catch (...) { std::lock_guard<std::mutex> lk(startup.mu); startup.err = std::current_exception(); }
        { std::lock_guard<std::mutex> lk(startup.mu); startup.done = true; }
        startup.cv.notify_all();
    });
    // (r05, measured and dropped: keeping what the parser threads produce before the engine is up in a host-side list -- so that the parse
    //  hides behind the device start-up -- costs a copy of every chunk under the sink's mutex: 0.1 s MORE on the 3 GB FASTQ file.)
    bool pushing = false;
    double t_first_push = 0, t_wait_engine = 0, t_in_push = 0;       // when the first chunk was ready (since t1), how long it waited for the engine, time inside the engine's push calls
    auto gate = [&]() { if (!pushing) { const double a = now_s(); t_first_push = a - t1; startup.wait_push(); t_wait_engine = now_s() - a; pushing = true; } };
    auto push = [&](const char* d, size_t n) { gate(); const double a = now_s(); be->push(d, n); t_in_push += now_s() - a; nbytes += n; };      // (callers serialise the sink)
    // chunk handed to the sink: a parser thread fills its own buffer of this size, so 32 threads first-touch 32 of them (64 MB chunks: 2 GB
    // of fresh pages for a 1.5 GB stream); small chunks stop the threads early while the engine starts up.  DSK_CHUNK_MB: experiments.
    size_t chunk_bytes = (size_t)8 << 20;        // (measured on the 3 GB FASTQ file, time from "engine up" to "all reads on the device": 8 MB 0.09 s, 16 MB 0.11, 32 MB 0.17, 64 MB 0.13)
    if (const char* e = getenv("DSK_CHUNK_MB")) chunk_bytes = (size_t)std::max(1, atoi(e)) << 20;
    uint64_t nseq = 0;
    std::vector<IBank*> subs = bank_->banks();   // one bank per comma-separated input (README.md:52-58)
    const bool per_bank = cfg.solidity_kind != 0 || cfg.histo2d;
    // -device-parse 1 (or DSK_DEVICE_PARSE=1): no host parser -- the text of every file is pushed as it is and the device leaves the
    // read stream (dskgpu_push_raw).  A bank the device parser gives back (DSKGPU_E_FORMAT) is parsed here as before.
    const bool dev_parse = cfg.nb_gpus == 1 && be->parsesOnDevice() && ((input_.has("-device-parse") && input_.getInt("-device-parse") != 0) || getenv("DSK_DEVICE_PARSE") != nullptr);
    unsigned raw_banks = 0; uint64_t raw_text_bytes = 0;
    // A bank that parsed in parallel says whether a serial parse would have handed on the same records (IBank::stream's `exact`: a
    // damaged FASTQ file on which the range cutter and a parser that reads qualities by count part ways).  If not, what it pushed is
    // dropped and the bank is parsed again by ONE thread: the reference's parser is serial, its reading of a damaged file is the one.
    unsigned serial_reparses = 0;
    auto stream_exact = [&](IBank* sub) -> uint64_t {
        be->markBank();
        const uint64_t before = nbytes;
        bool exact = true;
        uint64_t n = sub->stream(chunk_bytes, push, &exact);
        if (!exact && be->rewindBank()) { nbytes = before; ++serial_reparses; n = sub->streamSerial(chunk_bytes, push); }
        return n;
    };
    if (per_bank || subs.size() < 2 || dev_parse) {           // bank boundaries matter: stream the banks in order
        for (IBank* sub : subs) {
            bool raw_done = false;
            if (dev_parse) {
                const uint64_t before = nbytes;
                auto rawsink = [&](const char* d, size_t n, int fmt, bool nf) {
                    gate();
                    const double a = now_s();
                    if (!be->pushRaw(d, n, fmt, nf)) throw Exception("the engine does not parse on the device");
                    t_in_push += now_s() - a;
                    raw_text_bytes += n;
                };
                if (sub->streamRaw(rawsink)) {
                    uint64_t recs = 0, stream_bytes = 0;
                    if (be->rawFinish(recs, stream_bytes)) { nseq += recs; nbytes = stream_bytes; raw_done = true; ++raw_banks; }
                    else nbytes = before;
                }
            }
            if (!raw_done) nseq += stream_exact(sub);
            gate();
            be->nextBank();
        }
    } else {                                     // plain sum: inflate / parse the files concurrently (host thread pool)
        std::mutex mu; std::atomic<size_t> next(0); std::atomic<uint64_t> seqs(0); std::atomic<bool> all_exact(true);
        std::string err;
        auto worker = [&]() {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= subs.size()) return;
                try {
                    bool exact = true;
                    seqs += subs[i]->stream(chunk_bytes, [&](const char* d, size_t n) { std::lock_guard<std::mutex> g(mu); push(d, n); }, &exact);
                    if (!exact) all_exact = false;
                } catch (Exception& e) { std::lock_guard<std::mutex> g(mu); if (err.empty()) err = e.getMessage(); next = subs.size(); }
                catch (std::exception& e) { std::lock_guard<std::mutex> g(mu); if (err.empty()) err = e.what(); next = subs.size(); }
                catch (...) { std::lock_guard<std::mutex> g(mu); if (err.empty()) err = "unknown failure while reading the input"; next = subs.size(); }
            }
        };
        const unsigned nt = (unsigned)std::min<size_t>(subs.size(), std::max(1u, Bank::parseThreads()));
        be->markBank();
        const uint64_t before = nbytes;
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t) th.emplace_back(worker);
        for (auto& x : th) x.join();
        if (!err.empty()) throw Exception(err);
        nseq = seqs;
        if (!all_exact.load() && be->rewindBank()) {      // (the files' chunks are interleaved in the stream: all of them again, one after the other, one thread each)
            nbytes = before; ++serial_reparses; nseq = 0;
            for (IBank* sub : subs) nseq += sub->streamSerial(chunk_bytes, push);
        }
    }
    startup.wait_done();
    const double t2 = now_s();
    const bool phase_rss = getenv("DSK_PHASE_TIMES") != nullptr;
    auto rss = [&](const char* what) {       // (DSK_PHASE_TIMES: anonymous memory of the process after each phase -- what the kernel has to take apart after exit)
        if (!phase_rss) return;
        if (FILE* f = fopen("/proc/self/status", "r")) { char line[256]; while (fgets(line, sizeof line, f)) if (!strncmp(line, "RssAnon", 7)) fprintf(stderr, "[dsk] after %s: %s", what, line); fclose(f); }
    };
    rss("ingest");
    be->finish();
    const double t3 = now_s();
    rss("count");

    be->histogram(histo_);
    unsigned cutoff = 0, firstPeak = 0;
    autoCutoff(histo_, cutoff, firstPeak);
    unsigned amin = cfg.abundance_min;
    if (autoMin) amin = std::max(cfg.abundance_min, cutoff);

    // CountProcessor chain outputs: histogram ...
    {
        std::vector<HistoEntry> rows(cfg.histo_max);
        for (unsigned i = 1; i <= cfg.histo_max; ++i) { rows[i - 1].index = (uint16_t)i; rows[i - 1].abundance = histo_[i]; }
        hid_t t = H5Row<HistoEntry>::make();
        Group& hg = storage_->getGroup("histogram");
        hg.writeDataset("histogram", t, rows.data(), rows.size(), 0);
        H5Tclose(t);
        hg.setProperty("cutoff", std::to_string(cutoff));
        hg.setProperty("first_peak", std::to_string(firstPeak));
        uint64_t autoSolids = 0; for (size_t i = std::max(1u, cutoff); i < histo_.size(); ++i) autoSolids += histo_[i];
        hg.setProperty("nbsolids_auto", std::to_string(autoSolids));
    }
    // ... and the solid rows, one dataset per partition
    const uint32_t np = be->numPartitions();
    openPartitions(np);
    nb_solid_ = 0;
    // the engine returns ceil(k/32) words per k-mer; the row type of this span has span/32 (one more
    // when k is a multiple of 32, because span k serves k < span): zero-extend
    const size_t bw = (k + 31) / 32;
    // Partition p + 1 is fetched from the engine (device -> host) by a helper thread while partition p is turned into rows
    // (several threads) and written: two buffer slots, handed back and forth under one mutex.
    struct Slot { std::vector<uint64_t> k; std::vector<uint32_t> a; uint64_t n = 0; bool ready = false; };
    Slot slot[2];
    std::mutex mu; std::condition_variable cv;
    std::exception_ptr fetch_err;
    bool stop = false;
    std::thread fetcher([&]() {
        try {
            for (uint32_t p = 0; p < np; ++p) {
                Slot& sl = slot[p & 1];
                { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return !sl.ready || stop; }); if (stop) return; }
                sl.n = be->partitionSize(p);
                sl.k.resize(sl.n * bw + 1); sl.a.resize(sl.n + 1);
                if (sl.n) be->partitionCopy(p, sl.k.data(), sl.a.data());
                { std::lock_guard<std::mutex> lk(mu); sl.ready = true; }
                cv.notify_all();
            }
        } catch (...) { std::lock_guard<std::mutex> lk(mu); fetch_err = std::current_exception(); cv.notify_all(); }
    });
    struct Joiner { std::thread& t; std::mutex& mu; std::condition_variable& cv; bool& stop;
                    ~Joiner() { { std::lock_guard<std::mutex> lk(mu); stop = true; } cv.notify_all(); if (t.joinable()) t.join(); } } joiner{fetcher, mu, cv, stop};
    std::vector<uint64_t> wide;
    for (uint32_t p = 0; p < np; ++p) {
        Slot& sl = slot[p & 1];
        { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return sl.ready || fetch_err; }); if (fetch_err) std::rethrow_exception(fetch_err); }
        const uint64_t n = sl.n;
        const uint64_t* rows = sl.k.data();
        if (bw != words_) {
            wide.assign(n * words_ + 1, 0);
            for (uint64_t i = 0; i < n; ++i) for (size_t w = 0; w < bw; ++w) wide[i * words_ + w] = sl.k[i * bw + w];
            rows = wide.data();
        }
        writePartition(p, rows, sl.a.data(), n, amin, compress);
        { std::lock_guard<std::mutex> lk(mu); sl.ready = false; }
        cv.notify_all();
    }
    Group& dg = storage_->getGroup("dsk");
    dg.setProperty("kmer_size", std::to_string(k));
    const double t4 = now_s();
    rss("write");

    if (cfg.histo2d) {   // README.md:98-102, utils/plot-histo2D.R:22-30: rows = abundance in the reads, columns = in the genome (0..10)
        std::vector<uint64_t> h2;
        be->histogram2d(h2);
        if (h2.empty()) throw Exception("-histo2D needs at least two input files (genome first, then reads)");
        std::ofstream hf(out + ".histo2D");
        for (unsigned i = 0; i <= cfg.histo_max; ++i) {
            hf << i;
            for (unsigned g = 0; g < 11; ++g) hf << "\t" << h2[(size_t)i * 11 + g];
            hf << "\n";
        }
    }
    if (input_.has("-histo") && input_.getInt("-histo") != 0) {   // README.md:90-96, utils/plot-histo.R:24
        std::ofstream hf(out + ".histo");
        for (unsigned i = 1; i <= cfg.histo_max; ++i) hf << i << "\t" << histo_[i] << "\n";
    }

    config_.props_ = IProperties();
    config_.props_.add(0, "config");
    config_.props_.add(1, "kmer_size", "%zu", k);
    config_.props_.add(1, "abundance_min", "%u", amin);
    config_.props_.add(1, "abundance_max", "%u", cfg.abundance_max);
    config_.props_.add(1, "histo_max", "%u", cfg.histo_max);
    config_.props_.add(1, "storage_type", "hdf5");
    config_.props_.add(1, "nb_passes", "1");
    config_.props_.add(1, "nb_partitions", "%u", np);
    config_.props_.add(1, "partition_medium", "HBM (no disk spill)");

    info_ = IProperties();
    info_.add(0, getName());
    info_.add(1, "bank");
    info_.add(2, "uri", bank_->getId());
    info_.add(2, "nb_sequences", "%llu", (unsigned long long)nseq);
    info_.add(2, "read_stream_bytes", "%llu", (unsigned long long)nbytes);
    if (serial_reparses) info_.add(2, "banks_parsed_again_by_one_thread", "%u", serial_reparses);
    if (dev_parse) { info_.add(2, "banks_parsed_on_device", "%u", raw_banks); info_.add(2, "text_bytes_pushed", "%llu", (unsigned long long)raw_text_bytes); }
    info_.add(1, "stats");
    be->stats(info_, 2);
    info_.add(2, "solid_kmers_written", "%llu", (unsigned long long)nb_solid_);
    info_.add(2, "cutoff_auto", "%u", cutoff);
    info_.add(1, "time");
    info_.add(2, "setup_s", "%.3f", t1 - t0);
    info_.add(2, "ingest_s", "%.3f", t2 - t1);
    info_.add(3, "first_chunk_ready_s", "%.3f", t_first_push);
    info_.add(3, "waited_for_engine_s", "%.3f", t_wait_engine);
    info_.add(3, "inside_push_calls_s", "%.3f", t_in_push);
    info_.add(3, "engine_startup_s", "%.3f", startup.t_cfg);
    info_.add(3, "reserve_reads_s", "%.3f", startup.t_res);
    info_.add(3, "reserve_work_s", "%.3f", startup.t_prep);
    info_.add(2, "count_s", "%.3f", t3 - t2);
    info_.add(2, "write_s", "%.3f", t4 - t3);
    info_.add(2, "total_s", "%.3f", t4 - t0);
}

}  // namespace dsk
