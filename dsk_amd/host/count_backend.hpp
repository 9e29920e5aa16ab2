// count_backend.hpp -- seam between the host layer and the counting engine.
// The product backend is GpuBackend (gpu_backend.cpp -> C-ABI of
// include/dskgpu.h -> HIP kernels).  The interface exists so that the host
// plumbing (bank, options, HDF5) can be exercised on a machine without a GPU by
// tests that plug a checker backend; no product binary links anything else.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "tool.hpp"

namespace dsk {

struct CountConfig {
    unsigned kmer_size = 31;
    unsigned abundance_min = 2;
    unsigned abundance_max = 2147483647u;
    unsigned histo_max = 10000;
    unsigned nb_partitions = 0;   // 0 = engine default
    int device = 0;
    unsigned solidity_kind = 0;   // 0 sum, 1 min, 2 max, 3 one, 4 all, 5 custom (include/dskgpu.h DSKGPU_SOLIDITY_*)
    unsigned solidity_custom = 0; // custom: bit b = bank b must hold the k-mer
    bool histo2d = false;         // also build the 2-D histogram (bank 0 = genome, others = reads)
    unsigned nb_gpus = 1;         // -nb-gpus: ranks of the sharded count inside this process (k-mer space split by minimizer owner)
    unsigned minimizer_size = 0;  // -minimizer-size (owner map of the multi-GPU exchange); 0 = engine default
};

class ICountBackend {
public:
    virtual ~ICountBackend() {}
    virtual std::string name() const = 0;
    virtual void configure(const CountConfig& cfg) = 0;
    virtual void reserve(uint64_t nbytes) {}                  // optional hint: total read-stream bytes to come
    virtual void prepare(uint64_t nbytes) {}                  // optional: set up the work buffers of a count over that many bytes now (may run
                                                              // on another thread while push() is called; done before finish())
    virtual void push(const char* data, size_t nbytes) = 0;   // read-stream chunk, whole records
    virtual bool parsesOnDevice() const { return false; }     // pushRaw / rawFinish are offered (with the configuration given)
    // optional: file TEXT (FASTA 1 / FASTQ 2, cut anywhere) for an engine that parses on the device; false = not offered, use push()
    virtual bool pushRaw(const char* text, size_t nbytes, int format, bool new_file) { (void)text; (void)nbytes; (void)format; (void)new_file; return false; }
    // after the raw pushes of a bank: true + the number of records; false = the engine gave the text back (nothing of it was kept: parse on the host and push())
    virtual bool rawFinish(uint64_t& records, uint64_t& stream_bytes) { records = 0; stream_bytes = 0; return false; }
    // markBank: remember where the read stream stands; rewindBank: forget what was pushed since (the bank parsed a damaged file in
    // parallel and a serial parse would differ: it is pushed again) -- false when the backend cannot
    virtual void markBank() {}
    virtual bool rewindBank() { return false; }
    virtual void nextBank() = 0;                              // what was pushed so far is one bank (comma-separated input)
    virtual void finish() = 0;                                // run the count; results valid afterwards
    virtual void histogram(std::vector<uint64_t>& h) = 0;     // histo_max + 1 entries, h[0] == 0
    virtual void histogram2d(std::vector<uint64_t>& h) = 0;   // (histo_max + 1) x 11, row-major; empty when not requested
    virtual uint32_t numPartitions() = 0;
    virtual uint64_t partitionSize(uint32_t p) = 0;
    // kmers: n * words u64 (least significant word first); abundance: n u32; ascending k-mer order
    virtual void partitionCopy(uint32_t p, uint64_t* kmers, uint32_t* abundance) = 0;
    virtual void stats(IProperties& info, size_t depth) = 0;
};

typedef ICountBackend* (*BackendFactory)();
void setBackendFactory(BackendFactory f);     // set once by the executable's main()
ICountBackend* createBackend();               // throws dsk::Exception when none was set

ICountBackend* createGpuBackend();            // gpu_backend.cpp (links libdskgpu.so)

// The `dsk` executable leaves with _exit() right after the run and says so here.  (What a backend may do with that is its business; the GPU
// backend still gives its buffers back -- see ~GpuBackend.)
void setProcessExitsAfterRun(bool yes);
bool processExitsAfterRun();

}  // namespace dsk
