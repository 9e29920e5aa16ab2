// bank.hpp -- sequence banks: FASTA / FASTQ, plain or gzip'ed, comma lists and
// file-of-files ("albums").  Stands in for gatb-core's Bank::open / BankFasta /
// BankAlbum on the count path (call site src/DSK.cpp:51; formats README.md:50-61;
// multi-line records test/longread.fasta; album fixtures test/file_index*).
//
// A bank does not hand out per-sequence objects: the count path only needs the
// "read stream" of include/dskgpu.h (sequence bytes, records separated by one
// '\n'), so banks stream chunks of that form, cut at record boundaries.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace dsk {

class IBank {
public:
    typedef std::function<void(const char* data, size_t nbytes)> Sink;
    virtual ~IBank() {}
    virtual std::string getId() const = 0;
    // total size in bytes of the underlying files (compressed size for .gz)
    virtual uint64_t getSize() const = 0;
    // Push the whole bank through `sink` in chunks of about chunkBytes, each
    // chunk a whole number of records.  Returns the number of sequences.
    // exact (optional): set to false when the bank parsed in parallel and cannot vouch that a serial parse would have handed on the
    // same records -- a damaged FASTQ file, on which a parser that reads qualities by count is not where the range cutter believes it
    // to be.  The caller then drops what it was given and calls streamSerial (the reference's parser is serial: its result is the one).
    virtual uint64_t stream(size_t chunkBytes, const Sink& sink, bool* exact = nullptr) = 0;
    virtual uint64_t streamSerial(size_t chunkBytes, const Sink& sink) { return stream(chunkBytes, sink, nullptr); }
    // The bank's TEXT as it lies in the (inflated) file, headers and quality lines included, in pieces cut anywhere -- for an engine
    // that parses on the device (dskgpu_push_raw).  format: 1 FASTA, 2 FASTQ (include/dskgpu.h DSKGPU_RAW_*); new_file: the piece
    // begins a file.  Returns false -- and has handed on NOTHING -- when the bank does not do this (a bank of several files, a BGZF file, text that
    // does not start like FASTA / FASTQ): the caller uses stream().
    typedef std::function<void(const char* text, size_t nbytes, int format, bool new_file)> RawSink;
    virtual bool streamRaw(const RawSink& sink) { (void)sink; return false; }
    // (number of sequences, total bases, longest sequence); exact, by a full pass
    virtual void estimate(uint64_t& number, uint64_t& totalSize, uint64_t& maxSize);
    // file names (flattened)
    virtual std::vector<std::string> files() const = 0;
    // the banks this one is made of (comma-separated inputs / album lines); a plain file is its own single bank
    virtual std::vector<IBank*> banks() { return std::vector<IBank*>(1, this); }
};

class Bank {
public:
    // uri: "a.fa", "a.fa,b.fq.gz", or a file whose lines are file names
    // (README.md:52-61).  Throws dsk::Exception when a file cannot be read.
    static IBank* open(const std::string& uri);
    // host threads used to parse large uncompressed files (-nb-cores; 0 = all, capped at 32)
    static void setParseThreads(unsigned n);
    static unsigned parseThreads();
};

}  // namespace dsk
