// pgzip.hpp -- parallel inflate of ONE gzip member (an ordinary `gzip reads.fastq` file) on a thread pool.
//
// The reference's normal input is a gzip'ed FASTA/FASTQ file (README.md:52-61; scripts/simple_test.sh:36: T1 and BASELINE configs[0]
// are .fasta.gz), and gatb-core's bank inflates it with zlib on one thread.  A deflate stream has no index, but it can still be
// inflated from the middle (the two-pass scheme of pugz / rapidgzip, restated here from the deflate format, RFC 1951):
//   1. cut the compressed bytes into chunks; every chunk but the first SEARCHES the first deflate block that starts inside it --
//      bit by bit: a non-final dynamic-Huffman header whose code lengths form complete prefix codes, followed by symbols that
//      decode to text -- (parallel)
//   2. every chunk inflates from its block start to the next chunk's, with the 32 KB window it cannot know filled with MARKERS
//      (16-bit symbols: < 256 a byte, 0x8000 | j = "byte j of the window before this chunk"); back-references copy markers like
//      bytes (parallel)
//   3. the windows are resolved chunk after chunk (32 KB each: the only serial step), then every chunk's symbols become bytes
//      (parallel), the CRC-32 of the chunks is combined and checked against the member's trailer together with ISIZE.
// Anything the scheme does not cover -- no dynamic block found in a chunk, a chunk that does not end on its successor's start,
// several members in one file, a CRC mismatch -- makes the call return false BEFORE any byte is handed on (slab 0) or throw
// (later slabs, where bytes have been consumed): the caller then inflates with zlib as before.  Exactness never rests on the
// heuristics: every slab is verified by the running CRC-32 at the end.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace dsk {

// Inflate the gzip file image [file, file + n) slab by slab on `nthreads` threads.  For every slab, `consume(bytes, len, last)` is
// called once, in stream order (the buffer is reused afterwards).  chunk_bytes = compressed bytes per chunk (0 = default 2 MB).
// Returns false (and has called consume for NOTHING) when the file is not a single-member gzip this scheme handles or its first
// slab does not pass; throws std::runtime_error when a later slab fails (corrupt file).
bool pgz_inflate(const uint8_t* file, size_t n, unsigned nthreads, size_t chunk_bytes,
                 const std::function<void(const char*, size_t, bool)>& consume);

}  // namespace dsk
