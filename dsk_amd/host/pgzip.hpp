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
// What the scheme does not cover: a file whose first slab does not pass (no dynamic block found where one is searched, a chunk that
// does not end on its successor's start -- e.g. data that is not text) makes the call return false BEFORE any byte is handed on, and
// the caller inflates with zlib as before; a LATER slab that does not pass is inflated by zlib from the exact bit position and window
// reached (inflatePrime + inflateSetDictionary).  Several members (`cat a.gz b.gz`) are handled: each member's trailer is checked
// where the member ends.  Exactness never rests on the heuristics: every member is verified by its CRC-32 and size.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>
#include <vector>

namespace dsk {

// Inflate the gzip file image [file, file + n) slab by slab on `nthreads` threads.  For every slab, `consume(bytes, len, last)` is
// called once, in stream order; `headroom` writable bytes lie in front of `bytes` (the caller's cut-off last record of the previous
// slab goes there: one contiguous buffer to parse, no copy of the slab); the buffer is reused afterwards.  chunk_bytes = compressed
// bytes per chunk (0 = default 2 MB).  Members that follow the first one are inflated the same way.
// Returns false (and has called consume for NOTHING) when the file is not a gzip this scheme handles or its first slab does not
// pass; a later slab that does not pass is inflated by zlib from the exact position and window reached; throws std::runtime_error
// for a corrupt file (CRC-32 / size mismatch, truncation).
bool pgz_inflate(const uint8_t* file, size_t n, unsigned nthreads, size_t chunk_bytes, size_t headroom,
                 const std::function<void(char*, size_t, bool)>& consume);

}  // namespace dsk
