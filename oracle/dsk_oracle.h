/*
 * dsk_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the DSK count path: reads -> canonical k-mers ->
 * (kmer, abundance) rows + abundance histogram.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (dsk_amd/, include/dskgpu.h) never links or calls it.
 *
 * Parity status: PINNED for counts + histogram by the reference's own golden
 * files (test/k27.histo, test/rlong.histo, test/readN.histo,
 * test/short.parse_results; scripts/simple_test.sh:35-135), see
 * tests/test_oracle_golden.py.  UNPINNED (no runnable reference, gatb-core
 * submodule absent): the A<C<T<G order (README.md:106-112 prose only) and the
 * per-partition row order of dsk2ascii (utils/dsk2ascii.cpp:77-104).
 *
 * Semantics followed (reference file:line):
 *   - 2-bit code A=0,C=1,T=2,G=3; canonical = min(fwd, revcomp) -- README.md:104-112
 *   - k-mer printed MSB-first, "<kmer> <count>\n"            -- utils/dsk2ascii.cpp:104,
 *                                                               test/short.parse_results:1
 *   - non-ACGT base breaks the window (no substitution)      -- test/readN.fasta + readN.histo
 *   - multi-line FASTA records joined; FASTQ 4-line records  -- README.md:52-61, test/longread.fasta
 *   - comma-separated file list = one summed count           -- scripts/simple_test.sh:52
 *   - histogram over ALL distinct k-mers, rows 1..histo_max  -- test/k27.histo:1,10000
 *   - solid <=> abundance_min <= count <= abundance_max      -- scripts/simple_test.sh:88
 *   - 32-bit abundance, partition-then-count algorithm       -- doc/paper.tex:60-97,104
 */
#ifndef DSK_ORACLE_H
#define DSK_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsko_result dsko_result;

/* Load a bank URI (comma-separated list of FASTA/FASTQ files, optionally
 * gzip'ed) into one byte stream: the sequence of every record, records
 * separated by a single '\n'.  Caller frees with dsko_free_stream.
 * Returns 0 on success. */
int dsko_load_bank(const char* uri, uint8_t** stream, uint64_t* nbytes, uint64_t* nreads);
void dsko_free_stream(uint8_t* stream);

/* Count canonical k-mers (1 <= k <= dsko_max_kmer_size()) of a byte stream in which every byte
 * outside "ACGTacgt" terminates the current window.  nthreads >= 1.  k in 65..128 uses 256-bit keys
 * (C23 _BitInt, clang builds only -- the Makefile picks ROCm's clang when it is there). */
int dsko_max_kmer_size(void);                       /* 128, or 64 for a gcc build */
dsko_result* dsko_count(const uint8_t* stream, uint64_t nbytes, int k, int nthreads);
void dsko_free(dsko_result* r);

uint64_t dsko_total_kmers(const dsko_result* r);    /* k-mer occurrences   */
uint64_t dsko_num_distinct(const dsko_result* r);   /* distinct canonical  */
/* Rows in ascending k-mer value (A<C<T<G, first base most significant).
 * lo = low 64 bits, hi = high 64 bits (0 when k <= 32). */
void dsko_rows(const dsko_result* r, uint64_t* lo, uint64_t* hi, uint32_t* abundance);
/* Same with all four 64-bit words of the value (w2 = w3 = 0 when k <= 64). */
void dsko_rows4(const dsko_result* r, uint64_t* w0, uint64_t* w1, uint64_t* w2, uint64_t* w3, uint32_t* abundance);
/* out[i] for i in 0..histo_max: number of distinct k-mers with
 * min(count, histo_max) == i  (out[0] is always 0). */
void dsko_histogram(const dsko_result* r, uint64_t* out, uint32_t histo_max);
/* Number of rows with amin <= abundance <= amax. */
uint64_t dsko_num_solid(const dsko_result* r, uint32_t amin, uint32_t amax);

/* Small helpers used by kernel-level parity tests. */
void dsko_kmer_to_string(uint64_t lo, uint64_t hi, int k, char* out /* k+1 bytes */);
/* Canonical k-mer ending at every position of the stream (in stream order):
 * valid[i]=1 and (lo[i],hi[i]) set iff a full ACGT window ends at byte i. */
void dsko_enumerate(const uint8_t* stream, uint64_t nbytes, int k,
                    uint64_t* lo, uint64_t* hi, uint8_t* valid);
/* k up to 128: words[4*i + x] = word x of the canonical k-mer ending at byte i.  Returns -1 in a gcc build. */
int dsko_enumerate4(const uint8_t* stream, uint64_t nbytes, int k, uint64_t* words, uint8_t* valid);
/* Minimizer (m <= 16) of every valid k-mer: smallest canonical m-mer value in
 * the window, A<C<T<G numeric order (restates the idea of
 * doc/paper.tex:60-76 "partition by a function of the k-mer"; the exact
 * gatb-core ordering is not in the reference tree => unpinned). */
void dsko_minimizers(const uint8_t* stream, uint64_t nbytes, int k, int m,
                     uint32_t* minim, uint8_t* valid);

#ifdef __cplusplus
}
#endif
#endif
