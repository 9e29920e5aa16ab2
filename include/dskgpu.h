/*
 * dskgpu.h -- C-ABI of the MI355X k-mer counting engine (libdskgpu.so).
 *
 * This is the drop-in boundary for the DSK count path.  In the reference the
 * whole path is ONE C++ call, `SortingCountAlgorithm<span>::execute()`
 * (src/DSK.cpp:55-60), fed by `Bank::open(-file)` (src/DSK.cpp:51) and read
 * back through `Partition<Kmer<span>::Count> "solid"` + the histogram
 * (utils/dsk2ascii.cpp:61-104; scripts/simple_test.sh:37).  The reference has
 * no FFI of its own (it is all in-process C++), so each entry point below cites
 * the reference interface it stands in for.  Plain pointers and sizes only; no
 * C++/torch types cross.  Every function returns DSKGPU_OK (0) or a negative
 * error code; `dskgpu_last_error` gives the text.  One ctx per device; a ctx
 * is not thread-safe, distinct ctxs are independent.
 *
 * Input convention ("read stream"): a byte string in which every maximal run
 * of [ACGTacgt] is one sequence fragment; ANY other byte (N, IUPAC codes,
 * '\n' between reads) ends the current k-mer window (test/readN.fasta +
 * test/readN.histo).  A bank front-end therefore only has to concatenate the
 * sequence lines of its records separated by one non-ACGT byte -- or hand over
 * the file's text as it is and let the device do that (dskgpu_push_raw).
 *
 * K-mer value convention (README.md:104-112, utils/dsk2ascii.cpp:104): A=0,
 * C=1, T=2, G=3, first base most significant; canonical = min(fwd, revcomp).
 * A k-mer is `words` = ceil(k / 32) 64-bit words, least-significant word first
 * (1 for k <= 32, 2 for k <= 64, 3 for k <= 96, 4 for k <= 128).
 */
#ifndef DSKGPU_H
#define DSKGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSKGPU_OK 0
#define DSKGPU_E_ARG (-1)       /* bad argument / unsupported k            */
#define DSKGPU_E_DEVICE (-2)    /* HIP error (text in dskgpu_last_error)   */
#define DSKGPU_E_NOMEM (-3)     /* device allocation failed                */
#define DSKGPU_E_STATE (-4)     /* call out of order                       */
#define DSKGPU_E_OVERFLOW (-5)  /* internal table overflow after retries   */
#define DSKGPU_E_FORMAT (-6)    /* dskgpu_push_raw: the text is not what was declared (see there)            */
#define DSKGPU_NOT_RESERVED 1   /* dskgpu_reserve_work only: nothing was reserved (the request exceeds 60 % of the free HBM) -- not an
                                   error: dskgpu_count sizes its own buffers, in several passes if need be; nothing was allocated */

typedef struct dskgpu_ctx dskgpu_ctx;

/* Options of SortingCountAlgorithm<>::getOptionsParser() that reach the
 * count path (src/DSK.cpp:83; README.md:12,56; scripts/simple_test.sh:36,88). */
typedef struct dskgpu_config {
    uint32_t kmer_size;       /* -kmer-size, 1..128 (gatb spans 32/64/96/128 serve k < span; README.md:115-122, CMakeLists.txt:42) */
    uint32_t abundance_min;   /* -abundance-min (solid <=> min <= count <= max) */
    uint32_t abundance_max;   /* -abundance-max                                */
    uint32_t histo_max;       /* -histo-max: histogram rows 1..histo_max (10000) */
    int32_t  device;          /* HIP device ordinal                            */
    uint32_t nb_partitions;   /* number of output partitions (dsk/solid/<p>); 0 = auto */
    uint32_t minimizer_size;  /* -minimizer-size (used by the owner map), 0 = default 10 */
    uint32_t flags;           /* DSKGPU_F_* */
    uint32_t world_size;      /* number of GPUs sharing the k-mer space (1 = single) */
    uint32_t rank;            /* this GPU's index in [0, world_size)           */
    uint32_t max_pass_mkeys;  /* most k-mers (in millions) one pass may hold; 0 = 4026 (32-bit offsets).
                                 Larger inputs are counted in several passes over the key space, the
                                 in-HBM counterpart of DSK's disk passes (README.md:126-130) */
    uint32_t solidity_kind;   /* DSKGPU_SOLIDITY_*: how the counts of several banks decide solidity (-solidity-kind) */
    uint32_t solidity_custom; /* DSKGPU_SOLIDITY_CUSTOM: bit b set = bank b must hold the k-mer (-solidity-custom) */
    uint32_t reserved[3];
} dskgpu_config;

#define DSKGPU_F_TIMING 1u        /* record per-stage HIP-event timings        */
#define DSKGPU_F_NO_SORT 2u       /* leave solid rows unsorted (bench ablation) */
#define DSKGPU_F_MG_EXPLICIT 8u   /* multi-GPU: exchange one explicit key per k-mer instead of super-k-mer records */
#define DSKGPU_F_HISTO2D 4u       /* also build the 2-D histogram: bank 0 (genome) x the other banks (reads), -histo2D */
#define DSKGPU_F_PARTITION_ORDER 32u /* the REFERENCE's row order instead of the global one: rows ascending inside an output partition, partitions = runs of
                                   * hash sub-partitions of at most 4096 rows (dskgpu_num_partitions of them: thousands) -- what Partition<Count> "solid"
                                   * guarantees its readers (utils/dsk2ascii.cpp:61,77,85-104: partition after partition, the rows of each as they
                                   * come; gatb-core's partitions are classes of the minimizer hash).  One pass over the rows instead of three
                                   * (csrc/partsort.h).  Honoured by a single-pass count of one- and two-word k-mers (k <= 64; 2048 rows per partition above k = 32); every other path -- and an
                                   * input on which a partition would exceed what one block orders -- keeps the global order, which satisfies the
                                   * same contract with nb_partitions value ranges. */
#define DSKGPU_F_PLACE 16u        /* pick the place of every big device buffer: where a buffer lies in HBM changes the rate of
                                   * scattered stores into it by up to 40 % (tools/micro/write_place.hip; the "two speeds" of the
                                   * partition kernels).  Each allocation >= 256 MB becomes the best of up to 8 candidates, timed
                                   * with the store pattern of the level-1 scatter; the others are freed.  One-off cost at the
                                   * first count (~3-6 s for 30 GB of buffers): for contexts that count many times.  Process-wide
                                   * once a context asked for it; DSKGPU_PLACE=<candidates> in the environment does the same. */

/* -solidity-kind (gatb-core option; only `sum` is exercised by the reference's tests, README.md:12).
 * Banks = the inputs separated with dskgpu_next_bank / dskgpu_set_banks; one bank => plain counting. */
#define DSKGPU_SOLIDITY_SUM 0u    /* amin <= sum of the banks' counts <= amax (default)            */
#define DSKGPU_SOLIDITY_MIN 1u    /* amin <= smallest per-bank count <= amax                       */
#define DSKGPU_SOLIDITY_MAX 2u    /* amin <= largest per-bank count <= amax                        */
#define DSKGPU_SOLIDITY_ONE 3u    /* at least one bank has amin <= count <= amax                   */
#define DSKGPU_SOLIDITY_ALL 4u    /* every bank has amin <= count <= amax                          */
#define DSKGPU_SOLIDITY_CUSTOM 5u /* banks in solidity_custom have count >= amin, the others 0     */

/* Lifetime: stands where `SortingCountAlgorithm<span> sortingCount(bank, props)`
 * is constructed / destroyed (src/DSK.cpp:55). */
int  dskgpu_create(const dskgpu_config* cfg, dskgpu_ctx** out);
void dskgpu_destroy(dskgpu_ctx* ctx);
const char* dskgpu_last_error(const dskgpu_ctx* ctx);   /* ctx may be NULL: create-time error */
const char* dskgpu_version(void);
int dskgpu_device_count(void);                          /* HIP devices visible to this process (0 when there is none) */

/* Launch all device work of this ctx on an existing HIP stream (hipStream_t
 * passed as void*); NULL = a stream owned by the ctx. */
int dskgpu_set_stream(dskgpu_ctx* ctx, void* hip_stream);

/* ---- input: replaces the Bank iteration inside execute() (src/DSK.cpp:51,60) */
/* Append host bytes of the read stream (copied to the device; may be called
 * repeatedly; a separator is implied between calls).  `bytes` may be reused as soon as the call returns (it is staged), but the
 * DMA to the device may still be in flight then: an asynchronous copy error is reported by the next call that synchronises
 * (dskgpu_count, dskgpu_encode_reads, dskgpu_set_stream, a buffer growth), not by this one. */
int dskgpu_push_reads(dskgpu_ctx* ctx, const char* bytes, uint64_t nbytes);
/* Append FILE TEXT -- FASTA or FASTQ exactly as it lies in the (inflated) file, headers and quality lines included -- and let the
 * device turn it into the read stream: replaces BankFasta's parser (gatb-core BankFasta behind src/DSK.cpp:51 Bank::open; the
 * formats of README.md:52-61) together with dskgpu_push_reads' clean stream.  The text may be cut ANYWHERE between calls (mid
 * line, mid record): the parser's state is kept on the device.  `new_file` != 0 says that `text` begins a new file (line
 * counting restarts, the records of two files never join); the first raw push of a read set is a new file by itself.
 *   DSKGPU_RAW_FASTQ  four lines per record ('@' header, sequence, '+' line, qualities): the sequence lines are kept
 *   DSKGPU_RAW_FASTA  '>' header lines are dropped, all other lines are sequence, joined when a record is wrapped over lines
 * Asynchronous like dskgpu_push_reads: nothing is known about the result until something needs the stream's length -- every
 * call that reads the reads (dskgpu_count, dskgpu_encode_reads, dskgpu_mg_*, dskgpu_next_bank, dskgpu_push_reads) first does
 * what dskgpu_raw_finish does.  A text the device parser does not handle (a FASTQ file with sequences wrapped over several
 * lines or blanks inside them, one whose quality lines do not add up to its sequence lines -- a host parser reads as many
 * quality characters as the record has bases --, text that is neither format) is DETECTED, never mis-parsed: dskgpu_raw_finish returns
 * DSKGPU_E_FORMAT and the stream is what it was before the raw pushes -- the caller parses on the host and pushes the reads. */
#define DSKGPU_RAW_FASTA 1
#define DSKGPU_RAW_FASTQ 2
int dskgpu_push_raw(dskgpu_ctx* ctx, const char* text, uint64_t nbytes, int format, int new_file);
/* Wait for the raw pushes; -> the read stream's length in bytes and the number of records (header lines) the raw pushes since the
 * last finish held -- Bank::estimate's sequence count (both may be NULL). */
int dskgpu_raw_finish(dskgpu_ctx* ctx, uint64_t* stream_bytes, uint64_t* records);
/* The length of the pushed read stream in bytes (waits for raw pushes), and its rewind: the stream is cut back to its first
 * `stream_bytes` bytes -- what was pushed behind them is forgotten (a bank front-end that parsed a damaged file in parallel and
 * found out that a serial parse would differ pushes that file again; src/DSK.cpp:51: the reference's parser is serial). */
int dskgpu_stream_bytes(dskgpu_ctx* ctx, uint64_t* stream_bytes);
int dskgpu_rewind_reads(dskgpu_ctx* ctx, uint64_t stream_bytes);
/* Optional: size the device-side read buffer once (e.g. from Bank::getSize) instead of growing it push by push. */
int dskgpu_reserve_reads(dskgpu_ctx* ctx, uint64_t nbytes);
/* Optional: allocate the partition buffers of a count over up to `nbytes` read-stream bytes now (tens of GB of HBM: 0.2 s of
 * hipMalloc on a 10 M-read input) instead of inside the first dskgpu_count.  May run on another host thread WHILE the reads
 * are pushed -- it touches nothing dskgpu_push_reads / dskgpu_reserve_reads use -- but must have returned before dskgpu_count.
 * Call order: before dskgpu_encode_reads or never -- after it the kept encoding is the only copy of the reads, and this call
 * then leaves the encoded stream's buffers as they are (it only sizes the partition buffers).
 * Stands where SortingCountAlgorithm's configure step sizes its passes and partitions before execute() fills them. */
int dskgpu_reserve_work(dskgpu_ctx* ctx, uint64_t nbytes);   /* DSKGPU_OK, DSKGPU_NOT_RESERVED (a soft refusal, see above), or an error */
/* Use a read stream already resident in HBM (caller keeps ownership and must
 * keep it alive until dskgpu_count returns).  Replaces any pushed reads.  The call waits for all device work this
 * process has submitted so far (hipDeviceSynchronize): bytes that another stream is still writing when it is made
 * are complete when it returns.  Bytes written AFTER the call must be ordered by the caller (same stream as
 * dskgpu_set_stream, or a synchronisation of its own) before dskgpu_count. */
int dskgpu_set_reads_device(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes);

/* Optional: turn the current reads into their 2-bit form NOW (the first stage of every count: 0.375 bytes per base) and let go of
 * the bytes: a device-resident stream handed over with dskgpu_set_reads_device is never read again after this call returns -- the
 * caller may free or overwrite it --, reads that were pushed lose their copy in HBM.  Every later dskgpu_count / dskgpu_mg_* of
 * these reads starts from the kept encoding.  For inputs whose ASCII form would crowd the partitions out of HBM: 90 Gbp of reads
 * are 90 GB as bytes and 34 GB encoded, and the difference decides how many sweeps over the reads a multi-pass count needs
 * (README.md:126-130: DSK reads its bank once per pass and keeps nothing of it in memory).  Not with per-bank modes
 * (-solidity-kind other than sum, -histo2D), which re-read bank by bank: DSKGPU_E_STATE there.  A new dskgpu_push_reads /
 * dskgpu_set_reads_device starts a new read set. */
int dskgpu_encode_reads(dskgpu_ctx* ctx);

/* Banks: the comma-separated inputs of `-file` are separate banks (README.md:52-58).  Call
 * dskgpu_next_bank between the pushes of two banks (or behind every bank, the last one too: a bank may be EMPTY -- a file without
 * reads, a rank's empty share of a small bank -- and is only known to exist by its call), or give the end offset of every bank of a
 * device-resident stream.  Only needed for -solidity-kind != sum and -histo2D; at most 32 banks. */
int dskgpu_next_bank(dskgpu_ctx* ctx);
int dskgpu_set_banks(dskgpu_ctx* ctx, const uint64_t* end_offsets, uint32_t n_banks);

/* ---- the hot path: replaces SortingCountAlgorithm<span>::execute() (src/DSK.cpp:60) */
/* Single-GPU: encode -> canonical k-mers -> partition -> count -> histogram +
 * solidity filter (+ sort).  Synchronous on return. */
int dskgpu_count(dskgpu_ctx* ctx);

/* Multi-GPU (world_size > 1): the k-mer space is split over owners in [0, world_size).  The owner of a
 * k-mer is a function of the MINIMIZER of its window (m = minimizer_size), so runs of consecutive k-mers
 * share an owner and travel as one super-k-mer record of 2-bit packed bases (2-3 words for up to 16
 * k-mers) -- DSK v2's minimizer repartition + super-k-mers (CHANGELOG.md:13) used as the wire format.
 * For k < 20 or with DSKGPU_F_MG_EXPLICIT the records are explicit keys (1-2 words per k-mer, owner =
 * a bit field of the mixed k-mer).  Step 1 writes this rank's records grouped by owner into caller
 * memory `d_send` (capacity in 8-byte words) and the per-owner word counts into send_words[world_size]
 * (host).  The caller exchanges the groups (RCCL all-to-all) and hands the received words to step 2.
 * dskgpu_mg_send_capacity_words runs the sizing pass over the current reads and returns the words
 * dskgpu_mg_scatter will need (0 on error). */
int dskgpu_mg_scatter(dskgpu_ctx* ctx, void* d_send, uint64_t capacity_words, uint64_t* send_words);
/* Minimizer repartition (gatb-core's RepartitorAlgorithm; the call site that reads its result is src/DSK.cpp:63 getConfig):
 * the owner of a window = table[bucket of its minimizer], DSKGPU_MG_BUCKETS buckets.  Default: the bucket scaled to
 * world_size.  A balanced table comes from sampled loads: every rank calls dskgpu_mg_sample on its reads (k-mers per bucket,
 * estimated from every 16th tile), the loads are summed over the ranks (an all-reduce of 32 KB), dskgpu_mg_make_table turns
 * the sum into a table -- deterministic, so every rank computes the same one -- and dskgpu_mg_set_table installs it before
 * dskgpu_mg_scatter.  A bucket holding more than a quarter of a fair share (poly-A, microsatellite minimizers) gets
 * DSKGPU_MG_SPLIT: its windows go to the owner of their own k-mer as one-k-mer records; the others are placed largest first
 * on the least loaded owner.  The table must be the same on every rank; results never depend on which table is used.
 * (dskgpu_group_count does all of this by itself.) */
#define DSKGPU_MG_BUCKETS 4096
#define DSKGPU_MG_SPLIT 255
int dskgpu_mg_sample(dskgpu_ctx* ctx, uint64_t* loads /* [DSKGPU_MG_BUCKETS] */);
void dskgpu_mg_make_table(const uint64_t* summed_loads, uint32_t world_size, uint8_t* table /* [DSKGPU_MG_BUCKETS] */);
int dskgpu_mg_set_table(dskgpu_ctx* ctx, const uint8_t* table /* [DSKGPU_MG_BUCKETS], NULL = default */);
uint64_t dskgpu_mg_send_capacity_words(dskgpu_ctx* ctx);
int dskgpu_mg_count(dskgpu_ctx* ctx, const void* d_recv, uint64_t recv_words);
/* The senders know how many k-mers they packed for every owner (counted while the records are written):
 * dskgpu_mg_sent_kmers returns that row after dskgpu_mg_scatter (host array of world_size entries), the caller sends it
 * along with the word counts, and the receiver passes the column sum to dskgpu_mg_count_sized -- which then skips its own
 * pass over the received records (0.6 ms of a 20 ms step).  n_kmers = 0 behaves as dskgpu_mg_count; a figure that does not
 * match the records is reported as DSKGPU_E_ARG, never as a wrong result. */
int dskgpu_mg_sent_kmers(dskgpu_ctx* ctx, uint64_t* kmers /* [world_size] */);
int dskgpu_mg_count_sized(dskgpu_ctx* ctx, const void* d_recv, uint64_t recv_words, uint64_t n_kmers);
/* ---- the same step in SLICES, so that the exchange overlaps both of its neighbours: the exchange of slice i runs while the
 * sender writes slice i + 1 and while the receiver partitions slice i - 1.  Needs the sampled send layout (its sizes are known
 * before a record exists: the word counts of ALL slices go round once, up front):
 *   dskgpu_mg_slices_prepare  sizing pass; *nslices = 0 when this input takes the one-piece path (small input, explicit keys, a
 *                             send slice overflowed before) -- then use dskgpu_mg_scatter / dskgpu_mg_count.  Otherwise
 *                             send_words[s * world_size + o] = words slice s holds for owner o (slice-major, owner-major inside: the
 *                             send buffer is the concatenation, dskgpu_mg_send_capacity_words in all), kmers_est[o] = estimated
 *                             k-mers for owner o over all slices (sizes the receiver; an estimate, never a correctness input).
 *   dskgpu_mg_scatter_slice   launches the sender of slice s on the context's stream and returns (no synchronisation): the caller
 *                             records an event behind it and starts that slice's exchange on another stream.
 *   dskgpu_mg_count_sliced    the receiver: d_recv holds the slices one after the other (slice_words[s] words each, whatever the
 *                             order of the sources inside a slice); gate(user, s) is called on the host right before the first
 *                             device work that reads slice s is enqueued -- the callee makes the context's stream wait for the
 *                             arrival of slice s (hipStreamWaitEvent / a torch work handle's wait()).  Slices are gated in order,
 *                             each once; a path that needs all records at once gates all of them first.  A gate that returns
 *                             non-zero (the wait failed: a collective timed out or was aborted) stops the count: no further device
 *                             work is enqueued, the remaining gates are still called (their result ignored), the call returns
 *                             DSKGPU_E_STATE and the context holds no result.
 *   dskgpu_mg_slices_finish   after the exchange: *overflowed != 0 when a slice of the SEND layout overflowed -- the records of this
 *                             step are then incomplete on some receivers, and every rank must repeat the step in one piece (the
 *                             decision is the caller's collective: OR the flags); the context will use exact counts from then on. */
typedef int (*dskgpu_slice_gate)(void* user, uint32_t slice);   /* 0 = the stream now waits for the slice; non-zero = it will never arrive */
int dskgpu_mg_slices_prepare(dskgpu_ctx* ctx, uint32_t want_slices, uint32_t* nslices, uint64_t* send_words /* [want_slices * world_size] */,
                             uint64_t* kmers_est /* [world_size] */);
int dskgpu_mg_scatter_slice(dskgpu_ctx* ctx, void* d_send, uint64_t capacity_words, uint32_t slice);
int dskgpu_mg_slices_finish(dskgpu_ctx* ctx, int* overflowed);
int dskgpu_mg_count_sliced(dskgpu_ctx* ctx, const void* d_recv, uint32_t nslices, const uint64_t* slice_words, uint64_t n_kmers_est,
                           dskgpu_slice_gate gate, void* user);

/* ---- results: replace the CountProcessor outputs read back through
 * Storage (src/DSK.cpp:68; utils/dsk2ascii.cpp:61-104; simple_test.sh:37) */
typedef struct dskgpu_stats {
    uint64_t n_bytes;        /* read-stream bytes processed              */
    uint64_t n_kmers;        /* valid k-mer occurrences                  */
    uint64_t n_distinct;     /* distinct canonical k-mers                */
    uint64_t n_solid;        /* rows passing the abundance filter        */
    uint32_t n_partitions;   /* output partitions                        */
    uint32_t n_levels;       /* radix-partition levels used              */
    uint32_t n_final_bins;   /* hash-aggregate sub-partitions            */
    uint32_t n_retries;      /* attempts repeated: a count table overflowed (finer plan), a sampled slice / the region pool overflowed (exact path), or (two-word keys) the top-word table met two k-mers with one top word (count stage again with full compares) */
    uint64_t sort_fallback;  /* 1 if the row sort needed its full-width fallback */
    uint64_t n_passes;       /* passes over the key space (1 unless the input exceeds a pass) */
    uint64_t n_ext_regions;  /* extension regions taken by sub-partitions that outgrew their home region (repeat-rich
                                inputs: heavy k-mers stay on the histogram-free path; 0 on repeat-free reads)   */
    uint64_t n_heavy;        /* k-mers counted apart from the partitions (found heavy in the sample pass)       */
    uint64_t n_read_sweeps;  /* times the (2-bit) reads were walked to generate k-mers: 1 for a single pass; a multi-pass
                                count materialises the keys of up to 16 passes per sweep -- this is what DSK calls a pass
                                (each one re-reads the input; README.md:126-130 "below 10")                             */
} dskgpu_stats;
int dskgpu_get_stats(const dskgpu_ctx* ctx, dskgpu_stats* out);

/* out[0..nbins-1]; out[i] = number of distinct k-mers whose
 * min(count, histo_max) == i; nbins must be histo_max+1 (out[0] == 0).
 * Same content as the `histogram/histogram` dataset (test/k27.histo). */
int dskgpu_histogram(const dskgpu_ctx* ctx, uint64_t* out, uint32_t nbins);

/* 2-D histogram (flag DSKGPU_F_HISTO2D): out[r * 11 + g] = number of distinct k-mers seen min(r, histo_max)
 * times in the read banks (banks 1..) and min(g, 10) times in bank 0; nrows must be histo_max + 1.
 * Text form `<out>.histo2D` (README.md:98-102; utils/plot-histo2D.R:22-30). */
int dskgpu_histogram2d(const dskgpu_ctx* ctx, uint64_t* out, uint32_t nrows);

/* Output partitions = `Partition<Count> "solid"` (utils/dsk2ascii.cpp:61,77).
 * Rows are ascending by k-mer value inside a partition; by default the partitions are
 * ascending value ranges too, so the concatenation is globally sorted (with
 * DSKGPU_F_PARTITION_ORDER only the order inside a partition is guaranteed). */
/* Row order of the NEXT counts of this context: non-zero = DSKGPU_F_PARTITION_ORDER (see the flag), 0 = the global order. */
int dskgpu_set_row_order(dskgpu_ctx* ctx, int partition_order);
uint32_t dskgpu_num_partitions(const dskgpu_ctx* ctx);
uint64_t dskgpu_partition_size(const dskgpu_ctx* ctx, uint32_t p);
/* offsets[p] = first row of partition p in the result arrays (dskgpu_result_device), offsets[num_partitions] = all rows:
 * what a caller of DSKGPU_F_PARTITION_ORDER walks (thousands of partitions: one call instead of one per partition). */
int dskgpu_partition_offsets(const dskgpu_ctx* ctx, uint64_t* offsets /* [dskgpu_num_partitions + 1] */);
/* kmers: size*words u64 (row-major, LSW first, words = ceil(k/32)); abundance: size u32. Host memory. */
int dskgpu_partition_copy(const dskgpu_ctx* ctx, uint32_t p, uint64_t* kmers, uint32_t* abundance);
/* Device pointers to the full sorted result (valid until the next count/destroy).  d_kmers = word 0 of
 * every row; the device keeps one array per word (struct of arrays), higher words via dskgpu_partition_copy. */
int dskgpu_result_device(const dskgpu_ctx* ctx, const void** d_kmers, const void** d_abundance, uint64_t* n_rows);

/* Per-stage device time of the last count (flag DSKGPU_F_TIMING).  Returns the
 * number of stages; fills up to `cap` entries.  names[i] are static strings. */
int dskgpu_stage_times(const dskgpu_ctx* ctx, const char** names, float* ms, int cap);

/* ---- the same call on N GPUs of one node, inside ONE process (what `dsk -nb-gpus N` runs): the reference's
 * single `execute()` (src/DSK.cpp:55-60) still leaves ONE storage with a flat list of solid partitions
 * (utils/dsk2ascii.cpp:61,77).  A group owns one ctx per rank (world_size = n_ranks, rank r on devices[r], its own
 * stream and host thread).  Feed every rank its share of the reads through dskgpu_group_ctx(g, r) with the input
 * calls above (any split of whole records is valid: counting is a group-by on the canonical k-mer), then
 * dskgpu_group_count runs mg_scatter -> exchange -> mg_count on all ranks at once.  The exchange is an
 * all-to-all-v of super-k-mer records: grouped ncclSend / ncclRecv over RCCL (one communicator per rank from
 * ncclCommInitAll; librccl is loaded on first use) when every rank has its own device, device-to-device copies
 * when ranks share a device (RCCL refuses duplicate devices: the multi-rank tests on a 1-GPU box).
 * DSKGPU_GROUP_TRANSPORT=rccl|copy (environment, read at create) overrides the choice.  n_ranks: power of two <= 64.
 * cfg->device / world_size / rank are ignored (set per rank).  Results: per rank through dskgpu_group_ctx, or merged:
 * the histogram is the element-wise sum; global partition P = p * n_ranks + r is local partition p of rank r. */
typedef struct dskgpu_group dskgpu_group;
int  dskgpu_group_create(const dskgpu_config* cfg, const int32_t* devices, uint32_t n_ranks, dskgpu_group** out);
void dskgpu_group_destroy(dskgpu_group* g);
const char* dskgpu_group_last_error(const dskgpu_group* g);   /* g may be NULL: create-time error */
uint32_t dskgpu_group_size(const dskgpu_group* g);
dskgpu_ctx* dskgpu_group_ctx(dskgpu_group* g, uint32_t rank);
const char* dskgpu_group_transport(const dskgpu_group* g);     /* "rccl" or "copy" */
int dskgpu_group_count(dskgpu_group* g);
uint64_t dskgpu_group_exchanged_words(const dskgpu_group* g);  /* 8-byte words that changed rank in the last count */
/* steps of the last count that ran in slices -- exchange overlapped with the sender and the receiver's level 1 (dskgpu_mg_slices_*;
 * DSKGPU_GROUP_SLICES = slices per step, default 4, < 2 = every step in one piece); 0 when the input took the one-piece path */
uint32_t dskgpu_group_sliced_steps(const dskgpu_group* g);
int dskgpu_group_histogram(const dskgpu_group* g, uint64_t* out, uint32_t nbins);
/* Per-bank modes (solidity_kind != sum, DSKGPU_F_HISTO2D; banks = dskgpu_next_bank on EVERY rank's context at the same
 * points of the stream): dskgpu_group_count counts the banks one by one with one repartition table, every rank applies the
 * solidity kind to the k-mers it owns; the 2-D histogram is the element-wise sum (README.md:98-102). */
int dskgpu_group_histogram2d(const dskgpu_group* g, uint64_t* out, uint32_t nrows);
int dskgpu_group_get_stats(const dskgpu_group* g, dskgpu_stats* out);   /* sums over the ranks */
uint32_t dskgpu_group_num_partitions(const dskgpu_group* g);
uint64_t dskgpu_group_partition_size(const dskgpu_group* g, uint32_t P);
int dskgpu_group_partition_copy(const dskgpu_group* g, uint32_t P, uint64_t* kmers, uint32_t* abundance);

/* ---- kernel-level entry points used by the parity tests (device pointers) */
/* ASCII -> 2-bit packed words + invalid mask, one u64 / u32 per 32 bases. */
int dskgpu_k_encode(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes, void* d_packed, void* d_invalid);
/* Canonical k-mer (words = ceil(k/32) u64, LSW first) + validity byte for the window ending at every byte. */
int dskgpu_k_enumerate(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes, void* d_kmers, void* d_valid);
/* Minimizer (u32) of the window ending at every byte (0 when invalid). */
int dskgpu_k_minimizers(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes, void* d_minim, void* d_valid);

#ifdef __cplusplus
}
#endif
#endif
