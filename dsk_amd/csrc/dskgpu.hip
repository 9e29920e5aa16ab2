// dskgpu.hip -- C-ABI (include/dskgpu.h) over the HIP kernels in kernels.h.
// Host-side orchestration of the count path that stands where
// SortingCountAlgorithm<span>::execute() is called (src/DSK.cpp:60).
// gfx950 only; there is no CPU fallback: every entry point needs a HIP device.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <string>
#include <vector>

#include <thread>
#include <system_error>
#include "../../include/dskgpu.h"
#include "kernels.h"
#include "superkmer.h"
#include "rowsort.h"
#include "rowsort2.h"
#include "partsort.h"
#include "rawparse.h"

#include <rocprim/device/device_radix_sort.hpp>

#define DSKGPU_VERSION "dskgpu 0.1 (gfx950)"

namespace {

thread_local std::string g_create_err;

// Where a big buffer lies in HBM decides how fast scattered stores into it go: of several 9.6 GB allocations of one process some take
// the store pattern of the level-1 scatter (256 blocks x 512 streams x 256-byte runs) in 3.14 ms and others in 3.41 (a streaming
// fill: 1.70 / 1.79), the same virtual address changes class after a free + malloc, and the kernels' "two speeds" (level 1: 3.8 /
// 4.5 ms) follow (tools/micro/write_place.hip).  With DSKGPU_PLACE = K > 1 every allocation of >= 256 MB is the best of up to K
// candidates, each timed with that store pattern (one-off: ~0.1 s per candidate of 10 GB); the others are freed.
// the row sort's device scalars (matrix length, two work counters [, the ties flag]) set from KERNEL ARGUMENTS: the sorts are called many
// times back to back from host loops, and an asynchronous copy from one host-side array would only be correct as long as the runtime
// stages pageable copies synchronously
// (z0, z1: two more words to clear -- the sort's "not finished in place" flag and the count of listed sub-buckets -- or null)
__global__ void k_set_rs_scalars(u32* __restrict__ sc4, u32 len, u32 nwords, u32* __restrict__ z0 = nullptr, u32* __restrict__ z1 = nullptr) {
    if (threadIdx.x == 0) sc4[0] = len;
    else if (threadIdx.x < nwords) sc4[threadIdx.x] = 0u;
    if (threadIdx.x == 32 && z0) *z0 = 0u;
    if (threadIdx.x == 33 && z1) *z1 = 0u;
}
__global__ __launch_bounds__(1024) void k_place_probe(unsigned long long* __restrict__ out, unsigned long long n) {
    const unsigned long long per_block = n / gridDim.x, per_stream = per_block / 512;
    unsigned long long* base = out + (unsigned long long)blockIdx.x * per_block;
    const int r = threadIdx.x >> 5, l = threadIdx.x & 31;
    for (unsigned long long off = 0; off + 32 <= per_stream; off += 32)
        for (int p = r; p < 512; p += 32) base[(unsigned long long)p * per_stream + off + l] = off;
}
int g_place_k = -1;          // DSKGPU_PLACE (read once)
float place_probe_ms(void* p, size_t bytes) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 0.f;
    float best = 1e30f;
    for (int i = 0; i < 3; ++i) {
        (void)hipEventRecord(a, 0);
        hipLaunchKernelGGL(k_place_probe, dim3(256), dim3(1024), 0, 0, static_cast<unsigned long long*>(p), (unsigned long long)(bytes / 8));
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0.f; (void)hipEventElapsedTime(&ms, a, b);
        if (i && ms < best) best = ms;
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best;
}
hipError_t placed_malloc(void** out, size_t bytes) {
    if (g_place_k < 0) { const char* e = getenv("DSKGPU_PLACE"); g_place_k = e ? atoi(e) : 0; }      // (dskgpu_create with DSKGPU_F_PLACE sets 8)
    size_t free_b = 0, total_b = 0;
    if (g_place_k < 2 || bytes < (size_t(1) << 28) || hipMemGetInfo(&free_b, &total_b) != hipSuccess) return hipMalloc(out, bytes);
    const int K = (int)std::min<size_t>((size_t)g_place_k, free_b / 2 / bytes);      // candidates held at once: at most half of what is free
    if (K < 2) return hipMalloc(out, bytes);
    std::vector<void*> cand; std::vector<float> ms;
    for (int i = 0; i < K; ++i) {
        void* q = nullptr;
        if (hipMalloc(&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        static const size_t probe_max = getenv("DSKGPU_PLACE_GB") ? (size_t)(atof(getenv("DSKGPU_PLACE_GB")) * (1ull << 30)) : ~size_t(0);
        cand.push_back(q); ms.push_back(place_probe_ms(q, std::min(bytes, probe_max)));
    }
    if (cand.empty()) return hipErrorOutOfMemory;
    size_t best = 0;
    for (size_t i = 1; i < cand.size(); ++i) if (ms[i] < ms[best]) best = i;
    for (size_t i = 0; i < cand.size(); ++i) if (i != best) (void)hipFree(cand[i]);
    if (getenv("DSKGPU_VERBOSE")) {
        fprintf(stderr, "[dskgpu] placement of %.2f GB: probe ms", bytes * 1e-9);
        for (size_t i = 0; i < cand.size(); ++i) fprintf(stderr, " %.3f%s", ms[i], i == best ? "*" : "");
        fprintf(stderr, "\n");
    }
    *out = cand[best];
    return hipSuccess;
}

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        // growing a buffer that exists: 6 % on top, so that a size that wobbles by a few per cent from pass to pass (slices from
        // sampled loads) does not free and allocate tens of GB again -- near a full HBM that took a second
        size_t want = ((p ? bytes + bytes / 16 : bytes) + 255) & ~size_t(255);
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = placed_malloc(&p, want);
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    // grow, keeping the first `keep` bytes (device-to-device copy on `stream`); returns 0 on success
    int ensure_keep(size_t bytes, size_t keep, hipStream_t stream) {
        if (bytes <= cap) return 0;
        void* np = nullptr;
        size_t want = ((std::max(bytes, cap + cap / 2) + 255) & ~size_t(255));
        if (hipMalloc(&np, want) != hipSuccess) return -1;
        if (keep && p) {
            if (hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, stream) != hipSuccess || hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(np); return -1; }
        }
        if (p) (void)hipFree(p);
        p = np; cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum Scalar { SC_NCH1 = 0, SC_MLEN1, SC_NCH2, SC_MLEN2, SC_OVERFLOW, SC_F, SC_SORTFLAG, SC_OVF2, SC_OVF1, SC_RSLEN, SC_RSWORK, SC_RSWORK2, SC_RSTIES, SC_EXT, SC_NCHAINED, SC_NCH_S, SC_MLEN_S, SC_WORK2, SC_COUNT = 24 };
// everything the host wants to know after the count stage of a pass, gathered into ONE 80-byte record: seven 4- and 32-byte copies from
// four buffers into pageable host memory cost ~20 us of idle GPU each (the runtime stages every one of them), 0.12 ms of a 14 ms step
__global__ void k_gather_back(const u32* __restrict__ sc, const u32* __restrict__ nsolid_total, const u32* __restrict__ nk, const u64* __restrict__ gstats,
                              u64* __restrict__ out) {
    if (threadIdx.x == 0) {
        out[0] = sc[SC_OVERFLOW]; out[1] = *nsolid_total; out[2] = nk ? *nk : 0u; out[3] = sc[SC_OVF2]; out[4] = sc[SC_OVF1]; out[5] = sc[SC_EXT];
        out[6] = gstats[0]; out[7] = gstats[1]; out[8] = gstats[2]; out[9] = gstats[3];
    }
}
// start of a pass attempt: the device scalars (from kernel arguments), an empty abundance histogram, zeroed statistics -- one launch where a
// copy and two memsets were four (a memset of 80 008 bytes is two fill kernels)
struct ScalarSet { u32 v[SC_COUNT]; };
// (zero / nzero: the level-2 region fill counts, zeroed here as well when the pass takes the fixed-capacity regions)
__global__ __launch_bounds__(256) void k_setup_pass(u32* __restrict__ sc, ScalarSet h, u64* __restrict__ ghist, u32 nh, u64* __restrict__ gstats, u32 nstats,
                                                    u32* __restrict__ zero, u64 nzero) {
    const u32 t = blockIdx.x * 256 + threadIdx.x;
    if (t < SC_COUNT) sc[t] = h.v[t];
    if (t < nstats) gstats[t] = 0ull;
    for (u32 i = t; i < nh; i += gridDim.x * 256) ghist[i] = 0ull;
    for (u64 i = t; i < nzero; i += (u64)gridDim.x * 256) zero[i] = 0u;
}
// the row sort's two words for the host (its "could not finish in place" flag, the number of listed sub-buckets), put behind the abundance
// histogram so that the end of a step is ONE copy
__global__ void k_sort_back(const u32* __restrict__ sc, const u32* __restrict__ ovs, u64* __restrict__ out) {
    if (threadIdx.x == 0) { out[0] = sc[SC_SORTFLAG]; out[1] = ovs ? ovs[0] : 0u; }
}
#ifndef SORT_TOP_BITS
#define SORT_TOP_BITS 32u      // 4 radix passes; 40 bits (5 passes) cost 0.3 ms more on 43 M rows, the in-place run fix-up absorbs the extra ties
#endif

struct Stage { const char* name; hipEvent_t ev; };

}  // namespace

// Test / experiment switches (NOTEBOOK.md "Environment switches"): read ONCE from the environment when a context is
// created; none of them changes results.  The launch paths only look at this struct.
struct Tuning {
    bool no_opt1 = false, no_opt2 = false;      // DSKGPU_NO_OPT1 / _NO_OPT2: exact histogram + scan path at level 1 / at both levels
    bool no_aligned = false;                    // DSKGPU_NO_ALIGNED: plain write-out for key-array scatters
    u32 rs_heavy = 0;                           // DSKGPU_RS_HEAVY: rows of a first-digit bucket above which the row sort gives up (tests)
    u32 rs_bbits = 0;                           // DSKGPU_RS_BBITS: forced width of the row sort's second digit (8..10; tests)
    u32 rs_block_rows = 0;                      // DSKGPU_RS_BLOCK_ROWS: largest sub-bucket the hand-written row sort orders itself (tests: provoke its fallback)
    bool sk_exact = false, no_recsrc = false;   // DSKGPU_SK_EXACT, DSKGPU_NO_RECSRC (multi-GPU sender layout / receiver source)
    u32 opt_cap = 0;                            // DSKGPU_OPT_CAP: forced level-2 region size (keys)
    u64 opt_slice = 0;                          // DSKGPU_OPT_SLICE: forced level-1 slice size (keys)
    u64 sk_slice = 0, sk_minslice = 2000;       // DSKGPU_SK_SLICE, DSKGPU_SK_MINSLICE
    u32 table_maxload = 0;                      // DSKGPU_TABLE_MAXLOAD: distinct keys a count table may hold (forces the finer-partition retry)
    long long max_ext = -1;                     // DSKGPU_MAX_EXT: size of the extension-region pool of the level-2 scatter (tests: 0 = no chains)
    bool no_sample = false;                     // DSKGPU_NO_SAMPLE: level-1 slices from the mean load instead of the sampled per-bin loads
    bool no_heavy = false;                      // DSKGPU_NO_HEAVY: no k-mer is counted apart by the level-1 scatter
    bool verbose = false;                       // DSKGPU_VERBOSE: trace of the plan decisions on stderr
    bool no_level0 = false; u32 l0_passes = 0;  // DSKGPU_NO_LEVEL0: every pass of a multi-pass count re-generates its keys; DSKGPU_L0_PASSES=n: passes per level-0 sweep (tests)
    u64 rs_max_rows = 0;                        // DSKGPU_RS_MAX_ROWS: most rows the MSD row sort takes in one piece (tests: the group-wise path of huge row sets on a small input)
    u32 ps_maxc = 0;                            // DSKGPU_PS_MAXC: rows sharing a value bin that the partition-order sort still orders (tests: 1 provokes its fallback to the global order)
    bool sk_generic = false;                    // DSKGPU_SK_GENERIC: the sender kernels with k and m at run time even for k = 31 / 63 (tests: both forms write the same records)
    bool l0_keys = false;                       // DSKGPU_L0_KEYS: level 0 as key arrays (k_level0) even where the record-based one applies (experiments, tests)
    u32 mp_pass_mkeys = 0;                      // DSKGPU_MP_PASS_MKEYS: keys (millions) per pass of an input that needs several passes (default 1000)
    u64 rs_slab_rows = 0;                       // DSKGPU_RS_SLAB_ROWS: rows per slab of the row sort for >= 2^32 rows (tests: forces that path, with small slabs, on a small input)
    void read() {
        auto on = [](const char* n) { return getenv(n) != nullptr; };
        auto num = [](const char* n, u64 dflt) { const char* e = getenv(n); return e ? (u64)atoll(e) : dflt; };
        no_opt1 = on("DSKGPU_NO_OPT1"); no_opt2 = on("DSKGPU_NO_OPT2"); no_aligned = on("DSKGPU_NO_ALIGNED");
        sk_exact = on("DSKGPU_SK_EXACT");
        no_recsrc = on("DSKGPU_NO_RECSRC");
        opt_cap = (u32)num("DSKGPU_OPT_CAP", 0) & ~7u; opt_slice = num("DSKGPU_OPT_SLICE", 0) & ~7ull;
        sk_slice = num("DSKGPU_SK_SLICE", 0); sk_minslice = num("DSKGPU_SK_MINSLICE", 2000);
        table_maxload = (u32)num("DSKGPU_TABLE_MAXLOAD", 0);
        max_ext = getenv("DSKGPU_MAX_EXT") ? atoll(getenv("DSKGPU_MAX_EXT")) : -1;
        no_sample = on("DSKGPU_NO_SAMPLE"); no_heavy = on("DSKGPU_NO_HEAVY"); verbose = on("DSKGPU_VERBOSE"); no_level0 = on("DSKGPU_NO_LEVEL0"); l0_passes = (u32)num("DSKGPU_L0_PASSES", 0); mp_pass_mkeys = (u32)num("DSKGPU_MP_PASS_MKEYS", 0); rs_max_rows = num("DSKGPU_RS_MAX_ROWS", 0); l0_keys = on("DSKGPU_L0_KEYS"); sk_generic = on("DSKGPU_SK_GENERIC"); ps_maxc = (u32)num("DSKGPU_PS_MAXC", 0);
        rs_slab_rows = num("DSKGPU_RS_SLAB_ROWS", 0);
        rs_block_rows = (u32)num("DSKGPU_RS_BLOCK_ROWS", 0); rs_bbits = (u32)num("DSKGPU_RS_BBITS", 0); rs_heavy = (u32)num("DSKGPU_RS_HEAVY", 0);
    }
};

// state of a per-bank count in steps (banks_begin .. banks_finish below)
struct BankJob { dskgpu_config cfg; const uint8_t* base; u64 total; std::vector<u64> ends; u64 nu, tot_kmers; u32 passes, retries; bool active = false; };

struct dskgpu_ctx {
    dskgpu_config cfg{};
    BankJob bank_job{};
    Tuning tune;
    int W = 1;
    int words_out = 1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cu = 256;
    std::string err;

    // input
    DevBuf reads_own; u64 reads_len = 0;
    void* pin[2] = {nullptr, nullptr}; hipEvent_t pin_ev[2] = {nullptr, nullptr}; bool pin_used[2] = {false, false}; int pin_next = 0;   // pinned H2D staging
    const uint8_t* d_reads = nullptr; u64 n_bytes = 0;
    // dskgpu_push_raw: file text parsed on the device; the stream's length is on the device (RawState) until raw_finish reads it back
    DevBuf raw_in, raw_blk, raw_boff, raw_bstate, raw_state;
    bool raw_pending = false; u64 raw_base = 0, raw_ub = 0;      // the stream's length before the raw pushes / an upper bound of it now

    DevBuf packed, inval;          // K1 output
    DevBuf bufA, bufB;             // partition ping-pong
    DevBuf mat1, mat2, sums, descs1, descs2, seg, fstart, nsolid, scalars, ghist, gstats, chain_next;
    DevBuf fix_list;               // multi-word row sort: [count | (first row, rows) x FIX_LIST_CAP] of the prefix runs above FIX_CAP rows
    DevBuf rs_g[4];                // row sort: the gathered rows of the listed sub-buckets, one buffer per round (sort_oversize)
    DevBuf rs_del, rs_lens;        // row sort of >= 2^32 rows: per (slab, bin) 64-bit output offsets; the slabs' matrix lengths
    std::vector<u64> h_rs_del; std::vector<u32> h_rs_lens, h_rs_lin;
    DevBuf rs_ovs;                 // row sort: [count | (offset, rows, bits left) x RS_OVS_CAP] of the sub-buckets listed for another round
    std::vector<u32> h_ovs;
    u64* rs_res_k = nullptr; u32* rs_res_v = nullptr; u64* rs_tmp_k = nullptr; u32* rs_tmp_v = nullptr;   // the partially sorted rows and their scratch twin (sort_oversize)
    DevBuf smp_keys;               // records: the sample expanded to a key array (16 slots per candidate record, sentinel pads)
    DevBuf smp_mat, smp_descs, boff;   // sampled level-1 loads: chunk x bin matrix of the sample tiles, their descriptors; per-bin slice offsets
    DevBuf dbg, l0buf; DevBuf hv_lut, hv_collect, hv_buf; // heavy k-mers: bin -> collect slot, collected sample keys; [keys | counts | rows] of the k-mers counted apart
    std::vector<unsigned char> h_hv_lut; std::vector<u32> h_hv_cnt, h_hv_step; std::vector<u64> h_hv_coll, h_hv_keys;
    std::vector<ChunkDesc> h_descs_s; std::vector<u64> h_cbeg; std::vector<u32> h_boff; std::vector<u64> h_mom; std::vector<double> h_load, h_spread, h_seg_work;
    DevBuf out_w[4], srt_w[4], acc_w[4];   // rows as struct-of-arrays: word i of every row in [i]
    DevBuf out_ab, srt_ab, srt_tmp, srt_idx, srt_idx2, srt_k, srt_k2, abund2, acc_ab;   // srt_k2: one record per row for the multi-word gather
    u64 max_keys_per_pass = 0;     // 0 = as many as 32-bit offsets allow
    // multi-bank mode (solidity kinds, 2-D histogram)
    std::vector<u64> bank_ends;    // end offset of every declared bank in the read stream
    DevBuf u_w[4], s_w[4], u_val, s_val, m_flag, m_pos, m_sum, gh2d;
    std::vector<u64> hist2d;
    std::vector<u32> h_starts;     // explicit-key exchange: first key of every owner in the send buffer
    std::vector<u64> h_rstart;     // record exchange: first RECORD of every owner in the send buffer (64-bit: no limit on a rank's shard)
    std::vector<u32> h_sk_cells;   // ... records per (owner, chunk) as the sizing pass counted them
    std::vector<u64> h_sk_cb64;    // ... exact layout: their exclusive scan (owner-major) = record index of every (owner, chunk) pair
    DevBuf sk_cb64;                // ... the same on the device (k_sk_scatter<false>)
    // multi-GPU exchange as super-k-mer records (superkmer.h)
    bool enc_keep = false;         // dskgpu_encode_reads: packed / inval hold the 2-bit form of the current reads and the ASCII bytes are gone (d_reads == nullptr)
    bool enc_fresh = false;        // packed / inval hold the encoding of the current reads, left by dskgpu_mg_sample for the sender's sizing pass of the same step
    bool sk_mode = false, sk_prepared = false;
    bool sk_slices = false;        // the prepared send layout is slices from a sampled estimate (else exact offsets)
    bool sk_exact = false;         // a slice overflowed on these reads: exact counts from now on
    SkParams sk_sp{};
    DevBuf sk_sums, sk_cbase, sk_keys, sk_table, sk_load, sk_sent, sk_lay;      // sk_lay: [region base per owner: u64 x 64][slice per owner: u32 x 64] of a record-based level-0 sweep
    u64 last_rows = 0;             // solid rows of the last count of the current reads (0 = not counted yet): sizes what a multi-pass count keeps free for its rows
    bool rec_l0_off = false;       // these reads do not take the record-based level 0 (a slice of its sampled layout overflowed)
    u64 h_sk_sent[SK_MAX_OWNERS] = {0};      // k-mers inside the records the last mg_scatter wrote for every owner
    u64 h_sk_est[SK_MAX_OWNERS] = {0};       // sampled layout: estimated k-mers per owner (k_sk_hist on every 16th tile, scaled)
    u32 sk_nslices = 0;                      // dskgpu_mg_slices_prepare: slices of the prepared step (0 = none prepared)
    u64 rec_hint = 0;                        // dskgpu_mg_count_sized: the caller's k-mer total of the records (0 = none)
    bool rec_hint_est = false;               // ... an estimate (sliced step): sizes the fast path only, never checked against the result
    std::vector<u64> rec_slice_end;          // dskgpu_mg_count_sliced: record index where every slice ends; empty = one piece
    dskgpu_slice_gate rec_gate = nullptr; void* rec_gate_user = nullptr; u32 rec_gated = 0;      // slices whose arrival the stream already waits for
    bool rec_gate_failed = false;            // a gate said its slice will never arrive: the count stops (DSKGPU_E_STATE)
    std::vector<u32> h_slice_chunk;          // first level-1 chunk of every slice (+ the end)
    DevBuf cur_state;                        // parked write cursors of the level-1 blocks between the launches of a sliced receive
    bool rec_sized = false;                  // per-chunk k-mer sums of the records are on the device (k_sk_count ran)
    std::vector<uint8_t> h_table;  // the repartition table in use (SK_BUCKETS owners; default: bucket scaled to the world size)
    bool table_dirty = true;       // h_table not yet copied to sk_table
    std::vector<u32> h_sk_sums; std::vector<u64> h_sk_cbase;
    // records handed to dskgpu_mg_count: the level-1 scatter reads them directly (SRC 2); expanded lazily for the exact path
    const u64* rec_src = nullptr; u64 rec_n = 0; u64 rec_nch = 0, rec_rpc = 0; bool rec_expanded = false;
    int sort_back = 0; u64* hist_pin = nullptr; size_t hist_pin_n = 0;      // (sort_back: 1 = flag + sub-bucket count still on the device, 2 = flag only; see k_sort_back)
    u64* land = nullptr;                             // 64 KB of pinned host memory: where the small per-step read-backs land (landing())
    DevBuf back_dev; u64* back_host = nullptr;      // the count stage's read-back record (k_gather_back) and its pinned landing zone
    u32 h_back[4] = {0}; u64 h_stats[4] = {0}; u32 h_ovf2 = 0, h_ovf1 = 0, h_ext = 0; u64 h_nvalid = 0; bool have_nvalid = false;
    u64* fb_src_k = nullptr; u32* fb_src_v = nullptr; u64* fb_dst_k = nullptr; u32* fb_dst_v = nullptr;   // one-word row sort: where the full-width fallback finds a permutation of the rows / leaves them sorted
    bool sentinel_ok = true;       // the all-ones key is not the mixed form of a canonical k-mer of this k (checked at create)
    bool opt1_off = false;         // same for the histogram-free level-1 scatter (block-owned slices)
    bool mw_v3_off = false;        // the top-word table of k_count2v3 met two k-mers it cannot tell apart on these reads: k_count_mw from now on
    bool opt2_off = false;         // the fixed-capacity level-2 scatter overflowed on these reads: use the exact path   // host landing zone of the async size read-back
    std::vector<ChunkDesc> h_descs1, h_descs2;
    std::vector<const void*> big_lds_fns;   // kernels whose dynamic-LDS limit this context has raised (allow_big_lds)
    u32 h_sc[SC_COUNT] = {0};      // host mirror of the device scalars (kept alive across async copies)

    // the solid rows of a single one-word pass where the count kernel left them (run_one_pass): the row sort's first step reads them there
    // instead of a dense copy made by k_compact (rowsort.h: RsSparse)
    u32 job_passes = 1;            // passes of the running count as run_pipeline sees them (a pass of a record-based multi-pass count runs as "pass 0 of 1" inside run_one_pass)
    struct SparseRows { bool valid = false; RsSparse s{}; u64 n_sparse = 0; const u64* tail_k = nullptr; const u32* tail_v = nullptr; u32 n_tail = 0; } sp_rows;
    // multi-pass jobs: where a pass may put its dense rows straight away -- the job's accumulators, from row `rows` on (run_pipeline sets it
    // once they are sized; run_one_pass sets `took` when it did: the pass's rows are then already appended)
    struct RowSink { bool active = false, took = false; u32* ab = nullptr; u64* w[4] = {nullptr, nullptr, nullptr, nullptr}; u64 rows = 0, cap = 0; } sink;
    struct SparseRows2 { bool valid = false; Rs2Sparse s{}; u64 n_sparse = 0; Rows2C tail{nullptr, nullptr, nullptr}; u32 n_tail = 0; } sp_rows2;      // (two-word rows)
    // results
    bool have_result = false;
    bool sort_partial = false;
    // DSKGPU_F_PARTITION_ORDER (partsort.h): the rows ascending inside each output partition only.  part_mode: the last result is laid
    // out that way, h_part_off[p] = first row of partition p (n_parts + 1 entries, pinned); part_off_this_count: the flag was raised
    // (a partition or a bin above what a block orders) and this count takes the global sort instead
    bool part_mode = false, part_off_this_count = false; u32 n_parts = 0; u32* h_part_off = nullptr; size_t h_part_cap = 0; DevBuf part_off;
    SparseRows sp_rows_saved; SparseRows2 sp_rows2_saved;
    // ... and for the passes of a multi-pass count: every pass orders its rows partition by partition straight into the job's row arrays
    // (one k_part_sort launch instead of k_compact), the offsets of its partitions -- relative to the pass's first row -- go to
    // mp_part_off[off_index ..], one flag (mp_flag) serves the whole job; at the end the offsets become 64-bit row numbers (h_part_off64)
    struct MpPart { u64 row_base; u32 nparts, off_index; };
    std::vector<MpPart> mp_parts; DevBuf mp_part_off, mp_flag; u32 mp_off_used = 0; bool mp_part_ok = false;
    std::vector<u64> h_part_off64; std::vector<u32> h_mp_off;
    bool rows2_in_scratch = false; Rows2 rows2_scratch{};      // two-word rows above RS_MAX_ROWS: the result (and the fallback's input) is the scratch copy
    u64 n_rows = 0;
    const u64* res_w[4] = {nullptr, nullptr, nullptr, nullptr}; const u32* res_ab = nullptr;
    dskgpu_stats stats{};
    std::vector<u64> hist;

    // timing
    std::vector<Stage> marks;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    std::vector<const char*> st_names; std::vector<float> st_ms;

    void mark(const char* name) {
        if (!(cfg.flags & DSKGPU_F_TIMING)) return;
        if (ev_used == ev_pool.size()) { hipEvent_t e; (void)hipEventCreate(&e); ev_pool.push_back(e); }
        hipEvent_t e = ev_pool[ev_used++];
        (void)hipEventRecord(e, stream);
        marks.push_back({name, e});
    }
    void resolve_marks() {   // after a stream sync; appends to st_names/st_ms
        for (size_t i = 0; i + 1 < marks.size(); ++i) {
            float ms = 0; (void)hipEventElapsedTime(&ms, marks[i].ev, marks[i + 1].ev);
            // one entry per stage name: the passes of a multi-pass count (and the retries of a pass) add up
            size_t at = 0;
            while (at < st_names.size() && std::strcmp(st_names[at], marks[i + 1].name) != 0) ++at;
            if (at == st_names.size()) { st_names.push_back(marks[i + 1].name); st_ms.push_back(ms); } else st_ms[at] += ms;
        }
        marks.clear(); ev_used = 0;
    }
};

namespace {

#define CK(expr)                                                                             \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return (e_ == hipErrorOutOfMemory) ? DSKGPU_E_NOMEM : DSKGPU_E_DEVICE;           \
        }                                                                                    \
    } while (0)

#define CKL(what)                                                                            \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) {                                                              \
            ctx->err = std::string(what) + ": " + hipGetErrorString(e_);                     \
            return DSKGPU_E_DEVICE;                                                          \
        }                                                                                    \
    } while (0)

int fail(dskgpu_ctx* ctx, int code, const std::string& msg) { ctx->err = msg; return code; }

// Pinned host memory for the small read-backs a step waits on (a copy into pageable memory is staged by the runtime: ~10 us more of idle
// GPU per host round trip).  nullptr when the allocation fails or the request is larger: the caller then copies into its own buffer.
#define LAND_BYTES (64u << 10)
void* landing(dskgpu_ctx* ctx, size_t bytes) {
    if (bytes > LAND_BYTES) return nullptr;
    if (!ctx->land && hipHostMalloc(reinterpret_cast<void**>(&ctx->land), LAND_BYTES, hipHostMallocDefault) != hipSuccess) { ctx->land = nullptr; (void)hipGetLastError(); }
    return ctx->land;
}

// Is the all-ones key (the pad / empty-slot sentinel) the mixed form of a real canonical k-mer of this k?
// (It never is for k <= 32; for wider keys it is checked here once and the sentinel-based paths are switched off if so.)
bool sentinel_is_a_kmer(int W, unsigned k) {
    u64 v[4] = {0, 0, 0, 0};
    if (W == 1) v[0] = kunmix(~0ull);
    else if (W == 2) { KN<2> x; x.w[0] = x.w[1] = ~0ull; kunmixN(x); v[0] = x.w[0]; v[1] = x.w[1]; }
    else { KN<4> x; for (int i = 0; i < 4; ++i) x.w[i] = ~0ull; kunmixN(x); for (int i = 0; i < 4; ++i) v[i] = x.w[i]; }
    auto base = [&](const u64* a, unsigned i) { return (unsigned)((a[i >> 5] >> (2 * (i & 31))) & 3u); };   // base i counted from the LAST base
    for (unsigned i = k; i < 128; ++i) if (base(v, i)) return false;          // does not fit in 2k bits
    u64 r[4] = {0, 0, 0, 0};                                                  // reverse complement
    for (unsigned i = 0; i < k; ++i) { const unsigned c = base(v, i) ^ 2u, j = k - 1 - i; r[j >> 5] |= (u64)c << (2 * (j & 31)); }
    for (int x = 3; x >= 0; --x) if (v[x] != r[x]) return v[x] < r[x];        // canonical iff value <= its reverse complement
    return true;
}


// ---- K1 launcher
int run_encode(dskgpu_ctx* ctx, const uint8_t* d_bytes, u64 n, u64* nwords_out) {
    const u64 nwords = (n + 31) / 32;
    ctx->enc_fresh = false;
    CK(ctx->packed.ensure((nwords + 1) * 8));
    CK(ctx->inval.ensure((nwords + 1) * 4));
    if (nwords) {
        const u64 nb = (nwords + 255) / 256;
        const unsigned grid = (unsigned)std::min<u64>(nb, (u64)ctx->num_cu * 16);
        if ((reinterpret_cast<uintptr_t>(d_bytes) & 15) == 0)
            hipLaunchKernelGGL(k_encode<true>, dim3(grid), dim3(256), 0, ctx->stream, d_bytes, n,
                               ctx->packed.as<u64>(), ctx->inval.as<u32>(), nwords);
        else
            hipLaunchKernelGGL(k_encode<false>, dim3(grid), dim3(256), 0, ctx->stream, d_bytes, n,
                               ctx->packed.as<u64>(), ctx->inval.as<u32>(), nwords);
        CKL("k_encode");
    }
    *nwords_out = nwords;
    return DSKGPU_OK;
}

// the 2-bit form of the context's current reads: encoded now, or kept from dskgpu_encode_reads (the ASCII bytes may be gone by then)
int encode_current(dskgpu_ctx* ctx, u64* nwords_out) {
    if (ctx->enc_keep) { *nwords_out = (ctx->n_bytes + 31) / 32; ctx->enc_fresh = false; return DSKGPU_OK; }
    if (!ctx->d_reads && ctx->n_bytes) return fail(ctx, DSKGPU_E_STATE, "the reads were released (dskgpu_encode_reads) and their 2-bit form has been overwritten: set the reads again");
    return run_encode(ctx, ctx->d_reads, ctx->n_bytes, nwords_out);
}

// ---- scan launcher: exclusive scan of a[0..*d_len) in place, total -> a[*d_len]
int run_scan(dskgpu_ctx* ctx, u32* a, const u32* d_len, u64 max_len) {
    const u64 nb = std::max<u64>(1, (max_len + SCAN_BLK - 1) / SCAN_BLK);
    CK(ctx->sums.ensure((nb + 1) * 4));
    hipLaunchKernelGGL(k_scan_reduce, dim3((unsigned)nb), dim3(SCAN_NT), 0, ctx->stream, a, d_len, ctx->sums.as<u32>());
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, ctx->stream, ctx->sums.as<u32>(), d_len, a);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(SCAN_NT), 0, ctx->stream, a, d_len, ctx->sums.as<u32>());
    CKL("k_scan");
    return DSKGPU_OK;
}

// Kernels that stage a whole tile need more dynamic LDS than the 64 KB default: raise the limit once per context
// (a context is bound to one device and driven by one thread, so no process-wide flag is involved).
int allow_big_lds(dskgpu_ctx* ctx, const void* fn, int bytes = 160 * 1024) {      // (bytes: kernels with static LDS next to the dynamic block ask for what they use)
    for (const void* f : ctx->big_lds_fns) if (f == fn) return DSKGPU_OK;
    CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    ctx->big_lds_fns.push_back(fn);
    return DSKGPU_OK;
}

size_t scatter_lds(int W, u32 P, bool opt = false) { return (size_t)SC_NT * (16 / W) * 8 * W + (size_t)P * (opt ? 20 : 16) + 4 + 17 * 4 + 16 + (opt ? 8 + L0_MAX_PASSES * 8 : 0); }   // opt: + the slice ends (+ the region bases of a level-0 sweep)

template <int W, int SRC, int MODE>
int launch_hist_m(dskgpu_ctx* ctx, const typename KeyT<W>::T* keys, const ChunkDesc* descs, const u32* d_nch,
                  u64 max_chunks, u32* matrix, DigitSpec ds, u32 P) {
    const unsigned grid = (unsigned)std::max<u64>(1, std::min<u64>(max_chunks, (u64)ctx->num_cu));
    hipLaunchKernelGGL((k_hist<W, SRC, MODE>), dim3(grid), dim3(SC_NT), 0, ctx->stream, ctx->packed.as<u64>(),
                       ctx->inval.as<u32>(), keys, descs, d_nch, matrix, (int)ctx->cfg.kmer_size, ds, P);
    CKL("k_hist");
    return DSKGPU_OK;
}
// digit modes in use: reads -> owner (0) or level 1 (1); key array -> level 1 (1, multi-GPU receive side) or level 2 (2)
template <int W, int SRC>
int launch_hist(dskgpu_ctx* ctx, const typename KeyT<W>::T* keys, const ChunkDesc* descs, const u32* d_nch,
                u64 max_chunks, u32* matrix, DigitSpec ds, u32 P) {
    const bool mp = ds.mode == 1 && ds.npass > 1;     // level 1 of a multi-pass count: instantiation with the pass filter
    if (SRC == 0) return ds.mode == 0 ? launch_hist_m<W, 0, 0>(ctx, keys, descs, d_nch, max_chunks, matrix, ds, P)
                       : mp ? launch_hist_m<W, 0, 3>(ctx, keys, descs, d_nch, max_chunks, matrix, ds, P)
                            : launch_hist_m<W, 0, 1>(ctx, keys, descs, d_nch, max_chunks, matrix, ds, P);
    return ds.mode == 2 ? launch_hist_m<W, 1, 2>(ctx, keys, descs, d_nch, max_chunks, matrix, ds, P)
                   : mp ? launch_hist_m<W, 1, 3>(ctx, keys, descs, d_nch, max_chunks, matrix, ds, P)
                        : launch_hist_m<W, 1, 1>(ctx, keys, descs, d_nch, max_chunks, matrix, ds, P);
}

unsigned scatter_grid(const dskgpu_ctx* ctx, int W, u32 P, u64 max_chunks, bool opt = false) {
    const size_t lds = scatter_lds(W, P, opt);
    const u64 per_cu = std::max<u64>(1, std::min<u64>(2048 / SC_NT, (160 * 1024) / lds));   // resident blocks per CU
    return (unsigned)std::max<u64>(1, std::min<u64>(max_chunks, (u64)ctx->num_cu * per_cu));
}
template <int W, int SRC, int MODE, bool OPT = false, bool HEAVY = false>
int launch_scatter_m(dskgpu_ctx* ctx, const typename KeyT<W>::T* keys, const ChunkDesc* descs, const u32* d_nch,
                     u64 max_chunks, const u32* scanned, typename KeyT<W>::T* out, DigitSpec ds, u32 P, Opt1Spec o1 = Opt1Spec{nullptr, 0u, 0u, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, 0u, nullptr, 0ull, {0ull, 0ull, 0ull, 0ull}, 0u}) {
    const size_t lds = scatter_lds(W, P, OPT && !o1.uslice);
    const unsigned grid = scatter_grid(ctx, W, P, max_chunks, OPT && !o1.uslice);
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_scatter<W, SRC, MODE, OPT, HEAVY>)); if (e) return e; }
    hipLaunchKernelGGL((k_scatter<W, SRC, MODE, OPT, HEAVY>), dim3(grid), dim3(SC_NT), lds, ctx->stream, ctx->packed.as<u64>(),
                       ctx->inval.as<u32>(), keys, descs, d_nch, scanned, out, (int)ctx->cfg.kmer_size, ds, P, o1);
    CKL("k_scatter");
    return DSKGPU_OK;
}
// super-k-mer records as the source of the histogram-free level-1 scatter (one- and two-word keys); HEAVY: with k-mers counted apart
// (one-word keys); MODE 3: one of several passes over the records of a multi-GPU receive side (the pass filter on every key)
template <int W, bool HEAVY = false, int MODE = 1>
int launch_scatter_rec(dskgpu_ctx* ctx, const ChunkDesc* descs, const u32* d_nch, u64 max_chunks, typename KeyT<W>::T* out, DigitSpec ds, u32 P, Opt1Spec o1) {
    const size_t lds = scatter_lds(W, P, !o1.uslice);
    const unsigned grid = scatter_grid(ctx, W, P, max_chunks, !o1.uslice);
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_scatter<W, 2, MODE, true, HEAVY>)); if (e) return e; }
    hipLaunchKernelGGL((k_scatter<W, 2, MODE, true, HEAVY>), dim3(grid), dim3(SC_NT), lds, ctx->stream, ctx->rec_src, (const u32*)nullptr,
                       (const typename KeyT<W>::T*)nullptr, descs, d_nch, (const u32*)nullptr, out, (int)ctx->cfg.kmer_size, ds, P, o1);
    CKL("k_scatter(records)");
    return DSKGPU_OK;
}
template <int W>
int launch_scatter_rec_h(dskgpu_ctx* ctx, bool heavy, const ChunkDesc* descs, const u32* d_nch, u64 max_chunks, typename KeyT<W>::T* out, DigitSpec ds, u32 P, Opt1Spec o1) {
    const bool mp = ds.npass > 1;
    if constexpr (W <= 2) {
        if (heavy) return mp ? launch_scatter_rec<W, true, 3>(ctx, descs, d_nch, max_chunks, out, ds, P, o1) : launch_scatter_rec<W, true, 1>(ctx, descs, d_nch, max_chunks, out, ds, P, o1);
    }
    if constexpr (W <= 2) return mp ? launch_scatter_rec<W, false, 3>(ctx, descs, d_nch, max_chunks, out, ds, P, o1) : launch_scatter_rec<W, false, 1>(ctx, descs, d_nch, max_chunks, out, ds, P, o1);
    else return DSKGPU_E_STATE;                     // (records carry k <= 64)
}

// key-array source with aligned write-out (k_scatter_al) when its LDS footprint fits one CU
template <int W, int MODE, bool OPT = false, bool SLICED = false>
int launch_scatter_al(dskgpu_ctx* ctx, const typename KeyT<W>::T* keys, const ChunkDesc* descs, const u32* d_nch,
                      u64 max_chunks, const u32* scanned, typename KeyT<W>::T* out, DigitSpec ds, u32 P, OptSpec os = OptSpec{0u, nullptr, nullptr, nullptr, 0u, 0u, 0ull, 0u, 0u, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}) {
    const size_t lds = ascatter_lds(W, P);
    const unsigned grid = (unsigned)std::max<u64>(1, std::min<u64>(max_chunks, (u64)ctx->num_cu));
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_scatter_al<W, MODE, OPT, SLICED>)); if (e) return e; }
    hipLaunchKernelGGL((k_scatter_al<W, MODE, OPT, SLICED>), dim3(grid), dim3(SC_NT), lds, ctx->stream, keys, descs, d_nch, scanned, out, ds, P, os);
    CKL("k_scatter_al");
    return DSKGPU_OK;
}

template <int W, int SRC>
int launch_scatter(dskgpu_ctx* ctx, const typename KeyT<W>::T* keys, const ChunkDesc* descs, const u32* d_nch,
                   u64 max_chunks, const u32* scanned, typename KeyT<W>::T* out, DigitSpec ds, u32 P) {
    const bool mp = ds.mode == 1 && ds.npass > 1;
    if (SRC == 1 && ascatter_lds(W, P) <= 160 * 1024 && !ctx->tune.no_aligned)
        return ds.mode == 2 ? launch_scatter_al<W, 2>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P)
                       : mp ? launch_scatter_al<W, 3>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P)
                            : launch_scatter_al<W, 1>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P);
    if (SRC == 0) return ds.mode == 0 ? launch_scatter_m<W, 0, 0>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P)
                       : mp ? launch_scatter_m<W, 0, 3>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P)
                            : launch_scatter_m<W, 0, 1>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P);
    return ds.mode == 2 ? launch_scatter_m<W, 1, 2>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P)
                   : mp ? launch_scatter_m<W, 1, 3>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P)
                        : launch_scatter_m<W, 1, 1>(ctx, keys, descs, d_nch, max_chunks, scanned, out, ds, P);
}

// fixed-capacity regions or exact offsets: a compile-time switch of the count kernels (k_count1 / k_count_mw)
inline void launch_count_impl(dskgpu_ctx* ctx, unsigned grid, u64* keys, u64* solid_keys, u32* solid_ab, u32* ovf, const CountParams& cp) {
    if (cp.cap && cp.cap <= CNT_V3_KEYS * CNT_NT) {      // regions: the list-free kernel
        hipLaunchKernelGGL((k_count1v3<CNT_NT, CNT_KPT, CNT_V3_KEYS>), dim3(grid), dim3(CNT_NT), 0, ctx->stream, keys, solid_keys, solid_ab,
                           ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), ovf, cp, cp.subcnt);
        return;
    }
    if (cp.cap) hipLaunchKernelGGL(k_count1<true>, dim3(grid), dim3(CNT_NT), 0, ctx->stream, keys, solid_keys, ctx->fstart.as<u32>(), solid_ab,
                                   ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), ovf, cp, cp.subcnt);
    else hipLaunchKernelGGL(k_count1<false>, dim3(grid), dim3(CNT_NT), 0, ctx->stream, keys, solid_keys, ctx->fstart.as<u32>(), solid_ab,
                            ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), ovf, cp, (const u32*)ctx->fstart.as<u32>());
}
template <int W>
inline void launch_count_impl(dskgpu_ctx* ctx, unsigned grid, KN<W>* keys, KN<W>* solid_keys, u32* solid_ab, u32* ovf, const CountParams& cp) {
    if constexpr (W == 2) {
        if (cp.cap && cp.cap <= C2V_NKEYS * CNT_NT && !ctx->mw_v3_off) {      // regions: the table keyed by the mixed top word
            CountParams c2 = cp; c2.maxload = std::min<u32>(cp.maxload, C2V_MAXLOAD);
            hipLaunchKernelGGL((k_count2v3<CNT_NT, C2V_KPT, C2V_NKEYS>), dim3(grid), dim3(CNT_NT), 0, ctx->stream, (const K2*)keys, solid_keys, solid_ab,
                               ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), ovf, c2, cp.subcnt);
            return;
        }
    }
    if (cp.cap) hipLaunchKernelGGL((k_count_mw<W, true>), dim3(grid), dim3(CNT_NT), 0, ctx->stream, keys, solid_keys, ctx->fstart.as<u32>(), solid_ab,
                                   ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), ovf, cp);
    else hipLaunchKernelGGL((k_count_mw<W, false>), dim3(grid), dim3(CNT_NT), 0, ctx->stream, keys, solid_keys, ctx->fstart.as<u32>(), solid_ab,
                            ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), ovf, cp);
}
template <int W>
inline void launch_count(dskgpu_ctx* ctx, unsigned grid, typename KeyT<W>::T* keys, typename KeyT<W>::T* solid_keys, u32* solid_ab, u32* ovf, const CountParams& cp) {
    launch_count_impl(ctx, grid, keys, solid_keys, solid_ab, ovf, cp);
}

struct Plan {
    int levels;
    u32 P1, P2, F;
    DigitSpec d1, d2;
};

#define TARGET_KEYS 2900      // mean keys per final sub-partition (the count kernel prefetches 3072 per sub-partition; table: 4096 slots, 3584 usable)
#define TARGET_KEYS2 2560     // two-word keys: 3072-slot top-word table (k_count2v3), 2688 usable; 4096-slot index table (k_count_mw)
#ifndef TARGET_KEYS4
#define TARGET_KEYS4 640      // four-word keys: 1024 staged per sub-partition
#endif
#ifndef OPT_GROUPS4
#define OPT_GROUPS4 OPT_GROUPS
#endif
#define MAX_LEVEL_BINS 2048
#define ONE_LEVEL_BINS 1024
#define CH2 65536u            // keys per level-2 chunk
#define OPT_GROUPS 545u
#define OPT_GROUPS2 1023u     // two-word keys: regions of 1023 groups of 64 B = 4092 keys (the count kernels are paced by their two barriers per
                              // sub-partition, not by its keys: half as many sub-partitions of twice the size -- see NOTEBOOK.md section 6 "k = 63")
inline u32 opt_groups(int W) { return W == 2 ? (AL_G2 == 8 ? 1022u : OPT_GROUPS2) : (W == 1 && AL_G1 == 16) ? 546u : W == 4 ? OPT_GROUPS4 : OPT_GROUPS; }      // (AL_G2 == 8, experiments: 511 groups of 128 B)
inline u64 target_keys(int W) { return W == 1 ? TARGET_KEYS : W == 2 ? TARGET_KEYS2 : TARGET_KEYS4; }
#define OPT_CAP 4360u          // segment-owned level-2 scatter: keys per sub-partition region (mean <= TARGET_KEYS).
                              // 545 groups of 64 B -- an ODD number, so the region starts (and the write fronts that advance
                              // through all regions in step) spread over every HBM channel instead of camping on a few

// Final sub-partitions F = P1 * P2 sized to the input (any integer, not a power
// of two: digits use the multiply-shift reduction of key_digit()).
bool make_plan(u64 n_upper, int extra_bits, int W, u32 num_cu, Plan* pl) {
    const u64 target = target_keys(W);
    u64 F = ((n_upper + target - 1) / target) << extra_bits;
    if (F < 2) F = 2;
    if (F <= ONE_LEVEL_BINS) { pl->levels = 1; pl->P1 = (u32)F; pl->P2 = 1; }
    else {
        // Level 1 costs more per key the more bins it has (3.05 ps + 1.7 fs per bin and key, NOTEBOOK.md section 6: every tile leaves a
        // partial line per bin), level 2 costs the same for any number of bins its kernel holds (the per-bin carry must fit LDS:
        // p2max).  So: the FEWEST level-1 bins that keep level 2 inside that kernel -- but at least one segment per CU, and a
        // multiple of the CU count: level 2 gives every block whole segments (level-1 bins), one block per CU, and a count that is
        // not a multiple leaves some blocks one segment more than the rest (770 segments on 256 CUs cost 20 %).
        u64 p2max = MAX_LEVEL_BINS; while (p2max > 64 && ascatter_lds(W, (u32)p2max) > 160 * 1024) --p2max;
        u64 p1 = 1; while (p1 * p1 < F) ++p1;                                  // balanced split: small inputs
        const bool large = F >= (u64)num_cu * 64;
        if (large) {
            p1 = std::max<u64>((F + p2max - 2) / (p2max - 1), num_cu);         // (p2max - 1: room for the odd-P2 adjustment below)
            p1 = (p1 + num_cu - 1) / num_cu * num_cu;
            if (p1 > MAX_LEVEL_BINS - 8) p1 = MAX_LEVEL_BINS - 8;
        }
        u64 p2 = (F + p1 - 1) / p1;
        // An ODD number of level-2 bins: the regions of sub-partition q start q * 545 groups of 64 B into the buffer, and the
        // blocks walk their segments in step, so with an even P2 (768 * 545 * 64 B = a multiple of 8 KB between segments) all
        // write fronts sit on the same few HBM channels -- measured 6.0 ms at P2 = 768 against 4.7-5.0 ms at P2 = 769.
        if (p2 % 2 == 0 && p2 + 1 <= p2max) ++p2;
        if (p1 > MAX_LEVEL_BINS || p2 > MAX_LEVEL_BINS) return false;
        pl->levels = 2; pl->P1 = (u32)p1; pl->P2 = (u32)p2;
    }
    pl->F = pl->P1 * pl->P2;
    pl->d1 = DigitSpec{1u, pl->P1, 0u, 1u, 1u, 0u};
    pl->d2 = DigitSpec{2u, pl->P1, pl->P2, 1u, 1u, 0u};
    return true;
}

// Build level-1 chunk descriptors on the host (ranges are static).
// unit = tile granularity of the source (Tile<W>::WORDS words or Tile<W>::KEYS keys).
void build_descs1(dskgpu_ctx* ctx, u64 n_units_total, u64 tile, u64 max_chunks, u32* nch_out) {
    const u64 ntiles = std::max<u64>(1, (n_units_total + tile - 1) / tile);
    u64 nch = std::min<u64>(ntiles, max_chunks);
    const u64 tpc = (ntiles + nch - 1) / nch;
    nch = (ntiles + tpc - 1) / tpc;
    ctx->h_descs1.resize(nch);
    for (u64 c = 0; c < nch; ++c) {
        ChunkDesc d;
        d.begin = c * tpc * tile;
        d.end = std::min<u64>(n_units_total, (c + 1) * tpc * tile);
        if (d.begin > d.end) d.begin = d.end;
        d.flat_base = (u32)c;
        d.stride = (u32)nch;
        ctx->h_descs1[c] = d;
    }
    *nch_out = (u32)nch;
}

// the same for records that arrive in slices (ctx->rec_slice_end): every slice gets its own chunks (a launch per slice walks
// them); h_slice_chunk[s] = first chunk of slice s
void build_descs1_slices(dskgpu_ctx* ctx, u64 tile, u64 max_chunks, u32* nch_out) {
    const size_t S = ctx->rec_slice_end.size();
    ctx->h_descs1.clear(); ctx->h_slice_chunk.assign(S + 1, 0);
    const u64 per_slice = std::max<u64>(1, max_chunks / S);
    u64 rb = 0;
    for (size_t sl = 0; sl < S; ++sl) {
        const u64 re = ctx->rec_slice_end[sl];
        ctx->h_slice_chunk[sl] = (u32)ctx->h_descs1.size();
        if (re > rb) {
            const u64 ntiles = (re - rb + tile - 1) / tile;
            u64 nch = std::min<u64>(ntiles, per_slice);
            const u64 tpc = (ntiles + nch - 1) / nch;
            nch = (ntiles + tpc - 1) / tpc;
            for (u64 c = 0; c < nch; ++c) {
                ChunkDesc d;
                d.begin = rb + c * tpc * tile; d.end = std::min<u64>(re, rb + (c + 1) * tpc * tile);
                d.flat_base = 0; d.stride = 0;                      // (only the histogram-free scatter reads these chunks)
                ctx->h_descs1.push_back(d);
            }
        }
        rb = re;
    }
    ctx->h_slice_chunk[S] = (u32)ctx->h_descs1.size();
    for (auto& d : ctx->h_descs1) d.stride = (u32)ctx->h_descs1.size();
    *nch_out = (u32)ctx->h_descs1.size();
}

}  // namespace

namespace {

// Index permutation that sorts n rows of W words (struct-of-arrays in `rows`) ascending: W stable 64-bit
// radix passes, least significant word first.  Result in ctx->srt_idx.
int sort_index_multiword(dskgpu_ctx* ctx, const u64* const* rows, u64 n, int W);
int sort_index_multiword(dskgpu_ctx* ctx, DevBuf* rows, u64 n, int W) {
    const u64* p[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int x = 0; x < W; ++x) p[x] = rows[x].as<u64>();
    return sort_index_multiword(ctx, p, n, W);
}
int sort_index_multiword(dskgpu_ctx* ctx, const u64* const* rows, u64 n, int W) {
    CK(ctx->srt_k.ensure(n * 8));
    CK(ctx->srt_idx.ensure(n * 4));
    CK(ctx->srt_idx2.ensure(n * 4));
    DevBuf& keys_sorted = ctx->s_val;           // scratch for the sorted keys of a pass (not needed afterwards)
    CK(keys_sorted.ensure(n * 8));
    const unsigned gb = (unsigned)((n + 255) / 256);
    const unsigned top_bits = std::min(64u, std::max(1u, 2u * ctx->cfg.kmer_size - 64u * (unsigned)(W - 1)));
    u32* idx = ctx->srt_idx.as<u32>(); u32* idx2 = ctx->srt_idx2.as<u32>();
    hipLaunchKernelGGL(k_iota, dim3(gb), dim3(256), 0, ctx->stream, idx, n);
    size_t tmp = 0, tmp2 = 0;
    // LIBRARY SORT (rocprim), labelled FALLBACK: index permutation of multi-word rows in full-width order (W stable passes), behind a raised flag only
    CK(rocprim::radix_sort_pairs(nullptr, tmp, ctx->srt_k.as<u64>(), keys_sorted.as<u64>(), idx, idx2, (size_t)n, 0u, 64u, ctx->stream));
    CK(rocprim::radix_sort_pairs(nullptr, tmp2, ctx->srt_k.as<u64>(), keys_sorted.as<u64>(), idx, idx2, (size_t)n, 0u, top_bits, ctx->stream));
    CK(ctx->srt_tmp.ensure(std::max(tmp, tmp2)));
    for (int x = 0; x < W; ++x) {
        const u64* src = rows[x];
        const unsigned bits = x == W - 1 ? top_bits : 64u;
        if (x == 0) {
            CK(rocprim::radix_sort_pairs(ctx->srt_tmp.p, tmp, src, keys_sorted.as<u64>(), idx, idx2, (size_t)n, 0u, bits, ctx->stream));
        } else {
            hipLaunchKernelGGL(k_gather<u64>, dim3(gb), dim3(256), 0, ctx->stream, ctx->srt_k.as<u64>(), src, idx, n);
            size_t t = x == W - 1 ? tmp2 : tmp;
            CK(rocprim::radix_sort_pairs(ctx->srt_tmp.p, t, ctx->srt_k.as<u64>(), keys_sorted.as<u64>(), idx, idx2, (size_t)n, 0u, bits, ctx->stream));
        }
        std::swap(idx, idx2);
    }
    if (idx != ctx->srt_idx.as<u32>()) std::swap(ctx->srt_idx, ctx->srt_idx2);
    CKL("sort_index_multiword");
    return DSKGPU_OK;
}

// full-width order of multi-word rows: W stable radix passes over an index permutation, then gathers
int sort_rows_full_multiword(dskgpu_ctx* ctx, u64 n) {
    const int W = ctx->W;
    int rc = sort_index_multiword(ctx, ctx->out_w, n, W);
    if (rc) return rc;
    const unsigned gb = (unsigned)((n + 255) / 256);
    const u32* idx = ctx->srt_idx.as<u32>();
    CK(ctx->srt_ab.ensure(n * 4));
    for (int x = 0; x < W; ++x) {
        CK(ctx->srt_w[x].ensure(n * 8));
        hipLaunchKernelGGL(k_gather<u64>, dim3(gb), dim3(256), 0, ctx->stream, ctx->srt_w[x].as<u64>(), ctx->out_w[x].as<u64>(), idx, n);
        ctx->res_w[x] = ctx->srt_w[x].as<u64>();
    }
    hipLaunchKernelGGL(k_gather<u32>, dim3(gb), dim3(256), 0, ctx->stream, ctx->srt_ab.as<u32>(), ctx->out_ab.as<u32>(), idx, n);
    CKL("sort_rows");
    ctx->res_ab = ctx->srt_ab.as<u32>();
    return DSKGPU_OK;
}

// ---- hand-written MSD radix sort (rowsort.h) of n (64-bit key, 32-bit value) pairs on the `total` low bits of the key, in place
// (tk / tv: scratch of the same size).  One-word rows: (k-mer value, abundance); multi-word rows: (top 63 bits of the value,
// row index).  Whatever the kernels do not order themselves raises SC_SORTFLAG (zeroed here): the caller falls back to a
// full-width library sort (k / v and tk / tv each hold a complete permutation of the pairs either way).
int msd_sort_pairs(dskgpu_ctx* ctx, u64* k, u32* v, u64* tk, u32* tv, u64 n, int total, bool reset_flags = true, u32 base = 0, const dskgpu_ctx::SparseRows* spr = nullptr) {
    // second digit: 8 bits up to 96 M rows, 9 up to 192 M, 10 beyond (sub-buckets stay near 200 rows: one wave each in step C)
    int wantB = n <= (96ull << 20) ? 8 : n <= (192ull << 20) ? 9 : 10;
    if (ctx->tune.rs_bbits >= 8 && ctx->tune.rs_bbits <= 10) wantB = (int)ctx->tune.rs_bbits;      // tests
    const int bA = std::min(RS_ABITS, total), r1 = total - bA, bB = std::min(wantB, r1), r2 = r1 - bB, bC = std::min(8, r2), r3 = r2 - bC;
    const u32 BB = 1u << wantB;
    RsSpec sp{r1, r2, r3, (1u << bA) - 1u, (1u << bB) - 1u, (1u << bC) - 1u};
    // chunks of step A: about 64 K rows each, a multiple of the CU count of them (whole rounds of blocks), at least one tile each
    const u64 ncu = (u64)ctx->num_cu;
    u64 nch = (n + 65535) / 65536;
    nch = (nch + ncu - 1) / ncu * ncu;
    nch = std::max<u64>(1, std::min<u64>(nch, (n + RS_TILE - 1) / RS_TILE));
    u64 chunk = (n + nch - 1) / nch;
    nch = (n + chunk - 1) / chunk;
    // sparse source (spr: the rows still lie in the count kernel's regions): chunks = groups of qpc consecutive sub-partitions (about 64 K
    // rows, at most RS_SP_MAXQ sub-partitions), + one chunk for the dense tail (the rows of the k-mers counted apart)
    RsSparse sps{}; u64 nch_sp = 0;
    if (spr) {
        sps = spr->s;
        const u64 F = sps.F;
        u64 want = std::max<u64>(1, (spr->n_sparse + 65535) / 65536);
        want = (want + ncu - 1) / ncu * ncu;
        u64 qpc = std::max<u64>(1, (F + want - 1) / want);
        if (qpc > RS_SP_MAXQ) qpc = RS_SP_MAXQ;
        nch_sp = (F + qpc - 1) / qpc;
        sps.qpc = (u32)qpc;
        nch = nch_sp + (spr->n_tail ? 1 : 0);
        chunk = spr->n_tail;                                              // (the tail is one chunk)
    }
    const u64 M = (u64)RS_ABINS * nch;
    if (M >= 0xFFFFFFF0ull) return fail(ctx, DSKGPU_E_ARG, "row sort: chunk matrix too large");
    const u64 nsubw = (u64)RS_ABINS * (BB + 1);                           // sub-bucket starts; behind them the list of large sub-buckets
    CK(ctx->srt_tmp.ensure((M + 2 + 2 * nsubw + 16) * 4));
    u32* matrix = static_cast<u32*>(ctx->srt_tmp.p);
    u32* sub = matrix + M + 2;
    u32* biglist = sub + nsubw;
    u32* sc = ctx->scalars.as<u32>();
    CK(ctx->rs_ovs.ensure((1 + 3 * RS_OVS_CAP) * 4));
    hipLaunchKernelGGL(k_set_rs_scalars, dim3(1), dim3(64), 0, ctx->stream, sc + SC_RSLEN, (u32)M, reset_flags ? 4u : 3u,      // (length, two work counters [, ties seen])
                       reset_flags ? sc + SC_SORTFLAG : (u32*)nullptr, reset_flags ? ctx->rs_ovs.as<u32>() : (u32*)nullptr);
    const size_t ldsA = RsLds<RS_ABINS, RS_TILE>::bytes;
    const size_t ldsB = BB == 256 ? RsLds<256, RS_BTILE>::bytes : BB == 512 ? RsLds<512, RS_BTILE>::bytes : RsLds<1024, RS_BTILE>::bytes;
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_rs_scatter<false>)); if (e) return e; }
    if (spr) {
        const size_t ldsS = ldsA + ((size_t)RS_SP_MAXQ + 1) * 4;
        { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_rs_scatter_sp)); if (e) return e; }
        hipLaunchKernelGGL(k_rs_hist_sp, dim3((unsigned)nch_sp), dim3(RS_NT), 0, ctx->stream, sps, (u32)nch, matrix, sp);
        if (spr->n_tail) hipLaunchKernelGGL(k_rs_hist, dim3(1), dim3(RS_NT), 0, ctx->stream, spr->tail_k, (u64)spr->n_tail, (u32)chunk, (u32)nch, matrix, sp, (u32)nch_sp);
        CKL("k_rs_hist_sp");
        { const int e = run_scan(ctx, matrix, sc + SC_RSLEN, M); if (e) return e; }
        hipLaunchKernelGGL(k_rs_scatter_sp, dim3((unsigned)nch_sp), dim3(RS_NT), ldsS, ctx->stream, sps, (u32)nch, (const u32*)matrix, tk, tv, sp);
        if (spr->n_tail) hipLaunchKernelGGL(k_rs_scatter<false>, dim3(1), dim3(RS_NT), ldsA, ctx->stream, spr->tail_k, spr->tail_v, (u64)spr->n_tail, (u32)chunk, (u32)nch, (const u32*)matrix, tk, tv, sp, (const u64*)nullptr, (u32)nch_sp);
        CKL("k_rs_scatter_sp");
    } else {
        hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)nch), dim3(RS_NT), 0, ctx->stream, k, n, (u32)chunk, (u32)nch, matrix, sp, 0u);
        CKL("k_rs_hist");
        { const int e = run_scan(ctx, matrix, sc + SC_RSLEN, M); if (e) return e; }
        hipLaunchKernelGGL(k_rs_scatter<false>, dim3((unsigned)nch), dim3(RS_NT), ldsA, ctx->stream, k, v, n, (u32)chunk, (u32)nch, matrix, tk, tv, sp, (const u64*)nullptr, 0u);
        CKL("k_rs_scatter");
    }
    // a bucket above 64 x the mean (+ 256 K rows) is not a k-mer spectrum any more (canonical k-mers: at most ~2 x; a low-complexity stretch of
    // 200 kb puts 180 K rows under AAAAA: that is still one block's 0.2 ms -- the limit was 16 x + 64 K until seeds 208 / 292 / 319 of
    // tools/stress_random.py took the 12 ms library fallback for it): one block would
    // walk it alone, so it goes to the full-width fallback instead
    u32 heavy = (u32)std::min<u64>(0xFFFFFFFFull, n / RS_ABINS * 64 + 262144);
    if (ctx->tune.rs_heavy) heavy = ctx->tune.rs_heavy;
    const unsigned gridB = (unsigned)std::min<u64>(ncu * (160 * 1024 / (ldsB + 1024)), RS_ABINS);
    auto split = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(gridB), dim3(RS_BNT), ldsB, ctx->stream, tk, tv, (u32)nch, matrix, k, v, sub, sp, sc + SC_RSWORK2, heavy, sc + SC_SORTFLAG);
    };
    if (BB == 1024 && ldsB > 64 * 1024) { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_rs_split<1024>)); if (e) return e; }
    if (BB == 256) split(k_rs_split<256>); else if (BB == 512) split(k_rs_split<512>); else split(k_rs_split<1024>);
    CKL("k_rs_split");
    const u32 nsub = RS_ABINS * BB;
    hipLaunchKernelGGL(k_rs_cells, dim3(nsub / (RS_CNT / 64)), dim3(RS_CNT), 0, ctx->stream, k, v, sub, nsub, BB, sp, biglist, sc + SC_RSWORK, sc + SC_SORTFLAG, sc + SC_RSTIES);
    CKL("k_rs_cells");
    const u32 block_rows = ctx->tune.rs_block_rows ? std::min<u32>(ctx->tune.rs_block_rows, RS_BLOCK_ROWS) : RS_BLOCK_ROWS;
    hipLaunchKernelGGL(k_rs_big, dim3((unsigned)std::min<u64>(ncu, 256)), dim3(RS_NT), 0, ctx->stream, k, v, sub, sp, biglist, sc + SC_RSWORK, sc + SC_SORTFLAG, block_rows, sc + SC_RSTIES, base, ctx->rs_ovs.as<u32>());
    CKL("k_rs_big");
    return DSKGPU_OK;
}

// ---- the same sort for two-word rows (rowsort2.h): the rows themselves -- (hi, lo, abundance) in three arrays -- ordered in place on
// the `total` low bits of hi:lo (t: scratch of the same size).  What it lists (sub-buckets above RS_BLOCK_ROWS rows) is left in
// ctx->rs_ovs for the caller's next round; a heavy first-digit bucket or a full list raises SC_SORTFLAG.  k holds a complete
// permutation of the rows either way.
// spr: step A reads the rows in the count kernel's regions (+ a dense tail: the rows of the k-mers counted apart), as in msd_sort_pairs
int msd_sort_rows2(dskgpu_ctx* ctx, Rows2 k, Rows2 t, u64 n, int total, bool reset_flags, u32 base, const dskgpu_ctx::SparseRows2* spr = nullptr) {
    int wantB = n <= (96ull << 20) ? 8 : n <= (192ull << 20) ? 9 : 10;
    if (ctx->tune.rs_bbits >= 8 && ctx->tune.rs_bbits <= 10) wantB = (int)ctx->tune.rs_bbits;      // tests
    const int bA = std::min(RS2_ABITS, total), r1 = total - bA, bB = std::min(wantB, r1), r2 = r1 - bB, bC = std::min(8, r2), r3 = r2 - bC;
    const u32 BB = 1u << wantB;
    RsSpec sp{r1, r2, r3, (1u << bA) - 1u, (1u << bB) - 1u, (1u << bC) - 1u};
    const u64 ncu = (u64)ctx->num_cu;
    u64 nch = (n + 65535) / 65536;
    nch = (nch + ncu - 1) / ncu * ncu;
    nch = std::max<u64>(1, std::min<u64>(nch, (n + RS2_TILE - 1) / RS2_TILE));
    u64 chunk = (n + nch - 1) / nch;
    nch = (n + chunk - 1) / chunk;
    Rs2Sparse sps{}; u64 nch_sp = 0;
    if (spr) {      // chunks = groups of qpc consecutive sub-partitions (about 64 K rows, at most RS_SP_MAXQ of them) + one chunk for the dense tail
        sps = spr->s;
        const u64 F = sps.F;
        u64 want = std::max<u64>(1, (spr->n_sparse + 65535) / 65536);
        want = (want + ncu - 1) / ncu * ncu;
        u64 qpc = std::max<u64>(1, (F + want - 1) / want);
        if (qpc > RS_SP_MAXQ) qpc = RS_SP_MAXQ;
        nch_sp = (F + qpc - 1) / qpc;
        sps.qpc = (u32)qpc;
        nch = nch_sp + (spr->n_tail ? 1 : 0);
        chunk = spr->n_tail;
    }
    const u64 M = (u64)RS2_ABINS * nch;
    const u64 nsubw = (u64)RS2_ABINS * (BB + 1);
    CK(ctx->srt_tmp.ensure((M + 2 + 2 * nsubw + 16) * 4));
    u32* matrix = static_cast<u32*>(ctx->srt_tmp.p);
    u32* sub = matrix + M + 2;
    u32* biglist = sub + nsubw;
    u32* sc = ctx->scalars.as<u32>();
    CK(ctx->rs_ovs.ensure((1 + 3 * RS_OVS_CAP) * 4));
    hipLaunchKernelGGL(k_set_rs_scalars, dim3(1), dim3(64), 0, ctx->stream, sc + SC_RSLEN, (u32)M, reset_flags ? 4u : 3u,
                       reset_flags ? sc + SC_SORTFLAG : (u32*)nullptr, reset_flags ? ctx->rs_ovs.as<u32>() : (u32*)nullptr);
    const size_t ldsA = Rs2Lds<RS2_ABINS, RS2_TILE>::bytes;
    const size_t ldsB = BB == 256 ? Rs2Lds<256, RS2_BTILE>::bytes : BB == 512 ? Rs2Lds<512, RS2_BTILE>::bytes : Rs2Lds<1024, RS2_BTILE>::bytes;
    const size_t ldsBig = (size_t)RS_BLOCK_ROWS * 20 + 2 * RS_CELLS * 4;
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k2_scatter<false>)); if (e) return e; }
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k2_big), (int)ldsBig); if (e) return e; }
    const Rows2C kc{k.hi, k.lo, k.ab}, tc{t.hi, t.lo, t.ab};
    if (spr) {
        const size_t ldsS = ldsA + ((size_t)RS_SP_MAXQ + 1) * 4;
        { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k2_scatter_sp)); if (e) return e; }
        hipLaunchKernelGGL(k2_hist_sp, dim3((unsigned)nch_sp), dim3(RS_NT), 0, ctx->stream, sps, (u32)nch, matrix, sp);
        if (spr->n_tail) hipLaunchKernelGGL(k2_hist, dim3(1), dim3(RS_NT), 0, ctx->stream, spr->tail, (u64)spr->n_tail, (u32)chunk, (u32)nch, matrix, sp, (u32)nch_sp);
        CKL("k2_hist_sp");
        { const int e = run_scan(ctx, matrix, sc + SC_RSLEN, M); if (e) return e; }
        hipLaunchKernelGGL(k2_scatter_sp, dim3((unsigned)nch_sp), dim3(RS_NT), ldsS, ctx->stream, sps, (u32)nch, (const u32*)matrix, t, sp);
        if (spr->n_tail) hipLaunchKernelGGL(k2_scatter<false>, dim3(1), dim3(RS_NT), ldsA, ctx->stream, spr->tail, (u64)spr->n_tail, (u32)chunk, (u32)nch, matrix, t, sp, (const u64*)nullptr, (u32)nch_sp);
        CKL("k2_scatter_sp");
    } else {
        hipLaunchKernelGGL(k2_hist, dim3((unsigned)nch), dim3(RS_NT), 0, ctx->stream, kc, n, (u32)chunk, (u32)nch, matrix, sp, 0u);
        CKL("k2_hist");
        { const int e = run_scan(ctx, matrix, sc + SC_RSLEN, M); if (e) return e; }
        hipLaunchKernelGGL(k2_scatter<false>, dim3((unsigned)nch), dim3(RS_NT), ldsA, ctx->stream, kc, n, (u32)chunk, (u32)nch, matrix, t, sp, (const u64*)nullptr, 0u);
        CKL("k2_scatter");
    }
    u32 heavy = (u32)std::min<u64>(0xFFFFFFFFull, n / RS2_ABINS * 64 + 262144);
    if (ctx->tune.rs_heavy) heavy = ctx->tune.rs_heavy;
    const unsigned gridB = (unsigned)std::min<u64>(ncu * (160 * 1024 / (ldsB + 1024)), RS2_ABINS);
    auto split = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(gridB), dim3(RS_BNT), ldsB, ctx->stream, tc, (u32)nch, matrix, k, sub, sp, sc + SC_RSWORK2, heavy, sc + SC_SORTFLAG);
    };
    if (BB == 256) split(k2_split<256>); else if (BB == 512) split(k2_split<512>); else split(k2_split<1024>);
    CKL("k2_split");
    const u32 nsub = RS2_ABINS * BB;
    hipLaunchKernelGGL(k2_cells, dim3(nsub / (RS_CNT / 64)), dim3(RS_CNT), 0, ctx->stream, k, sub, nsub, BB, sp, biglist, sc + SC_RSWORK);
    CKL("k2_cells");
    const u32 block_rows = ctx->tune.rs_block_rows ? std::min<u32>(ctx->tune.rs_block_rows, RS_BLOCK_ROWS) : RS_BLOCK_ROWS;
    hipLaunchKernelGGL(k2_big, dim3((unsigned)std::min<u64>(ncu, 256)), dim3(RS_NT), ldsBig, ctx->stream, k, sub, sp, biglist, sc + SC_RSWORK, sc + SC_SORTFLAG, block_rows, base, ctx->rs_ovs.as<u32>());
    CKL("k2_big");
    return DSKGPU_OK;
}

// the sub-buckets a two-word sort listed (ctx->rs_ovs; offsets are absolute rows of R, the array that holds the result; S = scratch of
// the same shape): every listed range goes round again on the bits it has not used, until nothing is listed any more
int rows2_rounds(dskgpu_ctx* ctx, Rows2 R, Rows2 S) {
    u32* sc = ctx->scalars.as<u32>();
    std::vector<u32> list;
    for (int round = 0; ; ++round) {
        u32 cnt = 0;
        CK(hipMemcpyAsync(&ctx->h_back[3], sc + SC_SORTFLAG, 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipMemcpyAsync(&cnt, ctx->rs_ovs.p, 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        if (ctx->h_back[3] || cnt == 0) break;
        if (cnt > RS2_OVS_CAP || round >= 8) { ctx->h_back[3] = 1; CK(hipMemsetAsync(sc + SC_SORTFLAG, 0xFF, 4, ctx->stream)); break; }
        list.resize(3 * (size_t)cnt);
        CK(hipMemcpyAsync(list.data(), ctx->rs_ovs.as<u32>() + 1, list.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        CK(hipMemsetAsync(ctx->rs_ovs.p, 0, 4, ctx->stream));
        if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] two-word row sort, round %d: %u sub-bucket(s) above %u rows go round again (first: %u rows, %u bits left)\n", round + 1, cnt, (u32)RS_BLOCK_ROWS, list[1], list[2]);
        for (u32 r = 0; r < cnt; ++r) {
            const u64 off = list[3 * r]; const u32 len = list[3 * r + 1], bits = list[3 * r + 2];
            const Rows2 k2{R.hi + off, R.lo + off, R.ab + off}, t2{S.hi + off, S.lo + off, S.ab + off};
            const int e = msd_sort_rows2(ctx, k2, t2, len, (int)bits, false, (u32)off);
            if (e) return e;
        }
    }
    return DSKGPU_OK;
}

// two-word rows, <= RS_MAX_ROWS: out_* ordered in place; the sub-buckets the sort lists go round again on their remaining bits,
// range by range (a handful on real reads; each round consumes 26-28 bits: at most ceil(128 / 18) rounds).  Leaves the flag
// read-back in flight like the other sorts (ctx->h_back[3] != 0 after the caller's sync: the full-width fallback, which reads out_*)
int sort_rows2_msd(dskgpu_ctx* ctx, u64 n) {
    for (int x = 0; x < 2; ++x) CK(ctx->srt_w[x].ensure(n * 8));
    CK(ctx->srt_ab.ensure(n * 4));
    const Rows2 K{ctx->out_w[1].as<u64>(), ctx->out_w[0].as<u64>(), ctx->out_ab.as<u32>()};
    const Rows2 T{ctx->srt_w[1].as<u64>(), ctx->srt_w[0].as<u64>(), ctx->srt_ab.as<u32>()};
    u32* sc = ctx->scalars.as<u32>();
    { const int e = msd_sort_rows2(ctx, K, T, n, 2 * (int)ctx->cfg.kmer_size, true, 0u, ctx->sp_rows2.valid ? &ctx->sp_rows2 : nullptr); ctx->sp_rows2.valid = false; if (e) return e; }
    { const int e = rows2_rounds(ctx, K, T); if (e) return e; }
    (void)sc;
    ctx->h_ovs.assign(1, 0);
    ctx->sort_back = 2;
    ctx->sort_partial = true;
    return DSKGPU_OK;
}

// two-word row sets above RS_MAX_ROWS (all of configs[3]'s volume on one GPU: 7 * 10^8 rows): sort_rows_big's scheme with the
// rowsort2.h kernels -- step A once over all rows (out_* -> scratch), then the groups of 10-bit buckets that share their top `sb` bits
// are ordered one by one (in the scratch copy, out_* their scratch), then the listed sub-buckets' rounds.  Result: ctx->res_* = the
// scratch copy; on a raised flag the caller copies it back to out_* for the full-width fallback (ctx->rows2_in_scratch).
int sort_rows2_big(dskgpu_ctx* ctx, u64 n) {
    const Rows2 K{ctx->out_w[1].as<u64>(), ctx->out_w[0].as<u64>(), ctx->out_ab.as<u32>()};
    const size_t n_al = (size_t)((n + 31) & ~(u64)31), need = n_al * 20 + 256;
    Rows2 T;
    if (ctx->l0buf.cap >= need) { T.hi = ctx->l0buf.as<u64>(); T.lo = T.hi + n_al; T.ab = reinterpret_cast<u32*>(T.lo + n_al); }
    else {
        size_t free_b = 0, total_b = 0;
        CK(hipMemGetInfo(&free_b, &total_b));
        if (ctx->srt_w[0].cap + ctx->srt_w[1].cap + ctx->srt_ab.cap + free_b < need + ((size_t)2 << 30)) { ctx->l0buf.release(); ctx->bufA.release(); ctx->bufB.release(); }
        for (int x = 0; x < 2; ++x) CK(ctx->srt_w[x].ensure(n * 8));
        CK(ctx->srt_ab.ensure(n * 4));
        T = Rows2{ctx->srt_w[1].as<u64>(), ctx->srt_w[0].as<u64>(), ctx->srt_ab.as<u32>()};
    }
    const int total = 2 * (int)ctx->cfg.kmer_size;
    const int bA = std::min(RS2_ABITS, total);
    RsSpec sp{total - bA, 0, 0, (1u << bA) - 1u, 0u, 0u};
    const u64 ncu = (u64)ctx->num_cu;
    u64 nch = (n + 65535) / 65536;
    nch = (nch + ncu - 1) / ncu * ncu;
    const u64 chunk = (n + nch - 1) / nch;
    nch = (n + chunk - 1) / chunk;
    const u64 M = (u64)RS2_ABINS * nch;
    if (M >= 0xFFFFFFF0ull) return fail(ctx, DSKGPU_E_ARG, "row sort: too many rows");
    CK(ctx->mat2.ensure((M + 2) * 4));
    u32* matrix = ctx->mat2.as<u32>();
    u32* sc = ctx->scalars.as<u32>();
    CK(ctx->rs_ovs.ensure((1 + 3 * RS_OVS_CAP) * 4));
    hipLaunchKernelGGL(k_set_rs_scalars, dim3(1), dim3(64), 0, ctx->stream, sc + SC_RSLEN, (u32)M, 4u, sc + SC_SORTFLAG, ctx->rs_ovs.as<u32>());
    ctx->h_ovs.assign(1, 0);
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k2_scatter<false>)); if (e) return e; }
    const Rows2C kc{K.hi, K.lo, K.ab};
    hipLaunchKernelGGL(k2_hist, dim3((unsigned)nch), dim3(RS_NT), 0, ctx->stream, kc, n, (u32)chunk, (u32)nch, matrix, sp);
    CKL("k2_hist");
    { const int e = run_scan(ctx, matrix, sc + SC_RSLEN, M); if (e) return e; }
    const size_t ldsA = Rs2Lds<RS2_ABINS, RS2_TILE>::bytes;
    hipLaunchKernelGGL(k2_scatter<false>, dim3((unsigned)nch), dim3(RS_NT), ldsA, ctx->stream, kc, n, (u32)chunk, (u32)nch, matrix, T, sp, (const u64*)nullptr);
    CKL("k2_scatter");
    std::vector<u32> start(RS2_ABINS + 1);
    CK(hipMemcpy2DAsync(start.data(), 4, matrix, nch * 4, 4, RS2_ABINS + 1, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    start[RS2_ABINS] = (u32)n;
    const u64 rs_max = ctx->tune.rs_max_rows ? std::min<u64>(ctx->tune.rs_max_rows, RS_MAX_ROWS) : RS_MAX_ROWS;
    int sb = -1;
    for (int bits = 0; bits <= bA && sb < 0; ++bits) {
        const u32 per = (1u << bA) >> bits;
        bool ok = true;
        for (u32 g0 = 0; g0 < (1u << bA) && ok; g0 += per) ok = (u64)start[g0 + per] - start[g0] <= rs_max;
        if (ok) sb = bits;
    }
    ctx->res_w[1] = T.hi; ctx->res_w[0] = T.lo; ctx->res_ab = T.ab; ctx->sort_partial = true;
    ctx->rows2_in_scratch = true; ctx->rows2_scratch = T;
    if (sb < 0) { ctx->h_back[3] = 1; return DSKGPU_OK; }
    const u32 per = (1u << bA) >> sb;
    for (u32 g0 = 0; g0 < (1u << bA); g0 += per) {
        const u64 b = start[g0], e = start[g0 + per];
        if (e - b < 2 || total - sb < 1) continue;
        const Rows2 r2{T.hi + b, T.lo + b, T.ab + b}, s2{K.hi + b, K.lo + b, K.ab + b};
        const int rc = msd_sort_rows2(ctx, r2, s2, e - b, total - sb, false, (u32)b);
        if (rc) return rc;
    }
    { const int e = rows2_rounds(ctx, T, K); if (e) return e; }
    CK(hipMemcpyAsync(&ctx->h_back[3], sc + SC_SORTFLAG, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] two-word row sort: %llu rows in %u groups on their top %d bits, MSD sort per group\n", (unsigned long long)n, 1u << sb, sb);
    return DSKGPU_OK;
}

// one-word rows: out_* ordered in place (srt_* = scratch and, for run_pipeline's fallback, a complete permutation of the rows)
int sort_rows_msd(dskgpu_ctx* ctx, u64 n) {
    const int e = msd_sort_pairs(ctx, ctx->out_w[0].as<u64>(), ctx->out_ab.as<u32>(), ctx->srt_w[0].as<u64>(), ctx->srt_ab.as<u32>(), n,
                                 (int)std::min(64u, 2u * ctx->cfg.kmer_size), true, 0u, ctx->sp_rows.valid ? &ctx->sp_rows : nullptr);
    ctx->sp_rows.valid = false; ctx->sp_rows2.valid = false;
    if (e) return e;
    ctx->h_ovs.assign(1, 0);
    ctx->sort_back = 1;      // (flag and sub-bucket count travel with the histogram: run_pipeline)
    ctx->sort_partial = true;
    ctx->res_w[0] = ctx->out_w[0].as<u64>(); ctx->res_ab = ctx->out_ab.as<u32>();
    ctx->rs_res_k = ctx->out_w[0].as<u64>(); ctx->rs_res_v = ctx->out_ab.as<u32>(); ctx->rs_tmp_k = ctx->srt_w[0].as<u64>(); ctx->rs_tmp_v = ctx->srt_ab.as<u32>();
    return DSKGPU_OK;
}

// The sub-buckets a row sort listed instead of ordering them (k_rs_big: more than RS_BLOCK_ROWS rows share two digits -- the error
// variants of a k-mer with 10^8 occurrences share 13 and more leading bases): the rows of all listed ranges are gathered under
// the composite key (range number, value bits the range has not used yet), ordered by ONE more MSD sort and put back
// (k_ovs_gather / k_ovs_scatter) -- which may list ranges of the gathered array again: a few rounds at most, the rows of a range
// share ever longer prefixes.  Called after the host has seen a non-zero count (ctx->h_ovs[0]); synchronous.  Leaves
// ctx->h_back[3] != 0 when rounds or list run out: the caller's full-width fallback.
int sort_oversize_round(dskgpu_ctx* ctx, u64* res_k, u32* res_v, int depth) {
    u32* sc = ctx->scalars.as<u32>();
    const u32 cnt = ctx->h_ovs[0];
    if (cnt == 0) return DSKGPU_OK;
    if (cnt > RS_OVS_CAP || depth >= 4) { ctx->h_back[3] = 1; return DSKGPU_OK; }
    std::vector<u32> list(3 * (size_t)cnt), starts(cnt + 1, 0);
    CK(hipMemcpyAsync(list.data(), ctx->rs_ovs.as<u32>() + 1, list.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    u64 tot = 0; u32 maxbits = 1;
    for (u32 i = 0; i < cnt; ++i) { starts[i] = (u32)tot; tot += list[3 * i + 1]; maxbits = std::max(maxbits, list[3 * i + 2]); }
    if (tot >= 0xFFFF0000ull || maxbits > RS_OVS_SHIFT) { ctx->h_back[3] = 1; return DSKGPU_OK; }
    if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] row sort: %u sub-bucket(s) above %u rows (%llu rows in all, up to %u bits left) go round again\n", cnt, (u32)RS_BLOCK_ROWS, (unsigned long long)tot, maxbits);
    DevBuf& g = ctx->rs_g[depth];
    const size_t n_al = (size_t)((tot + 31) & ~(u64)31);
    const size_t meta = ((size_t)cnt * 4 * 4 + (size_t)cnt * 8 + 255) & ~size_t(255);      // list (3 words) + starts, then the prefixes
    CK(g.ensure(2 * n_al * 12 + meta + 256));
    u64* gk = g.as<u64>(); u64* tk = gk + n_al; u32* gv = reinterpret_cast<u32*>(tk + n_al); u32* tv = gv + n_al;
    u32* d_list = tv + n_al; u32* d_starts = d_list + 3 * (size_t)cnt;
    u64* d_prefix = reinterpret_cast<u64*>(reinterpret_cast<char*>(d_list) + (((size_t)cnt * 16 + 7) & ~size_t(7)));
    CK(hipMemcpyAsync(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_starts, starts.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, ctx->stream));
    const unsigned gridr = (unsigned)std::min<u64>(cnt, (u64)ctx->num_cu * 8);
    hipLaunchKernelGGL(k_ovs_gather, dim3(gridr), dim3(256), 0, ctx->stream, (const u64*)res_k, (const u32*)res_v, (const u32*)d_list, (const u32*)d_starts, cnt, gk, gv, d_prefix, maxbits);
    CKL("k_ovs_gather");
    CK(hipStreamSynchronize(ctx->stream));          // (list / starts were read from host vectors)
    CK(hipMemsetAsync(ctx->rs_ovs.p, 0, 4, ctx->stream));
    int idbits = 1; while ((1u << idbits) < cnt) ++idbits;
    { const int rc = msd_sort_pairs(ctx, gk, gv, tk, tv, tot, (int)maxbits + idbits, false, 0u); if (rc) return rc; }
    ctx->h_ovs.assign(1, 0);
    CK(hipMemcpyAsync(ctx->h_ovs.data(), ctx->rs_ovs.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(&ctx->h_back[3], sc + SC_SORTFLAG, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (!ctx->h_back[3] && ctx->h_ovs[0]) { const int rc = sort_oversize_round(ctx, gk, gv, depth + 1); if (rc) return rc; }
    if (ctx->h_back[3]) return DSKGPU_OK;
    hipLaunchKernelGGL(k_ovs_scatter, dim3(gridr), dim3(256), 0, ctx->stream, res_k, res_v, (const u32*)d_list, (const u32*)d_starts, cnt, (const u64*)gk, (const u32*)gv, (const u64*)d_prefix, maxbits);
    CKL("k_ovs_scatter");
    return DSKGPU_OK;
}
int sort_oversize(dskgpu_ctx* ctx) { return sort_oversize_round(ctx, ctx->rs_res_k, ctx->rs_res_v, 0); }

// Row sets above RS_MAX_ROWS (one-word rows; the 3 * 10^9 solid k-mers of a 30x human run, the 6 * 10^8 of 200 M reads): step A of
// the MSD sort ONCE over all rows -- 1024 buckets on the top 10 value bits, exact offsets (rows < 2^32) -- then the rows that
// share their top `sb` bits (sb = the fewest bits for which every such group holds <= RS_MAX_ROWS rows: 3-5 for those inputs)
// are ordered group by group with the MSD sort on the remaining bits.  No library kernel; what the MSD kernels do not order
// themselves raises the same flag as for small row sets (full-width fallback in run_pipeline).  Scratch for the second copy of
// the rows: the level-0 buffer of a multi-pass count when it is large enough (its key arrays are dead by now), else srt_* --
// after giving back the partition buffers when HBM is short (the next count allocates them again).
// In: rows in out_w[0] / out_ab.  Out: ctx->res_* (the scratch copy), ctx->fb_*: where the fallback finds a permutation of the rows.
int sort_rows_big(dskgpu_ctx* ctx, u64 n) {
    u64* k = ctx->out_w[0].as<u64>(); u32* v = ctx->out_ab.as<u32>();
    const size_t n_al = (size_t)((n + 31) & ~(u64)31), need = n_al * 12 + 256;
    u64* tk; u32* tv;
    if (ctx->l0buf.cap >= need) { tk = ctx->l0buf.as<u64>(); tv = reinterpret_cast<u32*>(tk + n_al); }
    else {
        size_t free_b = 0, total_b = 0;
        CK(hipMemGetInfo(&free_b, &total_b));
        if (ctx->srt_w[0].cap + ctx->srt_ab.cap + free_b < need + ((size_t)2 << 30)) { ctx->l0buf.release(); ctx->bufA.release(); ctx->bufB.release(); }
        CK(ctx->srt_w[0].ensure(n * 8)); CK(ctx->srt_ab.ensure(n * 4));
        tk = ctx->srt_w[0].as<u64>(); tv = ctx->srt_ab.as<u32>();
    }
    const int total = (int)std::min(64u, 2u * ctx->cfg.kmer_size);
    const int bA = std::min(RS_ABITS, total);
    RsSpec sp{total - bA, 0, 0, (1u << bA) - 1u, 0u, 0u};
    const u64 ncu = (u64)ctx->num_cu;
    u64 nch = (n + 65535) / 65536;
    nch = (nch + ncu - 1) / ncu * ncu;
    const u64 chunk = (n + nch - 1) / nch;
    nch = (n + chunk - 1) / chunk;
    const u64 M = (u64)RS_ABINS * nch;
    if (M >= 0xFFFFFFF0ull) return fail(ctx, DSKGPU_E_ARG, "row sort: too many rows");
    CK(ctx->mat2.ensure((M + 2) * 4));
    u32* matrix = ctx->mat2.as<u32>();
    u32* sc = ctx->scalars.as<u32>();
    CK(ctx->rs_ovs.ensure((1 + 3 * RS_OVS_CAP) * 4));
    hipLaunchKernelGGL(k_set_rs_scalars, dim3(1), dim3(64), 0, ctx->stream, sc + SC_RSLEN, (u32)M, 4u, sc + SC_SORTFLAG, ctx->rs_ovs.as<u32>());
    ctx->h_ovs.assign(1, 0);
    ctx->rs_res_k = tk; ctx->rs_res_v = tv; ctx->rs_tmp_k = k; ctx->rs_tmp_v = v;
    { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_rs_scatter<false>)); if (e) return e; }
    hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)nch), dim3(RS_NT), 0, ctx->stream, k, n, (u32)chunk, (u32)nch, matrix, sp, 0u);
    CKL("k_rs_hist");
    { const int e = run_scan(ctx, matrix, sc + SC_RSLEN, M); if (e) return e; }
    const size_t ldsA = RsLds<RS_ABINS, RS_TILE>::bytes;
    hipLaunchKernelGGL(k_rs_scatter<false>, dim3((unsigned)nch), dim3(RS_NT), ldsA, ctx->stream, k, v, n, (u32)chunk, (u32)nch, matrix, tk, tv, sp, (const u64*)nullptr, 0u);
    CKL("k_rs_scatter");
    // bucket starts -> host (entry b * nch of the scanned matrix; the scan leaves the total behind the last entry)
    std::vector<u32> start(RS_ABINS + 1);
    CK(hipMemcpy2DAsync(start.data(), 4, matrix, nch * 4, 4, RS_ABINS + 1, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    start[RS_ABINS] = (u32)n;
    const u64 rs_max = ctx->tune.rs_max_rows ? std::min<u64>(ctx->tune.rs_max_rows, RS_MAX_ROWS) : RS_MAX_ROWS;
    int sb = -1;
    for (int bits = 0; bits <= bA && sb < 0; ++bits) {
        const u32 per = (1u << bA) >> bits;               // 10-bit buckets per group
        bool ok = true;
        for (u32 g0 = 0; g0 < (1u << bA) && ok; g0 += per) ok = (u64)start[g0 + per] - start[g0] <= rs_max;
        if (ok) sb = bits;
    }
    ctx->fb_src_k = tk; ctx->fb_src_v = tv; ctx->fb_dst_k = k; ctx->fb_dst_v = v;
    ctx->res_w[0] = tk; ctx->res_ab = tv; ctx->sort_partial = true;
    if (sb < 0) {                                          // one 10-bit bucket alone exceeds the MSD sort: not a k-mer spectrum -> full-width fallback
        ctx->h_back[3] = 1;
        return DSKGPU_OK;
    }
    const u32 per = (1u << bA) >> sb;
    bool first = true;
    for (u32 g0 = 0; g0 < (1u << bA); g0 += per) {
        const u64 b = start[g0], e = start[g0 + per];
        if (e - b < 2 || total - sb < 1) continue;
        const int rc = msd_sort_pairs(ctx, tk + b, tv + b, k + b, v + b, e - b, total - sb, false, (u32)b);      // (in place in tk / tv; k / v = its scratch)
        if (rc) return rc;
        first = false;
    }
    (void)first;
    CK(hipMemcpyAsync(&ctx->h_back[3], sc + SC_SORTFLAG, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(ctx->h_ovs.data(), ctx->rs_ovs.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] row sort: %llu rows in %u groups on their top %d bits, MSD sort per group\n", (unsigned long long)n, 1u << sb, sb);
    return DSKGPU_OK;
}


// ---- row sets of 2^32 rows and more (one- and two-word rows): `-abundance-min 1` on a 200 M-read input, the solid k-mers of a deeper
// human run.  The reference streams rows to Partition<Count> without any such bound (utils/dsk2ascii.cpp:61,77).  The MSD kernels
// index rows with 32 bits, so step A runs SLAB by slab (2^31 rows each): per slab the first-digit histogram and its (32-bit, slab-
// local) scan, then the host lays the 1024 buckets out over ALL rows in 64 bits -- bucket b = [B[b], B[b + 1]), inside it the slabs'
// shares in slab order -- and hands every slab's scatter a 64-bit offset per bin (k_rs_scatter<true>: output index = slab-local
// index + gdel[bin]).  After that every bucket is far below 2^32 rows and the groups of buckets that share their top bits (<= RS_MAX_ROWS
// rows each) are ordered one by one with the ordinary MSD sort, each group with its own flag / list round trip: what a group's sort
// does not order itself goes through the library radix sort FOR THAT GROUP (one-word rows; a group is < 2^32 rows) -- exact for any
// value distribution.  Scratch = a second copy of the rows (the level-0 buffer of the multi-pass count when it is large enough).
template <int W>
int sort_rows_huge(dskgpu_ctx* ctx, u64 n) {
    static_assert(W == 1 || W == 2, "one- and two-word rows");
    constexpr int ABITS = W == 1 ? RS_ABITS : RS2_ABITS;       // first digit of the one- / two-word kernels
    constexpr u32 ABINS = 1u << ABITS;
    const size_t n_al = (size_t)((n + 31) & ~(u64)31), row_bytes = W == 1 ? 12 : 20, need = n_al * row_bytes + 256;
    u64* k = ctx->out_w[0].as<u64>(); u32* v = ctx->out_ab.as<u32>();
    u64* tk = nullptr; u32* tv = nullptr;              // one-word rows: the scratch copy (becomes the result)
    Rows2 K{nullptr, nullptr, nullptr}, T{nullptr, nullptr, nullptr};
    if (W == 2) K = Rows2{ctx->out_w[1].as<u64>(), ctx->out_w[0].as<u64>(), ctx->out_ab.as<u32>()};
    if (ctx->l0buf.cap >= need) {
        u64* base = ctx->l0buf.as<u64>();
        if (W == 1) { tk = base; tv = reinterpret_cast<u32*>(tk + n_al); }
        else { T.hi = base; T.lo = T.hi + n_al; T.ab = reinterpret_cast<u32*>(T.lo + n_al); }
    } else {
        ctx->l0buf.release(); ctx->bufA.release(); ctx->bufB.release();       // (the next count allocates them again)
        for (int x = 0; x < W; ++x) CK(ctx->srt_w[x].ensure(n * 8));
        CK(ctx->srt_ab.ensure(n * 4));
        if (W == 1) { tk = ctx->srt_w[0].as<u64>(); tv = ctx->srt_ab.as<u32>(); }
        else T = Rows2{ctx->srt_w[1].as<u64>(), ctx->srt_w[0].as<u64>(), ctx->srt_ab.as<u32>()};
    }
    const int total = W == 1 ? (int)std::min(64u, 2u * ctx->cfg.kmer_size) : 2 * (int)ctx->cfg.kmer_size;
    const int bA = std::min(ABITS, total);
    const u32 NB = 1u << bA;
    RsSpec sp{total - bA, 0, 0, NB - 1u, 0u, 0u};
    const u64 ncu = (u64)ctx->num_cu;
    const u64 slab = ctx->tune.rs_slab_rows ? std::max<u64>(ctx->tune.rs_slab_rows, 1024) : (1ull << 31);
    const u32 S = (u32)((n + slab - 1) / slab);
    // per slab: chunks of ~64 K rows (a multiple of the CU count of them), matrix of ABINS x nch counters
    std::vector<u64> s_n(S), s_nch(S), s_chunk(S), s_moff(S + 1, 0);
    for (u32 sl = 0; sl < S; ++sl) {
        const u64 ns = std::min<u64>(slab, n - (u64)sl * slab);
        u64 nch = (ns + 65535) / 65536;
        nch = (nch + ncu - 1) / ncu * ncu;
        const u64 tile = W == 1 ? RS_TILE : RS2_TILE;
        nch = std::max<u64>(1, std::min<u64>(nch, (ns + tile - 1) / tile));
        const u64 chunk = (ns + nch - 1) / nch;
        nch = (ns + chunk - 1) / chunk;
        s_n[sl] = ns; s_nch[sl] = nch; s_chunk[sl] = chunk;
        s_moff[sl + 1] = s_moff[sl] + (u64)ABINS * nch + 2;
    }
    CK(ctx->mat2.ensure(s_moff[S] * 4));
    CK(ctx->rs_lens.ensure((size_t)S * 4));
    ctx->h_rs_lens.resize(S);
    for (u32 sl = 0; sl < S; ++sl) ctx->h_rs_lens[sl] = (u32)((u64)ABINS * s_nch[sl]);
    CK(hipMemcpyAsync(ctx->rs_lens.p, ctx->h_rs_lens.data(), (size_t)S * 4, hipMemcpyHostToDevice, ctx->stream));
    u32* sc = ctx->scalars.as<u32>();
    CK(ctx->rs_ovs.ensure((1 + 3 * RS_OVS_CAP) * 4));
    ctx->h_rs_lin.assign((size_t)S * (ABINS + 1), 0);
    for (u32 sl = 0; sl < S; ++sl) {
        const u64 r0 = (u64)sl * slab;
        u32* matrix = ctx->mat2.as<u32>() + s_moff[sl];
        if (W == 1) hipLaunchKernelGGL(k_rs_hist, dim3((unsigned)s_nch[sl]), dim3(RS_NT), 0, ctx->stream, (const u64*)(k + r0), s_n[sl], (u32)s_chunk[sl], (u32)s_nch[sl], matrix, sp, 0u);
        else { const Rows2C kc{K.hi + r0, K.lo + r0, K.ab + r0}; hipLaunchKernelGGL(k2_hist, dim3((unsigned)s_nch[sl]), dim3(RS_NT), 0, ctx->stream, kc, s_n[sl], (u32)s_chunk[sl], (u32)s_nch[sl], matrix, sp); }
        CKL("k_rs_hist(slab)");
        { const int e = run_scan(ctx, matrix, ctx->rs_lens.as<u32>() + sl, (u64)ABINS * s_nch[sl]); if (e) return e; }
        // bucket starts inside the slab (entry b * nch of the scanned matrix; the scan leaves the slab's total behind the last entry)
        CK(hipMemcpy2DAsync(ctx->h_rs_lin.data() + (size_t)sl * (ABINS + 1), 4, matrix, s_nch[sl] * 4, 4, ABINS + 1, hipMemcpyDeviceToHost, ctx->stream));
    }
    CK(hipStreamSynchronize(ctx->stream));
    // the buckets over all rows, and every (slab, bin) pair's place inside its bucket
    std::vector<u64> B(ABINS + 1, 0);
    for (u32 b = 0; b < ABINS; ++b) {
        u64 t = 0;
        for (u32 sl = 0; sl < S; ++sl) { const u32* lin = ctx->h_rs_lin.data() + (size_t)sl * (ABINS + 1); t += (u64)(lin[b + 1] - lin[b]); }
        B[b + 1] = B[b] + t;
    }
    if (B[ABINS] != n) return fail(ctx, DSKGPU_E_DEVICE, "row sort: the slabs' histograms do not add up to the rows");
    ctx->h_rs_del.assign((size_t)S * ABINS, 0);
    {
        std::vector<u64> at(B.begin(), B.end() - 1);              // next free row of every bucket
        for (u32 sl = 0; sl < S; ++sl) {
            const u32* lin = ctx->h_rs_lin.data() + (size_t)sl * (ABINS + 1);
            for (u32 b = 0; b < ABINS; ++b) { ctx->h_rs_del[(size_t)sl * ABINS + b] = at[b] - (u64)lin[b]; at[b] += (u64)(lin[b + 1] - lin[b]); }      // (wraps in 64 bits: added back by the kernel)
        }
    }
    CK(ctx->rs_del.ensure(ctx->h_rs_del.size() * 8));
    CK(hipMemcpyAsync(ctx->rs_del.p, ctx->h_rs_del.data(), ctx->h_rs_del.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (W == 1) { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k_rs_scatter<true>)); if (e) return e; }
    else { const int e = allow_big_lds(ctx, reinterpret_cast<const void*>(&k2_scatter<true>)); if (e) return e; }
    const size_t ldsA1 = RsLds<RS_ABINS, RS_TILE>::bytes, ldsA2 = Rs2Lds<RS2_ABINS, RS2_TILE>::bytes;
    for (u32 sl = 0; sl < S; ++sl) {
        const u64 r0 = (u64)sl * slab;
        const u32* matrix = ctx->mat2.as<u32>() + s_moff[sl];
        const u64* gdel = ctx->rs_del.as<u64>() + (size_t)sl * ABINS;
        if (W == 1) hipLaunchKernelGGL(k_rs_scatter<true>, dim3((unsigned)s_nch[sl]), dim3(RS_NT), ldsA1, ctx->stream, (const u64*)(k + r0), (const u32*)(v + r0), s_n[sl],
                                       (u32)s_chunk[sl], (u32)s_nch[sl], matrix, tk, tv, sp, gdel, 0u);
        else { const Rows2C kc{K.hi + r0, K.lo + r0, K.ab + r0};
               hipLaunchKernelGGL(k2_scatter<true>, dim3((unsigned)s_nch[sl]), dim3(RS_NT), ldsA2, ctx->stream, kc, s_n[sl], (u32)s_chunk[sl], (u32)s_nch[sl], matrix, T, sp, gdel); }
        CKL("k_rs_scatter(slab)");
    }
    CK(hipStreamSynchronize(ctx->stream));                    // (h_rs_del / h_rs_lens were read from host vectors)
    // groups of buckets that share their top sb bits, each <= RS_MAX_ROWS rows; a single bucket above that stands alone (library sort)
    const u64 rs_max = ctx->tune.rs_max_rows ? std::min<u64>(ctx->tune.rs_max_rows, RS_MAX_ROWS) : RS_MAX_ROWS;
    int sb = bA;
    for (int bits = 0; bits <= bA; ++bits) {
        const u32 per = NB >> bits;
        bool ok = true;
        for (u32 g0 = 0; g0 < NB && ok; g0 += per) ok = B[g0 + per] - B[g0] <= rs_max;
        if (ok) { sb = bits; break; }
    }
    const u32 per = NB >> sb;
    ctx->h_ovs.assign(1, 0);
    u32 ngroups = 0, nlib = 0;
    for (u32 g0 = 0; g0 < NB; g0 += per) {
        const u64 b = B[g0], e = B[g0 + per], m = e - b;
        if (m < 2 || total - sb < 1) continue;
        ++ngroups;
        bool lib = m > rs_max;
        if (!lib) {
            CK(hipMemsetAsync(sc + SC_SORTFLAG, 0, 4, ctx->stream)); CK(hipMemsetAsync(ctx->rs_ovs.p, 0, 4, ctx->stream));
            if (W == 1) {
                const int rc = msd_sort_pairs(ctx, tk + b, tv + b, k + b, v + b, m, total - sb, false, 0u);
                if (rc) return rc;
                ctx->h_ovs.assign(1, 0);
                CK(hipMemcpyAsync(&ctx->h_back[3], sc + SC_SORTFLAG, 4, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipMemcpyAsync(ctx->h_ovs.data(), ctx->rs_ovs.p, 4, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipStreamSynchronize(ctx->stream));
                if (!ctx->h_back[3] && ctx->h_ovs[0]) { const int e2 = sort_oversize_round(ctx, tk + b, tv + b, 0); if (e2) return e2; CK(hipStreamSynchronize(ctx->stream)); }
            } else {
                const Rows2 r2{T.hi + b, T.lo + b, T.ab + b}, s2{K.hi + b, K.lo + b, K.ab + b};
                const int rc = msd_sort_rows2(ctx, r2, s2, m, total - sb, false, 0u);
                if (rc) return rc;
                { const int e2 = rows2_rounds(ctx, r2, s2); if (e2) return e2; }      // (synchronises; leaves the flag in h_back[3])
            }
            lib = ctx->h_back[3] != 0;
        }
        if (lib && m >= 0xFFFF0000ull) return fail(ctx, DSKGPU_E_OVERFLOW, "row sort: one value range of the >= 2^32 rows holds 2^32 rows itself (not a k-mer spectrum)");
        if (lib && W == 2) {
            // two-word rows of this group in full-width order: W stable radix passes over an index permutation, gathered into the
            // other copy and copied back (the group's rows in T are a complete permutation whatever the MSD kernels did to them)
            const u64* src[2] = {T.lo + b, T.hi + b};
            { const int rc = sort_index_multiword(ctx, src, m, 2); if (rc) return rc; }
            const unsigned gb = (unsigned)((m + 255) / 256);
            const u32* idx = ctx->srt_idx.as<u32>();
            hipLaunchKernelGGL(k_gather<u64>, dim3(gb), dim3(256), 0, ctx->stream, K.lo + b, (const u64*)(T.lo + b), idx, m);
            hipLaunchKernelGGL(k_gather<u64>, dim3(gb), dim3(256), 0, ctx->stream, K.hi + b, (const u64*)(T.hi + b), idx, m);
            hipLaunchKernelGGL(k_gather<u32>, dim3(gb), dim3(256), 0, ctx->stream, K.ab + b, (const u32*)(T.ab + b), idx, m);
            CKL("row sort: group gather");
            CK(hipMemcpyAsync(T.lo + b, K.lo + b, m * 8, hipMemcpyDeviceToDevice, ctx->stream));
            CK(hipMemcpyAsync(T.hi + b, K.hi + b, m * 8, hipMemcpyDeviceToDevice, ctx->stream));
            CK(hipMemcpyAsync(T.ab + b, K.ab + b, m * 4, hipMemcpyDeviceToDevice, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            ctx->stats.sort_fallback = 1; ++nlib;
        } else if (lib) {
            size_t tmp = 0;
            const unsigned end_bit = (unsigned)(total - sb);
            // LIBRARY SORT (rocprim), labelled FALLBACK: one group of a >= 2^32-row set that the MSD kernels gave up on (or that exceeds them alone)
            CK(rocprim::radix_sort_pairs(nullptr, tmp, tk + b, k + b, tv + b, v + b, (size_t)m, 0u, end_bit, ctx->stream));
            CK(ctx->srt_tmp.ensure(tmp));
            CK(rocprim::radix_sort_pairs(ctx->srt_tmp.p, tmp, tk + b, k + b, tv + b, v + b, (size_t)m, 0u, end_bit, ctx->stream));
            CK(hipMemcpyAsync(tk + b, k + b, m * 8, hipMemcpyDeviceToDevice, ctx->stream));
            CK(hipMemcpyAsync(tv + b, v + b, m * 4, hipMemcpyDeviceToDevice, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            ctx->stats.sort_fallback = 1; ++nlib;
        }
    }
    if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] row sort: %llu rows (>= 2^32 path) in %u slab(s), %u group(s) on their top %d bits, %u through the library sort\n", (unsigned long long)n, S, ngroups, sb, nlib);
    ctx->h_back[3] = 0; ctx->h_ovs.assign(1, 0);
    ctx->sort_partial = false; ctx->rows2_in_scratch = false;
    if (W == 1) { ctx->res_w[0] = tk; ctx->res_ab = tv; }
    else { ctx->res_w[1] = T.hi; ctx->res_w[0] = T.lo; ctx->res_ab = T.ab; }
    return DSKGPU_OK;
}

// ---- DSKGPU_F_PARTITION_ORDER: one-word rows of a single pass, ordered inside output partitions of <= PS_CAP rows by one LDS pass
// (partsort.h).  In: the sparse rows (ctx->sp_rows).  Out: srt_w[0] / srt_ab dense, partition after partition; part_off on the device
// and (after the caller's synchronisation) in h_part_off; SC_SORTFLAG raised when a block could not order its partition.
// one launch: the sparse rows `spr` / `spr2` (W = 1 / 2) -> dense rows at ov / o2, partition offsets (relative to the first row) at d_part_off
// [0 .. *nparts], *d_flag raised when a block could not order its partition (the rows are complete either way)
int launch_part_sort(dskgpu_ctx* ctx, int W, const dskgpu_ctx::SparseRows& spr, const dskgpu_ctx::SparseRows2& spr2, u64* ov, u32* oab, Rows2 o2,
                     u32* d_part_off, u32* d_flag, u32* nparts_out, u32* qpp_out) {
    const u64 F = W == 1 ? spr.s.F : spr2.s.F, n_sparse = W == 1 ? spr.n_sparse : spr2.n_sparse;
    const u32 n_tail = W == 1 ? spr.n_tail : spr2.n_tail;
    const u64 cap_rows = W == 1 ? PS_CAP : PS2_CAP;
    const u64 mean = std::max<u64>(1, (n_sparse + F - 1) / std::max<u64>(F, 1));
    const u32 qpp = (u32)std::min<u64>(std::max<u64>(1, (cap_rows * 3 / 4) / mean), PS_MAXQ);
    const u32 nps = (u32)((F + qpp - 1) / qpp), nparts = nps + (n_tail ? 1u : 0u);
    const int sh = std::max(0, 2 * (int)ctx->cfg.kmer_size - 12);
    const PsParams pp{qpp, nps, W == 1 ? std::min(sh, 52) : sh, n_tail, ctx->tune.ps_maxc ? std::min<u32>(ctx->tune.ps_maxc, PS_MAXC) : PS_MAXC};
    if (W == 1) hipLaunchKernelGGL(k_part_sort, dim3(nparts), dim3(PS_NT), 0, ctx->stream, spr.s, spr.tail_k, spr.tail_v, pp, ov, oab, d_part_off, d_flag);
    else hipLaunchKernelGGL(k_part_sort2, dim3(nparts), dim3(PS_NT), 0, ctx->stream, spr2.s, spr2.tail, pp, o2, d_part_off, d_flag);
    CKL("k_part_sort");
    *nparts_out = nparts; if (qpp_out) *qpp_out = qpp;
    return DSKGPU_OK;
}
// partitions a launch will make for F sub-partitions holding n_sparse rows (+ a tail)
u32 part_sort_nparts(int W, u64 F, u64 n_sparse, u32 n_tail) {
    const u64 cap_rows = W == 1 ? PS_CAP : PS2_CAP;
    const u64 mean = std::max<u64>(1, (n_sparse + F - 1) / std::max<u64>(F, 1));
    const u32 qpp = (u32)std::min<u64>(std::max<u64>(1, (cap_rows * 3 / 4) / mean), PS_MAXQ);
    return (u32)((F + qpp - 1) / qpp) + (n_tail ? 1u : 0u);
}
int sort_rows_partition_order(dskgpu_ctx* ctx, u64 n) {
    const int W = ctx->W;
    const dskgpu_ctx::SparseRows spr = ctx->sp_rows;
    const dskgpu_ctx::SparseRows2 spr2 = ctx->sp_rows2;
    ctx->sp_rows_saved = spr; ctx->sp_rows2_saved = spr2;
    ctx->sp_rows.valid = false; ctx->sp_rows2.valid = false;
    const u32 nparts = part_sort_nparts(W, W == 1 ? spr.s.F : spr2.s.F, W == 1 ? spr.n_sparse : spr2.n_sparse, W == 1 ? spr.n_tail : spr2.n_tail);
    for (int x = 0; x < W; ++x) CK(ctx->srt_w[x].ensure(n * 8));
    CK(ctx->srt_ab.ensure(n * 4));
    CK(ctx->part_off.ensure(((size_t)nparts + 2) * 4));
    if (ctx->h_part_cap < (size_t)nparts + 2) {
        if (ctx->h_part_off) CK(hipHostFree(ctx->h_part_off));
        ctx->h_part_off = nullptr; ctx->h_part_cap = 0;
        CK(hipHostMalloc(reinterpret_cast<void**>(&ctx->h_part_off), ((size_t)nparts + 2) * 4 * 2, hipHostMallocDefault));
        ctx->h_part_cap = ((size_t)nparts + 2) * 2;
    }
    u32* sc = ctx->scalars.as<u32>();
    hipLaunchKernelGGL(k_set_rs_scalars, dim3(1), dim3(64), 0, ctx->stream, sc + SC_RSLEN, 0u, 1u, sc + SC_SORTFLAG, (u32*)nullptr);
    u32 np = 0;
    { const int rc = launch_part_sort(ctx, W, spr, spr2, ctx->srt_w[0].as<u64>(), ctx->srt_ab.as<u32>(), Rows2{ctx->srt_w[1].as<u64>(), ctx->srt_w[0].as<u64>(), ctx->srt_ab.as<u32>()},
                                      ctx->part_off.as<u32>(), sc + SC_SORTFLAG, &np, nullptr); if (rc) return rc; }
    CK(hipMemcpyAsync(ctx->h_part_off, ctx->part_off.p, ((size_t)nparts + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
    ctx->part_mode = true; ctx->n_parts = nparts; ctx->h_part_off64.clear();
    ctx->h_ovs.assign(1, 0);
    ctx->sort_back = 2;            // (the flag travels with the histogram: run_pipeline)
    ctx->sort_partial = false;
    for (int x = 0; x < W; ++x) ctx->res_w[x] = ctx->srt_w[x].as<u64>();
    ctx->res_ab = ctx->srt_ab.as<u32>();
    return DSKGPU_OK;
}

// ---- result post-processing: sort rows by k-mer value
int sort_rows(dskgpu_ctx* ctx, u64 n) {
    const int W = ctx->W;
    for (int x = 0; x < 4; ++x) ctx->res_w[x] = x < W ? ctx->out_w[x].as<u64>() : nullptr;
    ctx->res_ab = ctx->out_ab.as<u32>();
    ctx->sort_partial = false;
    ctx->rows2_in_scratch = false;
    ctx->part_mode = false;
    ctx->h_ovs.assign(1, 0);
    if ((ctx->sp_rows.valid && (n == 0 || (ctx->cfg.flags & DSKGPU_F_NO_SORT) || W != 1)) || (ctx->sp_rows2.valid && (n == 0 || (ctx->cfg.flags & DSKGPU_F_NO_SORT) || W != 2)))
        return fail(ctx, DSKGPU_E_STATE, "row sort: sparse rows on a path that cannot read them");
    if (n == 0 || (ctx->cfg.flags & DSKGPU_F_NO_SORT)) return DSKGPU_OK;
    if ((ctx->sp_rows.valid || ctx->sp_rows2.valid) && (ctx->cfg.flags & DSKGPU_F_PARTITION_ORDER) && !ctx->part_off_this_count && n < 0xFFFF0000ull) return sort_rows_partition_order(ctx, n);
    if (ctx->sp_rows.valid) {      // the rows of a single one-word pass, still in the count kernel's regions (run_one_pass made sure this sort takes them)
        CK(ctx->srt_w[0].ensure(n * 8)); CK(ctx->srt_ab.ensure(n * 4));
        ctx->fb_src_k = ctx->srt_w[0].as<u64>(); ctx->fb_src_v = ctx->srt_ab.as<u32>(); ctx->fb_dst_k = ctx->out_w[0].as<u64>(); ctx->fb_dst_v = ctx->out_ab.as<u32>();
        return sort_rows_msd(ctx, n);
    }
    if (ctx->sp_rows2.valid) return sort_rows2_msd(ctx, n);      // (the two-word twin)
    const u64 rs_max = ctx->tune.rs_max_rows ? std::min<u64>(ctx->tune.rs_max_rows, RS_MAX_ROWS) : RS_MAX_ROWS;
    // 2^32 rows and more (or DSKGPU_RS_SLAB_ROWS: tests): step A slab by slab with 64-bit bucket offsets
    if ((n >= 0xFFFF0000ull || ctx->tune.rs_slab_rows) && W <= 2 && (W == 1 || 2u * ctx->cfg.kmer_size > 64u)) return W == 1 ? sort_rows_huge<1>(ctx, n) : sort_rows_huge<2>(ctx, n);
    if (n >= 0xFFFF0000ull) return fail(ctx, DSKGPU_E_ARG, "row sort: 2^32 rows and more are supported for k <= 64");
    if (W == 1 && n > rs_max && n < 0xFFFF0000ull) return sort_rows_big(ctx, n);
    CK(ctx->srt_w[0].ensure(n * 8));
    CK(ctx->srt_ab.ensure(n * 4));
    ctx->fb_src_k = ctx->srt_w[0].as<u64>(); ctx->fb_src_v = ctx->srt_ab.as<u32>(); ctx->fb_dst_k = ctx->out_w[0].as<u64>(); ctx->fb_dst_v = ctx->out_ab.as<u32>();
    size_t tmp = 0;
    // one-word rows: the hand-written MSD sort (rowsort.h) -- in one piece here, group by group / slab by slab above
    if (W == 1) return sort_rows_msd(ctx, n);
    // two-word rows: the rows themselves through the MSD sort (rowsort2.h)
    if (W == 2 && n < 0xFFFF0000ull)
        return n <= rs_max ? sort_rows2_msd(ctx, n) : sort_rows2_big(ctx, n);
    // four-word rows (k > 64): (top 63 bits of the value, row index) pairs through the MSD sort, gather, then the runs of equal prefix are
    // ordered in place by full comparison (exact fallback: sort_rows_full_multiword).  Row sets above RS_MAX_ROWS: the pairs through the
    // LIBRARY radix sort on their top 32 bits instead (rocprim: the one library sort on a non-fallback path; 384 M four-word rows are 14 GB)
    {
        CK(ctx->srt_k.ensure(n * 8)); CK(ctx->s_val.ensure(n * 8));
        CK(ctx->srt_idx.ensure(n * 4)); CK(ctx->srt_idx2.ensure(n * 4));
        const unsigned gb = (unsigned)((n + 255) / 256);
        RowsIn ri{}; RowsOut ro{};
        for (int x = 0; x < W; ++x) { CK(ctx->srt_w[x].ensure(n * 8)); ri.w[x] = ctx->out_w[x].as<u64>(); ro.w[x] = ctx->srt_w[x].as<u64>(); }
        const int bits = 2 * (int)ctx->cfg.kmer_size;
        const bool msd = n <= RS_MAX_ROWS;
        u64* aos = nullptr;
        if (msd) {      // + one record per row for the gather (the full-width fallback buffers are free until then)
            CK(ctx->srt_k2.ensure(n * 8 * (size_t)(W == 2 ? AosRow<2>::WORDS : AosRow<4>::WORDS)));
            aos = ctx->srt_k2.as<u64>();
            if (W == 2) hipLaunchKernelGGL(k_top_key_aos<2>, dim3(gb), dim3(256), 0, ctx->stream, ri, ctx->out_ab.as<u32>(), n, bits, ctx->srt_k.as<u64>(), ctx->srt_idx.as<u32>(), aos);
            else hipLaunchKernelGGL(k_top_key_aos<4>, dim3(gb), dim3(256), 0, ctx->stream, ri, ctx->out_ab.as<u32>(), n, bits, ctx->srt_k.as<u64>(), ctx->srt_idx.as<u32>(), aos);
        } else if (W == 2) hipLaunchKernelGGL(k_top_key<2>, dim3(gb), dim3(256), 0, ctx->stream, ri, n, bits, ctx->srt_k.as<u64>(), ctx->srt_idx.as<u32>());
        else hipLaunchKernelGGL(k_top_key<4>, dim3(gb), dim3(256), 0, ctx->stream, ri, n, bits, ctx->srt_k.as<u64>(), ctx->srt_idx.as<u32>());
        u32* flag = ctx->scalars.as<u32>() + SC_SORTFLAG;
        const u32* ties = nullptr;                     // MSD path: the tie pass returns at once unless the sort met equal keys
        const u32* idx; const u64* skey; int run_shift;
        if (msd) {
            // the hand-written MSD sort on (top 63 bits, row index): fully ordered by those 63 bits, what is left to k_fix_runs_multi
            // are the rows that share all of them
            const int e = msd_sort_pairs(ctx, ctx->srt_k.as<u64>(), ctx->srt_idx.as<u32>(), ctx->s_val.as<u64>(), ctx->srt_idx2.as<u32>(), n, 63);
            if (e) return e;
            // sub-buckets the sort listed for another round (thousands of rows sharing 26 and more leading bits): ordered before the rows are
            // gathered by index (one host round trip; a 63-bit prefix leaves ties to k_fix_runs_multi either way)
            CK(hipMemcpyAsync(&ctx->h_back[3], flag, 4, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipMemcpyAsync(ctx->h_ovs.data(), ctx->rs_ovs.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            if (!ctx->h_back[3] && ctx->h_ovs[0]) {
                ctx->rs_res_k = ctx->srt_k.as<u64>(); ctx->rs_res_v = ctx->srt_idx.as<u32>(); ctx->rs_tmp_k = ctx->s_val.as<u64>(); ctx->rs_tmp_v = ctx->srt_idx2.as<u32>();
                const int e2 = sort_oversize(ctx);
                if (e2) return e2;
                ctx->h_ovs.assign(1, 0);
            }
            idx = ctx->srt_idx.as<u32>(); skey = ctx->srt_k.as<u64>(); run_shift = 0; ties = ctx->scalars.as<u32>() + SC_RSTIES;
        } else {
            const unsigned begin_bit = 63u - SORT_TOP_BITS;
            // LIBRARY SORT (rocprim), labelled: four-word rows (k > 64) above RS_MAX_ROWS = 384 M rows -- the (63-bit key, index) pairs on their top 32 bits
            CK(rocprim::radix_sort_pairs(nullptr, tmp, ctx->srt_k.as<u64>(), ctx->s_val.as<u64>(), ctx->srt_idx.as<u32>(), ctx->srt_idx2.as<u32>(),
                                         (size_t)n, begin_bit, 63u, ctx->stream));
            CK(ctx->srt_tmp.ensure(tmp));
            CK(rocprim::radix_sort_pairs(ctx->srt_tmp.p, tmp, ctx->srt_k.as<u64>(), ctx->s_val.as<u64>(), ctx->srt_idx.as<u32>(), ctx->srt_idx2.as<u32>(),
                                         (size_t)n, begin_bit, 63u, ctx->stream));
            CK(hipMemsetAsync(flag, 0, 4, ctx->stream));
            idx = ctx->srt_idx2.as<u32>(); skey = ctx->s_val.as<u64>(); run_shift = (int)begin_bit;
        }
        if (aos) {
            if (W == 2) hipLaunchKernelGGL(k_gather_aos<2>, dim3(gb), dim3(256), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), aos, idx, n);
            else hipLaunchKernelGGL(k_gather_aos<4>, dim3(gb), dim3(256), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), aos, idx, n);
        } else {
            const unsigned gb4 = (unsigned)((n + 1023) / 1024);
            if (W == 2) hipLaunchKernelGGL(k_gather_rows<2>, dim3(gb4), dim3(256), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), ri, ctx->out_ab.as<u32>(), idx, n);
            else hipLaunchKernelGGL(k_gather_rows<4>, dim3(gb4), dim3(256), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), ri, ctx->out_ab.as<u32>(), idx, n);
        }
        const unsigned gfix = (unsigned)std::min<u64>(gb, (u64)ctx->num_cu * 32);      // grid-stride kernel: when there are no ties its blocks leave at once
        CK(ctx->fix_list.ensure((1 + 2 * (size_t)FIX_LIST_CAP) * 4));
        CK(hipMemsetAsync(ctx->fix_list.p, 0, 4, ctx->stream));
        u32* fl = ctx->fix_list.as<u32>();
        const unsigned glong = (unsigned)ctx->num_cu;
        if (W == 2) {
            hipLaunchKernelGGL(k_fix_runs_multi<2>, dim3(gfix), dim3(256), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), skey, n, run_shift, flag, ties, fl);
            hipLaunchKernelGGL(k_fix_long_runs<2>, dim3(glong), dim3(1024), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), (const u32*)fl, ties);
        } else {
            hipLaunchKernelGGL(k_fix_runs_multi<4>, dim3(gfix), dim3(256), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), skey, n, run_shift, flag, ties, fl);
            hipLaunchKernelGGL(k_fix_long_runs<4>, dim3(glong), dim3(1024), 0, ctx->stream, ro, ctx->srt_ab.as<u32>(), (const u32*)fl, ties);
        }
        CKL("sort_rows");
        CK(hipMemcpyAsync(&ctx->h_back[3], flag, 4, hipMemcpyDeviceToHost, ctx->stream));
        ctx->sort_partial = true;
        for (int x = 0; x < W; ++x) ctx->res_w[x] = ctx->srt_w[x].as<u64>();
        ctx->res_ab = ctx->srt_ab.as<u32>();
        return DSKGPU_OK;
    }
}

// k-mers per chunk of received records (k_sk_count) -> chunk bases for the expansion, and the total.  Needed up front only when
// the caller did not pass the total (dskgpu_mg_count); otherwise only by paths that expand the records.
#define REC_RESIZE 1001            // expand_records: the k-mer total the pipeline was sized with was an estimate and is off -- ctx->rec_hint holds the real one
// the stream waits for the arrival of the slices up to `upto` (exclusive) -- in order, each once
// -> DSKGPU_OK, or DSKGPU_E_STATE once a gate has failed (the caller enqueues nothing that reads the records)
int rec_gate_upto(dskgpu_ctx* ctx, u32 upto) {
    if (!ctx->rec_gate) return DSKGPU_OK;
    const u32 n = (u32)ctx->rec_slice_end.size();
    for (; ctx->rec_gated < std::min(upto, n); ++ctx->rec_gated)
        if (ctx->rec_gate(ctx->rec_gate_user, ctx->rec_gated) != 0) ctx->rec_gate_failed = true;
    return ctx->rec_gate_failed ? fail(ctx, DSKGPU_E_STATE, "a slice of the exchange did not arrive (the caller's gate failed)") : DSKGPU_OK;
}
int rec_gate_all(dskgpu_ctx* ctx) { return rec_gate_upto(ctx, 0xFFFFFFFFu); }

int sk_sizes(dskgpu_ctx* ctx, u64* total_out) {
    const u64 nrec = ctx->rec_n; const u32 R = ctx->sk_sp.R;
    { const int e = rec_gate_all(ctx); if (e) return e; }      // (a sliced receive: every record has to be there)
    u64 nch = std::min<u64>((nrec + SKX_NT - 1) / SKX_NT, (u64)ctx->num_cu * 16);
    u64 rpc = (nrec + nch - 1) / nch;
    rpc = (rpc + SKX_NT - 1) / SKX_NT * SKX_NT;
    nch = (nrec + rpc - 1) / rpc;
    if (rpc * SK_MAXN >= 0xFFFFFFFFull) return fail(ctx, DSKGPU_E_ARG, "too many records per chunk");
    CK(ctx->sk_sums.ensure(nch * 4)); CK(ctx->sk_cbase.ensure(nch * 8));
    ctx->h_sk_sums.assign(nch, 0); ctx->h_sk_cbase.assign(nch, 0);
    hipLaunchKernelGGL(k_sk_count, dim3((unsigned)nch), dim3(SKX_NT), 0, ctx->stream, ctx->rec_src, nrec, R, (u32)rpc, ctx->sk_sums.as<u32>());
    CKL("k_sk_count");
    CK(hipMemcpyAsync(ctx->h_sk_sums.data(), ctx->sk_sums.p, nch * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    u64 total = 0;
    for (u64 c = 0; c < nch; ++c) { ctx->h_sk_cbase[c] = total; total += ctx->h_sk_sums[c]; }
    CK(hipMemcpyAsync(ctx->sk_cbase.p, ctx->h_sk_cbase.data(), nch * 8, hipMemcpyHostToDevice, ctx->stream));
    ctx->rec_nch = nch; ctx->rec_rpc = rpc; ctx->rec_sized = true;
    *total_out = total;
    return DSKGPU_OK;
}

// records -> dense key array (only when the level-1 scatter cannot read the records itself: exact / multi-pass path)
template <int W>
int expand_records(dskgpu_ctx* ctx, u64 total) {
    typedef typename KeyT<W>::T Key;
    if (ctx->rec_expanded) return DSKGPU_OK;
    if (!ctx->rec_sized) {
        u64 real = 0;
        const int e = sk_sizes(ctx, &real);
        if (e) return e;
        if (real != total && ctx->rec_hint_est) { ctx->rec_hint = real; return REC_RESIZE; }      // the estimate sized the fast path only: once more with the real figure
        if (real != total) return fail(ctx, DSKGPU_E_ARG, "dskgpu_mg_count_sized: n_kmers does not match the k-mers inside the records");
    }
    { const int e = rec_gate_all(ctx); if (e) return e; }
    CK(ctx->sk_keys.ensure((total + 1) * sizeof(Key)));
    hipLaunchKernelGGL(k_sk_expand<W>, dim3((unsigned)ctx->rec_nch), dim3(SKX_NT), 0, ctx->stream, ctx->rec_src, ctx->rec_n, ctx->sk_sp.R,
                       (int)ctx->cfg.kmer_size, (u32)ctx->rec_rpc, ctx->sk_cbase.as<u64>(), ctx->sk_keys.as<Key>());
    CKL("k_sk_expand");
    ctx->rec_expanded = true;
    ctx->mark("mg_expand");
    return DSKGPU_OK;
}
template <> int expand_records<4>(dskgpu_ctx*, u64) { return DSKGPU_E_STATE; }      // records carry k <= 64 only

// Heavy k-mers of a pass (one-word keys): level-1 bins whose sampled load stands 20 % above the median hold a k-mer that alone is a
// large share of a bin.  k_collect_heavy gathers about HV_COLLECT sampled keys of up to HV_SLOTS such bins; a k-mer that makes up
// >= 5 % of a bin's collected keys is heavy: the HV_KEYS heaviest go to hv_buf = [keys | counts | rows], and the level-1
// scatter counts them apart.  ctx->h_load is reduced by what they take away (slice sizes, order of the level-2 segments).
// hv_buf (u64 words): [keys: HV_KEYS x W][counts: HV_KEYS][rows, word x of row r at (W + 1 + x) * HV_KEYS + r][abundances: HV_KEYS x u32]
template <int W> struct HvLayout { static constexpr size_t keys = 0, counts = (size_t)HV_KEYS * W, rows = (size_t)HV_KEYS * (W + 1), ab = (size_t)HV_KEYS * (2 * W + 1), words = (size_t)HV_KEYS * (2 * W + 2); };
template <int W>
int find_heavy(dskgpu_ctx* ctx, bool from_reads, const typename KeyT<W>::T* d_keys_in, u32 nts, const Plan& pl, u32* nheavy_out) {
    typedef typename KeyT<W>::T Key;
    struct HK { u64 w[W]; bool operator<(const HK& o) const { for (int x = W - 1; x >= 0; --x) if (w[x] != o.w[x]) return w[x] < o.w[x]; return false; }
                bool operator==(const HK& o) const { for (int x = 0; x < W; ++x) if (w[x] != o.w[x]) return false; return true; } };
    static_assert(sizeof(HK) == sizeof(Key), "host mirror of a device key");
    *nheavy_out = 0;
    const u32 P1 = pl.P1;
    std::vector<double> sorted(ctx->h_load);
    std::nth_element(sorted.begin(), sorted.begin() + P1 / 2, sorted.end());
    const double median = sorted[P1 / 2];
    std::vector<u32> flagged;
    for (u32 b = 0; b < P1; ++b) if (ctx->h_load[b] > 1.2 * median + 4096.0) flagged.push_back(b);
    if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] find_heavy: median load %.0f, %zu bins above 1.2 x\n", median, flagged.size());
    u64* hvb = nullptr;
    auto reset_buf = [&]() -> int {      // keys all-ones (= unused), counts zero
        CK(ctx->hv_buf.ensure(HvLayout<W>::words * 8));
        hvb = ctx->hv_buf.as<u64>();
        CK(hipMemsetAsync(hvb + HvLayout<W>::keys, 0xFF, (size_t)HV_KEYS * W * 8, ctx->stream));
        CK(hipMemsetAsync(hvb + HvLayout<W>::counts, 0, HV_KEYS * 8, ctx->stream));
        return DSKGPU_OK;
    };
    if (flagged.empty()) return DSKGPU_OK;
    std::sort(flagged.begin(), flagged.end(), [&](u32 a, u32 b) { return ctx->h_load[a] > ctx->h_load[b]; });
    if (flagged.size() > HV_SLOTS) flagged.resize(HV_SLOTS);
    ctx->h_hv_lut.assign(P1, 0xFF);
    for (size_t f = 0; f < flagged.size(); ++f) ctx->h_hv_lut[flagged[f]] = (unsigned char)f;
    const size_t nf = flagged.size();
    CK(ctx->hv_lut.ensure(P1));
    const unsigned grid = (unsigned)std::min<u64>(nts, (u64)ctx->num_cu * 2);
    const size_t per_slot = (size_t)grid * HV_BLOCK_KEYS;
    CK(ctx->hv_collect.ensure((size_t)HV_SLOTS * per_slot * sizeof(Key) + (size_t)HV_SLOTS * grid * 4 + HV_SLOTS * 4));
    u32* d_kept = reinterpret_cast<u32*>(ctx->hv_collect.as<Key>() + (size_t)HV_SLOTS * per_slot);
    u32* d_step = d_kept + (size_t)HV_SLOTS * grid;
    // keep every step-th sampled key of a bin, so that about HV_COLLECT are collected (h_mom: the bin's keys in the sample)
    ctx->h_hv_step.assign(HV_SLOTS, 1);
    for (size_t f = 0; f < flagged.size(); ++f) ctx->h_hv_step[f] = (u32)std::max<u64>(1, ctx->h_mom[2 * (size_t)flagged[f]] / HV_COLLECT);
    CK(hipMemcpyAsync(ctx->hv_lut.p, ctx->h_hv_lut.data(), P1, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_step, ctx->h_hv_step.data(), HV_SLOTS * 4, hipMemcpyHostToDevice, ctx->stream));
    u32* sc = ctx->scalars.as<u32>();
    const bool mp = pl.d1.npass > 1;
    auto launch = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SC_NT), 0, ctx->stream, (const u64*)ctx->packed.as<u64>(), (const u32*)ctx->inval.as<u32>(), d_keys_in,
                           (const ChunkDesc*)ctx->smp_descs.as<ChunkDesc>(), (const u32*)(sc + SC_NCH_S), (int)ctx->cfg.kmer_size, pl.d1, P1,
                           (const unsigned char*)ctx->hv_lut.as<unsigned char>(), d_kept, ctx->hv_collect.as<Key>(), (const u32*)d_step);
    };
    if (from_reads) { if (mp) launch(k_collect_heavy<W, 0, 3>); else launch(k_collect_heavy<W, 0, 1>); }
    else { if (mp) launch(k_collect_heavy<W, 1, 3>); else launch(k_collect_heavy<W, 1, 1>); }
    CKL("k_collect_heavy");
    ctx->h_hv_cnt.resize(nf * grid);
    ctx->h_hv_coll.resize(nf * per_slot * W);
    CK(hipMemcpyAsync(ctx->h_hv_cnt.data(), d_kept, nf * grid * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(ctx->h_hv_coll.data(), ctx->hv_collect.p, nf * per_slot * sizeof(Key), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    // candidates over all flagged bins: (estimated occurrences, key, bin); the HV_KEYS largest are counted apart
    struct Cand { double est; HK key; u32 bin; };
    std::vector<Cand> cands;
    for (size_t f = 0; f < nf; ++f) {
        HK* kk = reinterpret_cast<HK*>(ctx->h_hv_coll.data()) + f * per_slot;       // the blocks' kept keys, made dense in place
        u32 n = 0;
        for (unsigned g = 0; g < grid; ++g) {
            const u32 c = std::min<u32>(ctx->h_hv_cnt[f * grid + g], HV_BLOCK_KEYS);
            for (u32 i = 0; i < c; ++i) kk[n++] = kk[(size_t)g * HV_BLOCK_KEYS + i];
        }
        if (n < 256) continue;
        std::sort(kk, kk + n);
        const u32 bin = flagged[f];
        u32 found = 0;
        for (u32 i = 0; i < n;) {
            u32 j = i + 1;
            while (j < n && kk[j] == kk[i]) ++j;
            if ((u64)(j - i) * 20 >= n) { cands.push_back({ctx->h_load[bin] * (double)(j - i) / (double)n, kk[i], bin}); ++found; }     // >= 5 % of the bin's collected keys
            i = j;
        }
        if (ctx->tune.verbose) fprintf(stderr, "[dskgpu]   bin %u load %.0f: %u keys collected, %u dominant\n", bin, ctx->h_load[bin], n, found);
    }
    std::sort(cands.begin(), cands.end(), [](const Cand& a, const Cand& b) { return a.est > b.est; });
    if (cands.size() > HV_KEYS) cands.resize(HV_KEYS);
    ctx->h_hv_keys.assign((size_t)HV_KEYS * W, DSK_EMPTY);
    for (size_t x = 0; x < cands.size(); ++x) {
        for (int y = 0; y < W; ++y) ctx->h_hv_keys[x * W + y] = cands[x].key.w[y];
        ctx->h_load[cands[x].bin] -= cands[x].est;            // they never reach the bin: slices and the order of the level-2 segments follow
        ctx->h_seg_work[cands[x].bin] -= cands[x].est;
    }
    u32 nheavy = (u32)cands.size();
    if (nheavy) {
        { const int e = reset_buf(); if (e) return e; }
        CK(hipMemcpyAsync(hvb + HvLayout<W>::keys, ctx->h_hv_keys.data(), (size_t)HV_KEYS * W * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    *nheavy_out = nheavy;
    return DSKGPU_OK;
}

// One pass: partition + count the keys of pass `pass` (of `npass`) and leave its solid rows
// (unsorted) in out_w[0]/out_w[1]/out_ab.  Returns PASS_TOO_BIG when the pass holds more keys than `cap`.
#define PASS_TOO_BIG 1000
template <int W>
int run_one_pass(dskgpu_ctx* ctx, bool from_reads, const typename KeyT<W>::T* d_keys_in, u64 nkeys_in, u64 nwords,
                 u32 pass, u32 npass, u64 cap, u64* ns_out, u64* nk_out, Plan* plan_out) {
    typedef typename KeyT<W>::T Key;
    u32* sc = ctx->scalars.as<u32>();
    ctx->sp_rows.valid = false; ctx->sp_rows2.valid = false;
    int extra_bits = 0;
    for (int attempt = 0;; ++attempt) {
        Plan pl;
        // keys the plan is sized for: the exact number of valid windows of a single pass from the reads; of several passes its share
        // + 2 % (cap holds 6 % head-room: sized for it, 7 passes of 200 M reads needed 1792 level-1 bins, more than the LDS of the
        // histogram-free level 1 holds)
        const u64 plan_n = (from_reads && ctx->have_nvalid) ? std::min<u64>(cap, npass == 1 ? ctx->h_nvalid + 1 : ctx->h_nvalid / npass + ctx->h_nvalid / npass / 50 + 4096) : cap;
        if (!make_plan(plan_n, extra_bits, W, (u32)ctx->num_cu, &pl))
            return fail(ctx, DSKGPU_E_OVERFLOW, "cannot partition finer (table overflow persists)");
        pl.d1.world = pl.d2.world = ctx->cfg.world_size; pl.d1.npass = pl.d2.npass = npass; pl.d1.pass = pl.d2.pass = pass;
        // ---------------- level 1
        u32 nch1 = 0;
        const u64 max_chunks1 = (u64)ctx->num_cu * 8;
        // key source: the encoded reads, a key array, or (multi-GPU receive side) super-k-mer records that the
        // histogram-free level-1 scatter reads directly; any other path expands them to a key array first
        bool from_rec = !from_reads && d_keys_in == nullptr;
        auto upload_descs1 = [&]() -> int {
            if (from_reads) build_descs1(ctx, nwords, Tile<W>::WORDS, max_chunks1, &nch1);
            else if (from_rec && ctx->rec_slice_end.size() > 1) build_descs1_slices(ctx, RecTile<W>::NR, max_chunks1, &nch1);
            else if (from_rec) build_descs1(ctx, ctx->rec_n, RecTile<W>::NR, max_chunks1, &nch1);
            else build_descs1(ctx, nkeys_in, Tile<W>::KEYS, max_chunks1, &nch1);
            CK(ctx->descs1.ensure(ctx->h_descs1.size() * sizeof(ChunkDesc)));
            CK(hipMemcpyAsync(ctx->descs1.p, ctx->h_descs1.data(), ctx->h_descs1.size() * sizeof(ChunkDesc),
                              hipMemcpyHostToDevice, ctx->stream));
            return DSKGPU_OK;
        };
        auto records_to_keys = [&]() -> int {        // leave the records path: expand once, then it is a key array
            int e = expand_records<W>(ctx, nkeys_in);
            if (e) return e;
            d_keys_in = ctx->sk_keys.as<Key>(); from_rec = false;
            return upload_descs1();
        };
        { int e = upload_descs1(); if (e) return e; }
        // One-word keys straight from the reads, one pass, two levels: both scatters run without a histogram
        // pass (block-owned slices at level 1, segment-owned regions at level 2); any overflow sends the whole
        // attempt back through the exact histogram + scan path.
        u32 opt_cap = 0;                 // level 2: keys per sub-partition region (0 = exact offsets)
        if (pl.levels == 2 && ctx->sentinel_ok && !ctx->opt2_off && !ctx->tune.no_opt2 && ascatter_lds(W, pl.P2) <= 160 * 1024)
            opt_cap = opt_groups(W) * (8u / W);                                                 // 545 (two-word keys: 1091) groups of 64 B
        if (opt_cap && ctx->tune.opt_cap) opt_cap = ctx->tune.opt_cap;                       // experiments / tests
        if (W > 1 && (u64)pl.F * opt_cap >= 0xFFFF0000ull) opt_cap = 0;                       // k_count<W> keeps 32-bit offsets
        // extension regions behind the home regions (region chains, kernels.h): an eighth of the home regions + 4096; their offsets
        // inside the pool stay below 2^31.  Two-word keys (k_count_mw / k_count_chained_mw index keys with 32 bits): home regions
        // and pool together below 2^32 keys.  Four-word keys: none (tiles of 4096 keys: the scan over the bins dominates anyway).
        u32 max_ext = 0;
        if (opt_cap && W <= 2) {
            u64 want = std::min<u64>((u64)pl.F / 8 + 4096, 0x7FFFFFFFull / opt_cap - 1);
            if (ctx->tune.max_ext >= 0) want = (u64)ctx->tune.max_ext;                                 // tests
            if (W > 1) { const u64 room = 0xFFFF0000ull / opt_cap; want = room > (u64)pl.F + 1 ? std::min<u64>(want, room - pl.F - 1) : 0; }
            max_ext = (u32)want;
        }
        const u64 nregions = (u64)pl.F + max_ext;
        if (!opt_cap) { CK(ctx->bufA.ensure((cap + 1) * sizeof(Key))); CK(ctx->bufB.ensure((cap + 1) * sizeof(Key))); }      // exact offsets: the keys of the pass, twice
        if (opt_cap) {
            // the rows of the solid k-mers land at the region offsets too: abundances (one-word keys: in bufA, the free
            // ping-pong buffer) or keys + abundances (multi-word keys: bufA + abund2).  Size everything BEFORE level 1 writes bufA.
            const u64 slots = (u64)pl.F * opt_cap + ATile<W>::KEYS + 16;
            CK(ctx->bufA.ensure(std::max<u64>(W == 1 ? slots * 4 : slots * sizeof(Key), (cap + 1) * sizeof(Key))));
            if (W > 1) CK(ctx->abund2.ensure(slots * 4));
        }
        bool opt1 = opt_cap && !ctx->opt1_off && !ctx->tune.no_opt1 && (npass == 1 || from_reads || W <= 2);   // several passes over records / a key array (the multi-GPU receive side): one- and two-word keys
        if (from_rec && (!opt1 || W > 2 || ctx->tune.no_recsrc)) { int e = records_to_keys(); if (e) return e; }
        Opt1Spec o1{nullptr, 0u, 0u, sc + SC_OVF1, nullptr, ctx->sk_sp.R, ctx->gstats.as<u64>() + 2, nullptr, nullptr, 0u, nullptr, 0ull, {0ull, 0ull, 0ull, 0ull}, 0u};
        unsigned grid1 = 0;
        u32 nheavy = 0;                  // k-mers the level-2 scatter counts apart (find_heavy)
        // (the per-bin slice ends need 4 more bytes of LDS per bin: plans above 1634 level-1 bins keep UNIFORM slices, mean-sized, no sample)
        const bool uniform1 = scatter_lds(W, pl.P1, true) > 160 * 1024;
        if (opt1 && scatter_lds(W, pl.P1, false) > 160 * 1024) opt1 = false;
        if (opt1 && !from_reads) ctx->h_nvalid = nkeys_in;
        if (opt1) {
            grid1 = scatter_grid(ctx, W, pl.P1, nch1, !uniform1);
            // a block's share of the input: it walks chunks blockIdx, blockIdx + grid, .. (equal chunks, the busiest block has ceil(nch/grid))
            const u64 cpb = (nch1 + grid1 - 1) / grid1;
            double share = (double)cpb / (double)nch1;
            if (from_rec && ctx->rec_slice_end.size() > 1) {      // a launch per slice: the busiest block's chunks of every slice add up
                share = 0.0;
                u64 rb = 0;
                for (size_t sl = 0; sl < ctx->rec_slice_end.size(); ++sl) {
                    const u64 re = ctx->rec_slice_end[sl], nc = ctx->h_slice_chunk[sl + 1] - ctx->h_slice_chunk[sl];
                    if (nc) share += (double)((nc + grid1 - 1) / grid1) / (double)nc * (double)(re - rb) / (double)ctx->rec_n;
                    rb = re;
                }
            }
            // ---- level-1 loads of this pass, per bin ("PartiInfo" before the spill): a positional sample -- the level-1 digit
            // histogram of <= 1024 tiles spread over the source -- scaled to the pass.  Every bin's slices are sized from ITS load,
            // so a bin that holds a repeat family (or poly-A) gets longer slices instead of overflowing the mean-sized ones.
            // Records (multi-GPU receive side) have no histogram kernel: uniform loads.
            const double pass_keys = (double)(ctx->h_nvalid / npass);
            std::vector<double>& load = ctx->h_load;
            load.assign(pl.P1, pass_keys / pl.P1);
            ctx->h_seg_work = load;
            std::vector<double>& spread = ctx->h_spread;
            spread.assign(pl.P1, 0.0);
            bool sampled = false;
            // records (the multi-GPU receive side, the passes of a record-based multi-pass count): a positional sample of the records is
            // expanded into a key array with pads (k_sk_sample_keys: SK_MAXN slots per candidate record) and sampled like any key array.
            // Records that arrive in slices: the sample comes from the first slice (the slices are positional cuts of every sender's
            // reads: alike), so only that slice has to have arrived.
            const Key* d_keys_s = d_keys_in;             // the key array the sample kernels read
            double smp_density = 1.0;                    // records: real keys per slot of the sample array (a level-1 tile is full, a sample tile is not)
            bool rec_sample_ok = ctx->sentinel_ok;       // (the sample array is padded with the all-ones key, which the key-array kernels skip: only when it is no k-mer of this k)
            u64 rec_units = 0;
            if constexpr (W <= 2) {
              if (from_rec && !ctx->tune.no_sample && !uniform1) {
                const u64 nrec_s = ctx->rec_slice_end.size() > 1 ? ctx->rec_slice_end[0] : ctx->rec_n;
                const u64 NR = RecTile<W>::NR;
                const u64 nchk_all = (nrec_s + NR - 1) / NR;
                const u64 nchk = std::min<u64>(nchk_all, 256);
                if (nchk == 0) rec_sample_ok = false;
                else {
                    { const int e = rec_gate_upto(ctx, 1); if (e) return e; }
                    std::vector<u64>& cbeg = ctx->h_cbeg;        // (the context's: it outlives the asynchronous copy below; the next use is behind this pass's next synchronisation)
                    cbeg.resize(nchk);
                    for (u64 i = 0; i < nchk; ++i) cbeg[i] = (i * nchk_all / nchk) * NR;
                    rec_units = nchk * NR * SK_MAXN;
                    CK(ctx->smp_keys.ensure(rec_units * sizeof(Key) + nchk * 8 + 64));
                    u64* d_cbeg = reinterpret_cast<u64*>(ctx->smp_keys.as<char>() + rec_units * sizeof(Key));
                    CK(hipMemcpyAsync(d_cbeg, cbeg.data(), nchk * 8, hipMemcpyHostToDevice, ctx->stream));
                    hipLaunchKernelGGL(k_sk_sample_keys<W>, dim3((unsigned)nchk), dim3(SKX_NT), 0, ctx->stream, ctx->rec_src, ctx->rec_n, ctx->sk_sp.R, (int)ctx->cfg.kmer_size,
                                       (const u64*)d_cbeg, (u32)NR, ctx->smp_keys.as<Key>());
                    CKL("k_sk_sample_keys");
                    d_keys_s = ctx->smp_keys.as<Key>();
                }
              }
            }
            const bool smp_rec = from_rec && rec_units != 0;
            if ((!from_rec || smp_rec) && rec_sample_ok && !ctx->tune.no_sample && !uniform1) {
                const u64 units = from_reads ? nwords : smp_rec ? rec_units : nkeys_in, tile = from_reads ? Tile<W>::WORDS : Tile<W>::KEYS;
                const u64 ntiles = std::max<u64>(1, (units + tile - 1) / tile);
                const u64 nts = std::min<u64>(ntiles, 1024);
                ctx->h_descs_s.resize(nts);
                for (u64 i = 0; i < nts; ++i) {
                    const u64 t = i * ntiles / nts;
                    ChunkDesc d; d.begin = t * tile; d.end = std::min<u64>(units, (t + 1) * tile); d.flat_base = (u32)i; d.stride = (u32)nts;
                    ctx->h_descs_s[i] = d;
                }
                const u64 Ms = (u64)pl.P1 * nts;
                CK(ctx->smp_descs.ensure(nts * sizeof(ChunkDesc)));
                CK(ctx->smp_mat.ensure((Ms + 1) * 4 + (size_t)pl.P1 * 16));
                CK(hipMemcpyAsync(ctx->smp_descs.p, ctx->h_descs_s.data(), nts * sizeof(ChunkDesc), hipMemcpyHostToDevice, ctx->stream));
                ctx->h_sc[SC_NCH_S] = (u32)nts;
                CK(hipMemcpyAsync(sc + SC_NCH_S, &ctx->h_sc[SC_NCH_S], 4, hipMemcpyHostToDevice, ctx->stream));
                int e;
                if (from_reads) e = launch_hist<W, 0>(ctx, nullptr, ctx->smp_descs.as<ChunkDesc>(), sc + SC_NCH_S, nts, ctx->smp_mat.as<u32>(), pl.d1, pl.P1);
                else e = launch_hist<W, 1>(ctx, d_keys_s, ctx->smp_descs.as<ChunkDesc>(), sc + SC_NCH_S, nts, ctx->smp_mat.as<u32>(), pl.d1, pl.P1);
                if (e) return e;
                u64* mom = reinterpret_cast<u64*>(ctx->smp_mat.as<u32>() + ((Ms + 2) & ~(u64)1));
                hipLaunchKernelGGL(k_bin_moments, dim3((pl.P1 + 3) / 4), dim3(256), 0, ctx->stream, (const u32*)ctx->smp_mat.as<u32>(), (u32)nts, pl.P1, mom);
                CKL("k_bin_moments");
                ctx->h_mom.resize((size_t)pl.P1 * 2);
                {
                    void* lz = landing(ctx, (size_t)pl.P1 * 16);
                    CK(hipMemcpyAsync(lz ? lz : (void*)ctx->h_mom.data(), mom, (size_t)pl.P1 * 16, hipMemcpyDeviceToHost, ctx->stream));
                    CK(hipStreamSynchronize(ctx->stream));
                    if (lz) std::memcpy(ctx->h_mom.data(), lz, (size_t)pl.P1 * 16);
                }
                u64 stot = 0;
                for (u32 b = 0; b < pl.P1; ++b) stot += ctx->h_mom[2 * b];
                if (stot >= (u64)pl.P1 * 64) {        // enough sampled keys to say something per bin
                    // one pass from the reads: scaled so that the loads add up to the exact number of valid k-mers; otherwise by position
                    const double scale = (from_reads && npass > 1) ? (double)ntiles / (double)nts : pass_keys / (double)stot;
                    // tiles a block walks (the busiest one), and how far a bin's keys on those tiles may be from share * load:
                    //   the block's own spread: 5 sigma of the sum over its tiles of the per-tile count (variance measured on the sample),
                    //   the estimate's error  : 4 sigma of the sampled sum, scaled to the block's share
                    // (records: a level-1 tile is full, a tile of the sample array holds smp_density of its slots: the block walks
                    //  pass_keys * share / KEYS full tiles, each with 1 / density times the variance of a sample tile)
                    if (smp_rec) smp_density = std::max(0.05, (double)stot / ((double)nts * (double)Tile<W>::KEYS));
                    const double tiles_per_block = smp_rec ? pass_keys * share / (double)Tile<W>::KEYS : (double)ntiles * share;
                    for (u32 b = 0; b < pl.P1; ++b) {
                        const double sum = (double)ctx->h_mom[2 * b], sq = (double)ctx->h_mom[2 * b + 1];
                        const double mean = sum / (double)nts, var = std::max(mean, sq / (double)nts - mean * mean);      // (at least Poisson)
                        load[b] = sum * scale;
                        spread[b] = 5.0 * std::sqrt(tiles_per_block * var / smp_density) + 4.0 * std::sqrt((double)nts * var) * scale * share;
                    }
                    ctx->h_seg_work = load;
                    sampled = true;
                }
                // ---- a k-mer that alone is a large share of a level-1 bin (poly-A reads, a satellite: millions of occurrences):
                // its bin stands far above the others.  Collect sampled keys of those bins, find the dominant k-mer(s) on the host
                // and let the level-1 scatter count them apart (k_scatter<.., HEAVY>) -- everything lighter is what the region
                // chains are for.
                if constexpr (W <= 2) {
                    if (sampled && opt_cap && !ctx->tune.no_heavy) { const int e2 = find_heavy<W>(ctx, from_reads, d_keys_s, (u32)nts, pl, &nheavy); if (e2) return e2; }
                }
                ctx->mark("sample1");
            }
            // slice of bin b = the busiest block's share of its load + the spread above + 1 % + 64 keys; without a sample: + 6 % + 160
            // (what uniform reads need), in whole groups of 8 keys
            ctx->h_boff.resize(pl.P1 + 1);
            u64 area = 0;
            if (uniform1) o1.uslice = 1u;          // (set below)
            for (u32 b = 0; b < pl.P1; ++b) {
                double sl = load[b] * share;
                // (records: the senders' zero-length pad records sit at the ends of their slices, so a level-1 chunk holds anything from
                //  no pads to ~10 % -- a block that walks only a few chunks does not average that out: more room there)
                const double few = (from_rec && share * (double)nch1 < 16.0) ? 0.08 : 0.0;
                sl += sampled ? spread[b] + sl * 0.01 + 64.0 : sl * (0.06 + few) + 160.0;
                u64 slice = ((u64)sl + 8) & ~7ull;
                if (uniform1) o1.uslice = (u32)slice;
                if (ctx->tune.opt_slice) slice = ctx->tune.opt_slice;                                        // experiments / tests
                ctx->h_boff[b] = (u32)std::min<u64>(area, 0xFFFFFFFFull); area += slice;
            }
            ctx->h_boff[pl.P1] = (u32)std::min<u64>(area, 0xFFFFFFFFull);
            const u64 tail = 2 * Tile<W>::KEYS;                         // the dump zone behind the last slice (a tile's keys of a bin that outgrew its slice land there)
            const u64 cells = (u64)pl.P1 * grid1;
            if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] pass %u/%u: P1 %u P2 %u, %u level-1 blocks, area %llu keys per block (%.2f GB of slices), sampled %d, keys %llu, %u chunks, share %.6f, %zu slices\n", pass, npass, pl.P1, pl.P2, grid1,
                                           (unsigned long long)area, (double)area * grid1 * sizeof(Key) * 1e-9, (int)sampled, (unsigned long long)ctx->h_nvalid, nch1, share, ctx->rec_slice_end.size());
            if (area < 8 || area * grid1 + tail >= 0xFFFF0000ull) opt1 = false;
            else {
                o1.area = (u32)area; o1.dump = (u32)(area * grid1);
                CK(ctx->boff.ensure(((size_t)pl.P1 + 1) * 4));
                CK(hipMemcpyAsync(ctx->boff.p, ctx->h_boff.data(), ((size_t)pl.P1 + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
                o1.boff = ctx->boff.as<u32>();
                CK(ctx->mat1.ensure((cells + 1) * 4));                      // here: keys per (bin, block) slice
                o1.fill = ctx->mat1.as<u32>();
                if (grid1 + 1 > SLICED_MAX) opt1 = false;                   // the level-2 loader keeps the slice bounds in LDS
                CK(ctx->bufA.ensure((area * grid1 + tail + 1) * sizeof(Key)));
            }
        }
        if (from_rec && !opt1) { int e = records_to_keys(); if (e) return e; }
        const u64 M1 = (u64)pl.P1 * nch1;
        u32* h_sc = ctx->h_sc;
        std::memset(h_sc, 0, sizeof(ctx->h_sc));
        h_sc[SC_NCH1] = nch1; h_sc[SC_MLEN1] = (u32)M1; h_sc[SC_F] = pl.F;
        if (opt1) h_sc[SC_NCH2] = pl.P1;                       // level-2 chunks = the level-1 bin regions
        h_sc[SC_WORK2] = (u32)std::min<u64>(pl.P1, (u64)ctx->num_cu);   // work counter of the segment-owned level-2 scatter: first segment not taken in the first round
        // (with them: histogram / distinct counters of THIS pass attempt -- a table-overflow retry must not double count)
        {
            static_assert(sizeof(ctx->h_sc) == sizeof(ScalarSet), "scalar block");
            ScalarSet hs; std::memcpy(hs.v, h_sc, sizeof hs.v);
            const u32 nh = ctx->cfg.histo_max + 1;
            u64 nzero = 0;
            if (pl.levels == 2 && opt_cap) { nzero = nregions + 1; CK(ctx->mat2.ensure((size_t)nzero * 4)); }      // level 2's keys per region (home regions, then the extension pool)
            const u64 work = std::max<u64>(nh, nzero);
            hipLaunchKernelGGL(k_setup_pass, dim3((unsigned)std::min<u64>(1024, (work + 255) / 256)), dim3(256), 0, ctx->stream, sc, hs, ctx->ghist.as<u64>(), nh, ctx->gstats.as<u64>(), 4u,
                               nzero ? ctx->mat2.as<u32>() : (u32*)nullptr, nzero);
            CKL("k_setup_pass");
        }
        ctx->mark("setup");
        int rc;
        const ChunkDesc* dd1 = ctx->descs1.as<ChunkDesc>();
        if (opt1) {
            if constexpr (W <= 2) { if (nheavy) { o1.hv_keys = ctx->hv_buf.as<u64>() + HvLayout<W>::keys; o1.hv_cnt = reinterpret_cast<unsigned long long*>(ctx->hv_buf.as<u64>() + HvLayout<W>::counts); } }
            if (from_rec && ctx->rec_slice_end.size() > 1) {
                // the records arrive in slices: one launch per slice, each behind the arrival of its slice (rec_gate), the blocks'
                // write cursors parked in between
                const size_t S = ctx->rec_slice_end.size();
                CK(ctx->cur_state.ensure((size_t)grid1 * pl.P1 * 4));
                o1.cur_state = ctx->cur_state.as<u32>();
                rc = DSKGPU_OK;
                for (size_t sl = 0; sl < S && rc == DSKGPU_OK; ++sl) {
                    if ((rc = rec_gate_upto(ctx, (u32)sl + 1))) break;
                    o1.g0 = ctx->h_slice_chunk[sl]; o1.gn = ctx->h_slice_chunk[sl + 1] - ctx->h_slice_chunk[sl];
                    o1.resume = sl > 0 ? 1u : 0u; o1.last = sl + 1 == S ? 1u : 0u;
                    rc = launch_scatter_rec_h<W>(ctx, nheavy != 0, dd1, sc + SC_NCH1, nch1, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
                }
            }
            else if (from_rec) { if (!(rc = rec_gate_all(ctx))) rc = launch_scatter_rec_h<W>(ctx, nheavy != 0, dd1, sc + SC_NCH1, nch1, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1); }
            else if (nheavy && from_reads) {
                if constexpr (W <= 2) rc = npass > 1 ? launch_scatter_m<W, 0, 3, true, true>(ctx, nullptr, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1)
                                                     : launch_scatter_m<W, 0, 1, true, true>(ctx, nullptr, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
                else rc = DSKGPU_E_STATE;
            } else if (nheavy) {
                if constexpr (W <= 2) rc = npass > 1 ? launch_scatter_m<W, 1, 3, true, true>(ctx, d_keys_in, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1)
                                                     : launch_scatter_m<W, 1, 1, true, true>(ctx, d_keys_in, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
                else rc = DSKGPU_E_STATE;
            }
            else if (from_reads && npass > 1) rc = launch_scatter_m<W, 0, 3, true>(ctx, nullptr, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
            else if (from_reads) rc = launch_scatter_m<W, 0, 1, true>(ctx, nullptr, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
            else if (npass > 1) {
                if constexpr (W <= 2) rc = launch_scatter_m<W, 1, 3, true>(ctx, d_keys_in, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
                else rc = DSKGPU_E_STATE;
            }
            else rc = launch_scatter_m<W, 1, 1, true>(ctx, d_keys_in, dd1, sc + SC_NCH1, nch1, nullptr, ctx->bufA.as<Key>(), pl.d1, pl.P1, o1);
            if (rc) return rc;
            ctx->mark("scatter1");
        } else {
            CK(ctx->mat1.ensure((M1 + 1) * 4));
            if (from_reads) rc = launch_hist<W, 0>(ctx, nullptr, dd1, sc + SC_NCH1, nch1, ctx->mat1.as<u32>(), pl.d1, pl.P1);
            else rc = launch_hist<W, 1>(ctx, d_keys_in, dd1, sc + SC_NCH1, nch1, ctx->mat1.as<u32>(), pl.d1, pl.P1);
            if (rc) return rc;
            ctx->mark("hist1");
            if ((rc = run_scan(ctx, ctx->mat1.as<u32>(), sc + SC_MLEN1, M1))) return rc;
            ctx->mark("scan1");
            if (npass > 1) {      // the pass must fit the buffers sized for it (skewed inputs can overfill one pass)
                CK(hipMemcpyAsync(&ctx->h_back[2], ctx->mat1.as<u32>() + M1, 4, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipStreamSynchronize(ctx->stream));
                if ((u64)ctx->h_back[2] > cap) { ctx->resolve_marks(); return PASS_TOO_BIG; }
            }
            if (from_reads) rc = launch_scatter<W, 0>(ctx, nullptr, dd1, sc + SC_NCH1, nch1, ctx->mat1.as<u32>(), ctx->bufA.as<Key>(), pl.d1, pl.P1);
            else rc = launch_scatter<W, 1>(ctx, d_keys_in, dd1, sc + SC_NCH1, nch1, ctx->mat1.as<u32>(), ctx->bufA.as<Key>(), pl.d1, pl.P1);
            if (rc) return rc;
            ctx->mark("scatter1");
        }
        Key* fkeys = ctx->bufA.as<Key>();
        DevBuf* scratch = &ctx->bufB;
        CK(ctx->fstart.ensure(((size_t)pl.F + 2) * 4));
        CK(ctx->nsolid.ensure(((size_t)pl.F + 2) * 4));
        // ---------------- level 2
        // One-word keys take the segment-owned scatter (k_scatter_al<.., OPT>): no histogram pass, every
        // sub-partition gets a fixed region of OPT_CAP keys; a region that overflows (heavy repeats) sends the
        // level back through the exact histogram + scan path, and the context remembers it for these reads.
        if (pl.levels == 2 && opt_cap) {
            CK(ctx->bufB.ensure((nregions * opt_cap + ATile<W>::KEYS + 16) * sizeof(Key)));
            CK(ctx->descs2.ensure(((size_t)pl.P1 * 2 + 1) * sizeof(ChunkDesc)));
            CK(ctx->seg.ensure((size_t)pl.P1 * sizeof(SegInfo)));
            CK(ctx->mat2.ensure(((size_t)nregions + 1) * 4));                  // here: keys per region (home regions, then the extension pool)
            CK(ctx->chain_next.ensure(((size_t)nregions + 1 + max_ext + 1) * 4));   // links (only read where subcnt has its chain bit set), then the list of chained sub-partitions
            // (zeroed by k_setup_pass)
            if (opt1) {      // segments = the level-1 bin regions (slices + sentinel tails)
                ctx->h_descs2.resize(pl.P1);
                // heaviest segments first (the kernel hands them out by a work counter): a segment that holds a repeat family takes a
                // block longer than the others, so it must not be the last thing a block starts
                std::vector<u32> order(pl.P1);
                for (u32 sgm = 0; sgm < pl.P1; ++sgm) order[sgm] = sgm;
                // (in steps of 5 % of the mean, stable: the segments of uniform reads keep their natural order, neighbours in memory run together)
                double mean_work = 0.0; for (double w : ctx->h_seg_work) mean_work += w; mean_work = std::max(1.0, mean_work / pl.P1);
                auto wclass = [&](u32 a) { return (long long)(ctx->h_seg_work[a] / (0.05 * mean_work)); };
                std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return wclass(a) > wclass(b); });
                for (u32 i = 0; i < pl.P1; ++i) {
                    const u32 sgm = order[i];
                    ChunkDesc d; d.begin = ctx->h_boff[sgm]; d.end = ctx->h_boff[sgm + 1]; d.flat_base = sgm * pl.P2; d.stride = 1;   // slice i of the segment: + i * area
                    ctx->h_descs2[i] = d;
                }
                CK(hipMemcpyAsync(ctx->descs2.p, ctx->h_descs2.data(), (size_t)pl.P1 * sizeof(ChunkDesc), hipMemcpyHostToDevice, ctx->stream));
            } else {
                hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, ctx->stream, ctx->mat1.as<u32>(), nch1, pl.P1, 0x7FFFFFFFu, pl.P2,
                                   ctx->seg.as<SegInfo>(), ctx->descs2.as<ChunkDesc>(), sc + SC_NCH2, sc + SC_MLEN2, 1u);
                CKL("k_plan");
            }
            ctx->mark("plan2");
            OptSpec os{opt_cap, ctx->mat2.as<u32>(), sc + SC_OVF2, opt1 ? o1.fill : nullptr, 0u, grid1, (u64)o1.area,
                       pl.F, max_ext, ctx->chain_next.as<u32>(), sc + SC_EXT, ctx->chain_next.as<u32>() + nregions + 1, sc + SC_NCHAINED,
                       sc + SC_WORK2, nullptr};
            if (ctx->tune.verbose && opt1) { CK(ctx->dbg.ensure((size_t)pl.P1 * 24)); os.dbg = ctx->dbg.as<unsigned long long>(); }
            if (opt1) rc = launch_scatter_al<W, 2, true, true>(ctx, ctx->bufA.as<Key>(), ctx->descs2.as<ChunkDesc>(), sc + SC_NCH2, (u64)pl.P1 * 2, nullptr,
                                                               ctx->bufB.as<Key>(), pl.d2, pl.P2, os);
            else rc = launch_scatter_al<W, 2, true, false>(ctx, ctx->bufA.as<Key>(), ctx->descs2.as<ChunkDesc>(), sc + SC_NCH2, (u64)pl.P1 * 2, nullptr,
                                                           ctx->bufB.as<Key>(), pl.d2, pl.P2, os);
            if (rc) return rc;
            ctx->mark("scatter2");
            if (ctx->tune.verbose && opt1) {      // per-segment times of the level-2 scatter, in hand-out order
                std::vector<unsigned long long> t((size_t)pl.P1 * 3);
                CK(hipMemcpyAsync(t.data(), ctx->dbg.p, t.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
                CK(hipStreamSynchronize(ctx->stream));
                unsigned long long t0 = ~0ull, t1 = 0; for (u32 i = 0; i < pl.P1; ++i) { t0 = std::min(t0, t[3 * i]); t1 = std::max(t1, t[3 * i + 1]); }
                fprintf(stderr, "[dskgpu] level 2: %u segments, %.3f ms first start -> last end\n", pl.P1, (t1 - t0) * 1e-5);
                for (u32 i = 0; i < pl.P1; ++i)
                    if (i < 6 || i + 3 >= pl.P1 || t[3 * i + 1] + 20000 > t1)
                        fprintf(stderr, "[dskgpu]   desc %u (segment %u, load %.0f) block %llu: %.3f .. %.3f ms\n", i, (u32)(ctx->h_descs2[i].flat_base / pl.P2),
                                ctx->h_load[ctx->h_descs2[i].flat_base / pl.P2], t[3 * i + 2], (t[3 * i] - t0) * 1e-5, (t[3 * i + 1] - t0) * 1e-5);
            }
            fkeys = ctx->bufB.as<Key>();
            scratch = &ctx->bufA;
        } else if (pl.levels == 2) {
            const u64 max_chunks2 = cap / CH2 + pl.P1 + 1;
            const u64 M2 = max_chunks2 * pl.P2;
            if (M2 >= 0xFFFFFFFFull) return fail(ctx, DSKGPU_E_ARG, "level-2 matrix too large");
            CK(ctx->descs2.ensure(max_chunks2 * sizeof(ChunkDesc)));
            CK(ctx->seg.ensure((size_t)pl.P1 * sizeof(SegInfo)));
            CK(ctx->mat2.ensure((M2 + 1) * 4));
            hipLaunchKernelGGL(k_plan, dim3(1), dim3(1024), 0, ctx->stream, ctx->mat1.as<u32>(), nch1, pl.P1, CH2, pl.P2,
                               ctx->seg.as<SegInfo>(), ctx->descs2.as<ChunkDesc>(), sc + SC_NCH2, sc + SC_MLEN2, 0u);
            CKL("k_plan");
            ctx->mark("plan2");
            const ChunkDesc* dd2 = ctx->descs2.as<ChunkDesc>();
            if ((rc = launch_hist<W, 1>(ctx, ctx->bufA.as<Key>(), dd2, sc + SC_NCH2, max_chunks2, ctx->mat2.as<u32>(), pl.d2, pl.P2))) return rc;
            ctx->mark("hist2");
            if ((rc = run_scan(ctx, ctx->mat2.as<u32>(), sc + SC_MLEN2, M2))) return rc;
            ctx->mark("scan2");
            if ((rc = launch_scatter<W, 1>(ctx, ctx->bufA.as<Key>(), dd2, sc + SC_NCH2, max_chunks2, ctx->mat2.as<u32>(), ctx->bufB.as<Key>(), pl.d2, pl.P2))) return rc;
            ctx->mark("scatter2");
            fkeys = ctx->bufB.as<Key>();
            scratch = &ctx->bufA;
            hipLaunchKernelGGL(k_final_offsets, dim3((pl.F + 256) / 256), dim3(256), 0, ctx->stream, ctx->mat2.as<u32>(),
                               ctx->seg.as<SegInfo>(), pl.P2, 0u, sc + SC_MLEN2, ctx->fstart.as<u32>(), pl.F);
        } else {
            hipLaunchKernelGGL(k_final_offsets, dim3((pl.F + 256) / 256), dim3(256), 0, ctx->stream, ctx->mat1.as<u32>(),
                               (const SegInfo*)nullptr, pl.P1, nch1, sc + SC_MLEN1, ctx->fstart.as<u32>(), pl.F);
        }
        CKL("k_final_offsets");
        ctx->mark("offsets");
        // ---------------- count: one-word keys write solid rows in place (+ abundance into the
        // free ping-pong buffer); two-word keys write rows into the free buffer (+ abund2)
        CountParams cp;
        cp.F = pl.F;
        cp.maxload = W == 1 ? CNT_MAXLOAD : C2_MAXLOAD;
        if (ctx->tune.table_maxload) cp.maxload = std::min<u32>(cp.maxload, ctx->tune.table_maxload);
        cp.amin = ctx->cfg.abundance_min; cp.amax = ctx->cfg.abundance_max; cp.histo_max = ctx->cfg.histo_max;
        cp.cap = opt_cap; cp.subcnt = opt_cap ? ctx->mat2.as<u32>() : nullptr;
        const unsigned cgrid = (unsigned)std::min<u64>(pl.F, (u64)ctx->num_cu * 2);
        Key* solid_keys = W == 1 ? fkeys : scratch->as<Key>();
        u32* solid_ab = W == 1 ? scratch->as<u32>() : ctx->abund2.as<u32>();
        // the count stage as one unit: two-word keys may run it twice (k_count2v3, then -- when its verification bit went up -- k_count_mw:
        // the keys are still there, two-word rows go to the free buffer)
        auto count_stage = [&]() -> int {
            launch_count<W>(ctx, cgrid, fkeys, solid_keys, solid_ab, sc + SC_OVERFLOW, cp);
            CKL("k_count");
            if constexpr (W == 2) {
                if (opt_cap && max_ext) {
                    hipLaunchKernelGGL(k_count_chained_mw<2>, dim3((unsigned)std::min<u64>(max_ext, (u64)ctx->num_cu * 2)), dim3(CNT_NT), 0, ctx->stream, (const Key*)fkeys, solid_keys, solid_ab,
                                       ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), sc + SC_OVERFLOW, cp, (const u32*)cp.subcnt,
                                       (const u32*)ctx->chain_next.as<u32>(), (const u32*)(ctx->chain_next.as<u32>() + nregions + 1), (const u32*)(sc + SC_NCHAINED), max_ext);
                    CKL("k_count_chained_mw");
                }
            }
            if constexpr (W == 1) {
                if (opt_cap && max_ext) {      // the sub-partitions that went on in extension regions (none on repeat-free reads: the blocks leave at once)
                    hipLaunchKernelGGL(k_count_chained, dim3((unsigned)std::min<u64>(max_ext, (u64)ctx->num_cu * 2)), dim3(CNT_NT), 0, ctx->stream, fkeys, solid_keys, solid_ab,
                                       ctx->nsolid.as<u32>(), ctx->ghist.as<u64>(), ctx->gstats.as<u64>(), sc + SC_OVERFLOW, cp, (const u32*)cp.subcnt,
                                       (const u32*)ctx->chain_next.as<u32>(), (const u32*)(ctx->chain_next.as<u32>() + nregions + 1), (const u32*)(sc + SC_NCHAINED), max_ext);
                    CKL("k_count_chained");
                }
            }
            if constexpr (W <= 2) {
              if (nheavy) {      // the k-mers the level-1 scatter counted apart: histogram, distinct count, rows (appended behind the compacted ones below)
                const u32 slots = HV_KEYS;
                u64* hvb = ctx->hv_buf.as<u64>();
                hipLaunchKernelGGL(k_heavy_rows<W>, dim3((slots + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const Key*>(hvb + HvLayout<W>::keys),
                                   (const unsigned long long*)(hvb + HvLayout<W>::counts), slots, cp.amin, cp.amax, cp.histo_max, ctx->ghist.as<u64>(), ctx->gstats.as<u64>(),
                                   hvb + HvLayout<W>::rows, reinterpret_cast<u32*>(hvb + HvLayout<W>::ab));
                CKL("k_heavy_rows");
              }
            }
            return DSKGPU_OK;
        };
        // count, scan of the solid rows per sub-partition, sizes back to the host (one sync)
        auto count_and_sizes = [&]() -> int {
            if (int e = count_stage()) return e;
            ctx->mark("count");
            if (int e = run_scan(ctx, ctx->nsolid.as<u32>(), sc + SC_F, pl.F)) return e;
            ctx->mark("scan_solid");
            // (k-mers of the pass: the last sub-partition offset, or -- fixed-capacity regions -- the level-1 total; with block-owned slices
            //  the scatter's own count in gstats[2])
            CK(ctx->back_dev.ensure(16 * 8));
            if (!ctx->back_host) CK(hipHostMalloc(reinterpret_cast<void**>(&ctx->back_host), 16 * 8, hipHostMallocDefault));
            const u32* nkp = opt1 ? nullptr : (opt_cap ? ctx->mat1.as<u32>() + M1 : ctx->fstart.as<u32>() + pl.F);
            hipLaunchKernelGGL(k_gather_back, dim3(1), dim3(64), 0, ctx->stream, (const u32*)sc, (const u32*)(ctx->nsolid.as<u32>() + pl.F), nkp,
                               (const u64*)ctx->gstats.as<u64>(), ctx->back_dev.as<u64>());
            CKL("k_gather_back");
            CK(hipMemcpyAsync(ctx->back_host, ctx->back_dev.p, 10 * 8, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            const u64* bh = ctx->back_host;
            ctx->h_back[0] = (u32)bh[0]; ctx->h_back[1] = (u32)bh[1]; if (!opt1) ctx->h_back[2] = (u32)bh[2];
            ctx->h_ovf2 = (u32)bh[3]; ctx->h_ovf1 = (u32)bh[4]; ctx->h_ext = (u32)bh[5];
            for (int x = 0; x < 4; ++x) ctx->h_stats[x] = bh[6 + x];
            return DSKGPU_OK;
        };
        if constexpr (W == 2) CK(hipMemcpyAsync(ctx->gstats.as<u64>() + 3, ctx->gstats.as<u64>() + 2, 8, hipMemcpyDeviceToDevice, ctx->stream));      // (the keys level 1 placed, before k_heavy_rows adds to them: see below)
        if ((rc = count_and_sizes())) return rc;
        if constexpr (W == 2) {
            // k_count2v3 keys its table by the mixed top word alone and checks every key's low word afterwards.  Two different k-mers of the pass with
            // the same top word (birthday bound of a 64-bit hash: n^2 / 2^65 -- 0.5 % of the runs at 4 * 10^8 distinct k-mers), or one whose top word
            // is the empty-slot value: the COUNT STAGE runs again with the index-table kernel (the keys are untouched: two-word rows go to the other
            // buffer), and the rest of the reads' passes use that kernel too.  Tests craft both cases.
            if ((ctx->h_back[0] & (CNT_OVF_VERIFY | CNT_OVF_SENTINEL)) && !((opt_cap && ctx->h_ovf2) || (opt1 && ctx->h_ovf1))) {
                if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] pass %u/%u: the top-word table met k-mers it cannot tell apart (flags %u): counting again with k_count_mw\n", pass, npass, ctx->h_back[0]);
                ctx->mw_v3_off = true;
                ctx->stats.n_retries += 1;
                CK(hipMemsetAsync(ctx->ghist.p, 0, ((size_t)ctx->cfg.histo_max + 1) * 8, ctx->stream));
                CK(hipMemsetAsync(ctx->gstats.p, 0, 2 * 8, ctx->stream));
                CK(hipMemcpyAsync(ctx->gstats.as<u64>() + 2, ctx->gstats.as<u64>() + 3, 8, hipMemcpyDeviceToDevice, ctx->stream));
                CK(hipMemsetAsync(sc + SC_OVERFLOW, 0, 4, ctx->stream));
                if ((rc = count_and_sizes())) return rc;
            }
        }
        const u32 h_ovf = ctx->h_back[0], h_nsolid = ctx->h_back[1], h_nk = opt1 ? (u32)ctx->h_stats[2] : ctx->h_back[2];
        if ((opt_cap && ctx->h_ovf2) || (opt1 && ctx->h_ovf1)) {       // a slice / region overflowed: repeat this attempt with exact offsets
            ctx->resolve_marks();
            if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] pass %u/%u: %s overflowed: the exact path takes over\n", pass, npass, (opt1 && ctx->h_ovf1) ? "a level-1 slice" : "the level-2 extension pool");
            if (opt1 && ctx->h_ovf1) ctx->opt1_off = true;
            else { ctx->opt2_off = true; ctx->opt1_off = true; }     // the exact level 2 cannot read sentinel-padded slices
            ctx->stats.n_retries += 1;
            ctx->mark("start");
            --attempt;
            continue;
        }
        if (h_ovf & 1u) {
            ctx->resolve_marks();
            if (attempt >= 3) return fail(ctx, DSKGPU_E_OVERFLOW, "hash table overflow after 3 retries");
            extra_bits += 1;
            ctx->stats.n_retries += 1;
            ctx->mark("start");
            continue;
        }
        // ---------------- dense rows of this pass
        const u64 nhs = nheavy ? ctx->h_stats[1] : 0;                    // solid rows of the k-mers counted apart
        const u64 ns = h_nsolid + nhs;
        // (a pass of a multi-pass job whose accumulators have room: the rows go there directly -- no copy of 0.7 GB per pass afterwards)
        const bool to_sink = ctx->sink.active && ctx->sink.rows + ns + 1 <= ctx->sink.cap;
        RowsOut ro{};
        u32* rows_ab = nullptr;
        if (to_sink) {
            rows_ab = ctx->sink.ab + ctx->sink.rows;
            for (int x = 0; x < W; ++x) ro.w[x] = ctx->sink.w[x] + ctx->sink.rows;
            ctx->sink.took = true;
        } else {
            CK(ctx->out_ab.ensure((ns + 1) * 4));
            rows_ab = ctx->out_ab.as<u32>();
            for (int x = 0; x < W; ++x) { CK(ctx->out_w[x].ensure((ns + 1) * 8)); ro.w[x] = ctx->out_w[x].as<u64>(); }
        }
        if constexpr (W <= 2) {
          if (nhs) {
            for (int x = 0; x < W; ++x)
                CK(hipMemcpyAsync(ro.w[x] + h_nsolid, ctx->hv_buf.as<u64>() + HvLayout<W>::rows + (size_t)x * HV_KEYS, nhs * 8, hipMemcpyDeviceToDevice, ctx->stream));
            CK(hipMemcpyAsync(rows_ab + h_nsolid, ctx->hv_buf.as<u64>() + HvLayout<W>::ab, nhs * 4, hipMemcpyDeviceToDevice, ctx->stream));
          }
        }
        ctx->stats.n_heavy += nheavy;
        // One-word rows of a single pass that the hand-written MSD sort will order: its first step reads them where they lie (the regions /
        // exact ranges of the count kernel + the few rows of the k-mers counted apart as a dense tail) -- no dense copy is made first
        // (k_compact: 0.5 GB read + 0.5 GB written, 0.30 ms of a 14 ms step).  Several passes accumulate dense rows as before.
        bool sparse_sort = false;
        if constexpr (W == 1) {
            const u64 rs_max = ctx->tune.rs_max_rows ? std::min<u64>(ctx->tune.rs_max_rows, RS_MAX_ROWS) : RS_MAX_ROWS;
            sparse_sort = npass == 1 && ctx->job_passes == 1 && ns > 0 && ns <= rs_max && !(ctx->cfg.flags & DSKGPU_F_NO_SORT) &&
                          !ctx->tune.rs_slab_rows && !ctx->bank_job.active;
            if (sparse_sort) {
                ctx->sp_rows.valid = true;
                ctx->sp_rows.s = RsSparse{(const u64*)solid_keys, (const u32*)solid_ab, (const u32*)ctx->nsolid.as<u32>(), (const u32*)ctx->fstart.as<u32>(), opt_cap, pl.F, 0u};
                ctx->sp_rows.n_sparse = h_nsolid;
                ctx->sp_rows.n_tail = (u32)nhs;
                ctx->sp_rows.tail_k = nhs ? ro.w[0] + h_nsolid : nullptr;        // (copied there above: dense, already un-mixed)
                ctx->sp_rows.tail_v = nhs ? rows_ab + h_nsolid : nullptr;
            }
        }
        if constexpr (W == 2) {      // (two-word rows: the same, through rowsort2.h's sparse step A -- sort_rows2_msd is what sort_rows picks under these conditions)
            const u64 rs_max = ctx->tune.rs_max_rows ? std::min<u64>(ctx->tune.rs_max_rows, RS_MAX_ROWS) : RS_MAX_ROWS;
            sparse_sort = npass == 1 && ctx->job_passes == 1 && ns > 0 && ns <= rs_max && !(ctx->cfg.flags & DSKGPU_F_NO_SORT) &&
                          !ctx->tune.rs_slab_rows && !ctx->bank_job.active && 2u * ctx->cfg.kmer_size > 64u;
            if (sparse_sort) {
                ctx->sp_rows2.valid = true;
                ctx->sp_rows2.s = Rs2Sparse{(const K2*)solid_keys, (const u32*)solid_ab, (const u32*)ctx->nsolid.as<u32>(), (const u32*)ctx->fstart.as<u32>(), opt_cap, pl.F, 0u};
                ctx->sp_rows2.n_sparse = h_nsolid;
                ctx->sp_rows2.n_tail = (u32)nhs;
                ctx->sp_rows2.tail = nhs ? Rows2C{ro.w[1] + h_nsolid, ro.w[0] + h_nsolid, rows_ab + h_nsolid} : Rows2C{nullptr, nullptr, nullptr};
            }
        }
        bool mp_part = false;
        if constexpr (W <= 2) {      // DSKGPU_F_PARTITION_ORDER in a multi-pass count: the pass's rows ordered partition by partition on their way into the dense arrays
            mp_part = !sparse_sort && ctx->job_passes > 1 && ctx->mp_part_ok && ns > 0 && h_nsolid < 0xFFFF0000ull && (W == 1 || 2u * ctx->cfg.kmer_size > 64u);
            if (!sparse_sort && ctx->job_passes > 1 && ns > 0 && !mp_part) ctx->mp_part_ok = false;      // (one pass outside the scheme: the job keeps the global order)
            if (mp_part) {
                dskgpu_ctx::SparseRows spr{}; dskgpu_ctx::SparseRows2 spr2{};
                if constexpr (W == 1) {
                    spr.s = RsSparse{(const u64*)solid_keys, (const u32*)solid_ab, (const u32*)ctx->nsolid.as<u32>(), (const u32*)ctx->fstart.as<u32>(), opt_cap, pl.F, 0u};
                    spr.n_sparse = h_nsolid; spr.n_tail = (u32)nhs; spr.tail_k = nhs ? ro.w[0] + h_nsolid : nullptr; spr.tail_v = nhs ? rows_ab + h_nsolid : nullptr;
                } else {
                    spr2.s = Rs2Sparse{(const K2*)solid_keys, (const u32*)solid_ab, (const u32*)ctx->nsolid.as<u32>(), (const u32*)ctx->fstart.as<u32>(), opt_cap, pl.F, 0u};
                    spr2.n_sparse = h_nsolid; spr2.n_tail = (u32)nhs;
                    spr2.tail = nhs ? Rows2C{ro.w[1] + h_nsolid, ro.w[0] + h_nsolid, rows_ab + h_nsolid} : Rows2C{nullptr, nullptr, nullptr};
                }
                const u32 np_est = part_sort_nparts(W, pl.F, h_nsolid, (u32)nhs);
                if (ctx->mp_part_off.ensure_keep(((size_t)ctx->mp_off_used + np_est + 2) * 4, (size_t)ctx->mp_off_used * 4, ctx->stream)) return fail(ctx, DSKGPU_E_NOMEM, "partition offsets");
                u32 np = 0;
                const int prc = launch_part_sort(ctx, W, spr, spr2, ro.w[0], rows_ab, Rows2{W == 2 ? ro.w[1] : nullptr, ro.w[0], rows_ab},
                                                 ctx->mp_part_off.as<u32>() + ctx->mp_off_used, ctx->mp_flag.as<u32>(), &np, nullptr);
                if (prc) return prc;
                ctx->mp_parts.push_back(dskgpu_ctx::MpPart{0ull, np, ctx->mp_off_used});      // (row_base: the caller knows where the pass's rows start in the job)
                ctx->mp_off_used += np + 1;
            }
        }
        if (!sparse_sort && !mp_part) {
            hipLaunchKernelGGL(k_compact<W>, dim3((pl.F + 3) / 4), dim3(256), 0, ctx->stream, (const Key*)solid_keys, (const u32*)solid_ab,
                               ctx->fstart.as<u32>(), ctx->nsolid.as<u32>(), pl.F, ro, rows_ab, opt_cap);
            CKL("k_compact");
        }
        ctx->mark("compact");
        ctx->stats.n_ext_regions += std::min<u32>(ctx->h_ext, max_ext);
        *ns_out = ns; *nk_out = h_nk; *plan_out = pl;
        return DSKGPU_OK;
    }
}

// "Level 0" of a multi-pass count from reads (one-word keys): ONE sweep over the encoded reads writes the mixed keys of passes
// [lo, lo + G) of npass into ctx->l0buf, grouped by pass -- the histogram-free scatter with the pass as its digit (MODE 4): every
// (block, pass) pair owns a slice inside the pass's region, the unused tails are padded with the sentinel, so a region is one key
// array.  The passes of the group then run from those arrays (run_one_pass with a key source) instead of re-generating every k-mer
// once per pass -- the in-HBM counterpart of DSK writing every k-mer to its partition file ONCE (doc/paper.tex:65-67;
// README.md:126-130 asks for few passes because each one re-reads the input: a sweep here is what a pass is there).
// G = as many passes as the free HBM holds next to what a pass itself needs and `reserve_bytes` (rows still to come), at most
// L0_MAX_PASSES.  The slices of pass i are sized from ITS sampled load and its measured spread (a positional sample of <= 1024
// tiles, as at level 1): the pass that holds a k-mer with 10^8 occurrences (poly-A reads of a 30x human run) gets longer slices
// instead of sending its whole group back to the reads.
// -> region_keys[i]: keys (pads included) of pass lo + i's array; obase[i]: its offset in l0buf.  DSKGPU_OK with *G_out = 0: no room.
int level0_materialise(dskgpu_ctx* ctx, u64 nwords, u32 lo, u32 npass, u64 reserve_bytes, u32* G_out, u64 (&region_keys)[L0_MAX_PASSES], u64 (&obase)[L0_MAX_PASSES]) {
    *G_out = 0;
    const u64 nper = ctx->h_nvalid / npass + 1;
    u32 nch1 = 0;
    build_descs1(ctx, nwords, Tile<1>::WORDS, (u64)ctx->num_cu * 8, &nch1);
    // (k_level0 keeps nothing but its cursors in LDS: two blocks per CU, 32 waves, hide each other's atomics and stores)
    const unsigned grid = (unsigned)std::max<u64>(1, std::min<u64>(nch1, (u64)ctx->num_cu));
    const u64 cpb = (nch1 + grid - 1) / grid;
    const double share = (double)cpb / (double)nch1;                                // the busiest block's share of the chunks
    const u64 tail = 2 * Tile<1>::KEYS;
    const u64 uslice = (((u64)((double)nper * share * 1.02) + 4096) + 7) & ~7ull;   // hash-uniform passes: 2 % + 4096 over the busiest block's share
    const u64 uregion = uslice * grid + tail;                                       // keys; + the dump zone
    if (uregion >= 0xFFFF0000ull) return DSKGPU_OK;
    // how many passes fit beside what a pass itself needs (slices, regions + pool, rows): about 30 bytes per key of a pass
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    const u64 have = ctx->bufA.cap + ctx->bufB.cap + ctx->l0buf.cap;
    const u64 need = nper * 30ull + (4ull << 30) + reserve_bytes;
    const u64 room = free_b + have > need ? free_b + have - need : 0;
    u32 G = (u32)std::min<u64>(std::min<u64>(L0_MAX_PASSES, npass - lo), room / (uregion * 8));
    if (ctx->tune.l0_passes) G = std::min<u32>(std::min<u32>(ctx->tune.l0_passes, L0_MAX_PASSES), npass - lo);      // tests
    if (G < 2 && !(ctx->tune.l0_passes)) return DSKGPU_OK;                          // one pass at a time gains nothing over reading the reads
    if (G == 0) return DSKGPU_OK;
    CK(ctx->descs1.ensure(ctx->h_descs1.size() * sizeof(ChunkDesc)));
    CK(hipMemcpyAsync(ctx->descs1.p, ctx->h_descs1.data(), ctx->h_descs1.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice, ctx->stream));
    u32* sc = ctx->scalars.as<u32>();
    DigitSpec ds{4u, G, 0u, ctx->cfg.world_size, npass, lo};
    // ---- the passes' loads in this sweep, sampled
    std::vector<u64> slice(G, uslice);
    if (!ctx->tune.no_sample) {
        const u64 ntiles = std::max<u64>(1, (nwords + Tile<1>::WORDS - 1) / Tile<1>::WORDS);
        const u64 nts = std::min<u64>(ntiles, 1024);
        ctx->h_descs_s.resize(nts);
        for (u64 i = 0; i < nts; ++i) {
            const u64 t = i * ntiles / nts;
            ChunkDesc d; d.begin = t * Tile<1>::WORDS; d.end = std::min<u64>(nwords, (t + 1) * Tile<1>::WORDS); d.flat_base = (u32)i; d.stride = (u32)nts;
            ctx->h_descs_s[i] = d;
        }
        const u64 Ms = (u64)G * nts;
        CK(ctx->smp_descs.ensure(nts * sizeof(ChunkDesc)));
        CK(ctx->smp_mat.ensure((Ms + 4) * 4 + (size_t)G * 16));
        CK(hipMemcpyAsync(ctx->smp_descs.p, ctx->h_descs_s.data(), nts * sizeof(ChunkDesc), hipMemcpyHostToDevice, ctx->stream));
        ctx->h_sc[SC_NCH_S] = (u32)nts;
        CK(hipMemcpyAsync(sc + SC_NCH_S, &ctx->h_sc[SC_NCH_S], 4, hipMemcpyHostToDevice, ctx->stream));
        { const int e = launch_hist_m<1, 0, 4>(ctx, nullptr, ctx->smp_descs.as<ChunkDesc>(), sc + SC_NCH_S, nts, ctx->smp_mat.as<u32>(), ds, G); if (e) return e; }
        u64* mom = reinterpret_cast<u64*>(ctx->smp_mat.as<u32>() + ((Ms + 2) & ~(u64)1));
        hipLaunchKernelGGL(k_bin_moments, dim3((G + 3) / 4), dim3(256), 0, ctx->stream, (const u32*)ctx->smp_mat.as<u32>(), (u32)nts, G, mom);
        CKL("k_bin_moments");
        ctx->h_mom.resize((size_t)G * 2);
        CK(hipMemcpyAsync(ctx->h_mom.data(), mom, (size_t)G * 16, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        u64 stot = 0;
        for (u32 b = 0; b < G; ++b) stot += ctx->h_mom[2 * b];
        if (stot >= (u64)G * 4096) {
            const double scale = (double)ntiles / (double)nts, tiles_per_block = (double)ntiles * share;
            for (u32 b = 0; b < G; ++b) {
                const double sum = (double)ctx->h_mom[2 * b], sq = (double)ctx->h_mom[2 * b + 1];
                const double mean = sum / (double)nts, var = std::max(mean, sq / (double)nts - mean * mean);
                const double sl = sum * scale * share * 1.01 + 5.0 * std::sqrt(tiles_per_block * var) + 4.0 * std::sqrt((double)nts * var) * scale * share + 64.0;
                slice[b] = ((u64)sl + 8) & ~7ull;
            }
        }
        ctx->mark("sample0");
    }
    // regions, largest G whose regions fit the room
    for (;;) {
        u64 tot = 0; bool ok = true;
        for (u32 b = 0; b < G; ++b) { const u64 region = slice[b] * grid + tail; if (region >= 0xFFFF0000ull) ok = false; obase[b] = tot; region_keys[b] = slice[b] * grid; tot += region; }
        if (ok && (tot * 8 <= room || ctx->tune.l0_passes)) { CK(ctx->l0buf.ensure(tot * 8 + 64)); break; }
        if (--G < 2) return DSKGPU_OK;
    }
    ds.pa = G;
    ctx->h_sc[SC_NCH1] = nch1; ctx->h_sc[SC_OVF1] = 0;
    CK(hipMemcpyAsync(sc + SC_NCH1, &ctx->h_sc[SC_NCH1], 4, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemsetAsync(sc + SC_OVF1, 0, 4, ctx->stream));
    // device arrays of the sweep: [slice length per bin: u32 x (L0_MAX_PASSES + 1)] [region base per bin: u64 x L0_MAX_PASSES]
    ctx->h_boff.assign(L0_MAX_PASSES + 1 + 3 + 2 * L0_MAX_PASSES, 0);
    u64 hob[L0_MAX_PASSES];
    for (u32 i = 0; i < L0_MAX_PASSES; ++i) { ctx->h_boff[i] = (u32)slice[std::min(i, G - 1)]; hob[i] = obase[std::min(i, G - 1)]; }
    std::memcpy(&ctx->h_boff[L0_MAX_PASSES + 4], hob, sizeof hob);          // (at byte 80: 8-byte aligned)
    CK(ctx->boff.ensure(ctx->h_boff.size() * 4));
    CK(hipMemcpyAsync(ctx->boff.p, ctx->h_boff.data(), ctx->h_boff.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    const u32* d_slen = ctx->boff.as<u32>();
    const u64* d_obase = reinterpret_cast<const u64*>(ctx->boff.as<u32>() + L0_MAX_PASSES + 4);
    hipLaunchKernelGGL(k_level0, dim3(grid), dim3(SC_NT), 0, ctx->stream, (const u64*)ctx->packed.as<u64>(), (const u32*)ctx->inval.as<u32>(), (const ChunkDesc*)ctx->descs1.as<ChunkDesc>(),
                       (const u32*)(sc + SC_NCH1), ctx->l0buf.as<u64>(), (int)ctx->cfg.kmer_size, ds, G, d_slen, d_obase, sc + SC_OVF1);
    CKL("k_level0");
    ctx->mark("level0");
    CK(hipMemcpyAsync(&ctx->h_ovf1, sc + SC_OVF1, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (ctx->tune.verbose) {
        u64 tot = 0; for (u32 b = 0; b < G; ++b) tot += region_keys[b];
        fprintf(stderr, "[dskgpu] level 0: passes %u..%u of %u materialised, %.2f GB in all (%.2f GB free before)%s\n", lo, lo + G - 1, npass, (double)tot * 8e-9, (double)free_b * 1e-9,
                ctx->h_ovf1 ? " -- a slice overflowed: these passes read the reads" : "");
    }
    if (ctx->h_ovf1) return DSKGPU_OK;                                             // (a skewed key space beyond the sampled spread: the passes of this group re-generate their keys)
    *G_out = G;
    return DSKGPU_OK;
}

// ---- level 0 of a multi-pass count as super-k-mer RECORDS: the passes are "virtual owners"
// What the multi-GPU step does between GPUs, one GPU does between its passes: the pass of a k-mer is the OWNER that the minimizer
// repartition gives its window (owner = table[bucket of the minimizer], G owners = G passes; heavy buckets are split by k-mer),
// a sweep over the 2-bit reads writes the records of as many owners as HBM holds (k_sk_scatter with an owner window, every owner
// with its own slice length and region), and every pass then runs its level 1 straight from its records (k_scatter<W, 2, 1>),
// sized from a sample of them.  Records are 2.3-2.5 bytes per k-mer where a key array takes 8 (16 for two-word keys): a 90 Gbp
// input goes through in 2-3 sweeps instead of 7, 30 Gbp in one -- and the sender never forms a k-mer, which makes a sweep cheaper
// than the key-array one as well.  DSK writes super-k-mers to its partition files for the same reason (CHANGELOG.md:13;
// doc/paper.tex:65-67 for the passes).  Needs 20 <= k <= 64 (records) and <= SK_MAX_OWNERS passes; anything else, or a slice
// of the sampled layout that overflows, takes the key-array level 0 / the pass filter instead (ctx->rec_l0_off).
void sk_geometry(dskgpu_ctx* ctx, u64 nwords);
// the sender kernels with k and m at compile time for the BASELINE configs (superkmer.h: sk_tile_fx), at run time otherwise
#define SK_DISPATCH(ctx_, SP_, CALL) do { \
        if (!(ctx_)->tune.sk_generic && (SP_).k == 31 && (SP_).m == 10) { CALL(31, 10); } \
        else if (!(ctx_)->tune.sk_generic && (SP_).k == 63 && (SP_).m == 10) { CALL(63, 10); } \
        else { CALL(0, 0); } } while (0)
int upload_table(dskgpu_ctx* ctx);
struct RecL0 { u32 G = 0; u64 nch = 0; u32 slice[SK_MAX_OWNERS] = {0}; u64 region[SK_MAX_OWNERS] = {0}; };      // region[o]: records of owner o's region (nch * slice[o])
#define REC_L0_NO 2001             // rec_l0_prepare / _sweep: this input does not take the record path (not an error)

int rec_l0_prepare(dskgpu_ctx* ctx, u64 nwords, u32 G, RecL0* rl) {
    SkParams& sp = ctx->sk_sp;
    sk_geometry(ctx, nwords);
    sp.G = G; sp.olo = 0; sp.ohi = G; sp.oslice = nullptr; sp.obase = nullptr;
    const u64 nch = sp.nchunks, tpc = sp.tiles_per_chunk;
    const u32 step = tpc >= 16 ? 16u : 1u;
    // 1. the repartition table for G owners, from the sampled k-mer load of every minimizer bucket
    {
        SkParams ss = sp; ss.sample_step = step; ss.table = nullptr;
        CK(ctx->sk_load.ensure((size_t)SK_BUCKETS * 8));
        CK(hipMemsetAsync(ctx->sk_load.p, 0, (size_t)SK_BUCKETS * 8, ctx->stream));
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_sample<K_, M_>), dim3(ss.nchunks), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), ss, ctx->sk_load.as<unsigned long long>())
        SK_DISPATCH(ctx, ss, SK_CALL);
#undef SK_CALL
        CKL("k_sk_sample");
        std::vector<uint64_t> loads(SK_BUCKETS);
        CK(hipMemcpyAsync(loads.data(), ctx->sk_load.p, (size_t)SK_BUCKETS * 8, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        ctx->h_table.resize(SK_BUCKETS);
        dskgpu_mg_make_table(loads.data(), G, ctx->h_table.data());
        ctx->table_dirty = true;
        const int rc = upload_table(ctx);
        if (rc) return rc;
    }
    // 2. records per (owner, chunk), counted on every 16th tile (all tiles of a small input): the slice of an (owner, chunk) pair
    const u64 M = (u64)G * nch;
    CK(ctx->mat1.ensure((M + 1) * 4));
    CK(ctx->sk_sent.ensure(3 * SK_MAX_OWNERS * 8));
    CK(hipMemsetAsync(ctx->sk_sent.as<u64>() + SK_MAX_OWNERS, 0, SK_MAX_OWNERS * 8, ctx->stream));
    sp.sample_step = step;
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_hist<K_, M_>), dim3((unsigned)nch), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, ctx->mat1.as<u32>(), \
                                          ctx->sk_sent.as<unsigned long long>() + SK_MAX_OWNERS)
    SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    CKL("k_sk_hist");
    std::vector<u32> cells(M);
    CK(hipMemcpyAsync(cells.data(), ctx->mat1.p, M * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    const u64 sampled_tiles = (tpc + step - 1) / step;
    rl->G = G; rl->nch = nch;
    for (u32 o = 0; o < G; ++o) {
        u64 tot = 0, mx = 0;
        for (u64 c = 0; c < nch; ++c) { const u64 v = cells[(size_t)o * nch + c]; tot += v; mx = std::max(mx, v); }
        u64 sl;
        if (step == 1) sl = mx + 8;                                                        // every tile counted: the busiest chunk's figure is exact
        else { sl = tot * tpc / (sampled_tiles * nch) + 1; sl += sl * 2 / 25 + 128; }      // the mean per chunk, scaled, + 8 % + 128 (as the multi-GPU sender)
        if (ctx->tune.sk_slice) sl = ctx->tune.sk_slice;                                   // tests
        if (sl * nch >= 0xFFFF0000ull) return REC_L0_NO;                                   // (record positions inside an owner's region stay 32-bit on the reading side)
        rl->slice[o] = (u32)sl; rl->region[o] = sl * nch;
    }
    ctx->mark("level0_size");
    return DSKGPU_OK;
}

// one sweep: the records of owners [olo, ohi) into ctx->l0buf; base[o] = first 8-byte word of owner o's region.  -> REC_L0_NO when a
// slice overflowed (the sampled layout did not hold: the caller starts over on the key-array path)
int rec_l0_sweep(dskgpu_ctx* ctx, const RecL0& rl, u32 olo, u32 ohi, u64 (&base_words)[SK_MAX_OWNERS]) {
    SkParams sp = ctx->sk_sp;
    u64 obase[SK_MAX_OWNERS] = {0}; u32 osl[SK_MAX_OWNERS] = {0};
    u64 tot = 0;
    for (u32 o = 0; o < rl.G; ++o) { osl[o] = rl.slice[o]; obase[o] = tot; if (o >= olo && o < ohi) tot += rl.region[o]; base_words[o] = obase[o] * sp.R; }
    CK(ctx->l0buf.ensure(tot * sp.R * 8 + 64));
    CK(ctx->sk_lay.ensure(SK_MAX_OWNERS * 12));
    CK(hipMemcpyAsync(ctx->sk_lay.p, obase, sizeof obase, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(ctx->sk_lay.as<u64>() + SK_MAX_OWNERS, osl, sizeof osl, hipMemcpyHostToDevice, ctx->stream));
    sp.olo = olo; sp.ohi = ohi; sp.obase = ctx->sk_lay.as<unsigned long long>(); sp.oslice = reinterpret_cast<const u32*>(ctx->sk_lay.as<u64>() + SK_MAX_OWNERS);
    sp.c0 = 0; sp.c0g = 0; sp.clen = (u32)rl.nch; sp.rbase = 0; sp.slice = 0;
    u32* sc = ctx->scalars.as<u32>();
    CK(hipMemsetAsync(ctx->sk_sent.p, 0, SK_MAX_OWNERS * 8, ctx->stream));
    CK(hipMemsetAsync(sc + SC_OVF1, 0, 4, ctx->stream));
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_scatter<true, K_, M_>), dim3(sp.nchunks), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, \
                                          (const unsigned long long*)nullptr, ctx->l0buf.as<u64>(), sc + SC_OVF1, ctx->sk_sent.as<unsigned long long>())
    SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    CKL("k_sk_scatter(passes)");
    ctx->mark("level0");
    CK(hipMemcpyAsync(&ctx->h_ovf1, sc + SC_OVF1, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(ctx->h_sk_sent, ctx->sk_sent.p, SK_MAX_OWNERS * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));        // (obase / osl are locals)
    if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] level 0 (records): passes %u..%u of %u materialised, %.2f GB%s\n", olo, ohi - 1, rl.G, (double)tot * sp.R * 8e-9,
                                   ctx->h_ovf1 ? " -- a slice overflowed: starting over on the key-array path" : "");
    return ctx->h_ovf1 ? REC_L0_NO : DSKGPU_OK;
}

// The pipeline behind dskgpu_count / dskgpu_mg_count: encode once, then one or several passes over
// the key space (several when the input holds more k-mers than a pass may: < 2^32 offsets, and the
// ping-pong buffers must fit HBM -- the in-memory counterpart of DSK's disk passes), then the row sort.
template <int W>
int run_pipeline(dskgpu_ctx* ctx, bool from_reads, const typename KeyT<W>::T* d_keys_in, u64 nkeys_in) {
    typedef typename KeyT<W>::T Key;
    ctx->have_result = false;
    ctx->sink = dskgpu_ctx::RowSink{};
    if (from_reads) { ctx->st_names.clear(); ctx->st_ms.clear(); }   // from keys: keep the mg_scatter stages of this step
    ctx->marks.clear(); ctx->ev_used = 0;
    const u64 n_upper = from_reads ? ctx->n_bytes : nkeys_in;
    u64 nwords = 0;
    ctx->mark("start");
    if (from_reads) {
        int rc = encode_current(ctx, &nwords);
        if (rc) return rc;
        ctx->mark("encode");
    }
    CK(ctx->scalars.ensure(SC_COUNT * 4));
    CK(ctx->ghist.ensure(((size_t)ctx->cfg.histo_max + 1 + 2) * 8));      // (+ 2: k_sort_back's words)
    CK(ctx->gstats.ensure(4 * 8));
    ctx->have_nvalid = false;
    if (from_reads && nwords) {
        // the exact number of valid k-mer windows (0.08 ms): the plan is sized from it (the byte count over-states the k-mers of
        // 150 bp reads by a quarter: 590 K sub-partitions of 2030 keys instead of 469 K of 2560), and so are the level-1 slices
        CK(hipMemsetAsync(ctx->gstats.p, 0, 4 * 8, ctx->stream));
        hipLaunchKernelGGL(k_count_valid, dim3((unsigned)std::min<u64>((nwords + 255) / 256, (u64)ctx->num_cu * 8)), dim3(256), 0, ctx->stream,
                           ctx->inval.as<u32>(), nwords, (int)ctx->cfg.kmer_size, ctx->gstats.as<u64>() + 3);
        CKL("k_count_valid");
        {
            void* lz = landing(ctx, 8);
            CK(hipMemcpyAsync(lz ? lz : (void*)&ctx->h_nvalid, ctx->gstats.as<u64>() + 3, 8, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            if (lz) std::memcpy(&ctx->h_nvalid, lz, 8);
        }
        ctx->have_nvalid = true;
    }
    const u64 max_keys = ctx->max_keys_per_pass ? ctx->max_keys_per_pass : 0xD0000000ull;      // (3.49 G: level 1 then needs <= 1536 bins, what its LDS holds with the slice ends)
    // Passes over the key space: as few as hold the keys -- any number, not a power of two (key_in_pass maps a bit field of the
    // mixed key onto [0, npass)) -- counted from the EXACT number of valid k-mer windows when the keys come from reads: the byte
    // count over-states the k-mers of 150 bp reads by a quarter, which together with the doubling below once made 16 passes of
    // what 7 hold (200 M x 150 bp on one GPU: every pass re-generates all k-mers).  A pass that turns out too big (a skewed
    // key space) doubles the count.
    const u64 n_keys = (from_reads && ctx->have_nvalid) ? std::max<u64>(1, ctx->h_nvalid) : n_upper;
    // An input that needs several passes anyway (one-word keys from reads) takes SMALL ones -- 10^9 keys, the size of the bench
    // workload: a level-0 sweep then materialises up to 16 of them (what the free HBM holds: a pass of 10^9 keys works in 22 GB,
    // one of 3.5 * 10^9 in 75 GB, which left room for one or two key arrays beside it on a 90 Gbp input), and level 1 of a pass
    // has 512 bins instead of 1536 (its cost per key grows with the bin count: 3.8 against 5.9 ps).
    u64 pass_keys = max_keys;
    if (W == 1 && from_reads && ctx->have_nvalid && n_keys > max_keys && !ctx->max_keys_per_pass && !ctx->tune.no_level0)
        pass_keys = ctx->tune.mp_pass_mkeys ? (u64)ctx->tune.mp_pass_mkeys * 1000000ull : 1000000000ull;
    // two-word keys: a pass holds what 32-bit key indices address (regions of 2180 keys at <= 60 % fill)
    const u64 hard_max = W == 2 && !ctx->max_keys_per_pass ? std::min<u64>(max_keys, 2400000000ull) : max_keys;
    u32 npass = (u32)std::max<u64>(1, (n_keys + pass_keys - 1) / pass_keys);
    // Several passes from reads, 20 <= k <= 64: the passes are virtual OWNERS and a sweep materialises super-k-mer records (above).
    bool rec_l0 = W <= 2 && from_reads && ctx->have_nvalid && ctx->sk_mode && ctx->cfg.world_size == 1 && !ctx->tune.no_level0 && !ctx->tune.l0_keys && !ctx->rec_l0_off &&
                  (npass > 1 || n_keys > hard_max);
    RecL0 rl;
    // The record-based level 0 borrows the multi-GPU sender state of this context (sk_sp: G "owners" = passes, the owner window, the
    // repartition table for G owners).  Whatever way this function is left, the context gets back the state dskgpu_create /
    // dskgpu_mg_set_table gave it: a later dskgpu_mg_* call on a world_size = 1 context must see ONE owner again (it
    // looped over up to 64 owners into the caller's world_size-sized arrays).
    struct SenderStateGuard {
        dskgpu_ctx* c; SkParams sp; std::vector<uint8_t> table; bool armed = false;
        ~SenderStateGuard() { if (armed) { c->sk_sp = sp; c->h_table.swap(table); c->table_dirty = true; c->sk_prepared = false; c->enc_fresh = false; } }
    } sender_guard{ctx, ctx->sk_sp, ctx->h_table};
    if (rec_l0) {
        sender_guard.armed = true;
        u64 want = ctx->max_keys_per_pass ? max_keys : ctx->tune.mp_pass_mkeys ? (u64)ctx->tune.mp_pass_mkeys * 1000000ull : (W == 1 ? 1200000000ull : 1000000000ull);
        u64 G = (n_keys + want - 1) / want;
        if (G < 2) G = 2;
        if (G > SK_MAX_OWNERS) G = SK_MAX_OWNERS;
        // (owners are balanced to a few per cent by the repartition table: 15 % head-room under what a pass may hold)
        if (n_keys / G + n_keys / G / 7 > hard_max && !ctx->max_keys_per_pass) rec_l0 = false;
        else {
            const int rc = rec_l0_prepare(ctx, nwords, (u32)G, &rl);
            if (rc == REC_L0_NO) rec_l0 = false; else if (rc) return rc;
            else npass = (u32)G;
        }
    }
    u64 cap_floor = 0;       // keys the largest pass seen so far really holds (a k-mer with millions of occurrences sits in ONE pass whatever their number)
    for (;;) {
        if (npass > 4096) return fail(ctx, DSKGPU_E_OVERFLOW, "too many passes (one k-mer alone exceeds a pass)");
        // buffers of one pass: every position could yield a key when there is a single pass; with several, the hash spreads
        // the keys evenly: 6 % + 1 M head-room, checked after the level-1 histogram (exact path) or by the slices (sampled path)
        const u64 cap = npass == 1 ? n_upper : std::min<u64>(n_upper, std::max<u64>(cap_floor, n_keys / npass + n_keys / npass / 16 + (1u << 20)));
        if (cap >= 0xFFFF0000ull) { npass *= 2; cap_floor = 0; continue; }
        // (bufA / bufB are sized by the pass itself: the histogram-free path wants slices and regions, not `cap` keys)
        if (W > 1) CK(ctx->abund2.ensure((cap + 1) * 4));
        ctx->hist.assign((size_t)ctx->cfg.histo_max + 1, 0);
        std::vector<u64> pass_hist(ctx->hist.size());
        u64 tot_rows = 0, tot_kmers = 0, tot_distinct = 0;
        Plan pl{};
        bool too_big = false;
        u32 l0_lo = 0, l0_n = 0; u64 l0_keys[L0_MAX_PASSES] = {0}; u64 l0_base[L0_MAX_PASSES] = {0};       // passes materialised by the last level-0 sweep
        bool l0_try = W == 1 && from_reads && npass > 1 && ctx->have_nvalid && !ctx->tune.no_level0 && !rec_l0;
        u32 r_hi = 0; u64 r_base[SK_MAX_OWNERS] = {0};     // record-based level 0: owners materialised by the last sweep, first word of each one's region
        bool restart = false;
        u64 sweeps = 0;          // times the encoded reads were walked to generate k-mers (DSK's notion of a pass: README.md:126-130)
        bool rows_sized = false; // the row accumulators are sized for all passes (known after the first one)
        ctx->job_passes = npass;
        ctx->mp_parts.clear(); ctx->mp_off_used = 0;
        ctx->mp_part_ok = npass > 1 && W <= 2 && (ctx->cfg.flags & DSKGPU_F_PARTITION_ORDER) && !(ctx->cfg.flags & DSKGPU_F_NO_SORT) && !ctx->bank_job.active;
        if (ctx->mp_part_ok) { CK(ctx->mp_flag.ensure(256)); CK(hipMemsetAsync(ctx->mp_flag.p, 0, 4, ctx->stream)); }
        for (u32 p = 0; p < npass; ++p) {
            u64 ns = 0, nk = 0;
            int rc;
            const size_t mp_before = ctx->mp_parts.size();
            ctx->sink = dskgpu_ctx::RowSink{};
            if (npass > 1 && rows_sized) {      // the pass compacts its rows straight behind the job's (when they fit: run_one_pass)
                ctx->sink.active = true; ctx->sink.rows = tot_rows; ctx->sink.ab = ctx->acc_ab.as<u32>();
                u64 cap_rows = ctx->acc_ab.cap / 4;
                for (int x = 0; x < W; ++x) { ctx->sink.w[x] = ctx->acc_w[x].as<u64>(); cap_rows = std::min<u64>(cap_rows, ctx->acc_w[x].cap / 8); }
                ctx->sink.cap = cap_rows;
            }
            if (npass > 1) { ctx->opt1_off = false; ctx->opt2_off = false; ctx->mw_v3_off = false; }      // an overflow is a property of ONE pass (the one that holds a k-mer with 10^8 occurrences): the others keep the fast path
            if (rec_l0) {
                if (p >= r_hi) {       // the next sweep: as many owners as HBM holds beside a pass's own buffers and the rows still to come
                    size_t free_b = 0, total_b = 0;
                    CK(hipMemGetInfo(&free_b, &total_b));
                    const u64 nper = n_keys / npass + 1;
                    const u64 have = ctx->bufA.cap + ctx->bufB.cap + ctx->l0buf.cap;
                    const u64 rows_have = ctx->acc_ab.cap + ctx->acc_w[0].cap + (W > 1 ? ctx->acc_w[1].cap : 0);
                    const u64 rows_want = rows_sized ? 0 : n_keys / 16 * (8ull * W + 4);
                    const u64 need = nper * (W == 1 ? 30ull : 50ull) + (4ull << 30) + (rows_want > rows_have ? rows_want - rows_have : 0);
                    const u64 room = free_b + have > need ? free_b + have - need : 0;
                    const u64 R8 = (u64)ctx->sk_sp.R * 8;
                    u64 left = 0; for (u32 o = p; o < npass; ++o) left += rl.region[o] * R8;
                    const u64 nsw = std::max<u64>(1, (left + std::max<u64>(room, 1) - 1) / std::max<u64>(room, 1));      // sweeps still needed: equal shares
                    const u64 target = (left + nsw - 1) / nsw;
                    u64 acc = 0; u32 hi = p;
                    while (hi < npass && (hi == p || (acc + rl.region[hi] * R8 <= room && acc < target))) { acc += rl.region[hi] * R8; ++hi; }
                    if (ctx->tune.l0_passes) hi = std::min<u32>(npass, p + ctx->tune.l0_passes);      // tests
                    rc = rec_l0_sweep(ctx, rl, p, hi, r_base);
                    if (rc == REC_L0_NO) { restart = true; break; }
                    if (rc) return rc;
                    r_hi = hi; ++sweeps;
                }
                const u64 nvalid = ctx->h_nvalid;
                const u64 nk_in = ctx->h_sk_sent[p];
                ctx->rec_src = ctx->l0buf.as<u64>() + r_base[p]; ctx->rec_n = rl.region[p]; ctx->rec_expanded = false; ctx->rec_sized = false;
                ctx->rec_hint = 0; ctx->rec_hint_est = false; ctx->rec_slice_end.clear(); ctx->rec_gate = nullptr;
                if (nk_in == 0) { ns = 0; nk = 0; ctx->h_stats[0] = 0; rc = DSKGPU_OK; CK(hipMemsetAsync(ctx->ghist.p, 0, ((size_t)ctx->cfg.histo_max + 1) * 8, ctx->stream)); }
                else rc = run_one_pass<W>(ctx, false, nullptr, nk_in, 0, 0u, 1u, nk_in, &ns, &nk, &pl);
                ctx->rec_src = nullptr;
                ctx->h_nvalid = nvalid;
            } else
            if (l0_try && p >= l0_lo + l0_n) {       // the next group of passes: one sweep over the reads writes their keys
                // (until the first pass has told how many rows a pass leaves, room is kept for one solid row per sixteen k-mers -- what
                //  20x coverage leaves; should the rows need more, the group's remaining key arrays give way: below)
                const u64 reserve = rows_sized ? 0 : n_keys / 16 * 12;
                if constexpr (W == 1) { if ((rc = level0_materialise(ctx, nwords, p, npass, reserve, &l0_n, l0_keys, l0_base))) return rc; }
                l0_lo = p;
                if (l0_n == 0) l0_try = false;       // no room (or a skewed key space): every pass reads the reads
                else ++sweeps;
            }
            if (rec_l0) { /* done above */ }
            else if (l0_try && p < l0_lo + l0_n) {
                const u64 nvalid = ctx->h_nvalid;    // (a pass from a key array sizes itself from its own key count)
                const u64 nk_in = l0_keys[p - l0_lo];
                rc = run_one_pass<W>(ctx, false, reinterpret_cast<const Key*>(ctx->l0buf.as<u64>() + l0_base[p - l0_lo]), nk_in, 0, 0u, 1u, nk_in, &ns, &nk, &pl);
                ctx->h_nvalid = nvalid;
            } else { rc = run_one_pass<W>(ctx, from_reads, d_keys_in, nkeys_in, nwords, p, npass, cap, &ns, &nk, &pl); if (from_reads) ++sweeps; }
            if (rc == PASS_TOO_BIG) { too_big = true; break; }
            if (rc) return rc;
            tot_kmers += nk; tot_distinct += ctx->h_stats[0];
            if (ctx->mp_parts.size() > mp_before) ctx->mp_parts.back().row_base = tot_rows;
            if (npass > 1) {      // append this pass's rows and histogram to the job's
                CK(hipMemcpyAsync(pass_hist.data(), ctx->ghist.p, pass_hist.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
                // grow once: the passes hold similar numbers of rows (hash-uniform), so size for all of them after the first
                const u64 want_rows = std::max<u64>(tot_rows + ns + 1, p == 0 ? (ns + ns / 8 + 1024) * npass : 0);
                const bool took = ctx->sink.took;                        // (the rows are in the accumulators already)
                const u64 keep_rows = tot_rows + (took ? ns : 0);
                ctx->sink = dskgpu_ctx::RowSink{};
                auto grow_rows = [&]() {
                    bool ok = ctx->acc_ab.ensure_keep(want_rows * 4, keep_rows * 4, ctx->stream) == 0;
                    for (int x = 0; x < W && ok; ++x) ok = ctx->acc_w[x].ensure_keep(want_rows * 8, keep_rows * 8, ctx->stream) == 0;
                    return ok;
                };
                if (!grow_rows()) {      // the rows need more than was kept for them: the key arrays of the group's remaining passes give way (those passes get a sweep of their own)
                    (void)hipGetLastError();
                    if (l0_n && ctx->l0buf.p) { ctx->l0buf.release(); l0_n = p + 1 - l0_lo; }
                    if (rec_l0 && ctx->l0buf.p) { ctx->l0buf.release(); r_hi = p + 1; }
                    if (!grow_rows()) return fail(ctx, DSKGPU_E_NOMEM, "row accumulation");
                }
                rows_sized = true;
                if (ns && !took) {
                    CK(hipMemcpyAsync(ctx->acc_ab.as<u32>() + tot_rows, ctx->out_ab.p, ns * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    for (int x = 0; x < W; ++x)
                        CK(hipMemcpyAsync(ctx->acc_w[x].as<u64>() + tot_rows, ctx->out_w[x].p, ns * 8, hipMemcpyDeviceToDevice, ctx->stream));
                }
                CK(hipStreamSynchronize(ctx->stream));
                for (size_t i = 0; i < pass_hist.size(); ++i) ctx->hist[i] += pass_hist[i];
                ctx->resolve_marks();
                ctx->mark("start");
            }
            tot_rows += ns;
        }
        if (rec_l0) { ctx->sk_prepared = false; ctx->enc_fresh = false; }      // (the sender state of this context was used for the passes)
        if (restart) {          // the record layout did not hold for these reads: the same count on the key-array path
            ctx->rec_l0_off = true;
            ctx->resolve_marks();
            return run_pipeline<W>(ctx, from_reads, d_keys_in, nkeys_in);
        }
        if (too_big) {
            // the pass holds more keys than its buffers: give the passes that capacity (more passes would not make THAT pass
            // smaller); only when it exceeds what 32-bit offsets address, more passes
            const u64 seen = (u64)ctx->h_back[2];
            if (seen + (1u << 20) < 0xFFFF0000ull && seen > cap_floor) cap_floor = seen + (1u << 20); else { npass *= 2; cap_floor = 0; }
            continue;
        }
        // ---------------- partition order over all passes: every pass ordered its partitions on the way in -- nothing left to sort unless a block gave up
        bool mp_done = false;
        if (npass > 1 && ctx->mp_part_ok && !ctx->mp_parts.empty()) {
            u32 h_flag = 1;
            ctx->h_mp_off.resize(ctx->mp_off_used);
            CK(hipMemcpyAsync(&h_flag, ctx->mp_flag.p, 4, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipMemcpyAsync(ctx->h_mp_off.data(), ctx->mp_part_off.p, (size_t)ctx->mp_off_used * 4, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            if (!h_flag) {
                ctx->h_part_off64.clear();
                for (const auto& mp : ctx->mp_parts)
                    for (u32 i = 0; i < mp.nparts; ++i) ctx->h_part_off64.push_back(mp.row_base + ctx->h_mp_off[mp.off_index + i]);
                ctx->h_part_off64.push_back(tot_rows);
                ctx->part_mode = true; ctx->n_parts = (u32)(ctx->h_part_off64.size() - 1);
                for (int x = 0; x < 4; ++x) ctx->res_w[x] = x < W ? ctx->acc_w[x].as<u64>() : nullptr;
                ctx->res_ab = ctx->acc_ab.as<u32>();
                ctx->sort_partial = false; ctx->sp_rows.valid = false; ctx->sp_rows2.valid = false;
                ctx->h_back[3] = 0; ctx->h_ovs.assign(1, 0);
                mp_done = true;
            } else if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] row order: a partition of one pass exceeds what one block orders -- global order over all passes instead\n");
        }
        // ---------------- row sort over all passes
        if (npass > 1 && !mp_done) {      // make the accumulated rows the sort input
            std::swap(ctx->out_ab, ctx->acc_ab);
            for (int x = 0; x < W; ++x) std::swap(ctx->out_w[x], ctx->acc_w[x]);
        }
        int rc;
        ctx->part_off_this_count = false;
        if (mp_done) goto sort_done;
      sort_again:
        ctx->sort_back = 0;
        if ((rc = sort_rows(ctx, tot_rows))) return rc;
        ctx->mark("sort");
        if (ctx->sort_back) {      // histogram + the sort's two words: one copy into pinned memory (three pageable ones were ~25 us of idle GPU each)
            const size_t nh = ctx->hist.size();
            if (ctx->hist_pin_n < nh + 2) {
                if (ctx->hist_pin) CK(hipHostFree(ctx->hist_pin));
                ctx->hist_pin = nullptr; ctx->hist_pin_n = 0;
                CK(hipHostMalloc(reinterpret_cast<void**>(&ctx->hist_pin), (nh + 2) * 8, hipHostMallocDefault));
                ctx->hist_pin_n = nh + 2;
            }
            u64* gh = ctx->ghist.as<u64>();
            hipLaunchKernelGGL(k_sort_back, dim3(1), dim3(64), 0, ctx->stream, (const u32*)ctx->scalars.as<u32>(),
                               ctx->sort_back == 1 ? (const u32*)ctx->rs_ovs.as<u32>() : (const u32*)nullptr, gh + nh);
            CKL("k_sort_back");
            const size_t from = npass == 1 ? 0 : nh;
            CK(hipMemcpyAsync(ctx->hist_pin + from, gh + from, (nh + 2 - from) * 8, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            ctx->h_back[3] = (u32)ctx->hist_pin[nh]; ctx->h_ovs.assign(1, (u32)ctx->hist_pin[nh + 1]);
            if (npass == 1) memcpy(ctx->hist.data(), ctx->hist_pin, nh * 8);
            ctx->sort_back = 0;
        } else {
            if (npass == 1) CK(hipMemcpyAsync(ctx->hist.data(), ctx->ghist.p, ctx->hist.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
        }
        if (ctx->part_mode && ctx->h_back[3]) {      // a partition (or a value bin of one) above what a block orders in LDS: the global sort, on the same sparse rows
            if (ctx->tune.verbose) fprintf(stderr, "[dskgpu] row sort: a partition exceeds what one block orders -- global order instead of partition order\n");
            ctx->part_mode = false; ctx->part_off_this_count = true; ctx->h_back[3] = 0;
            ctx->sp_rows = ctx->sp_rows_saved; ctx->sp_rows2 = ctx->sp_rows2_saved;
            goto sort_again;
        }
        if (W == 1 && ctx->sort_partial && tot_rows && !ctx->h_back[3] && !ctx->h_ovs.empty() && ctx->h_ovs[0]) {
            if ((rc = sort_oversize(ctx))) return rc;      // sub-buckets the sort listed for another round on their remaining bits
        } else if (W > 1 && ctx->sort_partial && !ctx->h_ovs.empty() && ctx->h_ovs[0]) ctx->h_back[3] = 1;      // (index pairs of multi-word rows: the full-width order)
        if (W == 1 && ctx->sort_partial && tot_rows && ctx->h_back[3]) {
            // a run of equal 32-bit prefixes was too long for the in-place fix-up: sort full width
            // (srt_* holds a permutation of the rows; sort it back into out_*)
            size_t tmp = 0;
            const unsigned end_bit = std::min(64u, 2u * ctx->cfg.kmer_size);
            // LIBRARY SORT (rocprim), labelled FALLBACK: the one-word row sort raised its flag (a value distribution no k-mer spectrum has): full-width order of all rows
            CK(rocprim::radix_sort_pairs(nullptr, tmp, ctx->fb_src_k, ctx->fb_dst_k, ctx->fb_src_v, ctx->fb_dst_v, (size_t)tot_rows, 0u, end_bit, ctx->stream));
            CK(ctx->srt_tmp.ensure(tmp));
            CK(rocprim::radix_sort_pairs(ctx->srt_tmp.p, tmp, ctx->fb_src_k, ctx->fb_dst_k, ctx->fb_src_v, ctx->fb_dst_v, (size_t)tot_rows, 0u, end_bit, ctx->stream));
            CK(hipStreamSynchronize(ctx->stream));
            ctx->res_w[0] = ctx->fb_dst_k; ctx->res_ab = ctx->fb_dst_v;
            ctx->stats.sort_fallback = 1;
        } else if (W > 1 && ctx->sort_partial && tot_rows && ctx->h_back[3]) {
            if (ctx->rows2_in_scratch) {      // (two-word rows above RS_MAX_ROWS: the complete permutation is the scratch copy)
                CK(hipMemcpyAsync(ctx->out_w[1].p, ctx->rows2_scratch.hi, tot_rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
                CK(hipMemcpyAsync(ctx->out_w[0].p, ctx->rows2_scratch.lo, tot_rows * 8, hipMemcpyDeviceToDevice, ctx->stream));
                CK(hipMemcpyAsync(ctx->out_ab.p, ctx->rows2_scratch.ab, tot_rows * 4, hipMemcpyDeviceToDevice, ctx->stream));
            }
            if ((rc = sort_rows_full_multiword(ctx, tot_rows))) return rc;      // out_w holds the unsorted rows
            CK(hipStreamSynchronize(ctx->stream));
            ctx->stats.sort_fallback = 1;
        }
      sort_done:
        ctx->resolve_marks();
        ctx->n_rows = tot_rows;
        ctx->stats.n_bytes = from_reads ? ctx->n_bytes : 0;
        ctx->stats.n_kmers = tot_kmers;
        ctx->stats.n_distinct = tot_distinct;
        ctx->stats.n_solid = tot_rows;
        ctx->stats.n_levels = (u32)pl.levels;
        ctx->stats.n_final_bins = pl.F;
        ctx->stats.n_passes = npass;
        ctx->stats.n_read_sweeps = npass > 1 ? sweeps : (from_reads ? 1 : 0);
        if (from_reads) ctx->last_rows = tot_rows;
        if (npass > 1 && !mp_done) {      // the names go back: acc_* stays the job-sized buffer (it holds the result now), out_* the pass-sized one --
            std::swap(ctx->out_ab, ctx->acc_ab);      // left swapped, the next count grew the small one to job size again (10 GB of hipMalloc + hipFree per call)
            for (int x = 0; x < W; ++x) std::swap(ctx->out_w[x], ctx->acc_w[x]);
        }
        u32 np = ctx->cfg.nb_partitions ? ctx->cfg.nb_partitions : 4u;
        if (ctx->part_mode) np = ctx->n_parts;      // (partition order: the partitions are what the blocks of k_part_sort ordered)
        ctx->stats.n_partitions = np;
        ctx->have_result = true;
        return DSKGPU_OK;
    }
}

}  // namespace

namespace {
template <int W>
int mg_scatter_impl(dskgpu_ctx* ctx, void* d_send, uint64_t* send_words) {
    typedef typename KeyT<W>::T Key;
    ctx->st_names.clear(); ctx->st_ms.clear(); ctx->marks.clear(); ctx->ev_used = 0;
    ctx->mark("start");
    u64 nwords = 0;
    int rc = encode_current(ctx, &nwords);
    if (rc) return rc;
    ctx->mark("encode");
    const u32 G = ctx->cfg.world_size;
    u32 nch1 = 0;
    build_descs1(ctx, nwords, Tile<W>::WORDS, (u64)ctx->num_cu * 8, &nch1);
    const u64 M1 = (u64)G * nch1;
    CK(ctx->scalars.ensure(SC_COUNT * 4));
    CK(ctx->descs1.ensure(ctx->h_descs1.size() * sizeof(ChunkDesc)));
    CK(hipMemcpyAsync(ctx->descs1.p, ctx->h_descs1.data(), ctx->h_descs1.size() * sizeof(ChunkDesc), hipMemcpyHostToDevice, ctx->stream));
    u32* h_sc = ctx->h_sc;
    std::memset(h_sc, 0, sizeof(ctx->h_sc));
    h_sc[SC_NCH1] = nch1; h_sc[SC_MLEN1] = (u32)M1;
    u32* sc = ctx->scalars.as<u32>();
    CK(hipMemcpyAsync(sc, h_sc, sizeof(ctx->h_sc), hipMemcpyHostToDevice, ctx->stream));
    CK(ctx->mat1.ensure((M1 + 1) * 4));
    const DigitSpec owner = DigitSpec{0u, G, 0u, G, 1u, 0u};
    if ((rc = launch_hist<W, 0>(ctx, nullptr, ctx->descs1.as<ChunkDesc>(), sc + SC_NCH1, nch1, ctx->mat1.as<u32>(), owner, G))) return rc;
    ctx->mark("mg_hist");
    if ((rc = run_scan(ctx, ctx->mat1.as<u32>(), sc + SC_MLEN1, M1))) return rc;
    if ((rc = launch_scatter<W, 0>(ctx, nullptr, ctx->descs1.as<ChunkDesc>(), sc + SC_NCH1, nch1, ctx->mat1.as<u32>(), static_cast<Key*>(d_send), owner, G))) return rc;
    ctx->mark("mg_scatter");
    ctx->h_starts.assign(G + 1, 0);
    for (u32 o = 0; o <= G; ++o)
        CK(hipMemcpyAsync(&ctx->h_starts[o], ctx->mat1.as<u32>() + (u64)o * nch1, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->resolve_marks();
    for (u32 o = 0; o < G; ++o) send_words[o] = (u64)(ctx->h_starts[o + 1] - ctx->h_starts[o]) * W;
    return DSKGPU_OK;
}

// ---- repartition table of the super-k-mer owner map (superkmer.h)
void default_table(uint32_t world, uint8_t* table) { for (u32 b = 0; b < SK_BUCKETS; ++b) table[b] = (uint8_t)((b * world) / SK_BUCKETS); }
int upload_table(dskgpu_ctx* ctx) {
    if (ctx->h_table.size() != SK_BUCKETS) { ctx->h_table.resize(SK_BUCKETS); default_table(ctx->cfg.world_size, ctx->h_table.data()); ctx->table_dirty = true; }
    if (ctx->table_dirty) {
        CK(ctx->sk_table.ensure(SK_BUCKETS));
        CK(hipMemcpyAsync(ctx->sk_table.p, ctx->h_table.data(), SK_BUCKETS, hipMemcpyHostToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        ctx->table_dirty = false;
    }
    ctx->sk_sp.table = ctx->sk_table.as<unsigned char>();
    ctx->sk_sp.has_split = std::find(ctx->h_table.begin(), ctx->h_table.end(), (uint8_t)SK_SPLIT) != ctx->h_table.end() ? 1u : 0u;
    return DSKGPU_OK;
}
// tiles / chunks of the sender kernels over the encoded stream
void sk_geometry(dskgpu_ctx* ctx, u64 nwords) {
    SkParams& sp = ctx->sk_sp;
    sp.ngroups = nwords * 2;
    sp.ntiles = std::max<u64>(1, (sp.ngroups + SK_GROUPS - 1) / SK_GROUPS);
    // whole rounds of blocks: the k = 31 kernels (68 VGPRs, 50 KB of LDS) run three 512-thread blocks per CU, the others two
    const u64 per_cu = (!ctx->tune.sk_generic && sp.k == 31 && sp.m == 10) ? 9 : 8;
    u64 nch = std::min<u64>(std::max<u64>(1, sp.ntiles / 8), (u64)ctx->num_cu * per_cu);     // >= 8 tiles per chunk when there are that many
    const u64 tpc = (sp.ntiles + nch - 1) / nch;
    nch = (sp.ntiles + tpc - 1) / tpc;
    sp.tiles_per_chunk = (u32)tpc; sp.nchunks = (u32)nch;
    sp.c0 = 0; sp.c0g = 0; sp.clen = (u32)nch; sp.rbase = 0;      // one layout group: the whole step
}

// ---- multi-GPU exchange as super-k-mer records (superkmer.h)
// Sender, step 1: encode + count the records per (owner, chunk) + scan.  Leaves the record range of every
// owner in h_starts; the exact send size is known before the caller allocates the send buffer.
int sk_prepare(dskgpu_ctx* ctx) {
    ctx->st_names.clear(); ctx->st_ms.clear(); ctx->marks.clear(); ctx->ev_used = 0;
    ctx->sk_prepared = false;
    ctx->mark("start");
    u64 nwords = 0;
    int rc = DSKGPU_OK;
    if (ctx->enc_fresh) { nwords = (ctx->n_bytes + 31) / 32; ctx->enc_fresh = false; }      // (the repartition sample of this step just encoded these reads)
    else if ((rc = encode_current(ctx, &nwords))) return rc;
    ctx->mark("encode");
    SkParams& sp = ctx->sk_sp;
    sk_geometry(ctx, nwords);
    if ((rc = upload_table(ctx))) return rc;
    const u64 nch = sp.nchunks, tpc = sp.tiles_per_chunk;
    const u64 M = (u64)sp.G * nch;
    CK(ctx->scalars.ensure(SC_COUNT * 4));
    u32* h_sc = ctx->h_sc;
    std::memset(h_sc, 0, sizeof(ctx->h_sc));
    u32* sc = ctx->scalars.as<u32>();
    CK(hipMemcpyAsync(sc, h_sc, sizeof(ctx->h_sc), hipMemcpyHostToDevice, ctx->stream));
    CK(ctx->mat1.ensure((M + 1) * 4));
    // Exact layout: count every record, one scan places them.  Slice layout (default): count the records of every
    // 16th tile only, give every (owner, chunk) pair one slice of the estimated mean + 8 % + 128 records; the scatter
    // pads the slices with zero-length records.  Saves the full counting pass (1.95 of 5 ms); ~8 % more words to send.
    // Either way every record position is 64-bit from here on (the count matrix -- <= 64 owners x 2048 chunks of u32 -- comes to
    // the host, where it is summed / scanned in 64 bits): a rank's shard may be of any size.  The reference's own human run is
    // ONE execute() over 160 GB of reads (doc/human_log:3-4,20-24; README.md:126-130); on 8 GPUs that is 11.3 GB per rank.
    const bool slices = !ctx->sk_exact && !ctx->tune.sk_exact && tpc >= 8;
    sp.sample_step = slices ? 16u : 1u;
    CK(ctx->sk_sent.ensure(3 * SK_MAX_OWNERS * 8));            // [k-mers sent per owner | sampled k-mers per owner | overflow flag of a sliced step]
    CK(hipMemsetAsync(ctx->sk_sent.as<u64>() + SK_MAX_OWNERS, 0, SK_MAX_OWNERS * 8, ctx->stream));
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_hist<K_, M_>), dim3((unsigned)nch), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, ctx->mat1.as<u32>(), \
                                          ctx->sk_sent.as<unsigned long long>() + SK_MAX_OWNERS)
    SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    CKL("k_sk_hist");
    CK(hipMemcpyAsync(ctx->h_sk_est, ctx->sk_sent.as<u64>() + SK_MAX_OWNERS, SK_MAX_OWNERS * 8, hipMemcpyDeviceToHost, ctx->stream));
    ctx->mark("mg_hist");
    ctx->h_sk_cells.resize(M);
    CK(hipMemcpyAsync(ctx->h_sk_cells.data(), ctx->mat1.p, M * 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->h_rstart.assign(sp.G + 1, 0);
    ctx->sk_slices = false;
    if (slices) {
        u64 worst = 0;                                       // sampled records of the busiest owner
        for (u32 o = 0; o < sp.G; ++o) { u64 t = 0; for (u64 c = 0; c < nch; ++c) t += ctx->h_sk_cells[(size_t)o * nch + c]; worst = std::max(worst, t); }
        const u64 sampled_tiles = (tpc + sp.sample_step - 1) / sp.sample_step;            // per chunk
        u64 slice = worst * tpc / (sampled_tiles * nch) + 1;                              // records per (owner, chunk), estimated
        const u64 min_slice = ctx->tune.sk_minslice;                                      // (tests lower it)
        const bool small = slice < min_slice || worst < 20000;    // fixed slack too visible in the send volume, or too few sampled records to trust the estimate
        slice += slice * 2 / 25 + 128;
        if (ctx->tune.sk_slice) slice = ctx->tune.sk_slice;                               // tests
        if (!small && slice < 0xFFFF0000ull) {                                            // (a block's cursor inside ONE slice is 32-bit)
            sp.slice = (u32)slice;
            for (u32 o = 0; o < sp.G; ++o) ctx->h_sk_est[o] = ctx->h_sk_est[o] * tpc / sampled_tiles;      // sampled tiles -> all tiles
            for (u32 o = 0; o <= sp.G; ++o) ctx->h_rstart[o] = (u64)o * nch * slice;
            ctx->sk_slices = true;
        } else {                                             // small input: count exactly after all
            ctx->sk_exact = true;
            return sk_prepare(ctx);
        }
    } else {
        // exact layout: the 64-bit exclusive scan of the owner-major count matrix
        ctx->h_sk_cb64.resize(M + 1);
        u64 run = 0;
        for (u64 i = 0; i < M; ++i) { ctx->h_sk_cb64[i] = run; run += ctx->h_sk_cells[i]; }
        ctx->h_sk_cb64[M] = run;
        for (u32 o = 0; o <= sp.G; ++o) ctx->h_rstart[o] = ctx->h_sk_cb64[(u64)o * nch];
        CK(ctx->sk_cb64.ensure((M + 1) * 8));
        CK(hipMemcpyAsync(ctx->sk_cb64.p, ctx->h_sk_cb64.data(), (M + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
    }
    ctx->resolve_marks();
    ctx->sk_prepared = true;
    return DSKGPU_OK;
}

// Sender, step 2: write the records, grouped by owner, into the caller's buffer.
int sk_scatter(dskgpu_ctx* ctx, void* d_send, uint64_t capacity_words, uint64_t* send_words) {
    int rc;
    if (!ctx->sk_prepared && (rc = sk_prepare(ctx))) return rc;
    const SkParams& sp = ctx->sk_sp;
    if (capacity_words < ctx->h_rstart[sp.G] * sp.R) return fail(ctx, DSKGPU_E_ARG, "send buffer too small");
    ctx->marks.clear(); ctx->ev_used = 0;
    ctx->mark("start");
    u32* sc = ctx->scalars.as<u32>();
    CK(hipMemsetAsync(ctx->sk_sent.p, 0, SK_MAX_OWNERS * 8, ctx->stream));
    if (ctx->sk_slices) {
        CK(hipMemsetAsync(sc + SC_OVF1, 0, 4, ctx->stream));
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_scatter<true, K_, M_>), dim3(sp.nchunks), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, \
                                          (const unsigned long long*)nullptr, static_cast<u64*>(d_send), sc + SC_OVF1, ctx->sk_sent.as<unsigned long long>())
        SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    } else {
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_scatter<false, K_, M_>), dim3(sp.nchunks), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, \
                                          (const unsigned long long*)ctx->sk_cb64.as<unsigned long long>(), static_cast<u64*>(d_send), sc + SC_OVF1, ctx->sk_sent.as<unsigned long long>())
        SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    }
    CKL("k_sk_scatter");
    ctx->mark("mg_scatter");
    if (ctx->sk_slices) CK(hipMemcpyAsync(&ctx->h_ovf1, sc + SC_OVF1, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(ctx->h_sk_sent, ctx->sk_sent.p, SK_MAX_OWNERS * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->resolve_marks();
    if (ctx->sk_slices && ctx->h_ovf1) {      // a slice overflowed: exact counts for these reads from now on (and right away)
        ctx->sk_exact = true; ctx->sk_prepared = false;
        if ((rc = sk_prepare(ctx))) return rc;
        return sk_scatter(ctx, d_send, capacity_words, send_words);      // may report "send buffer too small": ask for the capacity again
    }
    for (u32 o = 0; o < sp.G; ++o) send_words[o] = (ctx->h_rstart[o + 1] - ctx->h_rstart[o]) * sp.R;
    ctx->sk_prepared = false;      // packed/mat1 are scratch of the next call
    return DSKGPU_OK;
}

// ---- a step in slices (the exchange of slice i overlaps the sender's slice i + 1 and the receiver's level 1 of slice i - 1).
// Only with the sampled send layout (its sizes are known before a record exists): slice s = the chunks [s * nch / S, (s + 1) * nch / S),
// a layout group of its own in the send buffer (SkParams::c0g, clen, rbase), owner-major inside.
void sk_slice_range(const SkParams& sp, u32 S, u32 s, u32* cb, u32* ce) { *cb = (u32)((u64)s * sp.nchunks / S); *ce = (u32)((u64)(s + 1) * sp.nchunks / S); }

int sk_slices_prepare(dskgpu_ctx* ctx, u32 want, u32* nslices, uint64_t* send_words, uint64_t* kmers_est) {
    int rc;
    *nslices = 0; ctx->sk_nslices = 0;
    if (!ctx->sk_prepared && (rc = sk_prepare(ctx))) return rc;
    const SkParams& sp = ctx->sk_sp;
    if (!ctx->sk_slices || want < 2) return DSKGPU_OK;          // exact layout (small input, or a slice overflowed before): one piece
    const u32 S = std::min<u32>(want, sp.nchunks);
    if (S < 2) return DSKGPU_OK;
    for (u32 sl = 0; sl < S; ++sl) {
        u32 cb, ce; sk_slice_range(sp, S, sl, &cb, &ce);
        for (u32 o = 0; o < sp.G; ++o) send_words[(size_t)sl * sp.G + o] = (u64)(ce - cb) * sp.slice * sp.R;
    }
    for (u32 o = 0; o < sp.G; ++o) kmers_est[o] = ctx->h_sk_est[o];
    *nslices = S; ctx->sk_nslices = S;
    return DSKGPU_OK;
}

// launch the scatter of slice s (asynchronous on the context's stream: the caller records an event behind it and starts the exchange)
int sk_scatter_slice(dskgpu_ctx* ctx, void* d_send, uint64_t capacity_words, u32 sl) {
    if (!ctx->sk_prepared || !ctx->sk_slices || sl >= ctx->sk_nslices) return fail(ctx, DSKGPU_E_STATE, "dskgpu_mg_scatter_slice without dskgpu_mg_slices_prepare");
    SkParams sp = ctx->sk_sp;
    if (capacity_words < ctx->h_rstart[sp.G] * sp.R) return fail(ctx, DSKGPU_E_ARG, "send buffer too small");
    if (sl == 0) {
        ctx->marks.clear(); ctx->ev_used = 0;
        ctx->mark("start");
        CK(hipMemsetAsync(ctx->sk_sent.p, 0, SK_MAX_OWNERS * 8, ctx->stream));
        CK(hipMemsetAsync(ctx->sk_sent.as<u64>() + 2 * SK_MAX_OWNERS, 0, 8, ctx->stream));
    }
    u32 cb, ce; sk_slice_range(sp, ctx->sk_nslices, sl, &cb, &ce);
    sp.c0 = cb; sp.c0g = cb; sp.clen = ce - cb; sp.rbase = (u64)cb * sp.G * sp.slice;
    // (the overflow flag of a sliced step lives apart from the scalars: the receiver's pipeline, which runs before the flag is
    //  read, resets those)
    if (ce > cb) {
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_scatter<true, K_, M_>), dim3(ce - cb), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, \
                                          (const unsigned long long*)nullptr, static_cast<u64*>(d_send), reinterpret_cast<u32*>(ctx->sk_sent.as<u64>() + 2 * SK_MAX_OWNERS), \
                                          ctx->sk_sent.as<unsigned long long>())
        SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    }
    CKL("k_sk_scatter");
    if (sl + 1 == ctx->sk_nslices) ctx->mark("mg_scatter");
    return DSKGPU_OK;
}

// end of the sender's part: did a slice of the send layout overflow (then the records of this step are incomplete -- every rank
// repeats the step in one piece; this context will use exact counts), and the k-mers that were packed
int sk_slices_finish(dskgpu_ctx* ctx, int* overflowed) {
    if (!ctx->sk_nslices) return fail(ctx, DSKGPU_E_STATE, "dskgpu_mg_slices_finish without dskgpu_mg_slices_prepare");
    CK(hipMemcpyAsync(&ctx->h_ovf1, ctx->sk_sent.as<u64>() + 2 * SK_MAX_OWNERS, 4, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(ctx->h_sk_sent, ctx->sk_sent.p, SK_MAX_OWNERS * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    *overflowed = ctx->h_ovf1 ? 1 : 0;
    if (ctx->h_ovf1) ctx->sk_exact = true;
    ctx->sk_prepared = false; ctx->sk_nslices = 0;
    return DSKGPU_OK;
}

// Receiver: records -> dense mixed keys -> the ordinary partition + count over a key array.
template <int W>
int sk_count(dskgpu_ctx* ctx, const u64* d_rec, u64 recv_words, u64 n_kmers_hint, bool hint_is_estimate = false) {
    typedef typename KeyT<W>::T Key;
    const u32 R = ctx->sk_sp.R;
    if (recv_words % R) return fail(ctx, DSKGPU_E_ARG, "recv_words is not a whole number of super-k-mer records");
    const u64 nrec = recv_words / R;
    if (hint_is_estimate && ctx->marks.size() > 1) {      // a sliced step: the sender's marks are still open (its launches returned at once)
        CK(hipStreamSynchronize(ctx->stream));
        ctx->resolve_marks();
    }
    ctx->marks.clear(); ctx->ev_used = 0;
    ctx->mark("start");
    u64 total = 0;
    ctx->rec_hint = 0; ctx->rec_sized = false; ctx->rec_hint_est = hint_is_estimate;
    if (nrec) {
        ctx->rec_src = d_rec; ctx->rec_n = nrec; ctx->rec_expanded = false;
        if (n_kmers_hint) { total = n_kmers_hint; ctx->rec_hint = n_kmers_hint; }      // the senders counted while they wrote the records
        else { const int e = sk_sizes(ctx, &total); if (e) return e; }
    } else {
        ctx->rec_src = nullptr; ctx->rec_n = 0;
        CK(ctx->sk_keys.ensure(sizeof(Key)));
    }
    ctx->mark("mg_sizes");
    CK(hipStreamSynchronize(ctx->stream));
    ctx->resolve_marks();
    if (!ctx->rec_src) { const int e = rec_gate_all(ctx); if (e) return e; return run_pipeline<W>(ctx, false, ctx->sk_keys.as<Key>(), 0); }
    int rc = run_pipeline<W>(ctx, false, nullptr, total);      // nullptr: keys come from ctx->rec_src
    if (rc == REC_RESIZE) { ctx->rec_hint_est = false; rc = run_pipeline<W>(ctx, false, nullptr, ctx->rec_hint); ctx->rec_hint = 0; }
    ctx->rec_src = nullptr;
    if (rc == DSKGPU_OK && ctx->rec_hint && !ctx->rec_hint_est && ctx->stats.n_kmers != ctx->rec_hint) {
        ctx->have_result = false;
        return fail(ctx, DSKGPU_E_ARG, "dskgpu_mg_count_sized: n_kmers does not match the k-mers inside the records");
    }
    return rc;
}



}  // namespace

namespace {

// Multi-bank count: every bank is counted on its own (all distinct k-mers kept), the per-bank rows are
// united and sorted by k-mer, and k_merge_banks applies the solidity kind / builds the histograms.  In steps, so that the
// in-process group (group.hip) can put its own count -- scatter, exchange, mg_count -- between them:
//   banks_begin   the per-bank counts keep every k-mer (abundance window 1 .. max, rows unsorted); the union is empty
//   banks_select  the context's read stream = bank b of the stream it was given (or all of it again: b = ~0u)
//   banks_add     the rows of the count just finished join the union as bank b
//   banks_finish  configuration and read stream restored; union sorted by k-mer, merged -> the result of the context

u32 banks_of(dskgpu_ctx* ctx) {
    std::vector<u64> ends = ctx->bank_ends;
    if (ends.empty() || ends.back() < ctx->n_bytes) ends.push_back(ctx->n_bytes);
    return (u32)ends.size();
}
int banks_begin(dskgpu_ctx* ctx) {
    BankJob& j = ctx->bank_job;
    j.ends = ctx->bank_ends;
    if (j.ends.empty() || j.ends.back() < ctx->n_bytes) j.ends.push_back(ctx->n_bytes);
    if (j.ends.size() > 32) return fail(ctx, DSKGPU_E_ARG, "at most 32 banks are supported by the solidity kinds");
    if (ctx->enc_keep) return fail(ctx, DSKGPU_E_STATE, "per-bank counts (-solidity-kind, -histo2D) need the reads themselves: not after dskgpu_encode_reads");
    j.cfg = ctx->cfg; j.base = ctx->d_reads; j.total = ctx->n_bytes;
    j.nu = 0; j.tot_kmers = 0; j.passes = 1; j.retries = 0; j.active = true;
    ctx->cfg.abundance_min = 1; ctx->cfg.abundance_max = 0xFFFFFFFFu; ctx->cfg.flags |= DSKGPU_F_NO_SORT;
    return DSKGPU_OK;
}
void banks_select(dskgpu_ctx* ctx, u32 b) {
    BankJob& j = ctx->bank_job;
    if (!j.active) return;
    if (b >= j.ends.size()) { ctx->d_reads = j.base; ctx->n_bytes = j.total; }
    else { const u64 beg = b ? j.ends[b - 1] : 0; ctx->d_reads = j.base + beg; ctx->n_bytes = j.ends[b] - beg; }
    ctx->enc_fresh = false; ctx->sk_prepared = false;
}
void banks_abort(dskgpu_ctx* ctx) {      // (an error inside a per-bank count: the context gets its configuration and its read stream back)
    BankJob& j = ctx->bank_job;
    if (!j.active) return;
    ctx->cfg = j.cfg; ctx->d_reads = j.base; ctx->n_bytes = j.total; j.active = false;
}
template <int W>
int banks_add(dskgpu_ctx* ctx, u32 b) {
    BankJob& j = ctx->bank_job;
    const u64 n = ctx->n_rows, nu = j.nu;
    j.tot_kmers += ctx->stats.n_kmers; j.passes = std::max<u32>(j.passes, (u32)ctx->stats.n_passes); j.retries += ctx->stats.n_retries;
    bool nomem = ctx->u_val.ensure_keep((nu + n + 1) * 8, nu * 8, ctx->stream) != 0;
    for (int x = 0; x < W; ++x) nomem = nomem || ctx->u_w[x].ensure_keep((nu + n + 1) * 8, nu * 8, ctx->stream) != 0;
    if (nomem) return fail(ctx, DSKGPU_E_NOMEM, "bank rows");
    if (n) {
        for (int x = 0; x < W; ++x)
            CK(hipMemcpyAsync(ctx->u_w[x].as<u64>() + nu, ctx->res_w[x], n * 8, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(k_pack_bank, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ctx->u_val.as<u64>() + nu, ctx->res_ab, n, b);
        CK(hipStreamSynchronize(ctx->stream));
    }
    j.nu += n;
    return DSKGPU_OK;
}
template <int W>
int banks_finish(dskgpu_ctx* ctx) {
    BankJob& j = ctx->bank_job;
    const u32 B = (u32)j.ends.size();
    const u64 nu = j.nu, total = j.total, tot_kmers = j.tot_kmers; const u32 passes = j.passes, retries = j.retries;
    banks_abort(ctx);                          // configuration and read stream back
    ctx->have_result = false;
    int rc = DSKGPU_OK;
    if (nu >= 0xFFFF0000ull) return fail(ctx, DSKGPU_E_ARG, "too many distinct k-mers over the banks for the merge");
    // ---- sort the union by k-mer
    CK(ctx->s_w[0].ensure((nu + 1) * 8)); CK(ctx->s_val.ensure((nu + 1) * 8));
    const unsigned gb = (unsigned)std::max<u64>(1, (nu + 255) / 256);
    if (nu) {
        size_t tmp = 0;
        if (W == 1) {
            const unsigned end_bit = std::min(64u, 2u * ctx->cfg.kmer_size);
            // LIBRARY SORT (rocprim), labelled: bank merges (-solidity-kind other than sum / -histo2D) order (value, bank) pairs once per job -- not on the sum path
            CK(rocprim::radix_sort_pairs(nullptr, tmp, ctx->u_w[0].as<u64>(), ctx->s_w[0].as<u64>(), ctx->u_val.as<u64>(), ctx->s_val.as<u64>(), (size_t)nu, 0u, end_bit, ctx->stream));
            CK(ctx->srt_tmp.ensure(tmp));
            CK(rocprim::radix_sort_pairs(ctx->srt_tmp.p, tmp, ctx->u_w[0].as<u64>(), ctx->s_w[0].as<u64>(), ctx->u_val.as<u64>(), ctx->s_val.as<u64>(), (size_t)nu, 0u, end_bit, ctx->stream));
        } else {
            if ((rc = sort_index_multiword(ctx, ctx->u_w, nu, W))) return rc;
            const u32* idx = ctx->srt_idx.as<u32>();
            for (int x = 0; x < W; ++x) {
                CK(ctx->s_w[x].ensure((nu + 1) * 8));
                hipLaunchKernelGGL(k_gather<u64>, dim3(gb), dim3(256), 0, ctx->stream, ctx->s_w[x].as<u64>(), ctx->u_w[x].as<u64>(), idx, nu);
            }
            CK(ctx->s_val.ensure((nu + 1) * 8));
            hipLaunchKernelGGL(k_gather<u64>, dim3(gb), dim3(256), 0, ctx->stream, ctx->s_val.as<u64>(), ctx->u_val.as<u64>(), idx, nu);
        }
        CKL("bank sort");
    }
    // ---- merge: solidity + histograms
    const size_t nh = (size_t)ctx->cfg.histo_max + 1;
    CK(ctx->m_flag.ensure((nu + 2) * 4)); CK(ctx->m_pos.ensure((nu + 2) * 4)); CK(ctx->m_sum.ensure((nu + 2) * 4));
    CK(ctx->gh2d.ensure(nh * 11 * 8));
    CK(hipMemsetAsync(ctx->ghist.p, 0, nh * 8, ctx->stream));
    CK(hipMemsetAsync(ctx->gh2d.p, 0, nh * 11 * 8, ctx->stream));
    CK(hipMemsetAsync(ctx->gstats.p, 0, 32, ctx->stream));
    MergeParams mp{B, ctx->cfg.solidity_kind, ctx->cfg.solidity_custom, ctx->cfg.abundance_min, ctx->cfg.abundance_max, ctx->cfg.histo_max,
                   (ctx->cfg.flags & DSKGPU_F_HISTO2D) ? 1u : 0u};
    u64 n_solid = 0;
    if (nu) {
        RowsIn ri{}; RowsOut ro{};
        for (int x = 0; x < W; ++x) ri.w[x] = ctx->s_w[x].as<u64>();
        hipLaunchKernelGGL(k_merge_banks<W>, dim3(gb), dim3(256), 0, ctx->stream, ri,
                           ctx->s_val.as<u64>(), nu, mp, ctx->m_flag.as<u32>(), ctx->m_sum.as<u32>(), ctx->ghist.as<u64>(), ctx->gh2d.as<u64>(), ctx->gstats.as<u64>());
        hipLaunchKernelGGL(k_copy_u32, dim3(gb), dim3(256), 0, ctx->stream, ctx->m_pos.as<u32>(), ctx->m_flag.as<u32>(), nu);
        CKL("k_merge_banks");
        ctx->h_sc[SC_F] = (u32)nu;
        CK(hipMemcpyAsync(ctx->scalars.as<u32>() + SC_F, &ctx->h_sc[SC_F], 4, hipMemcpyHostToDevice, ctx->stream));
        if ((rc = run_scan(ctx, ctx->m_pos.as<u32>(), ctx->scalars.as<u32>() + SC_F, nu))) return rc;
        CK(hipMemcpyAsync(&ctx->h_back[1], ctx->m_pos.as<u32>() + nu, 4, hipMemcpyDeviceToHost, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        n_solid = ctx->h_back[1];
        CK(ctx->out_ab.ensure((n_solid + 1) * 4));
        for (int x = 0; x < W; ++x) { CK(ctx->out_w[x].ensure((n_solid + 1) * 8)); ro.w[x] = ctx->out_w[x].as<u64>(); }
        hipLaunchKernelGGL(k_pick_rows<W>, dim3(gb), dim3(256), 0, ctx->stream, ri,
                           ctx->m_sum.as<u32>(), ctx->m_flag.as<u32>(), ctx->m_pos.as<u32>(), nu, ro, ctx->out_ab.as<u32>());
        CKL("k_pick_rows");
    }
    ctx->hist.assign(nh, 0); ctx->hist2d.assign(nh * 11, 0);
    CK(hipMemcpyAsync(ctx->hist.data(), ctx->ghist.p, nh * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(ctx->hist2d.data(), ctx->gh2d.p, nh * 11 * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipMemcpyAsync(&ctx->h_stats[0], ctx->gstats.p, 32, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    for (int x = 0; x < 4; ++x) ctx->res_w[x] = x < W ? ctx->out_w[x].as<u64>() : nullptr;
    ctx->res_ab = ctx->out_ab.as<u32>();
    ctx->n_rows = n_solid;
    ctx->stats.n_bytes = total; ctx->stats.n_kmers = tot_kmers; ctx->stats.n_distinct = ctx->h_stats[0]; ctx->stats.n_solid = n_solid;
    ctx->stats.n_passes = passes; ctx->stats.n_retries = retries;
    ctx->stats.n_partitions = ctx->cfg.nb_partitions ? ctx->cfg.nb_partitions : 4u;
    ctx->have_result = true;
    return DSKGPU_OK;
}
template <int W>
int run_banks(dskgpu_ctx* ctx) {
    int rc = banks_begin(ctx);
    if (rc) return rc;
    const u32 B = (u32)ctx->bank_job.ends.size();
    for (u32 b = 0; b < B; ++b) {
        banks_select(ctx, b);
        if ((rc = run_pipeline<W>(ctx, true, nullptr, 0)) || (rc = banks_add<W>(ctx, b))) { banks_abort(ctx); return rc; }
    }
    return banks_finish<W>(ctx);
}

}  // namespace

// ---- the same steps for group.hip (one library, not part of the C-ABI)
__attribute__((visibility("hidden"))) bool dskgpu_i_per_bank(dskgpu_ctx* ctx) {
    const bool banks = ctx->bank_ends.size() > 1 || (!ctx->bank_ends.empty() && ctx->bank_ends.back() < ctx->n_bytes);
    return banks && (ctx->cfg.solidity_kind != DSKGPU_SOLIDITY_SUM || (ctx->cfg.flags & DSKGPU_F_HISTO2D));
}
__attribute__((visibility("hidden"))) uint32_t dskgpu_i_banks(dskgpu_ctx* ctx) { return banks_of(ctx); }
__attribute__((visibility("hidden"))) int dskgpu_i_banks_begin(dskgpu_ctx* ctx) { return banks_begin(ctx); }
__attribute__((visibility("hidden"))) void dskgpu_i_banks_select(dskgpu_ctx* ctx, uint32_t b) { banks_select(ctx, b); }
__attribute__((visibility("hidden"))) void dskgpu_i_banks_abort(dskgpu_ctx* ctx) { banks_abort(ctx); }
__attribute__((visibility("hidden"))) int dskgpu_i_banks_add(dskgpu_ctx* ctx, uint32_t b) {
    return ctx->W == 1 ? banks_add<1>(ctx, b) : ctx->W == 2 ? banks_add<2>(ctx, b) : banks_add<4>(ctx, b);
}
__attribute__((visibility("hidden"))) int dskgpu_i_banks_finish(dskgpu_ctx* ctx) {
    return ctx->W == 1 ? banks_finish<1>(ctx) : ctx->W == 2 ? banks_finish<2>(ctx) : banks_finish<4>(ctx);
}

// =============================================================== C-ABI
static_assert(((size_t)32 << 20) / RP_BLOCK <= (size_t)RP_NT * RP_SCAN_PER, "k_rp_scan walks the block summaries of one staging piece (PIN_CHUNK) with RP_SCAN_PER per thread");
template <int FMT>
static void launch_raw_chunk(dskgpu_ctx* ctx, u32 n, uint8_t* out) {
    const u32 nb = (n + RP_BLOCK - 1) / RP_BLOCK;
    const unsigned char* in = ctx->raw_in.as<unsigned char>();
    RawState* st = ctx->raw_state.as<RawState>();
    hipLaunchKernelGGL((k_rp_count<FMT>), dim3(nb), dim3(RP_NT), 0, ctx->stream, in, n, ctx->raw_blk.as<RpBlock>());
    hipLaunchKernelGGL((k_rp_scan<FMT>), dim3(1), dim3(RP_NT), 0, ctx->stream, in, n, nb, ctx->raw_blk.as<RpBlock>(), st,
                       ctx->raw_boff.as<unsigned long long>(), ctx->raw_bstate.as<u32>(), out);
    hipLaunchKernelGGL((k_rp_write<FMT>), dim3(nb), dim3(RP_NT), 0, ctx->stream, in, n, ctx->raw_boff.as<unsigned long long>(),
                       ctx->raw_bstate.as<u32>(), st, out);
}

extern "C" {

const char* dskgpu_version(void) { return DSKGPU_VERSION; }

int dskgpu_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

const char* dskgpu_last_error(const dskgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int dskgpu_create(const dskgpu_config* cfg, dskgpu_ctx** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return DSKGPU_E_ARG; }
    *out = nullptr;
    if (cfg->kmer_size < 1 || cfg->kmer_size > 128) { g_create_err = "kmer_size must be in 1..128"; return DSKGPU_E_ARG; }
    if (cfg->solidity_kind > DSKGPU_SOLIDITY_CUSTOM) { g_create_err = "unknown solidity_kind"; return DSKGPU_E_ARG; }
    if (g_place_k < 0) { const char* e = getenv("DSKGPU_PLACE"); g_place_k = e ? atoi(e) : 0; }
    if ((cfg->flags & DSKGPU_F_PLACE) && g_place_k < 2) g_place_k = 8;
    const u32 ws = cfg->world_size ? cfg->world_size : 1;
    if ((ws & (ws - 1)) != 0 || ws > 64 || cfg->rank >= ws) { g_create_err = "world_size must be a power of two <= 64 and rank < world_size"; return DSKGPU_E_ARG; }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) { g_create_err = std::string("no HIP device: ") + hipGetErrorString(e); return DSKGPU_E_DEVICE; }
    if (cfg->device < 0 || cfg->device >= ndev) { g_create_err = "bad device ordinal"; return DSKGPU_E_ARG; }
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); return DSKGPU_E_DEVICE; }
    dskgpu_ctx* ctx = new dskgpu_ctx();
    ctx->cfg = *cfg;
    ctx->cfg.world_size = ws;
    if (ctx->cfg.histo_max == 0) ctx->cfg.histo_max = 10000;
    if (ctx->cfg.abundance_max == 0) ctx->cfg.abundance_max = 0x7FFFFFFFu;
    if (ctx->cfg.minimizer_size == 0) ctx->cfg.minimizer_size = 10;
    ctx->W = cfg->kmer_size <= 32 ? 1 : cfg->kmer_size <= 64 ? 2 : 4;     // device keys: 1, 2 or 4 words (three-word keys for 65 <= k <= 96 were built and measured in r06: slower than the four-word path)
    ctx->words_out = (int)((cfg->kmer_size + 31) / 32);
    ctx->sentinel_ok = !sentinel_is_a_kmer(ctx->W, cfg->kmer_size);                   // words of a k-mer at the ABI (3 for k <= 96)
    ctx->max_keys_per_pass = (u64)cfg->max_pass_mkeys * 1000000ull;
    ctx->tune.read();
    // super-k-mer records need >= 16 m-mers per window (superkmer.h); shorter k-mers travel as explicit keys
    // (world_size == 1 is the degenerate exchange: every record goes to owner 0; dskgpu_count never looks at sk_mode)
    ctx->sk_mode = cfg->kmer_size >= 20 && cfg->kmer_size <= 64 && !(cfg->flags & DSKGPU_F_MG_EXPLICIT);
    if (ctx->sk_mode) {
        ctx->sk_sp.k = cfg->kmer_size; ctx->sk_sp.G = ws;
        ctx->sk_sp.m = std::min<u32>(std::min<u32>(ctx->cfg.minimizer_size, 16u), cfg->kmer_size - 15u);
        ctx->sk_sp.R = sk_record_words(cfg->kmer_size);
    }
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) ctx->num_cu = cus; }      // (one attribute, not the whole property block)
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_err = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete ctx; return DSKGPU_E_DEVICE; }
    ctx->own_stream = true;
    *out = ctx;
    return DSKGPU_OK;
}

void dskgpu_destroy(dskgpu_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->cfg.device);
    (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->reads_own, &ctx->raw_in, &ctx->raw_blk, &ctx->raw_boff, &ctx->raw_bstate, &ctx->raw_state, &ctx->packed, &ctx->inval, &ctx->bufA, &ctx->bufB, &ctx->mat1, &ctx->mat2, &ctx->sums,
                      &ctx->descs1, &ctx->descs2, &ctx->seg, &ctx->fstart, &ctx->nsolid, &ctx->scalars, &ctx->ghist, &ctx->gstats, &ctx->chain_next, &ctx->smp_mat, &ctx->smp_descs, &ctx->boff, &ctx->hv_lut, &ctx->hv_collect, &ctx->hv_buf, &ctx->dbg, &ctx->l0buf,
                      &ctx->out_ab, &ctx->srt_ab, &ctx->srt_tmp,
                      &ctx->srt_idx, &ctx->srt_idx2, &ctx->srt_k, &ctx->srt_k2, &ctx->abund2, &ctx->acc_ab, &ctx->u_val,
                      &ctx->s_val, &ctx->m_flag, &ctx->m_pos, &ctx->m_sum, &ctx->gh2d,
                      &ctx->sk_sums, &ctx->sk_cbase, &ctx->sk_keys, &ctx->sk_table, &ctx->sk_load, &ctx->sk_sent, &ctx->cur_state, &ctx->rs_ovs, &ctx->smp_keys, &ctx->sk_lay, &ctx->fix_list, &ctx->sk_cb64, &ctx->rs_del, &ctx->rs_lens, &ctx->back_dev};
    if (ctx->back_host) (void)hipHostFree(ctx->back_host);
    if (ctx->hist_pin) (void)hipHostFree(ctx->hist_pin);
    if (ctx->h_part_off) (void)hipHostFree(ctx->h_part_off);
    if (ctx->land) (void)hipHostFree(ctx->land);
    for (DevBuf* b : bufs) b->release();
    for (int i = 0; i < 4; ++i) { ctx->rs_g[i].release(); ctx->out_w[i].release(); ctx->srt_w[i].release(); ctx->acc_w[i].release(); ctx->u_w[i].release(); ctx->s_w[i].release(); }
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) { if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]); if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int dskgpu_set_stream(dskgpu_ctx* ctx, void* hip_stream) {
    if (!ctx) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    CK(hipStreamSynchronize(ctx->stream));
    if (hip_stream) {
        if (ctx->own_stream) { (void)hipStreamDestroy(ctx->stream); ctx->own_stream = false; }
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
    } else if (!ctx->own_stream) {
        CK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return DSKGPU_OK;
}

#define PIN_CHUNK ((size_t)32 << 20)
static_assert(PIN_CHUNK == ((size_t)32 << 20), "launch_raw_chunk's capacity check is written for 32 MB pieces");
// pageable -> pinned: one thread copies ~10 GB/s on the host this was measured on, the link takes 55: big pieces are copied by up to eight (a memory-mapped file's pages are also faulted in by the copy)
static void stage_copy(void* dst, const void* src, size_t n) {
    const unsigned T = n >= ((size_t)16 << 20) ? 8u : n >= ((size_t)8 << 20) ? 4u : n >= ((size_t)2 << 20) ? 2u : 1u;
    if (T == 1) { std::memcpy(dst, src, n); return; }
    const size_t per = (((n + T - 1) / T) + 4095) & ~(size_t)4095;          // T * per >= n
    auto part = [=](unsigned t) { const size_t a = (size_t)t * per; if (a < n) std::memcpy((char*)dst + a, (const char*)src + a, std::min(per, n - a)); };
    std::thread th[7];
    unsigned started = 1;
    try { for (; started < T; ++started) th[started - 1] = std::thread(part, started); }
    catch (const std::system_error&) {}                      // (no more threads to be had: the parts that got none are copied here)
    part(0);
    for (unsigned t = started; t < T; ++t) part(t);
    for (unsigned t = 1; t < started; ++t) th[t - 1].join();
}

static int ensure_pinned(dskgpu_ctx* ctx) {      // the two pinned staging buffers of dskgpu_push_reads (also set up by dskgpu_reserve_reads: off the first push's path)
    if (ctx->pin[0] && ctx->pin[1]) return DSKGPU_OK;
    for (int i = 0; i < 2; ++i) {
        if (!ctx->pin[i]) CK(hipHostMalloc(&ctx->pin[i], PIN_CHUNK, hipHostMallocDefault));
        if (!ctx->pin_ev[i]) CK(hipEventCreateWithFlags(&ctx->pin_ev[i], hipEventDisableTiming));
    }
    return DSKGPU_OK;
}

static int raw_finish(dskgpu_ctx* ctx, u64* lines);
int dskgpu_push_reads(dskgpu_ctx* ctx, const char* bytes, uint64_t nbytes) {
    if (!ctx || (!bytes && nbytes)) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    if (ctx->raw_pending) { const int e = raw_finish(ctx, nullptr); if (e) return e; }
    const u64 need = ctx->reads_len + nbytes + 1;
    if (need > ctx->reads_own.cap) {
        DevBuf nb;
        CK(nb.ensure(std::max<u64>(need, std::max<u64>(ctx->reads_own.cap * 2, (u64)256 << 20))));
        if (ctx->reads_len) CK(hipMemcpyAsync(nb.p, ctx->reads_own.p, ctx->reads_len, hipMemcpyDeviceToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        ctx->reads_own.release();
        ctx->reads_own = nb;
    }
    uint8_t* dst = ctx->reads_own.as<uint8_t>();
    if (nbytes) {
        // pageable host memory -> two pinned staging buffers -> HBM: the CPU copy of one chunk
        // overlaps the DMA of the previous one (a direct pageable hipMemcpy reached 4.5 GB/s)
        const size_t CH = PIN_CHUNK;
        { const int e = ensure_pinned(ctx); if (e) return e; }
        for (u64 off = 0; off < nbytes; off += CH, ctx->pin_next ^= 1) {
            const int slot = ctx->pin_next;       // (alternates ACROSS calls too: the copy of this call's first piece overlaps the DMA of the last call's last one)
            const size_t len = (size_t)std::min<u64>(CH, nbytes - off);
            if (ctx->pin_used[slot]) CK(hipEventSynchronize(ctx->pin_ev[slot]));
            stage_copy(ctx->pin[slot], bytes + off, len);
            CK(hipMemcpyAsync(dst + ctx->reads_len + off, ctx->pin[slot], len, hipMemcpyHostToDevice, ctx->stream));
            CK(hipEventRecord(ctx->pin_ev[slot], ctx->stream));
            ctx->pin_used[slot] = true;
        }
    }
    CK(hipMemsetAsync(dst + ctx->reads_len + nbytes, '\n', 1, ctx->stream));
    // (no synchronisation here: the caller's bytes were copied to the pinned staging buffers above, and everything that reads the
    //  device copy -- the count, a later growth of the buffer -- is ordered behind the DMA on the context's stream.  A bank that
    //  hands over a few MB per call keeps the link busy this way instead of paying a round trip per call.)
    ctx->reads_len += nbytes + 1;
    ctx->d_reads = dst; ctx->n_bytes = ctx->reads_len; ctx->enc_keep = false; ctx->enc_fresh = false; ctx->sk_prepared = false; ctx->sk_exact = false; ctx->opt2_off = false; ctx->opt1_off = false; ctx->mw_v3_off = false; ctx->rec_l0_off = false; ctx->last_rows = 0;
    return DSKGPU_OK;
}

// The raw pushes' result: the stream's length comes back from the device (the one synchronisation of a raw ingest), the terminator is set.
static int raw_finish(dskgpu_ctx* ctx, u64* lines) {
    if (!ctx->raw_pending) return DSKGPU_OK;
    RawState s;
    CK(hipMemcpyAsync(&s, ctx->raw_state.p, sizeof(RawState), hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->raw_pending = false;
    uint8_t* dst = ctx->reads_own.as<uint8_t>();
    if (!rp_file_ok(s)) s.bad = 1;      // (the last file, now that it is complete: its quality lines must add up to its sequence lines)
    if (s.bad) {             // the text is not what the device parser handles: the raw pushes are dropped, the stream is what it was before them
        ctx->reads_len = ctx->raw_base;
        ctx->d_reads = dst; ctx->n_bytes = ctx->reads_len;
        return fail(ctx, DSKGPU_E_FORMAT, "dskgpu_push_raw: the text is not 4-line FASTQ / FASTA as declared (the raw pushes were dropped: parse on the host and push the reads)");
    }
    if (s.out_len + 1 > ctx->reads_own.cap) return fail(ctx, DSKGPU_E_STATE, "dskgpu_push_raw: stream longer than its bound");
    CK(hipMemsetAsync(dst + s.out_len, '\n', 1, ctx->stream));
    ctx->reads_len = s.out_len + 1;
    if (lines) *lines = s.recs;
    ctx->d_reads = dst; ctx->n_bytes = ctx->reads_len; ctx->enc_keep = false; ctx->enc_fresh = false; ctx->sk_prepared = false; ctx->sk_exact = false; ctx->opt2_off = false; ctx->opt1_off = false; ctx->mw_v3_off = false; ctx->rec_l0_off = false; ctx->last_rows = 0;
    return DSKGPU_OK;
}
#define RAW_SYNC(ctx) do { if ((ctx)->raw_pending) { const int e_ = raw_finish(ctx, nullptr); if (e_) return e_; } } while (0)

int dskgpu_raw_finish(dskgpu_ctx* ctx, uint64_t* stream_bytes, uint64_t* records) {
    if (!ctx) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    u64 ln = 0;
    const int rc = raw_finish(ctx, &ln);
    if (stream_bytes) *stream_bytes = ctx->reads_len;
    if (records) *records = ln;
    return rc;
}

int dskgpu_push_raw(dskgpu_ctx* ctx, const char* text, uint64_t nbytes, int format, int new_file) {
    if (!ctx || (!text && nbytes) || (format != DSKGPU_RAW_FASTA && format != DSKGPU_RAW_FASTQ)) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    if (!ctx->raw_pending) {
        CK(ctx->raw_state.ensure(sizeof(RawState)));
        CK(ctx->raw_in.ensure(PIN_CHUNK));
        const u64 maxb = PIN_CHUNK / RP_BLOCK;
        CK(ctx->raw_blk.ensure(maxb * sizeof(RpBlock)));
        CK(ctx->raw_boff.ensure(maxb * 8));
        CK(ctx->raw_bstate.ensure((maxb + 1) * 4));
        ctx->raw_base = ctx->raw_ub = ctx->reads_len;
        hipLaunchKernelGGL(k_rp_init, dim3(1), dim3(1), 0, ctx->stream, ctx->raw_state.as<RawState>(), (unsigned long long)ctx->reads_len);
        CKL("k_rp_init");
        ctx->raw_pending = true;
    } else if (new_file) {
        hipLaunchKernelGGL(k_rp_fresh, dim3(1), dim3(1), 0, ctx->stream, ctx->raw_state.as<RawState>());
        CKL("k_rp_fresh");
    }
    // what the text can leave at most: every byte, the separator in front of a new file, the terminator
    const u64 need = ctx->raw_ub + nbytes + 2;
    if (need > ctx->reads_own.cap) {
        DevBuf nb;
        CK(nb.ensure(std::max<u64>(need, std::max<u64>(ctx->reads_own.cap * 2, (u64)256 << 20))));
        if (ctx->raw_ub) CK(hipMemcpyAsync(nb.p, ctx->reads_own.p, ctx->raw_ub, hipMemcpyDeviceToDevice, ctx->stream));
        CK(hipStreamSynchronize(ctx->stream));
        ctx->reads_own.release();
        ctx->reads_own = nb;
    }
    uint8_t* out = ctx->reads_own.as<uint8_t>();
    if (nbytes) {
        const size_t CH = PIN_CHUNK;
        { const int e = ensure_pinned(ctx); if (e) return e; }
        for (u64 off = 0; off < nbytes; off += CH, ctx->pin_next ^= 1) {
            const int slot = ctx->pin_next;
            const size_t len = (size_t)std::min<u64>(CH, nbytes - off);
            if (ctx->pin_used[slot]) CK(hipEventSynchronize(ctx->pin_ev[slot]));
            stage_copy(ctx->pin[slot], text + off, len);
            CK(hipMemcpyAsync(ctx->raw_in.p, ctx->pin[slot], len, hipMemcpyHostToDevice, ctx->stream));      // (one device buffer: the stream orders the next piece's DMA behind this piece's kernels, which take a fraction of the DMA's time)
            CK(hipEventRecord(ctx->pin_ev[slot], ctx->stream));
            ctx->pin_used[slot] = true;
            if (format == DSKGPU_RAW_FASTQ) launch_raw_chunk<RP_FASTQ>(ctx, (u32)len, out);
            else launch_raw_chunk<RP_FASTA>(ctx, (u32)len, out);
            CKL("k_rp_write");
        }
    }
    ctx->raw_ub += nbytes + 1;
    return DSKGPU_OK;
}

int dskgpu_stream_bytes(dskgpu_ctx* ctx, uint64_t* stream_bytes) {
    if (!ctx || !stream_bytes) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    *stream_bytes = ctx->reads_len;
    return DSKGPU_OK;
}

int dskgpu_rewind_reads(dskgpu_ctx* ctx, uint64_t stream_bytes) {
    if (!ctx) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    if (ctx->enc_keep) return fail(ctx, DSKGPU_E_STATE, "dskgpu_rewind_reads: the reads were encoded and released (dskgpu_encode_reads)");
    if (stream_bytes > ctx->reads_len) return fail(ctx, DSKGPU_E_ARG, "dskgpu_rewind_reads: the stream is shorter than that");
    ctx->reads_len = stream_bytes;
    while (!ctx->bank_ends.empty() && ctx->bank_ends.back() > stream_bytes) ctx->bank_ends.pop_back();
    ctx->d_reads = ctx->reads_own.as<uint8_t>(); ctx->n_bytes = ctx->reads_len; ctx->enc_fresh = false; ctx->sk_prepared = false; ctx->sk_exact = false; ctx->opt2_off = false; ctx->opt1_off = false; ctx->mw_v3_off = false; ctx->rec_l0_off = false; ctx->last_rows = 0;
    return DSKGPU_OK;
}

int dskgpu_reserve_reads(dskgpu_ctx* ctx, uint64_t nbytes) {
    if (!ctx) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    { const int e = ensure_pinned(ctx); if (e) return e; }
    if (nbytes + 1 <= ctx->reads_own.cap) return DSKGPU_OK;
    DevBuf nb;
    CK(nb.ensure(nbytes + 1));
    if (ctx->reads_len) CK(hipMemcpyAsync(nb.p, ctx->reads_own.p, ctx->reads_len, hipMemcpyDeviceToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    ctx->reads_own.release();
    ctx->reads_own = nb;
    if (ctx->reads_len) ctx->d_reads = nb.as<uint8_t>();
    return DSKGPU_OK;
}

int dskgpu_reserve_work(dskgpu_ctx* ctx, uint64_t nbytes) {
    if (!ctx) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    // upper bounds from the byte count (every byte could end a k-mer): the encoded stream, the level-1 slices / the keys of a pass
    // (bufA), the level-2 regions + extension pool (bufB); DevBuf only ever grows, so dskgpu_count finds them in place
    const u64 max_keys = ctx->max_keys_per_pass ? ctx->max_keys_per_pass : 0xD0000000ull;
    const u64 n = std::min<u64>(nbytes + 1, max_keys + max_keys / 4);
    const u64 key = 8ull * (u64)ctx->W;
    const u64 nwords = (nbytes + 31) / 32;
    const u64 target = target_keys(ctx->W);
    const u64 F = n / target + 2, cap = opt_groups(ctx->W) * (8u / (u64)ctx->W);
    const u64 regions = F + (ctx->W == 1 ? F / 8 + 4096 : 0);
    {   // a reservation is a convenience: never more than 60 % of what is free (the sizes are upper bounds from a byte count; ranks that
        // share a device and DSKGPU_PLACE's candidates need room too) -- beyond that dskgpu_count sizes the buffers itself
        size_t free_b = 0, total_b = 0;
        const u64 want = (nwords + 1) * 12 + std::max<u64>((n + n / 8 + (1u << 20)) * key, ctx->W == 1 ? F * cap * 4 : 0) + (regions * cap + (1u << 16)) * key;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const u64 have = ctx->packed.cap + ctx->inval.cap + ctx->bufA.cap + ctx->bufB.cap;
            if (want > have && want - have > (u64)free_b * 6 / 10) return fail(ctx, DSKGPU_NOT_RESERVED, "dskgpu_reserve_work: the reservation exceeds 60 % of the free device memory (nothing was reserved; dskgpu_count sizes its own buffers)");
        }
    }
    // (after dskgpu_encode_reads the encoded stream is the ONLY copy of the reads and DevBuf::ensure does not keep contents: it stays)
    if (!ctx->enc_keep) {
        CK(ctx->packed.ensure((nwords + 1) * 8));
        CK(ctx->inval.ensure((nwords + 1) * 4));
    }
    CK(ctx->bufA.ensure(std::max<u64>((n + n / 8 + (1u << 20)) * key, ctx->W == 1 ? F * cap * 4 : 0)));
    CK(ctx->bufB.ensure((regions * cap + (1u << 16)) * key));
    if (ctx->W > 1) CK(ctx->abund2.ensure((F * cap + (1u << 16)) * 4));
    return DSKGPU_OK;
}

int dskgpu_set_reads_device(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes) {
    if (!ctx || (!d_bytes && nbytes)) return DSKGPU_E_ARG;
    // The caller usually just PRODUCED these bytes on a stream of its own (torch's), and the context's stream is non-blocking: nothing
    // orders the two.  Wait for the device once, here, so that a count can never read reads that are still being written (the r03
    // intermittent failure of the multi-process test was exactly that, in the test's reference count).  Once per read set, not per count.
    CK(hipSetDevice(ctx->cfg.device));
    CK(hipDeviceSynchronize());
    ctx->raw_pending = false;
    ctx->d_reads = static_cast<const uint8_t*>(d_bytes); ctx->enc_keep = false; ctx->enc_fresh = false; ctx->sk_prepared = false; ctx->sk_exact = false; ctx->opt2_off = false; ctx->opt1_off = false; ctx->mw_v3_off = false; ctx->rec_l0_off = false; ctx->last_rows = 0;
    ctx->n_bytes = nbytes;
    ctx->reads_len = 0;
    ctx->bank_ends.clear();
    return DSKGPU_OK;
}

int dskgpu_encode_reads(dskgpu_ctx* ctx) {
    if (!ctx) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    if (ctx->enc_keep) return DSKGPU_OK;
    if (!ctx->d_reads && ctx->n_bytes) return fail(ctx, DSKGPU_E_STATE, "no reads to encode");
    u64 nwords = 0;
    const int rc = run_encode(ctx, ctx->d_reads, ctx->n_bytes, &nwords);
    if (rc) return rc;
    CK(hipStreamSynchronize(ctx->stream));
    ctx->enc_keep = true;
    ctx->d_reads = nullptr;                    // the caller's buffer is never read again
    ctx->reads_own.release(); ctx->reads_len = 0;      // pushed reads: their ASCII copy in HBM goes too (a later push starts a new read set)
    return DSKGPU_OK;
}

int dskgpu_count(dskgpu_ctx* ctx) {
    if (!ctx) return DSKGPU_E_ARG;
    if (ctx->cfg.world_size != 1) return fail(ctx, DSKGPU_E_STATE, "dskgpu_count needs world_size == 1; use dskgpu_mg_scatter/_mg_count");
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    ctx->stats = dskgpu_stats{};
    const bool banks = ctx->bank_ends.size() > 1 || (!ctx->bank_ends.empty() && ctx->bank_ends.back() < ctx->n_bytes);
    if (banks && (ctx->cfg.solidity_kind != DSKGPU_SOLIDITY_SUM || (ctx->cfg.flags & DSKGPU_F_HISTO2D)))
        return ctx->W == 1 ? run_banks<1>(ctx) : ctx->W == 2 ? run_banks<2>(ctx) : run_banks<4>(ctx);
    if (ctx->cfg.flags & DSKGPU_F_HISTO2D) ctx->hist2d.clear();
    if (ctx->W == 1) return run_pipeline<1>(ctx, true, nullptr, 0);
    if (ctx->W == 2) return run_pipeline<2>(ctx, true, nullptr, 0);
    return run_pipeline<4>(ctx, true, nullptr, 0);
}

int dskgpu_next_bank(dskgpu_ctx* ctx) {
    if (!ctx) return DSKGPU_E_ARG;
    if (ctx->raw_pending) { CK(hipSetDevice(ctx->cfg.device)); RAW_SYNC(ctx); }
    // (every call ends a bank, an empty one too: a rank of a group that was handed nothing of a small bank must count as many banks as
    //  the others -- until r06 a bank that added no bytes was not recorded: the ranks of `dsk -nb-gpus 4 -solidity-kind min` then ran
    //  different numbers of per-bank steps and waited for each other for ever; on one GPU an empty input file was not a bank at all)
    ctx->bank_ends.push_back(ctx->reads_len);
    return DSKGPU_OK;
}

int dskgpu_set_banks(dskgpu_ctx* ctx, const uint64_t* end_offsets, uint32_t n_banks) {
    if (!ctx || (!end_offsets && n_banks)) return DSKGPU_E_ARG;
    ctx->bank_ends.assign(end_offsets, end_offsets + n_banks);
    for (size_t i = 1; i < ctx->bank_ends.size(); ++i)
        if (ctx->bank_ends[i] < ctx->bank_ends[i - 1]) { ctx->bank_ends.clear(); return fail(ctx, DSKGPU_E_ARG, "bank end offsets must be non-decreasing"); }
    return DSKGPU_OK;
}

int dskgpu_histogram2d(const dskgpu_ctx* ctx, uint64_t* out, uint32_t nrows) {
    if (!ctx || !out) return DSKGPU_E_ARG;
    if (!ctx->have_result || ctx->hist2d.empty()) return DSKGPU_E_STATE;
    if (nrows != ctx->cfg.histo_max + 1) return DSKGPU_E_ARG;
    std::memcpy(out, ctx->hist2d.data(), ctx->hist2d.size() * 8);
    return DSKGPU_OK;
}

uint64_t dskgpu_mg_send_capacity_words(dskgpu_ctx* ctx) {
    if (!ctx) return 0;
    if (ctx->raw_pending && (hipSetDevice(ctx->cfg.device) != hipSuccess || raw_finish(ctx, nullptr) != DSKGPU_OK)) return 0;
    if (!ctx->sk_mode) return (ctx->n_bytes + 1) * (u64)ctx->W;
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) return 0;
    if (!ctx->sk_prepared && sk_prepare(ctx) != DSKGPU_OK) return 0;     // the error text stays in the ctx; dskgpu_mg_scatter reports it
    return ctx->h_rstart[ctx->sk_sp.G] * ctx->sk_sp.R + 1;
}

int dskgpu_mg_scatter(dskgpu_ctx* ctx, void* d_send, uint64_t capacity_words, uint64_t* send_words) {
    if (!ctx || !d_send || !send_words) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    if (ctx->sk_mode) return sk_scatter(ctx, d_send, capacity_words, send_words);
    // (explicit keys -- k < 20, k > 64 or DSKGPU_F_MG_EXPLICIT -- keep 32-bit key offsets in the send buffer: one key per byte at most)
    if (ctx->n_bytes >= 0xFFFF0000ull) return fail(ctx, DSKGPU_E_ARG, "explicit-key exchange: a rank's read shard must stay below 4.29 GB (super-k-mer records, 20 <= k <= 64, have no such limit)");
    if (capacity_words < dskgpu_mg_send_capacity_words(ctx)) return fail(ctx, DSKGPU_E_ARG, "send buffer too small");
    const int rc = ctx->W == 1 ? mg_scatter_impl<1>(ctx, d_send, send_words) : ctx->W == 2 ? mg_scatter_impl<2>(ctx, d_send, send_words) : mg_scatter_impl<4>(ctx, d_send, send_words);
    if (rc == DSKGPU_OK) for (u32 o = 0; o < ctx->cfg.world_size; ++o) ctx->h_sk_sent[o] = send_words[o] / (u64)ctx->W;
    return rc;
}

int dskgpu_mg_sample(dskgpu_ctx* ctx, uint64_t* loads) {
    if (!ctx || !loads) return DSKGPU_E_ARG;
    std::memset(loads, 0, (size_t)SK_BUCKETS * 8);
    if (!ctx->sk_mode) return DSKGPU_OK;            // explicit keys: the owner is a bit field of the k-mer hash, balanced by construction
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    u64 nwords = 0;
    int rc = encode_current(ctx, &nwords);
    if (rc) return rc;
    sk_geometry(ctx, nwords);
    SkParams sp = ctx->sk_sp;
    sp.sample_step = sp.tiles_per_chunk >= 16 ? 16u : 1u;        // every 16th tile of a large shard, all tiles of a small one
    sp.table = nullptr;
    CK(ctx->sk_load.ensure((size_t)SK_BUCKETS * 8));
    CK(hipMemsetAsync(ctx->sk_load.p, 0, (size_t)SK_BUCKETS * 8, ctx->stream));
#define SK_CALL(K_, M_) hipLaunchKernelGGL((k_sk_sample<K_, M_>), dim3(sp.nchunks), dim3(SK_NT), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(), sp, \
                                          ctx->sk_load.as<unsigned long long>())
    SK_DISPATCH(ctx, sp, SK_CALL);
#undef SK_CALL
    CKL("k_sk_sample");
    CK(hipMemcpyAsync(loads, ctx->sk_load.p, (size_t)SK_BUCKETS * 8, hipMemcpyDeviceToHost, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    for (u32 b = 0; b < SK_BUCKETS; ++b) loads[b] *= sp.sample_step;      // an estimate of the whole shard's load
    ctx->sk_prepared = false;                       // packed / inval were rewritten
    ctx->enc_fresh = true;                          // ... with the encoding of the current reads: the sender's sizing pass reuses it
    return DSKGPU_OK;
}

void dskgpu_mg_make_table(const uint64_t* loads, uint32_t world, uint8_t* table) {
    if (!table) return;
    if (world == 0) world = 1;
    default_table(world, table);
    if (!loads || world == 1) return;
    u64 total = 0;
    for (u32 b = 0; b < SK_BUCKETS; ++b) total += loads[b];
    if (total == 0) return;
    // a bucket that alone holds more than a quarter of an owner's fair share is split over all owners by k-mer
    const u64 heavy = std::max<u64>(1, total / world / 4);
    std::vector<u64> owner_load(world, 0);
    std::vector<u32> order;
    u64 split_load = 0;
    for (u32 b = 0; b < SK_BUCKETS; ++b) {
        if (loads[b] > heavy) { table[b] = (uint8_t)SK_SPLIT; split_load += loads[b]; }
        else if (loads[b]) order.push_back(b);        // (buckets the sample did not see keep their default owner)
    }
    for (u32 o = 0; o < world; ++o) owner_load[o] = split_load / world;
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return loads[a] > loads[b]; });    // largest first, ties by index
    for (u32 b : order) {
        u32 best = 0;
        for (u32 o = 1; o < world; ++o) if (owner_load[o] < owner_load[best]) best = o;
        table[b] = (uint8_t)best;
        owner_load[best] += loads[b];
    }
}

int dskgpu_mg_set_table(dskgpu_ctx* ctx, const uint8_t* table) {
    if (!ctx) return DSKGPU_E_ARG;
    if (table)          // validate before the table in use is touched: a rejected table leaves the context as it was
        for (u32 b = 0; b < SK_BUCKETS; ++b)
            if (table[b] != SK_SPLIT && table[b] >= ctx->cfg.world_size) return fail(ctx, DSKGPU_E_ARG, "repartition table names an owner outside the world");
    ctx->h_table.resize(SK_BUCKETS);
    if (table) {
        std::memcpy(ctx->h_table.data(), table, SK_BUCKETS);
    } else default_table(ctx->cfg.world_size, ctx->h_table.data());
    ctx->table_dirty = true;
    ctx->sk_prepared = false;
    return DSKGPU_OK;
}

int dskgpu_mg_sent_kmers(dskgpu_ctx* ctx, uint64_t* kmers) {
    if (!ctx || !kmers) return DSKGPU_E_ARG;
    for (u32 o = 0; o < ctx->cfg.world_size; ++o) kmers[o] = ctx->h_sk_sent[o];
    return DSKGPU_OK;
}

int dskgpu_mg_count(dskgpu_ctx* ctx, const void* d_recv, uint64_t recv_words) { return dskgpu_mg_count_sized(ctx, d_recv, recv_words, 0); }

int dskgpu_mg_slices_prepare(dskgpu_ctx* ctx, uint32_t want_slices, uint32_t* nslices, uint64_t* send_words, uint64_t* kmers_est) {
    if (!ctx || !nslices || !send_words || !kmers_est) return DSKGPU_E_ARG;
    *nslices = 0;
    if (!ctx->sk_mode) return DSKGPU_OK;                 // explicit keys: one piece
    CK(hipSetDevice(ctx->cfg.device));
    RAW_SYNC(ctx);
    return sk_slices_prepare(ctx, want_slices, nslices, send_words, kmers_est);
}
int dskgpu_mg_scatter_slice(dskgpu_ctx* ctx, void* d_send, uint64_t capacity_words, uint32_t slice) {
    if (!ctx || !d_send) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    return sk_scatter_slice(ctx, d_send, capacity_words, slice);
}
int dskgpu_mg_slices_finish(dskgpu_ctx* ctx, int* overflowed) {
    if (!ctx || !overflowed) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    return sk_slices_finish(ctx, overflowed);
}
int dskgpu_mg_count_sliced(dskgpu_ctx* ctx, const void* d_recv, uint32_t nslices, const uint64_t* slice_words, uint64_t n_kmers_est,
                           dskgpu_slice_gate gate, void* user) {
    if (!ctx || !nslices || !slice_words || !gate) return DSKGPU_E_ARG;
    if (!ctx->sk_mode) return fail(ctx, DSKGPU_E_STATE, "dskgpu_mg_count_sliced needs super-k-mer records (20 <= k <= 64, no DSKGPU_F_MG_EXPLICIT)");
    CK(hipSetDevice(ctx->cfg.device));
    ctx->stats = dskgpu_stats{};
    const u32 R = ctx->sk_sp.R;
    u64 words = 0;
    ctx->rec_slice_end.clear();
    for (u32 sl = 0; sl < nslices; ++sl) {
        if (slice_words[sl] % R) { ctx->rec_slice_end.clear(); return fail(ctx, DSKGPU_E_ARG, "a slice is not a whole number of super-k-mer records"); }
        words += slice_words[sl];
        ctx->rec_slice_end.push_back(words / R);
    }
    if (!d_recv && words) { ctx->rec_slice_end.clear(); return DSKGPU_E_ARG; }
    ctx->rec_gate = gate; ctx->rec_gate_user = user; ctx->rec_gated = 0; ctx->rec_gate_failed = false;
    int rc = ctx->W == 1 ? sk_count<1>(ctx, static_cast<const u64*>(d_recv), words, n_kmers_est, true)
                         : sk_count<2>(ctx, static_cast<const u64*>(d_recv), words, n_kmers_est, true);
    const std::string err = ctx->err;
    (void)rec_gate_all(ctx);                             // (an error path may have left early: the caller's gates are all passed when this returns)
    if (ctx->rec_gate_failed) { ctx->have_result = false; if (rc == DSKGPU_OK) rc = DSKGPU_E_STATE; else ctx->err = err; }
    ctx->rec_gate = nullptr; ctx->rec_gate_user = nullptr; ctx->rec_slice_end.clear();
    return rc;
}

int dskgpu_mg_count_sized(dskgpu_ctx* ctx, const void* d_recv, uint64_t recv_words, uint64_t n_kmers) {
    if (!ctx || (!d_recv && recv_words)) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    ctx->stats = dskgpu_stats{};
    if (ctx->sk_mode)
        return ctx->W == 1 ? sk_count<1>(ctx, static_cast<const u64*>(d_recv), recv_words, n_kmers) : sk_count<2>(ctx, static_cast<const u64*>(d_recv), recv_words, n_kmers);
    if (n_kmers && n_kmers != recv_words / (u64)ctx->W) return fail(ctx, DSKGPU_E_ARG, "dskgpu_mg_count_sized: n_kmers does not match the k-mers received");
    if (recv_words % (u64)ctx->W) return fail(ctx, DSKGPU_E_ARG, "recv_words is not a whole number of k-mer records");
    if (!d_recv) { CK(ctx->sk_keys.ensure(64)); d_recv = ctx->sk_keys.p; }      // nothing received: a null key array would read as "keys come from records"
    if (ctx->W == 1) return run_pipeline<1>(ctx, false, static_cast<const u64*>(d_recv), recv_words);
    if (ctx->W == 2) return run_pipeline<2>(ctx, false, static_cast<const K2*>(d_recv), recv_words / 2);
    return run_pipeline<4>(ctx, false, static_cast<const KN<4>*>(d_recv), recv_words / 4);
}

int dskgpu_get_stats(const dskgpu_ctx* ctx, dskgpu_stats* out) {
    if (!ctx || !out) return DSKGPU_E_ARG;
    if (!ctx->have_result) return DSKGPU_E_STATE;
    *out = ctx->stats;
    return DSKGPU_OK;
}

int dskgpu_histogram(const dskgpu_ctx* ctx, uint64_t* out, uint32_t nbins) {
    if (!ctx || !out) return DSKGPU_E_ARG;
    if (!ctx->have_result) return DSKGPU_E_STATE;
    if (nbins != ctx->cfg.histo_max + 1) return DSKGPU_E_ARG;
    std::memcpy(out, ctx->hist.data(), (size_t)nbins * 8);
    return DSKGPU_OK;
}

int dskgpu_set_row_order(dskgpu_ctx* ctx, int partition_order) {
    if (!ctx) return DSKGPU_E_ARG;
    if (partition_order) ctx->cfg.flags |= DSKGPU_F_PARTITION_ORDER; else ctx->cfg.flags &= ~DSKGPU_F_PARTITION_ORDER;
    return DSKGPU_OK;
}

uint32_t dskgpu_num_partitions(const dskgpu_ctx* ctx) { return (ctx && ctx->have_result) ? ctx->stats.n_partitions : 0; }

static void part_range(const dskgpu_ctx* ctx, uint32_t p, u64* b, u64* e) {
    if (ctx->part_mode) {
        if (!ctx->h_part_off64.empty()) { *b = ctx->h_part_off64[p]; *e = ctx->h_part_off64[p + 1]; }      // (several passes: 64-bit row numbers)
        else { *b = ctx->h_part_off[p]; *e = ctx->h_part_off[p + 1]; }
        return;
    }
    const u64 P = ctx->stats.n_partitions, n = ctx->n_rows;
    *b = n * p / P; *e = n * (p + 1) / P;
}

uint64_t dskgpu_partition_size(const dskgpu_ctx* ctx, uint32_t p) {
    if (!ctx || !ctx->have_result || p >= ctx->stats.n_partitions) return 0;
    u64 b, e; part_range(ctx, p, &b, &e); return e - b;
}

int dskgpu_partition_offsets(const dskgpu_ctx* ctx, uint64_t* offsets) {
    if (!ctx || !offsets) return DSKGPU_E_ARG;
    if (!ctx->have_result) return DSKGPU_E_STATE;
    const u32 P = ctx->stats.n_partitions;
    for (u32 p = 0; p < P; ++p) { u64 b, e; part_range(ctx, p, &b, &e); offsets[p] = b; if (p + 1 == P) offsets[P] = e; }
    if (P == 0) offsets[0] = 0;
    return DSKGPU_OK;
}

int dskgpu_partition_copy(const dskgpu_ctx* cctx, uint32_t p, uint64_t* kmers, uint32_t* abundance) {
    dskgpu_ctx* ctx = const_cast<dskgpu_ctx*>(cctx);
    if (!ctx) return DSKGPU_E_ARG;
    if (!ctx->have_result) return DSKGPU_E_STATE;
    if (p >= ctx->stats.n_partitions) return DSKGPU_E_ARG;
    u64 b, e; part_range(ctx, p, &b, &e);
    const u64 n = e - b;
    if (n == 0) return DSKGPU_OK;
    CK(hipSetDevice(ctx->cfg.device));
    if (kmers) {
        if (ctx->W == 1) CK(hipMemcpy(kmers, ctx->res_w[0] + b, n * 8, hipMemcpyDeviceToHost));
        else {
            // the device keeps one array per word; rows are words_out consecutive words on the host side
            const int wo = ctx->words_out;
            std::vector<u64> col(n);
            for (int x = 0; x < wo; ++x) {
                CK(hipMemcpy(col.data(), ctx->res_w[x] + b, n * 8, hipMemcpyDeviceToHost));
                for (u64 i = 0; i < n; ++i) kmers[(u64)wo * i + x] = col[i];
            }
        }
    }
    if (abundance) CK(hipMemcpy(abundance, ctx->res_ab + b, n * 4, hipMemcpyDeviceToHost));
    return DSKGPU_OK;
}

int dskgpu_result_device(const dskgpu_ctx* ctx, const void** d_kmers, const void** d_abundance, uint64_t* n_rows) {
    if (!ctx) return DSKGPU_E_ARG;
    if (!ctx->have_result) return DSKGPU_E_STATE;
    if (d_kmers) *d_kmers = ctx->res_w[0];
    if (d_abundance) *d_abundance = ctx->res_ab;
    if (n_rows) *n_rows = ctx->n_rows;
    return DSKGPU_OK;
}

int dskgpu_stage_times(const dskgpu_ctx* ctx, const char** names, float* ms, int cap) {
    if (!ctx) return DSKGPU_E_ARG;
    const int n = (int)ctx->st_names.size();
    for (int i = 0; i < n && i < cap; ++i) { if (names) names[i] = ctx->st_names[i]; if (ms) ms[i] = ctx->st_ms[i]; }
    return n;
}

int dskgpu_k_encode(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes, void* d_packed, void* d_invalid) {
    if (!ctx || !d_packed || !d_invalid) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    u64 nwords = 0;
    if (ctx->enc_keep) { ctx->enc_keep = false; }      // (a test hook that encodes other bytes: the kept 2-bit form of the reads is overwritten)
    int rc = run_encode(ctx, static_cast<const uint8_t*>(d_bytes), nbytes, &nwords);
    if (rc) return rc;
    CK(hipMemcpyAsync(d_packed, ctx->packed.p, nwords * 8, hipMemcpyDeviceToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_invalid, ctx->inval.p, nwords * 4, hipMemcpyDeviceToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    return DSKGPU_OK;
}

int dskgpu_k_enumerate(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes, void* d_kmers, void* d_valid) {
    if (!ctx || !d_kmers || !d_valid) return DSKGPU_E_ARG;
    CK(hipSetDevice(ctx->cfg.device));
    u64 nwords = 0;
    if (ctx->enc_keep) { ctx->enc_keep = false; }      // (a test hook that encodes other bytes: the kept 2-bit form of the reads is overwritten)
    int rc = run_encode(ctx, static_cast<const uint8_t*>(d_bytes), nbytes, &nwords);
    if (rc) return rc;
    if (nwords) {
        const unsigned grid = (unsigned)((nwords * 2 + 255) / 256);
        if (ctx->W == 1)
            hipLaunchKernelGGL(k_enumerate<1>, dim3(grid), dim3(256), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(),
                               nwords, (u64)nbytes, (int)ctx->cfg.kmer_size, static_cast<u64*>(d_kmers), static_cast<uint8_t*>(d_valid), 1);
        else if (ctx->W == 2)
            hipLaunchKernelGGL(k_enumerate<2>, dim3(grid), dim3(256), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(),
                               nwords, (u64)nbytes, (int)ctx->cfg.kmer_size, static_cast<u64*>(d_kmers), static_cast<uint8_t*>(d_valid), ctx->words_out);
        else
            hipLaunchKernelGGL(k_enumerate<4>, dim3(grid), dim3(256), 0, ctx->stream, ctx->packed.as<u64>(), ctx->inval.as<u32>(),
                               nwords, (u64)nbytes, (int)ctx->cfg.kmer_size, static_cast<u64*>(d_kmers), static_cast<uint8_t*>(d_valid), ctx->words_out);
        CKL("k_enumerate");
    }
    CK(hipStreamSynchronize(ctx->stream));
    return DSKGPU_OK;
}

int dskgpu_k_minimizers(dskgpu_ctx* ctx, const void* d_bytes, uint64_t nbytes, void* d_minim, void* d_valid) {
    if (!ctx || !d_minim || !d_valid) return DSKGPU_E_ARG;
    const int m = (int)ctx->cfg.minimizer_size, k = (int)ctx->cfg.kmer_size;
    if (m < 1 || m > 16 || m > k) return fail(ctx, DSKGPU_E_ARG, "minimizer_size must be in 1..16 and <= kmer_size");
    CK(hipSetDevice(ctx->cfg.device));
    u64 nwords = 0;
    if (ctx->enc_keep) { ctx->enc_keep = false; }      // (a test hook that encodes other bytes: the kept 2-bit form of the reads is overwritten)
    int rc = run_encode(ctx, static_cast<const uint8_t*>(d_bytes), nbytes, &nwords);
    if (rc) return rc;
    if (nbytes) {
        const u64 nthreads = (nbytes + 15) / 16;
        hipLaunchKernelGGL(k_minimizers, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, ctx->stream, ctx->packed.as<u64>(),
                           ctx->inval.as<u32>(), nwords, (u64)nbytes, k, m, static_cast<u32*>(d_minim), static_cast<uint8_t*>(d_valid));
        CKL("k_minimizers");
    }
    CK(hipStreamSynchronize(ctx->stream));
    return DSKGPU_OK;
}

}  // extern "C"
