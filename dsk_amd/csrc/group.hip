// group.hip -- node-level sharded count inside ONE process (C-ABI: dskgpu_group_* in include/dskgpu.h).
//
// The reference's one call `SortingCountAlgorithm<span>::execute()` (src/DSK.cpp:55-60) leaves ONE storage with a
// flat list of solid partitions (read back at utils/dsk2ascii.cpp:61,77).  This file is what lets the `dsk` binary keep
// that shape on N GPUs: N contexts (one per rank, each with its own device, stream and host thread), every rank cuts
// its share of the reads into super-k-mer records grouped by owner (dskgpu_mg_scatter), the records are exchanged, and
// every rank counts the k-mers it owns (dskgpu_mg_count).  The exchange is the path's one real collective:
//   transport "rccl": one communicator per rank (ncclCommInitAll), counts through host memory (same process), payload
//                     as grouped ncclSend / ncclRecv on the rank's stream -- an all-to-all-v over xGMI.  librccl is
//                     loaded with dlopen the first time a group asks for it, so single-GPU runs never pay for it.
//   transport "copy": hipMemcpyAsync from the peers' send buffers (same device, or peer access) -- used when several
//                     ranks share one device (RCCL refuses duplicate devices), i.e. the multi-rank tests on a 1-GPU box.
// Built only on the public entry points of dskgpu.h: no kernels here.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/dskgpu.h"

// the per-bank steps of dskgpu.hip (same library, not part of the C-ABI): see "Multi-bank count" there
bool dskgpu_i_per_bank(dskgpu_ctx* ctx);
uint32_t dskgpu_i_banks(dskgpu_ctx* ctx);
int dskgpu_i_banks_begin(dskgpu_ctx* ctx);
void dskgpu_i_banks_select(dskgpu_ctx* ctx, uint32_t b);
void dskgpu_i_banks_abort(dskgpu_ctx* ctx);
int dskgpu_i_banks_add(dskgpu_ctx* ctx, uint32_t b);
int dskgpu_i_banks_finish(dskgpu_ctx* ctx);

namespace {

struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
    bool load() {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { err = std::string("dlopen(librccl): ") + dlerror(); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(handle, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        CommAbort = reinterpret_cast<decltype(CommAbort)>(sym("ncclCommAbort"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return CommInitAll && CommDestroy && CommAbort && GroupStart && GroupEnd && Send && Recv && GetErrorString;
    }
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

// All ranks meet here; reusable.  wait(failed) returns the OR of what the ranks brought to THIS meeting, the same value to every
// rank: whether to go on is decided collectively -- a rank that looked at the others' result codes on its own could see a failure
// that a faster rank missed, leave, and let the rest wait forever at the next meeting (or inside an RCCL send).
class Barrier {
public:
    explicit Barrier(unsigned n) : n_(n) {}
    bool wait(bool failed) { return (wait_bits(failed ? 1u : 0u) & 1u) != 0; }
    // the same with a few flags: bit 0 = "this rank failed", the others as the meeting defines them
    unsigned wait_bits(unsigned bits) {
        std::unique_lock<std::mutex> lk(mu_);
        const unsigned gen = gen_;
        pending_ |= bits;
        if (++count_ == n_) { count_ = 0; result_ = pending_; pending_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(lk, [&] { return gen_ != gen; });
        return result_;          // (cannot change before every rank of this meeting has left: the next one needs all of them)
    }
private:
    std::mutex mu_; std::condition_variable cv_; unsigned n_, count_ = 0, gen_ = 0; unsigned pending_ = 0, result_ = 0;
};

struct DevMem {
    void* p = nullptr; size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = ((bytes + bytes / 8 + 4095) & ~size_t(4095));      // head-room: sizes wobble from call to call
        if (hipMalloc(&p, want) != hipSuccess) return false;
        cap = want;
        return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

thread_local std::string g_group_create_err;

}  // namespace

struct dskgpu_group {
    uint32_t n = 0;
    bool use_rccl = false;
    std::vector<int> dev;
    std::vector<dskgpu_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<ncclComm_t> comm;
    std::vector<DevMem> send, recv;
    std::vector<std::vector<uint64_t>> counts;      // counts[src][dst], 8-byte words
    std::vector<std::vector<uint64_t>> kmers;       // kmers[src][dst], k-mers inside those words (the receiver's sizing)
    // a step in slices (dskgpu_mg_slices_*): the exchange of slice i runs on the rank's second stream while its first writes
    // slice i + 1 and, later, partitions slice i - 1
    uint32_t nslices = 4;                            // DSKGPU_GROUP_SLICES (< 2: every step in one piece)
    std::vector<hipStream_t> cstream;                // the exchange stream of every rank
    std::vector<std::vector<hipEvent_t>> ev_sent, ev_recv;      // [rank][slice]: slice written by the sender / arrived at the receiver
    std::vector<std::vector<uint64_t>> swords;       // swords[src][slice * n + dst], 8-byte words
    std::vector<std::vector<uint64_t>> kest;         // kest[src][dst]: estimated k-mers over all slices
    uint32_t sliced_steps = 0;                       // steps of the last count that ran in slices
    std::vector<int> rc;
    std::vector<std::string> rank_err;
    std::string err;
    uint32_t histo_max = 10000;
    bool balance = true;                             // build the repartition table from sampled loads before every count
    std::vector<std::vector<uint64_t>> loads;        // loads[rank][bucket]
    std::vector<uint8_t> table;
    uint64_t exchanged_words = 0;                    // words that crossed ranks in the last count (off-diagonal of counts)
    bool have_result = false;
    // A rank that fails between the meetings of an RCCL exchange (device error, a failed ncclSend) cannot just leave: its peers have
    // already posted the matching ncclRecv and would wait in it forever.  It ABORTS every communicator of the group instead
    // (ncclCommAbort is the call RCCL offers for exactly this, from any thread): the peers' pending transfers end with an error,
    // their stream synchronisation returns, the step fails on every rank with a message -- and the group, whose communicators
    // are gone, refuses further counts (create a new one).
    std::atomic<int> comms_aborted{0};
    // comm_mu[r] is held by rank r's thread while it is inside RCCL host calls on comm[r] (GroupStart .. GroupEnd), after it has seen
    // comms_aborted == 0 under the lock.  abort_comms TRIES each lock: a communicator whose rank is inside RCCL is aborted under its
    // feet -- the use ncclCommAbort exists for -- and one whose rank is outside stays locked for the abort, so that rank cannot
    // enter RCCL with a handle that is being freed.  The handles are never written after creation; group destroy skips
    // ncclCommDestroy for aborted ones.
    std::vector<std::unique_ptr<std::mutex>> comm_mu;
};

namespace {

int group_fail(dskgpu_group* g, int code, const std::string& msg) { g->err = msg; return code; }

void abort_comms(dskgpu_group* g) {
    if (!g->use_rccl || g->comms_aborted.exchange(1)) return;
    for (size_t r = 0; r < g->comm.size(); ++r) {
        if (!g->comm[r]) continue;
        const bool got = g->comm_mu[r]->try_lock();
        (void)g_rccl.CommAbort(g->comm[r]);
        if (got) g->comm_mu[r]->unlock();
    }
}
// rank r may use its communicator: lock held on return (release with comm_leave); false = the communicators were aborted
bool comm_enter(dskgpu_group* g, uint32_t r) {
    g->comm_mu[r]->lock();
    if (g->comms_aborted.load()) { g->comm_mu[r]->unlock(); return false; }
    return true;
}
void comm_leave(dskgpu_group* g, uint32_t r) { g->comm_mu[r]->unlock(); }

// one rank of one sharded count
void rank_body(dskgpu_group* g, uint32_t r, Barrier* bar) {
    auto fail = [&](int code, const std::string& msg) { g->rc[r] = code; g->rank_err[r] = msg; };      // (a rank writes its own slot only; the slots are read after the join)
    auto failed = [&]() { return g->rc[r] != DSKGPU_OK; };
    const uint32_t n = g->n;
    dskgpu_ctx* ctx = g->ctx[r];
    if (hipSetDevice(g->dev[r]) != hipSuccess) fail(DSKGPU_E_DEVICE, "hipSetDevice");
    // ---- step 0: the minimizer repartition table -- sampled bucket loads of every rank, summed, turned into ONE table
    if (g->balance) {
        if (g->rc[r] == DSKGPU_OK) {
            const int rc = dskgpu_mg_sample(ctx, g->loads[r].data());
            if (rc != DSKGPU_OK) fail(rc, std::string("mg_sample: ") + dskgpu_last_error(ctx));
        }
        if (bar->wait(failed())) return;
        if (r == 0) {
            std::vector<uint64_t> sum(DSKGPU_MG_BUCKETS, 0);
            for (uint32_t s = 0; s < n; ++s) for (uint32_t b = 0; b < DSKGPU_MG_BUCKETS; ++b) sum[b] += g->loads[s][b];
            dskgpu_mg_make_table(sum.data(), n, g->table.data());
        }
        (void)bar->wait(false);
        const int rc = dskgpu_mg_set_table(ctx, g->table.data());
        if (rc != DSKGPU_OK) fail(rc, std::string("mg_set_table: ") + dskgpu_last_error(ctx));
    }
    // ---- per-bank modes (-solidity-kind other than sum, -histo2D): every bank goes through steps 1-3 on its own -- with the ONE
    // repartition table built above from all the reads, so a k-mer has the same owner in every bank -- and joins the rank's union
    // of per-bank rows; the rank then applies the solidity kind to the k-mers it owns (banks_finish), exactly as one GPU does.
    const bool per_bank = dskgpu_i_per_bank(ctx);
    const uint32_t nbanks = per_bank ? dskgpu_i_banks(ctx) : 1u;
    if (per_bank && g->rc[r] == DSKGPU_OK) { const int rc = dskgpu_i_banks_begin(ctx); if (rc != DSKGPU_OK) fail(rc, std::string("banks: ") + dskgpu_last_error(ctx)); }
    struct Unwind { dskgpu_ctx* c; bool on; ~Unwind() { if (on) dskgpu_i_banks_abort(c); } } unwind{ctx, per_bank};      // (whatever way the body is left)
  for (uint32_t bank = 0; bank < nbanks; ++bank) {
    if (per_bank) { dskgpu_i_banks_select(ctx, bank); for (auto& c : g->counts[r]) c = 0; }
    for (auto& c : g->kmers[r]) c = 0;
    // ---- the step in slices when every rank's input allows it (sampled send layout); else, or when a send slice overflowed,
    // the step in one piece below
    bool done_in_slices = false;
    if (g->nslices >= 2) {
        const uint32_t S = g->nslices;
        uint32_t ns = 0;
        if (g->rc[r] == DSKGPU_OK) {
            const int rc = dskgpu_mg_slices_prepare(ctx, S, &ns, g->swords[r].data(), g->kest[r].data());
            if (rc != DSKGPU_OK) fail(rc, std::string("mg_slices_prepare: ") + dskgpu_last_error(ctx));
        }
        unsigned bits = bar->wait_bits((failed() ? 1u : 0u) | (ns != S ? 2u : 0u));
        if (bits & 1u) return;
        if (!(bits & 2u)) {
            const std::vector<uint64_t>& mine = g->swords[r];
            std::vector<uint64_t> rw(S, 0), rbase(S + 1, 0), sbase(S + 1, 0);
            for (uint32_t sl = 0; sl < S; ++sl) {
                uint64_t sw = 0;
                for (uint32_t p = 0; p < n; ++p) { rw[sl] += g->swords[p][(size_t)sl * n + r]; sw += mine[(size_t)sl * n + p]; }
                rbase[sl + 1] = rbase[sl] + rw[sl]; sbase[sl + 1] = sbase[sl] + sw;
            }
            const uint64_t cap = dskgpu_mg_send_capacity_words(ctx);
            if (cap == 0) fail(DSKGPU_E_DEVICE, std::string("send capacity: ") + dskgpu_last_error(ctx));
            else if (!g->send[r].ensure(cap * 8)) fail(DSKGPU_E_NOMEM, "send buffer");
            if (!g->recv[r].ensure(std::max<uint64_t>(rbase[S], 1) * 8)) fail(DSKGPU_E_NOMEM, "receive buffer");
            if (bar->wait(failed())) return;
            uint64_t* sb = static_cast<uint64_t*>(g->send[r].p);
            uint64_t* rb = static_cast<uint64_t*>(g->recv[r].p);
            hipStream_t cs = g->cstream[r];
            for (uint32_t sl = 0; sl < S; ++sl) {
                if (g->rc[r] == DSKGPU_OK) {
                    const int rc = dskgpu_mg_scatter_slice(ctx, sb, g->send[r].cap / 8, sl);
                    if (rc != DSKGPU_OK) fail(rc, std::string("mg_scatter_slice: ") + dskgpu_last_error(ctx));
                    else if (hipEventRecord(g->ev_sent[r][sl], g->stream[r]) != hipSuccess) fail(DSKGPU_E_DEVICE, "hipEventRecord");
                }
                if (g->use_rccl) {
                    if (g->rc[r] != DSKGPU_OK || g->comms_aborted.load()) { abort_comms(g); break; }      // (the peers' posted receives end with an error instead of waiting for this rank)
                    RcclApi& a = g_rccl;
                    (void)hipStreamWaitEvent(cs, g->ev_sent[r][sl], 0);
                    if (!comm_enter(g, r)) { fail(DSKGPU_E_DEVICE, "the exchange was aborted: another rank failed"); break; }
                    ncclResult_t e = a.GroupStart();
                    uint64_t so = sbase[sl], ro = rbase[sl];
                    for (uint32_t p = 0; p < n && e == ncclSuccess; ++p) {
                        const uint64_t ws = mine[(size_t)sl * n + p], wr = g->swords[p][(size_t)sl * n + r];
                        if (ws) e = a.Send(sb + so, ws, ncclUint64, (int)p, g->comm[r], cs);
                        if (e == ncclSuccess && wr) e = a.Recv(rb + ro, wr, ncclUint64, (int)p, g->comm[r], cs);
                        so += ws; ro += wr;
                    }
                    const ncclResult_t e2 = a.GroupEnd();
                    comm_leave(g, r);
                    if (e == ncclSuccess) e = e2;
                    if (e != ncclSuccess) { fail(DSKGPU_E_DEVICE, std::string("RCCL exchange: ") + a.GetErrorString(e)); abort_comms(g); break; }
                } else {
                    if (bar->wait(failed())) return;               // every rank's event of this slice is recorded
                    uint64_t ro = rbase[sl];
                    for (uint32_t p = 0; p < n; ++p) {
                        const uint64_t wr = g->swords[p][(size_t)sl * n + r];
                        if (!wr) continue;
                        uint64_t so = 0;                              // where slice sl, owner r starts in p's send buffer
                        for (uint32_t x = 0; x < sl; ++x) for (uint32_t o = 0; o < n; ++o) so += g->swords[p][(size_t)x * n + o];
                        for (uint32_t o = 0; o < r; ++o) so += g->swords[p][(size_t)sl * n + o];
                        hipError_t e = hipStreamWaitEvent(cs, g->ev_sent[p][sl], 0);
                        if (e == hipSuccess) e = hipMemcpyAsync(rb + ro, static_cast<const uint64_t*>(g->send[p].p) + so, wr * 8, hipMemcpyDefault, cs);
                        if (e != hipSuccess) { fail(DSKGPU_E_DEVICE, std::string("exchange copy: ") + hipGetErrorString(e)); break; }
                        ro += wr;
                    }
                }
                if (hipEventRecord(g->ev_recv[r][sl], cs) != hipSuccess) fail(DSKGPU_E_DEVICE, "hipEventRecord");
            }
            // the receiver: one level-1 launch per slice, each behind the arrival of its slice
            uint64_t est = 0;
            for (uint32_t p = 0; p < n; ++p) est += g->kest[p][r];
            struct Gate { hipStream_t st; hipEvent_t* ev; } gs{g->stream[r], g->ev_recv[r].data()};
            auto gate = [](void* u, uint32_t sl) -> int { Gate* x = static_cast<Gate*>(u); return hipStreamWaitEvent(x->st, x->ev[sl], 0) == hipSuccess ? 0 : 1; };
            if (g->rc[r] == DSKGPU_OK && !g->comms_aborted.load()) {
                const int rc = dskgpu_mg_count_sliced(ctx, rbase[S] ? rb : nullptr, S, rw.data(), est, gate, &gs);
                if (rc != DSKGPU_OK) { fail(rc, std::string("mg_count_sliced: ") + dskgpu_last_error(ctx)); abort_comms(g); }      // (a peer may still wait for a slice this rank was to send)
            }
            if (g->comms_aborted.load() && g->rc[r] == DSKGPU_OK) fail(DSKGPU_E_DEVICE, "the exchange was aborted: another rank failed");
            int ovf = 0;
            if (g->rc[r] == DSKGPU_OK) {
                const int rc = dskgpu_mg_slices_finish(ctx, &ovf);
                if (rc != DSKGPU_OK) fail(rc, std::string("mg_slices_finish: ") + dskgpu_last_error(ctx));
            }
            if (hipStreamSynchronize(cs) != hipSuccess) fail(DSKGPU_E_DEVICE, "exchange: stream synchronize");
            bits = bar->wait_bits((failed() ? 1u : 0u) | (ovf ? 2u : 0u));      // (also: nobody overwrites a send buffer a peer still reads)
            if (bits & 1u) return;
            if (!(bits & 2u)) {
                done_in_slices = true;
                for (uint32_t p = 0; p < n; ++p) { g->counts[r][p] = 0; for (uint32_t sl = 0; sl < S; ++sl) g->counts[r][p] += mine[(size_t)sl * n + p]; }
                if (r == 0) ++g->sliced_steps;
            }
        }
    }
  if (!done_in_slices) {
    // ---- step 1: this rank's records, grouped by owner
    if (g->rc[r] == DSKGPU_OK) {
        for (int attempt = 0; attempt < 2; ++attempt) {
            const uint64_t cap = dskgpu_mg_send_capacity_words(ctx);
            if (cap == 0) { fail(DSKGPU_E_DEVICE, std::string("send capacity: ") + dskgpu_last_error(ctx)); break; }
            if (!g->send[r].ensure(cap * 8)) { fail(DSKGPU_E_NOMEM, "send buffer"); break; }
            const int rc = dskgpu_mg_scatter(ctx, g->send[r].p, g->send[r].cap / 8, g->counts[r].data());
            if (rc == DSKGPU_OK) { (void)dskgpu_mg_sent_kmers(ctx, g->kmers[r].data()); break; }
            // a slice of the sampled send layout overflowed and the exact layout needs more room: ask again, once
            if (attempt == 0 && rc == DSKGPU_E_ARG && std::strstr(dskgpu_last_error(ctx), "send buffer too small")) continue;
            fail(rc, std::string("mg_scatter: ") + dskgpu_last_error(ctx));
            break;
        }
    }
    if (bar->wait(failed())) return;                 // every rank's counts row is final, every send buffer complete
    // ---- step 2: the exchange (all-to-all-v)
    uint64_t recv_words = 0;
    std::vector<uint64_t> roff(n + 1, 0), soff(n + 1, 0);
    for (uint32_t s = 0; s < n; ++s) { roff[s + 1] = roff[s] + g->counts[s][r]; soff[s + 1] = soff[s] + g->counts[r][s]; }
    recv_words = roff[n];
    if (!g->recv[r].ensure(std::max<uint64_t>(recv_words, 1) * 8)) fail(DSKGPU_E_NOMEM, "receive buffer");
    if (bar->wait(failed())) return;                 // (a failed allocation must stop everybody before RCCL would hang)
    uint64_t* sb = static_cast<uint64_t*>(g->send[r].p);
    uint64_t* rb = static_cast<uint64_t*>(g->recv[r].p);
    if (g->use_rccl) {
        RcclApi& a = g_rccl;
        if (!comm_enter(g, r)) fail(DSKGPU_E_DEVICE, "the exchange was aborted: another rank failed");
        else {
            ncclResult_t e = a.GroupStart();
            for (uint32_t p = 0; p < n && e == ncclSuccess; ++p) {
                if (g->counts[r][p]) e = a.Send(sb + soff[p], g->counts[r][p], ncclUint64, (int)p, g->comm[r], g->stream[r]);
                if (e == ncclSuccess && g->counts[p][r]) e = a.Recv(rb + roff[p], g->counts[p][r], ncclUint64, (int)p, g->comm[r], g->stream[r]);
            }
            const ncclResult_t e2 = a.GroupEnd();
            comm_leave(g, r);
            if (e == ncclSuccess) e = e2;
            if (e != ncclSuccess) { fail(DSKGPU_E_DEVICE, std::string("RCCL exchange: ") + a.GetErrorString(e)); abort_comms(g); }
        }
    } else {
        for (uint32_t s = 0; s < n; ++s) {
            const uint64_t w = g->counts[s][r];
            if (!w) continue;
            uint64_t off = 0;
            for (uint32_t d = 0; d < r; ++d) off += g->counts[s][d];
            const hipError_t e = hipMemcpyAsync(rb + roff[s], static_cast<const uint64_t*>(g->send[s].p) + off, w * 8, hipMemcpyDefault, g->stream[r]);
            if (e != hipSuccess) { fail(DSKGPU_E_DEVICE, std::string("exchange copy: ") + hipGetErrorString(e)); break; }
        }
    }
    if (g->rc[r] == DSKGPU_OK && hipStreamSynchronize(g->stream[r]) != hipSuccess) fail(DSKGPU_E_DEVICE, "exchange: stream synchronize");
    if (bar->wait(failed())) return;                 // nobody overwrites a send buffer a peer still reads
    // ---- step 3: count what this rank owns
    uint64_t recv_kmers = 0;
    for (uint32_t s = 0; s < n; ++s) recv_kmers += g->kmers[s][r];
    int rc = dskgpu_mg_count_sized(ctx, recv_words ? g->recv[r].p : nullptr, recv_words, recv_kmers);
    if (rc != DSKGPU_OK) fail(rc, std::string("mg_count: ") + dskgpu_last_error(ctx));
  }
    if (per_bank) {
        int rc;
        if (g->rc[r] == DSKGPU_OK && (rc = dskgpu_i_banks_add(ctx, bank)) != DSKGPU_OK) fail(rc, std::string("banks: ") + dskgpu_last_error(ctx));
        if (bar->wait(failed())) return;             // (the next bank re-uses the send buffers and the counts matrix)
    }
  }
    if (per_bank) {
        unwind.on = false;
        const int rc = dskgpu_i_banks_finish(ctx);
        if (rc != DSKGPU_OK) fail(rc, std::string("banks: ") + dskgpu_last_error(ctx));
    }
}

}  // namespace

extern "C" {

const char* dskgpu_group_last_error(const dskgpu_group* g) { return g ? g->err.c_str() : g_group_create_err.c_str(); }

int dskgpu_group_create(const dskgpu_config* cfg, const int32_t* devices, uint32_t n_ranks, dskgpu_group** out) {
    if (!cfg || !out || !devices || n_ranks == 0) { g_group_create_err = "null argument"; return DSKGPU_E_ARG; }
    *out = nullptr;
    if ((n_ranks & (n_ranks - 1)) != 0 || n_ranks > 64) { g_group_create_err = "the number of ranks must be a power of two <= 64"; return DSKGPU_E_ARG; }
    dskgpu_group* g = new dskgpu_group();
    g->n = n_ranks;
    g->dev.assign(devices, devices + n_ranks);
    g->histo_max = cfg->histo_max ? cfg->histo_max : 10000;
    bool distinct = true;
    for (uint32_t a = 0; a < n_ranks; ++a) for (uint32_t b = a + 1; b < n_ranks; ++b) distinct = distinct && devices[a] != devices[b];
    const char* want = getenv("DSKGPU_GROUP_TRANSPORT");           // "rccl" | "copy"; default: rccl when every rank has its own device
    g->use_rccl = want ? std::strcmp(want, "rccl") == 0 : (distinct && n_ranks > 1);
    auto bail = [&](int code, const std::string& msg) { g_group_create_err = msg; dskgpu_group_destroy(g); return code; };
    if (g->use_rccl && !distinct) return bail(DSKGPU_E_ARG, "transport rccl needs one device per rank");
    g->ctx.assign(n_ranks, nullptr); g->stream.assign(n_ranks, nullptr); g->comm.assign(n_ranks, nullptr);
    for (uint32_t r = 0; r < n_ranks; ++r) g->comm_mu.emplace_back(new std::mutex());
    g->send.resize(n_ranks); g->recv.resize(n_ranks);
    g->counts.assign(n_ranks, std::vector<uint64_t>(n_ranks, 0));
    g->kmers.assign(n_ranks, std::vector<uint64_t>(n_ranks, 0));
    g->rc.assign(n_ranks, DSKGPU_OK); g->rank_err.assign(n_ranks, "");
    if (const char* e = getenv("DSKGPU_GROUP_SLICES")) g->nslices = (uint32_t)std::min<long>(64, std::max<long>(0, std::atol(e)));
    g->cstream.assign(n_ranks, nullptr);
    g->ev_sent.assign(n_ranks, std::vector<hipEvent_t>(g->nslices, nullptr)); g->ev_recv = g->ev_sent;
    g->swords.assign(n_ranks, std::vector<uint64_t>((size_t)std::max<uint32_t>(g->nslices, 1) * n_ranks, 0));
    g->kest.assign(n_ranks, std::vector<uint64_t>(n_ranks, 0));
    g->loads.assign(n_ranks, std::vector<uint64_t>(DSKGPU_MG_BUCKETS, 0)); g->table.assign(DSKGPU_MG_BUCKETS, 0);
    if (const char* e = getenv("DSKGPU_GROUP_BALANCE")) g->balance = std::strcmp(e, "0") != 0;     // "0": keep the default table (tests)
    for (uint32_t r = 0; r < n_ranks; ++r) {
        dskgpu_config c = *cfg;
        c.world_size = n_ranks; c.rank = r; c.device = devices[r];
        const int rc = dskgpu_create(&c, &g->ctx[r]);
        if (rc != DSKGPU_OK) return bail(rc, std::string("rank ") + std::to_string(r) + ": " + dskgpu_last_error(nullptr));
        if (hipSetDevice(devices[r]) != hipSuccess || hipStreamCreateWithFlags(&g->stream[r], hipStreamNonBlocking) != hipSuccess)
            return bail(DSKGPU_E_DEVICE, "stream of rank " + std::to_string(r));
        if (dskgpu_set_stream(g->ctx[r], g->stream[r]) != DSKGPU_OK) return bail(DSKGPU_E_DEVICE, "set_stream");
        if (g->nslices >= 2) {
            if (hipStreamCreateWithFlags(&g->cstream[r], hipStreamNonBlocking) != hipSuccess) return bail(DSKGPU_E_DEVICE, "exchange stream of rank " + std::to_string(r));
            for (uint32_t sl = 0; sl < g->nslices; ++sl)
                if (hipEventCreateWithFlags(&g->ev_sent[r][sl], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&g->ev_recv[r][sl], hipEventDisableTiming) != hipSuccess) return bail(DSKGPU_E_DEVICE, "events of rank " + std::to_string(r));
        }
    }
    if (!g->use_rccl) {                                  // the copy transport reads the peers' send buffers directly
        for (uint32_t a = 0; a < n_ranks; ++a)
            for (uint32_t b = 0; b < n_ranks; ++b) {
                if (devices[a] == devices[b]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[a], devices[b]) == hipSuccess && can) {
                    (void)hipSetDevice(devices[a]);
                    const hipError_t e = hipDeviceEnablePeerAccess(devices[b], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return bail(DSKGPU_E_DEVICE, "hipDeviceEnablePeerAccess");
                    (void)hipGetLastError();
                }
            }
    } else {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        if (!g_rccl.load()) return bail(DSKGPU_E_DEVICE, g_rccl.err);
        const ncclResult_t e = g_rccl.CommInitAll(g->comm.data(), (int)n_ranks, g->dev.data());
        if (e != ncclSuccess) { for (auto& c : g->comm) c = nullptr; return bail(DSKGPU_E_DEVICE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(e)); }
    }
    *out = g;
    return DSKGPU_OK;
}

void dskgpu_group_destroy(dskgpu_group* g) {
    if (!g) return;
    for (uint32_t r = 0; r < g->ctx.size(); ++r) {
        if (r < g->dev.size()) (void)hipSetDevice(g->dev[r]);
        if (r < g->stream.size() && g->stream[r]) (void)hipStreamSynchronize(g->stream[r]);
        if (r < g->comm.size() && g->comm[r] && !g->comms_aborted.load()) (void)g_rccl.CommDestroy(g->comm[r]);      // (an aborted communicator is already freed)
        if (g->ctx[r]) { (void)dskgpu_set_stream(g->ctx[r], nullptr); dskgpu_destroy(g->ctx[r]); }
        if (r < g->send.size()) { g->send[r].release(); g->recv[r].release(); }
        if (r < g->stream.size() && g->stream[r]) (void)hipStreamDestroy(g->stream[r]);
        if (r < g->cstream.size() && g->cstream[r]) { (void)hipStreamSynchronize(g->cstream[r]); (void)hipStreamDestroy(g->cstream[r]); }
        if (r < g->ev_sent.size()) for (hipEvent_t e : g->ev_sent[r]) if (e) (void)hipEventDestroy(e);
        if (r < g->ev_recv.size()) for (hipEvent_t e : g->ev_recv[r]) if (e) (void)hipEventDestroy(e);
    }
    delete g;
}

uint32_t dskgpu_group_size(const dskgpu_group* g) { return g ? g->n : 0; }
dskgpu_ctx* dskgpu_group_ctx(dskgpu_group* g, uint32_t rank) { return (g && rank < g->n) ? g->ctx[rank] : nullptr; }
const char* dskgpu_group_transport(const dskgpu_group* g) { return !g ? "" : g->use_rccl ? "rccl" : "copy"; }
uint64_t dskgpu_group_exchanged_words(const dskgpu_group* g) { return g ? g->exchanged_words : 0; }
uint32_t dskgpu_group_sliced_steps(const dskgpu_group* g) { return g ? g->sliced_steps : 0; }

int dskgpu_group_count(dskgpu_group* g) {
    if (!g) return DSKGPU_E_ARG;
    if (g->comms_aborted.load()) return group_fail(g, DSKGPU_E_STATE, "the group's RCCL communicators were aborted by a failed step: create a new group");
    g->have_result = false; g->sliced_steps = 0;
    std::fill(g->rc.begin(), g->rc.end(), DSKGPU_OK);
    for (auto& row : g->counts) std::fill(row.begin(), row.end(), 0);
    Barrier bar(g->n);
    std::vector<std::thread> th;
    for (uint32_t r = 1; r < g->n; ++r) th.emplace_back(rank_body, g, r, &bar);
    int caller_dev = -1;
    (void)hipGetDevice(&caller_dev);                 // rank 0 runs on the caller's thread: its current device is put back
    rank_body(g, 0, &bar);
    if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
    for (auto& t : th) t.join();
    for (uint32_t r = 0; r < g->n; ++r)
        if (g->rc[r] != DSKGPU_OK) return group_fail(g, g->rc[r], "rank " + std::to_string(r) + ": " + g->rank_err[r]);
    g->exchanged_words = 0;
    for (uint32_t s = 0; s < g->n; ++s) for (uint32_t d = 0; d < g->n; ++d) if (s != d) g->exchanged_words += g->counts[s][d];
    g->have_result = true;
    return DSKGPU_OK;
}

int dskgpu_group_histogram(const dskgpu_group* g, uint64_t* out, uint32_t nbins) {
    if (!g || !out) return DSKGPU_E_ARG;
    if (!g->have_result) return DSKGPU_E_STATE;
    if (nbins != g->histo_max + 1) return DSKGPU_E_ARG;
    std::vector<uint64_t> one(nbins);
    std::memset(out, 0, (size_t)nbins * 8);
    for (uint32_t r = 0; r < g->n; ++r) {
        const int rc = dskgpu_histogram(g->ctx[r], one.data(), nbins);
        if (rc != DSKGPU_OK) return rc;
        for (uint32_t i = 0; i < nbins; ++i) out[i] += one[i];
    }
    return DSKGPU_OK;
}

int dskgpu_group_histogram2d(const dskgpu_group* g, uint64_t* out, uint32_t nrows) {
    if (!g || !out) return DSKGPU_E_ARG;
    if (!g->have_result) return DSKGPU_E_STATE;
    if (nrows != g->histo_max + 1) return DSKGPU_E_ARG;
    std::vector<uint64_t> one((size_t)nrows * 11);
    std::memset(out, 0, (size_t)nrows * 11 * 8);
    for (uint32_t r = 0; r < g->n; ++r) {
        const int rc = dskgpu_histogram2d(g->ctx[r], one.data(), nrows);
        if (rc != DSKGPU_OK) return rc;
        for (size_t i = 0; i < one.size(); ++i) out[i] += one[i];
    }
    return DSKGPU_OK;
}

int dskgpu_group_get_stats(const dskgpu_group* g, dskgpu_stats* out) {
    if (!g || !out) return DSKGPU_E_ARG;
    if (!g->have_result) return DSKGPU_E_STATE;
    dskgpu_stats t{};
    for (uint32_t r = 0; r < g->n; ++r) {
        dskgpu_stats s{};
        const int rc = dskgpu_get_stats(g->ctx[r], &s);
        if (rc != DSKGPU_OK) return rc;
        t.n_bytes += s.n_bytes; t.n_kmers += s.n_kmers; t.n_distinct += s.n_distinct; t.n_solid += s.n_solid;
        t.n_partitions += s.n_partitions; t.n_retries += s.n_retries; t.sort_fallback += s.sort_fallback; t.n_ext_regions += s.n_ext_regions; t.n_heavy += s.n_heavy;
        t.n_levels = std::max(t.n_levels, s.n_levels); t.n_final_bins += s.n_final_bins; t.n_passes = std::max(t.n_passes, s.n_passes); t.n_read_sweeps = std::max(t.n_read_sweeps, s.n_read_sweeps);
    }
    *out = t;
    return DSKGPU_OK;
}

// Global partition ids: rank r's local partition p is partition p * n_ranks + r of the job, so `dsk/solid/<P>` stays one
// flat list (utils/dsk2ascii.cpp:61,77) whatever the number of GPUs; rows ascend by k-mer value inside a partition.
uint32_t dskgpu_group_num_partitions(const dskgpu_group* g) {
    if (!g || !g->have_result) return 0;
    return dskgpu_num_partitions(g->ctx[0]) * g->n;          // every rank uses the same nb_partitions
}
uint64_t dskgpu_group_partition_size(const dskgpu_group* g, uint32_t P) {
    if (!g || !g->have_result) return 0;
    return dskgpu_partition_size(g->ctx[P % g->n], P / g->n);
}
int dskgpu_group_partition_copy(const dskgpu_group* g, uint32_t P, uint64_t* kmers, uint32_t* abundance) {
    if (!g) return DSKGPU_E_ARG;
    if (!g->have_result) return DSKGPU_E_STATE;
    return dskgpu_partition_copy(g->ctx[P % g->n], P / g->n, kmers, abundance);
}

}  // extern "C"
