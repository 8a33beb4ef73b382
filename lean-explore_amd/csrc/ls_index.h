// ls_index.h — private: the index handle behind include/leansearch.h's opaque `ls_index`, shared by
// ls_api.hip (single-device orchestration; ls_callers.hip, ls_batched.hip, ls_debug.hip are its other parts) and
// ls_shard.hip (the row-sharded group handle).
#pragma once
#include "ls_common.h"

#include <chrono>
#include <condition_variable>
#include <cstddef>
#include <deque>
#include <mutex>
#include <vector>

struct ls_shard_group;  // ls_shard.hip
struct ls_req;          // ls_callers.hip: one queued synchronous host search

#define LS_NSETS 2
// Unchecked batched calls a handle carries before it checks them itself (a check drains the pipeline:
// ~2 batch times). Every slot owns a flag slice (capped at 16 MB in total: fewer slots for huge
// batches) and a copy of the raw queries; the copies are allocated in blocks of LS_BC_QKEEP_BLOCK slots
// as the backlog actually grows, so a caller that checks regularly never pays for the rest.
#define LS_BC_SLOTS 1024
#define LS_BC_QKEEP_BLOCK 64
#ifndef LS_BC_LANES
// Scratch sets that consecutive LS_FLAG_PIPELINE batches rotate over. Four, so that nothing a batch
// has to wait for is younger than two batches: prep(i+4) rewrites the prepared queries pass(i) read,
// pass(i+4) the queues select(i) read. (With two, prep(i+2) had to wait for pass(i) itself, and the
// ~20 us a cross-stream dependency takes to resolve landed between two passes.)
#define LS_BC_LANES 4
#endif
#define LS_BC_SETS (1 + LS_BC_LANES)  // batched scratch sets: 0 = caller's stream, then the lanes
#define LS_PROF_MAX 4096

// Buffers of ONE synchronous host search (ls_search) in flight. Two slots: while the call that owns slot
// A still polls for its results, the next caller may already queue its launch with slot B (round 5:
// two callers used to ping-pong as two single launches with the GPU idle in between). Slot c also owns
// scan scratch generation c: a same-launch selection's retry reads its generation's score vector and
// granules after the host has seen its answer, so nothing else may touch that generation meanwhile.
// Lock order: slot mutex(es) first, then ls_index::mu.
struct ls_host_slot {
    std::mutex mu;  // held from the enqueue until the results have been handed back
    hipStream_t stream = nullptr;  // overlapped calls run on their slot's own stream: the next call's scan starts
                                   // while this call's selection workgroup and tail are still running
    float* h_q = nullptr;      size_t h_q_cap = 0;      // pinned query copy (the kernels may read it directly)
    float* d_qraw = nullptr;   size_t qraw_cap = 0;     // device query copy (floats)
    float* d_out_s = nullptr;  int64_t* d_out_i = nullptr;  size_t out_cap = 0;  // nq*k (large results)
    float* h_out_s = nullptr;  int64_t* h_out_i = nullptr;  size_t h_out_cap = 0;  // pinned
    u32* h_done = nullptr;     // pinned [LS_QUERIES_PER_LAUNCH_MAX]: completion words
    u32 done_seq = 0;
    ls_out_gran* h_out_g = nullptr;  size_t h_out_g_cap = 0;  // pinned: result granules of a spinning call
    std::vector<ls_fin_batch> retry_groups;  // the call's same-launch jobs, one batch per launch (LS_DONE_RETRY)
};
#define LS_HOST_SLOTS 2

struct ls_index {
    int32_t device = 0;
    int32_t n_cu = 256;
    int64_t n = 0;
    int32_t dtype = 0;
    int64_t base = 0;
    ls_geom g{};
    void* d_corpus = nullptr;
    int64_t cap_rows = 0;  // rows d_corpus has room for (+ LS_CORPUS_PAD_ROWS); >= n
    std::mutex mu;
    hipStream_t own_stream = nullptr;
    // synchronous host searches that arrive together are served as one batch (ls_search)
    std::mutex q_mu;
    std::condition_variable q_cv;
    std::deque<ls_req*> req_q;
    bool leader_active = false;
    std::atomic<uint64_t> q_epoch{0};  // bumped whenever the queue's state changes (arrivals too: the gathering leader counts them)
    std::atomic<int64_t> q_len{0};     // req_q.size(), readable without q_mu (the gathering leader polls it)
    std::atomic<uint64_t> lead_epoch{0};  // bumped when the leadership or a host slot comes free: what a QUEUED waiter polls
    int32_t calls_in_flight = 0;       // batches queued whose results have not been handed back yet (under q_mu)
    int64_t requests_in_flight = 0;    // ... and the requests in them
    int64_t peak_callers = 0;          // decaying maximum of (requests in flight + queued): the callers around lately
    std::chrono::steady_clock::time_point last_arrival{};  // (under q_mu) ...
    double arrival_gap_us = 1e4;       // ... running mean of the time between two ls_search arrivals
    uint64_t arrivals_seen = 0;
    uint32_t peak_decay = 0;
    int32_t opt_combine = 1;
    uint64_t n_combined_batches = 0, n_combined_requests = 0;
    std::atomic<int32_t> spinners{0};  // waiters polling right now (ls_spin_cap: as many as the process has CPUs for)
    uint64_t n_waiter_parks = 0;       // (under q_mu) debug counter 33
    uint64_t n_lead_wait_ns = 0, n_lead_call_ns = 0, n_lead_relock_ns = 0, n_lead_begin_ns = 0, n_lead_finish_ns = 0;  // (under q_mu) debug counters 28-30
    // non-null: this handle is a row-sharded GROUP (ls_create_sharded): `n`, `dtype`, `g` and
    // `device` (the primary shard's) describe the whole index, every other member below is unused
    // and the per-device sub-handles live in the group (ls_shard.hip)
    ls_shard_group* group = nullptr;

    // scratch (grown on demand, reused by every search on this handle)
    // per-query scan scratch, LS_NSETS generations: launch i's piggy-backed finalize of group i-1
    // reads one generation while its scan of group i fills the other
    struct scratch_set {
        float* d_S = nullptr;        // s_vecs score vectors, s_stride floats apart
        u64* d_cand = nullptr;       // max_blocks * LS_KP_MAX
        u64* d_bound = nullptr;      // max_blocks
        void* d_gran = nullptr;      // same-launch selection: LS_QUERIES_PER_LAUNCH_MAX * LS_GRAN_MAX tagged
                                     // 16-byte granules (ls_fin_params::gran), allocated at first use
        float* d_qpad = nullptr;     // padded query group of a ragged VALU group (8 x d floats)
        size_t qpad_cap = 0;
        hipStream_t last_stream = nullptr;  // stream of the last launch that used this generation: a launch
                                            // on another stream first waits (on the host) for that one
    } sets[LS_NSETS];
    uint64_t set_rr = 0;
    int32_t last_set = 0;
    // batched (MFMA) path scratch, allocated on first use. Set 0 serves plain calls on the
    // caller's stream: a handle that only ever sees one stream pays nothing for sharing it; the
    // first call on a second stream synchronises the previous one and switches the set to
    // multi-stream mode, where `done` is recorded behind the last kernel of every call and
    // waited for by the next call's stream (an event record costs a few us of GPU time per
    // batch: a barrier packet with a release). Sets 1 .. LS_BC_LANES serve LS_FLAG_PIPELINE calls:
    // consecutive batches rotate over them (the three-stream chain, batched_search_on_stream).
    struct bc_set {
        void* d_qh = nullptr;      size_t qh_cap = 0;       // bytes: fp16 queries [nq_pad, d_pad]
        u64* d_queues = nullptr;   size_t queues_cap = 0;   // private candidate queues
        u32* d_counts = nullptr;   size_t counts_cap = 0;
        float* d_tau = nullptr;    size_t tau_cap = 0;
        u32* d_sample_top = nullptr; size_t sample_top_cap = 0;  // best sample scores per lane
        hipEvent_t done = nullptr;        // set 0, multi-stream mode
        hipStream_t last_stream = nullptr;  // where the set's results become final (the select's stream)
        bool used = false;
        bool multi_stream = false;
        // sets 1 .. LS_BC_LANES (LS_FLAG_PIPELINE): the three-stream chain, see batched_search_on_stream
        bool chain = false;
        bool sel_recorded = false;        // ev_sel carries a record
        hipEvent_t ev_prep = nullptr;     // the set's prep kernel is done: the caller may reuse its query buffer
        hipEvent_t ev_pass = nullptr;     // the set's pass is done (attached to its dispatch): the select may start
        hipEvent_t ev_sel = nullptr;      // the set's select is done (recorded on the select stream)
    } bc_sets[LS_BC_SETS];
    // one batch between "sample pass + tau queued" and "pass + select queued" (batched_search_on_stream)
    struct bc_stage {
        bool active = false;
        int set_id = 0, lane = 0;
        bool f32 = false, top2 = false;
        int64_t nq = 0, nq_pad = 0, rps = 0;
        int32_t k = 0;
        int nsplits = 0, sample_stride = 0, jrank = 0, keys_need = 0;
        u32* d_flags = nullptr;
        float* d_out_s = nullptr;
        int64_t* d_out_i = nullptr;
    };
    std::deque<bc_stage> held;  // pipelined batches whose pass is held back (at most two: see batched_search_on_stream)
    // LS_FLAG_PIPELINE batches: prep kernels, filter stages (MFMA passes, back to back) and selects each
    // have their own stream; the scratch sets 1 .. LS_BC_LANES alternate between consecutive batches
    hipStream_t chain_main[2] = {nullptr, nullptr};  // the chain's two lanes
    hipStream_t chain_sel = nullptr;                 // ... and its select stream
    hipEvent_t chain_in = nullptr;        // recorded on the caller's stream, waited for by the prep stream
    int32_t opt_fused = 1;                // pipelined fp16 batches of rows <= 768 bytes: a later batch's sample phase rides on the pass launch
    int32_t opt_wave_select = 1;          // one-wave select kernel (<= 48 VGPRs) where the shape allows
    uint64_t bc_lane_rr = 0;
    int32_t bc_last_set = 0;  // the set of the most recent batched call (ls_export_flags)
    u32* d_overflow = nullptr; size_t overflow_cap = 0;
    u32* h_overflow = nullptr; size_t h_overflow_cap = 0;  // pinned
    // async batched calls not yet checked: each keeps its own flag slice of d_overflow AND its own
    // copy of the raw queries (d_qkeep), so that several batches can be in flight before one
    // ls_check repairs whatever was flagged, whatever the caller did to its query buffer meanwhile
    struct batched_call {
        int64_t nq = 0;
        int32_t k = 0;
        uint32_t flags = 0;
        float* d_out_s = nullptr;
        int64_t* d_out_i = nullptr;
        hipStream_t stream = nullptr;
        int32_t slot = 0;
    };
    std::vector<batched_call> bc_pending;
    int64_t bc_slot_stride = 0;  // u32 per flag slot
    float* d_qkeep_blk[LS_BC_SLOTS / LS_BC_QKEEP_BLOCK] = {};  // each LS_BC_QKEEP_BLOCK x bc_qkeep_stride floats
    int64_t bc_qkeep_stride = 0;
    int bc_slots() const {  // slots in use for the current slot stride
        const int64_t by_mem = (4ll << 20) / (bc_slot_stride > 0 ? bc_slot_stride : 1);
        return (int)(by_mem < 16 ? 16 : (by_mem > LS_BC_SLOTS ? LS_BC_SLOTS : by_mem));
    }
    u32* d_last_flags = nullptr;  // flag slice of the most recent batched call (ls_export_flags)
    int64_t last_flags_n = 0;
    uint64_t n_batched_fallback = 0;  // queries repaired by the scan path (host counter)
    int32_t n_batched_launches = 0;   // kernel launches of the most recent batched call (counted)
    int32_t last_path = 0;            // most recent search: 1 scan path, 2 fp16 MFMA path, 3 fp32 MFMA path
    uint64_t n_chunked_calls = 0;     // batched calls cut into sub-batches (candidate-queue capacity)
    uint64_t n_launches_total = 0;    // kernel launches queued by searches on this handle
    int32_t opt_gemm = 1;             // allow the batched MFMA path
    int32_t opt_spec_tau = 1;         // speculative (verified) sample threshold

    int n_pending = 0;                 // queries whose finalize has not been launched yet
    ls_fin_batch pending{};
    hipStream_t pending_stream = nullptr;
    long long s_stride = 0;            // floats between the score vectors of one generation
    int32_t s_vecs = 0;                // score vectors per generation: 8 (one VALU scan group) until a launch that keeps
                                       // its score vectors serves more (ls_mq, LS_FLAG_ASYNC-only calls): grow_score_vectors
    int32_t opt_multi_query = 1;       // several queries per corpus pass (groups of 8 / 4)
    int32_t opt_mq = 1;                // fp32 index: 2..16 queries per pass on the f32 matrix cores (ls_mq.hip)
    int32_t opt_mq32 = 1;              // ... up to 32 queries per pass (two B blocks; debug option 22)
    uint64_t n_mq_launches = 0;
    int32_t opt_scan_skip_scores = 1;  // ... single-query launches of pipelined / synchronous device calls too
    int32_t opt_mq_skip_scores = 1;    // ... whose selection jobs ride along write no score vectors (debug option 19)
    uint64_t n_mq_reserved = 0;        // queries of such launches served again on the scan kernel (counter 25)
    uint64_t n_mq_skipped_repairs = 0; // ... whose output rows a later pipelined call had been given meanwhile (counter 26)
    // (read under q_mu by ls_search, written under the handle's mutex by ls_debug_option: atomics)
    std::atomic<int32_t> opt_early_cap{8};   // ls_search: callers up to which a second batch goes early (debug option 21)
    std::atomic<int32_t> opt_full_early{1};  // ls_search: a queue that fills a pass goes at once behind the call in flight (debug option 23)
    std::atomic<int32_t> opt_gather{1};      // ls_search: concurrent callers are gathered into one pass (debug option 20)
    double call_us_est = 0.0;          // running estimate of one combined call, begin to finish (under q_mu)
    bool reserving = false;            // (that second serve is being queued: its selection takes its own launch)
    // ... and so do pipelined / synchronous DEVICE-output calls: the launch keeps its raw queries (slot of
    // d_mq_keep), an unproven query raises its word in d_mq_flags, mq_repair (ls_check, the end of a
    // synchronous call, ring full) serves it again in place. LS_FLAG_ASYNC alone keeps the score vectors:
    // its results are promised in stream order.
    struct mq_pending_call {
        int slot, nq;
        int32_t k;
        uint32_t flags;
        float* d_out_s;
        int64_t* d_out_i;
        hipStream_t stream;
    };
    std::vector<mq_pending_call> mq_pend;
    float* d_mq_keep = nullptr;        // [LS_MQ_KEEP_SLOTS][16][mq_keep_d]
    int32_t mq_keep_d = 0;
    u32* d_mq_flags = nullptr;         // [LS_MQ_KEEP_SLOTS][16]
    u32* h_mq_flags = nullptr;         // pinned mirror; its LAST word is raised by any flagged job (ls_fin_params::repair_any)
    bool dev_call_repairable = false;  // set by ls_search_device around a call whose results may be repaired later
    int32_t opt_same_launch = 1;       // synchronous host calls: the selection rides on its own query's scan launch
    uint64_t n_forced_checks = 0;           // checks of pending batched calls the library ran on its own
    uint64_t n_same_launch_retries = 0;     // host calls that had to launch the stand-alone finalize
    unsigned long long n_spin_timeouts = 0; // host calls whose 2 ms of polling ran out (they slept in hipStreamSynchronize; counter 27)
    u32 gran_tag = 0;                  // tag of the last same-launch selection (ls_fin_params::tag; never 0)
    int32_t max_blocks = 0;
    u32* d_counters = nullptr;                        // [0] finalize slow-path count
    ls_host_slot hs[LS_HOST_SLOTS];
    std::atomic<unsigned> hs_rr{0};
    int32_t opt_overlap_calls = 1;     // synchronous host calls may overlap two deep (debug option 17)
    uint64_t n_overlapped_calls = 0;   // host calls that were queued while another one was still in flight
    int32_t force_gen = -1;            // >= 0: the scan scratch generation the call being queued must use
    std::vector<ls_fin_batch>* cur_retry = nullptr;  // where the call being queued keeps its same-launch jobs
    u32 cur_done_seq = 0;              // ... and the sequence number its completion words / granules carry
    u32* done_base = nullptr;  // set by ls_search around its scan-path call, else null
    ls_out_gran* gran_out_base = nullptr;                      // set together with done_base

    // options / instrumentation
    int32_t opt_kprime = 0;  // 0 = automatic
    int32_t opt_blocks = 0;  // 0 = automatic (scan workgroups per launch)
    int32_t opt_force_slow = 0;
    int32_t opt_overlap = 1;    // finalize of group i-1 rides on the scan launch of group i
    int32_t opt_alternate = 0;  // alternate sweep direction between consecutive scans
    uint64_t sweep_count = 0;
    // profiling: hipEvent pairs around EVERY scan launch (and the finalize after it), recorded
    // on the stream the kernels run on, up to LS_PROF_MAX launches; read by ls_last_kernel_ms
    bool profiling = false;
    std::vector<hipEvent_t> prof_ev;  // 2 events per launch: begin, end
    size_t prof_n = 0;
};


// Every synchronous host call in flight has handed its results back (both host slots), then the handle's
// mutex: what every entry point that touches the scan scratch, the corpus or the base takes.
struct ls_quiesce {
    std::unique_lock<std::mutex> a, b, m;
    explicit ls_quiesce(ls_index* ix) {
        std::lock(ix->hs[0].mu, ix->hs[1].mu);
        a = std::unique_lock<std::mutex>(ix->hs[0].mu, std::adopt_lock);
        b = std::unique_lock<std::mutex>(ix->hs[1].mu, std::adopt_lock);
        m = std::unique_lock<std::mutex>(ix->mu);
    }
};

template <typename T>
static inline int ls_grow(T** p, size_t* cap, size_t need) {
    if (need <= *cap) return LS_OK;
    if (*p) LS_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 2;
    LS_HIP(hipMalloc((void**)p, want * sizeof(T)));
    *cap = want;
    return LS_OK;
}
template <typename T>
static inline int ls_grow_pinned(T** p, size_t* cap, size_t need, unsigned flags = hipHostMallocDefault) {
    if (need <= *cap) return LS_OK;
    if (*p) LS_HIP(hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    size_t want = need + need / 2;
    LS_HIP(hipHostMalloc((void**)p, want * sizeof(T), flags));
    *cap = want;
    return LS_OK;
}

// ---- internals of ls_api.hip that the group handle drives. The caller holds the mutex of the
// handle it passes and has made that handle's device current. ---------------------------------
int ls_i_check_device(int32_t device);
int ls_i_check_search_args(const ls_index* ix, const void* q, int64_t nq, int32_t k, uint32_t flags,
                           const void* os, const void* oi);
bool ls_i_batched_eligible(const ls_index* ix, int64_t nq, int32_t k);
int ls_i_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k, uint32_t flags,
                          float* d_out_s, int64_t* d_out_i, hipStream_t s, bool host_api);
int ls_i_flush_pending(ls_index* ix);
int ls_i_flush_deferred(ls_index* ix);
int ls_i_batched_repair(ls_index* ix);
int ls_i_export_flags(ls_index* ix, void* d_dst, int64_t nq, hipStream_t s);
int ls_i_grow_score_vectors(ls_index* ix, int need);  // score vectors per scratch generation, grown on demand
// (ls_batched.hip)
int64_t ls_i_bc_chunk(const ls_index* ix, int64_t nq, int32_t k);  // queries one batched call takes at once (0: none)
int ls_i_batched_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k, uint32_t flags,
                                  float* d_out_s, int64_t* d_out_i, hipStream_t s);
// (what the synchronous host path, ls_callers.hip, asks about a call's shape)
int ls_i_scan_path_max_nq(const ls_index* ix, int32_t k);            // queries ONE scan-path launch carries at this k
int64_t ls_i_scan_group_count(const ls_index* ix, int64_t nq, int32_t k);  // launches the scan path cuts nq queries into

// ---- the group handle (ls_shard.hip); every function takes the GROUP's ls_index -----------------
int ls_group_search(ls_index* ix, const float* q, bool q_on_host, int64_t nq, int32_t k,
                    uint32_t flags, float* out_s, int64_t* out_i, hipStream_t s);
int ls_replica_search(ls_index* ix, const float* q, int64_t nq, int32_t k, uint32_t flags, float* out_s,
                      int64_t* out_i);  // replicated handles: synchronous host calls, round-robin
bool ls_group_is_replicated(const ls_index* ix);
int ls_group_check(ls_index* ix, hipStream_t s);
void ls_group_destroy(ls_index* ix);
int ls_group_add(ls_index* ix, const float* rows, int64_t n_add);
int ls_group_reconstruct(ls_index* ix, int64_t row0, int64_t count, float* out);
int ls_group_set_base(ls_index* ix, int64_t base);
int ls_group_debug_option(ls_index* ix, int32_t which, int32_t value);
int64_t ls_group_debug_counter(ls_index* ix, int32_t which);
int ls_group_set_profiling(ls_index* ix, int32_t enabled);
int ls_group_last_kernel_ms(ls_index* ix, float* scan_ms, float* total_ms);
