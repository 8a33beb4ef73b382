// ls_callers.hip - the synchronous host search: ls_search (the reference's blocking index.search,
// search/engine.py:250). Two parts: ONE call on one of the handle's two host slots (begin: stage + queue the
// launch; finish: poll the pinned results), and the queue that serves concurrent callers as ONE pass
// (leaders, waiters, the gather window). Everything the GPU does is queued through ls_i_search_on_stream
// (ls_api.hip); this file is host code only.
#include "ls_index.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>
#include <sched.h>

#define LS_GATHER_SLOW_US 110.0  // a combined call longer than this is a "long pass" (ls_search)
#ifndef LS_GATHER_MAX_US
#define LS_GATHER_MAX_US 60.0    // the longest a leader waits for the callers seen lately ...
#endif
#ifndef LS_GATHER_DIV
#define LS_GATHER_DIV 3.0        // ... and the fraction of a call's running estimate it may spend on that
#endif
#ifndef LS_GATHER_QUIET_US
#define LS_GATHER_QUIET_US 6.0   // with >= LS_GATHER_QUIET_MIN callers around, 3/4 of them queued and no arrival for this long: go
#endif
#define LS_GATHER_QUIET_MIN 16

// ---- the synchronous host search (ls_search: the reference's index.search, search/engine.py:250) -----
// A call runs in two phases on one of the handle's two host slots (ls_host_slot, ls_index.h):
//   begin  - under the slot's mutex AND the handle's mutex: stage the query, queue the launch(es);
//   finish - under the slot's mutex only: poll the pinned completion words / result granules (or sleep in
//            hipStreamSynchronize), run the rare same-launch retry, hand the results back.
// Between the two the handle's mutex is free, so the NEXT call can queue its launch behind this one while
// this one still waits for its answer ("overlapped": single-group scan-path calls that the host can poll;
// everything else - batched calls, several query groups, group handles - is "exclusive": it waits for both
// slots and keeps the handle's mutex to the end, exactly the round-4 behaviour). Two single-query callers
// used to alternate as lone launches with the GPU idle from the end of one call's selection to the next
// call's launch (~15 us of every 65: profiles/ab/r04_concurrent_callers_replicas.txt); overlapped, the
// second launch is already queued when the first one's scan ends.
struct ls_host_call {
    ls_index* ix = nullptr;
    ls_host_slot* S = nullptr;
    std::unique_lock<std::mutex> slot_lk, other_lk, mu_lk;
    const float* q = nullptr;
    int64_t nq = 0;
    int32_t k = 0;
    uint32_t flags = 0;
    float* out_scores = nullptr;
    int64_t* out_indices = nullptr;
    bool group = false, spin = false, out_direct = false, queued = false, in_direct = false;
    hipStream_t stream = nullptr;
    int rc = LS_OK;
    int gen = -1;  // the scratch generation an overlapped call was given (ls_index::force_gen)
};

#ifdef LS_LEAD_TRACE  // (variant build: where a leader's begin / finish goes, debug counters 40-47, cumulative ns)
std::atomic<uint64_t> g_lead_trace[8];
struct ls_trace_clock {
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(int i) {
        const auto n = std::chrono::steady_clock::now();
        g_lead_trace[i].fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(n - t).count(), std::memory_order_relaxed);
        t = n;
    }
};
#define LS_LAP(c, i) (c).lap(i)
#else
struct ls_trace_clock { };
#define LS_LAP(c, i) ((void)(c))
#endif

static int host_call_begin_impl(ls_host_call& c) {
    ls_index* ix = c.ix;
    const int64_t nq = c.nq;
    const int32_t k = c.k;
    int rc = LS_OK;
    ls_trace_clock tc;
    if (ix->group) {  // the group handle has its own host path (ls_shard.hip); nothing to overlap here
        c.group = true;
        c.mu_lk = std::unique_lock<std::mutex>(ix->mu);
        return c.rc = ls_group_search(ix, c.q, true, nq, k, c.flags & LS_FLAG_NORMALIZE, c.out_scores,
                                      c.out_indices, nullptr);
    }
    // a slot: the one whose turn it is, or the other one if that is free right now
    unsigned si = ix->hs_rr.fetch_add(1, std::memory_order_relaxed) % LS_HOST_SLOTS;
    c.slot_lk = std::unique_lock<std::mutex>(ix->hs[si].mu, std::try_to_lock);
    if (!c.slot_lk.owns_lock()) {
        const unsigned sj = (si + 1) % LS_HOST_SLOTS;
        c.slot_lk = std::unique_lock<std::mutex>(ix->hs[sj].mu, std::try_to_lock);
        if (c.slot_lk.owns_lock()) si = sj;
        else c.slot_lk = std::unique_lock<std::mutex>(ix->hs[si].mu);
    }
    c.mu_lk = std::unique_lock<std::mutex>(ix->mu);
    const size_t qn = (size_t)nq * ix->g.d, on = (size_t)nq * k;
    const bool small_call = nq <= ls_i_scan_path_max_nq(ix, k);
    c.out_direct = on <= (size_t)(1 << 16);
    // Small scan-path calls: the finalize workgroup of every query writes tagged result granules
    // (k <= LS_OUT_GRAN_MAX_K) or drained rows + a completion word into pinned host memory; the host
    // spins on those instead of sleeping in hipStreamSynchronize (whose wake-up costs more than the
    // 47 us scan's launch). Falls back to the stream sync after 2 ms.
    c.spin = small_call && c.out_direct && !ls_i_batched_eligible(ix, nq, k) && ix->n > 0;
    const bool overlapped = ix->opt_overlap_calls && c.spin && ls_i_scan_group_count(ix, nq, k) == 1;
    if (!overlapped) {
        // exclusive: no other host call in flight (lock order: both slots, then the handle)
        c.mu_lk.unlock();
        c.slot_lk.unlock();
        std::lock(ix->hs[0].mu, ix->hs[1].mu);
        c.slot_lk = std::unique_lock<std::mutex>(ix->hs[0].mu, std::adopt_lock);
        c.other_lk = std::unique_lock<std::mutex>(ix->hs[1].mu, std::adopt_lock);
        c.mu_lk.lock();
        si = 0;
    }
    ls_host_slot& S = ix->hs[si];
    c.S = &S;
    LS_HIP(hipSetDevice(ix->device));
    if (overlapped && !S.stream) LS_HIP(hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking));
    // an overlapped call runs on its slot's stream: the next call's scan workgroups move onto the CUs as this
    // call's retire, under its selection workgroup and its tail (one stream would order kernel behind kernel)
    hipStream_t s = overlapped ? S.stream : ix->own_stream;
    c.stream = s;
    if ((rc = ls_grow(&S.d_qraw, &S.qraw_cap, qn)) != LS_OK) return c.rc = rc;
    if ((rc = ls_grow_pinned(&S.h_q, &S.h_q_cap, qn)) != LS_OK) return c.rc = rc;
    if (on > S.out_cap) {
        size_t c1 = S.out_cap, c2 = S.out_cap;
        if ((rc = ls_grow(&S.d_out_s, &c1, on)) != LS_OK) return c.rc = rc;
        if ((rc = ls_grow(&S.d_out_i, &c2, on)) != LS_OK) return c.rc = rc;
        S.out_cap = std::min(c1, c2);
    }
    if (on > S.h_out_cap) {
        size_t c1 = S.h_out_cap, c2 = S.h_out_cap;
        if ((rc = ls_grow_pinned(&S.h_out_s, &c1, on)) != LS_OK) return c.rc = rc;
        if ((rc = ls_grow_pinned(&S.h_out_i, &c2, on)) != LS_OK) return c.rc = rc;
        S.h_out_cap = std::min(c1, c2);
    }
    // Pinned host buffers are device-visible. Results: the selection writes the output rows into
    // h_out_* over PCIe itself (no copy command behind the kernel). Queries: the scan workgroups read
    // the pinned copy themselves (single queries). A stand-alone probe (tools/host_roundtrip_probe.hip)
    // prices that read at 7.4 us for 448 idle workgroups against 2.7 us behind a copy command and
    // 1.5 us through the kernel arguments - but in the scan kernel the first corpus tile's loads are
    // in flight before the query is touched, so the read hides, and the copy command measured 1.6-2 us
    // SLOWER per call (profiles/ab/r04_hostapi_selection.txt; the copy path was removed in round 6).
    // (only single queries: every workgroup reads the whole query block - 256 x 16 x 4 KB over PCIe otherwise)
    const bool in_direct = c.in_direct = small_call && nq == 1;
    if (c.spin && !S.h_done) {
        LS_HIP(hipHostMalloc((void**)&S.h_done, sizeof(u32) * LS_QUERIES_PER_LAUNCH_MAX, hipHostMallocDefault));
        memset(S.h_done, 0, sizeof(u32) * LS_QUERIES_PER_LAUNCH_MAX);
    }
    LS_LAP(tc, 4);
    memcpy(S.h_q, c.q, qn * sizeof(float));
    if (!in_direct)
        LS_HIP(hipMemcpyAsync(S.d_qraw, S.h_q, qn * sizeof(float), hipMemcpyHostToDevice, s));
    LS_LAP(tc, 5);
    S.retry_groups.clear();
    if (c.spin) {
        if (on > S.h_out_g_cap) {
            if (S.h_out_g) (void)hipHostFree(S.h_out_g);
            S.h_out_g = nullptr;
            S.h_out_g_cap = 0;
            const size_t cap = std::max<size_t>(on, 4096);
            LS_HIP(hipHostMalloc((void**)&S.h_out_g, cap * sizeof(ls_out_gran), hipHostMallocDefault));
            memset(S.h_out_g, 0, cap * sizeof(ls_out_gran));
            S.h_out_g_cap = cap;
        }
        if (++S.done_seq >= LS_DONE_RETRY) S.done_seq = 1;  // the top bit is the retry answer
        ix->done_base = S.h_done;
        ix->gran_out_base = S.h_out_g;
    }
    ix->cur_retry = &S.retry_groups;
    ix->cur_done_seq = S.done_seq;
    ix->force_gen = c.gen = overlapped ? (int)si : -1;
    rc = ls_i_search_on_stream(ix, in_direct ? S.h_q : S.d_qraw, nq, k, c.flags & LS_FLAG_NORMALIZE,
                               c.out_direct ? S.h_out_s : S.d_out_s, c.out_direct ? S.h_out_i : S.d_out_i, s, true);
    ix->done_base = nullptr;
    ix->gran_out_base = nullptr;
    ix->cur_retry = nullptr;
    ix->force_gen = -1;
    LS_LAP(tc, 6);
    if (rc != LS_OK) return c.rc = rc;
    if (!c.out_direct) {
        LS_HIP(hipMemcpyAsync(S.h_out_s, S.d_out_s, on * sizeof(float), hipMemcpyDeviceToHost, s));
        LS_HIP(hipMemcpyAsync(S.h_out_i, S.d_out_i, on * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    }
    c.queued = true;
    if (overlapped) {
        if (ix->hs[(si + 1) % LS_HOST_SLOTS].mu.try_lock()) ix->hs[(si + 1) % LS_HOST_SLOTS].mu.unlock();
        else ix->n_overlapped_calls++;  // (the other slot's call is still in flight)
        c.mu_lk.unlock();  // the next call may queue its launch now
    }
    return LS_OK;
}

static int host_call_begin(ls_host_call& c) {
    const int rc = host_call_begin_impl(c);  // (LS_HIP returns straight out of it)
    if (rc != LS_OK) {
        c.rc = rc;
        if (c.ix && !c.ix->group) {
            c.ix->done_base = nullptr;
            c.ix->gran_out_base = nullptr;
            c.ix->cur_retry = nullptr;
            c.ix->force_gen = -1;
        }
    }
    return rc;
}

static int host_call_finish(ls_host_call& c) {
    if (c.group || c.rc != LS_OK || !c.queued) return c.rc;
    ls_index* ix = c.ix;
    ls_host_slot& S = *c.S;
    const int64_t nq = c.nq;
    const int32_t k = c.k;
    const size_t on = (size_t)nq * k;
    hipStream_t s = c.stream;
    float* out_scores = c.out_scores;
    int64_t* out_indices = c.out_indices;
    const int64_t base = ix->base;  // (ls_set_base waits for the calls in flight)
    int rc = LS_OK;
    // spin until every query is final: all k of its result granules carry the call's sequence number
    // in both halves (ls_fin_params::out_gran), or - if accepted - its completion word holds the retry
    // answer; gives up after 2 ms and lets the caller sleep in hipStreamSynchronize
    auto wait_words = [&](bool accept_retry, bool* retry) -> bool {
        const auto t0 = std::chrono::steady_clock::now();
        const u32 seq = S.done_seq;
        int64_t i = 0;   // queries below i are final (granules never change back)
        int32_t j = 0;   // granules below j of query i carry the tag
        bool any_retry = false;
        const bool granules = k <= LS_OUT_GRAN_MAX_K;
        for (unsigned it = 0;; ++it) {
            for (; i < nq; ++i, j = 0) {
                const u32 w = __atomic_load_n(&S.h_done[i], __ATOMIC_ACQUIRE);
                if (accept_retry && w == (seq | LS_DONE_RETRY)) {
                    any_retry = true;
                    continue;
                }
                if (!granules) {  // drained rows + completion word
                    if (w == seq) continue;
                    break;
                }
                // (decoded as they are recognised: at k = 1000 a second pass over 16 KB of granules would
                // cost the host more than the drain + completion word it replaces cost the GPU)
                const ls_out_gran* g = S.h_out_g + (size_t)i * k;
                float* os = out_scores + (size_t)i * k;
                int64_t* oi = out_indices + (size_t)i * k;
                for (; j < k; ++j) {
                    if (__atomic_load_n(&g[j].tag_lo, __ATOMIC_ACQUIRE) != seq ||
                        __atomic_load_n(&g[j].tag_hi, __ATOMIC_ACQUIRE) != seq)
                        break;
                    os[j] = g[j].score;
                    oi[j] = g[j].row == 0xffffffffu ? (int64_t)-1 : base + (int64_t)g[j].row;
                }
                if (j < k) break;
            }
            if (i == nq) {
                if (retry) *retry = any_retry;
                return true;
            }
            ls_cpu_relax();
            if ((it & 1023) == 1023 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2))
                return false;
        }
    };
    bool done = false;
    if (c.spin) {
        bool retry = false;
        done = wait_words(true, &retry);
        if (!done) {  // slow launch: sleep until the stream has drained, then every word is final
            __atomic_fetch_add(&ix->n_spin_timeouts, 1ull, __ATOMIC_RELAXED);
            LS_HIP(hipStreamSynchronize(s));
            done = wait_words(true, &retry);
        }
        if (done && retry) {
            // Same-launch selection jobs that could not prove their emitted keys complete (clustered
            // rows, ties) or gave up waiting: the stand-alone finalize behind the scan - the kernel
            // boundary makes the score vector visible - answering through the same completion words.
            // (Queued under the handle's mutex: another call may be queueing its launch right now. This
            // call still owns its slot, i.e. its scratch generation: the retry's inputs are intact.)
            std::unique_lock<std::mutex> relock;
            if (!c.mu_lk.owns_lock()) relock = std::unique_lock<std::mutex>(ix->mu);
            LS_HIP(hipSetDevice(ix->device));
            for (const ls_fin_batch& gb : S.retry_groups) {  // one batch per launch of the call
                if (!gb.p0.done) continue;
                ls_fin_batch jobs = gb;  // the stand-alone finalize of the group's flagged jobs
                jobs.njobs = 0;
                jobs.p0.wait = 0;  // behind the kernel boundary every granule is there
                jobs.p0.keys_cap = LS_FINAL_CAP;
                const int64_t q0 = gb.p0.done - S.h_done;  // the group's first query within the call
                for (int j = 0; j < gb.njobs; ++j) {
                    const int64_t qi = q0 + gb.idx[j];
                    if (qi < 0 || qi >= nq || S.h_done[qi] != (S.done_seq | LS_DONE_RETRY)) continue;
                    if (!gb.p0.S) {
                        // the job rode on a launch that wrote no score vectors: serve the query again, alone
                        // (scan kernel: the same bits; its selection gets its own launch and answers through the
                        // same completion word / granules). The call still owns its slot: the query copy is intact.
                        ix->done_base = S.h_done + qi;
                        ix->gran_out_base = k <= LS_OUT_GRAN_MAX_K ? S.h_out_g + (size_t)qi * k : nullptr;
                        ix->cur_retry = nullptr;
                        ix->cur_done_seq = S.done_seq;
                        ix->force_gen = c.gen;
                        ix->reserving = true;
                        rc = ls_i_search_on_stream(ix, (c.in_direct ? S.h_q : S.d_qraw) + (size_t)qi * ix->g.d, 1, k, c.flags & LS_FLAG_NORMALIZE,
                                                   (c.out_direct ? S.h_out_s : S.d_out_s) + (size_t)qi * k,
                                                   (c.out_direct ? S.h_out_i : S.d_out_i) + (size_t)qi * k, s, true);
                        ix->reserving = false;
                        ix->done_base = nullptr;
                        ix->gran_out_base = nullptr;
                        ix->force_gen = -1;
                        if (rc != LS_OK) return rc;
                        ix->n_mq_reserved++;
                        continue;
                    }
                    jobs.idx[jobs.njobs++] = gb.idx[j];
                }
                if (jobs.njobs) {
                    if ((rc = ls_launch_finalize(jobs, s)) != LS_OK) return rc;
                    ix->n_launches_total++;
                }
            }
            ix->n_same_launch_retries++;
            if (relock.owns_lock()) relock.unlock();
            done = wait_words(false, nullptr);
        }
    }
    S.retry_groups.clear();
    if (!done) LS_HIP(hipStreamSynchronize(s));
    if (c.spin && k <= LS_OUT_GRAN_MAX_K) {
        if (done) return LS_OK;  // (unpacked while waiting)
        std::atomic_thread_fence(std::memory_order_acquire);  // behind the drained stream
        for (size_t e = 0; e < on; ++e) {
            const ls_out_gran& g = S.h_out_g[e];
            out_scores[e] = g.score;
            out_indices[e] = g.row == 0xffffffffu ? (int64_t)-1 : base + (int64_t)g.row;
        }
        return LS_OK;
    }
    memcpy(out_scores, S.h_out_s, on * sizeof(float));
    memcpy(out_indices, S.h_out_i, on * sizeof(int64_t));
    return LS_OK;
}

// One synchronous host search, begin + finish (callers that do not go through the combining queue)
static int host_search_locked(ls_index* ix, const float* q, int64_t nq, int32_t k, uint32_t flags,
                              float* out_scores, int64_t* out_indices) {
    ls_host_call c;
    c.ix = ix; c.q = q; c.nq = nq; c.k = k; c.flags = flags; c.out_scores = out_scores; c.out_indices = out_indices;
    host_call_begin(c);
    return host_call_finish(c);
}

// ---- combining concurrent callers ---------------------------------------------------------------
// The reference's event loop makes one blocking index.search per query (search/engine.py:250), but
// an MCP server with several clients, or a threaded caller, has several of them in flight. The scan
// path serves up to 32 queries per corpus pass (fp32: ls_mq.hip, ~50 us for 16, ~65 us for 32 at N = 200 k; one
// query alone: 47 us), so requests that arrive while a search is running are not queued behind the mutex one by
// one: they wait in a queue, and whoever holds the leadership serves ALL compatible waiters (same k, same
// flags, as many queries as one launch carries: ls_i_scan_path_max_nq) as ONE batch, then hands their results
// back. A lone caller becomes leader at once and pays nothing; waiters poll their own request's flag (as many of
// them as the process has CPUs for) or sleep on its condition variable (round 6, ls_spin_cap). Round 5: the
// leadership is released as soon as the batch's launch is QUEUED
// (host_call_begin), so the next leader queues the requests that arrived meanwhile behind it while the
// first one polls for its answers. Results are those of the separate calls: every query's arithmetic
// is the same whatever group it rides in.
struct ls_req {
    const float* q;
    int64_t nq;
    int32_t k;
    uint32_t flags;
    float* out_s;
    int64_t* out_i;
    int rc = LS_OK;
    std::atomic<bool> done{false};     // set LAST by the serving thread (release): the waiter may return - and its request,
                                       // which lives on its stack, vanish - the moment it sees it, without the queue's mutex
    std::atomic<bool> taken{false};    // popped into a batch some leader is serving (set under q_mu; a waiter polls it)
    char err[256] = "";
    std::condition_variable cv;        // where THIS request's caller sleeps once it may not (or no longer) spin ...
    bool parked = false;               // ... (under q_mu) and whether it does: whoever finishes its batch, or hands on the
                                       // leadership while it heads the queue, wakes it - and nobody else (round 6)
};

// A queued caller polls the queue's epoch before it sleeps on the condition variable: for about two calls' worth
// of the handle's running estimate (its answer is that far away at most when it is next in line), 40..300 us -
// not for a fixed 300 us whatever the call takes (16 callers kept 15 cores spinning)
// ... and only as many waiters poll at all as the process has CPUs for (round 6: the GPU box's container has a
// 16-CPU cgroup quota; 32 callers spinning were throttled for half of every period - cpu.stat nr_throttled - and ran at
// 80 k q/s where 16 ran at 120-160 k; 16 UNRELATED busy threads next to 16 callers halved them the same way). The
// CPUs: the affinity mask, capped by the cgroup's quota (v2 cpu.max, v1 cpu.cfs_quota_us), or LS_SPIN_CPUS; one is
// the leader's, two stay free for the callers' own work, the rest may spin; everyone else sleeps on his request's
// condition variable at once.
static int ls_spin_cap() {
    static const int cap = [] {
        long cpus = (long)std::thread::hardware_concurrency();
        cpu_set_t cs;
        if (sched_getaffinity(0, sizeof(cs), &cs) == 0 && CPU_COUNT(&cs) > 0) cpus = CPU_COUNT(&cs);
        long quota = -1, period = 0;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char qs[32] = "";
            if (fscanf(f, "%31s %ld", qs, &period) == 2 && strcmp(qs, "max") != 0) quota = atol(qs);
            fclose(f);
        } else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(f1, "%ld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(f2, "%ld", &period) != 1) period = 0;
                fclose(f2);
            }
        }
        if (quota > 0 && period > 0) cpus = std::min(cpus, (quota + period - 1) / period);
        if (const char* e = getenv("LS_SPIN_CPUS")) cpus = atol(e);
        return (int)std::max<long>(1, cpus - 3);  // (measured on the 16-CPU box, 16 / 32 callers, k q/s: 15 pollers 138 / 119 - throttled -,
                                                  // 13: 138 / 135, 11: 120 / 133, 8: 125 / 129, 4: 130 / 123)
    }();
    return cap;
}
#define LS_WAITER_SPIN_MIN_US 40.0
#define LS_WAITER_SPIN_MAX_US 300.0
struct ls_served {  // one batch between its begin and its finish
    ls_host_call call;
    std::vector<ls_req*> batch;
    bool combined = false;
};

static void serve_begin(ls_index* ix, ls_served& sv) {
    std::vector<ls_req*>& batch = sv.batch;
    ls_host_call& c = sv.call;
    c.ix = ix;
    c.k = batch[0]->k;
    c.flags = batch[0]->flags;
    if (batch.size() == 1) {
        ls_req* r = batch[0];
        c.q = r->q; c.nq = r->nq; c.out_scores = r->out_s; c.out_indices = r->out_i;
        host_call_begin(c);
        return;
    }
    sv.combined = true;
    ls_trace_clock tc;
    int64_t total = 0;
    for (ls_req* r : batch) total += r->nq;
    c.nq = total;
    // (the staging lives in the call's slot, which is only known inside begin: stage in a temporary
    // vector of the batch first - a few KB)
    static thread_local std::vector<float> tq;
    const int32_t d = ix->g.d;
    tq.resize((size_t)total * d);
    int64_t at = 0;
    for (ls_req* r : batch) {
        memcpy(tq.data() + at * d, r->q, (size_t)r->nq * d * sizeof(float));
        at += r->nq;
    }
    static thread_local std::vector<float> ts;
    static thread_local std::vector<int64_t> ti;
    ts.resize((size_t)total * c.k);
    ti.resize((size_t)total * c.k);
    c.q = tq.data(); c.out_scores = ts.data(); c.out_indices = ti.data();
    LS_LAP(tc, 0);
    host_call_begin(c);  // (copies the queries into the slot's pinned buffer before it returns)
    LS_LAP(tc, 1);
}

static void serve_finish(ls_index* ix, ls_served& sv) {
    ls_host_call& c = sv.call;
    ls_trace_clock tc;
    const int rc = host_call_finish(c);
    LS_LAP(tc, 2);
    const int32_t k = c.k;
    int64_t at = 0;
    for (ls_req* r : sv.batch) {
        r->rc = rc;
        if (rc != LS_OK) snprintf(r->err, sizeof(r->err), "%s", ls_last_error());
        else if (sv.combined) {
            memcpy(r->out_s, c.out_scores + at * k, (size_t)r->nq * k * sizeof(float));
            memcpy(r->out_i, c.out_indices + at * k, (size_t)r->nq * k * sizeof(int64_t));
        }
        at += r->nq;
    }
    LS_LAP(tc, 3);
    if (sv.combined) {
        std::lock_guard<std::mutex> ql(ix->q_mu);
        ix->n_combined_batches++;
        ix->n_combined_requests += sv.batch.size();
    }
}

extern "C" {

int ls_search(ls_index* ix, const float* q, int64_t nq, int32_t k, uint32_t flags,
              float* out_scores, int64_t* out_indices) {
    int rc = ls_i_check_search_args(ix, q, nq, k, flags & ~(LS_FLAG_ASYNC | LS_FLAG_PIPELINE), out_scores,
                                    out_indices);
    if (rc != LS_OK) return rc;
    if (nq == 0) return LS_OK;
    flags &= LS_FLAG_NORMALIZE;
    if (ls_group_is_replicated(ix)) return ls_replica_search(ix, q, nq, k, flags, out_scores, out_indices);
    if (!ix->opt_combine || nq > ls_i_scan_path_max_nq(ix, k))  // what fills a pass by itself gains nothing from company
        return host_search_locked(ix, q, nq, k, flags, out_scores, out_indices);
    ls_req me{q, nq, k, flags, out_scores, out_indices};
    // (round 6: 32 callers ran at 65 k q/s where 16 ran at 100 k - 470 us per batch for a 100 us pass. Every arrival
    // bumped the ONE epoch every waiter polled, so each of N arrivals sent the other waiters through the queue's
    // mutex: O(N^2) acquisitions per batch, the losers parked in futex_wait. Now a waiter whose request is in a
    // batch polls its own `done` flag and nothing else, a queued one polls `lead_epoch` - bumped only when the
    // leadership or a host slot comes free - and neither takes the mutex to find out; arrivals bump `q_epoch`,
    // which only the one gathering leader reads.)
    auto q_lock = [](std::unique_lock<std::mutex>& l) {
        for (int i = 0; i < 64; ++i) {  // (a few us: the sections are short; then park)
            if (l.try_lock()) return;
            for (int j = 0; j < 16; ++j) ls_cpu_relax();
        }
        l.lock();
    };
    std::unique_lock<std::mutex> lk(ix->q_mu, std::defer_lock);
    q_lock(lk);
    ix->req_q.push_back(&me);
    ix->q_len.store((int64_t)ix->req_q.size(), std::memory_order_release);
    ix->q_epoch.fetch_add(1, std::memory_order_release);  // (a leader waiting to form its batch counts the arrivals)
    {   // running estimate of the time between two arrivals (what the gather below asks before it waits)
        const auto now = std::chrono::steady_clock::now();
        if (ix->arrivals_seen++) {
            const double gap = std::min(1e4, std::chrono::duration<double, std::micro>(now - ix->last_arrival).count());
            ix->arrival_gap_us += (gap - ix->arrival_gap_us) / 8.0;
        }
        ix->last_arrival = now;
    }
    bool answered = false;  // seen `done` without holding the mutex
    while (!me.done.load(std::memory_order_acquire)) {
        if (ix->leader_active || me.taken.load(std::memory_order_relaxed)) {  // (taken: my request is in a batch someone is serving)
            // The answer is typically 50-150 us away and a futex wake-up costs tens of us (times the callers
            // woken at once): poll for a while before sleeping (round 5)
            const uint64_t seen = ix->lead_epoch.load(std::memory_order_acquire);
            const double spin_us = std::min(LS_WAITER_SPIN_MAX_US, std::max(LS_WAITER_SPIN_MIN_US, 2.0 * ix->call_us_est));
            // (... and fewer pollers the more callers there are beyond the CPUs: 64 callers on the 16-CPU box, pollers
            // 13 / 8 / 4 / 2 / 1: 149 / 171 / 197 / 211 / 197 k q/s - the pollers stood in the way of the callers that
            // were waking up; 8-32 callers are within 5-10 % of each other for every cap, 13 the best: half a poller
            // less per caller beyond the CPU count, two at least)
            const int cap = ls_spin_cap();
            const int64_t beyond = std::max<int64_t>(0, ix->peak_callers - (cap + 3));
            const int pollers = (int)std::max<int64_t>(std::min(2, cap), cap - beyond / 2);
            lk.unlock();
            bool changed = false;
            const bool may_spin = ix->spinners.fetch_add(1, std::memory_order_relaxed) < pollers;  // (a CPU to poll on)
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0; may_spin && !changed; ++it) {
                for (int i = 0; i < 32; ++i) ls_cpu_relax();
                if (me.done.load(std::memory_order_acquire)) {
                    answered = true;
                    break;
                }
                // (a request still in the queue may have to lead: it looks again when the leadership or a slot came free)
                changed = !me.taken.load(std::memory_order_acquire) && ix->lead_epoch.load(std::memory_order_acquire) != seen;
                if ((it & 15) == 15 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
            }
            ix->spinners.fetch_sub(1, std::memory_order_relaxed);
            if (answered) break;  // (nothing of the queue is touched any more: no mutex)
            q_lock(lk);
            // Sleep - on the request's own condition variable: its batch's server wakes it when the answer is there, a
            // leader that hands the leadership on wakes it if it heads the queue; nobody is woken for anything else.
            // (Checked under the mutex both wakers hold: a taken request sleeps until it is done, a queued one while
            // somebody leads.)
            // (a poller that saw the leadership change and lost the race for it goes on polling: `changed`)
            if (!changed && !me.done.load(std::memory_order_acquire) && (me.taken.load(std::memory_order_relaxed) || ix->leader_active)) {
                me.parked = true;
                ix->n_waiter_parks++;
                me.cv.wait(lk);
                me.parked = false;
            }
            continue;
        }
        // lead ONE batch: queue its launch, pass the leadership on, then wait for its results
        ix->leader_active = true;
        const auto t_lead = std::chrono::steady_clock::now();  // (phase clocks of the leader: debug counters 28-30)
        // A call is still in flight. Two callers taking turns (one request in flight, one waiting): queue
        // the waiting one's launch NOW, behind the running one - the GPU then goes from scan to scan instead
        // of idling from one call's last result to the next call's launch (2 callers: 15.5 k -> 18 k
        // queries/s, p50 130 -> 110 us). More callers than that: every pass costs the same 50-60 us however
        // many queries ride in it, so the batch is formed when the call in flight has handed its results
        // back and takes along everything that arrived meanwhile (queued at once behind the running call, a
        // batch held 1-2 requests and 8 callers fell from 51 k to 35 k queries/s; forming it "as late as
        // keeps the launches back to back" from a running estimate of the call time: 43 k).
        // (round 5, second step: the slots have their own streams, so a second batch may also go early when
        // EVERY caller the handle has seen lately is either in the running batch or already queued - waiting
        // for the running call could add nobody; `peak_callers` is a slowly decaying maximum of that count)
        // (round 6, more than 32 callers: once the queue alone fills a pass nobody can join this batch any more - it is
        // queued at once behind the running one, on the other slot: the GPU goes from pass to pass. 64 C-thread callers, same
        // box: d = 384 131 -> 151-174 k q/s, d = 1024 k = 1000 77 -> 148 k (p50 787 -> 400 us); 32 callers unchanged; debug option 23)
        const int64_t full_pass = ls_i_scan_path_max_nq(ix, ix->req_q.front()->k);
        auto go_early = [&]() {
            const int64_t total = ix->requests_in_flight + (int64_t)ix->req_q.size();
            if (ix->opt_overlap_calls && ix->opt_full_early && ix->calls_in_flight < LS_HOST_SLOTS && (int64_t)ix->req_q.size() >= full_pass) return true;
            // (up to 8 callers: with more, the two halves are big passes that only slow each other down -
            // 16 callers 76.5 k early vs 81.2 k waiting, 8 callers 51.4 k vs 45.9 k, 4 callers 32.9 k vs 25.6 k)
            // (long passes - d = 1024: 140 us - gain nothing from running two at once, they share one HBM; their
            // callers are gathered into ONE pass instead, below)
            if (ix->opt_gather && ix->call_us_est > LS_GATHER_SLOW_US) return false;
            // (with the gather on, only STRAGGLERS of a short pass go early: every caller seen lately is in flight
            // or queued, so waiting could add nobody - Python threads hand the GIL around and arrive spread over
            // more than the gather window: 8 Python callers 46.6 k -> 51-67 k q/s, C threads unchanged at 75-80 k)
            if (ix->opt_gather)
                return ix->opt_overlap_calls && ix->calls_in_flight < LS_HOST_SLOTS && total > 2 &&
                       total >= ix->peak_callers && total <= ix->opt_early_cap;
            return ix->opt_overlap_calls && ix->calls_in_flight < LS_HOST_SLOTS &&
                   (total <= 2 || (total >= ix->peak_callers && total <= ix->opt_early_cap));
        };
        while (ix->calls_in_flight > 0 && !go_early()) {
            // (what may end this wait: the call in flight hands its results back - lead_epoch - or, while few enough
            // callers are around for a second batch to go early, an arrival - q_epoch. With more callers than that
            // an arrival changes nothing, and a leader that re-took the mutex on each of 30 arrivals stood in their way)
            const bool arrivals_matter = ix->peak_callers <= ix->opt_early_cap;
            std::atomic<uint64_t>& ep = arrivals_matter ? ix->q_epoch : ix->lead_epoch;
            const uint64_t seen = ep.load(std::memory_order_acquire);  // (as above: poll, then sleep)
            const double spin_us = std::min(LS_WAITER_SPIN_MAX_US, std::max(LS_WAITER_SPIN_MIN_US, 2.0 * ix->call_us_est));
            // (a full queue matters only while a host slot is free: with both in flight - more than three passes' worth of
            // callers - it would end every poll at once and send this leader round and round through the mutex)
            const bool full_matters = ix->opt_overlap_calls && ix->opt_full_early && ix->calls_in_flight < LS_HOST_SLOTS;
            lk.unlock();
            bool changed = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0; !changed; ++it) {
                for (int i = 0; i < 32; ++i) ls_cpu_relax();
                // (... or the queue has grown to a full pass: read through the arrivals' atomic, not through the mutex)
                changed = ep.load(std::memory_order_acquire) != seen ||
                          (full_matters && ix->q_len.load(std::memory_order_acquire) >= full_pass);
                if ((it & 15) == 15 && std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
            }
            q_lock(lk);
            if (!changed && ix->calls_in_flight > 0 && ep.load(std::memory_order_acquire) == seen)
                ix->q_cv.wait(lk);
        }
        // Gather (round 5; first for long passes only, then for all - 2 / 4 / 8 callers at d = 384: 23.7 / 40.0 /
        // 64-75 k -> 26 / 46.5 / 70 k q/s, d = 1024: 8.7 / 15.0 / 30.0 -> 12.1 / 22.5 / 36.5 k; with it only stragglers
        // go early, above): nothing is in flight and fewer requests are queued than callers were
        // seen lately - the others are on their way back from the pass that just ended (their results were
        // handed out microseconds ago). Launching now would split the callers into two groups that wait for each
        // other's pass forever (4 callers, d = 1024: 2 + 2, every call 2 x 158 us); a short wait puts them all
        // into ONE pass. Bounded by a third of the running estimate of a call, at most LS_GATHER_MAX_US.
        // (round 6, open-loop record profiles/ab/r06_open_loop.txt: at 5-10 k requests/s - one arrival per 100-200 us -
        // the window mostly expired empty and cost the lone request it delayed 15-55 us: the leader waits only when
        // the arrival rate seen lately makes another request within the window likelier than not)
        const double budget_us = std::min<double>(LS_GATHER_MAX_US, ix->call_us_est / LS_GATHER_DIV);
        if (ix->opt_gather && ix->calls_in_flight == 0 && (int64_t)ix->req_q.size() < ix->peak_callers &&
            ix->arrival_gap_us < 3.5 * budget_us) {
            const auto t0 = std::chrono::steady_clock::now();
            while ((int64_t)ix->req_q.size() < ix->peak_callers) {
                // (the queue's length is read through an atomic the arrivals maintain: the leader stays out of their
                // way and takes the mutex ONCE, when everyone is there or the window is over)
                const int64_t want = ix->peak_callers;
                lk.unlock();
                bool late = false, quiet = false;
                int64_t len_seen = ix->q_len.load(std::memory_order_acquire);
                double changed_at = 0.0;
                for (unsigned it = 0; !late && !quiet && ix->q_len.load(std::memory_order_acquire) < want; ++it) {
                    for (int i = 0; i < 16; ++i) ls_cpu_relax();
                    if ((it & 7) == 7) {
                        const double now = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                        late = now > budget_us;
                        // (many callers, some of them asleep - more callers than CPUs: those take tens of us to come back;
                        // once three quarters are here and the arrivals have stopped, the pass goes without the rest)
                        const int64_t len = ix->q_len.load(std::memory_order_acquire);
                        if (len != len_seen) { len_seen = len; changed_at = now; }
                        quiet = want >= LS_GATHER_QUIET_MIN && len >= want - want / 4 && now - changed_at > LS_GATHER_QUIET_US;
                    }
                }
                q_lock(lk);
                if (quiet && !late) break;
                if (late) {
                    // the callers that did not come are gone (or slower than the window): stop waiting for them
                    // quickly - a lone caller after a burst of 16 would otherwise pay the window for ~500 calls
                    const int64_t here = ix->requests_in_flight + (int64_t)ix->req_q.size();
                    ix->peak_callers = std::max<int64_t>(here, ix->peak_callers - std::max<int64_t>(1, ix->peak_callers / 4));
                    break;
                }
            }
        }
        const auto t_formed = std::chrono::steady_clock::now();
        ix->n_lead_wait_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_formed - t_lead).count();
        ls_served sv;
        ls_req* head = ix->req_q.front();
        int64_t total = 0;
        const int batch_cap = ls_i_scan_path_max_nq(ix, head->k);  // queries ONE pass carries (32 where ls_mq serves the index)
        while (!ix->req_q.empty()) {
            ls_req* r = ix->req_q.front();
            // (the head request always goes: an option changed since it was queued may have lowered the cap under it)
            if (r != head && (r->k != head->k || r->flags != head->flags || total + r->nq > batch_cap)) break;
            sv.batch.push_back(r);
            r->taken.store(true, std::memory_order_release);
            total += r->nq;
            ix->req_q.pop_front();
        }
        ix->q_len.store((int64_t)ix->req_q.size(), std::memory_order_release);
        lk.unlock();
        const auto t_call = std::chrono::steady_clock::now();
        serve_begin(ix, sv);
        const auto t_begun = std::chrono::steady_clock::now();
        q_lock(lk);
        ix->n_lead_begin_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_begun - t_call).count();
        ix->calls_in_flight++;
        ix->requests_in_flight += (int64_t)sv.batch.size();
        {
            const int64_t total = ix->requests_in_flight + (int64_t)ix->req_q.size();
            if (total >= ix->peak_callers) ix->peak_callers = total;
            else if ((++ix->peak_decay & 31) == 0) ix->peak_callers--;
        }
        ix->leader_active = false;
        ix->q_epoch.fetch_add(1, std::memory_order_release);
        ix->lead_epoch.fetch_add(1, std::memory_order_release);
        // (a waiter whose request is still queued leads the next batch: the spinning ones see the epoch, a sleeping
        // head of the queue is woken - it goes whichever batch comes next, so it is the one worth a wake-up)
        if (!ix->req_q.empty() && ix->req_q.front()->parked) ix->req_q.front()->cv.notify_one();
        lk.unlock();
        const auto t_fin = std::chrono::steady_clock::now();
        serve_finish(ix, sv);
        const auto t_served = std::chrono::steady_clock::now();
        q_lock(lk);
        ix->n_lead_call_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_served - t_call).count();
        ix->n_lead_finish_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(t_served - t_fin).count();
        ix->n_lead_relock_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_served).count();
        {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_call).count();
            ix->call_us_est = ix->call_us_est <= 0.0 ? us : ix->call_us_est + (us - ix->call_us_est) / 8.0;
        }
        ix->calls_in_flight--;
        ix->requests_in_flight -= (int64_t)sv.batch.size();
        for (ls_req* r : sv.batch) {
            // (a sleeping caller cannot leave cv.wait before this thread lets go of the mutex: its request is still
            // there to be notified; for a polling one the store is the last access to *r)
            if (r->parked) {
                r->done.store(true, std::memory_order_release);
                r->cv.notify_one();
            } else {
                r->done.store(true, std::memory_order_release);
            }
        }
        ix->q_epoch.fetch_add(1, std::memory_order_release);
        ix->lead_epoch.fetch_add(1, std::memory_order_release);
        ix->q_cv.notify_all();  // (the next batch's leader, if it sleeps waiting for this call's slot: one thread at most)
        if (!ix->leader_active && !ix->req_q.empty() && ix->req_q.front()->parked) ix->req_q.front()->cv.notify_one();
    }
    if (lk.owns_lock()) lk.unlock();
    if (me.rc != LS_OK && me.err[0]) ls_set_error("%s", me.err);
    return me.rc;
}

}  // extern "C"
