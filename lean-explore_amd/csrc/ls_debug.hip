// ls_debug.hip - kernel timing, tuning hooks and counters (include/leansearch_debug.h): what the tests, the tools under
// tools/ and bench.py read or switch; nothing a search needs.
#include "ls_index.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#ifdef LS_LEAD_TRACE
extern std::atomic<uint64_t> g_lead_trace[8];  // (ls_callers.hip)
#endif

extern "C" {

int ls_set_profiling(ls_index* ix, int32_t enabled) {
    if (!ix) return LS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->group) return ls_group_set_profiling(ix, enabled);
    ix->profiling = enabled != 0;
    ix->prof_n = 0;
    return LS_OK;
}

int ls_last_kernel_ms(ls_index* ix, float* scan_ms, float* total_ms) {
    if (!ix || !scan_ms || !total_ms) return LS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->group) return ls_group_last_kernel_ms(ix, scan_ms, total_ms);
    if (ix->prof_n == 0) {
        ls_set_error("ls_last_kernel_ms: no profiled search recorded");
        return LS_ERR_INVALID_ARG;
    }
    LS_HIP(hipSetDevice(ix->device));
    double a = 0.0, b = 0.0;
    for (size_t i = 0; i < ix->prof_n; ++i) {
        hipEvent_t* pe = &ix->prof_ev[2 * i];
        LS_HIP(hipEventSynchronize(pe[1]));
        float x = 0.f;
        LS_HIP(hipEventElapsedTime(&x, pe[0], pe[1]));
        a += x;
        b += x;
    }
    *scan_ms = (float)(a / (double)ix->prof_n);
    *total_ms = (float)(b / (double)ix->prof_n);
    ix->prof_n = 0;
    return LS_OK;
}

// test / tuning hooks -------------------------------------------------------------------------
int ls_debug_option(ls_index* ix, int32_t which, int32_t value) {
    if (!ix) return LS_ERR_INVALID_ARG;
    if (which == 10) {  // combine concurrent synchronous host searches into shared corpus passes (default on)
        std::lock_guard<std::mutex> ql(ix->q_mu);
        ix->opt_combine = value != 0;
        return LS_OK;
    }
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group) return ls_group_debug_option(ix, which, value);
    if (which == 0) {  // force k' (0 = automatic)
        ix->opt_kprime = value;
        return LS_OK;
    }
    if (which == 13) {  // pipelined fp16 batches (rows <= 768 bytes): a later batch's sample phase rides on the pass launch (default on)
        ix->opt_fused = value != 0;
        return LS_OK;
    }
    if (which == 14) {  // one-wave select kernel (co-resident with a running pass): default on
        ix->opt_wave_select = value != 0;
        return LS_OK;
    }
    if (which == 9) {  // synchronous host calls: selection inside the scan launch of its own query (default on)
        ix->opt_same_launch = value != 0;
        return LS_OK;
    }
    if (which == 7) {  // force the number of scan workgroups per launch (0 = automatic)
        ix->opt_blocks = value;
        return LS_OK;
    }
#ifdef LS_VARIANT_RS2
    if (which == 18) {  // fp16 index, 48-chunk rows: the batched pass in the row-split, 64-queries-per-wave shape (variant builds)
        if (int rc = ls_i_batched_repair(ix)) return rc;  // nothing pending in the other geometry
        ix->g.qg4 = value != 0;
        return LS_OK;
    }
#endif
    if (which == 21) {  // most callers for which a second batch may go early on the other host slot (default 8)
        ix->opt_early_cap = value;
        return LS_OK;
    }
    if (which == 20) {  // concurrent callers are gathered into one pass (0 off; default on)
        ix->opt_gather = value != 0;
        return LS_OK;
    }
    if (which == 22) {  // fp32 index: one ls_mq pass carries up to 32 queries (two MFMA B blocks per A operand; default on)
        ix->opt_mq32 = value != 0;
        return LS_OK;
    }
    if (which == 19) {  // launches whose unproven queries can be served again write no score vectors (default on; 2: not the single-query device launches)
        ix->opt_mq_skip_scores = value != 0;
        ix->opt_scan_skip_scores = value == 1;
        // (with score vectors kept no launch may find them too few while a host call is in flight: all of them now)
        if (!value && ix->d_corpus) return ls_i_grow_score_vectors(ix, LS_QUERIES_PER_LAUNCH_MAX);
        return LS_OK;
    }
    if (which == 23) {  // ls_search: a queue that alone fills a pass is launched at once behind the call in flight (default on)
        ix->opt_full_early = value != 0;
        return LS_OK;
    }
    if (which == 17) {  // synchronous host calls overlap two deep (default on)
        ix->opt_overlap_calls = value != 0;
        return LS_OK;
    }
    if (which == 16) {  // fp32 index: small batches on the f32 matrix cores, 16 queries per pass (ls_mq.hip; default on)
        ix->opt_mq = value != 0;
        return LS_OK;
    }
    if (which == 6) {  // several queries per corpus pass on the scan path (default on)
        ix->opt_multi_query = value != 0;
        return LS_OK;
    }
    if (which == 5) {  // speculative sample threshold on the batched path (default on)
        ix->opt_spec_tau = value != 0;
        return LS_OK;
    }
    if (which == 4) {  // allow the batched MFMA path (default on)
        ix->opt_gemm = value != 0;
        return LS_OK;
    }
    if (which == 3) {  // piggy-back the finalize on the next scan launch (default on)
        ix->opt_overlap = value != 0;
        return LS_OK;
    }
    if (which == 2) {  // alternate the sweep direction of consecutive scans (default off)
        ix->opt_alternate = value != 0;
        return LS_OK;
    }
    if (which == 1) {  // force the finalize kernel's exact slow path
        ix->opt_force_slow = value != 0;
        return LS_OK;
    }
    ls_set_error("ls_debug_option: unknown option %d", which);
    return LS_ERR_INVALID_ARG;
}

int ls_debug_read_scores(ls_index* ix, float* out, int64_t count) {
    if (!ix || !out || count < 0 || count > ix->n || ix->group) return LS_ERR_INVALID_ARG;
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    LS_HIP(hipSetDevice(ix->device));
    LS_HIP(hipDeviceSynchronize());
    LS_HIP(hipMemcpy(out, ix->sets[ix->last_set].d_S, sizeof(float) * (size_t)count,
                     hipMemcpyDeviceToHost));
    return LS_OK;
}

int64_t ls_debug_counter(ls_index* ix, int32_t which) {
#ifdef LS_GEMM_TIMING
    if (ix && which >= 3000 && which < 3000 + 4096) {  // sample-pass phase stamps (ls_gemm.hip)
        static unsigned long long st[4096];
        if ((which - 3000) == 0 && ls_gemm_read_sample_stamps(st, 4096) != 0) return -1;
        return (int64_t)st[which - 3000];
    }
    if (ix && which >= 2000) {  // start / end tick of MFMA-pass workgroup (which - 2000) / 2
        u64 v = 0;
        if (hipMemcpy(&v, reinterpret_cast<const u64*>(ix->bc_sets[ix->bc_last_set].d_sample_top) + (which - 2000),
                      sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        return (int64_t)v;
    }
#endif
#ifdef LS_SCAN_TIMING
    if (ix && which >= 1000) {  // start / end tick of scan workgroup (which - 1000) / 2
        u64 v = 0;
        if (hipMemcpy(&v, reinterpret_cast<const u64*>(ix->sets[ix->last_set].d_S + 7 * ix->s_stride) +
                              (which - 1000), sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        return (int64_t)v;
    }
    if (ix && which >= 10 && which < 18) {  // phase stamps of the last scan launch (ls_scan.hip: 4; ls_mq.hip: 7)
        u64 v = 0;
        if (hipMemcpy(&v, ix->sets[ix->last_set].d_cand + (size_t)ix->max_blocks * LS_KP_MAX - 8 + (which - 10),
                      sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return (int64_t)v;
    }
#endif
#ifdef LS_LEAD_TRACE
    if (which >= 40 && which < 48) return (int64_t)g_lead_trace[which - 40].load(std::memory_order_relaxed);
#endif
    if (!ix || which < 0 || which > 33) return -1;
    if (which == 16 || which == 17) {
        std::lock_guard<std::mutex> ql(ix->q_mu);
        return (int64_t)(which == 16 ? ix->n_combined_batches : ix->n_combined_requests);
    }
    if (which == 33) {  // waiters that went to sleep on their request (more callers than CPUs to poll on, or a long wait)
        std::lock_guard<std::mutex> ql(ix->q_mu);
        return (int64_t)ix->n_waiter_parks;
    }
    if (which >= 28 && which <= 32) {  // the leaders' phase clocks, cumulative ns: waiting + gathering | begin..finish | re-taking the queue's
                                       // mutex | of begin..finish: the enqueue (serve_begin) | the wait for the results and their copy
        std::lock_guard<std::mutex> ql(ix->q_mu);
        return (int64_t)(which == 28 ? ix->n_lead_wait_ns : which == 29 ? ix->n_lead_call_ns : which == 30 ? ix->n_lead_relock_ns :
                         which == 31 ? ix->n_lead_begin_ns : ix->n_lead_finish_ns);
    }
    std::lock_guard<std::mutex> lk(ix->mu);
    if (ix->group) return ls_group_debug_counter(ix, which);
    if (which == 8) return (int64_t)ix->n_batched_fallback;
    if (which == 9) return (int64_t)ix->n_batched_launches;
    if (which == 10) return (int64_t)ix->last_path;
    if (which == 11) return (int64_t)ix->n_launches_total;
    if (which == 12) return (int64_t)ix->n_chunked_calls;
    if (which == 20) return (int64_t)ix->n_same_launch_retries;
    if (which == 22) return (int64_t)ix->n_forced_checks;
    if (which == 23) return (int64_t)ix->n_mq_launches;
    if (which == 24) return (int64_t)ix->n_overlapped_calls;
    if (which == 25) return (int64_t)ix->n_mq_reserved;
    if (which == 26) return (int64_t)ix->n_mq_skipped_repairs;
    if (which == 27) return (int64_t)__atomic_load_n(&ix->n_spin_timeouts, __ATOMIC_RELAXED);
    if (which > 9) return 0;  // 13..15, 18, 19 and 21 are group counters
    if (hipSetDevice(ix->device) != hipSuccess) return -1;
    u32 v = 0;
    if (hipMemcpy(&v, ix->d_counters + which, sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    return (int64_t)v;
}

}  // extern "C"
