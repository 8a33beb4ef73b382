// ls_api.hip — the C ABI of include/leansearch.h: index lifetime, HBM residency, and the
// host-side orchestration of one search (prep -> scan -> finalize per query).
//
// There is deliberately no CPU path in this library: with no HIP device every compute entry
// point returns LS_ERR_NO_DEVICE.
#include "ls_index.h"


#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <condition_variable>
#include <thread>
#include <sched.h>
#include <vector>

static thread_local char g_err[512] = "";

void ls_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ls_i_check_device(int32_t device) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0) {
        ls_set_error("no HIP device available (%s); libleansearch has no CPU path",
                     e != hipSuccess ? hipGetErrorString(e) : "device count 0");
        return LS_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= cnt) {
        ls_set_error("device %d out of range (have %d)", device, cnt);
        return LS_ERR_NO_DEVICE;
    }
    return LS_OK;
}

static int create_common(ls_index** out, int64_t n, int32_t d, int32_t dtype, int32_t device,
                         ls_index** pidx) {
    if (!out) {
        ls_set_error("ls_create: out is null");
        return LS_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (n < 0 || d <= 0) {
        ls_set_error("ls_create: bad shape n=%lld d=%d", (long long)n, d);
        return LS_ERR_INVALID_ARG;
    }
    if (n >= 0xffffffffll) {
        ls_set_error("ls_create: n=%lld exceeds the 2^32-1 rows one shard can index", (long long)n);
        return LS_ERR_INVALID_ARG;
    }
    ls_geom g;
    if (ls_pick_geom(d, dtype, &g) != LS_OK) {
        ls_set_error("ls_create: unsupported d=%d / dtype=%d (max stored row is 4096 bytes)", d,
                     dtype);
        return LS_ERR_INVALID_ARG;
    }
    int rc = ls_i_check_device(device);
    if (rc != LS_OK) return rc;
    LS_HIP(hipSetDevice(device));
    ls_index* ix = new (std::nothrow) ls_index();
    if (!ix) {
        ls_set_error("ls_create: out of host memory");
        return LS_ERR_INVALID_ARG;
    }
    ix->device = device;
    ix->n = n;
    ix->dtype = dtype;
    ix->g = g;
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess &&
        cu > 0)
        ix->n_cu = cu;
    *pidx = ix;
    return LS_OK;
}

// (Re)allocate what depends on the row count: the corpus with its zero pad rows (old rows are
// carried over device-to-device when the index grows) and the scan path's score vectors. Nothing
// of the handle changes unless every allocation succeeded. `amortise`: grow the capacity
// geometrically (index.add in a loop stays linear in the rows added).
static int alloc_rows(ls_index* ix, int64_t new_n, bool amortise) {
    const size_t row_bytes = (size_t)ix->g.chunks * 16;
    if (ix->max_blocks == 0) {
        // ls_scan_blocks() <= 2 workgroups per CU; room for the tuning hook (debug option 7) to
        // force up to 4 per CU
        ix->max_blocks = 4 * ix->n_cu;
        for (auto& st : ix->sets) {  // room for LS_QUERIES_PER_LAUNCH_MAX queries per generation
            LS_HIP(hipMalloc((void**)&st.d_cand,
                             sizeof(u64) * (size_t)ix->max_blocks * LS_KP_MAX * LS_QUERIES_PER_LAUNCH_MAX));
            LS_HIP(hipMalloc((void**)&st.d_bound,
                             sizeof(u64) * (size_t)ix->max_blocks * LS_QUERIES_PER_LAUNCH_MAX));
        }
    }
    if (new_n > ix->cap_rows || !ix->sets[0].d_S) {
        int64_t want = std::max<int64_t>(new_n, 1);
        if (amortise && ix->cap_rows > 0)
            want = std::min<int64_t>(std::max(want, ix->cap_rows + ix->cap_rows / 2), 0xfffffffell);
        void* d_new = nullptr;
        float* S_new[LS_NSETS] = {};
        long long stride = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            stride = (want + 63) / 64 * 64;
            bool ok = new_n == 0 ||
                      hipMalloc(&d_new, (size_t)(want + LS_CORPUS_PAD_ROWS) * row_bytes) == hipSuccess;
            if (ix->s_vecs <= 0) ix->s_vecs = 8;  // (one VALU scan group; grow_score_vectors when a launch needs more)
            for (int i = 0; ok && i < LS_NSETS; ++i)
                ok = hipMalloc((void**)&S_new[i], sizeof(float) * (size_t)stride * ix->s_vecs) == hipSuccess;
            if (ok) break;
            (void)hipGetLastError();
            (void)hipFree(d_new);
            d_new = nullptr;
            for (auto& p : S_new) {
                (void)hipFree(p);
                p = nullptr;
            }
            if (attempt == 1 || want == std::max<int64_t>(new_n, 1)) {
                ls_set_error("out of device memory for %lld rows of %zu bytes", (long long)want,
                             row_bytes);
                return LS_ERR_HIP;
            }
            want = std::max<int64_t>(new_n, 1);  // the amortised size did not fit: exact size
        }
        if (d_new && ix->d_corpus && ix->n > 0 &&
            hipMemcpy(d_new, ix->d_corpus, (size_t)std::min(ix->n, new_n) * row_bytes,
                      hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipFree(d_new);
            for (auto& p : S_new) (void)hipFree(p);
            ls_set_error("carrying the stored rows over failed");
            return LS_ERR_HIP;
        }
        if (ix->d_corpus) (void)hipFree(ix->d_corpus);
        ix->d_corpus = d_new;
        for (int i = 0; i < LS_NSETS; ++i) {
            if (ix->sets[i].d_S) (void)hipFree(ix->sets[i].d_S);
            ix->sets[i].d_S = S_new[i];
        }
        ix->s_stride = stride;
        ix->cap_rows = new_n > 0 ? want : 0;
    }
    // the batched path reads whole tiles: LS_CORPUS_PAD_ROWS zero rows follow row new_n
    if (ix->d_corpus)
        LS_HIP(hipMemset((char*)ix->d_corpus + (size_t)new_n * row_bytes, 0,
                         (size_t)LS_CORPUS_PAD_ROWS * row_bytes));
    return LS_OK;
}

static int alloc_index_buffers(ls_index* ix) {
    LS_HIP(hipStreamCreateWithFlags(&ix->own_stream, hipStreamNonBlocking));
    int rc = alloc_rows(ix, ix->n, false);
    if (rc != LS_OK) return rc;
    LS_HIP(hipMalloc((void**)&ix->d_counters, sizeof(u32) * 8));
    LS_HIP(hipMemset(ix->d_counters, 0, sizeof(u32) * 8));
    return LS_OK;
}

// Host fp32 rows [count, d] -> stored rows [row0, row0 + count) of the HBM corpus.
static int upload_rows(ls_index* ix, int64_t row0, const float* rows, int64_t count) {
    const ls_geom& g = ix->g;
    const int32_t d = g.d;
    char* dst = (char*)ix->d_corpus + (size_t)row0 * g.chunks * 16;
    if (ix->dtype == LS_DTYPE_F32 && g.d_pad == d) {
        // stored layout == caller's layout: one straight copy into HBM
        LS_HIP(hipMemcpy(dst, rows, (size_t)count * d * sizeof(float), hipMemcpyHostToDevice));
        return LS_OK;
    }
    // upload in slabs of rows through a staging buffer, converting on the device
    const int64_t slab = std::max<int64_t>(1, (int64_t)(256ll << 20) / ((int64_t)d * 4));
    float* stage = nullptr;
    LS_HIP(hipMalloc((void**)&stage, (size_t)std::min(slab, count) * d * sizeof(float)));
    int rc = LS_OK;
    for (int64_t r0 = 0; rc == LS_OK && r0 < count; r0 += slab) {
        const int64_t nr = std::min(slab, count - r0);
        if (hipMemcpy(stage, rows + r0 * d, (size_t)nr * d * sizeof(float),
                      hipMemcpyHostToDevice) != hipSuccess) {
            ls_set_error("corpus upload failed");
            rc = LS_ERR_HIP;
            break;
        }
        rc = ls_launch_convert(stage, dst + (size_t)r0 * g.chunks * 16, nr, g, ix->own_stream);
        if (rc == LS_OK && hipStreamSynchronize(ix->own_stream) != hipSuccess) {
            ls_set_error("corpus conversion failed");
            rc = LS_ERR_HIP;
        }
    }
    (void)hipFree(stage);
    return rc;
}

extern "C" {

void ls_destroy(ls_index* ix) {
    if (!ix) return;
    if (ix->group) {
        ls_group_destroy(ix);
        delete ix;
        return;
    }
    (void)hipSetDevice(ix->device);
    if (ix->own_stream) (void)hipStreamSynchronize(ix->own_stream);
    (void)hipFree(ix->d_corpus);
    for (auto& h : ix->hs) {
        (void)hipFree(h.d_qraw);
        (void)hipFree(h.d_out_s);
        (void)hipFree(h.d_out_i);
        if (h.h_q) (void)hipHostFree(h.h_q);
        if (h.h_done) (void)hipHostFree(h.h_done);
        if (h.h_out_g) (void)hipHostFree(h.h_out_g);
        if (h.h_out_s) (void)hipHostFree(h.h_out_s);
        if (h.h_out_i) (void)hipHostFree(h.h_out_i);
        if (h.stream) {
            (void)hipStreamSynchronize(h.stream);
            (void)hipStreamDestroy(h.stream);
        }
    }
    for (auto& st : ix->sets) {
        (void)hipFree(st.d_S);
        (void)hipFree(st.d_cand);
        (void)hipFree(st.d_bound);
        (void)hipFree(st.d_gran);
        (void)hipFree(st.d_qpad);
    }
    (void)hipFree(ix->d_counters);
    for (hipStream_t cs : {ix->chain_main[0], ix->chain_main[1], ix->chain_sel})
        if (cs) (void)hipStreamSynchronize(cs);
    for (auto& st : ix->bc_sets) {
        if (st.ev_prep) (void)hipEventDestroy(st.ev_prep);
        if (st.ev_pass) (void)hipEventDestroy(st.ev_pass);
        if (st.ev_sel) (void)hipEventDestroy(st.ev_sel);
        if (st.done) (void)hipEventDestroy(st.done);
        (void)hipFree(st.d_qh);
        (void)hipFree(st.d_queues);
        (void)hipFree(st.d_counts);
        (void)hipFree(st.d_tau);
        (void)hipFree(st.d_sample_top);
    }
    for (float* b : ix->d_qkeep_blk) (void)hipFree(b);
    (void)hipFree(ix->d_mq_keep);
    (void)hipFree(ix->d_mq_flags);
    if (ix->h_mq_flags) (void)hipHostFree(ix->h_mq_flags);
    (void)hipFree(ix->d_overflow);
    if (ix->h_overflow) (void)hipHostFree(ix->h_overflow);
    for (hipEvent_t e : ix->prof_ev) (void)hipEventDestroy(e);
    for (hipStream_t cs : {ix->chain_main[0], ix->chain_main[1], ix->chain_sel})
        if (cs) (void)hipStreamDestroy(cs);
    if (ix->chain_in) (void)hipEventDestroy(ix->chain_in);
    if (ix->own_stream) (void)hipStreamDestroy(ix->own_stream);
    delete ix;
}

int ls_create(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
              int32_t device) {
    if (n > 0 && !corpus) {
        ls_set_error("ls_create: corpus is null");
        return LS_ERR_INVALID_ARG;
    }
    ls_index* ix = nullptr;
    int rc = create_common(out, n, d, dtype, device, &ix);
    if (rc != LS_OK) return rc;
    rc = alloc_index_buffers(ix);
    if (rc == LS_OK && n > 0) rc = upload_rows(ix, 0, corpus, n);
    if (rc != LS_OK) {
        ls_destroy(ix);
        return rc;
    }
    *out = ix;
    return LS_OK;
}

int ls_create_from_device(ls_index** out, const void* d_corpus, int64_t n, int32_t d,
                          int32_t dtype, int32_t device) {
    if (n > 0 && !d_corpus) {
        ls_set_error("ls_create_from_device: corpus is null");
        return LS_ERR_INVALID_ARG;
    }
    ls_index* ix = nullptr;
    int rc = create_common(out, n, d, dtype, device, &ix);
    if (rc != LS_OK) return rc;
    rc = alloc_index_buffers(ix);
    if (rc == LS_OK && n > 0) {
        rc = ls_launch_convert((const float*)d_corpus, ix->d_corpus, n, ix->g, ix->own_stream);
        if (rc == LS_OK && hipStreamSynchronize(ix->own_stream) != hipSuccess) {
            ls_set_error("corpus conversion failed");
            rc = LS_ERR_HIP;
        }
    }
    if (rc != LS_OK) {
        ls_destroy(ix);
        return rc;
    }
    *out = ix;
    return LS_OK;
}

int64_t ls_ntotal(const ls_index* ix) { return ix ? ix->n : -1; }
int32_t ls_dim(const ls_index* ix) { return ix ? ix->g.d : -1; }
int32_t ls_dtype(const ls_index* ix) { return ix ? ix->dtype : -1; }
int32_t ls_device(const ls_index* ix) { return ix ? ix->device : -1; }

int ls_set_base(ls_index* ix, int64_t base) {
    if (!ix || base < 0) {
        ls_set_error("ls_set_base: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group) return ls_group_set_base(ix, base);
    // held-back pipelined batches launch their select - which adds the base - at the next flush: the
    // batches already submitted must be numbered with the base they were submitted under
    LS_HIP(hipSetDevice(ix->device));
    if (int rc = ls_i_flush_deferred(ix)) return rc;
    // ... and so must the queries a pending repair will serve again (ls_mq / scan launches without score vectors,
    // unchecked batched calls): repair them now, under the old base
    if (int rc = ls_i_flush_pending(ix)) return rc;
    if (int rc = ls_i_batched_repair(ix)) return rc;
    ix->base = base;
    return LS_OK;
}

}  // extern "C"

// choose k' (keys each scan workgroup emits) from lambda = expected top-k rows per workgroup
static int pick_kprime(const ls_index* ix, int blocks, int keff, int kp_max = LS_KP_MAX) {
    if (ix->opt_kprime > 0) return std::min(ix->opt_kprime, kp_max - 1);
    const double lam = (double)keff / (double)blocks;
    int kp = (int)(lam + 5.0 * __builtin_sqrt(lam) + 3.0);
    kp = std::max(kp, 2);
    kp = std::min(kp, kp_max - 1);
    while (kp > 1 && (int64_t)blocks * kp > LS_FINAL_CAP) --kp;
    return kp;
}

static size_t ls_fin_lds_bytes_host(int keys_cap, int keff) {
    int rc = 256;
    while (rc < keff) rc <<= 1;
    return ((size_t)keys_cap + (size_t)rc + 256 + 16) * sizeof(u64) + (8 * 256 + 64) * sizeof(u32);
}

// Launch the pending selection jobs on their own (1024 threads each, LDS for the full
// 8192-key capacity).
int ls_i_flush_pending(ls_index* ix) {
    const int np = ix->n_pending;
    if (np == 0) return LS_OK;
    ix->n_pending = 0;
    ls_fin_batch jobs = ix->pending;
    jobs.njobs = np;
    jobs.p0.keys_cap = LS_FINAL_CAP;
    ix->n_launches_total++;
    return ls_launch_finalize(jobs, ix->pending_stream);  // one launch, one workgroup per job
}

// The score vectors S (what a selection's rescue sweeps) are kept for 8 queries per generation - one VALU scan
// group - until a launch that writes them serves more: ls_mq launches of LS_FLAG_ASYNC-only / LS_FLAG_INORDER
// calls (every other ls_mq launch writes none: skip_scores below). 2 generations x 32 vectors x n floats would be
// 3.2 GB on a 12.5 M-row shard that never uses them. Drains the device: nothing queued may still
// use the old vectors.
static int grow_score_vectors(ls_index* ix, int need) {
    if (need <= ix->s_vecs) return LS_OK;
    const int want = need <= 8 ? 8 : (need <= 16 ? 16 : LS_QUERIES_PER_LAUNCH_MAX);
    if (int rc = ls_i_flush_pending(ix)) return rc;  // (its jobs name the old vectors)
    LS_HIP(hipDeviceSynchronize());
    float* S_new[LS_NSETS] = {};
    for (int i = 0; i < LS_NSETS; ++i) {
        if (hipMalloc((void**)&S_new[i], sizeof(float) * (size_t)ix->s_stride * want) != hipSuccess) {
            (void)hipGetLastError();
            for (auto& p : S_new) (void)hipFree(p);
            ls_set_error("out of device memory for %d score vectors of %lld rows", want, (long long)ix->s_stride);
            return LS_ERR_HIP;
        }
    }
    for (int i = 0; i < LS_NSETS; ++i) {
        (void)hipFree(ix->sets[i].d_S);
        ix->sets[i].d_S = S_new[i];
    }
    ix->s_vecs = want;
    return LS_OK;
}

// ls_mq launch geometry for `nq` queries of one pass (ls_mq.hip): workgroups, k', keys per lane (0: not usable)
struct mq_plan {
    int blocks, kprime, keys;
};
static mq_plan mq_make_plan(const ls_index* ix, int nq, int32_t k) {
    mq_plan p{0, 0, 0};
    if (!(ix->opt_mq && ix->opt_multi_query && ix->dtype == LS_DTYPE_F32 && ix->n >= LS_MQ_MIN_ROWS)) return p;
    const int keff = (int)std::max<int64_t>(std::min<int64_t>(k, ix->n), 1);
    p.blocks = ix->opt_blocks > 0 ? std::min(ix->opt_blocks, ix->max_blocks) : ls_mq_blocks(ix->n, ix->n_cu, nq, ix->g.chunks);
    // (an ls_mq workgroup ranks waves x keys of them and may emit more than the scan kernel's 15: at k = 1000 over
    // 224-241 workgroups - 4.2-4.5 of a query's top-k each on average - the 15-key cap left the proof unprovable for
    // 1.5e-3..3.6e-3 of the queries, each one a second serve of 140 us; 23 keys: 1e-9)
    p.kprime = pick_kprime(ix, p.blocks, keff, LS_MQ_KP_MAX);
    p.keys = ls_mq_lane_keys(p.blocks, keff, nq);
    while (p.keys > 0 && p.keys < 8 && p.kprime + 1 > ls_mq_waves(nq) * p.keys) p.keys = p.keys == 3 ? 5 : 8;  // k' + 1 of waves x keys go out
    if (p.keys > 0) p.kprime = std::min(p.kprime, ls_mq_waves(nq) * p.keys - 1);
    return p;
}
// Queries one ls_mq pass may carry on this index for this k: 32 (two 16-column MFMA blocks per A operand), or
// 0 when the kernel is not usable (fp16 storage, small shards, k too large for the shard).
static int mq_max_queries(const ls_index* ix, int32_t k) {
    if (!ix->opt_mq32) return mq_make_plan(ix, 16, k).keys > 0 ? 16 : 0;
    return mq_make_plan(ix, LS_QUERIES_PER_LAUNCH_MAX, k).keys > 0 ? LS_QUERIES_PER_LAUNCH_MAX
                                                                    : (mq_make_plan(ix, 16, k).keys > 0 ? 16 : 0);
}
// The most queries a call may bring and still be served by the exact scan path (ls_search's combining queue, the
// polling host call): one ls_mq pass where that kernel serves the index, else LS_SCAN_PATH_MAX_NQ.
static int scan_path_max_nq(const ls_index* ix, int32_t k) {
    return std::max(LS_SCAN_PATH_MAX_NQ, mq_max_queries(ix, k));
}

// Launches (query groups) the scan path will use for a call of nq queries: ONE definition, shared by the
// scheduling below and by the host API's decision to overlap a call (a call that owns one scratch generation
// must be a single group).
static int scan_group_size(const ls_index* ix, int64_t left, int mq_max) {
    if (mq_max > 0 && left >= 2) return (int)std::min<int64_t>(left, mq_max);
    return !ix->opt_multi_query ? 1 : (left >= 5 ? (int)std::min<int64_t>(left, 8) : (left >= 2 ? (int)std::min<int64_t>(left, 4) : 1));
}
static int64_t scan_group_count(const ls_index* ix, int64_t nq, int32_t k) {
    const int mq_max = mq_max_queries(ix, k);
    int64_t groups = 0;
    for (int64_t left = nq; left > 0; ++groups) left -= scan_group_size(ix, left, mq_max);
    return groups;
}

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
#define LS_MQ_KEEP_SLOTS 256  // ls_mq launches without score vectors between two repairs (device-output calls)
static int mq_repair(ls_index* ix);

// Queue one search on stream `s` through the scan path. d_q: device fp32 [nq, d]; outputs
// device [nq, k].
static int scan_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k, uint32_t flags,
                            float* d_out_s, int64_t* d_out_i, hipStream_t s) {
    const ls_geom& g = ix->g;
    int rc = LS_OK;
    const bool normalize = (flags & LS_FLAG_NORMALIZE) != 0;
    const int64_t keff = std::min<int64_t>(k, ix->n);
    const int scan_blocks = ix->opt_blocks > 0 ? std::min(ix->opt_blocks, ix->max_blocks)
                                               : ls_scan_blocks(ix->n > 0 ? ix->n : 1, g, ix->n_cu);
    const int scan_kprime = pick_kprime(ix, scan_blocks, (int)std::max<int64_t>(keff, 1));
    // fp32 index, two or more queries left: up to 32 of them share one pass on the f32 matrix cores
    // (ls_mq.hip; same bits as the scan kernel, so a query's results do not depend on its company)
    const int mq_max = mq_max_queries(ix, k);
    // scratch generations this call will use: a same-launch job's retry reads its generation's S and
    // granules after the host has seen its answer, so such a call must not wrap around the LS_NSETS
    // generations (one query per launch - debug option 6 - and 3+ queries would: ADVICE r4)
    const int64_t n_groups = scan_group_count(ix, nq, k);
    if (ix->force_gen >= 0 && n_groups != 1) {  // (host_call_begin asks the same function)
        ls_set_error("internal: an overlapped host call must be a single query group (%lld)", (long long)n_groups);
        return LS_ERR_INVALID_ARG;
    }
    // Everything is queued on the caller's stream. Queries go out in groups of 8, 4 or 1 that
    // share one pass over the corpus:
    //     launch i = { scan(group i)  +  one extra workgroup per query of group i-1: finalize }
    // so the selection step costs neither a launch nor a kernel boundary and hides under the
    // next scan. Two scratch generations alternate (group i-1's finalize reads generation A
    // while group i's scan fills B). The last group's finalizes are "pending": they ride on the
    // next call's first scan (LS_FLAG_PIPELINE) or are launched on their own right away.
    const bool pipeline = (flags & LS_FLAG_PIPELINE) != 0;
    // A scratch generation is ordered by stream order only: a launch that uses it on another stream than
    // its previous user first waits (on the host) for that stream (per generation, below).
    if (ix->n_pending && (ix->pending_stream != s || !ix->opt_overlap)) {
        hipStream_t old = ix->pending_stream;
        rc = ls_i_flush_pending(ix);
        if (rc != LS_OK) return rc;
        // a different stream takes over: do not let its scans race the old stream's finalize
        if (old != s) LS_HIP(hipStreamSynchronize(old));
    }
    for (int64_t q0 = 0; q0 < nq;) {
        const int64_t left = nq - q0;
        // queries per launch: 8 or 4 with the last real query repeated as padding, or 1
        const bool use_mq = mq_max > 0 && left >= 2;
        const int gsz = scan_group_size(ix, left, mq_max);
        const int NQ = use_mq ? gsz : (!ix->opt_multi_query ? 1 : (left >= 5 ? 8 : (left >= 2 ? 4 : 1)));
        const int real = (int)std::min<int64_t>(NQ, left);
        if ((use_mq || NQ == 1) && ix->dev_call_repairable && !ix->reserving &&
            ((int)ix->mq_pend.size() >= LS_MQ_KEEP_SLOTS || (ix->d_mq_keep && ix->mq_keep_d != g.d))) {
            rc = mq_repair(ix);  // the ring of kept queries is full: make what is pending final first
            if (rc != LS_OK) return rc;
        }
        // (an ls_mq launch's geometry depends on its query count: 17..32 queries run the two-block kernel, one
        // workgroup per CU with the selection workgroups' CUs left free)
        const mq_plan mp = use_mq ? mq_make_plan(ix, NQ, k) : mq_plan{0, 0, 0};
        const int mq_keys = mp.keys;
        const int blocks = use_mq ? mp.blocks : scan_blocks;
        const int kprime = use_mq ? mp.kprime : scan_kprime;
        const bool prof = ix->profiling && ix->prof_n < LS_PROF_MAX;
        hipEvent_t* pe = nullptr;
        if (prof) {
            while (ix->prof_ev.size() < 2 * (ix->prof_n + 1)) {
                hipEvent_t e;
                LS_HIP(hipEventCreate(&e));
                ix->prof_ev.push_back(e);
            }
            pe = &ix->prof_ev[2 * ix->prof_n];
        }
        const int gen = ix->force_gen >= 0 ? ix->force_gen : (int)(ix->set_rr++ % LS_NSETS);
        ls_index::scratch_set& st = ix->sets[gen];
        if (st.last_stream && st.last_stream != s) LS_HIP(hipStreamSynchronize(st.last_stream));
        st.last_stream = s;
        if (ix->n_pending && ix->n <= 0) {  // an empty index launches no scan to ride on
            rc = ls_i_flush_pending(ix);
            if (rc != LS_OK) return rc;
        }
        ls_scan_args a{};
        a.nfin = 0;
        // Synchronous host-API calls (ls_search: the reference's call, search/engine.py:250): the
        // selection jobs of THIS group ride on its own scan launch and wait inside the kernel for
        // the scan workgroups' keys - one launch per group instead of scan + selection, no kernel
        // boundary and no second launch latency in front of the selection. The hand-off is the
        // data itself (tagged write-through granules, no drain, no counter: ls_fin_params::gran); a job
        // whose emitted keys cannot be proven complete answers LS_DONE_RETRY in its completion word
        // and the host launches the stand-alone finalize behind the scan (host_search_locked).
        // Only on the library's own stream and only with completion words to answer through: a
        // caller's stream may be CU-masked, and device-output calls have no way to ask for a retry.
        const int own_keys_cap = std::max(256, blocks * kprime);
        const bool same_launch =
            !pipeline && ix->n > 0 && ix->opt_same_launch != 0 && !ix->reserving && ix->done_base != nullptr &&
            n_groups <= LS_NSETS &&
            (s == ix->own_stream || s == ix->hs[0].stream || s == ix->hs[1].stream) &&  // (the library's own streams)
            keff <= 256 &&  // (k > 256 orders its result on 1024 threads: own launch)
            (int64_t)blocks * (kprime + 1) <= LS_GRAN_MAX &&
            ls_fin_lds_bytes_host(own_keys_cap, (int)std::max<int64_t>(keff, 1)) <= LS_PIGGY_LDS_MAX;
        // An ls_mq launch whose selection jobs ride along (synchronous host calls) writes NO score vectors: such
        // a job never reads S inside the launch anyway (it answers LS_DONE_RETRY), and its retry serves the
        // query again, alone, on the scan kernel - the same bits (host_call_finish). The 16 x n x 4 bytes of
        // stores cost 3 us of 57 (d = 384) and 4..29 us of 140..167 (d = 1024, 2..16 queries) at N = 200 k.
        // (k > 256: the selection has its own launch behind the pass; without S it answers the same way)
        const bool host_words = !pipeline && ix->done_base != nullptr && ix->cur_retry != nullptr && !ix->reserving;
        // (device-output calls whose results ls_check may still repair: mq_repair; never a repair's own launch)
        const bool dev_keep = (use_mq || (NQ == 1 && ix->opt_scan_skip_scores)) && ix->dev_call_repairable &&
                              !ix->reserving && ix->done_base == nullptr && ix->opt_mq_skip_scores && ix->n > 0;
        // (single queries of synchronous host calls too - the reference's call: 0.5-1 us of 47 / 122)
        const bool skip_scores = ((same_launch || host_words) && ix->opt_mq_skip_scores) || dev_keep;
        if (!skip_scores && NQ > ix->s_vecs) {  // a wide launch that keeps its score vectors (rare: grow_score_vectors)
            if ((rc = grow_score_vectors(ix, NQ)) != LS_OK) return rc;  // (flushes the pending jobs, drains the device)
        }
        if (ix->n_pending && same_launch) {  // left by an earlier pipelined call: its own launch
            rc = ls_i_flush_pending(ix);
            if (rc != LS_OK) return rc;
        }
        if (ix->n_pending) {
            const int keff_p = (int)std::min<int64_t>(ix->pending.p0.k, ix->n);
            if (ls_fin_lds_bytes_host(ix->pending.p0.keys_cap, keff_p) <= LS_PIGGY_LDS_MAX) {
                a.nfin = ix->n_pending;
                a.fin = ix->pending;
                a.fin.njobs = ix->n_pending;
            } else {
                rc = ls_i_flush_pending(ix);
                if (rc != LS_OK) return rc;
            }
        }
        // padded query slots re-read the last real query (their results are never finalised)
        const float* qsrc = d_q + q0 * g.d;
        if (real < NQ) {  // (never for an ls_mq launch: it takes the real count)
            rc = ls_grow(&st.d_qpad, &st.qpad_cap, (size_t)LS_QUERIES_PER_LAUNCH_MAX * g.d);
            if (rc != LS_OK) return rc;
            for (int i = 0; i < NQ; ++i)
                LS_HIP(hipMemcpyAsync(st.d_qpad + (size_t)i * g.d,
                                      d_q + (q0 + std::min(i, real - 1)) * g.d,
                                      sizeof(float) * g.d, hipMemcpyDefault, s));
            qsrc = st.d_qpad;
        }
        a.d_q = qsrc;
        a.nq = NQ;
        a.normalize = normalize;
        a.reverse = ix->opt_alternate && (ix->sweep_count++ & 1);
        int keep_slot = -1;
        if (dev_keep) {
            if (!ix->d_mq_keep || ix->mq_keep_d != g.d) {
                if (ix->d_mq_keep) (void)hipFree(ix->d_mq_keep);
                ix->d_mq_keep = nullptr;
                LS_HIP(hipMalloc((void**)&ix->d_mq_keep, sizeof(float) * (size_t)LS_MQ_KEEP_SLOTS * LS_QUERIES_PER_LAUNCH_MAX * g.d));
                ix->mq_keep_d = g.d;
            }
            if (!ix->d_mq_flags) {
                const size_t fb = sizeof(u32) * (size_t)LS_MQ_KEEP_SLOTS * LS_QUERIES_PER_LAUNCH_MAX;
                LS_HIP(hipMalloc((void**)&ix->d_mq_flags, fb));
                LS_HIP(hipMemset(ix->d_mq_flags, 0, fb));
                LS_HIP(hipHostMalloc((void**)&ix->h_mq_flags, fb + sizeof(u32), hipHostMallocDefault));
                ix->h_mq_flags[LS_MQ_KEEP_SLOTS * LS_QUERIES_PER_LAUNCH_MAX] = 0u;
            }
            keep_slot = (int)ix->mq_pend.size();
            ix->mq_pend.push_back({keep_slot, real, k, flags, d_out_s + q0 * k, d_out_i + q0 * k, s});
        }
        a.d_S = skip_scores ? nullptr : st.d_S;
        a.s_stride = ix->s_stride;
        a.d_cand = st.d_cand;
        a.c_stride = (long long)ix->max_blocks * LS_KP_MAX;
        a.d_bound = st.d_bound;
        a.b_stride = ix->max_blocks;
        a.blocks = blocks;
        a.kprime = kprime;
        a.mq_keys = mq_keys;
        a.d_qkeep = keep_slot >= 0 ? ix->d_mq_keep + (size_t)keep_slot * LS_QUERIES_PER_LAUNCH_MAX * g.d : nullptr;
        ls_fin_batch& jobs = same_launch ? a.fin : ix->pending;
        if (same_launch) {
            if (!st.d_gran) {  // zeroed once: no granule of a later launch carries tag 0
                const size_t bytes = (size_t)LS_QUERIES_PER_LAUNCH_MAX * LS_GRAN_MAX * 16;
                LS_HIP(hipMalloc(&st.d_gran, bytes));
                LS_HIP(hipMemsetAsync(st.d_gran, 0, bytes, s));
            }
            if (++ix->gran_tag == 0) ix->gran_tag = 1;
            a.nfin = real;
            a.d_gran = st.d_gran;
            a.g_stride = LS_GRAN_MAX;
            a.tag = ix->gran_tag;
        }
        {
            // the group's jobs differ by whole-query displacements only: job 0 + strides (ls_fin_batch)
            ls_fin_params& p = jobs.p0;
            p.S = skip_scores ? nullptr : st.d_S;
            p.n = ix->n;
            p.cand = st.d_cand;
            p.bound = st.d_bound;
            p.blocks = blocks;
            p.kprime = kprime;
            p.k = k;
            p.keys_cap = own_keys_cap;
            p.force_slow = ix->opt_force_slow;
            p.base = ix->base;
            p.out_scores = d_out_s + q0 * k;
            p.out_indices = (long long*)(d_out_i + q0 * k);
            p.counters = ix->d_counters;
            p.done = ix->done_base ? ix->done_base + q0 : nullptr;
            p.out_gran = ix->gran_out_base && k <= LS_OUT_GRAN_MAX_K ? ix->gran_out_base + (size_t)q0 * k : nullptr;
            p.done_val = ix->cur_done_seq;
            p.gran = same_launch ? st.d_gran : nullptr;
            p.tag = same_launch ? ix->gran_tag : 0u;
            p.wait = same_launch ? 1u : 0u;
            p.repair = keep_slot >= 0 ? ix->d_mq_flags + (size_t)keep_slot * LS_QUERIES_PER_LAUNCH_MAX : nullptr;
            p.repair_any = keep_slot >= 0 ? ix->h_mq_flags + LS_MQ_KEEP_SLOTS * LS_QUERIES_PER_LAUNCH_MAX : nullptr;
            jobs.S_stride = a.s_stride;
            jobs.cand_stride = a.c_stride;
            jobs.bound_stride = a.b_stride;
            jobs.gran_stride = (long long)LS_GRAN_MAX * 16;
            jobs.njobs = real;
            for (int i = 0; i < LS_QUERIES_PER_LAUNCH_MAX; ++i) jobs.idx[i] = (unsigned char)i;
            if ((same_launch || skip_scores) && ix->cur_retry) ix->cur_retry->push_back(jobs);  // kept until the host has seen the answers
        }
        if (prof) LS_HIP(hipEventRecord(pe[0], s));
        rc = use_mq ? ls_launch_mq(ix->d_corpus, ix->n, g, a, s) : ls_launch_scan(ix->d_corpus, ix->n, g, a, s);
        if (rc != LS_OK) return rc;
        ix->n_launches_total++;
        if (use_mq) ix->n_mq_launches++;
        if (prof) {
            LS_HIP(hipEventRecord(pe[1], s));
            ix->prof_n++;
        }
        if (!same_launch) {
            ix->n_pending = real;
            ix->pending_stream = s;
        }
        ix->last_set = gen;
        q0 += real;
    }
    if (!pipeline || !ix->opt_overlap) return ls_i_flush_pending(ix);
    return LS_OK;
}

// ---- batched MFMA path (ls_gemm.hip) ---------------------------------------------------------------
bool ls_i_batched_eligible(const ls_index* ix, int64_t nq, int32_t k) {
    if (!ix->opt_gemm || k > LS_GEMM_MAX_K) return false;
    const bool big = ix->n >= LS_GEMM_MIN_ROWS ||
                     (ix->n >= LS_GEMM_MIN_ROWS_BIGNQ && nq >= LS_GEMM_BIGNQ);
    if (ix->dtype == LS_DTYPE_F16)
        return nq > LS_SCAN_PATH_MAX_NQ && ix->g.chunks <= LS_GEMM_MAX_CHUNKS && big;
    // fp32: exact f32 MFMA (ls_gemm32.hip), any row length - for batches past what ONE exact ls_mq pass carries
    // (32 queries; 16 with option 22 = 0), or from LS_GEMM32_MIN_NQ on where ls_mq does not serve this (index, k)
    const int one_pass = mq_max_queries(ix, k);
    return big && (one_pass > 0 ? nq > std::max(one_pass, LS_GEMM32_MIN_NQ - 1) : nq >= LS_GEMM32_MIN_NQ);
}

// ls_mq launches of device-output calls that wrote no score vectors: every selection job that could not
// prove its keys complete raised its word in d_mq_flags; those queries are served again here, alone, on the
// scan kernel (the same bits), from the launch's own copy of the raw queries, into the call's output rows.
// Synchronises the streams involved. No-op when nothing is pending.
static int mq_repair(ls_index* ix) {
    if (ix->mq_pend.empty()) return LS_OK;
    if (int rc = ls_i_flush_pending(ix)) return rc;  // the youngest launch's selection jobs
    // (the pending list stays in the handle until nothing below can fail before the repairs themselves: an early
    // return leaves it - and the raised flag word - for the next check)
    hipStream_t last = ix->mq_pend.back().stream;
    for (const auto& pc : ix->mq_pend)
        if (pc.stream != last) LS_HIP(hipStreamSynchronize(pc.stream));
    if (ix->pending_stream && ix->pending_stream != last) LS_HIP(hipStreamSynchronize(ix->pending_stream));
    const size_t used = ix->mq_pend.size() * LS_QUERIES_PER_LAUNCH_MAX;
    LS_HIP(hipStreamSynchronize(last));
    u32* any_word = ix->h_mq_flags + LS_MQ_KEEP_SLOTS * LS_QUERIES_PER_LAUNCH_MAX;
    if (__atomic_load_n(any_word, __ATOMIC_ACQUIRE) == 0u) {  // the common case: no copy, nothing to read
        ix->mq_pend.clear();
        return LS_OK;
    }
    LS_HIP(hipMemcpyAsync(ix->h_mq_flags, ix->d_mq_flags, sizeof(u32) * used, hipMemcpyDeviceToHost, last));
    LS_HIP(hipStreamSynchronize(last));
    __atomic_store_n(any_word, 0u, __ATOMIC_RELEASE);
    std::vector<ls_index::mq_pending_call> pend;
    pend.swap(ix->mq_pend);
    bool any = false;
    const bool was = ix->reserving, rep = ix->dev_call_repairable;
    ix->reserving = true;  // (its launches keep their score vectors and are final when they return)
    ix->dev_call_repairable = false;
    // (a repair never answers through a host call's completion words, whatever context it runs in)
    u32* const sv_done = ix->done_base;
    auto* const sv_gran = ix->gran_out_base;
    auto* const sv_retry = ix->cur_retry;
    const int sv_gen = ix->force_gen;
    ix->done_base = nullptr;
    ix->gran_out_base = nullptr;
    ix->cur_retry = nullptr;
    ix->force_gen = -1;
    int rc = LS_OK;
    for (size_t pi = 0; pi < pend.size() && rc == LS_OK; ++pi) {
        const auto& pc = pend[pi];
        const u32* fl = ix->h_mq_flags + (size_t)pc.slot * LS_QUERIES_PER_LAUNCH_MAX;
        const float* qk = ix->d_mq_keep + (size_t)pc.slot * LS_QUERIES_PER_LAUNCH_MAX * ix->mq_keep_d;
        for (int q = 0; q < pc.nq && rc == LS_OK; ++q) {
            if (!fl[q]) continue;
            any = true;
            //  the caller may have handed these output rows to a LATER pipelined call before this
            // check (a ring of output buffers shorter than the calls between two checks): the later call's results
            // own them now - a repair written there would replace a newer answer with an older one
            const float* r0 = pc.d_out_s + (size_t)q * pc.k;
            bool reused = false;
            for (size_t pj = pi + 1; pj < pend.size() && !reused; ++pj)
                reused = r0 < pend[pj].d_out_s + (size_t)pend[pj].nq * pend[pj].k && pend[pj].d_out_s < r0 + pc.k;
            if (reused) {
                ix->n_mq_skipped_repairs++;
                continue;
            }
            ix->n_mq_reserved++;
            rc = scan_search_on_stream(ix, qk + (size_t)q * ix->mq_keep_d, 1, pc.k, pc.flags & LS_FLAG_NORMALIZE,
                                       pc.d_out_s + (size_t)q * pc.k, pc.d_out_i + (size_t)q * pc.k, last);
        }
        if (rc != LS_OK) break;
    }
    ix->reserving = was;
    ix->dev_call_repairable = rep;
    ix->done_base = sv_done;
    ix->gran_out_base = sv_gran;
    ix->cur_retry = sv_retry;
    ix->force_gen = sv_gen;
    if (rc != LS_OK) {
        //  a failed repair must not look like a finished one: the pending launches and the raised flag
        // word come back (the device flags were not cleared), so the next ls_check tries again or fails again
        pend.insert(pend.end(), ix->mq_pend.begin(), ix->mq_pend.end());
        ix->mq_pend.swap(pend);
        __atomic_store_n(any_word, 1u, __ATOMIC_RELEASE);
        return rc;
    }
    if (any) {
        LS_HIP(hipMemsetAsync(ix->d_mq_flags, 0, sizeof(u32) * used, last));
        LS_HIP(hipStreamSynchronize(last));
    }
    return LS_OK;
}

// Re-run the queries of the pending batched calls whose candidate queues overflowed (or were
// short) through the exact per-query scan path, from the calls' OWN query copies. Synchronises.
int ls_i_batched_repair(ls_index* ix) {
    if (int rc = mq_repair(ix)) return rc;
    if (int rc = ls_i_flush_deferred(ix)) return rc;
    if (ix->bc_pending.empty()) return LS_OK;
    std::vector<ls_index::batched_call> pend;
    pend.swap(ix->bc_pending);
    hipStream_t s = pend.back().stream;
    for (const auto& bc : pend)  // all slots are read after the youngest call has drained
        if (bc.stream != s) LS_HIP(hipStreamSynchronize(bc.stream));
    LS_HIP(hipMemcpyAsync(ix->h_overflow, ix->d_overflow,  // (slots are handed out in order: the used prefix)
                          sizeof(u32) * (size_t)ix->bc_slot_stride * pend.size(), hipMemcpyDeviceToHost, s));
    LS_HIP(hipStreamSynchronize(s));
    bool any = false;
    for (const auto& bc : pend) {
        const u32* fl = ix->h_overflow + (size_t)bc.slot * ix->bc_slot_stride;
        const float* qk = ix->d_qkeep_blk[bc.slot / LS_BC_QKEEP_BLOCK] +
                          (size_t)(bc.slot % LS_BC_QKEEP_BLOCK) * ix->bc_qkeep_stride;
        for (int64_t q = 0; q < bc.nq;) {
            if (!fl[q] || LS_ABL_NOREPAIR) {
                ++q;
                continue;
            }
            // a run of flagged neighbours shares corpus passes (the scan path's 8- / 4-query groups)
            int64_t run = 1;
            while (q + run < bc.nq && fl[q + run]) ++run;
            ix->n_batched_fallback += (uint64_t)run;
            any = true;
            int rc = scan_search_on_stream(ix, qk + q * ix->g.d, run, bc.k,
                                           bc.flags & LS_FLAG_NORMALIZE, bc.d_out_s + q * bc.k,
                                           bc.d_out_i + q * bc.k, s);
            if (rc != LS_OK) return rc;
            q += run;
        }
    }
    if (any) LS_HIP(hipStreamSynchronize(s));
    return LS_OK;
}

// Geometry of one batched call: corpus slices, sample thinning, speculative rank, expected passes.
struct bc_plan {
    int QT, TM;               // queries per workgroup, corpus rows per LDS tile
    int64_t nq_pad, rps;      // padded queries, rows per slice
    int nsplits, sample_stride, jrank, keys_need;
    double per_queue;         // expected entries of one private candidate queue (capacity LS_GEMM_QCAP)
};
static bc_plan bc_make_plan(const ls_index* ix, int64_t nq, int32_t k) {
    const ls_geom& g = ix->g;
    const bool f32 = ix->dtype == LS_DTYPE_F32;
    const int QT = f32 ? 64 : ls_gemm_qt(g);  // queries per workgroup
    const int TM = f32 ? 64 : ls_gemm_tile_rows(g);  // rows a slice contributes to one tile
    const int64_t nq_pad = (nq + QT - 1) / QT * QT;
    const int nqt = (int)(nq_pad / QT);
    // corpus slices: one 8-wave workgroup per CU in total, a multiple of the 8 XCDs. The fp32
    // kernel is light on registers (two workgroups share a CU) and each workgroup walks two slices.
    int nsplits = (LS_GEMM_WG_PER_CU * ix->n_cu / nqt) / 8 * 8;
    nsplits = std::max(8, std::min(nsplits, 256));
    if (f32) nsplits = 2 * std::max(8, std::min(LS_GEMM_MAX_SPLITS / 2, (2 * ix->n_cu / nqt) / 8 * 8));
    else nsplits *= ls_gemm_rs(g);  // the row-split shape: every workgroup walks two slices
    int64_t rps = (ix->n + nsplits - 1) / nsplits;
    rps = (rps + TM - 1) / TM * TM;
    const int tiles_per_split = (int)(rps / TM);
    // the sample is a fixed FRACTION of the corpus (~1/24 of every slice, at least
    // LS_GEMM_SAMPLE_ROWS rows; ~1/48 for slices of more than 1536 tiles, where the sample pass
    // itself is what costs): the expected number of rows passing tau, ~j*N/M0, then does not
    // grow with N
    // The fraction thins out on long slices (about 16 tiles per slice up to 1/96 of the rows:
    // config 4's 12.5 M-row shard spent 7 % of its batch in a 1/24 sample): a thinner sample
    // passes more rows per query (~j*N/M0 +- that over sqrt(j)), which the select kernel's key
    // buffer must hold; the fraction is halved until the +5 sigma count fits LS_BSEL_MAX_KEYS.
    int frac = std::max(24, std::min(96, tiles_per_split / 16));
    int sample_tiles, sample_stride, jrank, keys_need;
    double pass_mean = 0.0;
    for (;; frac /= 2) {
        sample_tiles = std::max(std::max(1, LS_GEMM_SAMPLE_ROWS / TM),
                                (tiles_per_split + frac - 1) / frac);
        sample_stride = std::max(1, (tiles_per_split + sample_tiles - 1) / sample_tiles);
        // Speculative threshold. The k-th best SAMPLE score is a certified lower bound of the
        // final k-th best but passes ~k*N/M0 rows per query. The j-th best sample score (j < k)
        // passes only ~j*N/M0 rows; it is not certified, so the select kernel verifies that at
        // least k rows passed and flags the query for the exact scan path otherwise. j is the
        // smallest rank whose expected pass count exceeds k by 4.5 standard deviations (relative
        // sd of an order statistic ~ 1/sqrt(j)): a flag is a ~1e-5 event per query on
        // exchangeable rows.
        jrank = k;
        const int visited = (tiles_per_split + sample_stride - 1) / sample_stride;
        const double m0 = (double)nsplits * visited * TM;
        const double r = (double)ix->n / std::max(1.0, m0);
        if (ix->opt_spec_tau) {
            for (int j = 1; j <= k; ++j) {
                if ((double)j * r * (1.0 - 4.5 / __builtin_sqrt((double)j)) >= (double)k) {
                    jrank = j;
                    break;
                }
            }
        }
        pass_mean = (double)jrank * r;
        const double expect = pass_mean * (1.0 + 5.0 / __builtin_sqrt((double)jrank));
        keys_need = (int)std::min(expect, 1e9);
        if (keys_need <= LS_BSEL_MAX_KEYS || frac <= 24) break;
    }
    bc_plan p;
    p.QT = QT;
    p.TM = TM;
    p.nq_pad = nq_pad;
    p.rps = rps;
    p.nsplits = nsplits;
    p.sample_stride = sample_stride;
    p.jrank = jrank;
    p.keys_need = keys_need;
    p.per_queue = pass_mean / ((double)nsplits * 4.0);
    return p;
}

// The candidate queues are private per (query, slice, quarter) and hold LS_GEMM_QCAP entries: a
// batch fits when a queue's expected length leaves 5 sigma of Poisson headroom. nsplits shrinks
// as n_cu / query tiles, so big batches at big k (nq = 4096, k = 1000: 16 slices) would overflow
// every queue and send every query to the repair path; such calls are cut into sub-batches.
static bool bc_plan_fits(const bc_plan& p) {
    return p.per_queue + 5.0 * __builtin_sqrt(p.per_queue) <= (double)LS_GEMM_QCAP &&
           p.keys_need <= LS_BSEL_MAX_KEYS;
}
// Largest sub-batch (a multiple of the query tile) whose plan fits; 0 = not even one tile does.
static int64_t bc_chunk(const ls_index* ix, int64_t nq, int32_t k) {
    bc_plan p = bc_make_plan(ix, nq, k);
    if (bc_plan_fits(p)) return nq;
    for (int64_t tiles = p.nq_pad / p.QT / 2; tiles >= 1; tiles /= 2) {
        p = bc_make_plan(ix, tiles * p.QT, k);
        if (bc_plan_fits(p)) return tiles * p.QT;
    }
    return 0;
}

// ---- one batched call ------------------------------------------------------------------------------
// Kernels of a batch: query prep -> sample pass -> tau -> MFMA pass -> select. Plain calls queue all
// of it on the caller's stream.
//
// LS_FLAG_PIPELINE calls go through the handle's CHAIN (round 4): four streams, four scratch sets.
//   prep stream   : prep(i)        behind the caller's stream (chain_in) and behind the previous readers
//                                  of the scratch set's prepared queries; a one-wave, 37-register
//                                  kernel that runs INSIDE whatever pass is resident
//   two main lanes: lane i % 2 runs  ... L(i-2) -> tau(i) -> L(i) -> tau(i+2) -> L(i+2) ...
//                                  where L(i) = ONE launch: the MFMA pass of batch i, then the sample
//                                  phase of batch i+2 (LS_GEMM_FUSED). Nothing orders the two lanes:
//                                  the workgroups of L(i+1) move onto the CUs as those of L(i) retire,
//                                  so the kernel boundaries around a pass (5 us behind it, 7 us in
//                                  front), the tau kernel and the lane's event waits all hide under
//                                  the OTHER lane's pass. For that, batch i's pass is queued when call
//                                  i+2 arrives (or at the next flush: ls_check, a call of another shape,
//                                  ls_export_flags, ls_add ...): results of pipelined calls are
//                                  defined to be valid after ls_check anyway.
//   select stream : select(i)      behind L(i): the one-wave select kernel (<= 48 VGPRs, ls_wsel.hip)
//                                  runs INSIDE the passes that follow, in the registers and LDS a
//                                  pass leaves free, instead of between two passes
// The event the select stream waits for is attached to the pass's dispatch (hipExtLaunchKernelGGL):
// no extra packet behind a pass. Measured on the way here (tools/c3_timeline.sh, tools/
// coresidency_probe.hip, profiles/ab/r04_c3_chain.txt): a cross-stream dependency takes ~15-20 us to
// resolve on this runtime, a wait packet in front of a pass ~8 us, the boundaries around a pass 5 + 7
// us - with ONE main stream those sat between two passes (36 us per batch), hence two lanes. Round
// 3 rotated whole batches over two "lanes" as well, but each lane then ran its own sample pass - the
// same 8-wave, 96 KB kernel, which could only start on CUs the other lane's pass had left (~25 us
// per batch).
static int bc_launch_select(ls_index* ix, const ls_index::bc_stage& b, hipStream_t ss) {
    ls_index::bc_set& st = ix->bc_sets[b.set_id];
    ls_gemm_bufs bufs;
    bufs.d_queues = st.d_queues;
    bufs.d_counts = st.d_counts;
    bufs.d_overflow = b.d_flags;
    bufs.d_sample_top = st.d_sample_top;
    if (ix->opt_wave_select && !b.f32 && ls_wave_select_ok(b.nsplits, b.k, b.keys_need))
        return ls_launch_wave_select(bufs, b.nsplits, b.nq, b.k, ix->base, ix->n, b.rps, b.d_out_s,
                                     b.d_out_i, ss, nullptr);
    return ls_launch_batch_select(bufs, b.nsplits, b.nq, b.k, b.keys_need, ix->base, ix->n, b.rps,
                                  b.d_out_s, b.d_out_i, ss);
}

// Queue the MFMA pass and the select of batch `b` (its sample pass and tau are already queued on
// `sm`). `next` non-null: the pass launch also runs the sample phase of that batch (same plan).
static int bc_launch_pass_select(ls_index* ix, const ls_index::bc_stage& b, const ls_index::bc_stage* next,
                                 hipStream_t sm, hipStream_t ss, bool tau_next = false) {
    ls_index::bc_set& st = ix->bc_sets[b.set_id];
    const ls_geom& g = ix->g;
    int rc;
    ls_gemm_bufs bufs;
    bufs.d_queues = st.d_queues;
    bufs.d_counts = st.d_counts;
    bufs.d_overflow = b.d_flags;
    bufs.d_sample_top = st.d_sample_top;
    const bool prof = ix->profiling && ix->prof_n < LS_PROF_MAX;
    hipEvent_t* pe = nullptr;
    if (prof) {
        while (ix->prof_ev.size() < 2 * (ix->prof_n + 1)) {
            hipEvent_t e;
            LS_HIP(hipEventCreate(&e));
            ix->prof_ev.push_back(e);
        }
        pe = &ix->prof_ev[2 * ix->prof_n];
    }
    // The select of a chain batch runs on the chain's select stream, behind an event attached to the
    // pass's dispatch (no extra packet in the lane). While profiling, a timing pair rides there
    // instead and brackets exactly the kernel.
    const bool xsel = ss != sm;
    hipEvent_t const ev_start = prof ? pe[0] : nullptr;
    hipEvent_t const ev_stop = prof ? pe[1] : (xsel ? st.ev_pass : nullptr);
    // the set's previous select (four batches ago) has read the queues this pass refills
    if (xsel && st.sel_recorded) LS_HIP(hipStreamWaitEvent(sm, st.ev_sel, 0));
    if (b.f32) {
        rc = ls_launch_gemm32_filter(ix->d_corpus, ix->n, g, (const float*)st.d_qh, b.nq, b.nq_pad,
                                     st.d_tau, b.nsplits, b.rps, 1, bufs, sm, ev_start, ev_stop);
    } else if (next) {
        ls_index::bc_set& sn = ix->bc_sets[next->set_id];
        ls_gemm_fuse fz;
        fz.d_qh_next = sn.d_qh;
        fz.nq_next = next->nq;
        fz.d_sample_top_next = sn.d_sample_top;
        fz.sample_stride = next->sample_stride;
        rc = ls_launch_gemm_filter(ix->d_corpus, ix->n, g, st.d_qh, b.nq, b.nq_pad, st.d_tau, b.nsplits,
                                   b.rps, 1, bufs, b.top2, sm, &fz, ev_start, ev_stop);
    } else {
        rc = ls_launch_gemm_filter(ix->d_corpus, ix->n, g, st.d_qh, b.nq, b.nq_pad, st.d_tau, b.nsplits,
                                   b.rps, 1, bufs, b.top2, sm, nullptr, ev_start, ev_stop);
    }
    if (rc != LS_OK) return rc;
    ix->n_launches_total++;
    if (prof) {
        if (xsel) LS_HIP(hipEventRecord(st.ev_pass, sm));  // (profiling pass only: its own packet)
        ix->prof_n++;
    }
    if (tau_next) {  // the rider's tau goes in front of this batch's select: it is the lane's critical kernel
        ls_index::bc_set& sn = ix->bc_sets[next->set_id];
        if ((rc = ls_launch_tau(sn.d_sample_top, next->nsplits, next->nq, next->nq_pad, next->jrank,
                                sn.d_tau, sm)) != LS_OK)
            return rc;
        ix->n_launches_total++;
    }
    if (xsel) LS_HIP(hipStreamWaitEvent(ss, st.ev_pass, 0));
    if ((rc = bc_launch_select(ix, b, ss)) != LS_OK) return rc;
    ix->n_launches_total++;
    if (xsel) {
        LS_HIP(hipEventRecord(st.ev_sel, ss));
        st.sel_recorded = true;
    }
    if (st.multi_stream) LS_HIP(hipEventRecord(st.done, ss));
    return LS_OK;
}

// The pipelined batches whose passes are still held back (see above): queue their passes and selects now.
int ls_i_flush_deferred(ls_index* ix) {
    while (!ix->held.empty()) {
        const ls_index::bc_stage h = ix->held.front();
        ix->held.pop_front();
        if (int rc = bc_launch_pass_select(ix, h, nullptr, ix->chain_main[h.lane], ix->chain_sel)) return rc;
    }
    return LS_OK;
}

static int batched_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k,
                                    uint32_t flags, float* d_out_s, int64_t* d_out_i,
                                    hipStream_t s) {
    int rc = ls_i_flush_pending(ix);
    if (rc != LS_OK) return rc;
    const ls_geom& g = ix->g;
    const bool f32 = ix->dtype == LS_DTYPE_F32;
    const int QT = f32 ? 64 : ls_gemm_qt(g);  // queries per workgroup
    const int64_t nq_pad = (nq + QT - 1) / QT * QT;
    const int64_t qkeep_need = nq * g.d;
    if ((int)ix->bc_pending.size() >= ix->bc_slots() ||
        (!ix->bc_pending.empty() &&
         (nq_pad > ix->bc_slot_stride || qkeep_need > ix->bc_qkeep_stride))) {
        rc = ls_i_batched_repair(ix);  // slots exhausted (or too small): check what is pending
        if (rc != LS_OK) return rc;
        ix->n_forced_checks++;
    }
    const bool chain = (flags & LS_FLAG_PIPELINE) != 0;
    if (!chain && (rc = ls_i_flush_deferred(ix)) != LS_OK) return rc;
    const uint64_t seq = chain ? ix->bc_lane_rr++ : 0;
    const int lane = (int)(seq & 1);  // LS_BC_LANES is even: a scratch set always belongs to one lane
    const int set_id = chain ? 1 + (int)(seq % LS_BC_LANES) : 0;
    ls_index::bc_set& st = ix->bc_sets[set_id];
    hipStream_t const caller = s;
    hipStream_t sp = s, sm = s, ss = s;  // prep / sample, tau, pass / select
    if (chain) {
        if (!ix->chain_main[0]) {
            // (two streams of one priority class are given two hardware queues, in creation order;
            // kernel traces show them as queues 3 and 4)
            int least = 0, greatest = 0;
            LS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            LS_HIP(hipStreamCreateWithPriority(&ix->chain_main[0], hipStreamNonBlocking, greatest));
            LS_HIP(hipStreamCreateWithPriority(&ix->chain_main[1], hipStreamNonBlocking, greatest));
            // the selects' own stream, in the same class: a lower class is starved for as long as
            // workgroups of a pass are waiting for CUs, which with two lanes is always
            LS_HIP(hipStreamCreateWithPriority(&ix->chain_sel, hipStreamNonBlocking, greatest));
            LS_HIP(hipEventCreateWithFlags(&ix->chain_in, hipEventDisableTiming));
        }
        if (!st.ev_prep) {
            LS_HIP(hipEventCreateWithFlags(&st.ev_prep, hipEventDisableTiming));
            LS_HIP(hipEventCreateWithFlags(&st.ev_pass, hipEventDisableTiming));
            LS_HIP(hipEventCreateWithFlags(&st.ev_sel, hipEventDisableTiming));
        }
        sp = sm = ix->chain_main[lane];
        ss = ix->chain_sel;
    } else if (st.used && st.last_stream != s) {
        // set 0 is shared by plain calls: a call on another stream waits for the previous one
        if (!st.multi_stream) {
            LS_HIP(hipStreamSynchronize(st.last_stream));  // once: no event was recorded yet
            LS_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
            st.multi_stream = true;
        } else {
            LS_HIP(hipStreamWaitEvent(s, st.done, 0));
        }
    }
    const bc_plan plan = bc_make_plan(ix, nq, k);
    const int nsplits = plan.nsplits;
    const size_t nrec = (size_t)nq_pad * nsplits;

    // (growing a buffer frees it first: hipFree drains the device, whatever stream still uses it)
    size_t c;
    c = st.qh_cap;
    if ((rc = ls_grow((unsigned char**)&st.d_qh, &c, (size_t)nq_pad * g.d_pad * (f32 ? 4 : 2))) != LS_OK)
        return rc;
    st.qh_cap = c;
    if ((rc = ls_grow(&st.d_queues, &st.queues_cap, nrec * 4 * LS_GEMM_QCAP)) != LS_OK) return rc;
    if ((rc = ls_grow(&st.d_counts, &st.counts_cap, nrec * 4)) != LS_OK) return rc;
    if ((rc = ls_grow(&st.d_tau, &st.tau_cap, (size_t)nq_pad)) != LS_OK) return rc;
    if (ix->bc_pending.empty()) {
        if (nq_pad > ix->bc_slot_stride) ix->bc_slot_stride = nq_pad;
        if (qkeep_need > ix->bc_qkeep_stride) {  // the query copies are re-sliced: drop the old blocks
            ix->bc_qkeep_stride = qkeep_need;
            for (float*& blk : ix->d_qkeep_blk) {
                if (blk) LS_HIP(hipFree(blk));
                blk = nullptr;
            }
        }
    }
    if ((rc = ls_grow(&ix->d_overflow, &ix->overflow_cap,
                   (size_t)ix->bc_slot_stride * ix->bc_slots())) != LS_OK)
        return rc;
    const int slot = (int)ix->bc_pending.size();
    float*& qblk = ix->d_qkeep_blk[slot / LS_BC_QKEEP_BLOCK];
    if (!qblk) {
        // a block of LS_BC_QKEEP_BLOCK slots is 4 GB for nq = 16384, d = 1024: if it does not
        // fit, the backlog is checked (every slot freed) and the call retried by the caller
        if (hipMalloc((void**)&qblk, sizeof(float) * (size_t)ix->bc_qkeep_stride * LS_BC_QKEEP_BLOCK) != hipSuccess) {
            (void)hipGetLastError();
            qblk = nullptr;
            ls_set_error("batched call: out of device memory for the repair copies of %d queued batches "
                         "(%zu bytes per batch): check the pending calls (ls_check) and retry",
                         LS_BC_QKEEP_BLOCK, sizeof(float) * (size_t)ix->bc_qkeep_stride);
            return LS_ERR_HIP;
        }
    }
    u32* d_flags = ix->d_overflow + (size_t)slot * ix->bc_slot_stride;
    float* d_qkeep = qblk + (size_t)(slot % LS_BC_QKEEP_BLOCK) * ix->bc_qkeep_stride;
    if ((rc = ls_grow(&st.d_sample_top, &st.sample_top_cap, nrec * 16)) != LS_OK) return rc;
    if ((rc = ls_grow_pinned(&ix->h_overflow, &ix->h_overflow_cap,
                          (size_t)ix->bc_slot_stride * ix->bc_slots())) != LS_OK)
        return rc;

    ls_index::bc_stage b;
    b.active = true;
    b.lane = lane;
    b.set_id = set_id;
    b.f32 = f32;
    b.nq = nq;
    b.nq_pad = nq_pad;
    b.k = k;
    b.nsplits = nsplits;
    b.sample_stride = plan.sample_stride;
    b.jrank = plan.jrank;
    b.keys_need = plan.keys_need;
    b.rps = plan.rps;
    // two kept sample scores per lane are enough when a query has >= 4 j lanes (two of its best
    // j sample scores then share a lane with probability ~1/8 each)
    b.top2 = LS_GEMM_SAMPLE_TOP2 && (long long)nsplits * 4 >= 4ll * plan.jrank;
    b.d_flags = d_flags;
    b.d_out_s = d_out_s;
    b.d_out_i = d_out_i;
    ls_gemm_bufs bufs;
    bufs.d_queues = st.d_queues;
    bufs.d_counts = st.d_counts;
    bufs.d_overflow = d_flags;
    bufs.d_sample_top = st.d_sample_top;

    // ---- prep -------------------------------------------------------------------------------------
    if (chain) {
        // behind everything the caller has queued so far (its queries) ...
        LS_HIP(hipEventRecord(ix->chain_in, caller));
        LS_HIP(hipStreamWaitEvent(sp, ix->chain_in, 0));
        // (everything else that touches the set is earlier work of this very lane: stream order)
    }
    rc = f32 ? ls_launch_prep_f32(d_q, (float*)st.d_qh, d_qkeep, nq, nq_pad, g,
                                  (flags & LS_FLAG_NORMALIZE) != 0, d_flags, sp)
             : ls_launch_prep_f16(d_q, st.d_qh, d_qkeep, nq, nq_pad, g,
                                  (flags & LS_FLAG_NORMALIZE) != 0, d_flags, sp);
    if (rc != LS_OK) return rc;
    int launches = 1;  // counted, not assumed: debug counter 9
    if (chain) {
        // the prep kernel was the only reader of the caller's query buffer (it also took the
        // repair copy): work the caller queues on its stream from here on may overwrite it
        LS_HIP(hipEventRecord(st.ev_prep, sp));
        LS_HIP(hipStreamWaitEvent(caller, st.ev_prep, 0));
    }
    // ---- sample pass: a few tiles of every slice, spread over the slice ------------------------------
    // Pipelined fp16 batches of one plan: the sample phase rides on the pass launch of the batch TWO
    // calls back, which is held until now for that (the register-starved geometries - config 4's
    // 1.5 KiB rows - spend ~1 % of a multi-millisecond batch between passes and keep their own
    // sample launch)
    const bool fuse_ok = chain && !f32 && ix->opt_fused != 0 && g.chunks <= 48 && ls_gemm_rs(g) == 1;
    bool ride = false;
    if (fuse_ok && ix->held.size() == 2) {
        const ls_index::bc_stage& d = ix->held.front();
        ride = d.lane == lane && d.nq_pad == nq_pad && d.k == k && d.nsplits == nsplits && d.rps == b.rps &&
               d.sample_stride == b.sample_stride && d.jrank == b.jrank && d.top2 == b.top2 &&
               d.set_id != set_id;
    }
    if (ride) {
        const ls_index::bc_stage d = ix->held.front();
        ix->held.pop_front();
        if ((rc = bc_launch_pass_select(ix, d, &b, sm, ss, true)) != LS_OK) return rc;  // + tau(this batch)
    } else {
        // nothing to ride on (the first two batches of a run, another shape, a geometry that does not
        // fuse): a full pipeline is flushed, and the sample pass gets its own launch
        if (!fuse_ok || ix->held.size() == 2) {
            if ((rc = ls_i_flush_deferred(ix)) != LS_OK) return rc;
        }
        rc = f32 ? ls_launch_gemm32_filter(ix->d_corpus, ix->n, g, (const float*)st.d_qh, nq, nq_pad,
                                           nullptr, nsplits, b.rps, b.sample_stride, bufs, sm)
                 : ls_launch_gemm_filter(ix->d_corpus, ix->n, g, st.d_qh, nq, nq_pad, nullptr, nsplits,
                                         b.rps, b.sample_stride, bufs, b.top2, sm);
        if (rc != LS_OK) return rc;
        ++launches;
        if ((rc = ls_launch_tau(st.d_sample_top, nsplits, nq, nq_pad, b.jrank, st.d_tau, sm)) != LS_OK)
            return rc;
        ++launches;
    }
    // ---- pass + select: now, or (pipelined fp16 batches) two calls from now / at the next flush -----
    const bool hold = fuse_ok;
    if (hold) {
        ix->held.push_back(b);
    } else {
        const uint64_t before = ix->n_launches_total;
        if ((rc = bc_launch_pass_select(ix, b, nullptr, sm, ss)) != LS_OK) return rc;
        launches += (int)(ix->n_launches_total - before);
        ix->n_launches_total = before;
    }
    st.used = true;
    st.chain = chain;
    st.last_stream = ss;
    ix->bc_last_set = set_id;
    // kernels of THIS batch (a held-back batch: its pass and select follow; a rider's tau was queued
    // with the carrying pass)
    ix->n_batched_launches = launches + (hold ? 2 : 0) + (ride ? 1 : 0);
    ix->n_launches_total += (uint64_t)launches;
    ix->last_path = f32 ? 3 : 2;
    ix->d_last_flags = d_flags;
    ix->last_flags_n = nq;
    ls_index::batched_call bc;
    bc.nq = nq;
    bc.k = k;
    bc.flags = flags;
    bc.d_out_s = d_out_s;
    bc.d_out_i = d_out_i;
    bc.stream = ss;
    bc.slot = slot;
    ix->bc_pending.push_back(bc);
    if (!(flags & (LS_FLAG_ASYNC | LS_FLAG_PIPELINE))) return ls_i_batched_repair(ix);
    return LS_OK;
}

int ls_i_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k, uint32_t flags,
                          float* d_out_s, int64_t* d_out_i, hipStream_t s, bool host_api) {
    const int64_t chunk = ls_i_batched_eligible(ix, nq, k) ? bc_chunk(ix, nq, k) : 0;
    if (chunk >= nq) {
        // the host API synchronises anyway: repair right away
        uint32_t f = host_api ? (flags & ~(LS_FLAG_ASYNC | LS_FLAG_PIPELINE)) : flags;
        return batched_search_on_stream(ix, d_q, nq, k, f, d_out_s, d_out_i, s);
    }
    if (chunk > 0) {
        // The candidate queues cannot hold the whole batch (big nq x big k leaves too few corpus
        // slices per query tile): sub-batches, each verified and repaired before the next, so the
        // call is exact when it returns whatever the flags say (rare shape; it trades the
        // asynchrony for not sending every query through the repair path).
        for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
            const int64_t m = std::min(chunk, nq - q0);
            int rc = batched_search_on_stream(ix, d_q + q0 * ix->g.d, m, k, flags & LS_FLAG_NORMALIZE,
                                              d_out_s + q0 * k, d_out_i + q0 * k, s);
            if (rc != LS_OK) return rc;
        }
        ix->d_last_flags = nullptr;  // already repaired: nothing provisional to export
        ix->last_flags_n = 0;
        ix->n_chunked_calls++;
        return LS_OK;
    }
    ix->d_last_flags = nullptr;  // the scan path is exact in stream order: nothing to verify
    ix->last_flags_n = 0;
    ix->last_path = 1;
    return scan_search_on_stream(ix, d_q, nq, k, flags, d_out_s, d_out_i, s);
}

int ls_i_check_search_args(const ls_index* ix, const void* q, int64_t nq, int32_t k,
                             uint32_t flags, const void* os, const void* oi) {
    if (!ix) {
        ls_set_error("search: index is null");
        return LS_ERR_INVALID_ARG;
    }
    if (nq < 0 || k <= 0 || (nq > 0 && (!q || !os || !oi))) {
        ls_set_error("search: bad argument (nq=%lld k=%d)", (long long)nq, k);
        return LS_ERR_INVALID_ARG;
    }
    if (flags & ~(LS_FLAG_NORMALIZE | LS_FLAG_ASYNC | LS_FLAG_PIPELINE | LS_FLAG_INORDER)) {
        ls_set_error("search: unknown flags 0x%x", flags);
        return LS_ERR_INVALID_ARG;
    }
    if (std::min<int64_t>(k, ix->n) > LS_MAX_K || k > (1 << 20)) {
        ls_set_error("search: min(k, ntotal) = %lld exceeds LS_MAX_K = %d",
                     (long long)std::min<int64_t>(k, ix->n), LS_MAX_K);
        return LS_ERR_K_TOO_LARGE;
    }
    return LS_OK;
}


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
static std::atomic<uint64_t> g_lead_trace[8];
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
    const bool small_call = nq <= scan_path_max_nq(ix, k);
    c.out_direct = on <= (size_t)(1 << 16);
    // Small scan-path calls: the finalize workgroup of every query writes tagged result granules
    // (k <= LS_OUT_GRAN_MAX_K) or drained rows + a completion word into pinned host memory; the host
    // spins on those instead of sleeping in hipStreamSynchronize (whose wake-up costs more than the
    // 47 us scan's launch). Falls back to the stream sync after 2 ms.
    c.spin = small_call && c.out_direct && !ls_i_batched_eligible(ix, nq, k) && ix->n > 0;
    const bool overlapped = ix->opt_overlap_calls && c.spin && scan_group_count(ix, nq, k) == 1;
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
// path serves up to 16 queries per corpus pass (fp32: ls_mq.hip, ~60 us for 16 at N = 200 k; one query
// alone: 47 us), so requests that arrive while a search is running are not queued behind the mutex one by
// one: they wait in a queue, and whoever holds the leadership serves ALL compatible waiters (same k, same
// flags, <= LS_SCAN_PATH_MAX_NQ queries in total) as ONE batch, then hands their results back. A lone
// caller becomes leader at once and pays nothing; waiters sleep on a condition variable (no
// spinning on the mutex). Round 5: the leadership is released as soon as the batch's launch is QUEUED
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
        if (rc != LS_OK) snprintf(r->err, sizeof(r->err), "%s", g_err);
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
    if (!ix->opt_combine || nq > scan_path_max_nq(ix, k))  // what fills a pass by itself gains nothing from company
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
            lk.unlock();
            bool changed = false;
            const bool may_spin = ix->spinners.fetch_add(1, std::memory_order_relaxed) < ls_spin_cap();  // (a CPU to poll on)
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
            if (!me.done.load(std::memory_order_acquire) && (me.taken.load(std::memory_order_relaxed) || ix->leader_active)) {
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
        auto go_early = [&]() {
            const int64_t total = ix->requests_in_flight + (int64_t)ix->req_q.size();
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
            lk.unlock();
            bool changed = false;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned it = 0; !changed; ++it) {
                for (int i = 0; i < 32; ++i) ls_cpu_relax();
                changed = ep.load(std::memory_order_acquire) != seen;
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
        const int batch_cap = scan_path_max_nq(ix, head->k);  // queries ONE pass carries (32 where ls_mq serves the index)
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

int ls_search_device(ls_index* ix, const void* d_q, int64_t nq, int32_t k, uint32_t flags,
                     void* d_out_scores, void* d_out_indices, void* stream) {
    int rc = ls_i_check_search_args(ix, d_q, nq, k, flags, d_out_scores, d_out_indices);
    if (rc != LS_OK) return rc;
    if (nq == 0) return LS_OK;
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group)
        return ls_group_search(ix, (const float*)d_q, false, nq, k, flags, (float*)d_out_scores,
                               (int64_t*)d_out_indices, (hipStream_t)stream);
    LS_HIP(hipSetDevice(ix->device));
    hipStream_t s = (hipStream_t)stream;
    // pipelined results are final after ls_check, synchronous ones when this returns: both leave room for the
    // repair of an ls_mq launch without score vectors (mq_repair); LS_FLAG_ASYNC alone promises stream order
    ix->dev_call_repairable = !(flags & LS_FLAG_INORDER) && ((flags & LS_FLAG_PIPELINE) || !(flags & LS_FLAG_ASYNC));
    rc = ls_i_search_on_stream(ix, (const float*)d_q, nq, k, flags, (float*)d_out_scores,
                          (int64_t*)d_out_indices, s, false);
    ix->dev_call_repairable = false;
    if (rc != LS_OK) return rc;
    if (!(flags & (LS_FLAG_ASYNC | LS_FLAG_PIPELINE))) {
        LS_HIP(hipStreamSynchronize(s));
        if ((rc = mq_repair(ix)) != LS_OK) return rc;
    }
    return LS_OK;
}

int ls_check(ls_index* ix, void* stream) {
    if (!ix) {
        ls_set_error("ls_check: index is null");
        return LS_ERR_INVALID_ARG;
    }
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group) return ls_group_check(ix, (hipStream_t)stream);
    LS_HIP(hipSetDevice(ix->device));
    int rc = ls_i_flush_pending(ix);
    if (rc != LS_OK) return rc;
    rc = ls_i_batched_repair(ix);  // queries whose candidate queues overflowed: exact scan path
    if (rc != LS_OK) return rc;
    LS_HIP(hipStreamSynchronize((hipStream_t)stream));
    return LS_OK;
}

// index.add(x) on a built index (reference extract/index.py:116): the stored rows are carried over
// device-to-device, only the new rows cross PCIe. Synchronises the handle first.
int ls_add(ls_index* ix, const float* rows, int64_t n_add) {
    if (!ix || n_add < 0 || (n_add > 0 && !rows)) {
        ls_set_error("ls_add: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    if (n_add == 0) return LS_OK;
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group) return ls_group_add(ix, rows, n_add);
    if (ix->n + n_add >= 0xffffffffll) {
        ls_set_error("ls_add: %lld rows exceed the 2^32-1 rows one shard can index",
                     (long long)(ix->n + n_add));
        return LS_ERR_INVALID_ARG;
    }
    LS_HIP(hipSetDevice(ix->device));
    int rc = ls_i_flush_pending(ix);
    if (rc == LS_OK) rc = ls_i_batched_repair(ix);
    if (rc != LS_OK) return rc;
    LS_HIP(hipDeviceSynchronize());  // nothing queued on any stream may still read the old buffers
    const int64_t old_n = ix->n;
    rc = alloc_rows(ix, old_n + n_add, true);  // on failure the handle is unchanged
    if (rc != LS_OK) return rc;
    rc = upload_rows(ix, old_n, rows, n_add);
    if (rc != LS_OK) {
        // the new rows never became searchable: restore the zero pad rows behind the old ones
        (void)hipMemset((char*)ix->d_corpus + (size_t)old_n * ix->g.chunks * 16, 0,
                        (size_t)LS_CORPUS_PAD_ROWS * ix->g.chunks * 16);
        return rc;
    }
    ix->n = old_n + n_add;  // committed only after the rows are in HBM
    return LS_OK;
}

// index.reconstruct_n(row0, count): the stored rows as float32 [count, d] in host memory (fp16
// storage returns the rounded values). Used to write an index file without keeping a host copy.
int ls_reconstruct(ls_index* ix, int64_t row0, int64_t count, float* out) {
    if (!ix || row0 < 0 || count < 0 || row0 + count > ix->n || (count > 0 && !out)) {
        ls_set_error("ls_reconstruct: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    if (count == 0) return LS_OK;
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group) return ls_group_reconstruct(ix, row0, count, out);
    LS_HIP(hipSetDevice(ix->device));
    const ls_geom& g = ix->g;
    const char* src = (const char*)ix->d_corpus + (size_t)row0 * g.chunks * 16;
    if (ix->dtype == LS_DTYPE_F32 && g.d_pad == g.d) {
        LS_HIP(hipMemcpy(out, src, (size_t)count * g.d * sizeof(float), hipMemcpyDeviceToHost));
        return LS_OK;
    }
    const int64_t slab = std::max<int64_t>(1, (int64_t)(256ll << 20) / ((int64_t)g.d * 4));
    float* stage = nullptr;
    LS_HIP(hipMalloc((void**)&stage, (size_t)std::min(slab, count) * g.d * sizeof(float)));
    int rc = LS_OK;
    for (int64_t r0 = 0; rc == LS_OK && r0 < count; r0 += slab) {
        const int64_t nr = std::min(slab, count - r0);
        rc = ls_launch_unconvert(src + (size_t)r0 * g.chunks * 16, stage, nr, g, ix->own_stream);
        if (rc == LS_OK && (hipStreamSynchronize(ix->own_stream) != hipSuccess ||
                            hipMemcpy(out + r0 * g.d, stage, (size_t)nr * g.d * sizeof(float),
                                      hipMemcpyDeviceToHost) != hipSuccess)) {
            ls_set_error("ls_reconstruct: HIP copy/launch failed");
            rc = LS_ERR_HIP;
        }
    }
    (void)hipFree(stage);
    return rc;
}

// faiss.normalize_L2 on a host array: per-device pinned staging buffers that the kernel reads and
// writes over PCIe itself, cached across calls (no allocation, no copy command, one stream
// sync per call).
struct ls_norm_cache {
    std::mutex mu;
    float* h_in = nullptr;
    float* h_out = nullptr;
    size_t cap = 0;  // floats
    hipStream_t s = nullptr;
};
static ls_norm_cache g_norm[64];

int ls_normalize_l2(float* x, int64_t nq, int32_t d, int32_t device) {
    if (nq < 0 || d <= 0 || (nq > 0 && !x)) {
        ls_set_error("ls_normalize_l2: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    int rc = ls_i_check_device(device);
    if (rc != LS_OK) return rc;
    if (nq == 0) return LS_OK;
    if (device >= 64) {
        ls_set_error("ls_normalize_l2: device ordinal %d not supported", device);
        return LS_ERR_NO_DEVICE;
    }
    LS_HIP(hipSetDevice(device));
    ls_norm_cache& nc = g_norm[device];
    std::lock_guard<std::mutex> lk(nc.mu);
    if (!nc.s) LS_HIP(hipStreamCreateWithFlags(&nc.s, hipStreamNonBlocking));
    ls_geom g{};
    g.d = d;
    g.d_pad = d;
    const int64_t rows_per_pass = std::max<int64_t>(1, (int64_t)(64ll << 20) / ((int64_t)d * 4));
    for (int64_t r0 = 0; r0 < nq; r0 += rows_per_pass) {
        const int64_t rows = std::min(rows_per_pass, nq - r0);
        const size_t cnt = (size_t)rows * d;
        if (cnt > nc.cap) {
            size_t c1 = nc.cap, c2 = nc.cap;
            if ((rc = ls_grow_pinned(&nc.h_in, &c1, cnt)) != LS_OK) return rc;
            if ((rc = ls_grow_pinned(&nc.h_out, &c2, cnt)) != LS_OK) return rc;
            nc.cap = std::min(c1, c2);
        }
        memcpy(nc.h_in, x + r0 * d, cnt * sizeof(float));
        rc = ls_launch_prep(nc.h_in, nc.h_out, rows, g, true, false, nc.s);
        if (rc != LS_OK) return rc;
        LS_HIP(hipStreamSynchronize(nc.s));
        memcpy(x + r0 * d, nc.h_out, cnt * sizeof(float));
    }
    return LS_OK;
}

// Copy the per-query verification flags of the most recent search queued on this handle into
// d_dst (device memory, u32 [nq]) on `stream`: non-zero = that query will be repaired by the next
// ls_check. The scan path is always exact, so its calls export zeros.
int ls_export_flags(ls_index* ix, void* d_dst, int64_t nq, void* stream) {
    if (!ix || nq < 0 || (nq > 0 && !d_dst)) {
        ls_set_error("ls_export_flags: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    if (nq == 0) return LS_OK;
    ls_quiesce lk(ix);  // (no synchronous host call in flight, then the handle's mutex)
    if (ix->group) {
        ls_set_error("ls_export_flags: not available on a sharded handle (its shards' flags travel "
                     "with the exchange; ls_check repairs and re-merges)");
        return LS_ERR_INVALID_ARG;
    }
    LS_HIP(hipSetDevice(ix->device));
    return ls_i_export_flags(ix, d_dst, nq, (hipStream_t)stream);
}

}  // extern "C"

int ls_i_export_flags(ls_index* ix, void* d_dst, int64_t nq, hipStream_t s) {
    if (int rc = ls_i_flush_deferred(ix)) return rc;  // the flags are final behind the batch's select
    if (ix->d_last_flags && ix->last_flags_n == nq) {
        ls_index::bc_set& st = ix->bc_sets[ix->bc_last_set];
        if (st.last_stream != s) {  // flags are final on the stream the select ran on
            if (st.chain) {
                // the chain's select stream is the library's own: order `s` behind the set's select
                LS_HIP(hipStreamWaitEvent(s, st.ev_sel, 0));
            } else if (st.multi_stream) {
                LS_HIP(hipStreamWaitEvent(s, st.done, 0));
            } else {
                LS_HIP(hipStreamSynchronize(st.last_stream));
            }
        }
        LS_HIP(hipMemcpyAsync(d_dst, ix->d_last_flags, sizeof(u32) * (size_t)nq,
                              hipMemcpyDeviceToDevice, s));
    } else {
        LS_HIP(hipMemsetAsync(d_dst, 0, sizeof(u32) * (size_t)nq, s));
    }
    return LS_OK;
}

extern "C" {

int ls_merge_topk(const void* d_scores_in, const void* d_indices_in, int32_t n_lists, int64_t nq,
                  int32_t k, void* d_out_scores, void* d_out_indices, int32_t device,
                  void* stream) {
    if (n_lists <= 0 || nq < 0 || k <= 0 ||
        (nq > 0 && (!d_scores_in || !d_indices_in || !d_out_scores || !d_out_indices))) {
        ls_set_error("ls_merge_topk: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    int rc = ls_i_check_device(device);
    if (rc != LS_OK) return rc;
    LS_HIP(hipSetDevice(device));
    return ls_launch_merge((const float*)d_scores_in, (const int64_t*)d_indices_in,
                           nq * k * (int64_t)sizeof(float), nq * k * (int64_t)sizeof(int64_t),
                           n_lists, nq, k, (float*)d_out_scores, (int64_t*)d_out_indices,
                           (hipStream_t)stream);
}

int ls_merge_topk_strided(const void* d_scores_in, const void* d_indices_in,
                          int64_t list_stride_bytes, int32_t n_lists, int64_t nq, int32_t k,
                          void* d_out_scores, void* d_out_indices, int32_t device, void* stream) {
    if (n_lists <= 0 || nq < 0 || k <= 0 || list_stride_bytes < 0 || (list_stride_bytes & 7) ||
        (nq > 0 && (!d_scores_in || !d_indices_in || !d_out_scores || !d_out_indices))) {
        ls_set_error("ls_merge_topk_strided: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    int rc = ls_i_check_device(device);
    if (rc != LS_OK) return rc;
    LS_HIP(hipSetDevice(device));
    return ls_launch_merge((const float*)d_scores_in, (const int64_t*)d_indices_in,
                           list_stride_bytes, list_stride_bytes, n_lists, nq, k,
                           (float*)d_out_scores, (int64_t*)d_out_indices, (hipStream_t)stream);
}

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
        if (!value && ix->d_corpus) return grow_score_vectors(ix, LS_QUERIES_PER_LAUNCH_MAX);
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

const char* ls_last_error(void) { return g_err; }
const char* ls_version(void) { return "leansearch-mi355x 0.3.0 (gfx950)"; }
int32_t ls_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

}  // extern "C"
