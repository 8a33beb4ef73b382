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
int ls_i_grow_score_vectors(ls_index* ix, int need) {
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
int ls_i_scan_path_max_nq(const ls_index* ix, int32_t k) {
    return std::max(LS_SCAN_PATH_MAX_NQ, mq_max_queries(ix, k));
}

// Launches (query groups) the scan path will use for a call of nq queries: ONE definition, shared by the
// scheduling below and by the host API's decision to overlap a call (a call that owns one scratch generation
// must be a single group).
static int scan_group_size(const ls_index* ix, int64_t left, int mq_max) {
    if (mq_max > 0 && left >= 2) return (int)std::min<int64_t>(left, mq_max);
    return !ix->opt_multi_query ? 1 : (left >= 5 ? (int)std::min<int64_t>(left, 8) : (left >= 2 ? (int)std::min<int64_t>(left, 4) : 1));
}
int64_t ls_i_scan_group_count(const ls_index* ix, int64_t nq, int32_t k) {
    const int mq_max = mq_max_queries(ix, k);
    int64_t groups = 0;
    for (int64_t left = nq; left > 0; ++groups) left -= scan_group_size(ix, left, mq_max);
    return groups;
}

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
    const int64_t n_groups = ls_i_scan_group_count(ix, nq, k);
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
            if ((rc = ls_i_grow_score_vectors(ix, NQ)) != LS_OK) return rc;  // (flushes the pending jobs, drains the device)
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

int ls_i_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k, uint32_t flags,
                          float* d_out_s, int64_t* d_out_i, hipStream_t s, bool host_api) {
    const int64_t chunk = ls_i_batched_eligible(ix, nq, k) ? ls_i_bc_chunk(ix, nq, k) : 0;
    if (chunk >= nq) {
        // the host API synchronises anyway: repair right away
        uint32_t f = host_api ? (flags & ~(LS_FLAG_ASYNC | LS_FLAG_PIPELINE)) : flags;
        return ls_i_batched_search_on_stream(ix, d_q, nq, k, f, d_out_s, d_out_i, s);
    }
    if (chunk > 0) {
        // The candidate queues cannot hold the whole batch (big nq x big k leaves too few corpus
        // slices per query tile): sub-batches, each verified and repaired before the next, so the
        // call is exact when it returns whatever the flags say (rare shape; it trades the
        // asynchrony for not sending every query through the repair path).
        for (int64_t q0 = 0; q0 < nq; q0 += chunk) {
            const int64_t m = std::min(chunk, nq - q0);
            int rc = ls_i_batched_search_on_stream(ix, d_q + q0 * ix->g.d, m, k, flags & LS_FLAG_NORMALIZE,
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


extern "C" {

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

const char* ls_last_error(void) { return g_err; }
const char* ls_version(void) { return "leansearch-mi355x 0.3.0 (gfx950)"; }
int32_t ls_device_count(void) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

}  // extern "C"
