// ls_select.hip — selection kernels: the stand-alone finalize of one query and the G-way shard
// merge. The device code lives in ls_select_dev.h.
#include "ls_select_dev.h"

#include <algorithm>

// ---- stand-alone finalize: one workgroup of 1024 threads per job, dynamic LDS ---------------------
__global__ __launch_bounds__(LS_FINAL_THREADS) void ls_finalize_kernel(ls_fin_batch jobs) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_fin[];
    finalize_body<LS_FINAL_THREADS>(ls_fin_job(jobs, blockIdx.x), smem_fin, threadIdx.x);
}

int ls_launch_finalize(const ls_fin_batch& jobs, hipStream_t s) {
    if (jobs.njobs <= 0) return LS_OK;
    static ls_attr_once once;
    if (int rc = ls_set_max_dynamic_lds(once, (const void*)ls_finalize_kernel, 128 * 1024)) return rc;
    const ls_fin_params& p = jobs.p0;  // (k, n and the key capacity are the group's)
    const int keff = (int)((long long)p.k < p.n ? p.k : p.n);
    const size_t smem = ls_fin_lds_bytes(p.keys_cap, keff);
    hipLaunchKernelGGL(ls_finalize_kernel, dim3(jobs.njobs), dim3(LS_FINAL_THREADS), smem, s, jobs);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- G-way merge of per-shard results ------------------------------------------
// One workgroup per query: n_lists * k (score, global row) pairs -> keys in LDS -> lds_topk.
// Global rows must be < 2^32 - 1. Padded inputs (index -1) become key 0 and are ignored.
#define LS_MERGE_THREADS 256
__global__ __launch_bounds__(LS_MERGE_THREADS) void ls_merge_kernel(
    const float* __restrict__ sc, const long long* __restrict__ ix, long long stride_s,
    long long stride_i, int n_lists, long long nq, int k, float* __restrict__ out_scores,
    long long* __restrict__ out_indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int mc = n_lists * k;
    u64* keys = reinterpret_cast<u64*>(smem_raw);          // mc
    u64* res = keys + mc;                                   // LS_RES_CAP
    u64* tmp = res + LS_RES_CAP;                            // LS_RES_CAP
    u32* hist = reinterpret_cast<u32*>(tmp + LS_RES_CAP);   // 8 * 256
    u32* misc = hist + 8 * 256;                             // 8 * 8
    const long long q = blockIdx.x;
    const int tid = threadIdx.x;
    for (int i = tid; i < mc; i += LS_MERGE_THREADS) {
        const int l = i / k, j = i - l * k;
        const long long off = q * k + j;  // within list l; lists are stride_* BYTES apart
        const long long row = *(const long long*)((const char*)(ix + off) + l * stride_i);
        const float v = *(const float*)((const char*)(sc + off) + l * stride_s);
        keys[i] = row >= 0 ? ls_make_key(v, (u32)row) : 0ull;
    }
    __syncthreads();
    const int kk = k < LS_RES_CAP ? k : LS_RES_CAP;
    const int nvalid = lds_topk(keys, mc, kk, res, tmp, hist, misc, tid, LS_MERGE_THREADS);
    __syncthreads();
    for (int i = tid; i < k; i += LS_MERGE_THREADS) {
        const u64 key = i < nvalid ? res[i] : 0ull;
        out_scores[q * k + i] = ls_key_score(key);
        out_indices[q * k + i] = ls_key_index(key, 0);
    }
}

int ls_launch_merge(const float* d_scores_in, const int64_t* d_indices_in, int64_t stride_s_bytes,
                    int64_t stride_i_bytes, int32_t n_lists, int64_t nq, int32_t k,
                    float* d_out_scores, int64_t* d_out_indices, hipStream_t s) {
    if (nq <= 0) return LS_OK;
    const long long mc = (long long)n_lists * k;
    if (mc > LS_FINAL_CAP || k > LS_RES_CAP) {
        ls_set_error("ls_merge_topk: n_lists*k = %lld exceeds %d (or k > %d)", mc, LS_FINAL_CAP,
                     LS_RES_CAP);
        return LS_ERR_K_TOO_LARGE;
    }
    const size_t smem = ((size_t)mc + 2 * LS_RES_CAP) * sizeof(u64) + (8 * 256 + 64) * sizeof(u32);
    static ls_attr_once once;
    if (int rc = ls_set_max_dynamic_lds(once, (const void*)ls_merge_kernel, 112 * 1024)) return rc;
    hipLaunchKernelGGL(ls_merge_kernel, dim3((unsigned)nq), dim3(LS_MERGE_THREADS), smem, s,
                       d_scores_in, (const long long*)d_indices_in, (long long)stride_s_bytes,
                       (long long)stride_i_bytes, n_lists, (long long)nq, k, d_out_scores,
                       (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
