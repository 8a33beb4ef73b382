// ls_bm25.hip — lexical name retrieval on the GPU: the eager-sparse BM25+
// scoring of bm25s (reference src/lean_explore/search/engine.py:192-223 calls
// `bm25.retrieve([tokens], k=1000)`; indices built at src/lean_explore/extract/index.py:238-266).
//
// The index is a CSC matrix, one column per vocabulary token: rows = documents containing the
// token, data = idf * tf-part - nonoccurrence (float32). A query adds the columns of its tokens
// into an all-zero score vector IN TOKEN ORDER (one launch per token: a column never repeats a
// document, so the adds need no atomics and the float32 sum has the same order as bm25s's
// numpy loop -> bit-identical scores), adds sum(nonoccurrence[tokens]) and selects the top k
// with the same machinery as the dense path (per-workgroup candidates + bounds -> finalize).
//
// HBM-bound integer/byte work: bytes per query = sum over tokens of 8 B * |column| (row id +
// value) + 12 B * n_docs (the candidate sweep: read the sums, write the final scores, reset the
// accumulator to zero for the next query).
#include "ls_select_dev.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

struct ls_bm25 {
    int32_t device = 0, n_cu = 256;
    int64_t n_docs = 0, n_vocab = 0, nnz = 0;
    std::vector<int64_t> h_indptr;   // host copy: column extents are launch parameters
    std::vector<float> h_nonocc;
    int32_t* d_indices = nullptr;
    float* d_data = nullptr;
    float* d_S = nullptr;   // accumulator: all zeros between searches (the candidate sweep resets it)
    float* d_F = nullptr;   // final scores of the last search (the finalize step's rescue path)
    bool dirty = false;     // a search failed between its first add and its sweep
    float* h_out_s = nullptr;    // pinned, device-visible: the finalize step writes results here
    int64_t* h_out_i = nullptr;
    u64* d_cand = nullptr;
    u64* d_bound = nullptr;
    u32* d_counters = nullptr;
    int32_t blocks = 1;
    hipStream_t stream = nullptr;
    std::mutex mu;
};

__global__ __launch_bounds__(256) void bm25_zero_kernel(float* __restrict__ S, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        S[i] = 0.0f;
}

__global__ __launch_bounds__(256) void bm25_add_column_kernel(float* __restrict__ S,
                                                              const int32_t* __restrict__ rows,
                                                              const float* __restrict__ vals,
                                                              long long len) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < len; i += (long long)gridDim.x * 256)
        S[rows[i]] += vals[i];
}

// F[r] = S[r] + shift (the query's non-occurrence sum), S[r] = 0 for the next search (no separate
// zeroing launch), then per-workgroup best kprime keys + bound, exactly what the dense scan emits,
// so finalize_body can prove / complete the top-k.
__global__ __launch_bounds__(256) void bm25_candidates_kernel(float* __restrict__ S,
                                                              float* __restrict__ F, long long n,
                                                              float shift, u64* __restrict__ cand,
                                                              u64* __restrict__ bound, int kprime) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kp = kprime + 1;
    const long long W = (long long)gridDim.x * 4, gw = (long long)blockIdx.x * 4 + wave;
    const long long NT = (n + 63) / 64;
    u64 lst = 0, thr = 0;
    for (long long t = gw; t < NT; t += W) {
        const long long row = t * 64 + lane;
        const bool valid = row < n;
        float s = 0.0f;
        if (valid) {
            s = S[row] + shift;
            F[row] = s;
            S[row] = 0.0f;
        }
        const u64 key = valid ? ls_make_key(s, (u32)row) : 0ull;
        u64 mask = __ballot(key > thr);
        while (mask) {
            const int j = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            wave_insert(lst, readlane64(key, j), lane, kp);
            thr = readlane64(lst, kp - 1);
        }
    }
    __shared__ u64 sm[4 * LS_KP_MAX];
    if (lane < LS_KP_MAX) sm[wave * LS_KP_MAX + lane] = (lane < kp) ? lst : 0ull;
    __syncthreads();
    if (wave == 0) {
        const u64 mine = sm[lane];
        int rank = 0;
#pragma unroll 8
        for (int i = 0; i < 4 * LS_KP_MAX; ++i) {
            const u64 o = sm[i];
            rank += (o > mine) || (o == mine && i < lane);
        }
        if (rank < kprime) cand[(long long)blockIdx.x * kprime + rank] = mine;
        if (rank == kprime) bound[blockIdx.x] = mine;
    }
}

extern "C" {

void ls_bm25_destroy(ls_bm25* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    (void)hipFree(ix->d_indices);
    (void)hipFree(ix->d_data);
    (void)hipFree(ix->d_S);
    (void)hipFree(ix->d_F);
    if (ix->h_out_s) (void)hipHostFree(ix->h_out_s);
    if (ix->h_out_i) (void)hipHostFree(ix->h_out_i);
    (void)hipFree(ix->d_cand);
    (void)hipFree(ix->d_bound);
    (void)hipFree(ix->d_counters);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

int ls_bm25_create(ls_bm25** out, const int64_t* indptr, const int32_t* indices, const float* data,
                   const float* nonocc, int64_t n_docs, int64_t n_vocab, int32_t device) {
    if (!out) return LS_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_docs < 0 || n_vocab < 0 || !indptr || (n_vocab > 0 && !nonocc) || n_docs >= 0xffffffffll) {
        ls_set_error("ls_bm25_create: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    const int64_t nnz = indptr[n_vocab];
    if (nnz < 0 || (nnz > 0 && (!indices || !data))) {
        ls_set_error("ls_bm25_create: bad CSC arrays");
        return LS_ERR_INVALID_ARG;
    }
    for (int64_t t = 0; t < n_vocab; ++t)
        if (indptr[t + 1] < indptr[t]) {
            ls_set_error("ls_bm25_create: indptr is not monotone");
            return LS_ERR_INVALID_ARG;
        }
    for (int64_t i = 0; i < nnz; ++i)
        if (indices[i] < 0 || indices[i] >= n_docs) {
            ls_set_error("ls_bm25_create: document index out of range");
            return LS_ERR_INVALID_ARG;
        }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) {
        ls_set_error("no HIP device available; libleansearch has no CPU path");
        return LS_ERR_NO_DEVICE;
    }
    LS_HIP(hipSetDevice(device));
    ls_bm25* ix = new (std::nothrow) ls_bm25();
    if (!ix) return LS_ERR_INVALID_ARG;
    ix->device = device;
    ix->n_docs = n_docs;
    ix->n_vocab = n_vocab;
    ix->nnz = nnz;
    ix->h_indptr.assign(indptr, indptr + n_vocab + 1);
    ix->h_nonocc.assign(nonocc, nonocc + n_vocab);
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0)
        ix->n_cu = cu;
    ix->blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_docs + 255) / 256, 2 * ix->n_cu));
    auto fail = [&](const char* what) {
        ls_set_error("ls_bm25_create: %s failed", what);
        ls_bm25_destroy(ix);
        return LS_ERR_HIP;
    };
    const size_t nz = (size_t)std::max<int64_t>(nnz, 1), nd = (size_t)std::max<int64_t>(n_docs, 1);
    if (hipMalloc((void**)&ix->d_indices, nz * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_data, nz * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_S, nd * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_F, nd * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMemset(ix->d_S, 0, nd * 4) != hipSuccess) return fail("hipMemset");
    if (hipHostMalloc((void**)&ix->h_out_s, LS_MAX_K * 4, hipHostMallocDefault) != hipSuccess)
        return fail("hipHostMalloc");
    if (hipHostMalloc((void**)&ix->h_out_i, LS_MAX_K * 8, hipHostMallocDefault) != hipSuccess)
        return fail("hipHostMalloc");
    if (hipMalloc((void**)&ix->d_cand, (size_t)ix->blocks * LS_KP_MAX * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_bound, (size_t)ix->blocks * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_counters, 32) != hipSuccess) return fail("hipMalloc");
    if (hipMemset(ix->d_counters, 0, 32) != hipSuccess) return fail("hipMemset");
    if (nnz > 0) {
        if (hipMemcpy(ix->d_indices, indices, (size_t)nnz * 4, hipMemcpyHostToDevice) != hipSuccess)
            return fail("upload");
        if (hipMemcpy(ix->d_data, data, (size_t)nnz * 4, hipMemcpyHostToDevice) != hipSuccess)
            return fail("upload");
    }
    if (hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess) return fail("stream");
    *out = ix;
    return LS_OK;
}

int64_t ls_bm25_ntotal(const ls_bm25* ix) { return ix ? ix->n_docs : -1; }

// token_ids: host int32 [n_tokens], ids of the query's tokens in query order (duplicates count
// twice, like bm25s); out_scores host f32 [k], out_docs host i64 [k], (-FLT_MAX, -1) padded.
int ls_bm25_search(ls_bm25* ix, const int32_t* token_ids, int32_t n_tokens, int32_t k,
                   float* out_scores, int64_t* out_docs) {
    if (!ix || n_tokens < 0 || k <= 0 || (n_tokens > 0 && !token_ids) || !out_scores || !out_docs) {
        ls_set_error("ls_bm25_search: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    if (std::min<int64_t>(k, ix->n_docs) > LS_MAX_K) {
        ls_set_error("ls_bm25_search: min(k, n_docs) exceeds LS_MAX_K");
        return LS_ERR_K_TOO_LARGE;
    }
    for (int i = 0; i < n_tokens; ++i)
        if (token_ids[i] < 0 || token_ids[i] >= ix->n_vocab) {
            ls_set_error("ls_bm25_search: token id out of range");
            return LS_ERR_INVALID_ARG;
        }
    std::lock_guard<std::mutex> lk(ix->mu);
    LS_HIP(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    const long long n = ix->n_docs;
    // No query token has a posting (e.g. the raw-token index asked about a multi-word query, whose
    // single token is never a name): every document scores exactly `shift`, so under the total
    // order the answer is documents 0..k-1. The GPU path would produce the same bits the slow way
    // (200k tied keys defeat the candidate proof), so this one case is answered right here.
    int64_t postings = 0;
    for (int i = 0; i < n_tokens; ++i)
        postings += ix->h_indptr[token_ids[i] + 1] - ix->h_indptr[token_ids[i]];
    if (n > 0 && postings == 0) {
        float shift = 0.0f;
        for (int i = 0; i < n_tokens; ++i) shift = shift + ix->h_nonocc[token_ids[i]];
        shift = 0.0f + shift;              // what the sweep computes: S[row] (= 0) + shift
        const bool ok = shift > -FLT_MAX;  // (NaN / -inf rows are never returned)
        for (int i = 0; i < k; ++i) {
            const bool live = ok && i < n;
            out_scores[i] = live ? shift : -FLT_MAX;
            out_docs[i] = live ? i : -1;
        }
        return LS_OK;
    }
    if (n > 0) {
        if (ix->dirty)  // a failed search left partial sums behind
            hipLaunchKernelGGL(bm25_zero_kernel, dim3(ix->blocks), dim3(256), 0, s, ix->d_S, n);
        ix->dirty = true;
        float shift = 0.0f;
        for (int i = 0; i < n_tokens; ++i) {
            const int64_t a = ix->h_indptr[token_ids[i]], b = ix->h_indptr[token_ids[i] + 1];
            shift = shift + ix->h_nonocc[token_ids[i]];  // float32, query order (as bm25s sums)
            if (b > a) {
                const int grid = (int)std::min<int64_t>((b - a + 255) / 256, 8 * ix->n_cu);
                hipLaunchKernelGGL(bm25_add_column_kernel, dim3(grid), dim3(256), 0, s, ix->d_S,
                                   ix->d_indices + a, ix->d_data + a, (long long)(b - a));
            }
        }
        const int keff = (int)std::min<int64_t>(k, n);
        const double lam = (double)keff / ix->blocks;
        int kprime = (int)(lam + 5.0 * __builtin_sqrt(lam) + 3.0);
        kprime = std::max(2, std::min(kprime, LS_KP_MAX - 1));
        hipLaunchKernelGGL(bm25_candidates_kernel, dim3(ix->blocks), dim3(256), 0, s, ix->d_S, ix->d_F,
                           n, shift, ix->d_cand, ix->d_bound, kprime);
        LS_HIP(hipGetLastError());
        ix->dirty = false;
        ls_fin_batch jobs{};
        ls_fin_params& p = jobs.p[0];
        p.S = ix->d_F;
        p.n = n;
        p.cand = ix->d_cand;
        p.bound = ix->d_bound;
        p.blocks = ix->blocks;
        p.kprime = kprime;
        p.k = k;
        p.keys_cap = LS_FINAL_CAP;
        p.force_slow = 0;
        p.base = 0;
        p.out_scores = ix->h_out_s;  // pinned host memory, written by the kernel over PCIe
        p.out_indices = (long long*)ix->h_out_i;
        p.counters = ix->d_counters;
        if (k > LS_MAX_K) {  // only possible when k > n_docs: select LS_MAX_K >= n_docs, pad on host
            p.k = LS_MAX_K;
        }
        int rc = ls_launch_finalize(jobs, 1, s);
        if (rc != LS_OK) return rc;
        const int kk = std::min(k, LS_MAX_K);
        LS_HIP(hipStreamSynchronize(s));
        memcpy(out_scores, ix->h_out_s, (size_t)kk * 4);
        memcpy(out_docs, ix->h_out_i, (size_t)kk * 8);
        for (int i = kk; i < k; ++i) {
            out_scores[i] = -FLT_MAX;
            out_docs[i] = -1;
        }
    } else {
        for (int i = 0; i < k; ++i) {
            out_scores[i] = -FLT_MAX;
            out_docs[i] = -1;
        }
    }
    return LS_OK;
}

}  // extern "C"
