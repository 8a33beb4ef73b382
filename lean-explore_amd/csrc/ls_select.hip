// ls_select.hip — exact k-selection kernels: the heap + reorder half of faiss `index.search`
// (reference src/lean_explore/search/engine.py:250) and the G-way shard merge.
//
// All selection is on 64-bit keys (ls_common.h), so "top-k under (score desc, row asc)" is
// "k largest unsigned integers", bit-exact by construction.
//
// ls_finalize_kernel (one workgroup, 1024 threads, 64 KiB LDS)
//   fast path : sort the blocks*k' keys emitted by the scan; the k-th best key T is a lower
//               bound of the true k-th best. Every row the scan did NOT emit is <= its
//               workgroup's bound, so if max(bound) < T (or every bound is 0 = nothing was
//               withheld) the sorted prefix IS the global top-k. Traffic: blocks*(k'+1)*8 B.
//   slow path : otherwise (clustered corpus, huge k, NaN-heavy data) select exactly from the
//               score vector S[n] the scan wrote: 4-pass 8-bit radix select on ord(score) for
//               the k-th value, gather everything above it, then the lowest-index rows equal
//               to it, then sort. Always exact; costs ~6 single-workgroup sweeps of S.
#include "ls_common.h"

// ---- workgroup-wide bitonic sort, descending, m a power of two, keys in LDS ------------------
__device__ __forceinline__ void block_bitonic_desc(u64* a, int m, int tid, int nthreads) {
    for (int k2 = 2; k2 <= m; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (m >> 1); p += nthreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int x = i | j;
                const u64 ai = a[i], ax = a[x];
                const bool desc = (i & k2) == 0;
                if ((ai < ax) == desc) {
                    a[i] = ax;
                    a[x] = ai;
                }
            }
            __syncthreads();
        }
    }
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// exclusive prefix sum of one flag per thread over the workgroup; returns total via *total
__device__ __forceinline__ int block_excl_scan_flag(bool flag, int tid, int nthreads, u32* wsum,
                                                    int* total) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const u64 b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = (u32)__popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        const int c = (int)wsum[w];
        if (w < wave) off += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return off + within;
}

// Exact top-k of S[0..n) into keys[0..), sorted descending. Returns number of valid keys.
__device__ int slow_select(const float* __restrict__ S, long long n, int k, u64* keys, u32* hist,
                           u32* misc, int tid, int nthreads) {
    u32 prefix = 0, pmask = 0;
    int krem = (long long)k < n ? k : (int)n;
    int keff = krem;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += nthreads) hist[i] = 0;
        __syncthreads();
        for (long long r = tid; r < n; r += nthreads) {
            const float s = S[r];
            if (s > -FLT_MAX) {
                const u32 o = ls_ord(s);
                if ((o & pmask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            u32 total = 0;
            for (int b = 0; b < 256; ++b) total += hist[b];
            int kr = krem, ke = keff;
            if (pass == 0 && total < (u32)kr) {  // fewer valid rows than requested
                kr = (int)total;
                ke = (int)total;
            }
            u32 cum = 0;
            int bsel = 0;
            for (int b = 255; b >= 0; --b) {
                if (cum + hist[b] >= (u32)kr && kr > 0) {
                    bsel = b;
                    break;
                }
                cum += hist[b];
            }
            misc[0] = (u32)bsel;
            misc[1] = (u32)(kr - (int)cum);  // still needed inside the selected bin
            misc[2] = (u32)ke;
        }
        __syncthreads();
        const u32 bsel = misc[0];
        krem = (int)misc[1];
        keff = (int)misc[2];
        __syncthreads();
        if (keff == 0) return 0;
        prefix |= bsel << shift;
        pmask |= 255u << shift;
    }
    const u32 T = prefix;          // ord() of the k-th best score
    const int above = keff - krem;  // rows strictly better than T
    if (tid == 0) misc[3] = 0;
    __syncthreads();
    for (long long r = tid; r < n; r += nthreads) {
        const float s = S[r];
        if (s > -FLT_MAX && ls_ord(s) > T) {
            const u32 pos = atomicAdd(&misc[3], 1u);
            keys[pos] = ls_make_key(s, (u32)r);
        }
    }
    __syncthreads();
    // rows equal to T: take the krem lowest row indices (deterministic tie-break)
    int running = 0;
    for (long long b0 = 0; b0 < n && running < krem; b0 += nthreads) {
        const long long r = b0 + tid;
        float s = 0.0f;
        bool flag = false;
        if (r < n) {
            s = S[r];
            flag = (s > -FLT_MAX) && ls_ord(s) == T;
        }
        int total;
        const int pos = running + block_excl_scan_flag(flag, tid, nthreads, hist, &total);
        if (flag && pos < krem) keys[above + pos] = ls_make_key(s, (u32)r);
        running += total;
    }
    __syncthreads();
    const int m = next_pow2(keff);
    for (int i = keff + tid; i < m; i += nthreads) keys[i] = 0;
    __syncthreads();
    block_bitonic_desc(keys, m, tid, nthreads);
    return keff;
}

__global__ __launch_bounds__(LS_FINAL_THREADS) void ls_finalize_kernel(
    const float* __restrict__ S, long long n, const u64* __restrict__ cand,
    const u64* __restrict__ bound, int blocks, int kprime, int k, long long base,
    float* __restrict__ out_scores, long long* __restrict__ out_indices, u32* slow_count,
    int force_slow) {
    __shared__ u64 keys[LS_FINAL_CAP];
    __shared__ u32 hist[256];
    __shared__ u32 misc[8];
    __shared__ u64 red[LS_FINAL_THREADS / 64];
    const int tid = threadIdx.x;
    const int nt = LS_FINAL_THREADS;

    const int keff = (long long)k < n ? k : (int)n;
    const int mc = blocks * kprime;
    bool fast = !force_slow && n > 0 && mc >= keff && mc <= LS_FINAL_CAP;
    int nvalid = 0;

    if (fast) {
        const int m = next_pow2(mc);
        for (int i = tid; i < m; i += nt) keys[i] = (i < mc) ? cand[i] : 0ull;
        u64 mb = 0;
        for (int i = tid; i < blocks; i += nt) {
            const u64 b = bound[i];
            mb = b > mb ? b : mb;
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(mb, o, 64);
            mb = other > mb ? other : mb;
        }
        if ((tid & 63) == 0) red[tid >> 6] = mb;
        __syncthreads();
        mb = 0;
        for (int w = 0; w < nt / 64; ++w) mb = red[w] > mb ? red[w] : mb;
        block_bitonic_desc(keys, m, tid, nt);
        const u64 T = keff > 0 ? keys[keff - 1] : 0ull;
        fast = (mb == 0ull) || (mb < T);
        nvalid = keff;
    }
    if (!fast && n > 0) {
        __syncthreads();
        if (tid == 0 && slow_count) atomicAdd(slow_count, 1u);
        nvalid = slow_select(S, n, k, keys, hist, misc, tid, nt);
    }
    __syncthreads();
    for (int i = tid; i < k; i += nt) {
        const u64 key = (i < nvalid) ? keys[i] : 0ull;
        out_scores[i] = ls_key_score(key);
        out_indices[i] = ls_key_index(key, base);
    }
}

int ls_launch_finalize(const float* d_S, int64_t n, const u64* d_cand, const u64* d_bound,
                       int32_t blocks, int32_t kprime, int32_t k, int64_t base,
                       float* d_out_scores, int64_t* d_out_indices, u32* d_slow_count,
                       int32_t force_slow, hipStream_t s) {
    hipLaunchKernelGGL(ls_finalize_kernel, dim3(1), dim3(LS_FINAL_THREADS), 0, s, d_S,
                       (long long)n, d_cand, d_bound, blocks, kprime, k, (long long)base,
                       d_out_scores, (long long*)d_out_indices, d_slow_count, force_slow);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- G-way merge of per-shard results ------------------------------------------
// One workgroup per query: n_lists * k (score, global row) pairs -> keys -> sort -> best k.
// Global rows must be < 2^32 - 1. Padded inputs (index -1) become key 0 and sort last.
__global__ __launch_bounds__(256) void ls_merge_kernel(const float* __restrict__ sc,
                                                       const long long* __restrict__ ix,
                                                       int n_lists, long long nq, int k,
                                                       float* __restrict__ out_scores,
                                                       long long* __restrict__ out_indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* keys = reinterpret_cast<u64*>(smem_raw);
    const long long q = blockIdx.x;
    const int tid = threadIdx.x;
    const int mc = n_lists * k;
    const int m = next_pow2(mc);
    for (int i = tid; i < m; i += 256) {
        u64 key = 0;
        if (i < mc) {
            const int l = i / k, j = i % k;
            const long long off = ((long long)l * nq + q) * k + j;
            const long long row = ix[off];
            if (row >= 0) key = ls_make_key(sc[off], (u32)row);
        }
        keys[i] = key;
    }
    __syncthreads();
    block_bitonic_desc(keys, m, tid, 256);
    for (int i = tid; i < k; i += 256) {
        const u64 key = keys[i];
        out_scores[q * k + i] = ls_key_score(key);
        out_indices[q * k + i] = ls_key_index(key, 0);
    }
}

int ls_launch_merge(const float* d_scores_in, const int64_t* d_indices_in, int32_t n_lists,
                    int64_t nq, int32_t k, float* d_out_scores, int64_t* d_out_indices,
                    hipStream_t s) {
    if (nq <= 0) return LS_OK;
    int m = 1;
    while (m < n_lists * k) m <<= 1;
    if (m > LS_FINAL_CAP) {
        ls_set_error("ls_merge_topk: n_lists*k = %d exceeds %d", n_lists * k, LS_FINAL_CAP);
        return LS_ERR_K_TOO_LARGE;
    }
    hipLaunchKernelGGL(ls_merge_kernel, dim3((unsigned)nq), dim3(256), (size_t)m * sizeof(u64), s,
                       d_scores_in, (const long long*)d_indices_in, n_lists, (long long)nq, k,
                       d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
