// ls_select.hip — exact k-selection kernels: the heap + reorder half of faiss `index.search`
// (reference src/lean_explore/search/engine.py:250) and the G-way shard merge.
//
// All selection is on 64-bit keys (ls_common.h), so "top-k under (score desc, row asc)" is
// "k largest unsigned integers", bit-exact by construction. Non-zero keys are unique (one per
// row); key 0 means "no result".
//
// lds_topk: the workgroup primitive. Given up to 8192 keys in LDS it finds the exact k-th
// largest non-zero key T by MSB-first 8-bit radix select (4 passes over the score half; 4 more
// over the row half only when several candidates share the k-th score), compacts the keys >= T
// and orders them (rank-by-counting for k <= 256, bitonic sort above). No pass touches HBM.
//
// ls_finalize_kernel (one workgroup, 1024 threads)
//   fast path : the blocks*k' keys emitted by the scan go to LDS (one load per thread, all in
//               flight together) -> lds_topk. The k-th best emitted key T is a lower bound of
//               the true k-th best. Every row the scan did NOT emit is <= its workgroup's
//               bound, so if max(bound) < T (or every bound is 0: nothing was withheld) the
//               result is the global top-k. Traffic: blocks*(k'+1)*8 B.
//   rescue    : otherwise one sweep over the score vector S[n] collects every row with
//               key >= T (a superset of the answer) into LDS -> lds_topk.
//   general   : if even that overflows LDS (>8192 rows tie with or beat T, e.g. all scores
//               equal) or too few keys were emitted: 4-pass radix select over S itself.
//   Every path is exact; only their cost differs.
#include "ls_common.h"

#define LS_RES_CAP 2048  // == LS_MAX_K

// ---- workgroup-wide bitonic sort, descending, m a power of two, keys in LDS ------------------
// Pair p of a stage is handled by thread p % nthreads. For strides j <= 64 both elements of
// pair p lie in the 128-element block p >> 6, and all lanes of a wave share p >> 6, so those
// stages need only wave-level ordering (LDS is in-order per wave); only strides >= 128 cross
// waves and take a workgroup barrier: 10 barriers instead of 55 for m = 1024.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
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
            const int jn = j > 1 ? (j >> 1) : k2;  // stride of the next stage
            if (j > 64 || jn > 64)
                __syncthreads();
            else
                wave_lds_fence();
        }
    }
    __syncthreads();
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Wave 0 finds the histogram bin that holds the krem-th largest element (krem is clamped to
// the number of elements counted). out[0] = bin, out[1] = krem - (#elements in higher bins),
// out[2] = hist[bin], out[3] = total, out[4] = clamped krem.
// Must be followed by __syncthreads() before `out` is read.
__device__ __forceinline__ void find_bin(const u32* hist, u32 krem, u32* out, int tid) {
    if (tid < 64) {
        const u32 h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2],
                  h3 = hist[4 * tid + 3];
        const u32 mine = h0 + h1 + h2 + h3;
        u32 suf = mine;  // inclusive suffix sum over lanes >= tid
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_down(suf, o, 64);
            if (tid + o < 64) suf += t;
        }
        const u32 above = suf - mine;
        const u32 total = __shfl(suf, 0, 64);
        if (krem > total) krem = total;
        if (tid == 0) {
            out[3] = total;
            out[4] = krem;
        }
        if (krem > above && krem <= above + mine) {  // exactly one lane when krem >= 1
            u32 cum = above;
            int b;
            u32 hb;
            if (cum + h3 >= krem) { b = 4 * tid + 3; hb = h3; }
            else {
                cum += h3;
                if (cum + h2 >= krem) { b = 4 * tid + 2; hb = h2; }
                else {
                    cum += h2;
                    if (cum + h1 >= krem) { b = 4 * tid + 1; hb = h1; }
                    else { cum += h1; b = 4 * tid; hb = h0; }
                }
            }
            out[0] = (u32)b;
            out[1] = krem - cum;
            out[2] = hb;
        }
    }
}

// hist[digit] += 1 for every active lane. Candidates for one query have nearly equal scores, so
// in the leading passes most lanes share one digit and plain LDS atomics would serialise on it:
// the first active lane's digit is counted once per wave by ballot, the other lanes (spread
// over many digits, few conflicts) use ordinary LDS atomics.
__device__ __forceinline__ void wave_hist_add(u32* hist, u32 digit, bool active, int lane) {
    const u64 act = __ballot(active);
    if (act) {
        const int leader = __ffsll((long long)act) - 1;
        const u32 dsel = (u32)__builtin_amdgcn_readlane((int)digit, leader);
        const u64 same = __ballot(active && digit == dsel);
        if (lane == leader) atomicAdd(&hist[dsel], (u32)__popcll(same));
        if (active && digit != dsel) atomicAdd(&hist[digit], 1u);
    }
}

// Exact top-k of the non-zero keys in keys[0..cnt) (LDS, left untouched) -> res[0..kk) sorted
// descending, kk = min(k, #non-zero). tmp: LDS scratch of LS_RES_CAP keys. k <= LS_RES_CAP.
// hist: 8 * 256 counters (one histogram per radix pass, zeroed here), misc: 8 * 8 words.
// The caller must have synchronised the workgroup after writing keys.
__device__ int lds_topk(const u64* keys, int cnt, int k, u64* res, u64* tmp, u32* hist, u32* misc,
                        int tid, int nt) {
    if (k > cnt) k = cnt;
    if (k <= 0) return 0;
    const int lane = tid & 63;
    const int cnt_pad = (cnt + 63) & ~63;  // whole waves take part in the ballots
    for (int i = tid; i < 8 * 256; i += nt) hist[i] = 0;
    if (tid == 0) misc[7 * 8 + 7] = 0;  // survivor counter
    __syncthreads();
    u32 pref = 0, pmask = 0;
    u32 krem = (u32)k, neq = 0;
    int kk = k;
    for (int pass = 0; pass < 4; ++pass) {  // score half
        const int shift = 24 - 8 * pass;
        u32* h = hist + pass * 256;
        u32* ms = misc + pass * 8;
        for (int i = tid; i < cnt_pad; i += nt) {
            const u64 key = i < cnt ? keys[i] : 0ull;
            const u32 hi = (u32)(key >> 32);
            wave_hist_add(h, (hi >> shift) & 255u, key != 0ull && (hi & pmask) == pref, lane);
        }
        __syncthreads();
        find_bin(h, krem, ms, tid);
        __syncthreads();
        if (pass == 0) {
            kk = (int)ms[4];  // min(k, #non-zero keys)
            if (kk == 0) return 0;
        }
        pref |= ms[0] << shift;
        pmask |= 255u << shift;
        krem = ms[1];
        neq = ms[2];
    }
    const u32 T_hi = pref;
    u32 T_lo = 0;
    if (neq > krem) {  // several candidates share the k-th score: split them on the row half
        u32 lpref = 0, lmask = 0;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            u32* h = hist + (4 + pass) * 256;
            u32* ms = misc + (4 + pass) * 8;
            for (int i = tid; i < cnt_pad; i += nt) {
                const u64 key = i < cnt ? keys[i] : 0ull;
                const u32 lo = (u32)key;
                wave_hist_add(h, (lo >> shift) & 255u,
                              key != 0ull && (u32)(key >> 32) == T_hi && (lo & lmask) == lpref,
                              lane);
            }
            __syncthreads();
            find_bin(h, krem, ms, tid);
            __syncthreads();
            lpref |= ms[0] << shift;
            lmask |= 255u << shift;
            krem = ms[1];
        }
        T_lo = lpref;
    }
    const u64 T = ((u64)T_hi << 32) | (u64)T_lo;  // exactly kk non-zero keys are >= T
    u64* dst = (kk <= 256) ? tmp : res;
    for (int i = tid; i < cnt; i += nt) {
        const u64 key = keys[i];
        if (key != 0ull && key >= T) dst[atomicAdd(&misc[7 * 8 + 7], 1u)] = key;
    }
    __syncthreads();
    if (kk <= 256) {  // order by counting: rank = number of larger survivors
        if (tid < kk) {
            const u64 mine = tmp[tid];
            int rank = 0;
            for (int j = 0; j < kk; ++j) rank += tmp[j] > mine;
            res[rank] = mine;
        }
        __syncthreads();
    } else {
        const int m = next_pow2(kk);
        for (int i = kk + tid; i < m; i += nt) res[i] = 0ull;
        __syncthreads();
        block_bitonic_desc(res, m, tid, nt);
    }
    return kk;
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

// General exact top-k of S[0..n) -> res[0..) sorted descending; returns the number of valid keys.
// 4 radix passes over S for the k-th score, one gather of everything above it, one ordered
// gather of the lowest-index rows equal to it, then a sort. k <= LS_RES_CAP.
__device__ int general_select(const float* __restrict__ S, long long n, int k, u64* res, u32* hist,
                              u32* misc, int tid, int nthreads) {
    u32 prefix = 0, pmask = 0;
    u32 krem = (long long)k < n ? (u32)k : (u32)n;
    int keff = (int)krem;
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
        find_bin(hist, krem, misc, tid);
        __syncthreads();
        if (pass == 0) {
            keff = (int)misc[4];  // min(k, #valid rows)
            if (keff == 0) return 0;
        }
        prefix |= misc[0] << shift;
        pmask |= 255u << shift;
        krem = misc[1];
        __syncthreads();
    }
    const u32 T = prefix;                // ord() of the k-th best score
    const int above = keff - (int)krem;  // rows strictly better than T
    if (tid == 0) misc[4] = 0;
    __syncthreads();
    for (long long r = tid; r < n; r += nthreads) {
        const float s = S[r];
        if (s > -FLT_MAX && ls_ord(s) > T) res[atomicAdd(&misc[4], 1u)] = ls_make_key(s, (u32)r);
    }
    __syncthreads();
    // rows equal to T: take the krem lowest row indices (deterministic tie-break)
    int running = 0;
    for (long long b0 = 0; b0 < n && running < (int)krem; b0 += nthreads) {
        const long long r = b0 + tid;
        float s = 0.0f;
        bool flag = false;
        if (r < n) {
            s = S[r];
            flag = (s > -FLT_MAX) && ls_ord(s) == T;
        }
        int total;
        const int pos = running + block_excl_scan_flag(flag, tid, nthreads, hist, &total);
        if (flag && pos < (int)krem) res[above + pos] = ls_make_key(s, (u32)r);
        running += total;
    }
    __syncthreads();
    const int m = next_pow2(keff);
    for (int i = keff + tid; i < m; i += nthreads) res[i] = 0;
    __syncthreads();
    block_bitonic_desc(res, m, tid, nthreads);
    return keff;
}

__global__ __launch_bounds__(LS_FINAL_THREADS) void ls_finalize_kernel(
    const float* __restrict__ S, long long n, const u64* __restrict__ cand,
    const u64* __restrict__ bound, int blocks, int kprime, int k, long long base,
    float* __restrict__ out_scores, long long* __restrict__ out_indices, u32* counters,
    int force_slow) {
    __shared__ u64 keys[LS_FINAL_CAP];
    __shared__ u64 res[LS_RES_CAP];
    __shared__ u64 tmp[LS_RES_CAP];
    __shared__ u32 hist[8 * 256];
    __shared__ u32 misc[8 * 8];
    __shared__ u64 red[LS_FINAL_THREADS / 64];
    const int tid = threadIdx.x;
    const int nt = LS_FINAL_THREADS;

    const int keff = (long long)k < n ? k : (int)n;
    const int mc = blocks * kprime;
    int nvalid = 0;
    bool done = (n <= 0);
    u64 T = 0;  // k-th best emitted key: a lower bound of the true k-th best key

    if (!done && !force_slow && mc >= keff && mc <= LS_FINAL_CAP) {
        // all global loads first, one round trip
        u64 mb = 0;
        for (int i = tid; i < blocks; i += nt) {
            const u64 b = bound[i];
            mb = b > mb ? b : mb;
        }
        for (int i = tid; i < mc; i += nt) keys[i] = cand[i];
        for (int o = 32; o >= 1; o >>= 1) {
            const u64 other = __shfl_xor(mb, o, 64);
            mb = other > mb ? other : mb;
        }
        if ((tid & 63) == 0) red[tid >> 6] = mb;
        __syncthreads();
        mb = 0;
        for (int w = 0; w < nt / 64; ++w) mb = red[w] > mb ? red[w] : mb;
        nvalid = lds_topk(keys, mc, keff, res, tmp, hist, misc, tid, nt);
        T = (nvalid == keff && keff > 0) ? res[keff - 1] : 0ull;
        done = (mb == 0ull) || (T != 0ull && mb < T);
        __syncthreads();
    }
    if (!done) {
        if (tid == 0 && counters) atomicAdd(&counters[0], 1u);
        bool rescued = false;
        if (T != 0ull && !force_slow) {
            // rescue: every row with key >= T is a candidate; the answer is among them
            if (tid == 0) misc[5] = 0;
            __syncthreads();
            for (long long r = tid; r < n; r += nt) {
                const u64 key = ls_make_key(S[r], (u32)r);
                if (key >= T) {
                    const u32 pos = atomicAdd(&misc[5], 1u);
                    if (pos < LS_FINAL_CAP) keys[pos] = key;
                }
            }
            __syncthreads();
            const int c = (int)misc[5];
            __syncthreads();
            if (c <= LS_FINAL_CAP) {
                nvalid = lds_topk(keys, c, keff, res, tmp, hist, misc, tid, nt);
                rescued = true;
            }
        }
        if (!rescued) {
            if (tid == 0 && counters) atomicAdd(&counters[1], 1u);
            nvalid = general_select(S, n, k, res, hist, misc, tid, nt);
        }
    }
    __syncthreads();
    for (int i = tid; i < k; i += nt) {
        const u64 key = (i < nvalid) ? res[i] : 0ull;
        out_scores[i] = ls_key_score(key);
        out_indices[i] = ls_key_index(key, base);
    }
}

int ls_launch_finalize(const float* d_S, int64_t n, const u64* d_cand, const u64* d_bound,
                       int32_t blocks, int32_t kprime, int32_t k, int64_t base,
                       float* d_out_scores, int64_t* d_out_indices, u32* d_counters,
                       int32_t force_slow, hipStream_t s) {
    hipLaunchKernelGGL(ls_finalize_kernel, dim3(1), dim3(LS_FINAL_THREADS), 0, s, d_S,
                       (long long)n, d_cand, d_bound, blocks, kprime, k, (long long)base,
                       d_out_scores, (long long*)d_out_indices, d_counters, force_slow);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- G-way merge of per-shard results ------------------------------------------
// One workgroup per query: n_lists * k (score, global row) pairs -> keys in LDS -> lds_topk.
// Global rows must be < 2^32 - 1. Padded inputs (index -1) become key 0 and are ignored.
#define LS_MERGE_THREADS 256
__global__ __launch_bounds__(LS_MERGE_THREADS) void ls_merge_kernel(
    const float* __restrict__ sc, const long long* __restrict__ ix, int n_lists, long long nq, int k,
    float* __restrict__ out_scores, long long* __restrict__ out_indices) {
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
        const long long off = ((long long)l * nq + q) * k + j;
        const long long row = ix[off];
        keys[i] = row >= 0 ? ls_make_key(sc[off], (u32)row) : 0ull;
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

int ls_launch_merge(const float* d_scores_in, const int64_t* d_indices_in, int32_t n_lists,
                    int64_t nq, int32_t k, float* d_out_scores, int64_t* d_out_indices,
                    hipStream_t s) {
    if (nq <= 0) return LS_OK;
    const long long mc = (long long)n_lists * k;
    if (mc > LS_FINAL_CAP || k > LS_RES_CAP) {
        ls_set_error("ls_merge_topk: n_lists*k = %lld exceeds %d (or k > %d)", mc, LS_FINAL_CAP,
                     LS_RES_CAP);
        return LS_ERR_K_TOO_LARGE;
    }
    const size_t smem = ((size_t)mc + 2 * LS_RES_CAP) * sizeof(u64) + (8 * 256 + 64) * sizeof(u32);
    static bool attr_set = false;
    if (!attr_set) {
        LS_HIP(hipFuncSetAttribute((const void*)ls_merge_kernel,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL(ls_merge_kernel, dim3((unsigned)nq), dim3(LS_MERGE_THREADS), smem, s,
                       d_scores_in, (const long long*)d_indices_in, n_lists, (long long)nq, k,
                       d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
