// ls_gemm.hip — the batched path: Q[nq, d] x Corpus^T[d, N] on the matrix cores with the
// top-k selection fused into the epilogue (BASELINE config 3: N=200k, d=384 fp16, nq=1024,
// k=100). Stands in for faiss `index.search(x, k)` with a large nq
// (reference src/lean_explore/search/engine.py:250; the reference itself only ever sends nq=1).
//
// Why fused: the score matrix is nq*N fp32 = 819 MB for config 3; writing and re-reading it
// would cost more HBM time than the whole MFMA budget, so scores never leave registers.
//
// ls_gemm_filter_kernel — one workgroup = 4 waves = 128 queries x one corpus slice
//   - B operand (queries): each wave keeps its 32 queries' fp16 fragments in VGPRs for the
//     whole slice (KSTEPS x 4 registers), so B costs no LDS or HBM traffic in the loop.
//   - A operand (corpus): tiles of 32 rows stream HBM -> LDS with global_load_lds (16 B/lane,
//     double buffered) and are shared by the 4 waves. LDS rows are XOR-swizzled on the SOURCE
//     address (chunk ^ (row & 15)) so the ds_read_b128 fragment reads are bank-conflict free.
//   - v_mfma_f32_32x32x16_f16: D[row, query] accumulates in fp32; fp16 x fp16 products are exact.
//   - epilogue: lane (query j, half h) holds 16 row scores of ONE query. A score >= tau[j]
//     (tau = k-th best of a row sample, a certified lower bound of the final k-th best) is
//     appended to this lane's private queue in HBM: no atomics, no cross-lane traffic.
//   - workgroups that share a corpus slice sit on the same XCD (block % 8) so the slice is
//     fetched from HBM once and served to the other query tiles from that XCD's L2.
//
// Phases (ls_api.hip orchestrates): sample pass (tau = -inf over ~4% of the rows) -> tau kernel
// (k-th best sample score per query) -> full pass with tau -> select kernel (exact top-k of
// each query's queues). A queue that overflows flags its query; flagged queries are re-run by
// the exact per-query scan path, so the result is always exact.
#include "ls_select_dev.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// ---- queries -> fp16 [nq_pad, d_pad], normalised if asked, zero padded ---------------------------
__global__ __launch_bounds__(256) void ls_prep_f16_kernel(const float* __restrict__ qin,
                                                          _Float16* __restrict__ qout, int nq,
                                                          int d, int d_pad, int normalize) {
    __shared__ float red[4];
    const int qi = blockIdx.x;
    _Float16* dst = qout + (long long)qi * d_pad;
    if (qi >= nq) {
        for (int j = threadIdx.x; j < d_pad; j += 256) dst[j] = (_Float16)0.0f;
        return;
    }
    const float* src = qin + (long long)qi * d;
    float inv = 1.0f;
    if (normalize) {
        float ss = 0.0f;
        for (int j = threadIdx.x; j < d; j += 256) ss = fmaf(src[j], src[j], ss);
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        ss = (red[0] + red[1]) + (red[2] + red[3]);
        if (ss > 0.0f) inv = 1.0f / sqrtf(ss);
    }
    for (int j = threadIdx.x; j < d_pad; j += 256)
        dst[j] = (_Float16)(j < d ? (normalize ? src[j] * inv : src[j]) : 0.0f);
}

int ls_launch_prep_f16(const float* d_q, void* d_qh, int64_t nq, int64_t nq_pad, const ls_geom& g,
                       bool normalize, hipStream_t s) {
    hipLaunchKernelGGL(ls_prep_f16_kernel, dim3((unsigned)nq_pad), dim3(256), 0, s, d_q,
                       (_Float16*)d_qh, (int)nq, g.d, g.d_pad, normalize ? 1 : 0);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- geometry shared by the kernels ---------------------------------------------------------------
// workgroup b -> (corpus split, query tile): the nqt tiles of one split are consecutive on one XCD
__device__ __forceinline__ void wg_coords(int b, int nqt, int* split, int* qt) {
    const int xcd = b & 7, j = b >> 3;
    *split = (j / nqt) * 8 + xcd;
    *qt = j % nqt;
}
__host__ __device__ __forceinline__ int wg_index(int split, int qt, int nqt) {
    return (((split >> 3) * nqt + qt) << 3) | (split & 7);
}

template <int CHUNKS>
__global__ __launch_bounds__(LS_GEMM_THREADS, 2) void ls_gemm_filter_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride,
    u64* __restrict__ queues, u32* __restrict__ counts, int cap, u32* __restrict__ overflow) {
    constexpr int KSTEPS = CHUNKS / 2;            // 16 fp16 per MFMA k-step = 2 chunks
    constexpr int TILE_CHUNKS = LS_GEMM_TM * CHUNKS;  // 16-byte chunks per LDS tile
    constexpr int LOADS = TILE_CHUNKS / LS_GEMM_THREADS;  // global_load_lds per thread per tile
    static_assert(TILE_CHUNKS % LS_GEMM_THREADS == 0, "tile must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int split, qt;
    wg_coords((int)blockIdx.x, nqt, &split, &qt);
    const long long r_begin = (long long)split * rows_per_split;
    long long r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;
    const int ntiles_all = r_begin < r_end ? (int)((r_end - r_begin + LS_GEMM_TM - 1) / LS_GEMM_TM) : 0;
    const int nt = (ntiles_all + tile_stride - 1) / tile_stride;  // tiles this launch visits

    // B fragments: query j = qt*128 + wave*32 + (lane & 31); k-step kk -> chunk 2kk + (lane >> 5)
    const int qj = qt * 128 + wave * 32 + (lane & 31);
    half8 bq[KSTEPS];
    {
        const u32x4* qrow = qh + (long long)qj * CHUNKS + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const u32x4 v = qrow[2 * kk];
            bq[kk] = __builtin_bit_cast(half8, v);
        }
    }
    const bool qvalid = qj < nq;
    const float tauv = tau ? tau[qj] : -FLT_MAX;

    // private queue of this lane: entry e at queues[(b*256 + tid)*cap + e] (contiguous per lane,
    // so the select kernels read a queue as one coalesced run)
    u64* myq = queues + ((long long)blockIdx.x * LS_GEMM_THREADS + tid) * cap;
    int cnt = 0;

    // stage tile `ti` (index into this split's tiles) into LDS buffer `buf`
    auto stage = [&](int ti, int buf) {
        const long long row0 = r_begin + (long long)ti * LS_GEMM_TM;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int Lc = (wave * LOADS + j) * 64 + lane;  // LDS chunk this lane fills
            const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
            const int c = sl ^ (r & 15);                    // source chunk (swizzle on the source)
            long long row = row0 + r;
            row = row < n ? row : n - 1;
            const u32x4* src = corpus + row * CHUNKS + c;
            unsigned char* dst = smem + (size_t)buf * TILE_CHUNKS * 16 + (size_t)(wave * LOADS + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    if (nt > 0) stage(0, 0);
    __syncthreads();  // (the compiler drains the LDS-DMA before the barrier)

    const int ar = lane & 31;  // A fragment: row ar of the tile, chunk 2kk + (lane >> 5)
    for (int i = 0; i < nt; ++i) {
        const int buf = i & 1;
        if (i + 1 < nt) stage((i + 1) * tile_stride, buf ^ 1);
        const unsigned char* tb = smem + (size_t)buf * TILE_CHUNKS * 16 + (size_t)ar * CHUNKS * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int c = (2 * kk + (lane >> 5)) ^ (ar & 15);
            const u32x4 av = *reinterpret_cast<const u32x4*>(tb + c * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, av), bq[kk], acc,
                                                         0, 0, 0);
        }
        // epilogue: acc[r] = <corpus row, query qj>, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const long long row0 = r_begin + (long long)(i * tile_stride) * LS_GEMM_TM + 4 * (lane >> 5);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float s = acc[r];
            if (s >= tauv) {
                const long long row = row0 + (r & 3) + 8 * (r >> 2);
                const u64 key = ls_make_key(s, (u32)row);
                if (qvalid && row < r_end && key != 0ull) {
                    if (cnt < cap) myq[cnt] = key;
                    ++cnt;
                }
            }
        }
        __syncthreads();  // next tile landed (DMA drained before the barrier) / this one consumed
    }
    counts[(long long)blockIdx.x * LS_GEMM_THREADS + tid] = (u32)(cnt < cap ? cnt : cap);
    if (cnt > cap) overflow[qj] = 1u;
}

int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, u64* d_queues, u32* d_counts,
                          int cap, u32* d_overflow, hipStream_t s) {
    const int nqt = (int)(nq_pad / 128);
    const dim3 grid((unsigned)(nsplits * nqt)), block(LS_GEMM_THREADS);
    const size_t smem = (size_t)2 * LS_GEMM_TM * g.chunks * 16;
#define LS_GEMM_CASE(C)                                                                          \
    if (g.chunks == C) {                                                                         \
        hipLaunchKernelGGL((ls_gemm_filter_kernel<C>), grid, block, smem, s,                      \
                           (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq, nqt, \
                           d_tau, (long long)rows_per_split, tile_stride, d_queues, d_counts, cap, \
                           d_overflow);                                                          \
        LS_HIP(hipGetLastError());                                                               \
        return LS_OK;                                                                            \
    }
    LS_GEMM_CASE(16) LS_GEMM_CASE(32) LS_GEMM_CASE(48) LS_GEMM_CASE(64)
#undef LS_GEMM_CASE
    ls_set_error("batched path: unsupported row geometry (%d chunks)", g.chunks);
    return LS_ERR_INVALID_ARG;
}

// ---- tau: k-th best sample score of each query ----------------------------------------------------
// One workgroup per query; the query's queue entries (<= LS_TAU_PER_THREAD per thread) stay in
// registers; 4 radix passes over the score half. tau = -FLT_MAX when the sample holds < k scores.
#define LS_TAU_PER_THREAD 32
__global__ __launch_bounds__(256) void ls_tau_kernel(const u64* __restrict__ queues,
                                                     const u32* __restrict__ counts, int cap,
                                                     int nsplits, int nqt, int nq, int k,
                                                     float* __restrict__ tau) {
    __shared__ u32 hist[4 * 256];
    __shared__ u32 misc[4 * 8];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (q >= nq) {
        if (tid == 0) tau[q] = FLT_MAX;  // padded query: nothing passes
        return;
    }
    const int qt = q / 128, w = (q % 128) / 32, l = q % 32;
    // entry list of this query: (split, half, e) -> flattened index space nsplits * 2 * cap
    const int total = nsplits * 2 * cap;
    u32 hi[LS_TAU_PER_THREAD];
#pragma unroll
    for (int j = 0; j < LS_TAU_PER_THREAD; ++j) {
        const int idx = tid + j * 256;
        u32 v = 0;
        if (idx < total) {
            const int e = idx % cap, sh = idx / cap, half = sh & 1, split = sh >> 1;
            const int b = wg_index(split, qt, nqt);
            const int t = w * 64 + half * 32 + l;
            if ((u32)e < counts[(long long)b * LS_GEMM_THREADS + t])
                v = (u32)(queues[((long long)b * LS_GEMM_THREADS + t) * cap + e] >> 32);
        }
        hi[j] = v;  // 0 = no entry (ord() of a valid score is never 0: that is -NaN territory)
    }
    for (int i = tid; i < 4 * 256; i += 256) hist[i] = 0;
    __syncthreads();
    u32 pref = 0, pmask = 0, krem = (u32)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int j = 0; j < LS_TAU_PER_THREAD; ++j)
            wave_hist_add(hist + pass * 256, (hi[j] >> shift) & 255u,
                          hi[j] != 0u && (hi[j] & pmask) == pref, lane);
        __syncthreads();
        find_bin(hist + pass * 256, krem, misc + pass * 8, tid);
        __syncthreads();
        if (pass == 0 && misc[3] < (u32)k) {  // fewer than k sample scores: no usable bound
            if (tid == 0) tau[q] = -FLT_MAX;
            return;
        }
        pref |= misc[pass * 8] << shift;
        pmask |= 255u << shift;
        krem = misc[pass * 8 + 1];
    }
    if (tid == 0) tau[q] = ls_unord(pref);
}

int ls_launch_tau(const u64* d_queues, const u32* d_counts, int cap, int nsplits, int64_t nq,
                  int64_t nq_pad, int k, float* d_tau, hipStream_t s) {
    if ((long long)nsplits * 2 * cap > 256LL * LS_TAU_PER_THREAD) {
        ls_set_error("batched path: sample too large for the tau kernel");
        return LS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ls_tau_kernel, dim3((unsigned)nq_pad), dim3(256), 0, s, d_queues, d_counts,
                       cap, nsplits, (int)(nq_pad / 128), (int)nq, k, d_tau);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- exact top-k of each query's queues --------------------------------------------------------------
#define LS_BSEL_KEYS 6144
__global__ __launch_bounds__(256) void ls_batch_select_kernel(
    const u64* __restrict__ queues, const u32* __restrict__ counts, int cap, int nsplits, int nqt,
    int k, long long base, u32* __restrict__ overflow, float* __restrict__ out_scores,
    long long* __restrict__ out_indices) {
    __shared__ u64 keys[LS_BSEL_KEYS];
    __shared__ u64 res[256];
    __shared__ u64 tmp[256];
    __shared__ u32 hist[8 * 256];
    __shared__ u32 misc[64];
    __shared__ u32 nkeys;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int qt = q / 128, w = (q % 128) / 32, l = q % 32;
    if (tid == 0) nkeys = 0;
    __syncthreads();
    // walk the query's 2*nsplits queues in the flattened (queue, entry) index space: a wave
    // reads one queue's 64 slots as one coalesced run and appends the live ones to LDS
    const int total = nsplits * 2 * cap;
    for (int idx = tid; idx < total; idx += 256) {
        const int e = idx % cap, sh = idx / cap, half = sh & 1, split = sh >> 1;
        const int b = wg_index(split, qt, nqt);
        const int t = w * 64 + half * 32 + l;
        if ((u32)e < counts[(long long)b * LS_GEMM_THREADS + t]) {
            const u32 pos = atomicAdd(&nkeys, 1u);
            if (pos < LS_BSEL_KEYS) keys[pos] = queues[((long long)b * LS_GEMM_THREADS + t) * cap + e];
        }
    }
    __syncthreads();
    const int cnt = (int)nkeys;
    if (cnt > LS_BSEL_KEYS) {  // more candidates than fit: exact fallback handles this query
        if (tid == 0) overflow[q] = 1u;
        return;
    }
    if (overflow[q]) return;  // a queue overflowed in the GEMM pass
    __syncthreads();
    const int nvalid = lds_topk(keys, cnt, k, res, tmp, hist, misc, tid, 256);
    __syncthreads();
    if (nvalid < k && tid == 0) overflow[q] = 2u;  // too few candidates (cannot happen with a
                                                   // certified tau unless n < k): fall back
    for (int i = tid; i < k; i += 256) {
        const u64 key = i < nvalid ? res[i] : 0ull;
        out_scores[(long long)q * k + i] = ls_key_score(key);
        out_indices[(long long)q * k + i] = ls_key_index(key, base);
    }
}

int ls_launch_batch_select(const u64* d_queues, const u32* d_counts, int cap, int nsplits,
                           int64_t nq, int64_t nq_pad, int k, int64_t base, u32* d_overflow,
                           float* d_out_scores, int64_t* d_out_indices, hipStream_t s) {
    if (k > LS_GEMM_MAX_K) {
        ls_set_error("batched path: k > %d", LS_GEMM_MAX_K);
        return LS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ls_batch_select_kernel, dim3((unsigned)nq), dim3(256), 0, s, d_queues,
                       d_counts, cap, nsplits, (int)(nq_pad / 128), k, (long long)base, d_overflow,
                       d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
