// ls_gemm.hip — the batched path: Q[nq, d] x Corpus^T[d, N] on the matrix cores with the
// top-k selection fused into the epilogue (BASELINE config 3: N=200k, d=384 fp16, nq=1024,
// k=100). Stands in for faiss `index.search(x, k)` with a large nq
// (reference src/lean_explore/search/engine.py:250; the reference itself only ever sends nq=1).
//
// Why fused: the score matrix is nq*N fp32 = 819 MB for config 3; writing and re-reading it
// would cost more HBM time than the whole MFMA budget, so scores never leave registers.
//
// ls_gemm_filter_kernel — one workgroup = 8 waves = 256 queries x one corpus slice
//   - B operand (queries): each wave keeps its 32 queries' fp16 fragments in VGPRs for the
//     whole slice (KSTEPS x 4 registers), so B costs no LDS or HBM traffic in the loop.
//   - A operand (corpus): tiles of 64 rows stream HBM/L2 -> LDS by DMA (global_load_lds,
//     double buffered) and are shared by the 8 waves. LDS rows are XOR-swizzled on the SOURCE
//     address (chunk ^ (row & 15)) so the ds_read_b128 fragment reads are bank-conflict free.
//   - the two 32-row blocks of a tile run in lock step on two accumulators (independent MFMA
//     chains) while the previous tile's two accumulators are filtered one element per MFMA, in
//     the shadow of the matrix pipe (a microbenchmark of this pattern: tools/mfma_ub.hip).
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
                                                          int d, int d_pad, int normalize,
                                                          u32* __restrict__ overflow) {
    __shared__ float red[4];
    const int qi = blockIdx.x;
    if (threadIdx.x == 0) overflow[qi] = 0u;  // per-query repair flag, cleared for this batch
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
                       bool normalize, u32* d_overflow, hipStream_t s) {
    hipLaunchKernelGGL(ls_prep_f16_kernel, dim3((unsigned)nq_pad), dim3(256), 0, s, d_q,
                       (_Float16*)d_qh, (int)nq, g.d, g.d_pad, normalize ? 1 : 0, d_overflow);
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

// top-4 of a lane's sample scores, descending (branch-free insert)
__device__ __forceinline__ void top4_insert(u32 (&t)[4], u32 v) {
    u32 a = v > t[3] ? v : t[3];
    u32 hi = a > t[2] ? a : t[2], lo = a > t[2] ? t[2] : a;
    t[3] = lo;
    a = hi;
    hi = a > t[1] ? a : t[1];
    lo = a > t[1] ? t[1] : a;
    t[2] = lo;
    a = hi;
    hi = a > t[0] ? a : t[0];
    lo = a > t[0] ? t[0] : a;
    t[1] = lo;
    t[0] = hi;
}

template <int CHUNKS, bool SAMPLE>
__global__ __launch_bounds__(LS_GEMM_THREADS, 2) void ls_gemm_filter_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride,
    u64* __restrict__ queues, u32* __restrict__ counts, int cap, u32* __restrict__ overflow,
    u32* __restrict__ sample_top) {
    constexpr int KSTEPS = CHUNKS / 2;                    // 16 fp16 per MFMA k-step = 2 chunks
    constexpr int ROW_BYTES = CHUNKS * 16;
    constexpr int TILE_CHUNKS = LS_GEMM_TM * CHUNKS;      // 16-byte chunks per LDS tile
    constexpr int TILE_BYTES = TILE_CHUNKS * 16;
    constexpr int LOADS = TILE_CHUNKS / LS_GEMM_THREADS;  // 16-byte loads per thread per tile
    constexpr int NRB = LS_GEMM_TM / 32;                  // MFMA row blocks per tile
    static_assert(TILE_CHUNKS % LS_GEMM_THREADS == 0, "tile must split evenly over the threads");
    static_assert(NRB == 2, "the block pipeline below alternates two accumulators");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 tiles

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    int split, qt;
    wg_coords((int)blockIdx.x, nqt, &split, &qt);
    const long long r_begin = (long long)split * rows_per_split;
    long long r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;
    const int ntiles_all = r_begin < r_end ? (int)((r_end - r_begin + LS_GEMM_TM - 1) / LS_GEMM_TM) : 0;
    const int nt = (ntiles_all + tile_stride - 1) / tile_stride;  // tiles this launch visits

    // B fragments: query j = qt*QT + wave*32 + (lane & 31); k-step kk -> chunk 2kk + half
    const int qj = qt * LS_GEMM_QT + wave * 32 + (lane & 31);
    half8 bq[KSTEPS];
    {
        const u32x4* qrow = qh + (long long)qj * CHUNKS + half;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const u32x4 v = qrow[2 * kk];
            bq[kk] = __builtin_bit_cast(half8, v);
        }
    }
    const bool qvalid = qj < nq;
    const float tauv = SAMPLE ? 0.0f : tau[qj];

    u64* myq = queues + ((long long)blockIdx.x * LS_GEMM_THREADS + tid) * cap;
    int cnt = 0;
    u32 top[4] = {0u, 0u, 0u, 0u};

    // ---- corpus tiles: HBM/L2 -> LDS by DMA (global_load_lds, 16 B per lane), double buffered --
    // Wave w, load j fills the 64 consecutive LDS chunks starting at (w*LOADS + j)*64: chunk Lc
    // is tile row r = Lc / CHUNKS, slot sl = Lc % CHUNKS and receives SOURCE chunk sl ^ (r & 15)
    // (the swizzle is applied to the source address: the DMA destination is lane-linear). The
    // per-thread source offsets are loop invariant; the HBM copy is padded with zero rows past n
    // (ls_api.hip), so no clamping is needed. A 64-row tile takes ~3 us to consume, longer than
    // the DMA's flight time, so one tile of look-ahead suffices.
    int goff[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int Lc = (wave * LOADS + j) * 64 + lane;
        const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
        goff[j] = r * CHUNKS + (sl ^ (r & 15));
    }
    auto stage = [&](int ti, int buf) {
        const u32x4* base = corpus + (r_begin + (long long)ti * LS_GEMM_TM) * CHUNKS;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            unsigned char* dst = smem + buf * TILE_BYTES + (wave * LOADS + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + goff[j]), (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // A fragment of k-step kk, row block rb: row ar = lane & 31, chunk (2kk + half) ^ (ar & 15).
    // (2kk + half) & 15 takes 8 values per lane: 8 precomputed byte offsets + immediates.
    const int ar = lane & 31;
    int lo8[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) lo8[m] = ar * ROW_BYTES + (((2 * m + half) ^ (ar & 15)) * 16);
    auto a_frag = [&](int buf, int rb, int kk) -> half8 {
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + lo8[kk & 7] + buf * TILE_BYTES +
                                                        rb * 32 * ROW_BYTES + (kk >> 3) * 256);
        return __builtin_bit_cast(half8, v);
    };

    // ---- one element of a finished 32x32 block ------------------------------------------------------
    // element r: row = row0 + (r&3) + 8*(r>>2) (row0 already includes 4*half)
    auto check = [&](const f32x16& acc, int r, long long row0) {
        const float s = acc[r];
        const long long row = row0 + (r & 3) + 8 * (r >> 2);
        if (SAMPLE) {
            const u64 key = (qvalid && row < r_end) ? ls_make_key(s, 0u) : 0ull;
            top4_insert(top, (u32)(key >> 32));
        } else if (s >= tauv) {
            const u64 key = ls_make_key(s, (u32)row);
            if (qvalid && row < r_end && key != 0ull) {
                if (cnt < cap) myq[cnt] = key;
                ++cnt;
            }
        }
    };

    // One tile = two row blocks computed in lock step on two accumulators (independent chains:
    // consecutive MFMAs never wait on each other), with the PREVIOUS tile's two accumulators
    // checked one element per MFMA in the shadow of the matrix pipe.
    auto run_tile = [&](f32x16& c0, f32x16& c1, const f32x16& p0, const f32x16& p1, bool have_prev,
                        long long prev_row0, int buf) {
        half8 a0[LS_GEMM_APF], a1[LS_GEMM_APF];
#pragma unroll
        for (int kk = 0; kk < LS_GEMM_APF && kk < KSTEPS; ++kk) {
            a0[kk] = a_frag(buf, 0, kk);
            a1[kk] = a_frag(buf, 1, kk);
        }
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            if (kk == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[0], bq[0], z, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[kk % LS_GEMM_APF], bq[kk], c0, 0, 0, 0);
            }
            if (kk + LS_GEMM_APF < KSTEPS) a0[kk % LS_GEMM_APF] = a_frag(buf, 0, kk + LS_GEMM_APF);
            if (kk < 16 && have_prev) check(p0, kk, prev_row0);
            if (kk == 0) {
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.0f;
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[0], bq[0], z, 0, 0, 0);
            } else {
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[kk % LS_GEMM_APF], bq[kk], c1, 0, 0, 0);
            }
            if (kk + LS_GEMM_APF < KSTEPS) a1[kk % LS_GEMM_APF] = a_frag(buf, 1, kk + LS_GEMM_APF);
            if (kk < 16 && have_prev) check(p1, kk, prev_row0 + 32);
        }
        if (KSTEPS < 16 && have_prev) {
#pragma unroll
            for (int r = KSTEPS; r < 16; ++r) {
                check(p0, r, prev_row0);
                check(p1, r, prev_row0 + 32);
            }
        }
    };

    f32x16 accA0, accA1, accB0, accB1;  // (A*, B*) alternate between consecutive tiles
    auto tile_row0 = [&](int i) { return r_begin + (long long)(i * tile_stride) * LS_GEMM_TM + 4 * half; };
    if (nt > 0) stage(0, 0);
    __syncthreads();  // the compiler drains the DMA (vmcnt(0)) before the barrier
    for (int i = 0; i < nt; i += 2) {
        if (i + 1 < nt) stage((i + 1) * tile_stride, 1);
        run_tile(accA0, accA1, accB0, accB1, i > 0, tile_row0(i - 1), 0);
        __syncthreads();
        if (i + 1 < nt) {
            if (i + 2 < nt) stage((i + 2) * tile_stride, 0);
            run_tile(accB0, accB1, accA0, accA1, true, tile_row0(i), 1);
            __syncthreads();
        }
    }
    if (nt > 0) {  // the last tile still has to be checked
        const long long row0 = tile_row0(nt - 1);
        const bool last_in_b = ((nt - 1) & 1) != 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            check(last_in_b ? accB0 : accA0, r, row0);
            check(last_in_b ? accB1 : accA1, r, row0 + 32);
        }
    }
    if (SAMPLE) {
        uint4 t4 = make_uint4(top[0], top[1], top[2], top[3]);
        reinterpret_cast<uint4*>(sample_top)[(long long)blockIdx.x * LS_GEMM_THREADS + tid] = t4;
    } else {
        counts[(long long)blockIdx.x * LS_GEMM_THREADS + tid] = (u32)(cnt < cap ? cnt : cap);
        if (cnt > cap) overflow[qj] = 1u;
    }
}

int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, u64* d_queues, u32* d_counts,
                          int cap, u32* d_overflow, u32* d_sample_top, hipStream_t s) {
    const int nqt = (int)(nq_pad / LS_GEMM_QT);
    const dim3 grid((unsigned)(nsplits * nqt)), block(LS_GEMM_THREADS);
    const size_t smem = (size_t)2 * LS_GEMM_TM * g.chunks * 16;
#define LS_GEMM_CASE(C)                                                                          \
    if (g.chunks == C) {                                                                         \
        if (d_tau)                                                                               \
            hipLaunchKernelGGL((ls_gemm_filter_kernel<C, false>), grid, block, smem, s,           \
                               (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq,  \
                               nqt, d_tau, (long long)rows_per_split, tile_stride, d_queues,      \
                               d_counts, cap, d_overflow, d_sample_top);                         \
        else                                                                                     \
            hipLaunchKernelGGL((ls_gemm_filter_kernel<C, true>), grid, block, smem, s,            \
                               (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq,  \
                               nqt, d_tau, (long long)rows_per_split, tile_stride, d_queues,      \
                               d_counts, cap, d_overflow, d_sample_top);                         \
        LS_HIP(hipGetLastError());                                                               \
        return LS_OK;                                                                            \
    }
    LS_GEMM_CASE(16) LS_GEMM_CASE(32) LS_GEMM_CASE(48) LS_GEMM_CASE(64)
#undef LS_GEMM_CASE
    ls_set_error("batched path: unsupported row geometry (%d chunks)", g.chunks);
    return LS_ERR_INVALID_ARG;
}

#define LS_TAU_PER_THREAD 2
// ---- tau: j-th best sample score of each query --------------------------------------------------
// The sample pass left, for every (workgroup, lane), the 4 best sample scores that lane saw
// (ord() of the score, 0 = none). A query owns 2 lanes in each of its nsplits workgroups:
// <= 2*nsplits*4 values, <= 2 per thread. 4 radix passes find the j-th largest.
// Keeping only 4 per lane can only LOWER the result (if one lane held more than 4 of the best
// j), i.e. let more rows through: tau is a speculative, verified threshold either way.
__global__ __launch_bounds__(256) void ls_tau_kernel(const u32* __restrict__ sample_top, int nsplits,
                                                     int nqt, int nq, int j_rank,
                                                     float* __restrict__ tau) {
    __shared__ u32 hist[4 * 256];
    __shared__ u32 misc[4 * 8];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (q >= nq) {
        if (tid == 0) tau[q] = FLT_MAX;  // padded query: nothing passes
        return;
    }
    const int qt = q / LS_GEMM_QT, w = (q % LS_GEMM_QT) / 32, l = q % 32;
    const int total = nsplits * 2 * 4;  // values of this query
    u32 v[LS_TAU_PER_THREAD];
#pragma unroll
    for (int j = 0; j < LS_TAU_PER_THREAD; ++j) {
        const int idx = tid + j * 256;
        u32 x = 0;
        if (idx < total) {
            const int e = idx & 3, sh = idx >> 2, hf = sh & 1, split = sh >> 1;
            const int b = wg_index(split, qt, nqt);
            x = sample_top[((long long)b * LS_GEMM_THREADS + w * 64 + hf * 32 + l) * 4 + e];
        }
        v[j] = x;
    }
    for (int i = tid; i < 4 * 256; i += 256) hist[i] = 0;
    __syncthreads();
    u32 pref = 0, pmask = 0, krem = (u32)j_rank;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
#pragma unroll
        for (int j = 0; j < LS_TAU_PER_THREAD; ++j)
            wave_hist_add(hist + pass * 256, (v[j] >> shift) & 255u,
                          v[j] != 0u && (v[j] & pmask) == pref, lane);
        __syncthreads();
        find_bin(hist + pass * 256, krem, misc + pass * 8, tid);
        __syncthreads();
        if (pass == 0 && misc[3] < (u32)j_rank) {  // fewer than j sample scores: no bound
            if (tid == 0) tau[q] = -FLT_MAX;
            return;
        }
        pref |= misc[pass * 8] << shift;
        pmask |= 255u << shift;
        krem = misc[pass * 8 + 1];
    }
    if (tid == 0) tau[q] = ls_unord(pref);
}

int ls_launch_tau(const u32* d_sample_top, int nsplits, int64_t nq, int64_t nq_pad, int j_rank,
                  float* d_tau, hipStream_t s) {
    if ((long long)nsplits * 2 * 4 > 256LL * LS_TAU_PER_THREAD) {
        ls_set_error("batched path: sample too large for the tau kernel");
        return LS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ls_tau_kernel, dim3((unsigned)nq_pad), dim3(256), 0, s, d_sample_top,
                       nsplits, (int)(nq_pad / LS_GEMM_QT), (int)nq, j_rank, d_tau);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- exact top-k of each query's queues --------------------------------------------------------------
#define LS_BSEL_KEYS 2048
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
    const int qt = q / LS_GEMM_QT, w = (q % LS_GEMM_QT) / 32, l = q % 32;
    if (tid == 0) nkeys = 0;
    __syncthreads();
    // gather: thread t < 2*nsplits owns one of the query's queues: one load for its length, a
    // block-wide prefix for its slot range in LDS, then its (few) live entries
    __shared__ u32 wsum[8];
    const int nqueues = nsplits * 2;
    u32 c = 0;
    const u64* qptr = nullptr;
    if (tid < nqueues) {
        const int half = tid & 1, split = tid >> 1;
        const int b = wg_index(split, qt, nqt);
        const int t = w * 64 + half * 32 + l;
        c = counts[(long long)b * LS_GEMM_THREADS + t];
        qptr = queues + ((long long)b * LS_GEMM_THREADS + t) * cap;
    }
    {   // exclusive prefix sum of c over the 256 threads
        const int lane = tid & 63, wv = tid >> 6;
        u32 inc = c;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t2 = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t2;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        u32 off = 0;
        for (int i = 0; i < wv; ++i) off += wsum[i];
        if (tid == 0) nkeys = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const u32 start = off + inc - c;
        for (u32 e = 0; e < c; ++e)
            if (start + e < LS_BSEL_KEYS) keys[start + e] = qptr[e];
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
                       d_counts, cap, nsplits, (int)(nq_pad / LS_GEMM_QT), k, (long long)base, d_overflow,
                       d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
