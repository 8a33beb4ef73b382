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
//   - A operand (corpus): tiles of 32 rows stream HBM/L2 -> registers -> LDS, two tiles in
//     flight in registers while a third is consumed from LDS (a 1-tile look-ahead measured
//     latency-bound), and are shared by the 8 waves. LDS rows are XOR-swizzled
//     (chunk ^ (row & 15)) so the ds_read_b128 fragment reads are bank-conflict free.
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

template <int CHUNKS, bool SAMPLE>
__global__ __launch_bounds__(LS_GEMM_THREADS, 2) void ls_gemm_filter_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride,
    u64* __restrict__ queues, u32* __restrict__ counts, int cap, u32* __restrict__ overflow) {
    constexpr int KSTEPS = CHUNKS / 2;                    // 16 fp16 per MFMA k-step = 2 chunks
    constexpr int TILE_CHUNKS = LS_GEMM_TM * CHUNKS;      // 16-byte chunks per LDS tile
    constexpr int LOADS = TILE_CHUNKS / LS_GEMM_THREADS;  // 16-byte loads per thread per tile
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

    // B fragments: query j = qt*QT + wave*32 + (lane & 31); k-step kk -> chunk 2kk + (lane >> 5)
    const int qj = qt * LS_GEMM_QT + wave * 32 + (lane & 31);
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

    // private queue of this lane: entry e at queues[(b*THREADS + tid)*cap + e] (contiguous per
    // lane, so the select kernels read a queue as one coalesced run)
    u64* myq = queues + ((long long)blockIdx.x * LS_GEMM_THREADS + tid) * cap;
    int cnt = 0;

    // Corpus tiles travel HBM/L2 -> registers -> LDS. Two tiles are in flight in registers
    // (ra, rb) while a third is being consumed from LDS: the loads are ordinary global loads, so
    // hipcc counts them (vmcnt(N), not 0) next to the epilogue's queue stores.
    // Thread t fills LDS chunks Lc = j*THREADS + t of the tile: row r = Lc / CHUNKS, slot
    // sl = Lc % CHUNKS holds source chunk sl ^ (r & 15) (XOR swizzle: conflict-free ds_read_b128).
    auto load_tile = [&](int ti, u32x4 (&rg)[LOADS]) {
        const long long row0 = r_begin + (long long)ti * LS_GEMM_TM;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int Lc = j * LS_GEMM_THREADS + tid;
            const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
            long long row = row0 + r;
            row = row < n ? row : n - 1;
            rg[j] = corpus[row * CHUNKS + (sl ^ (r & 15))];
        }
    };
    auto write_tile = [&](int buf, const u32x4 (&rg)[LOADS]) {
        unsigned char* base = smem + (size_t)buf * TILE_CHUNKS * 16;
#pragma unroll
        for (int j = 0; j < LOADS; ++j)
            *reinterpret_cast<u32x4*>(base + (size_t)(j * LS_GEMM_THREADS + tid) * 16) = rg[j];
    };

    const int ar = lane & 31;  // A fragment: row ar of the tile, chunk 2kk + (lane >> 5)
    auto compute_tile = [&](int i, int buf) {
        const unsigned char* tb = smem + (size_t)buf * TILE_CHUNKS * 16 + (size_t)ar * CHUNKS * 16;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        // A fragments run LS_GEMM_APF k-steps ahead of the MFMA that consumes them
        u32x4 af[LS_GEMM_APF];
#pragma unroll
        for (int kk = 0; kk < LS_GEMM_APF && kk < KSTEPS; ++kk)
            af[kk] = *reinterpret_cast<const u32x4*>(tb + (((2 * kk + (lane >> 5)) ^ (ar & 15)) * 16));
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                __builtin_bit_cast(half8, af[kk % LS_GEMM_APF]), bq[kk], acc, 0, 0, 0);
            if (kk + LS_GEMM_APF < KSTEPS)
                af[kk % LS_GEMM_APF] = *reinterpret_cast<const u32x4*>(
                    tb + (((2 * (kk + LS_GEMM_APF) + (lane >> 5)) ^ (ar & 15)) * 16));
        }
        // pin the schedule hipcc would otherwise collapse to read->wait->mfma:
        // APF LDS reads up front, then one MFMA per LDS read (masks: 0x100 DS read, 0x008 MFMA)
        __builtin_amdgcn_sched_group_barrier(0x100, LS_GEMM_APF, 0);
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (kk + LS_GEMM_APF < KSTEPS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        // epilogue: acc[r] = <corpus row, query qj>, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
        const long long row0 = r_begin + (long long)(i * tile_stride) * LS_GEMM_TM + 4 * (lane >> 5);
        if (SAMPLE) {
            // sample pass: every score is kept (0 = no entry): 16 keys = one 128-byte line per
            // lane and tile, written as 8 x 16-byte stores
            u64 kk2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = row0 + (r & 3) + 8 * (r >> 2);
                kk2[r] = (qvalid && row < r_end) ? ls_make_key(acc[r], (u32)row) : 0ull;
            }
            if (cnt + 16 <= cap) {
                ulonglong2* dst = reinterpret_cast<ulonglong2*>(myq + cnt);
#pragma unroll
                for (int r = 0; r < 8; ++r) dst[r] = make_ulonglong2(kk2[2 * r], kk2[2 * r + 1]);
            }
            cnt += 16;
        } else {
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
        }
    };

    u32x4 ra[LOADS], rb[LOADS];
    if (nt > 0) load_tile(0, ra);
    if (nt > 1) load_tile(tile_stride, rb);
    if (nt > 0) write_tile(0, ra);
    if (nt > 2) load_tile(2 * tile_stride, ra);
    __syncthreads();
    // invariant at the top of iteration i: LDS[i&1] = tile i; tile i+1 in flight in rb (i even)
    // or ra (i odd); tile i+2 in flight in the other set
    for (int i = 0; i < nt; i += 2) {
        compute_tile(i, 0);
        if (i + 1 < nt) write_tile(1, rb);
        if (i + 3 < nt) load_tile((i + 3) * tile_stride, rb);
        __syncthreads();
        if (i + 1 < nt) {
            compute_tile(i + 1, 1);
            if (i + 2 < nt) write_tile(0, ra);
            if (i + 4 < nt) load_tile((i + 4) * tile_stride, ra);
            __syncthreads();
        }
    }
    counts[(long long)blockIdx.x * LS_GEMM_THREADS + tid] = (u32)(cnt < cap ? cnt : cap);
    if (cnt > cap) overflow[qj] = 1u;
}

int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, u64* d_queues, u32* d_counts,
                          int cap, u32* d_overflow, hipStream_t s) {
    const int nqt = (int)(nq_pad / LS_GEMM_QT);
    const dim3 grid((unsigned)(nsplits * nqt)), block(LS_GEMM_THREADS);
    const size_t smem = (size_t)2 * LS_GEMM_TM * g.chunks * 16;
#define LS_GEMM_CASE(C)                                                                          \
    if (g.chunks == C) {                                                                         \
        if (d_tau)                                                                               \
            hipLaunchKernelGGL((ls_gemm_filter_kernel<C, false>), grid, block, smem, s,           \
                               (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq,  \
                               nqt, d_tau, (long long)rows_per_split, tile_stride, d_queues,      \
                               d_counts, cap, d_overflow);                                       \
        else                                                                                     \
            hipLaunchKernelGGL((ls_gemm_filter_kernel<C, true>), grid, block, smem, s,            \
                               (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq,  \
                               nqt, d_tau, (long long)rows_per_split, tile_stride, d_queues,      \
                               d_counts, cap, d_overflow);                                       \
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
    const int qt = q / LS_GEMM_QT, w = (q % LS_GEMM_QT) / 32, l = q % 32;
    // entry list of this query: (split, half, e) -> flattened index space nsplits * 2 * cap
    const int total = nsplits * 2 * cap;
    // all loads are independent and issued together: first the queue lengths, then the entries
    u32 qc[LS_TAU_PER_THREAD];
    u64 ent[LS_TAU_PER_THREAD];
    u32 hi[LS_TAU_PER_THREAD];
#pragma unroll
    for (int j = 0; j < LS_TAU_PER_THREAD; ++j) {
        const int idx = tid + j * 256;
        const int sh = (idx < total ? idx : 0) / cap, half = sh & 1, split = sh >> 1;
        const int b = wg_index(split, qt, nqt);
        const int t = w * 64 + half * 32 + l;
        qc[j] = counts[(long long)b * LS_GEMM_THREADS + t];
        ent[j] = queues[((long long)b * LS_GEMM_THREADS + t) * cap + (idx < total ? idx % cap : 0)];
    }
#pragma unroll
    for (int j = 0; j < LS_TAU_PER_THREAD; ++j) {
        const int idx = tid + j * 256;
        hi[j] = (idx < total && (u32)(idx % cap) < qc[j]) ? (u32)(ent[j] >> 32) : 0u;
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
                       cap, nsplits, (int)(nq_pad / LS_GEMM_QT), (int)nq, k, d_tau);
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
    const int qt = q / LS_GEMM_QT, w = (q % LS_GEMM_QT) / 32, l = q % 32;
    if (tid == 0) nkeys = 0;
    __syncthreads();
    // walk the query's 2*nsplits queues in the flattened (queue, entry) index space: a wave
    // reads one queue's slots as one coalesced run. Loads are unconditional and issued in
    // batches of 8 so that the gather costs a few memory latencies, not one per queue.
    const int total = nsplits * 2 * cap;
    for (int i0 = 0; i0 < total; i0 += 256 * 8) {
        u32 qc[8];
        u64 ent[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = i0 + j * 256 + tid;
            const int ii = idx < total ? idx : 0;
            const int e = ii % cap, sh = ii / cap, half = sh & 1, split = sh >> 1;
            const int b = wg_index(split, qt, nqt);
            const int t = w * 64 + half * 32 + l;
            qc[j] = counts[(long long)b * LS_GEMM_THREADS + t];
            ent[j] = queues[((long long)b * LS_GEMM_THREADS + t) * cap + e];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = i0 + j * 256 + tid;
            if (idx < total && (u32)(idx % cap) < qc[j]) {
                const u32 pos = atomicAdd(&nkeys, 1u);
                if (pos < LS_BSEL_KEYS) keys[pos] = ent[j];
            }
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
                       d_counts, cap, nsplits, (int)(nq_pad / LS_GEMM_QT), k, (long long)base, d_overflow,
                       d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
