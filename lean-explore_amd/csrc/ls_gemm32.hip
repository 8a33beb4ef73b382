// ls_gemm32.hip — the batched path for an fp32 index (what the reference stores: fp32 rows,
// reference src/lean_explore/models/search_db.py:24-35; queried with k = 1000,
// search/engine.py:538): Q[nq, d] x Corpus^T[d, N] on the matrix cores in EXACT fp32 with the
// same fused top-k selection as ls_gemm.hip. Stands in for faiss `index.search(x, k)` with nq > 16.
//
// v_mfma_f32_16x16x4_f32: f32 in, f32 accumulate, bit-for-bit a k-ordered fmaf chain, at the
// fp32 vector rate (157 TFLOP/s peak, 1/16 of the fp16 MFMA rate) — so this kernel is bound by
// the matrix pipe at every batch size it serves and nothing else needs to be clever:
//   - one workgroup = 8 waves = 64 queries x TWO corpus slices; wave w owns query block w & 3
//     (16 queries) on slice half w >> 2, a 64-row tile of it at a time: 4 independent
//     accumulator chains (row blocks), 16 registers. A (query, slice, quarter) queue therefore
//     has exactly one producer lane, as in the fp16 kernel.
//   - K advances in chunks of 32 floats. Both operands of a chunk are staged through registers
//     into LDS (rows padded to 34 floats: conflict-free ds_read_b32 fragment reads), double
//     buffered, the next chunk's global loads in flight under the current chunk's 32 MFMAs.
//     The query chunk is re-read from L2 for every row tile (it is tiny); the corpus streams
//     from HBM once per query tile.
//   - epilogue per row tile: lane (query, quarter) holds 4 row scores per accumulator, exactly
//     the layout of the fp16 kernel, so the tau kernel and the select kernel of ls_gemm.hip are
//     shared as they are (same queues, same [query][slice][quarter] layout).
#include "ls_select_dev.h"

#include <hip/hip_ext.h>

typedef float f32x4v __attribute__((ext_vector_type(4)));

#define G32_TM 64       // corpus rows per tile (per slice half)
#define G32_BN 64       // queries per workgroup
#define G32_BK 32       // floats per K chunk
#define G32_LD 34       // padded LDS row length in floats (bank = (2*row + k) mod 32)
#define G32_THREADS 512

__host__ __device__ __forceinline__ long long g32_queue_id(int q, int split, int quarter, int nsplits) {
    return ((long long)q * nsplits + split) * 4 + quarter;
}
__device__ __forceinline__ void g32_top4_insert(float (&t)[4], float v) {
    float a = fmaxf(v, t[3]);
    t[3] = fminf(a, t[2]);
    a = fmaxf(a, t[2]);
    t[2] = fminf(a, t[1]);
    a = fmaxf(a, t[1]);
    t[1] = fminf(a, t[0]);
    t[0] = fmaxf(a, t[0]);
}
__device__ __forceinline__ uint4 g32_top4_keys(const float (&t)[4]) {  // 0 = "no sample"
    return make_uint4(t[0] == -FLT_MAX ? 0u : ls_ord(t[0]), t[1] == -FLT_MAX ? 0u : ls_ord(t[1]),
                      t[2] == -FLT_MAX ? 0u : ls_ord(t[2]), t[3] == -FLT_MAX ? 0u : ls_ord(t[3]));
}

// corpus: stored rows [n + pad][d_pad] fp32; qp: prepared queries [nq_pad][d_pad] fp32 (normalised
// if asked, zero padded). Workgroup b -> (slice pair, query tile): the query tiles of one pair are
// consecutive on one XCD (b % 8), as in ls_gemm.hip. nsplits = 2 * (gridDim.x / nqt) slices.
template <bool SAMPLE>
__global__ __launch_bounds__(G32_THREADS, 2) void ls_gemm32_filter_kernel(
    const float* __restrict__ corpus, long long n, const float* __restrict__ qp, int nq, int nqt,
    int d_pad, const float* __restrict__ tau, long long rows_per_split, int tile_stride,
    uint2* __restrict__ queues, u32* __restrict__ counts, u32* __restrict__ overflow,
    u32* __restrict__ sample_top) {
    __shared__ __attribute__((aligned(16))) float sA[2][2 * G32_TM * G32_LD];  // [buf][half][row]
    __shared__ __attribute__((aligned(16))) float sB[2][G32_BN * G32_LD];
    constexpr int cap = LS_GEMM_QCAP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qd = lane >> 4, li = lane & 15;
    const int g = wave & 3, half = wave >> 2;  // query block, slice half
    const int b = (int)blockIdx.x, xcd = b & 7, j0 = b >> 3;
    const int pair = (j0 / nqt) * 8 + xcd, qt = j0 % nqt;
    const int nsplits = 2 * ((int)gridDim.x / nqt);
    const int split = pair * 2 + half;  // this wave's slice
    // tiles of the two halves advance together: the loop runs for the longer one
    auto slice_tiles = [&](int sp, long long* rb, long long* re) {
        *rb = (long long)sp * rows_per_split;
        *re = *rb + rows_per_split;
        if (*re > n) *re = n;
        const int all = *rb < *re ? (int)((*re - *rb + G32_TM - 1) / G32_TM) : 0;
        return (all + tile_stride - 1) / tile_stride;
    };
    long long rbeg[2], rend[2];
    const int nt0 = slice_tiles(pair * 2, &rbeg[0], &rend[0]);
    const int nt1 = slice_tiles(pair * 2 + 1, &rbeg[1], &rend[1]);
    const int nt = nt0 > nt1 ? nt0 : nt1;
    const long long r_begin = rbeg[half], r_end = rend[half];
    const int nchunk = d_pad / G32_BK;

    const int qj = qt * G32_BN + g * 16 + li;  // this lane's query
    const float tauv = SAMPLE ? 0.0f : tau[qj];
    const long long qid = g32_queue_id(qj, split, qd, nsplits);
    uint2* myq = queues + qid * cap;
    int cnt = 0;
    float top[4] = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};

    // staging map: A chunk = 2 halves x 64 rows x 8 float4 -> thread t loads (row t/8 of half h,
    // float4 t%8), h = 0, 1; B chunk = 64 queries x 8 float4 -> (query t/8, float4 t%8).
    // A tile of a half that has run out of rows is read from the zero pad rows behind the
    // corpus (clamped), its scores are masked by the row bound below.
    const int s_row = tid >> 3, s_c4 = tid & 7;
    float4 ra[2], rq;
    auto load_chunk = [&](int ti, int c) {  // ti: tile index inside the slices
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            long long row = rbeg[h] + (long long)ti * G32_TM + s_row;
            row = row < n + LS_CORPUS_PAD_ROWS ? row : n;  // n .. n+pad-1 are zero rows
            ra[h] = *reinterpret_cast<const float4*>(corpus + row * d_pad + c * G32_BK + s_c4 * 4);
        }
        rq = *reinterpret_cast<const float4*>(qp + (long long)(qt * G32_BN + s_row) * d_pad +
                                              c * G32_BK + s_c4 * 4);
    };
    auto store_chunk = [&](int buf) {  // rows are 34 floats apart: 8-byte aligned pairs
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* p = &sA[buf][(h * G32_TM + s_row) * G32_LD + s_c4 * 4];
            *reinterpret_cast<float2*>(p) = make_float2(ra[h].x, ra[h].y);
            *reinterpret_cast<float2*>(p + 2) = make_float2(ra[h].z, ra[h].w);
        }
        float* p = &sB[buf][s_row * G32_LD + s_c4 * 4];
        *reinterpret_cast<float2*>(p) = make_float2(rq.x, rq.y);
        *reinterpret_cast<float2*>(p + 2) = make_float2(rq.z, rq.w);
    };

    // The append is the hot slow path: clamped slot, one store (see ls_gemm.hip). Padded queries
    // carry tau = FLT_MAX; rows past the slice end are dropped here (two slices share a tile
    // loop, so a half may run past its own end), rows past n by the select kernel as well.
    auto check1 = [&](float s, int lrow) {
        const bool inside = r_begin + lrow < r_end;
        if (SAMPLE) {
            g32_top4_insert(top, (qj < nq && inside) ? s : -FLT_MAX);
        } else if (s >= tauv && inside) {
            const int slot = cnt < cap ? cnt : cap - 1;
            myq[slot] = make_uint2(__float_as_uint(s), (u32)lrow);
            ++cnt;
        }
    };

    f32x4v acc[4];
    const int steps = nt * nchunk;  // (tile, chunk) pairs, one continuous stream
    if (steps > 0) {
        load_chunk(0, 0);
        store_chunk(0);
    }
    __syncthreads();
    int ti = 0, c = 0;
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 1;
        int ti_n = ti, c_n = c + 1;
        if (c_n == nchunk) {
            c_n = 0;
            ++ti_n;
        }
        const bool more = st + 1 < steps;
        if (more) load_chunk(ti_n * tile_stride, c_n);  // in flight under this chunk's MFMAs
        if (c == 0) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
        // A[row = rb*16 + li][k = 4ks + qd], B[k = 4ks + qd][query = li]
        const float* a_row = &sA[buf][(half * G32_TM + li) * G32_LD + qd];
        const float* b_row = &sB[buf][(g * 16 + li) * G32_LD + qd];
#pragma unroll
        for (int ks = 0; ks < G32_BK / 4; ++ks) {
            const float bv = b_row[ks * 4];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
                acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_row[rb * 16 * G32_LD + ks * 4], bv,
                                                               acc[rb], 0, 0, 0);
        }
        if (c == nchunk - 1) {  // the tile's scores are complete: filter them
            const int lrow0 = (ti * tile_stride) * G32_TM + 4 * qd;
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) check1(acc[rb][reg], lrow0 + rb * 16 + reg);
        }
        if (more) store_chunk(buf ^ 1);
        __syncthreads();
        ti = ti_n;
        c = c_n;
    }
    if (SAMPLE) {
        reinterpret_cast<uint4*>(sample_top)[qid] = g32_top4_keys(top);
    } else {
        counts[qid] = (u32)(cnt < cap ? cnt : cap);
        if (cnt > cap) overflow[qj] = 1u;
    }
}

int ls_launch_gemm32_filter(const void* d_corpus, int64_t n, const ls_geom& g, const float* d_qp,
                            int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                            int64_t rows_per_split, int tile_stride, const ls_gemm_bufs& b,
                            hipStream_t s, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int nqt = (int)(nq_pad / G32_BN);
    const dim3 grid((unsigned)(nsplits / 2 * nqt));
    if (d_tau)  // (the events ride on the dispatch: no extra packet on the stream)
        hipExtLaunchKernelGGL((ls_gemm32_filter_kernel<false>), grid, dim3(G32_THREADS), 0, s, ev_start,
                           ev_stop, 0, (const float*)d_corpus, (long long)n, d_qp, (int)nq, nqt, g.d_pad, d_tau,
                           (long long)rows_per_split, tile_stride, (uint2*)b.d_queues, b.d_counts,
                           b.d_overflow, b.d_sample_top);
    else
        hipLaunchKernelGGL((ls_gemm32_filter_kernel<true>), grid, dim3(G32_THREADS), 0, s,
                           (const float*)d_corpus, (long long)n, d_qp, (int)nq, nqt, g.d_pad, d_tau,
                           (long long)rows_per_split, tile_stride, (uint2*)b.d_queues, b.d_counts,
                           b.d_overflow, b.d_sample_top);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- queries -> padded fp32 [nq_pad, d_pad], normalised if asked; clears the call's repair flags
// and keeps the raw copy for repairs (the fp32 counterpart of ls_prep_f16_kernel) -----------------
__global__ __launch_bounds__(256) void ls_prep_f32_kernel(const float* __restrict__ qin,
                                                          float* __restrict__ qout,
                                                          float* __restrict__ qkeep, int nq,
                                                          int nq_pad, int d, int d_pad,
                                                          int normalize, u32* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq_pad) return;
    if (lane == 0) overflow[qi] = 0u;
    const float* src = qin + (long long)qi * d;
    const bool live = qi < nq;
    float inv = 1.0f;
    if (normalize && live) {
        const float ss = ls_wave_sumsq(src, d, lane);
        if (ss > 0.0f) inv = 1.0f / sqrtf(ss);
    }
    for (int j = lane; j < d_pad; j += 64) {
        const float v = (live && j < d) ? src[j] : 0.0f;
        if (live && j < d && qkeep) qkeep[(long long)qi * d + j] = v;
        qout[(long long)qi * d_pad + j] = normalize ? v * inv : v;
    }
}

int ls_launch_prep_f32(const float* d_q, float* d_qp, float* d_qkeep, int64_t nq, int64_t nq_pad,
                       const ls_geom& g, bool normalize, u32* d_overflow, hipStream_t s) {
    hipLaunchKernelGGL(ls_prep_f32_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, s, d_q,
                       d_qp, d_qkeep, (int)nq, (int)nq_pad, g.d, g.d_pad, normalize ? 1 : 0,
                       d_overflow);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
