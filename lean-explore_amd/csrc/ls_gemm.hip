// ls_gemm.hip — the batched path: Q[nq, d] x Corpus^T[d, N] on the matrix cores with the
// top-k selection fused into the epilogue (BASELINE config 3: N=200k, d=384 fp16, nq=1024,
// k=100; config 4: d=768 fp16, nq=256). Stands in for faiss `index.search(x, k)` with a large nq
// (reference src/lean_explore/search/engine.py:250; the reference itself only ever sends nq=1).
//
// Why fused: the score matrix is nq*N fp32 = 819 MB for config 3; writing and re-reading it
// would cost more HBM time than the whole MFMA budget, so scores never leave registers.
//
// ls_gemm_filter_kernel — one workgroup = 8 waves x (16*QG queries) x one corpus slice
//   - v_mfma_f32_16x16x32_f16. A wave owns QG groups of 16 queries (QG = 2 for stored rows
//     <= 1 KiB, else 1): their fp16 fragments stay in VGPRs for the whole slice, so B costs no
//     LDS or HBM traffic in the loop.
//   - A operand (corpus): tiles of TM rows (64, or 32 for long rows) stream HBM/L2 -> LDS by DMA
//     (global_load_lds, 16 B/lane, double buffered) and are shared by the 8 waves. LDS rows are
//     XOR-swizzled on the SOURCE address (chunk ^ (row & 15)): conflict-free ds_read_b128.
//   - a tile is TM/16 row blocks x QG query groups = up to 8 INDEPENDENT accumulator chains per
//     wave, each A fragment feeding QG MFMAs: the matrix pipe never waits on a dependent result
//     (tools/mfma_ub.hip: 4+ chains with interleaved LDS reads run at the pipe's ceiling).
//   - epilogue: lane (query, quarter) holds 4 row scores per accumulator. The PREVIOUS tile's
//     accumulators are filtered a few elements per k-step, in the shadow of the matrix pipe: a
//     score >= tau[query] is appended to the lane's private queue in HBM (no atomics).
//   - workgroups that share a corpus slice sit on the same XCD (block % 8) so the slice is
//     fetched from HBM once and served to the other query tiles from that XCD's L2.
//
// Phases (ls_api.hip orchestrates): sample pass (two tiles of every slice; each lane keeps its 4
// best sample scores in registers) -> tau kernel (j-th best sample score per query) -> full pass
// with tau -> select kernel (exact top-k of each query's queues, verifies >= k candidates). A
// flagged query (queue overflow / too few candidates) is re-run by the exact per-query scan
// path, so the result is always exact.
#include "ls_select_dev.h"

#ifndef LS_GEMM_LOADERS
#define LS_GEMM_LOADERS 2  // loader waves per workgroup for long rows (0 = MFMA waves issue the DMA)
#endif
#ifndef LS_GEMM_PF
#define LS_GEMM_PF 1
#endif
#ifndef LS_GEMM_CHECK_NUM
#define LS_GEMM_CHECK_NUM 1
#define LS_GEMM_CHECK_DEN 1
#endif
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// ---- queries -> fp16 MFMA B fragments, normalised if asked, zero padded ---------------------------
// Output layout = the order in which ls_gemm_filter_kernel consumes it: the 16-byte chunk c of
// query q (tile qt, wave w, group g2, li = q % 16) is fragment (qt, w, g2, kk = c / 4), lane
// (c % 4) * 16 + li. One wave load of a B fragment is then one contiguous 1 KiB read.
__device__ __forceinline__ long long qfrag_chunk(int q, int c, int QG, int KS) {
    const int QPW = 16 * QG, QT = LS_GEMM_WAVES * QPW;
    const int qt = q / QT, w = (q % QT) / QPW, g2 = (q % QPW) / 16, li = q % 16;
    return ((((long long)(qt * LS_GEMM_WAVES + w) * QG + g2) * KS + (c >> 2)) << 6) + ((c & 3) << 4) + li;
}
// One wave per query (4 queries per block): the norm is a wave reduction, every lane converts
// and stores whole 16-byte chunks.
__global__ __launch_bounds__(256) void ls_prep_f16_kernel(const float* __restrict__ qin,
                                                          u32x4* __restrict__ qout, int nq,
                                                          int nq_pad, int d, int d_pad, int QG,
                                                          int normalize, u32* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (qi >= nq_pad) return;
    const int KS = d_pad / 32, chunks = d_pad / 8;
    if (lane == 0) overflow[qi] = 0u;  // per-query repair flag, cleared for this batch
    const float* src = qin + (long long)qi * d;
    const bool live = qi < nq;
    float inv = 1.0f;
    if (normalize && live) {
        float ss = 0.0f;
        for (int j = lane; j < d; j += 64) ss = fmaf(src[j], src[j], ss);
        for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (ss > 0.0f) inv = 1.0f / sqrtf(ss);
    }
    for (int c = lane; c < chunks; c += 64) {
        half8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = c * 8 + e;
            const float v = (live && j < d) ? (normalize ? src[j] * inv : src[j]) : 0.0f;
            h[e] = (_Float16)v;
        }
        qout[qfrag_chunk(qi, c, QG, KS)] = __builtin_bit_cast(u32x4, h);
    }
}

int ls_gemm_qg(const ls_geom& g);
int ls_launch_prep_f16(const float* d_q, void* d_qh, int64_t nq, int64_t nq_pad, const ls_geom& g,
                       bool normalize, u32* d_overflow, hipStream_t s) {
    hipLaunchKernelGGL(ls_prep_f16_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, s, d_q,
                       (u32x4*)d_qh, (int)nq, (int)nq_pad, g.d, g.d_pad, ls_gemm_qg(g),
                       normalize ? 1 : 0, d_overflow);
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
// Query q of a launch with QG groups per wave lives in tile qt = q / (128*QG), wave
// w = (q % (128*QG)) / (16*QG), group qg = (q % (16*QG)) / 16, li = q % 16; in every workgroup of
// its tile 4 lanes (quarter = 0..3) own a private queue for it. Queues, their lengths and the
// sample tops are stored QUERY-MAJOR, [query][slice][quarter]: everything the tau and select
// kernels read for one query is contiguous (they are the consumers with a dependent round trip;
// the producers' scattered 16-byte stores cost nothing).
__host__ __device__ __forceinline__ long long queue_id(int q, int split, int quarter, int nsplits) {
    return ((long long)q * nsplits + split) * 4 + quarter;
}

// top-4 of a lane's sample scores, descending: branch-free insert on the floats themselves
// (v_max/v_min pairs); they become ord() keys once, when the kernel stores them
__device__ __forceinline__ void top4_insert(float (&t)[4], float v) {
    float a = fmaxf(v, t[3]);
    t[3] = fminf(a, t[2]);
    a = fmaxf(a, t[2]);
    t[2] = fminf(a, t[1]);
    a = fmaxf(a, t[1]);
    t[1] = fminf(a, t[0]);
    t[0] = fmaxf(a, t[0]);
}
__device__ __forceinline__ uint4 top4_keys(const float (&t)[4]) {  // 0 = "no sample"
    return make_uint4(t[0] == -FLT_MAX ? 0u : ls_ord(t[0]), t[1] == -FLT_MAX ? 0u : ls_ord(t[1]),
                      t[2] == -FLT_MAX ? 0u : ls_ord(t[2]), t[3] == -FLT_MAX ? 0u : ls_ord(t[3]));
}

// Long rows, full pass: two extra LOADER waves per workgroup issue the tile DMA. An LDS-DMA piece
// costs the issuing wave ~150 cycles (tools/pmc_c4_sq.sh), six pieces per wave are 900 of a
// long-row tile's ~2400 cycles; with dedicated loaders the eight MFMA waves never issue it.
// (10 waves = 3 on two of the SIMDs: needs <= 168 registers, which only the long-row kernel,
// with one query group per wave, has.)
__host__ __device__ constexpr int ls_gemm_loaders(int chunks, bool sample) {
    return (LS_GEMM_LOADERS && chunks == 96 && !sample) ? LS_GEMM_LOADERS : 0;
}
template <int CHUNKS, int QG, bool SAMPLE>
__global__ __launch_bounds__(LS_GEMM_THREADS + 64 * ls_gemm_loaders(CHUNKS, SAMPLE),
                             ls_gemm_loaders(CHUNKS, SAMPLE) ? 3 : LS_GEMM_WAVES_PER_SIMD)
void ls_gemm_filter_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride,
    u64* __restrict__ queues, u32* __restrict__ counts, int cap, u32* __restrict__ overflow,
    u32* __restrict__ sample_top) {
    constexpr int TM = CHUNKS <= 64 ? LS_GEMM_TM_SHORT : 32;  // corpus rows per LDS tile
    constexpr int NRB = TM / 16;                          // 16-row MFMA blocks per tile
    constexpr int KS = CHUNKS / 4;                        // k-steps: 32 fp16 = 4 chunks each
    constexpr int QPW = 16 * QG;                          // queries per wave
    constexpr int NV = NRB * QG * 4;                      // filter values per lane per tile
    // The filter's queue appends are global stores, and the tile hand-over barrier drains vmcnt:
    // a store issued in a tile's last k-steps would hold the barrier for its whole round trip.
    // So the previous tile is filtered in the FIRST LS_GEMM_CHECK_NUM/DEN of the k-steps only.
    constexpr int CKS = (KS * LS_GEMM_CHECK_NUM + LS_GEMM_CHECK_DEN - 1) / LS_GEMM_CHECK_DEN;
    constexpr int CPK = (NV + CKS - 1) / CKS;             // checks interleaved per k-step
    constexpr int ROW_BYTES = CHUNKS * 16;
    constexpr int TILE_CHUNKS = TM * CHUNKS;
    constexpr int TILE_BYTES = TILE_CHUNKS * 16;
    constexpr int LOADS = TILE_CHUNKS / LS_GEMM_THREADS;  // 16-byte DMA loads per thread per tile
    static_assert(TILE_CHUNKS % LS_GEMM_THREADS == 0, "tile must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 tiles

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar DMA addressing
    const int qd = lane >> 4, li = lane & 15;  // quarter (k-chunk / row group), index in group
    int split, qt;
    wg_coords((int)blockIdx.x, nqt, &split, &qt);
    const int nsplits = (int)gridDim.x / nqt;
    const long long r_begin = (long long)split * rows_per_split;
    long long r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;
    const int ntiles_all = r_begin < r_end ? (int)((r_end - r_begin + TM - 1) / TM) : 0;
#ifdef LS_GEMM_ABL_NOLOOP  // timing ablation: prologue + epilogue only (cap is never negative)
    const int nt = cap < 0 ? 1 : 0;
#elif defined(LS_GEMM_ABL_HALF)  // timing ablation: half the tiles
    const int nt = ((ntiles_all + tile_stride - 1) / tile_stride) / 2;
#else
    const int nt = (ntiles_all + tile_stride - 1) / tile_stride;  // tiles this launch visits
#endif

    constexpr int NLOAD = ls_gemm_loaders(CHUNKS, SAMPLE);
    if (NLOAD > 0 && wave >= LS_GEMM_WAVES) {  // ---- loader waves: the tile DMA and nothing else
        const int lw = wave - LS_GEMM_WAVES;
        constexpr int PPL = TILE_CHUNKS / 64 / (NLOAD > 0 ? NLOAD : 1);  // 1 KiB pieces per loader
        int lgoff[PPL];
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
            const int Lc = (lw + j * NLOAD) * 64 + lane;
            const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
            lgoff[j] = r * CHUNKS + (sl ^ (r & 15));
        }
        auto lstage = [&](int ti, int buf) {
            const u32x4* base = corpus + (r_begin + (long long)ti * TM) * CHUNKS;
#pragma unroll
            for (int j = 0; j < PPL; ++j) {
                unsigned char* dst = smem + buf * TILE_BYTES + (lw + j * NLOAD) * 1024;
                __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + lgoff[j]), (lds_ptr_t)dst, 16, 0, 0);
            }
        };
        // Ring of three tile buffers, two tiles ahead. Same barrier sequence as the MFMA waves
        // (one before tile 0, one per tile). The loader has no other vector-memory traffic, so
        // "tile i+1 has landed, tile i+2 may still fly" is exactly vmcnt(PPL): written by hand,
        // with the bare barrier (a __syncthreads() would drain everything).
        static_assert(NLOAD == 0 || PPL < 64, "vmcnt immediate");
        constexpr int W = NLOAD > 0 ? PPL : 0;
        auto hand_over = [&](bool newer_in_flight) {
            asm volatile("" ::: "memory");
            if (newer_in_flight)
                __builtin_amdgcn_s_waitcnt(0x0F70 | (W & 15) | ((W >> 4) << 14));
            else
                __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (nt > 0) lstage(0, 0);
        if (nt > 1) lstage(tile_stride, 1);
        hand_over(nt > 1);
        int b = 0;  // ring slot of tile i
        for (int i = 0; i < nt; ++i) {
            const int b2 = b == 0 ? 2 : b - 1;  // slot of tile i+2 == slot of tile i-1 (consumed)
            if (i + 2 < nt) lstage((i + 2) * tile_stride, b2);
            hand_over(i + 2 < nt);
            b = b == 2 ? 0 : b + 1;
        }
        return;
    }

    // ---- corpus tiles: HBM/L2 -> LDS by DMA (global_load_lds, 16 B per lane), double buffered --
    // Wave w, load j fills the 64 consecutive LDS chunks starting at (w*LOADS + j)*64: chunk Lc
    // is tile row r = Lc / CHUNKS, slot sl = Lc % CHUNKS and receives SOURCE chunk sl ^ (r & 15)
    // (the DMA destination is lane-linear, so the swizzle goes on the source address). The HBM
    // copy is padded with zero rows past n (ls_api.hip): no clamping. A tile takes ~3 us to
    // consume, longer than the DMA's flight, so one tile of look-ahead suffices.
    int goff[LOADS];
#pragma unroll
    for (int j = 0; j < LOADS; ++j) {
        const int Lc = (wave * LOADS + j) * 64 + lane;
        const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
        goff[j] = r * CHUNKS + (sl ^ (r & 15));
    }
    auto stage = [&](int ti, int buf) {
        if (NLOAD > 0) return;  // the loader waves do it
        const u32x4* base = corpus + (r_begin + (long long)ti * TM) * CHUNKS;
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            unsigned char* dst = smem + buf * TILE_BYTES + (wave * LOADS + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + goff[j]), (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // The first tile(s) are requested BEFORE the query fragments: the HBM round trip of tile 0
    // then overlaps the (L2-resident) query loads instead of queueing behind them.
    constexpr int NB_ALL = (144 * 1024) / TILE_BYTES;  // tiles that fit in the 144 KiB carve-out
    const bool sample_upfront = SAMPLE && nt > 0 && nt <= NB_ALL && nt <= 3;
    if (nt > 0) stage(0, 0);
    if (sample_upfront) {
        if (nt > 1) stage(tile_stride, 1);
        if (nt > 2) stage(2 * tile_stride, 2);
    }

    // B fragments: group qg holds query qt*8*QPW + wave*QPW + qg*16 + li; k-step kk -> chunk 4kk+qd
    half8 bq[QG][KS];
    int qj[QG];
    bool qvalid[QG];
    float tauv[QG];
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2) {
        qj[g2] = (qt * LS_GEMM_WAVES + wave) * QPW + g2 * 16 + li;
        // fragment-ordered by ls_prep_f16_kernel: each load below is one contiguous KiB per wave
        const u32x4* qfrag = qh + ((((long long)(qt * LS_GEMM_WAVES + wave) * QG + g2) * KS) << 6) + lane;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
#ifdef LS_GEMM_ABL_NOBQ  // timing ablation: no query-fragment loads
            const u32x4 v = {(u32)kk, (u32)cap, (u32)nq, (u32)lane};
#else
            const u32x4 v = qfrag[kk << 6];
#endif
            bq[g2][kk] = __builtin_bit_cast(half8, v);
        }
        qvalid[g2] = qj[g2] < nq;
        tauv[g2] = SAMPLE ? 0.0f : tau[qj[g2]];
    }

    // private queues of this lane (one per query group), contiguous per lane
    // entry = {score bits, row relative to the slice}; the select kernel turns it into a key
    uint2* myq[QG];
    int cnt[QG];
    float top[QG][4];
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2) {
        myq[g2] = reinterpret_cast<uint2*>(queues) + queue_id(qj[g2], split, qd, nsplits) * cap;
        cnt[g2] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) top[g2][e] = -FLT_MAX;
    }

    // A fragment of k-step kk, row block rb: tile row rb*16 + li, chunk (4kk + qd) ^ li.
    // (4kk + qd) & 15 takes 4 values per lane: 4 precomputed byte offsets + immediates.
    int lo4[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) lo4[m] = li * ROW_BYTES + (((4 * m + qd) ^ li) * 16);
    auto a_frag = [&](int bufoff, int rb, int kk) -> half8 {  // bufoff: byte offset of the tile
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + lo4[kk & 3] + bufoff +
                                                        rb * 16 * ROW_BYTES + (kk >> 2) * 256);
        return __builtin_bit_cast(half8, v);
    };

    // ---- one element of a finished tile: e -> (row block, query group, register) ------------------
    // The append is the hot slow path (a wave enters it for ~1 check in 5): no key building, no
    // bounds logic here. Padded queries carry tau = FLT_MAX and never pass; zero-pad rows past n
    // are dropped by the select kernel; a full queue keeps overwriting its last slot while the
    // count runs on, which is how the overflow is seen at the end.
    auto check = [&](const f32x4v (&acc)[NRB][QG], int e, int lrow0) {
        const int rb = e / (QG * 4), g2 = (e / 4) % QG, reg = e % 4;
        const float s = acc[rb][g2][reg];
        const int lrow = lrow0 + rb * 16 + reg;  // lrow0 already includes 4*qd
        if (SAMPLE) {
            // NaN never enters (fmaxf/fminf drop it); padded queries and zero-pad rows are masked
            top4_insert(top[g2], (qvalid[g2] && r_begin + lrow < r_end) ? s : -FLT_MAX);
        } else if (s >= tauv[g2]) {
            const int slot = cnt[g2] < cap ? cnt[g2] : cap - 1;
            myq[g2][slot] = make_uint2(__float_as_uint(s), (u32)lrow);
            ++cnt[g2];
        }
    };

    // One tile: NRB*QG independent accumulator chains advance together, one k-step at a time;
    // the PREVIOUS tile's accumulators are filtered CPK elements per k-step.
    auto run_tile = [&](f32x4v (&cur)[NRB][QG], const f32x4v (&prev)[NRB][QG], bool have_prev,
                        int prev_row0, int bufoff) {
        // A fragments are read LS_GEMM_PF k-steps ahead of their MFMAs
        constexpr int PF = LS_GEMM_PF, NA = PF + 1;
        half8 a[NA][NRB];
#pragma unroll
        for (int p0 = 0; p0 < PF; ++p0) {
            if (p0 < KS) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) a[p0][rb] = a_frag(bufoff, rb, p0);
            }
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + PF < KS) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) a[(kk + PF) % NA][rb] = a_frag(bufoff, rb, kk + PF);
            }
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int g2 = 0; g2 < QG; ++g2) {
                    f32x4v c;
                    if (kk == 0) {
                        c[0] = 0.0f; c[1] = 0.0f; c[2] = 0.0f; c[3] = 0.0f;
                    } else {
                        c = cur[rb][g2];
                    }
                    cur[rb][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk % NA][rb], bq[g2][kk],
                                                                        c, 0, 0, 0);
                }
            }
#ifdef LS_GEMM_ABL_NOCHECK  // timing ablation: a branch-free sink instead of the filter
            if (have_prev) {
#pragma unroll
                for (int c2 = 0; c2 < CPK; ++c2)
                    if (kk * CPK + c2 < NV) {
                        const int e = kk * CPK + c2;
                        tauv[0] += prev[e / (QG * 4)][(e / 4) % QG][e % 4];
                    }
            }
            if (false)
#else
            if (have_prev)
#endif
            {
#pragma unroll
                for (int c2 = 0; c2 < CPK; ++c2)
                    if (kk * CPK + c2 < NV) check(prev, kk * CPK + c2, prev_row0);
            }
        }
    };

    f32x4v accA[NRB][QG], accB[NRB][QG];  // alternate between consecutive tiles
    auto tile_row0 = [&](int i) { return (i * tile_stride) * TM + 4 * qd; };  // slice-relative
    // The sample pass visits only a few tiles, so their DMA latencies would be paid one by one:
    // when all of them fit in LDS together they are fetched up front and consumed back to back.
    if (sample_upfront) {
        __syncthreads();
        run_tile(accA, accB, false, 0, 0);
        if (nt > 1) run_tile(accB, accA, true, tile_row0(0), TILE_BYTES);
        if (nt > 2) run_tile(accA, accB, true, tile_row0(1), 2 * TILE_BYTES);
        const int row0 = tile_row0(nt - 1);
        if ((nt - 1) & 1) {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(accB, e, row0);
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(accA, e, row0);
        }
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2)
            reinterpret_cast<uint4*>(sample_top)[queue_id(qj[g2], split, qd, nsplits)] =
                top4_keys(top[g2]);
        return;
    }
    __syncthreads();  // the compiler drains the DMA (vmcnt(0)) before the barrier
#ifdef LS_GEMM_ABL_NOSTAGE  // timing ablation: no tile hand-over (results are garbage)
#define LS_STAGE(t, b) if (cap < 0) stage(t, b)
#define LS_TILE_BARRIER() if (cap < 0) __syncthreads()
#elif defined(LS_GEMM_ABL_NODMA)  // timing ablation: barriers but no tile traffic
#define LS_STAGE(t, b) if (cap < 0) stage(t, b)
#define LS_TILE_BARRIER() __syncthreads()
#elif defined(LS_GEMM_ABL_NOBARRIER)  // timing ablation: tile traffic but no barrier (racy)
#define LS_STAGE(t, b) stage(t, b)
#define LS_TILE_BARRIER() if (cap < 0) __syncthreads()
#else
#define LS_STAGE(t, b) stage(t, b)
#define LS_TILE_BARRIER() __syncthreads()
#endif
    if (NLOAD > 0) {  // loader waves fill a ring of three tiles, two ahead; this wave only computes
        int b = 0;
        for (int i = 0; i < nt; i += 2) {
            run_tile(accA, accB, i > 0, tile_row0(i - 1), b * TILE_BYTES);
            __syncthreads();
            b = b == 2 ? 0 : b + 1;
            if (i + 1 < nt) {
                run_tile(accB, accA, true, tile_row0(i), b * TILE_BYTES);
                __syncthreads();
                b = b == 2 ? 0 : b + 1;
            }
        }
    }
    for (int i = 0; i < (NLOAD > 0 ? 0 : nt); i += 2) {
        if (i + 1 < nt) LS_STAGE((i + 1) * tile_stride, 1);
        run_tile(accA, accB, i > 0, tile_row0(i - 1), 0);
        LS_TILE_BARRIER();
        if (i + 1 < nt) {
            if (i + 2 < nt) LS_STAGE((i + 2) * tile_stride, 0);
            run_tile(accB, accA, true, tile_row0(i), TILE_BYTES);
            LS_TILE_BARRIER();
        }
    }
    if (nt > 0) {  // the last tile still has to be filtered
        const int row0 = tile_row0(nt - 1);
        if ((nt - 1) & 1) {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(accB, e, row0);
        } else {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(accA, e, row0);
        }
    }
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2) {
        const long long qid = queue_id(qj[g2], split, qd, nsplits);
        if (SAMPLE) {
            reinterpret_cast<uint4*>(sample_top)[qid] = top4_keys(top[g2]);
        } else {
            counts[qid] = (u32)(cnt[g2] < cap ? cnt[g2] : cap);
#ifdef LS_GEMM_ABL_NOCHECK
            if (tauv[0] == 12345.0f) counts[qid] = 7u;
#endif
            if (cnt[g2] > cap) overflow[qj[g2]] = 1u;
        }
    }
}

int ls_gemm_qg(const ls_geom& g) { return g.chunks <= 64 ? 2 : 1; }
int ls_gemm_tile_rows(const ls_geom& g) { return g.chunks <= 64 ? LS_GEMM_TM_SHORT : 32; }

int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, u64* d_queues, u32* d_counts,
                          int cap, u32* d_overflow, u32* d_sample_top, hipStream_t s) {
    const int QG = ls_gemm_qg(g);
    const int nqt = (int)(nq_pad / (LS_GEMM_WAVES * 16 * QG));
    const dim3 grid((unsigned)(nsplits * nqt));
    // two tile buffers; the sample pass takes a third when it fits (all its tiles up front)
    const size_t tile_bytes = (size_t)ls_gemm_tile_rows(g) * g.chunks * 16;
    const size_t smem = tile_bytes * (((!d_tau || ls_gemm_loaders(g.chunks, false)) &&
                                       3 * tile_bytes <= 144 * 1024) ? 3 : 2);
#define LS_GEMM_LAUNCH(C, Q, SMP)                                                                 \
    {                                                                                             \
        auto kern = ls_gemm_filter_kernel<C, Q, SMP>;                                             \
        static bool attr_set = false;                                                             \
        if (!attr_set) {                                                                          \
            LS_HIP(hipFuncSetAttribute((const void*)kern,                                         \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));  \
            attr_set = true;                                                                      \
        }                                                                                         \
        const dim3 blk(LS_GEMM_THREADS + 64 * ls_gemm_loaders(C, SMP));                           \
        hipLaunchKernelGGL(kern, grid, blk, smem, s, (const u32x4*)d_corpus, (long long)n,        \
                           (const u32x4*)d_qh, (int)nq, nqt, d_tau, (long long)rows_per_split,     \
                           tile_stride, d_queues, d_counts, cap, d_overflow, d_sample_top);       \
        LS_HIP(hipGetLastError());                                                                \
        return LS_OK;                                                                             \
    }
#define LS_GEMM_CASE(C, Q)                     \
    if (g.chunks == C) {                       \
        if (d_tau) LS_GEMM_LAUNCH(C, Q, false) \
        else LS_GEMM_LAUNCH(C, Q, true)        \
    }
    LS_GEMM_CASE(16, 2) LS_GEMM_CASE(32, 2) LS_GEMM_CASE(48, 2) LS_GEMM_CASE(64, 2)
    LS_GEMM_CASE(96, 1) LS_GEMM_CASE(128, 1)
#undef LS_GEMM_CASE
#undef LS_GEMM_LAUNCH
    ls_set_error("batched path: unsupported row geometry (%d chunks)", g.chunks);
    return LS_ERR_INVALID_ARG;
}

// ---- tau: j-th best sample score of each query --------------------------------------------------
// The sample pass left, for every (workgroup, lane, query group), the 4 best sample scores that
// lane saw (ord() of the score, 0 = none). A query owns 4 lanes in each of its nsplits
// workgroups: 16*nsplits values, <= 4 per thread. 4 radix passes find the j-th largest.
// Keeping only 4 per lane can only LOWER the result (if one lane held more than 4 of the best
// j), i.e. let more rows through: tau is a speculative, verified threshold either way.
#define LS_TAU_PER_THREAD 8
#ifndef LS_TAU_PASSES
#define LS_TAU_PASSES 2
#endif
__global__ __launch_bounds__(256) void ls_tau_kernel(const u32* __restrict__ sample_top, int nsplits,
                                                     int nqt, int QG, int nq, int j_rank,
                                                     float* __restrict__ tau) {
    __shared__ u32 hist[4 * 256];
    __shared__ u32 misc[4 * 8];
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (q >= nq) {
        if (tid == 0) tau[q] = FLT_MAX;  // padded query: nothing passes
        return;
    }
    (void)nqt;
    (void)QG;
    const int total = nsplits * 4 * 4;  // values of this query, contiguous
    u32 v[LS_TAU_PER_THREAD];
#pragma unroll
    for (int j = 0; j < LS_TAU_PER_THREAD; ++j) {
        const int idx = tid + j * 256;
        u32 x = 0;
        if (idx < total) {
            const int e = idx & 3, sq = idx >> 2, quarter = sq & 3, split = sq >> 2;
            x = sample_top[queue_id(q, split, quarter, nsplits) * 4 + e];  // == q*total + idx
        }
        v[j] = x;
    }
    for (int i = tid; i < 4 * 256; i += 256) hist[i] = 0;
    __syncthreads();
    u32 pref = 0, pmask = 0, krem = (u32)j_rank;
    // Only the top 16 bits of the order key are resolved (sign, exponent, 7 mantissa bits): the
    // result is at most 0.8 % below the exact j-th sample score, i.e. still a valid (slightly
    // more permissive) speculative threshold, for half the passes.
    for (int pass = 0; pass < LS_TAU_PASSES; ++pass) {
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

int ls_launch_tau(const u32* d_sample_top, int nsplits, int64_t nq, int64_t nq_pad, const ls_geom& g,
                  int j_rank, float* d_tau, hipStream_t s) {
    const int QG = ls_gemm_qg(g);
    if ((long long)nsplits * 16 > 256LL * LS_TAU_PER_THREAD) {
        ls_set_error("batched path: sample too large for the tau kernel");
        return LS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ls_tau_kernel, dim3((unsigned)nq_pad), dim3(256), 0, s, d_sample_top,
                       nsplits, (int)(nq_pad / (LS_GEMM_WAVES * 16 * QG)), QG, (int)nq, j_rank, d_tau);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- exact top-k of each query's queues --------------------------------------------------------------
#define LS_BSEL_KEYS 2048
__global__ __launch_bounds__(256) void ls_batch_select_kernel(
    const u64* __restrict__ queues, const u32* __restrict__ counts, int cap, int nsplits, int nqt,
    int QG, int k, long long base, long long n, long long rows_per_split,
    u32* __restrict__ overflow, float* __restrict__ out_scores,
    long long* __restrict__ out_indices) {
    __shared__ u64 keys[LS_BSEL_KEYS];
    __shared__ u64 res[256];
    __shared__ u64 tmp[256];
    __shared__ u32 hist[8 * 256];
    __shared__ u32 misc[64];
    __shared__ u32 nkeys;
    __shared__ u32 wsum[8];
    const int q = blockIdx.x, tid = threadIdx.x;
    (void)nqt;
    (void)QG;
    // gather: the query owns 4 queues per slice (<= 512); thread t takes queues t and t + 256:
    // one load each for their lengths, a block-wide prefix for the slot ranges in LDS, then the
    // (few) live entries
    const int nqueues = nsplits * 4;
    u32 c[2] = {0, 0};
    const u64* qptr[2] = {nullptr, nullptr};
    long long r_begin[2] = {0, 0};
    // the first four entries of each queue are fetched together with its length (their address
    // does not depend on it): one round trip instead of two for the typical <= 4-entry queue
    uint4 ea[2], eb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int qi = tid + h * 256;
        ea[h] = eb[h] = make_uint4(0, 0, 0, 0);
        if (qi < nqueues) {
            const int quarter = qi & 3, split = qi >> 2;
            r_begin[h] = (long long)split * rows_per_split;
            const long long qid = queue_id(q, split, quarter, nsplits);
            c[h] = counts[qid];
            qptr[h] = queues + qid * cap;
            ea[h] = reinterpret_cast<const uint4*>(qptr[h])[0];
            eb[h] = reinterpret_cast<const uint4*>(qptr[h])[1];
        }
    }
    {
        const int lane = tid & 63, wv = tid >> 6;
        const u32 ct = c[0] + c[1];
        u32 inc = ct;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t2 = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t2;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        u32 off = 0;
        for (int i = 0; i < wv; ++i) off += wsum[i];
        if (tid == 0) nkeys = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        u32 start = off + inc - ct;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const u32 ch = c[h];
            const long long rb = r_begin[h];
            auto put = [&](u32 e, u32 bits, u32 lrow) {
                if (e < ch && start + e < LS_BSEL_KEYS) {
                    const long long row = rb + (long long)lrow;
                    keys[start + e] = row < n ? ls_make_key(__uint_as_float(bits), (u32)row) : 0ull;
                }
            };
            uint4 a = ea[h], b = eb[h];
            for (u32 e0 = 0; e0 < ch; e0 += 4) {  // cap is a multiple of 4: loads stay in the queue
                if (e0) {
                    a = reinterpret_cast<const uint4*>(qptr[h] + e0)[0];
                    b = reinterpret_cast<const uint4*>(qptr[h] + e0)[1];
                }
                put(e0, a.x, a.y);
                put(e0 + 1, a.z, a.w);
                put(e0 + 2, b.x, b.y);
                put(e0 + 3, b.z, b.w);
            }
            start += ch;
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
#if defined(LS_BSEL_ABL) && LS_BSEL_ABL == 1  // timing ablation: gather only
    if (cap > 0) return;
#endif
    const int nvalid = lds_topk(keys, cnt, k, res, tmp, hist, misc, tid, 256);
#if defined(LS_BSEL_ABL) && LS_BSEL_ABL == 2  // timing ablation: no output
    if (cap > 0) return;
#endif
    __syncthreads();
    if (nvalid < k && tid == 0) overflow[q] = 2u;  // the speculative tau let < k rows through
    for (int i = tid; i < k; i += 256) {
        const u64 key = i < nvalid ? res[i] : 0ull;
        out_scores[(long long)q * k + i] = ls_key_score(key);
        out_indices[(long long)q * k + i] = ls_key_index(key, base);
    }
}

int ls_launch_batch_select(const u64* d_queues, const u32* d_counts, int cap, int nsplits,
                           int64_t nq, int64_t nq_pad, const ls_geom& g, int k, int64_t base,
                           int64_t n, int64_t rows_per_split, u32* d_overflow, float* d_out_scores, int64_t* d_out_indices,
                           hipStream_t s) {
    const int QG = ls_gemm_qg(g);
    if (k > LS_GEMM_MAX_K || nsplits * 4 > 512) {
        ls_set_error("batched path: k > %d or too many slices", LS_GEMM_MAX_K);
        return LS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ls_batch_select_kernel, dim3((unsigned)nq), dim3(256), 0, s, d_queues,
                       d_counts, cap, nsplits, (int)(nq_pad / (LS_GEMM_WAVES * 16 * QG)), QG, k, (long long)base,
                       (long long)n, (long long)rows_per_split, d_overflow, d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
