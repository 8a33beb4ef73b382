// ls_gemm.hip — the batched path: Q[nq, d] x Corpus^T[d, N] on the matrix cores with the
// top-k selection fused into the epilogue (BASELINE config 3: N=200k, d=384 fp16, nq=1024,
// k=100; config 4: d=768 fp16, nq=256). Stands in for faiss `index.search(x, k)` with a large nq
// (reference src/lean_explore/search/engine.py:250; the reference itself only ever sends nq=1).
//
// Why fused: the score matrix is nq*N fp32 = 819 MB for config 3; writing and re-reading it
// would cost more HBM time than the whole MFMA budget, so scores never leave the CU.
//
// ls_gemm_filter_kernel — one workgroup = 8 waves x (16*QG queries) x one corpus slice
//   - v_mfma_f32_16x16x32_f16. A wave owns QG = 2 groups of 16 queries (1 only for stored rows of
//     2 KiB): their fp16 fragments stay in VGPRs for the whole slice, so B costs no LDS or HBM
//     traffic in the loop. For 1.5 KiB rows (config 4) that is 192 of the wave's 256 registers;
//     the kernel then keeps ONE accumulator set and the two waves of a SIMD filter at opposite
//     ends of a tile, so one of them always feeds the matrix pipe (see run_tile).
//   - A operand (corpus): tiles of TM rows (64, or 32 for long rows) stream HBM/L2 -> LDS by DMA
//     (global_load_lds, 16 B/lane, double buffered) and are shared by the 8 waves. LDS rows are
//     XOR-swizzled on the SOURCE address (chunk ^ (row & 15)): conflict-free ds_read_b128.
//   - epilogue: lane (query, quarter) holds 4 row scores per accumulator; a score >= tau[query]
//     is appended to the lane's private queue. The first QL entries of a queue live in the LDS
//     the tiles leave free (a ds_write_b64, no HBM traffic, no atomics); the rare lane that sees
//     more spills to a private HBM queue. When the slice is done the four quarter-queues of a
//     query are compacted into ONE contiguous record per (query, slice) in HBM: the select kernel
//     reads 1-2 lines per slice instead of walking 4 scattered queues.
//   - workgroups that share a corpus slice sit on the same XCD (block % 8) so the slice is
//     fetched from HBM once and served to the other query tiles from that XCD's L2.
//
// Phases (ls_api.hip orchestrates): sample pass (a few tiles of every slice; each lane keeps its 4
// best sample scores in registers) -> tau kernel (j-th best sample score per query) -> full pass
// with tau -> select kernel (exact top-k of each query's records, verifies >= k candidates). A
// flagged query (queue overflow / too few candidates) is re-run by the exact per-query scan
// path, so the result is always exact.
#include "ls_select_dev.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#define LS_GEMM_LDS_BYTES (160 * 1024)  // the whole LDS of a CU: one workgroup per CU

// ---- static geometry of one instantiation ------------------------------------------------------
__host__ __device__ constexpr int gemm_qg(int chunks) { return chunks <= 96 ? 2 : 1; }
__host__ __device__ constexpr int gemm_tm(int chunks) { return chunks <= 64 ? LS_GEMM_TM_SHORT : 32; }
__host__ __device__ constexpr int gemm_tile_bytes(int chunks) { return gemm_tm(chunks) * chunks * 16; }
// LDS queue entries per (lane, query group): what two tile buffers leave free, at most LS_GEMM_QL
__host__ __device__ constexpr int gemm_ql(int chunks) {
    const int left = LS_GEMM_LDS_BYTES - 2 * gemm_tile_bytes(chunks);
    const int per = left / (LS_GEMM_THREADS * gemm_qg(chunks) * 8);
    return per < 0 ? 0 : (per > LS_GEMM_QL ? LS_GEMM_QL : per);
}

// ---- queries -> fp16 MFMA B fragments, normalised if asked, zero padded ---------------------------
// Output layout = the order in which ls_gemm_filter_kernel consumes it: the 16-byte chunk c of
// query q (tile qt, wave w, group g2, li = q % 16) is fragment (qt, w, g2, kk = c / 4), lane
// (c % 4) * 16 + li. One wave load of a B fragment is then one contiguous 1 KiB read.
__device__ __forceinline__ long long qfrag_chunk(int q, int c, int QG, int KS) {
    const int QPW = 16 * QG, QT = LS_GEMM_WAVES * QPW;
    const int qt = q / QT, w = (q % QT) / QPW, g2 = (q % QPW) / 16, li = q % 16;
    return ((((long long)(qt * LS_GEMM_WAVES + w) * QG + g2) * KS + (c >> 2)) << 6) + ((c & 3) << 4) + li;
}
// One wave per query (4 queries per block): the norm is the library's canonical wave reduction
// (ls_wave_sumsq), every lane converts and stores whole 16-byte chunks. The raw fp32 queries are
// also copied into the call's own slot (`qkeep`): a later repair of a flagged query must not
// depend on the caller keeping its query buffer alive.
__global__ __launch_bounds__(256) void ls_prep_f16_kernel(const float* __restrict__ qin,
                                                          u32x4* __restrict__ qout,
                                                          float* __restrict__ qkeep, int nq,
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
        const float ss = ls_wave_sumsq(src, d, lane);
        if (ss > 0.0f) inv = 1.0f / sqrtf(ss);
    }
    if (live && qkeep)
        for (int j = lane; j < d; j += 64) qkeep[(long long)qi * d + j] = src[j];
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

int ls_launch_prep_f16(const float* d_q, void* d_qh, float* d_qkeep, int64_t nq, int64_t nq_pad,
                       const ls_geom& g, bool normalize, u32* d_overflow, hipStream_t s) {
    hipLaunchKernelGGL(ls_prep_f16_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, s, d_q,
                       (u32x4*)d_qh, d_qkeep, (int)nq, (int)nq_pad, g.d, g.d_pad, ls_gemm_qg(g),
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
// Query q of a launch with QG groups per wave lives in tile qt = q / (128*QG), wave
// w = (q % (128*QG)) / (16*QG), group qg = (q % (16*QG)) / 16, li = q % 16; in every workgroup of
// its tile 4 lanes (quarter = 0..3) filter for it. Everything the tau and select kernels read for
// one query is contiguous: sample tops and spill queues [query][slice][quarter], records and
// their lengths [query][slice].
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

// LDS store the compiler does not see as one: with a tile DMA (global_load_lds) in flight hipcc
// puts `s_waitcnt vmcnt(0)` in front of every ordinary LDS store (the DMA is a pending LDS write
// it cannot prove disjoint), which would park the wave behind the next tile's HBM fetch on every
// queue append. LDS operations of one wave execute in order, so the flush's later reads of the
// same queue need no wait either.
__device__ __forceinline__ void lds_store_b64_nowait(unsigned lds_addr, unsigned lo, unsigned hi) {
    const u64 v = ((u64)hi << 32) | lo;
    asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr), "v"(v) : "memory");
}

struct ls_gemm_out {
    uint2* rec;        // [nq_pad][nsplits][LS_GEMM_REC] (score bits, slice-relative row)
    u32* rcnt;         // [nq_pad][nsplits] entries in the record | spill mask << 8
    uint2* spill;      // [nq_pad][nsplits][4][LS_GEMM_SCAP] private HBM queues past the LDS part
    u32* scnt;         // [nq_pad][nsplits][4] entries a spilling lane wrote
    u32* overflow;     // [nq_pad] per-query repair flag
    u32* sample_top;   // [nq_pad][nsplits][4][4] (sample pass)
};

template <int CHUNKS, int QG, bool SAMPLE>
__global__ __launch_bounds__(LS_GEMM_THREADS, LS_GEMM_WAVES_PER_SIMD) void ls_gemm_filter_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride, ls_gemm_out out) {
    constexpr int TM = gemm_tm(CHUNKS);    // corpus rows per LDS tile
    constexpr int NRB = TM / 16;           // 16-row MFMA blocks per tile
    constexpr int KS = CHUNKS / 4;         // k-steps: 32 fp16 = 4 chunks each
    constexpr int QPW = 16 * QG;           // queries per wave
    constexpr int NV = NRB * QG * 4;       // filter values per lane per tile
    // B fragments of QG groups take KS*QG*4 registers. When that leaves too little for two
    // accumulator sets, one set is kept and a chain is filtered right before it restarts.
    constexpr bool ONE_ACC = KS * QG * 4 >= 128;
    constexpr int CPK = (NV + KS - 1) / KS;  // two sets: checks interleaved per k-step
    constexpr int ROW_BYTES = CHUNKS * 16;
    constexpr int TILE_CHUNKS = TM * CHUNKS;
    constexpr int TILE_BYTES = TILE_CHUNKS * 16;
    constexpr int LOADS = TILE_CHUNKS / LS_GEMM_THREADS;  // 16-byte DMA loads per thread per tile
    constexpr int QL = SAMPLE ? 0 : gemm_ql(CHUNKS);
    static_assert(TILE_CHUNKS % LS_GEMM_THREADS == 0, "tile must split evenly over the threads");
    static_assert(QL * 4 <= LS_GEMM_REC, "a record holds the four LDS quarter-queues");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // tiles | lane queues

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
    const int nt = (ntiles_all + tile_stride - 1) / tile_stride;  // tiles this launch visits

    // ---- corpus tiles: HBM/L2 -> LDS by DMA (global_load_lds, 16 B per lane), double buffered --
    // Wave w, load j fills the 64 consecutive LDS chunks starting at (w*LOADS + j)*64: chunk Lc
    // is tile row r = Lc / CHUNKS, slot sl = Lc % CHUNKS and receives SOURCE chunk sl ^ (r & 15)
    // (the DMA destination is lane-linear, so the swizzle goes on the source address). The HBM
    // copy is padded with zero rows past n (ls_api.hip): no clamping. A tile takes ~3 us to
    // consume, longer than the DMA's flight, so one tile of look-ahead suffices.
    // Register-starved instantiations (ONE_ACC) recompute the per-lane source offsets for every
    // tile (a handful of VALU ops behind an opaque copy of the lane id): hoisted out of the tile
    // loop they would be spilled and every reload would wait on the memory pipe.
    int goff[ONE_ACC ? 1 : LOADS];
    if constexpr (!ONE_ACC) {
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int Lc = (wave * LOADS + j) * 64 + lane;
            const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
            goff[j] = r * CHUNKS + (sl ^ (r & 15));
        }
    }
    auto stage = [&](int ti, int buf) {
        const u32x4* base = corpus + (r_begin + (long long)ti * TM) * CHUNKS;
        int lane_v = lane;
        if constexpr (ONE_ACC) asm volatile("" : "+v"(lane_v));
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            int off;
            if constexpr (ONE_ACC) {
                const int Lc = (wave * LOADS + j) * 64 + lane_v;
                const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
                off = r * CHUNKS + (sl ^ (r & 15));
            } else {
                off = goff[j];
            }
            unsigned char* dst = smem + buf * TILE_BYTES + (wave * LOADS + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + off), (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // The first tile(s) are requested BEFORE the query fragments: the HBM round trip of tile 0
    // then overlaps the (L2-resident) query loads instead of queueing behind them.
    constexpr int NB_ALL = LS_GEMM_LDS_BYTES / TILE_BYTES;  // tiles that fit in LDS together
    const bool sample_upfront = SAMPLE && nt > 0 && nt <= NB_ALL && nt <= 3;
    if (nt > 0) stage(0, 0);
    if (sample_upfront) {
        if (nt > 1) stage(tile_stride, 1);
        if (nt > 2) stage(2 * tile_stride, 2);
    }

    // B fragments: group qg holds query qt*8*QPW + wave*QPW + qg*16 + li; k-step kk -> chunk 4kk+qd
    half8 bq[QG][KS];
    int qj[QG];
    float tauv[QG];
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2) {
        qj[g2] = (qt * LS_GEMM_WAVES + wave) * QPW + g2 * 16 + li;
        // fragment-ordered by ls_prep_f16_kernel: each load below is one contiguous KiB per wave
        const u32x4* qfrag = qh + ((((long long)(qt * LS_GEMM_WAVES + wave) * QG + g2) * KS) << 6) + lane;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) bq[g2][kk] = __builtin_bit_cast(half8, qfrag[kk << 6]);
        tauv[g2] = SAMPLE ? 0.0f : tau[qj[g2]];
    }

    // This lane's candidate queues, one per query group: entry = {score bits, row relative to the
    // slice}. Entries 0..QL-1 in LDS (behind the two tile buffers), the rest in the HBM spill queue.
    uint2* lq = reinterpret_cast<uint2*>(smem + 2 * TILE_BYTES) + (size_t)tid * QG * (QL > 0 ? QL : 1);
    const unsigned lq_addr = (unsigned)(uintptr_t)(lds_ptr_t)lq;  // LDS byte address of the queue
    int cnt[QG];
    float top[QG][4];
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2) {
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

    // ---- one score of a finished tile -------------------------------------------------------------
    // The append is the hot slow path (a wave enters it for ~1 check in 5): no key building, no
    // bounds logic here. Padded queries carry tau = FLT_MAX and never pass; zero-pad rows past n
    // are dropped by the select kernel; a full spill queue keeps overwriting its last slot while
    // the count runs on, which is how the overflow is seen at the end.
    auto check1 = [&](float s, int g2, int lrow) {
        if (SAMPLE) {
            // NaN never enters (fmaxf/fminf drop it); padded queries and zero-pad rows are masked
            top4_insert(top[g2], (qj[g2] < nq && r_begin + lrow < r_end) ? s : -FLT_MAX);
        } else if (s >= tauv[g2]) {
            const int c = cnt[g2];
            const uint2 ent = make_uint2(__float_as_uint(s), (u32)lrow);
            if (c < QL) {
                lds_store_b64_nowait(lq_addr + (unsigned)(g2 * QL + c) * 8u, ent.x, ent.y);
            } else {
                const int slot = c - QL < LS_GEMM_SCAP ? c - QL : LS_GEMM_SCAP - 1;
                out.spill[queue_id(qj[g2], split, qd, nsplits) * LS_GEMM_SCAP + slot] = ent;
            }
            cnt[g2] = c + 1;
        }
    };
    auto check = [&](const f32x4v (&acc)[NRB][QG], int e, int lrow0) {  // e -> (block, group, reg)
        const int rb = e / (QG * 4), g2 = (e / 4) % QG, reg = e % 4;
        check1(acc[rb][g2][reg], g2, lrow0 + rb * 16 + reg);  // lrow0 already includes 4*qd
    };

    // One tile: NRB*QG independent accumulator chains advance together, one k-step at a time.
    // Two sets: the PREVIOUS tile's accumulators are filtered CPK elements per k-step, in the
    // shadow of the matrix pipe. One set (ONE_ACC): the filter cannot hide inside its own wave's
    // MFMA stream (every check is a branch the MFMAs are not scheduled across), so it hides under
    // the OTHER wave of the SIMD instead: waves 0-3 filter a tile right after the hand-over
    // barrier, before they overwrite the accumulators ("early"), waves 4-7 right after their
    // k-loop, before the barrier ("late"); waves w and w+4 share a SIMD, so one of the pair is
    // always feeding the matrix pipe while the other one filters.
    const bool late = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    auto run_tile = [&](f32x4v (&cur)[NRB][QG], const f32x4v (&prev)[NRB][QG], bool have_prev,
                        int prev_row0, int cur_row0, int bufoff) {
        if (ONE_ACC && !late && have_prev) {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(prev, e, prev_row0);
        }
        half8 a[2][NRB];  // A fragments are read one k-step ahead of their MFMAs
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb) a[0][rb] = a_frag(bufoff, rb, 0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) {
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb) a[(kk + 1) & 1][rb] = a_frag(bufoff, rb, kk + 1);
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
                    cur[rb][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk & 1][rb], bq[g2][kk],
                                                                        c, 0, 0, 0);
                }
            }
            if (!ONE_ACC && have_prev) {
#pragma unroll
                for (int c2 = 0; c2 < CPK; ++c2)
                    if (kk * CPK + c2 < NV) check(prev, kk * CPK + c2, prev_row0);
            }
        }
        if (ONE_ACC && late) {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(cur, e, cur_row0);
        }
    };

    f32x4v accA[NRB][QG], accB[NRB][QG];  // two sets alternate between tiles (ONE_ACC: accA only)
    auto tile_row0 = [&](int i) { return (i * tile_stride) * TM + 4 * qd; };  // slice-relative
    auto flush_last = [&](const f32x4v (&acc)[NRB][QG]) {  // the last tile still has to be filtered
        const int row0 = tile_row0(nt - 1);
#pragma unroll
        for (int e = 0; e < NV; ++e) check(acc, e, row0);
    };
    // The sample pass visits only a few tiles, so their DMA latencies would be paid one by one:
    // when all of them fit in LDS together they are fetched up front and consumed back to back.
    if constexpr (SAMPLE) {
        if (sample_upfront) {
            __syncthreads();
            if constexpr (ONE_ACC) {
                for (int i = 0; i < nt; ++i)
                    run_tile(accA, accA, i > 0, tile_row0(i - 1), tile_row0(i), i * TILE_BYTES);
                if (!late) flush_last(accA);
            } else {
                run_tile(accA, accB, false, 0, 0, 0);
                if (nt > 1) run_tile(accB, accA, true, tile_row0(0), 0, TILE_BYTES);
                if (nt > 2) run_tile(accA, accB, true, tile_row0(1), 0, 2 * TILE_BYTES);
                if ((nt - 1) & 1) flush_last(accB); else flush_last(accA);
            }
#pragma unroll
            for (int g2 = 0; g2 < QG; ++g2)
                reinterpret_cast<uint4*>(out.sample_top)[queue_id(qj[g2], split, qd, nsplits)] =
                    top4_keys(top[g2]);
            return;
        }
    }
    __syncthreads();  // the compiler drains the DMA (vmcnt(0)) before the barrier
    if constexpr (ONE_ACC) {
        for (int i = 0; i < nt; ++i) {
            if (i + 1 < nt) stage((i + 1) * tile_stride, (i + 1) & 1);
            run_tile(accA, accA, i > 0, tile_row0(i - 1), tile_row0(i), (i & 1) * TILE_BYTES);
            __syncthreads();
        }
        if (nt > 0 && !late) flush_last(accA);
    } else {
        for (int i = 0; i < nt; i += 2) {
            if (i + 1 < nt) stage((i + 1) * tile_stride, 1);
            run_tile(accA, accB, i > 0, tile_row0(i - 1), 0, 0);
            __syncthreads();
            if (i + 1 < nt) {
                if (i + 2 < nt) stage((i + 2) * tile_stride, 0);
                run_tile(accB, accA, true, tile_row0(i), 0, TILE_BYTES);
                __syncthreads();
            }
        }
        if (nt > 0) {
            if ((nt - 1) & 1) flush_last(accB); else flush_last(accA);
        }
    }
    if constexpr (SAMPLE) {
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2)
            reinterpret_cast<uint4*>(out.sample_top)[queue_id(qj[g2], split, qd, nsplits)] =
                top4_keys(top[g2]);
    } else {
        // ---- compact the four quarter-queues of every query into its (query, slice) record -------
        // LDS is in-order per wave, and a lane reads back only what it wrote itself: no barrier.
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2) {
            const int c = cnt[g2] < QL ? cnt[g2] : QL;  // entries in LDS
            const int spilled = cnt[g2] > QL;
            int off = 0, total = 0, smask = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cj = __shfl(c, li + 16 * j, 64);
                const int sj = __shfl(spilled, li + 16 * j, 64);
                off += j < qd ? cj : 0;
                total += cj;
                smask |= sj << j;
            }
            const long long rid = (long long)qj[g2] * nsplits + split;
            uint2* r = out.rec + rid * LS_GEMM_REC + off;
            for (int e = 0; e < c; ++e) r[e] = lq[g2 * QL + e];
            if (qd == 0) out.rcnt[rid] = (u32)total | ((u32)smask << 8);
            if (spilled) {
                const int sc = cnt[g2] - QL;
                out.scnt[queue_id(qj[g2], split, qd, nsplits)] = (u32)(sc < LS_GEMM_SCAP ? sc : LS_GEMM_SCAP);
                if (sc > LS_GEMM_SCAP) out.overflow[qj[g2]] = 1u;
            }
        }
    }
}

int ls_gemm_qg(const ls_geom& g) { return gemm_qg(g.chunks); }
int ls_gemm_tile_rows(const ls_geom& g) { return gemm_tm(g.chunks); }

int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, const ls_gemm_bufs& b,
                          hipStream_t s) {
    const int QG = ls_gemm_qg(g);
    const int nqt = (int)(nq_pad / (LS_GEMM_WAVES * 16 * QG));
    const dim3 grid((unsigned)(nsplits * nqt));
    ls_gemm_out o;
    o.rec = (uint2*)b.d_rec;
    o.rcnt = b.d_rcnt;
    o.spill = (uint2*)b.d_spill;
    o.scnt = b.d_scnt;
    o.overflow = b.d_overflow;
    o.sample_top = b.d_sample_top;
    // full pass: two tile buffers + the lane queues; sample pass: up to three tiles up front
    const size_t tile_bytes = (size_t)gemm_tile_bytes(g.chunks);
    size_t smem;
    if (d_tau)
        smem = 2 * tile_bytes + (size_t)LS_GEMM_THREADS * QG * (gemm_ql(g.chunks) > 0 ? gemm_ql(g.chunks) : 1) * 8;
    else
        smem = tile_bytes * (3 * tile_bytes <= LS_GEMM_LDS_BYTES ? 3 : 2);
#define LS_GEMM_LAUNCH(C, SMP)                                                                    \
    {                                                                                             \
        auto kern = ls_gemm_filter_kernel<C, gemm_qg(C), SMP>;                                    \
        static ls_attr_once once;                                                                 \
        if (int rc = ls_set_max_dynamic_lds(once, (const void*)kern, LS_GEMM_LDS_BYTES)) return rc; \
        hipLaunchKernelGGL(kern, grid, dim3(LS_GEMM_THREADS), smem, s, (const u32x4*)d_corpus,    \
                           (long long)n, (const u32x4*)d_qh, (int)nq, nqt, d_tau,                 \
                           (long long)rows_per_split, tile_stride, o);                            \
        LS_HIP(hipGetLastError());                                                                \
        return LS_OK;                                                                             \
    }
#define LS_GEMM_CASE(C)                     \
    if (g.chunks == C) {                    \
        if (d_tau) LS_GEMM_LAUNCH(C, false) \
        else LS_GEMM_LAUNCH(C, true)        \
    }
    LS_GEMM_CASE(16) LS_GEMM_CASE(32) LS_GEMM_CASE(48) LS_GEMM_CASE(64)
    LS_GEMM_CASE(96) LS_GEMM_CASE(128)
#undef LS_GEMM_CASE
#undef LS_GEMM_LAUNCH
    ls_set_error("batched path: unsupported row geometry (%d chunks)", g.chunks);
    return LS_ERR_INVALID_ARG;
}

// ---- tau: j-th best sample score of each query --------------------------------------------------
// The sample pass left, for every (workgroup, lane, query group), the 4 best sample scores that
// lane saw (ord() of the score, 0 = none). A query owns 4 lanes in each of its nsplits
// workgroups: 16*nsplits values. ONE WAVE per query (4 queries per workgroup): the values sit in
// registers, each radix pass is a wave-private LDS histogram + a 64-lane suffix scan: no
// workgroup barrier anywhere. Keeping only 4 per lane can only LOWER the result (if one lane held
// more than 4 of the best j), i.e. let more rows through: tau is a speculative, verified
// threshold either way.
#define LS_TAU_PER_LANE (LS_GEMM_MAX_SPLITS * 16 / 64)
__global__ __launch_bounds__(256) void ls_tau_kernel(const u32* __restrict__ sample_top, int nsplits,
                                                     int nq, int nq_pad, int j_rank,
                                                     float* __restrict__ tau) {
    __shared__ u32 hist_all[4][256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wv;
    if (q >= nq_pad) return;
    if (q >= nq) {
        if (lane == 0) tau[q] = FLT_MAX;  // padded query: nothing passes
        return;
    }
    u32* hist = hist_all[wv];
    const int total = nsplits * 16;  // values of this query, contiguous: [slice][quarter][4]
    const uint4* src = reinterpret_cast<const uint4*>(sample_top) + (long long)q * nsplits * 4;
    u32 v[LS_TAU_PER_LANE];
#pragma unroll
    for (int j = 0; j < LS_TAU_PER_LANE / 4; ++j) {
        const int idx4 = lane + j * 64;  // one (slice, quarter) group of 4 per load
        const uint4 x = idx4 * 4 < total ? src[idx4] : make_uint4(0, 0, 0, 0);
        v[4 * j] = x.x; v[4 * j + 1] = x.y; v[4 * j + 2] = x.z; v[4 * j + 3] = x.w;
    }
    u32 pref = 0, pmask = 0, krem = (u32)j_rank;
    // Only the top 16 bits of the order key are resolved (sign, exponent, 7 mantissa bits): the
    // result is at most 0.8 % below the exact j-th sample score, i.e. still a valid (slightly
    // more permissive) speculative threshold, for half the passes.
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = lane; i < 256; i += 64) hist[i] = 0;
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < LS_TAU_PER_LANE; ++j)
            if (v[j] != 0u && (v[j] & pmask) == pref) atomicAdd(&hist[(v[j] >> shift) & 255u], 1u);
        wave_lds_fence();
        // lane l owns bins 4l..4l+3; suffix sums locate the bin of the krem-th largest value
        const u32 h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2],
                  h3 = hist[4 * lane + 3];
        const u32 mine = h0 + h1 + h2 + h3;
        u32 suf = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_down(suf, o, 64);
            if (lane + o < 64) suf += t;
        }
        const u32 above = suf - mine;
        const u32 totalv = __shfl(suf, 0, 64);
        if (pass == 0 && totalv < (u32)j_rank) {  // fewer than j sample scores: no bound
            if (lane == 0) tau[q] = -FLT_MAX;
            return;
        }
        const bool owner = krem > above && krem <= above + mine;  // exactly one lane
        u32 bin = 0, rem = 0;
        if (owner) {
            u32 cum = above;
            if (cum + h3 >= krem) { bin = 4 * lane + 3; rem = krem - cum; }
            else {
                cum += h3;
                if (cum + h2 >= krem) { bin = 4 * lane + 2; rem = krem - cum; }
                else {
                    cum += h2;
                    if (cum + h1 >= krem) { bin = 4 * lane + 1; rem = krem - cum; }
                    else { cum += h1; bin = 4 * lane; rem = krem - cum; }
                }
            }
        }
        const int ol = __ffsll((long long)__ballot(owner)) - 1;
        bin = (u32)__shfl((int)bin, ol, 64);
        krem = (u32)__shfl((int)rem, ol, 64);
        pref |= bin << shift;
        pmask |= 255u << shift;
        wave_lds_fence();
    }
    if (lane == 0) tau[q] = ls_unord(pref);
}

int ls_launch_tau(const u32* d_sample_top, int nsplits, int64_t nq, int64_t nq_pad, int j_rank,
                  float* d_tau, hipStream_t s) {
    if (nsplits > LS_GEMM_MAX_SPLITS) {
        ls_set_error("batched path: too many slices for the tau kernel");
        return LS_ERR_INVALID_ARG;
    }
    hipLaunchKernelGGL(ls_tau_kernel, dim3((unsigned)((nq_pad + 3) / 4)), dim3(256), 0, s,
                       d_sample_top, nsplits, (int)nq, (int)nq_pad, j_rank, d_tau);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- exact top-k of each query's records -------------------------------------------------------------
// One workgroup per query. Its nsplits record lengths are contiguous (one load per thread), a
// block-wide prefix assigns LDS slots, then P = 256 / nsplits threads share each record's
// entries (16-byte loads of two entries). Spilled lanes (rare) are walked afterwards.
// LDS: keys[keys_cap] | res[res_cap] | tmp[res_cap] (u64), hist[8*256] | misc[64] (u32).
__global__ __launch_bounds__(256) void ls_batch_select_kernel(
    const uint2* __restrict__ rec, const u32* __restrict__ rcnt, const uint2* __restrict__ spill,
    const u32* __restrict__ scnt, int nsplits, int k, int keys_cap, int res_cap, long long base,
    long long n, long long rows_per_split, u32* __restrict__ overflow,
    float* __restrict__ out_scores, long long* __restrict__ out_indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sel[];
    u64* keys = reinterpret_cast<u64*>(smem_sel);
    u64* res = keys + keys_cap;
    u64* tmp = res + res_cap;
    u32* hist = reinterpret_cast<u32*>(tmp + res_cap);
    u32* misc = hist + 8 * 256;
    u32* wsum = misc + 64;  // [4] + nkeys: inside the dynamic region (a static __shared__ in front
    u32& nkeys = wsum[4];   // of it would shift its 16-byte alignment)
    const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const u32 word = tid < nsplits ? rcnt[(long long)q * nsplits + tid] : 0u;
    const u32 c_rec = word & 255u, smask = (word >> 8) & 15u;
    // spilled lanes' lengths ride along in the same prefix (their queue is walked by this thread)
    u32 c_sp[4] = {0, 0, 0, 0};
    u32 ct = c_rec;
    if (smask) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (smask >> j & 1) {
                c_sp[j] = scnt[queue_id(q, tid, j, nsplits)];
                ct += c_sp[j];
            }
    }
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
    const u32 start = off + inc - ct;  // first LDS slot of slice `tid`
    auto put = [&](u32 slot, uint2 ent, long long rb) {
        if (slot < (u32)keys_cap) {
            const long long row = rb + (long long)ent.y;
            keys[slot] = row < n ? ls_make_key(__uint_as_float(ent.x), (u32)row) : 0ull;
        }
    };
    // record entries: P threads per slice, two entries per 16-byte load
    const int P = nsplits <= 64 ? 4 : (nsplits <= 128 ? 2 : 1);
    {
        const int sl = tid / P, part = tid % P;
        // the owner thread's (start, c_rec) reach its helpers through LDS (hist is free until
        // lds_topk, which re-zeroes it behind the barrier below)
        if (tid < nsplits) reinterpret_cast<uint2*>(hist)[tid] = make_uint2(start, c_rec);
        __syncthreads();
        if (sl < nsplits) {
            const uint2 sc = reinterpret_cast<const uint2*>(hist)[sl];
            const long long rb = (long long)sl * rows_per_split;
            const uint4* r4 = reinterpret_cast<const uint4*>(rec + ((long long)q * nsplits + sl) * LS_GEMM_REC);
            for (u32 e = 2 * part; e < sc.y; e += 2 * P) {
                const uint4 x = r4[e >> 1];
                put(sc.x + e, make_uint2(x.x, x.y), rb);
                if (e + 1 < sc.y) put(sc.x + e + 1, make_uint2(x.z, x.w), rb);
            }
        }
    }
    if (smask) {  // rare: this slice has lanes that spilled past their LDS queue
        u32 slot = start + c_rec;
        const long long rb = (long long)tid * rows_per_split;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint2* sq = spill + queue_id(q, tid, j, nsplits) * LS_GEMM_SCAP;
            for (u32 e = 0; e < c_sp[j]; ++e) put(slot + e, sq[e], rb);
            slot += c_sp[j];
        }
    }
    __syncthreads();
    const int cnt = (int)nkeys;
    if (cnt > keys_cap) {  // more candidates than fit: the exact scan path handles this query
        if (tid == 0) overflow[q] = 1u;
        return;
    }
    if (overflow[q]) return;  // a spill queue overflowed in the GEMM pass
    const int nvalid = lds_topk(keys, cnt, k, res, tmp, hist, misc, tid, 256);
    __syncthreads();
    if (nvalid < k && tid == 0) overflow[q] = 2u;  // the speculative tau let < k rows through
    for (int i = tid; i < k; i += 256) {
        const u64 key = i < nvalid ? res[i] : 0ull;
        out_scores[(long long)q * k + i] = ls_key_score(key);
        out_indices[(long long)q * k + i] = ls_key_index(key, base);
    }
}

int ls_launch_batch_select(const ls_gemm_bufs& b, int nsplits, int64_t nq, int k, int64_t base,
                           int64_t n, int64_t rows_per_split, float* d_out_scores,
                           int64_t* d_out_indices, hipStream_t s) {
    if (k > LS_GEMM_MAX_K || nsplits > LS_GEMM_MAX_SPLITS) {
        ls_set_error("batched path: k > %d or too many slices", LS_GEMM_MAX_K);
        return LS_ERR_INVALID_ARG;
    }
    // candidates per query average ~2-5 k (the threshold's safety margin): room for 4 k, >= 2048
    int keys_cap = 2048;
    while (keys_cap < 4 * k) keys_cap <<= 1;
    int res_cap = 256;
    while (res_cap < k) res_cap <<= 1;
    const size_t smem = ((size_t)keys_cap + 2 * (size_t)res_cap) * sizeof(u64) +
                        (8 * 256 + 64 + 16) * sizeof(u32);
    static ls_attr_once once;
    if (int rc = ls_set_max_dynamic_lds(once, (const void*)ls_batch_select_kernel, 128 * 1024)) return rc;
    hipLaunchKernelGGL(ls_batch_select_kernel, dim3((unsigned)nq), dim3(256), smem, s,
                       (const uint2*)b.d_rec, b.d_rcnt, (const uint2*)b.d_spill, b.d_scnt, nsplits, k,
                       keys_cap, res_cap, (long long)base, (long long)n, (long long)rows_per_split,
                       b.d_overflow, d_out_scores, (long long*)d_out_indices);
    LS_HIP(hipGetLastError());
    return LS_OK;
}
