// ls_gemm.hip — the batched path: Q[nq, d] x Corpus^T[d, N] on the matrix cores with the
// top-k selection fused into the epilogue (BASELINE config 3: N=200k, d=384 fp16, nq=1024,
// k=100; config 4: d=768 fp16, nq=256). Stands in for faiss `index.search(x, k)` with a large nq
// (reference src/lean_explore/search/engine.py:250; the reference itself only ever sends nq=1).
//
// Why fused: the score matrix is nq*N fp32 = 819 MB for config 3; writing and re-reading it
// would cost more HBM time than the whole MFMA budget, so scores never leave registers.
//
// ls_gemm_filter_kernel — one workgroup = 8 waves x (16*QG queries) x one corpus slice
//   - v_mfma_f32_16x16x32_f16. A wave owns QG = 2 groups of 16 queries (1 only for stored rows of
//     2 KiB): their fp16 fragments stay in VGPRs for the whole slice, so B costs no LDS or HBM
//     traffic in the loop. For 1.5 KiB rows (config 4) that is 192 of the wave's 256 registers;
//     the kernel then keeps ONE accumulator set (ONE_ACC).
//   - A operand (corpus): tiles of TM rows (64, or 32 for long rows) stream HBM/L2 -> LDS by DMA
//     (global_load_lds, 16 B/lane) into TWO buffers (the shipped default, LS_GEMM_RING3 = 0): the
//     next tile's DMA pieces are issued one at a time between the current tile's k-steps, the
//     hand-over waits for them (vmcnt) and crosses one s_barrier per tile. (LS_GEMM_RING3 = 1, a
//     variant-build knob, keeps a ring of three buffers with the DMA two tiles ahead and a counted
//     vmcnt; it measured 1-2 % slower and is not what ships.)
//     LDS rows are XOR-swizzled on the SOURCE address (chunk ^ (row & 15)): conflict-free
//     ds_read_b128.
//   - epilogue: lane (query, quarter) holds 4 row scores per accumulator; a score >= tau[query]
//     is appended to the lane's private queue in HBM (no atomics) as a raw (score, slice-relative
//     row) pair. The append is the hot slow path — every instruction in it costs ~1 us per batch
//     (measured: an LDS-resident queue with an overflow test ran 10 us slower) — so it is a
//     clamped slot, one address add and one store.
//   - workgroups that share a corpus slice sit on the same XCD (block % 8) so the slice is
//     fetched from HBM once and served to the other query tiles from that XCD's L2.
//
// Phases (ls_api.hip orchestrates): sample pass (a few tiles of every slice; each lane keeps its 4
// best sample scores in registers) -> tau kernel (j-th best sample score per query) -> full pass
// with tau -> select kernel (exact top-k of each query's queues, verifies >= k candidates). A
// flagged query (queue overflow / too few candidates) is re-run by the exact per-query scan
// path, so the result is always exact.
#include "ls_select_dev.h"

#include <hip/hip_ext.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

#define LS_GEMM_LDS_BYTES (160 * 1024)  // the whole LDS of a CU: one workgroup per CU
#ifdef LS_GEMM_TIMING  // phase stamps (100 MHz) of the SAMPLE pass's workgroups: start, loads landed, tiles done, end
__device__ unsigned long long g_sample_stamps[1024 * 4];
int ls_gemm_read_sample_stamps(unsigned long long* out, int count) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sample_stamps), sizeof(unsigned long long) * count) == hipSuccess ? 0 : -1;
}
#define LS_SSTAMP_S(i) do { if (SAMPLE && tid == 0 && blockIdx.x < 1024) g_sample_stamps[blockIdx.x * 4 + (i)] = wall_clock64(); } while (0)
// fused launches: 8 stamp slots per workgroup (0 start, 2 full pass done, 7 end)
#define LS_SSTAMP_F(i) do { if (FUSED && tid == 0 && blockIdx.x < 512) g_sample_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#define LS_SSTAMP_F2(i) do { if (FUSED && tid == 0 && blockIdx.x < 256) g_sample_stamps[2048 + blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define LS_SSTAMP_S(i) do {} while (0)
#define LS_SSTAMP_F(i) do {} while (0)
#define LS_SSTAMP_F2(i) do {} while (0)
#endif

// ---- static geometry of one instantiation ------------------------------------------------------
__host__ __device__ constexpr int gemm_qg(int chunks) { return chunks <= LS_GEMM_QG2_MAX_CHUNKS ? 2 : 1; }
__host__ __device__ constexpr int gemm_tm(int chunks) { return chunks <= 48 ? LS_GEMM_TM_SHORT : 32; }
__host__ __device__ constexpr int gemm_tile_bytes(int chunks) { return gemm_tm(chunks) * chunks * 16; }
// tile buffers: a ring of three (two tiles of DMA look-ahead) when they fit, else two
__host__ __device__ constexpr int gemm_nbuf(int chunks) {
    return (LS_GEMM_RING3 && 3 * gemm_tile_bytes(chunks) <= LS_GEMM_LDS_BYTES) ? 3 : 2;
}

// Register budget (a build-time fact, checked by tests/test_abi.py): the full pass and the fused
// launch of the two-accumulator geometries (rows <= 768 bytes) need <= 232 registers, so two waves per
// SIMD leave 48 of its 512 for a third, small wave - the one-wave select kernel (ls_wsel.hip) then
// runs INSIDE a resident pass (tools/coresidency_probe.hip).

// ---- queries -> fp16 MFMA B fragments, normalised if asked, zero padded ---------------------------
// Output layout = the order in which ls_gemm_filter_kernel consumes it: the 16-byte chunk c of
// query q (tile qt, wave w, group g2, li = q % 16) is fragment (qt, w, g2, kk = c / 4), lane
// (c % 4) * 16 + li. One wave load of a B fragment is then one contiguous 1 KiB read.
// (WQ = waves of a workgroup that hold DIFFERENT queries: 8, or 4 in the row-split geometry where two waves
// share a query block and take a row half each)
__device__ __forceinline__ long long qfrag_chunk(int q, int c, int QG, int KS, int WQ) {
    const int QPW = 16 * QG, QT = WQ * QPW;
    const int qt = q / QT, w = (q % QT) / QPW, g2 = (q % QPW) / 16, li = q % 16;
    return ((((long long)(qt * WQ + w) * QG + g2) * KS + (c >> 2)) << 6) + ((c & 3) << 4) + li;
}
// One wave per query (4 queries per block): the norm is the library's canonical wave reduction
// (ls_wave_sumsq), every lane converts and stores whole 16-byte chunks. The raw fp32 queries are
// also copied into the call's own slot (`qkeep`): a later repair of a flagged query must not
// depend on the caller keeping its query buffer alive.
__global__ __launch_bounds__(256) void ls_prep_f16_kernel(const float* __restrict__ qin,
                                                          u32x4* __restrict__ qout,
                                                          float* __restrict__ qkeep, int nq,
                                                          int nq_pad, int d, int d_pad, int QG, int WQ,
                                                          int normalize, u32* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
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
        qout[qfrag_chunk(qi, c, QG, KS, WQ)] = __builtin_bit_cast(u32x4, h);
    }
}

int ls_launch_prep_f16(const float* d_q, void* d_qh, float* d_qkeep, int64_t nq, int64_t nq_pad,
                       const ls_geom& g, bool normalize, u32* d_overflow, hipStream_t s) {
    // one-wave workgroups: 37 registers and no LDS, i.e. a wave that fits beside a resident MFMA pass
    // (tools/coresidency_probe.hip), whatever SIMD has room for it
    hipLaunchKernelGGL(ls_prep_f16_kernel, dim3((unsigned)nq_pad), dim3(64), 0, s, d_q,
                       (u32x4*)d_qh, d_qkeep, (int)nq, (int)nq_pad, g.d, g.d_pad, ls_gemm_qg(g), LS_GEMM_WAVES / ls_gemm_rs(g),
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
// The register-starved geometries (1.5 KiB rows: 14 registers spilled with four) keep the
// lane's best TWO sample scores when the query has so many lanes that two of its best j
// rarely share one (ls_launch_gemm_filter); the other two slots of the record stay "no
// sample". Like the top-4, a shorter list can only lower tau.
__device__ __forceinline__ void top4_insert(float (&t)[2], float v) {
    const float a = fmaxf(v, t[1]);
    t[1] = fminf(a, t[0]);
    t[0] = fmaxf(a, t[0]);
}
__device__ __forceinline__ uint4 top4_keys(const float (&t)[2]) {
    return make_uint4(t[0] == -FLT_MAX ? 0u : ls_ord(t[0]), t[1] == -FLT_MAX ? 0u : ls_ord(t[1]), 0u, 0u);
}

struct ls_gemm_out {
    uint2* queues;     // [nq_pad][nsplits][4][LS_GEMM_QCAP] (score bits, slice-relative row)
    u32* counts;       // [nq_pad][nsplits][4]
    u32* overflow;     // [nq_pad] per-query repair flag
    u32* sample_top;   // [nq_pad][nsplits][4][4] (sample pass)
    // fused launches only (LS_GEMM_FUSED): behind its own full pass the launch runs the sample
    // phase of the NEXT batch (sample_top is then the next batch's buffer)
    const u32x4* qh_next;  // the next batch's prepared queries (same geometry as this batch)
    int nq_next;
    int sample_stride;     // the sample phase visits every sample_stride-th tile of a slice
};

// "at most n vector-memory operations outstanding" (gfx9 encoding: vmcnt in bits 3:0 and 15:14;
// expcnt and lgkmcnt fields left at "no wait")
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt immediate");
    __builtin_amdgcn_s_waitcnt(0x0F70 | (N & 15) | ((N >> 4) << 14));
}

// compile-time phase tags for the generic lambdas below
struct phase_sample { static constexpr bool value = true; };
struct phase_full { static constexpr bool value = false; };

#ifndef LS_GEMM_PF_QG4
#define LS_GEMM_PF_QG4 1  // A-fragment look-ahead of the sample pass and of the 64-queries-per-wave shape (registers)
#endif
#define LS_GEMM_PASS 0    // full pass against a given tau
#define LS_GEMM_SAMPLE 1  // sample pass: best sample scores per lane
#define LS_GEMM_FUSED 2   // full pass of this batch, then the sample phase of the NEXT batch

// TOPN: sample scores kept per lane. MODE stays the last parameter: the profile tooling tells the
// launches apart by the ", 0>" / ", 1>" / ", 2>" tail of the kernel name.
// NT: the tile DMA carries the non-temporal hint. Right when ONE workgroup reads each corpus slice
// once (a single query tile, config 4: -1.5 %); wrong when the query tiles of a slice share it
// through their XCD's L2 (config 3: +3.8 %).
//
// LS_GEMM_FUSED (round 4): this batch's full pass, then the NEXT batch's sample phase in the same
// launch. Under LS_FLAG_PIPELINE the sample pass of batch i+1 used to sit between the passes of
// batches i and i+1 - the same 8-wave, 96 KB kernel, so it could only start on CUs pass i had left,
// and it paid its own launch, B-fragment loads and boundary. Folded behind pass i it costs its tiles
// only; between two passes remain the tau kernel and two kernel boundaries. No workgroup ever waits
// for another one (an earlier form of this launch ran the sample phase FIRST and met the tile's
// workgroups on arrival counters inside the kernel: 28 us of start-up, waits and cross-XCD reads of
// freshly written-through scores per launch against 23 us for the two kernels it replaced -
// tools/fused_phases.py, profiles/ab/r04_c3_fused_forms.txt).
// RS (round 5, the "64 queries per wave" shape for stored rows of 768 bytes): RS = 2 splits the tile's ROWS over
// wave pairs - waves 0-3 take the first half of every tile, waves 4-7 the second, wave w and w + 4 hold the same
// QG = 4 query groups - so every A fragment read from LDS feeds 4 MFMAs instead of 2 (LDS reads per MFMA 0.5 ->
// 0.25; profiles/ab/r05_tile_shape.txt prices the reads at 6 % of the pass, by clock). B is then 192 VGPRs: one
// accumulator set, row blocks in sequence (the config-4 loop). The two halves of a tile come from TWO adjacent
// corpus slices (as in ls_gemm32.hip), so a (query, slice, quarter) queue still has exactly one producer lane
// and the tau / select kernels are shared unchanged; a launch then covers 2 x workgroups / query-tiles slices.
template <int CHUNKS, int QG, int RS, int TOPN, bool NT, int MODE>
__device__ __forceinline__ void gemm_filter_body(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride_arg, const ls_gemm_out& out) {
    // (nq is re-pointed at the next batch's query count by a fused launch's second phase)
    constexpr bool SAMPLE = MODE == LS_GEMM_SAMPLE;
    constexpr bool FUSED = MODE == LS_GEMM_FUSED;
    constexpr int TM = gemm_tm(CHUNKS);    // corpus rows per LDS tile
    constexpr int TMH = TM / RS;           // ... of which one slice contributes this many (a wave's rows)
    constexpr int NRB = TMH / 16;          // 16-row MFMA blocks per wave and tile
    constexpr int WQ = LS_GEMM_WAVES / RS; // waves holding different queries
    constexpr int KS = CHUNKS / 4;         // k-steps: 32 fp16 = 4 chunks each
    constexpr int QPW = 16 * QG;           // queries per wave
    constexpr int NV = NRB * QG * 4;       // filter values per lane per tile
    // B fragments of QG groups take KS*QG*4 registers. When that leaves too little for two
    // accumulator sets, one set is kept and filtered between a tile's k-loop and the next one's.
    constexpr bool ONE_ACC = KS * QG * 4 >= 128;
    constexpr int CPK = (NV + KS - 1) / KS;  // two sets: checks interleaved per k-step
    constexpr int ROW_BYTES = CHUNKS * 16;
    constexpr int TILE_CHUNKS = TM * CHUNKS;
    constexpr int TILE_BYTES = TILE_CHUNKS * 16;
    constexpr int LOADS = TILE_CHUNKS / LS_GEMM_THREADS;  // 16-byte DMA loads per thread per tile
    constexpr int NBUF = gemm_nbuf(CHUNKS);
    constexpr int cap = LS_GEMM_QCAP;
    static_assert(TILE_CHUNKS % LS_GEMM_THREADS == 0, "tile must split evenly over the threads");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // the tile ring

    const int tid = threadIdx.x, lane = tid & 63;
#ifdef LS_GEMM_TIMING  // developer instrumentation: start / end tick (100 MHz) of every workgroup
    const unsigned long long t_start = wall_clock64();
#endif
    LS_SSTAMP_S(0);
    LS_SSTAMP_F(0);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar DMA addressing
    const int qd = lane >> 4, li = lane & 15;  // quarter (k-chunk / row group), index in group
    int wsplit, qt;  // the workgroup's slice (RS = 2: slice PAIR) and query tile
    wg_coords((int)blockIdx.x, nqt, &wsplit, &qt);
    const int nsplits = RS * ((int)gridDim.x / nqt);
    const int rs = RS > 1 ? wave / WQ : 0;   // row half this wave computes (= its slice of the pair)
    const int wq = RS > 1 ? wave % WQ : wave;  // query block of the wave
    const int split = wsplit * RS + rs;
    const long long r_begin = (long long)split * rows_per_split;
    long long r_end = r_begin + rows_per_split;
    if (r_end > n) r_end = n;
    int ntiles_all = 0;  // of the longest slice of the workgroup (the tile loop is common: barriers)
#pragma unroll
    for (int h = 0; h < RS; ++h) {
        const long long b0 = (long long)(wsplit * RS + h) * rows_per_split;
        long long e0 = b0 + rows_per_split;
        if (e0 > n) e0 = n;
        const int th = b0 < e0 ? (int)((e0 - b0 + TMH - 1) / TMH) : 0;
        ntiles_all = th > ntiles_all ? th : ntiles_all;
    }
    // the slice whose rows this wave's DMA pieces fetch: piece rows 0 .. TMH-1 are the first slice's
    const long long dma_begin = RS > 1 ? (long long)(wsplit * RS + (wave * (TM * CHUNKS / LS_GEMM_THREADS) * 64) / (TMH * CHUNKS)) * rows_per_split
                                       : r_begin;
    // the phase being run: every tile_stride-th tile of the slice, nt of them (a fused launch runs
    // the sample phase first and resets both for the full pass)
    int tile_stride = tile_stride_arg;
    int nt = (ntiles_all + tile_stride - 1) / tile_stride;

    // ---- corpus tiles: HBM/L2 -> LDS by DMA (global_load_lds, 16 B per lane) ------------------
    // Wave w, load j fills the 64 consecutive LDS chunks starting at (w*LOADS + j)*64: chunk Lc
    // is tile row r = Lc / CHUNKS, slot sl = Lc % CHUNKS and receives SOURCE chunk sl ^ (r & 15)
    // (the DMA destination is lane-linear, so the swizzle goes on the source address). The HBM
    // copy is padded with zero rows past n (ls_api.hip): no clamping.
    // 1.5 KiB rows (config 4) are 1.5 DMA pieces each and the kernel has no registers to keep six
    // per-lane offsets: the tile is stored with logical rows r and r + 16 ADJACENT in LDS
    // (physical row 2*(r & 15) + (r >> 4)); wave w then stages {w, w+16} and {w+8, w+24} as two
    // runs of three whole pieces, and since both rows of a run share the swizzle key r & 15 every
    // source offset is  run base (scalar) + (lane ^ key) + constant : ~12 VALU per tile instead
    // of ~55 for the generic divide-by-row-length form. PAIRED changes a_frag's row stride too.
    constexpr bool PAIRED = CHUNKS == 96 && TM == 32 && 16 % LS_GEMM_WAVES == 0;
    // Other register-starved instantiations (ONE_ACC) recompute the generic per-lane offsets for
    // every tile behind an opaque copy of the lane id: hoisted out of the tile loop they would be
    // spilled and every reload would wait on the memory pipe.
    constexpr bool LEAN = ONE_ACC || LS_GEMM_LEAN;  // per-tile recomputation instead of registers
    int goff[(LEAN || PAIRED) ? 1 : LOADS];
    if constexpr (!LEAN && !PAIRED) {
#pragma unroll
        for (int j = 0; j < LOADS; ++j) {
            const int Lc = (wave * LOADS + j) * 64 + lane;
            const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
            goff[j] = (r % TMH) * CHUNKS + (sl ^ (r & 15));
        }
    }
    static_assert(RS == 1 || ((TMH * CHUNKS) % (64 * (TM * CHUNKS / LS_GEMM_THREADS)) == 0 && !(CHUNKS == 96)),
                  "row split: a wave's DMA pieces must lie in one half of the tile");
    // piece j (0 .. LOADS-1) of tile ti -> the tile buffer at LDS byte offset bufoff
    auto stage_piece = [&](int ti, int bufoff, int j) {
#if LS_ABL_NODMA
        if (!SAMPLE && ti > 0) return;
#endif
        long long row0 = dma_begin + (long long)ti * TMH;
        if (RS > 1) {  // a short (last) slice of the pair runs out of rows before the other: stay inside the pad rows
            const long long lim = n + LS_CORPUS_PAD_ROWS - TMH;
            row0 = row0 < lim ? row0 : lim;
        }
        const u32x4* base = corpus + row0 * CHUNKS;
        int lane_v = lane;
        if constexpr (LEAN || PAIRED) asm volatile("" : "+v"(lane_v));
        if constexpr (PAIRED) {
            const int run = j / 3, which = j % 3;
            const int r0 = wave + LS_GEMM_WAVES * run;  // logical rows r0 and r0 + 16, key r0
            const int t = lane_v ^ r0;
            const u32x4* rowp = base + r0 * CHUNKS;  // wave-uniform
            unsigned char* dst = smem + bufoff + (2 * r0) * ROW_BYTES + which * 1024;
            const int off = which == 0 ? t
                          : which == 1 ? (lane_v < 32 ? 64 + t : 16 * CHUNKS + t - 32)
                                       : 16 * CHUNKS + 32 + t;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(rowp + off), (lds_ptr_t)dst, 16, 0, NT ? 2 : 0);
        } else {
            int off;
            if constexpr (LEAN) {
                const int Lc = (wave * LOADS + j) * 64 + lane_v;
                const int r = Lc / CHUNKS, sl = Lc % CHUNKS;
                off = (r % TMH) * CHUNKS + (sl ^ (r & 15));
            } else {
                off = goff[j];
            }
            unsigned char* dst = smem + bufoff + (wave * LOADS + j) * 1024;
            __builtin_amdgcn_global_load_lds((glb_ptr_t)(base + off), (lds_ptr_t)dst, 16, 0, NT ? 2 : 0);
        }
    };
    auto stage = [&](int ti, int bufoff) {  // all pieces of tile ti, back to back
#pragma unroll
        for (int j = 0; j < LOADS; ++j) stage_piece(ti, bufoff, j);
    };
    // Hand-over between tiles: this wave's pieces of the NEXT tile have landed (`newer` = a
    // younger tile's LOADS pieces may still be in flight: loads return in order, so "<= LOADS
    // outstanding" means every older load is done, whatever stores were issued in between), then
    // the workgroup barrier: every wave's pieces are in LDS and every wave is done reading the
    // buffer that is refilled next. A bare s_barrier: __syncthreads() would drain vmcnt to 0.
    auto hand_over = [&](bool newer) {
        asm volatile("" ::: "memory");
        if (newer) wait_vmcnt<LOADS>(); else wait_vmcnt<0>();
#if !LS_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };
    // The hand-over in front of the first tile drains everything (vmcnt(0)). Tried and removed:
    // waiting only for the DMA ("at most QG*KS+QG younger loads outstanding") so that the first
    // tile starts while the query fragments stream in. It gained 0.3 % and was WRONG once the
    // compiler sank some fragment loads below the wait (they come from const __restrict__ memory,
    // an asm memory clobber does not pin them): fewer younger loads than counted, the wait passes
    // with DMA pieces still in flight, the first tile is read stale (caught by tests/test_fuzz_gpu.py
    // on two of three seeds). A counted vmcnt is only sound where this file issues every
    // vector-memory operation in between itself.
    auto first_hand_over = [&]() {
        asm volatile("" ::: "memory");
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // The first tile(s) are requested BEFORE the query fragments: the HBM round trip of tile 0
    // then overlaps the (L2-resident) query loads instead of queueing behind them.
    constexpr int NB_ALL = LS_GEMM_LDS_BYTES / TILE_BYTES;  // tiles that fit in LDS together
    const bool sample_upfront = SAMPLE && nt > 0 && nt <= NB_ALL && nt <= 3;
    // the tiles a phase requests before its first hand-over
    auto phase_prologue = [&]() {
        if (nt > 0) stage(0, 0);
        if (sample_upfront) {
            if (nt > 1) stage(tile_stride, TILE_BYTES);
            if (nt > 2) stage(2 * tile_stride, 2 * TILE_BYTES);
        } else if (NBUF == 3 && nt > 1) {
            stage(tile_stride, TILE_BYTES);
        }
    };
    phase_prologue();

    // B fragments: group qg holds query qt*8*QPW + wave*QPW + qg*16 + li; k-step kk -> chunk 4kk+qd.
    half8 bq[QG][KS];
    int qj[QG];
    float tauv[QG];
    const u32x4* qfrag[QG];
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2) {
        qj[g2] = (qt * WQ + wq) * QPW + g2 * 16 + li;
        // fragment-ordered by ls_prep_f16_kernel: each load below is one contiguous KiB per wave
        qfrag[g2] = qh + ((((long long)(qt * WQ + wq) * QG + g2) * KS) << 6) + lane;
        tauv[g2] = SAMPLE ? 0.0f : (LS_ABL_NOPASS ? FLT_MAX : tau[qj[g2]]);
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk)
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2) bq[g2][kk] = __builtin_bit_cast(half8, qfrag[g2][kk << 6]);
    // private queues of this lane (one per query group), contiguous per lane
    // entry = {score bits, row relative to the slice}; the select kernel turns it into a key
    // (kept as 32-bit entry offsets from out.queues - the whole queue array is nq_pad * nsplits KiB,
    // far below 2^32 entries - so that the append is base (scalar) + offset: two registers less than
    // two pointers, which is what keeps a fused launch inside 232 registers)
    u32 myq[QG];
    int cnt[QG];
    float top[QG][TOPN];  // sample scores kept per lane and query group
    auto init_full_state = [&]() {  // (a fused launch sets these up only after its sample phase)
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2) {
            myq[g2] = (u32)(queue_id(qj[g2], split, qd, nsplits) * cap);
            cnt[g2] = 0;
        }
    };
    init_full_state();
#pragma unroll
    for (int g2 = 0; g2 < QG; ++g2)
#pragma unroll
        for (int e = 0; e < TOPN; ++e) top[g2][e] = -FLT_MAX;

    // A fragment of k-step kk, row block rb: tile row rb*16 + li, chunk (4kk + qd) ^ li.
    // (4kk + qd) & 15 takes 4 values per lane: 4 precomputed byte offsets + immediates.
    int lo4[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
        lo4[m] = li * (PAIRED ? 2 : 1) * ROW_BYTES + (((4 * m + qd) ^ li) * 16);
    auto a_frag = [&](int bufoff, int rb, int kk) -> half8 {  // bufoff: byte offset of the tile
        const u32x4 v = *reinterpret_cast<const u32x4*>(smem + lo4[kk & 3] + bufoff + rs * (TMH * ROW_BYTES) +
                                                        rb * (PAIRED ? 1 : 16) * ROW_BYTES + (kk >> 2) * 256);
        return __builtin_bit_cast(half8, v);
    };

    // ---- one score of a finished tile -------------------------------------------------------------
    // `ph` (phase_sample / phase_full) selects at compile time what happens to a score.
    // The append is the hot slow path (a wave enters it for ~1 check in 5): no key building, no
    // bounds logic here. Padded queries carry tau = FLT_MAX and never pass; zero-pad rows past n
    // are dropped by the select kernel; a full queue keeps overwriting its last slot while the
    // count runs on, which is how the overflow is seen at the end.
    auto check1 = [&](auto ph, float s, int g2, int lrow) {
        if constexpr (decltype(ph)::value) {
            // NaN never enters (fmaxf/fminf drop it); padded queries and zero-pad rows are masked
            top4_insert(top[g2], (qj[g2] < nq && r_begin + lrow < r_end) ? s : -FLT_MAX);
        } else {
            if (s >= tauv[g2]) {
                const int slot = cnt[g2] < cap ? cnt[g2] : cap - 1;
                const u32 qo = LEAN ? (u32)(queue_id(qj[g2], split, qd, nsplits) * cap) : myq[g2];
#if LS_GEMM_APPEND_SC1  // variant builds: write-through appends (nothing dirty in L2 at the kernel's end)
                __hip_atomic_store(reinterpret_cast<u64*>(out.queues) + (qo + (u32)slot),
                                   ((u64)(u32)lrow << 32) | __float_as_uint(s), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
#else
                out.queues[qo + (u32)slot] = make_uint2(__float_as_uint(s), (u32)lrow);
#endif
                ++cnt[g2];
            }
        }
    };
    auto check = [&](auto ph, const f32x4v (&acc)[NRB][QG], int e, int lrow0) {  // e -> (block, group, reg)
        const int rb = e / (QG * 4), g2 = (e / 4) % QG, reg = e % 4;
        check1(ph, acc[rb][g2][reg], g2, lrow0 + rb * 16 + reg);  // lrow0 already includes 4*qd
    };

    // One tile: NRB*QG independent accumulator chains advance together, one k-step at a time.
    // Two sets: the PREVIOUS tile's accumulators are filtered CPK elements per k-step, in the
    // shadow of the matrix pipe.
    // One set, two query groups (SEQ_RB, config 4's geometry): the tile's two row blocks run one
    // after the other; while block rb's two chains advance, the 8 scores of the block that
    // finished just before (block rb-1 of this tile, or the last block of the previous tile,
    // still sitting in its registers) are filtered, one every KS/8 k-steps. The filter hides in
    // the MFMA stream again without a second accumulator set.
    // One set, one query group (2 KiB rows): a single chain per row block would stall on its own
    // MFMA latency, so all blocks advance together and the filter runs before the k-loop.
    constexpr bool SEQ_RB = ONE_ACC && QG >= 2;
    // SEQ_RB also spreads the DMA pieces of the tile that is fetched next over the first row
    // block's k-steps (one piece every other k-step) instead of issuing all of them right behind
    // the barrier, where both waves of a SIMD would do so at once with the matrix pipe idle.
    auto run_tile = [&](auto ph, f32x4v (&cur)[NRB][QG], const f32x4v (&prev)[NRB][QG], bool have_prev,
                        int prev_row0, int cur_row0, int bufoff, bool stage_more = false,
                        int stage_ti = 0, int stage_buf = 0) {
        constexpr bool SMP = decltype(ph)::value;
        if constexpr (SMP && FUSED) {
            // The sample phase of a fused launch is a handful of tiles, and whatever it keeps in
            // registers comes on top of the full pass's state (the B fragments alone are KS*QG*4
            // registers): ONE accumulator set, and the tile's own scores filtered right behind its
            // k-loop instead of inside the next tile's. The launch stays at the full pass's register
            // count - which is what lets the one-wave select kernel (ls_wsel.hip) run beside it.
            half8 a[2][NRB];  // A fragments one k-step ahead (the second accumulator set's registers are free here)
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
                        cur[rb][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk & 1][rb], bq[g2][kk], c, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < NV; ++e) check(ph, cur, e, cur_row0);
            return;
        }
        if constexpr (SEQ_RB) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                const int pb = rb == 0 ? NRB - 1 : rb - 1;  // block whose scores are filtered now
                const bool have = rb == 0 ? have_prev : true;
                const int prow0 = (rb == 0 ? prev_row0 : cur_row0) + pb * 16;
                constexpr int PF = (SMP || QG >= 4) ? LS_GEMM_PF_QG4 : LS_GEMM_PF, NA = PF + 1;  // A fragments are read PF k-steps ahead (the sample pass has 8 registers fewer)
                half8 a[NA];
#pragma unroll
                for (int p0 = 0; p0 < PF; ++p0) a[p0] = a_frag(bufoff, rb, p0);
#pragma unroll
                for (int kk = 0; kk < KS; ++kk) {
                    if (kk + PF < KS) a[(kk + PF) % NA] = a_frag(bufoff, rb, kk + PF);
#pragma unroll
                    for (int g2 = 0; g2 < QG; ++g2) {
                        f32x4v c;
                        if (kk == 0) {
                            c[0] = 0.0f; c[1] = 0.0f; c[2] = 0.0f; c[3] = 0.0f;
                        } else {
                            c = cur[rb][g2];
                        }
                        cur[rb][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk % NA], bq[g2][kk], c,
                                                                            0, 0, 0);
                    }
                    if (have) {
#pragma unroll
                        for (int e = 0; e < QG * 4; ++e)
                            if ((e * KS) / (QG * 4) == kk) check1(ph, cur[pb][e / 4][e % 4], e / 4, prow0 + e % 4);
                    }
                    constexpr int SD = 2 * LOADS <= KS ? 2 : 1;  // a piece every SD-th k-step
                    if (rb == 0 && kk % SD == SD - 1 && kk / SD < LOADS && stage_more)
                        stage_piece(stage_ti, stage_buf, kk / SD);
                }
            }
            return;
        }
        if (ONE_ACC && have_prev) {
#pragma unroll
            for (int e = 0; e < NV; ++e) check(ph, prev, e, prev_row0);
        }
        half8 a[2][NRB];  // A fragments are read one k-step ahead of their MFMAs
        // (LS_ABL_LDSREADS, timing ablation with wrong results: only the first NRB / LS_ABL_LDSREADS
        // row blocks are read from LDS, the other blocks' MFMAs reuse those registers - their chains
        // start from 1, 2, 4 instead of 0 so that the compiler cannot merge them: no extra instruction.
        // Build it together with LS_ABL_NOPASS (the shifted scores would all pass the filter): what a
        // tile shape with that many times fewer LDS reads per MFMA could gain at most)
        constexpr int NRD = LS_ABL_LDSREADS > 1 ? (NRB / LS_ABL_LDSREADS > 0 ? NRB / LS_ABL_LDSREADS : 1) : NRB;  // row blocks really read
        auto a_load = [&](half8 (&dst)[NRB], int kk) {
#pragma unroll
            for (int rb = 0; rb < NRD; ++rb) dst[rb] = a_frag(bufoff, rb, kk);
        };
        a_load(a[0], 0);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) a_load(a[(kk + 1) & 1], kk + 1);
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
#pragma unroll
                for (int g2 = 0; g2 < QG; ++g2) {
                    f32x4v c;
                    if (kk == 0) {
                        const float c0 = rb / NRD == 0 ? 0.0f : rb / NRD == 1 ? 1.0f : rb / NRD == 2 ? 2.0f : 4.0f;  // inline constants
                        c[0] = c0; c[1] = c0; c[2] = c0; c[3] = c0;
                    } else {
                        c = cur[rb][g2];
                    }
                    cur[rb][g2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[kk & 1][rb % NRD], bq[g2][kk],
                                                                        c, 0, 0, 0);
                }
            }
            if (!ONE_ACC && have_prev) {
#pragma unroll
                for (int c2 = 0; c2 < CPK; ++c2)
                    if (kk * CPK + c2 < NV) check(ph, prev, kk * CPK + c2, prev_row0);
            }
            if (!SEQ_RB && kk < LOADS && stage_more)
                stage_piece(stage_ti, stage_buf, kk);
        }
    };

    f32x4v accA[NRB][QG], accB[NRB][QG];  // two sets alternate between tiles (ONE_ACC: accA only)
    // The tile loop carries no "is there a previous tile" / "is there a next tile" branches (they
    // cut the loop body into basic blocks the scheduler cannot move MFMAs across): the
    // accumulators start at -inf, which passes no threshold (tau >= -FLT_MAX) and leaves a
    // top-4 list unchanged, so the first tile may filter them like a real previous tile; and
    // the full pass always requests the next tile, past the slice end too (the next slice's
    // rows or the zero pad rows, LS_CORPUS_PAD_ROWS >= 2 tiles: fetched, never read).
    auto reset_acc = [&]() {
#if LS_GEMM_STRAIGHT
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int g2 = 0; g2 < QG; ++g2)
#pragma unroll
                for (int e = 0; e < 4; ++e) accA[rb][g2][e] = accB[rb][g2][e] = -INFINITY;
#endif
    };
    reset_acc();
    auto tile_row0 = [&](int i) { return (i * tile_stride) * TMH + 4 * qd; };  // slice-relative
    auto flush_last = [&](auto ph, const f32x4v (&acc)[NRB][QG]) {  // the last tile still has to be filtered
        const int row0 = tile_row0(nt - 1);
#pragma unroll
        for (int e = 0; e < NV; ++e)
            if (!SEQ_RB || e / (QG * 4) == NRB - 1) check(ph, acc, e, row0);  // SEQ_RB: last block only
    };
    // The sample pass visits only a few tiles, so their DMA latencies would be paid one by one:
    // when all of them fit in LDS together they are fetched up front and consumed back to back.
    if constexpr (SAMPLE) {
        if (sample_upfront) {
            first_hand_over();
            LS_SSTAMP_S(1);
            if constexpr (ONE_ACC) {
                for (int i = 0; i < nt; ++i)
                    run_tile(phase_sample{}, accA, accA, i > 0, tile_row0(i - 1), tile_row0(i), i * TILE_BYTES);
                flush_last(phase_sample{}, accA);
            } else {
                run_tile(phase_sample{}, accA, accB, false, 0, 0, 0);
                if (nt > 1) run_tile(phase_sample{}, accB, accA, true, tile_row0(0), 0, TILE_BYTES);
                if (nt > 2) run_tile(phase_sample{}, accA, accB, true, tile_row0(1), 0, 2 * TILE_BYTES);
                if ((nt - 1) & 1) flush_last(phase_sample{}, accB); else flush_last(phase_sample{}, accA);
            }
            LS_SSTAMP_S(2);
#pragma unroll
            for (int g2 = 0; g2 < QG; ++g2)
                reinterpret_cast<uint4*>(out.sample_top)[queue_id(qj[g2], split, qd, nsplits)] =
                    top4_keys(top[g2]);
            LS_SSTAMP_S(3);
            return;
        }
    }
    // ---- the tile loop of one phase. Ring of NBUF buffers; tile i sits in buffer i % NBUF. With
    // three buffers tile i+2 is requested at the top of tile i (its buffer was last read during
    // tile i-1, which every wave has left through the barrier); with two, tile i+1.
    constexpr int AHEAD = NBUF - 1;
    int b_cur = 0, b_new = AHEAD * TILE_BYTES;  // LDS byte offsets of tile i and of tile i + AHEAD
    // `fresh`: the phase's first tile(s) were requested by phase_prologue (ring restarts at buffer 0).
    // Not fresh (a fused launch's sample phase, two buffers): the full pass's last iteration already
    // requested sample tile 0 as its "next tile" and handed it over - the ring just goes on.
    auto run_phase = [&](auto ph, bool fresh = true) {
        constexpr bool SMP = decltype(ph)::value;
        if (fresh) {
            first_hand_over();
            b_cur = 0;
            b_new = AHEAD * TILE_BYTES;
        }
        if constexpr (SMP) LS_SSTAMP_F(1);
        auto advance = [&](int& b) { b = b + TILE_BYTES == NBUF * TILE_BYTES ? 0 : b + TILE_BYTES; };
        auto one_tile = [&](f32x4v (&cur)[NRB][QG], const f32x4v (&prev)[NRB][QG], int i) {
            constexpr bool always = LS_GEMM_STRAIGHT && !SMP && (AHEAD + 1) * TMH <= LS_CORPUS_PAD_ROWS;
            const bool more = always ? true : i + AHEAD < nt;
            // the next tile's DMA pieces are issued between this tile's k-steps (run_tile), not in
            // one burst behind the barrier
            constexpr bool spread = (SEQ_RB || !SMP) && LOADS <= KS;
            // (a fused launch's full pass: the request past the slice's last tile fetches the sample
            // phase's first tile - tile 0 of the slice - instead of rows nobody reads)
            const int ti_next = (FUSED && !SMP && NBUF == 2 && i + AHEAD >= nt) ? 0 : (i + AHEAD) * tile_stride;
            if (more && !spread) stage(ti_next, b_new);
            run_tile(ph, cur, prev, LS_GEMM_STRAIGHT ? true : i > 0, tile_row0(i - 1), tile_row0(i), b_cur,
                     more && spread, ti_next, b_new);
            // tile i+1 must be complete before anyone reads it. NBUF == 3: only when a younger tile
            // was requested in this iteration may LOADS pieces stay in flight.
            hand_over(NBUF == 3 && more);
            advance(b_cur);
            advance(b_new);
        };
        if constexpr (SMP && FUSED) {  // every tile filters itself (run_tile): nothing left to flush
            for (int i = 0; i < nt; ++i) one_tile(accA, accA, i);
        } else if constexpr (ONE_ACC) {
            for (int i = 0; i < nt; ++i) one_tile(accA, accA, i);
            if (nt > 0) flush_last(ph, accA);
        } else {
            for (int i = 0; i < nt; i += 2) {
                one_tile(accA, accB, i);
                if (i + 1 < nt) one_tile(accB, accA, i + 1);
            }
            if (nt > 0) {
                if ((nt - 1) & 1) flush_last(ph, accB); else flush_last(ph, accA);
            }
        }
    };

    if constexpr (FUSED) {
        // ======== this batch's full pass, exactly as a LS_GEMM_PASS launch ============================
        run_phase(phase_full{});
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2) {
            const long long qid = queue_id(qj[g2], split, qd, nsplits);
            out.counts[qid] = (u32)(cnt[g2] < cap ? cnt[g2] : cap);
            if (cnt[g2] > cap) out.overflow[qj[g2]] = 1u;
        }
        LS_SSTAMP_F(2);
        // ======== the NEXT batch's sample phase over this slice's sample tiles =====================
        // Same geometry as this batch (the host fuses only then). Its prepared queries were written
        // by a prep kernel that completed before this launch started (stream wait on its event).
        tile_stride = out.sample_stride;
        nt = (ntiles_all + tile_stride - 1) / tile_stride;
        // the straight-line full pass (two buffers) has already fetched and handed over sample tile 0
        constexpr bool PREFETCHED = LS_GEMM_STRAIGHT && NBUF == 2 && 2 * TMH <= LS_CORPUS_PAD_ROWS;
        if constexpr (!PREFETCHED) {
            __builtin_amdgcn_s_barrier();  // every wave has left the tile ring
            phase_prologue();
        }
        {
            const u32x4* qn = out.qh_next;
            asm volatile("" : "+s"(qn));  // a different buffer: nothing of the first load may be reused
#pragma unroll
            for (int g2 = 0; g2 < QG; ++g2)
                qfrag[g2] = qn + ((((long long)(qt * WQ + wq) * QG + g2) * KS) << 6) + lane;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk)
#pragma unroll
                for (int g2 = 0; g2 < QG; ++g2) bq[g2][kk] = __builtin_bit_cast(half8, qfrag[g2][kk << 6]);
        }
        nq = out.nq_next;
        run_phase(phase_sample{}, !PREFETCHED);
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2)
            reinterpret_cast<uint4*>(out.sample_top)[queue_id(qj[g2], split, qd, nsplits)] = top4_keys(top[g2]);
    } else if constexpr (SAMPLE) {
        run_phase(phase_sample{});
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2)
            reinterpret_cast<uint4*>(out.sample_top)[queue_id(qj[g2], split, qd, nsplits)] = top4_keys(top[g2]);
    } else {
        run_phase(phase_full{});
#pragma unroll
        for (int g2 = 0; g2 < QG; ++g2) {
            const long long qid = queue_id(qj[g2], split, qd, nsplits);
            out.counts[qid] = (u32)(cnt[g2] < cap ? cnt[g2] : cap);
            if (cnt[g2] > cap) out.overflow[qj[g2]] = 1u;
        }
    }
    LS_SSTAMP_F(7);
#ifdef LS_GEMM_TIMING
    if (!SAMPLE && !FUSED && tid == 0) {  // the sample tops are dead once tau has been computed
        unsigned long long* life = reinterpret_cast<unsigned long long*>(out.sample_top);
        life[2 * blockIdx.x] = t_start;
        life[2 * blockIdx.x + 1] = wall_clock64();
    }
#endif
}

// The kernel proper: a thin entry point around the body (clang's amdgpu_num_vgpr attribute is not
// enforced by this toolchain, so the 232-register budget of gemm_vgpr_budget is checked on the built
// code object instead: tests/test_abi.py::test_register_budget_of_the_co_resident_kernels).
template <int CHUNKS, int QG, int TOPN, bool NT, int MODE>
__global__ __launch_bounds__(LS_GEMM_THREADS, LS_GEMM_WAVES_PER_SIMD) void ls_gemm_filter_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride, ls_gemm_out out) {
    gemm_filter_body<CHUNKS, QG, 1, TOPN, NT, MODE>(corpus, n, qh, nq, nqt, tau, rows_per_split, tile_stride, out);
}
#ifdef LS_VARIANT_RS2
// The row-split, 64-queries-per-wave shape (round 5; `make variant NAME=rs2 VFLAGS=-DLS_VARIANT_RS2`, debug option
// 18): exact, LDS / MFMA instruction ratio 0.25 instead of 0.5, but 21 registers spilled inside the tile loop -
// pass 185 us against 124 (profiles/ab/r05_tile_shape.txt). A measured loser: not in the shipped library.
template <int CHUNKS, int TOPN, bool NT, int MODE>
__global__ __launch_bounds__(LS_GEMM_THREADS, LS_GEMM_WAVES_PER_SIMD) void ls_gemm_filter_rs2_kernel(
    const u32x4* __restrict__ corpus, long long n, const u32x4* __restrict__ qh, int nq, int nqt,
    const float* __restrict__ tau, long long rows_per_split, int tile_stride, ls_gemm_out out) {
    gemm_filter_body<CHUNKS, 4, 2, TOPN, NT, MODE>(corpus, n, qh, nq, nqt, tau, rows_per_split, tile_stride, out);
}
#endif

#ifdef LS_VARIANT_RS2
static inline bool gemm_is_qg4(const ls_geom& g) { return g.qg4 != 0 && g.chunks == 48 && g.elem == 2; }
#else
static inline bool gemm_is_qg4(const ls_geom&) { return false; }
#endif
int ls_gemm_rs(const ls_geom& g) { return gemm_is_qg4(g) ? 2 : 1; }
int ls_gemm_qg(const ls_geom& g) { return gemm_is_qg4(g) ? 4 : gemm_qg(g.chunks); }
int ls_gemm_qt(const ls_geom& g) { return (LS_GEMM_WAVES / ls_gemm_rs(g)) * 16 * ls_gemm_qg(g); }
int ls_gemm_tile_rows(const ls_geom& g) { return gemm_tm(g.chunks) / ls_gemm_rs(g); }

// One launcher for the three kinds of launch. `fz` non-null (with d_tau) = a fused launch (this
// batch's full pass + the next batch's sample phase); else d_tau null = sample pass, non-null = full pass. ev_start /
// ev_stop (either may be null) are attached to the dispatch itself (hipExtLaunchKernelGGL): they
// cost no extra packet on the stream, and a pair of timing events brackets exactly the kernel.
int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, const ls_gemm_bufs& b,
                          bool sample_top2, hipStream_t s, const ls_gemm_fuse* fz, hipEvent_t ev_start,
                          hipEvent_t ev_stop) {
    const int nqt = (int)(nq_pad / ls_gemm_qt(g));
    const dim3 grid((unsigned)(nsplits / ls_gemm_rs(g) * nqt));
    ls_gemm_out o{};
    o.queues = (uint2*)b.d_queues;
    o.counts = b.d_counts;
    o.overflow = b.d_overflow;
    o.sample_top = b.d_sample_top;
    if (fz) {
        o.qh_next = (const u32x4*)fz->d_qh_next;
        o.nq_next = (int)fz->nq_next;
        o.sample_stride = fz->sample_stride;
        o.sample_top = fz->d_sample_top_next;
    }
    // full pass / fused: the tile ring; sample pass: up to three tiles up front
    const size_t tile_bytes = (size_t)gemm_tile_bytes(g.chunks);
    const size_t smem = d_tau ? tile_bytes * gemm_nbuf(g.chunks)
                              : tile_bytes * (3 * tile_bytes <= LS_GEMM_LDS_BYTES ? 3 : 2);
#define LS_GEMM_LAUNCH(C, MODE, TOPN, NT)                                                         \
    {                                                                                             \
        auto kern = ls_gemm_filter_kernel<C, gemm_qg(C), TOPN, NT, MODE>;                         \
        static ls_attr_once once;                                                                 \
        if (int rc = ls_set_max_dynamic_lds(once, (const void*)kern, LS_GEMM_LDS_BYTES)) return rc; \
        hipExtLaunchKernelGGL(kern, grid, dim3(LS_GEMM_THREADS), smem, s, ev_start, ev_stop, 0,   \
                              (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq,  \
                              nqt, d_tau, (long long)rows_per_split, tile_stride, o);             \
        LS_HIP(hipGetLastError());                                                                \
        return LS_OK;                                                                             \
    }
#ifdef LS_VARIANT_RS2
#define LS_GEMM_LAUNCH_RS2(C, MODE, TOPN, NT)                                                     \
    {                                                                                             \
        auto kern = ls_gemm_filter_rs2_kernel<C, TOPN, NT, MODE>;                                 \
        static ls_attr_once once;                                                                 \
        if (int rc = ls_set_max_dynamic_lds(once, (const void*)kern, LS_GEMM_LDS_BYTES)) return rc; \
        hipExtLaunchKernelGGL(kern, grid, dim3(LS_GEMM_THREADS), smem, s, ev_start, ev_stop, 0,   \
                              (const u32x4*)d_corpus, (long long)n, (const u32x4*)d_qh, (int)nq,  \
                              nqt, d_tau, (long long)rows_per_split, tile_stride, o);             \
        LS_HIP(hipGetLastError());                                                                \
        return LS_OK;                                                                             \
    }
    if (gemm_is_qg4(g)) {  // 48-chunk rows, 64 queries per wave, rows split over wave pairs
        if (fz) {
            ls_set_error("batched path: the row-split shape has no fused launch");
            return LS_ERR_INVALID_ARG;
        }
        if (d_tau && nqt == 1) LS_GEMM_LAUNCH_RS2(48, LS_GEMM_PASS, 4, true)
        else if (d_tau) LS_GEMM_LAUNCH_RS2(48, LS_GEMM_PASS, 4, false)
        else if (sample_top2) LS_GEMM_LAUNCH_RS2(48, LS_GEMM_SAMPLE, 2, false)
        else LS_GEMM_LAUNCH_RS2(48, LS_GEMM_SAMPLE, 4, false)
    }
#undef LS_GEMM_LAUNCH_RS2
#endif
#define LS_GEMM_TOP2(C) (C / 4 * gemm_qg(C) * 4 >= 128 && sample_top2)
// (the fused launch exists for the two-accumulator geometries only: rows of up to 768 bytes. The
// register-starved ones spend ~1 % of a multi-millisecond batch outside the pass; a forced fused run of
// config 4's geometry also mis-thresholded every query - exact after repairs, 40x slower - and was
// not pursued)
#define LS_GEMM_CASE(C)                                                                  \
    if (g.chunks == C) {                                                                 \
        if constexpr (C <= 48) {                                                         \
            if (fz && nqt == 1) LS_GEMM_LAUNCH(C, LS_GEMM_FUSED, 4, true)                \
            else if (fz) LS_GEMM_LAUNCH(C, LS_GEMM_FUSED, 4, false)                      \
        }                                                                                \
        if (fz) {                                                                        \
            ls_set_error("batched path: no fused launch for %d-chunk rows", g.chunks);   \
            return LS_ERR_INVALID_ARG;                                                   \
        }                                                                                \
        if (d_tau && nqt == 1) LS_GEMM_LAUNCH(C, LS_GEMM_PASS, 4, true)                  \
        else if (d_tau) LS_GEMM_LAUNCH(C, LS_GEMM_PASS, 4, false)                        \
        else if (LS_GEMM_TOP2(C)) LS_GEMM_LAUNCH(C, LS_GEMM_SAMPLE, 2, false)            \
        else LS_GEMM_LAUNCH(C, LS_GEMM_SAMPLE, 4, false)                                 \
    }
#ifdef LS_GEMM_ONLY_CASE  // developer builds: one geometry (compile time of a register experiment)
    LS_GEMM_CASE(LS_GEMM_ONLY_CASE)
#else
    LS_GEMM_CASE(16) LS_GEMM_CASE(32) LS_GEMM_CASE(48) LS_GEMM_CASE(64)
    LS_GEMM_CASE(96) LS_GEMM_CASE(128)
#endif
#undef LS_GEMM_CASE
#undef LS_GEMM_TOP2
#undef LS_GEMM_LAUNCH
    ls_set_error("batched path: unsupported row geometry (%d chunks)", g.chunks);
    return LS_ERR_INVALID_ARG;
}

// ---- tau: j-th best sample score of each query --------------------------------------------------
// The sample phase left, for every (workgroup, lane, query group), the 4 (or 2) best sample scores
// that lane saw (ord() of the score, 0 = none). A query owns 4 lanes in each of its nsplits
// workgroups: 16*nsplits values, contiguous. Keeping only a few per lane can only LOWER the result
// (if one lane held more of the best j), i.e. let more rows through: tau is a speculative,
// verified threshold either way.
// Round 4: ONE WAVE per query, values in registers, no LDS and no barrier: the j-th largest by a
// bit-wise search over the top 16 bits of the order key (sign, exponent, 7 mantissa bits) - 16 steps
// of "how many values are >= candidate" (compares + one wave sum). The result is at most 0.8 % below
// the exact j-th sample score, the same value the two 8-bit radix passes of the earlier 256-thread
// kernel produced; it sits between two MFMA passes of a pipelined run, so its ~3 us are critical
// path (the earlier kernel: 5.6 us).
template <int P>  // uint4 loads per lane: 16 * nsplits / 256, rounded up to 4, 8, 16, 32
__global__ __launch_bounds__(64) void ls_tau_kernel(const u32* __restrict__ sample_top, int nsplits,
                                                    int nq, int j_rank, float* __restrict__ tau) {
    const int q = blockIdx.x, lane = threadIdx.x;
    if (q >= nq) {
        if (lane == 0) tau[q] = FLT_MAX;  // padded query: nothing passes
        return;
    }
    const int total4 = nsplits * 4;  // uint4s of this query: [slice][quarter]
    const uint4* src = reinterpret_cast<const uint4*>(sample_top) + (long long)q * total4;
    uint4 v[P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int idx = lane + j * 64;
        v[j] = idx < total4 ? src[idx] : make_uint4(0, 0, 0, 0);
    }
    auto count_ge = [&](u32 cand) -> u32 {
        u32 c = 0;
#pragma unroll
        for (int j = 0; j < P; ++j)
            c += (u32)(v[j].x >= cand) + (u32)(v[j].y >= cand) + (u32)(v[j].z >= cand) + (u32)(v[j].w >= cand);
        return wave_sum(c);
    };
    float out;
    if (count_ge(1u) < (u32)j_rank) {
        out = -FLT_MAX;  // fewer than j sample scores: no bound
    } else {
        u32 t = 0;
#pragma unroll 1
        for (int bit = 31; bit >= 16; --bit) {
            const u32 cand = t | (1u << bit);
            if (count_ge(cand) >= (u32)j_rank) t = cand;
        }
        out = ls_unord(t);
    }
    if (lane == 0) tau[q] = out;
}

int ls_launch_tau(const u32* d_sample_top, int nsplits, int64_t nq, int64_t nq_pad, int j_rank,
                  float* d_tau, hipStream_t s) {
    if (nsplits > LS_GEMM_MAX_SPLITS) {
        ls_set_error("batched path: too many slices for the tau kernel");
        return LS_ERR_INVALID_ARG;
    }
#define LS_TAU_LAUNCH(P)                                                                      \
    hipLaunchKernelGGL(ls_tau_kernel<P>, dim3((unsigned)nq_pad), dim3(64), 0, s, d_sample_top, \
                       nsplits, (int)nq, j_rank, d_tau)
    if (nsplits <= 64) LS_TAU_LAUNCH(4);
    else if (nsplits <= 128) LS_TAU_LAUNCH(8);
    else if (nsplits <= 256) LS_TAU_LAUNCH(16);
    else LS_TAU_LAUNCH(32);
#undef LS_TAU_LAUNCH
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// ---- exact top-k of each query's queues --------------------------------------------------------------
// One 256-thread workgroup per query (one WAVE per query with the keys in registers and no
// barrier at all measured 19.7 us against 18.4 us: a lone wave has nothing to hide its LDS
// latencies behind). The query owns 4 queues per slice; thread t takes queues t, t + 256,
// ...: one load each for their lengths, a block-wide prefix for the slot ranges in LDS, then the
// (few) live entries. LDS: keys[keys_cap] | res[res_cap] | tmp[res_cap] (u64), hist[8*256] |
// misc[64] | 16 words (u32) — all dynamic (a static __shared__ would shift its alignment).
template <int LS_BSEL_QPT>  // queues per thread: 4 * nsplits / 256, rounded up to 1, 2, 4, 8
__global__ __launch_bounds__(256) void ls_batch_select_kernel(
    const uint2* __restrict__ queues, const u32* __restrict__ counts, int nsplits, int k,
    int keys_cap, int res_cap, long long base, long long n, long long rows_per_split,
    u32* __restrict__ overflow, float* __restrict__ out_scores,
    long long* __restrict__ out_indices) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sel[];
    u64* keys = reinterpret_cast<u64*>(smem_sel);
    u64* res = keys + keys_cap;
    u64* tmp = res + res_cap;
    u32* hist = reinterpret_cast<u32*>(tmp + res_cap);
    u32* misc = hist + 8 * 256;
    u32* wsum = misc + 64;
    u32& nkeys = wsum[4];
    constexpr int cap = LS_GEMM_QCAP;
    const int q = blockIdx.x, tid = threadIdx.x;
    const int nqueues = nsplits * 4;
    u32 c[LS_BSEL_QPT];
    // the first four entries of each queue are fetched together with its length (their address
    // does not depend on it): one round trip instead of two for the typical <= 4-entry queue
    uint4 ea[LS_BSEL_QPT], eb[LS_BSEL_QPT];
    const uint2* qbase = queues + (long long)q * nqueues * cap;  // queue qi starts at qbase + qi*cap
#pragma unroll
    for (int h = 0; h < LS_BSEL_QPT; ++h) {
        const int qi = tid + h * 256;  // == split * 4 + quarter
        c[h] = 0;
        ea[h] = eb[h] = make_uint4(0, 0, 0, 0);
        if (qi < nqueues) {
            c[h] = counts[(long long)q * nqueues + qi];
            ea[h] = reinterpret_cast<const uint4*>(qbase + (long long)qi * cap)[0];
            eb[h] = reinterpret_cast<const uint4*>(qbase + (long long)qi * cap)[1];
        }
    }
    {
        const int lane = tid & 63, wv = tid >> 6;
        u32 ct = 0;
#pragma unroll
        for (int h = 0; h < LS_BSEL_QPT; ++h) ct += c[h];
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
        for (int h = 0; h < LS_BSEL_QPT; ++h) {
            const u32 ch = c[h];
            const int qi = tid + h * 256;
            const long long rb = (long long)(qi >> 2) * rows_per_split;
            auto put = [&](u32 e, u32 bits, u32 lrow) {
                if (e < ch && start + e < (u32)keys_cap) {
                    const long long row = rb + (long long)lrow;
                    keys[start + e] = row < n ? ls_make_key(__uint_as_float(bits), (u32)row) : 0ull;
                }
            };
            uint4 a = ea[h], b = eb[h];
            for (u32 e0 = 0; e0 < ch; e0 += 4) {  // cap is a multiple of 4: loads stay in the queue
                if (e0) {
                    a = reinterpret_cast<const uint4*>(qbase + (long long)qi * cap + e0)[0];
                    b = reinterpret_cast<const uint4*>(qbase + (long long)qi * cap + e0)[1];
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
    if (cnt > keys_cap) {  // more candidates than fit: the exact scan path handles this query
        if (tid == 0) overflow[q] = 1u;
        return;
    }
    if (overflow[q]) return;  // a queue overflowed in the GEMM pass
    __syncthreads();
    const int nvalid = lds_topk(keys, cnt, k, res, tmp, hist, misc, tid, 256);
    __syncthreads();
    if (nvalid < k && tid == 0) overflow[q] = 2u;  // the speculative tau let < k rows through
    for (int i = tid; i < k; i += 256) {
        const u64 key = i < nvalid ? res[i] : 0ull;
        out_scores[(long long)q * k + i] = ls_key_score(key);
        out_indices[(long long)q * k + i] = ls_key_index(key, base);
    }
}

int ls_launch_batch_select(const ls_gemm_bufs& b, int nsplits, int64_t nq, int k, int keys_need,
                           int64_t base,
                           int64_t n, int64_t rows_per_split, float* d_out_scores,
                           int64_t* d_out_indices, hipStream_t s) {
    if (k > LS_GEMM_MAX_K || nsplits > LS_GEMM_MAX_SPLITS) {
        ls_set_error("batched path: k > %d or too many slices", LS_GEMM_MAX_K);
        return LS_ERR_INVALID_ARG;
    }
    // candidates per query: the caller's +5 sigma estimate for its sample (~2-5 k for a 1/16
    // sample, more for the thinner samples of long slices); at least 4 k and 2048
    int keys_cap = 2048;
    while (keys_cap < 4 * k || (keys_cap < keys_need && keys_cap < LS_BSEL_MAX_KEYS)) keys_cap <<= 1;
    int res_cap = 256;
    while (res_cap < k) res_cap <<= 1;
    const size_t smem = ((size_t)keys_cap + 2 * (size_t)res_cap) * sizeof(u64) +
                        (8 * 256 + 64 + 16) * sizeof(u32);
#define LS_BSEL_LAUNCH(P)                                                                      \
    {                                                                                          \
        static ls_attr_once once;                                                              \
        if (int rc = ls_set_max_dynamic_lds(once, (const void*)ls_batch_select_kernel<P>,     \
                                            128 * 1024))                                       \
            return rc;                                                                         \
        hipLaunchKernelGGL(ls_batch_select_kernel<P>, dim3((unsigned)nq), dim3(256), smem, s, \
                           (const uint2*)b.d_queues, b.d_counts, nsplits, k, keys_cap, res_cap, \
                           (long long)base, (long long)n, (long long)rows_per_split,          \
                           b.d_overflow, d_out_scores, (long long*)d_out_indices);            \
    }
    if (nsplits <= 64) LS_BSEL_LAUNCH(1)
    else if (nsplits <= 128) LS_BSEL_LAUNCH(2)
    else if (nsplits <= 256) LS_BSEL_LAUNCH(4)
    else LS_BSEL_LAUNCH(8)
#undef LS_BSEL_LAUNCH
    LS_HIP(hipGetLastError());
    return LS_OK;
}
