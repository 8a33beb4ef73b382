// ls_mq.hip — small batches on an fp32 index: 2..16 queries share ONE pass over the corpus, with the
// inner products on the f32 matrix cores and BIT-IDENTICAL to the single-query scan kernel.
//
// Replaces faiss `index.search(x, k)` for the batch sizes between the reference's own call (nq = 1,
// reference src/lean_explore/search/engine.py:238-250 -> ls_scan.hip) and the big-batch MFMA paths: what
// a multi-client server produces when this library combines concurrent callers (engine.py:250 called
// from several MCP clients, mcp/server.py:147-151), small explicit batches, and the repairs of the
// batched paths. Until round 5 these ran the VALU scan in groups of 8 queries: 75 us (d = 384) /
// 180 us (d = 1024) per 8-query pass at N = 200 k against 47 / 124 us of HBM time, 16 queries two passes.
//
// Roofline: HBM. One pass reads the corpus once (n * d * 4 bytes) for up to 16 queries;
// v_mfma_f32_16x16x4_f32 at 16 query columns needs n * d / 64 instructions of 32 cycles:
// 15.6 us (d = 384) / 41.7 us (d = 1024) of matrix time per SIMD-filled chip at N = 200 k, under the
// 47 / 124 us the HBM stream takes - the matrix pipe has slack, the VALU work below hides in it.
//
// Same bits as ls_scan.hip. The scan kernel's fp32 order (ls_scan.hip QueryRegs::dot + group_sum;
// restated by oracle/flat_ip_ref.c ORDER_SCAN) is: lane `sub` of the L lanes sharing a row runs one
// fmaf chain over its chunks sub, sub+L, .., sub+(V-1)L (4 floats each, memory order), then a balanced
// xor tree over the L partial sums. tools/arith_probe.hip shows v_mfma_f32_16x16x4_f32 to be, bit for
// bit, acc = fmaf(a[k], b[k], acc) for k = 0, 1, 2, 3 in that order. So ONE MFMA whose K dimension is the
// four floats of chunk c advances chain (c mod L) by exactly the scan kernel's four fmafs - for 16 rows x
// 16 queries at once - and L accumulators per (row block, query block) hold the L chains; the tree is L-1
// vector adds. A query's scores, and therefore its results and their order, do not depend on whether it
// was served alone (scan) or in company (here): tests/test_concurrent_gpu.py and the zero-excuse parity
// checks (oracle.compare_kernel_order) assert array_equal.
//
// Work decomposition (one workgroup = 4 waves, one workgroup per CU)
//   - a wave takes tiles of 16 consecutive rows, round-robin over all waves of the launch.
//   - MFMA operand layout: lane (i = lane % 16, kq = lane / 16) supplies A[row i][k = kq] and
//     B[k = kq][query i]. A lane loads ONE 16-byte chunk of its row per load instruction (chunk
//     cb + kq: the four lane groups cover four consecutive chunks = 64 contiguous bytes per row); a
//     4 x 4 transpose across the lane groups (2 x v_permlane32_swap + 2 x v_permlane16_swap on the four
//     registers) then leaves register m = element kq of chunk cb + m: four MFMA A operands for four
//     VALU instructions, no LDS round trip for the corpus. The loads stream through a static ring of P
//     "units" (12-16 in flight per lane, the scan kernel's depth), continuing into the wave's next tile.
//   - chains are processed in groups of 16 (64 accumulator registers): group g = chains 16g..16g+15,
//     their V chunks each, then the group's 15-add tree; the L/16 group sums are combined by the top
//     levels of the same tree. The unit order inside a tile is a compile-time permutation of the row's
//     chunks; both halves of every 128-byte line are fetched by consecutive units.
//   - B (the queries, normalised like the scan kernel's prologue: ls_wave_sumsq, one multiply per
//     element) lives in LDS as [chunk][kq][query]: one conflict-free ds_read_b32 per MFMA.
//   - selection: the score block leaves lane (kq, query) with rows 4kq..4kq+3 of its query. Every lane
//     keeps the best M keys it has seen (branch-free compare-exchange chain; M = 3, 5 or 8, chosen by the
//     host from k / lanes so that a lane holding M of a query's top-k is a 1e-3 event). At the end the
//     four lane groups of a wave merge their lists in registers (top M of the union + the best key that
//     dropped out), the four waves' lists meet in LDS, and the workgroup emits its best k' of those 4 M
//     keys (+ granule / bound exactly like the scan kernel: finalize_body, the same-launch hand-off and
//     the stand-alone finalize are shared unchanged). bound = max(best key not emitted, best key dropped
//     in a merge, every lane's M-th key): whatever the workgroup saw and did not emit lies under it, so
//     the proof in finalize_body holds; if it fails the rescue sweeps the score vector S, which this
//     kernel writes like the scan does.
#include "ls_select_dev.h"

#include <algorithm>

typedef float mq_f32x4 __attribute__((ext_vector_type(4)));

#define LS_MQ_THREADS 256
#define LS_MQ_WAVES 4
#define LS_MQ_NQ 16          // query columns of one MFMA block
#define LS_MQ_GC 16          // chains per accumulator group


// 4 x 4 transpose across the four 16-lane groups: in: lane group g, register m = T[m][g];
// out: register m of lane group g = T[g][m]
__device__ __forceinline__ void mq_transpose(const mq_f32x4& x, float (&r)[4]) {
    u32 r0 = __builtin_bit_cast(u32, (float)x[0]), r1 = __builtin_bit_cast(u32, (float)x[1]);
    u32 r2 = __builtin_bit_cast(u32, (float)x[2]), r3 = __builtin_bit_cast(u32, (float)x[3]);
    const auto a = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);  // rows 2,3 of r0 <-> rows 0,1 of r2
    const auto b = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
    const auto c = __builtin_amdgcn_permlane16_swap((u32)a[0], (u32)b[0], false, false);  // odd rows <-> even rows
    const auto e = __builtin_amdgcn_permlane16_swap((u32)a[1], (u32)b[1], false, false);
    r[0] = __builtin_bit_cast(float, (u32)c[0]);
    r[1] = __builtin_bit_cast(float, (u32)c[1]);
    r[2] = __builtin_bit_cast(float, (u32)e[0]);
    r[3] = __builtin_bit_cast(float, (u32)e[1]);
}

template <int L, int V, int M>
__global__ __launch_bounds__(LS_MQ_THREADS, 2) void ls_mq_kernel(
    const mq_f32x4* __restrict__ corpus, long long n, const float* __restrict__ qraw, int d, int nq,
    int normalize, float* __restrict__ S, long long s_stride, u64* __restrict__ cand, long long c_stride,
    u64* __restrict__ bound, long long b_stride, int kprime, int nfin, ls_fin_batch fin,
    void* __restrict__ gran, long long g_stride, u32 tag, float* __restrict__ qkeep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    // the first `nfin` workgroups run selection jobs (of the previous launch, or - same-launch hand-off -
    // of this launch's own queries), exactly as in ls_scan_kernel
    if ((int)blockIdx.x < nfin) {
        finalize_body<LS_MQ_THREADS>(fin.p[blockIdx.x], smem_dyn, threadIdx.x);
        return;
    }
#ifdef LS_SCAN_TIMING  // developer instrumentation: phase stamps (100 MHz ticks) of one workgroup
    unsigned long long stamp[8] = {};
    int tiles_done = 0;
#define LS_MQSTAMP(i) stamp[i] = wall_clock64()
#else
#define LS_MQSTAMP(i) do {} while (0)
#endif
    LS_MQSTAMP(0);
    constexpr int CH = L * V;              // 16-byte chunks per stored row
    constexpr int NU = CH / 4;             // load units per tile (4 chunks = 64 bytes per row each)
    constexpr int GC = LS_MQ_GC;
    constexpr int NG = L / GC;             // accumulator groups
    constexpr int UPG = V * GC / 4;        // units per group
#ifndef LS_MQ_RING_MULT
#define LS_MQ_RING_MULT 1
#endif
    constexpr int P = (LS_MQ_RING_MULT > 1 && NU % (LS_MQ_RING_MULT * UPG) == 0) ? LS_MQ_RING_MULT * UPG : UPG;  // units in flight per lane
    static_assert(L % GC == 0 && NU % P == 0 && NG * UPG == NU, "geometry");
    const int bid = (int)blockIdx.x - nfin;
    const int nblk = (int)gridDim.x - nfin;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kq = lane >> 4;

    float* Bs = reinterpret_cast<float*>(smem_dyn);                    // [CH][4][16]
    u64* Ks = reinterpret_cast<u64*>(smem_dyn + (size_t)CH * 256);     // [16 queries][4 waves][M], then the bounds

    // Tiles of 16 rows are dealt round-robin to the waves of the launch (adjacent tiles go to different
    // workgroups): a run of adjacent, similar rows spreads over many workgroups.
    // Measured alternatives (tools/multiq_time.py, N = 200 k, 2 / 16 queries per pass, d = 384 | d = 1024):
    //   this form with 448 workgroups of 4 waves (1.75 per CU; 256 are ~10 % faster) 62.0 / 66.1 | 150.9 / 165.6 us
    //   256 workgroups of 8 waves, the tiles of a workgroup dealt to its waves
    //     by an LDS counter (every CU the same load, tools/mq_lifetimes.py)     63.1 / 69.1 | 149.4 / 172.3 us
    //   the same with (tile, 16-chain group) tasks of 12-16 KB, the group sums
    //     met in LDS by the last arriver (uniform work items for every d)       67.5 / 82.2 | 154.3 / 192.0 us
    // The eight-wave forms balance the CUs but pay a workgroup-wide barrier at the end (the slowest of 8
    // waves), a per-task LDS round trip, and the riding selection workgroups then displace whole scan
    // workgroups (one workgroup per CU leaves no second slot).
    const long long W = (long long)nblk * LS_MQ_WAVES;
    const long long NT = (n + 15) / 16;
    // wave-major numbering: the launch's last, partial round of tiles (12 500 tiles over 1024 waves: 0.2 of
    // a round) goes to ONE wave in each of many workgroups instead of all four waves of a few - the tail is a
    // lone wave's tile (~2 us) in 212 CUs, not four waves' worth (~4 us) in 53
    long long t = (long long)wave * nblk + bid;

    // unit u of a tile -> first chunk: group-major, then the lane's V rounds, then 4-chunk steps
    auto unit_chunk = [](int u) constexpr -> int {
        const int grp = u / UPG, v = (u % UPG) / (GC / 4), j = u % (GC / 4);
        return L * v + GC * grp + 4 * j;
    };
    auto tile_ptr = [&](long long tile) -> const mq_f32x4* {
        // (LS_CORPUS_PAD_ROWS zero rows follow row n-1: the ragged last tile needs no clamping)
        const long long tc = tile < NT ? tile : NT - 1;  // a prefetch past the wave's last tile re-reads it
        return corpus + (tc * 16 + li) * CH + kq;
    };

    // ---- queries -> LDS in MFMA B layout, faiss.normalize_L2 fused (reference engine.py:242) exactly as
    // in ls_scan_kernel: canonical wave sum of squares (ls_wave_sumsq's order: lane l sums x[l], x[l+64], ..
    // by fused multiply-adds, then the xor tree 32..1), one correctly rounded 1/sqrt, one multiply per
    // element. Unused query columns (>= nq) and the row padding are zero. A wave stages queries
    // 4 wave .. 4 wave + 3: all their loads are issued before the first is used (one memory round trip
    // instead of one per query: 10.7 -> ~3 us at nq = 16, tools/mq_phases.py), and in FRONT of the corpus
    // ring's first loads - vector memory returns in order, and the queries (L2 hits for all but the first
    // workgroup) would otherwise arrive behind a cold HBM round trip. The ring's first P units then fly
    // while the queries are normalised and written.
    constexpr int EPL = CH * 4 / 64;             // elements per lane and query
    constexpr int QPW = LS_MQ_NQ / LS_MQ_WAVES;  // queries per wave
    static_assert(QPW == 4, "one 16-byte LDS write per element");
    float xq[QPW][EPL];
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        const int qi = QPW * wave + j;
        const float* src = qraw + (long long)(qi < nq ? qi : 0) * d;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int e = lane + 64 * i;
            xq[j][i] = src[e < d ? e : d - 1];  // (unconditional loads; masked below)
        }
    }
    // (a launch without score vectors keeps its raw queries for the repair: workgroup b copies query b)
    if (qkeep)
        for (int qq = bid; qq < nq; qq += nblk)
            for (int e = threadIdx.x; e < d; e += LS_MQ_THREADS) qkeep[(long long)qq * d + e] = qraw[(long long)qq * d + e];
    __builtin_amdgcn_sched_barrier(0);
    mq_f32x4 ring[P];
    {
        const mq_f32x4* p0 = tile_ptr(t);
#pragma unroll
        for (int u = 0; u < P; ++u) ring[u] = __builtin_nontemporal_load(p0 + unit_chunk(u));
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float inv[QPW];
#pragma unroll
        for (int j = 0; j < QPW; ++j) {
#pragma unroll
            for (int i = 0; i < EPL; ++i)
                if (QPW * wave + j >= nq || lane + 64 * i >= d) xq[j][i] = 0.0f;
            inv[j] = 1.0f;
            if (normalize) {
                float ss = 0.0f;
#pragma unroll
                for (int i = 0; i < EPL; ++i) ss = fmaf(xq[j][i], xq[j][i], ss);  // (zeros past d add nothing)
                ss = ls_wave_xor_sum(ss);
                if (ss > 0.0f) inv[j] = 1.0f / sqrtf(ss);
            }
        }
        // element e of the wave's four consecutive queries: ONE 16-byte LDS write (the four-byte form put the
        // 64 lanes of a write on 4 banks)
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            const int e = lane + 64 * i;
            mq_f32x4 w4;
#pragma unroll
            for (int j = 0; j < QPW; ++j) w4[j] = xq[j][i] * inv[j];
            *reinterpret_cast<mq_f32x4*>(&Bs[(e >> 2) * 64 + (e & 3) * 16 + QPW * wave]) = w4;
        }
    }
    __syncthreads();

    LS_MQSTAMP(1);
    u64 lst[M];  // this lane's best keys (query li, rows 4kq.. of the wave's tiles), descending
#pragma unroll
    for (int i = 0; i < M; ++i) lst[i] = 0ull;
    const bool live_q = li < nq;

    while (t < NT) {
        const mq_f32x4* pcur = tile_ptr(t);
        const mq_f32x4* pnext = tile_ptr(t + W);
        mq_f32x4 gs[NG];
        mq_f32x4 acc[GC];
        // (the B fragments do not change from tile to tile: left alone, the compiler hoists all CH reads
        // out of this loop - 96 to 256 registers, spilled. An opaque copy of the lane offset per tile keeps
        // them where they are: one ds_read_b32 in front of its MFMA. The opaque value is the OFFSET, not the
        // pointer: an opaque pointer loses its LDS address space, the reads become flat_load_dword, and a
        // pending flat load forces every wait to vmcnt(0).)
        int boff = lane;
        asm volatile("" : "+v"(boff));
        const float* bf = Bs + boff;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const mq_f32x4 x = ring[u % P];
            // refill the slot: a later unit of this tile, or the head of the wave's next tile
            if (u + P < NU)
                ring[u % P] = __builtin_nontemporal_load(pcur + unit_chunk(u + P));
            else
                ring[u % P] = __builtin_nontemporal_load(pnext + unit_chunk(u + P - NU));
            // The tile body is one basic block; left alone, the scheduler sinks every refill down to its
            // first use to shorten live ranges (it chases a higher occupancy), the waits become vmcnt(0)
            // and each unit pays a full memory round trip (measured: 93 us per 16-query pass at N = 200 k
            // instead of ~65). Nothing moves across this point: the refill stays P units ahead of its use.
            __builtin_amdgcn_sched_barrier(0);
            float a[4];
            mq_transpose(x, a);
            const int cb = unit_chunk(u);
            const int v = (u % UPG) / (GC / 4), j = u % (GC / 4), grp = u / UPG;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float bv = bf[(cb + m) * 64];
                mq_f32x4 c;
                if (v == 0) {
                    c[0] = 0.0f; c[1] = 0.0f; c[2] = 0.0f; c[3] = 0.0f;
                } else {
                    c = acc[4 * j + m];
                }
                acc[4 * j + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv, c, 0, 0, 0);
            }
            if (u % UPG == UPG - 1) {  // the group's chains are complete: xor tree 1, 2, 4, 8
#pragma unroll
                for (int o = 1; o < GC; o <<= 1)
#pragma unroll
                    for (int i = 0; i < GC; i += 2 * o) acc[i] = acc[i] + acc[i + o];
                gs[grp] = acc[0];
            }
        }
#pragma unroll
        for (int o = 1; o < NG; o <<= 1)  // the tree's top levels (16, 32)
#pragma unroll
            for (int i = 0; i < NG; i += 2 * o) gs[i] = gs[i] + gs[i + o];
        const mq_f32x4 sc = gs[0];  // rows t*16 + 4kq + 0..3 of query li
#ifdef LS_SCAN_TIMING
        if (tiles_done == 0) LS_MQSTAMP(2);
#endif

        const long long row0 = t * 16 + 4 * kq;
        // the score vector (what the selection's rescue sweeps). S == nullptr: the caller repairs a query whose
        // workgroup keys cannot be proven complete by serving it again on the single-query path (ls_api.hip).
        // What these stores cost is paid in the memory system, per write request, once the corpus no longer
        // fits the Infinity Cache: N = 200 k, d = 1024: 136 us without them for any query count, 140 / 144 /
        // 155 / 167 us with 2 / 4 / 8 / 16 queries (200 k half-line writes at 16: TCC_EA0_WRREQ_64B,
        // tools/mq_pmc.sh); d = 384 (307 MB): 54 -> 57.5 us. Tried, same times: a fifth wave that only stores
        // (fed through LDS: the scanning waves' in-order vmcnt never sees a store), quad-coalesced stores,
        // adjacent tiles paired into whole 128-byte lines per query (docs/EXPERIMENTS.md, round 5).
        if (live_q && S) {
            float* sp = S + (long long)li * s_stride + row0;
            if (row0 + 3 < n) {
                *reinterpret_cast<mq_f32x4*>(sp) = sc;  // s_stride is a multiple of 64 floats
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (row0 + r < n) sp[r] = sc[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = sc[r];
            u64 x = (live_q && row0 + r < n) ? ls_make_key(s, (u32)(row0 + r)) : 0ull;
#pragma unroll
            for (int i = 0; i < M; ++i) {  // branch-free insert: the larger key stays, the smaller moves on
                const bool gt = x > lst[i];
                const u64 hi = gt ? x : lst[i];
                x = gt ? lst[i] : x;
                lst[i] = hi;
            }
        }
        t += W;
#ifdef LS_SCAN_TIMING
        if (tiles_done++ == 0) LS_MQSTAMP(3);
#endif
    }
    LS_MQSTAMP(4);

    // ---- 16 lanes hold keys of one query: 4 lane groups x 4 waves ------------------------------------
    // In the wave first (registers, every query of the wave at once): the lane groups kq and kq ^ 1,
    // then ^ 2 merge their sorted lists - C[i] = max(A[i], B[M-1-i]) is the top M of the union (a bitonic
    // sequence, re-sorted by a small network), min(A[i], B[M-1-i]) are the keys that leave - and carry a
    // bound: the best key dropped anywhere below (a lane's own drops lie under its last key).
    u64 bnd = lst[M - 1];
    // (lane ^ 16 / lane ^ 32 by v_permlane16_swap / v_permlane32_swap: no LDS crossbar round trips)
    auto xor_lanes32 = [&](u32 v, int mask) -> u32 {
        if (mask == 32) {
            const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
            return lane < 32 ? (u32)r[1] : (u32)r[0];
        }
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane & 16) ? (u32)r[0] : (u32)r[1];
    };
    auto xor_lanes64 = [&](u64 v, int mask) -> u64 {
        return ((u64)xor_lanes32((u32)(v >> 32), mask) << 32) | xor_lanes32((u32)v, mask);
    };
#pragma unroll
    for (int mask = 16; mask <= 32; mask <<= 1) {
        u64 other[M];
#pragma unroll
        for (int i = 0; i < M; ++i) other[i] = xor_lanes64(lst[i], mask);
        const u64 obnd = xor_lanes64(bnd, mask);
        bnd = bnd > obnd ? bnd : obnd;
#pragma unroll
        for (int i = 0; i < M; ++i) {
            const u64 a = lst[i], b = other[M - 1 - i];
            const u64 lo = a < b ? a : b;
            lst[i] = a < b ? b : a;
            bnd = bnd > lo ? bnd : lo;
        }
#pragma unroll
        for (int pass = 0; pass < M; ++pass)  // odd-even transposition sort, descending (M <= 8)
#pragma unroll
            for (int i = pass & 1; i + 1 < M; i += 2) {
                const u64 a = lst[i], b = lst[i + 1];
                lst[i] = a > b ? a : b;
                lst[i + 1] = a > b ? b : a;
            }
    }
    // Across the waves through LDS: per query 4 lists of M keys + 4 bounds; thread (query = tid / 16,
    // slot = tid % 16) ranks key `slot` of its query among the 4M by counting; the best k' go out, the
    // bound is the best key that does not, or the best of the waves' bounds.
    constexpr int TK = LS_MQ_WAVES * M;  // keys per query (k' + 1 <= TK)
    constexpr int TPQ = LS_MQ_THREADS / LS_MQ_NQ;  // threads per query
    u64* Kb = Ks + LS_MQ_NQ * TK;  // [16 queries][4 waves] bounds
    if (kq == 0) {
#pragma unroll
        for (int i = 0; i < M; ++i) Ks[li * TK + wave * M + i] = lst[i];
        Kb[li * LS_MQ_WAVES + wave] = bnd;
    }
    __syncthreads();
    LS_MQSTAMP(5);
    {
        const int qi = threadIdx.x / TPQ, slot = threadIdx.x % TPQ;
        const u64* kk = Ks + qi * TK;
        u64 mine[2];
        int rank[2] = {0, 0};
        mine[0] = slot < TK ? kk[slot] : 0ull;
        mine[1] = slot + TPQ < TK ? kk[slot + TPQ] : 0ull;  // (M = 5 / 8: 20 / 32 keys over 16 threads)
#pragma unroll
        for (int i = 0; i < TK; ++i) {
            const u64 o = kk[i];
            rank[0] += (o > mine[0]) || (o == mine[0] && i < slot);  // ties exist only among the zeros
            if (TK > TPQ) rank[1] += (o > mine[1]) || (o == mine[1] && i < slot + TPQ);
        }
        u64 lb = Kb[qi * LS_MQ_WAVES];
#pragma unroll
        for (int w = 1; w < LS_MQ_WAVES; ++w) lb = Kb[qi * LS_MQ_WAVES + w] > lb ? Kb[qi * LS_MQ_WAVES + w] : lb;
#pragma unroll
        for (int c = 0; c < (TK > TPQ ? 2 : 1); ++c) {
            if (qi >= nq || slot + TPQ * c >= TK) continue;
            if (gran) {  // same-launch selection: tagged 16-byte granules, rank-major (ls_scan.hip)
                if (rank[c] <= kprime) {
                    const u64 out = rank[c] == kprime ? (mine[c] > lb ? mine[c] : lb) : mine[c];
                    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                        (char*)gran + (long long)qi * g_stride * 16, 0, nblk * (kprime + 1) * 16, LS_BUF_RSRC_FLAGS);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{(u32)out, (u32)(out >> 32), tag, 0u}, rsrc,
                                                           (rank[c] * nblk + bid) * 16, 0, LS_AUX_SC1);
                }
            } else {
                if (rank[c] < kprime) cand[qi * c_stride + (long long)bid * kprime + rank[c]] = mine[c];
                if (rank[c] == kprime) bound[qi * b_stride + bid] = mine[c] > lb ? mine[c] : lb;
            }
        }
    }
#ifdef LS_SCAN_TIMING
    LS_MQSTAMP(6);
    if (bid == nblk / 2 && threadIdx.x == 0) {
        for (int i = 0; i < 6; ++i) cand[c_stride - 8 + i] = stamp[i + 1] - stamp[i];
        cand[c_stride - 2] = (u64)tiles_done;
    }
    if (threadIdx.x == 0 && nq <= 7) {  // every workgroup's start / end tick: score vector 7 is unused
        unsigned long long* life = reinterpret_cast<unsigned long long*>(S + 7 * s_stride);
        life[2 * bid] = stamp[0];
        life[2 * bid + 1] = stamp[6];
    }
#endif
}

// ---- host side ------------------------------------------------------------------------------------
// workgroups of one launch: one per CU at most, at least two tiles per wave
int ls_mq_blocks(int64_t n, int32_t n_cu) {
    const int64_t NT = (n + 15) / 16;
    constexpr int tpw = 2;  // at least this many tiles per wave on small shards
    const int64_t b = (NT + LS_MQ_WAVES * tpw - 1) / (LS_MQ_WAVES * tpw);
    // (big shards: ONE workgroup per CU. tools/mq_blocks_sweep.py, N = 200 k, 16 queries, d = 384 / 768 / 1024:
    //  256 workgroups 60.3 / 128.6 / 167.0 us, 448: 67.2 / 140.9 / 179.7, 512: 65.0 / 137.0 / 176.9,
    //  768: 65.9 / 140.3 / 181.0 - four waves per CU with 12-16 KB in flight each already carry the HBM
    //  stream; more streams only add DRAM page conflicts and a longer tail)
    if (b <= n_cu) return (int)std::max<int64_t>(b, 1);
    // ... and among the counts in [0.92, 1] x CUs the one whose last round of tiles is the fullest (12 500 tiles
    // over 256 x 4 waves are 12.2 rounds: a fifth of the waves then runs a 13th tile alone; 241 workgroups make
    // it 12.97). Same box, 16 queries: 256 -> 241-250 workgroups 61.6 -> 59.4 us (d = 384), 142.3 -> 133.9 (d = 768),
    // 166.8 -> 167.7-170.5 (d = 1024: within the noise).
    int64_t best = n_cu;
    double best_fill = -1.0;
    for (int64_t c = n_cu; c >= (int64_t)n_cu * 92 / 100; --c) {
        const double rounds = (double)NT / (double)(c * LS_MQ_WAVES);
        double fill = rounds - (double)(int64_t)rounds;
        if (fill == 0.0) fill = 1.0;
        if (fill > best_fill + 0.02) {  // near-ties go to the larger count
            best_fill = fill;
            best = c;
        }
    }
    return (int)best;
}

// keys a lane keeps: the smallest of {3, 5, 8} for which "some lane of the launch holds that many of one
// query's top-k" is rarer than 2e-3 per query (Poisson tail, lambda = k / lanes per query); 0 = this
// kernel is the wrong tool (k too large for the shard: the scan path's groups take the call)
int ls_mq_lane_keys(int blocks, int keff) {
    const double lanes = 4.0 * LS_MQ_WAVES * blocks;
    const double lam = (double)keff / lanes;
    const double p3 = lam * lam * lam / 6.0 * lanes;
    if (p3 < 2e-3) return 3;
    const double p5 = lam * lam * lam * lam * lam / 120.0 * lanes;
    if (p5 < 4e-3) return 5;
    const double l4 = lam * lam * lam * lam;
    if (l4 * l4 / 40320.0 * lanes < 4e-3) return 8;  // (k = 1000 over 200 k rows: lambda = 0.24)
    return 0;
}

size_t ls_mq_lds_bytes(const ls_geom& g, int lane_keys) {
    return (size_t)g.chunks * 256 + (size_t)LS_MQ_NQ * LS_MQ_WAVES * (lane_keys + 1) * sizeof(u64);
}

template <int L, int V, int M>
static int mq_launch(const void* corpus, int64_t n, const ls_geom& g, const ls_scan_args& a, hipStream_t s) {
    size_t smem = ls_mq_lds_bytes(g, M);
    for (int i = 0; i < a.nfin; ++i) {
        const ls_fin_params& fp = a.fin.p[i];
        const int keff = (int)((long long)fp.k < fp.n ? fp.k : fp.n);
        smem = std::max(smem, ls_fin_lds_bytes(fp.keys_cap, keff));
    }
    auto kern = ls_mq_kernel<L, V, M>;
    static ls_attr_once once;
    if (int rc = ls_set_max_dynamic_lds(once, (const void*)kern, LS_PIGGY_LDS_MAX)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.blocks + a.nfin), dim3(LS_MQ_THREADS), smem, s, (const mq_f32x4*)corpus,
                       (long long)n, a.d_q, g.d, a.nq, a.normalize ? 1 : 0, a.d_S, (long long)a.s_stride,
                       a.d_cand, (long long)a.c_stride, a.d_bound, (long long)a.b_stride, a.kprime, a.nfin,
                       a.fin, a.d_gran, (long long)a.g_stride, a.tag, a.d_qkeep);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// a.nq = the real query count (2..16); a.mq_keys = ls_mq_lane_keys(a.blocks, k): 3, 5 or 8
int ls_launch_mq(const void* d_corpus, int64_t n, const ls_geom& g, const ls_scan_args& a, hipStream_t s) {
    if (n <= 0) return LS_OK;
    if (g.elem != 4 || a.nq < 1 || a.nq > LS_MQ_NQ || a.kprime < 1 || a.kprime + 1 > LS_KP_MAX ||
        (a.mq_keys != 3 && a.mq_keys != 5 && a.mq_keys != 8)) {
        ls_set_error("ls_launch_mq: bad arguments (elem %d nq %d kprime %d keys %d)", g.elem, a.nq, a.kprime, a.mq_keys);
        return LS_ERR_INVALID_ARG;
    }
#define LS_CASE(LL, VV)                                                          \
    if (g.L == LL && g.V == VV)                                                  \
        return a.mq_keys == 3 ? mq_launch<LL, VV, 3>(d_corpus, n, g, a, s)       \
             : a.mq_keys == 5 ? mq_launch<LL, VV, 5>(d_corpus, n, g, a, s)       \
                              : mq_launch<LL, VV, 8>(d_corpus, n, g, a, s);
    LS_CASE(16, 1) LS_CASE(16, 2) LS_CASE(16, 3) LS_CASE(16, 4)
    LS_CASE(32, 3) LS_CASE(32, 4)
    LS_CASE(64, 3) LS_CASE(64, 4)
#undef LS_CASE
    ls_set_error("ls_launch_mq: unsupported row geometry L=%d V=%d", g.L, g.V);
    return LS_ERR_INVALID_ARG;
}
