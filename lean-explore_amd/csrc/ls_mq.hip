// ls_mq.hip — small batches on an fp32 index: 2..32 queries share ONE pass over the corpus, with the
// inner products on the f32 matrix cores and BIT-IDENTICAL to the single-query scan kernel.
//
// Replaces faiss `index.search(x, k)` for the batch sizes between the reference's own call (nq = 1,
// reference src/lean_explore/search/engine.py:238-250 -> ls_scan.hip) and the big-batch MFMA paths: what
// a multi-client server produces when this library combines concurrent callers (engine.py:250 called
// from several MCP clients, mcp/server.py:147-151), small explicit batches, and the repairs of the
// batched paths. Until round 5 these ran the VALU scan in groups of 8 queries: 75 us (d = 384) /
// 180 us (d = 1024) per 8-query pass at N = 200 k against 47 / 124 us of HBM time, 16 queries two passes.
//
// Roofline: HBM for 2..16 queries (one MFMA B block, NB = 1): one pass reads the corpus once
// (n * d * 4 bytes); v_mfma_f32_16x16x4_f32 at 16 query columns needs n * d / 64 instructions of 32 cycles:
// 15.6 us (d = 384) / 41.7 us (d = 1024) of matrix time per SIMD-filled chip at N = 200 k, under the
// 47 / 124 us the HBM stream takes. 17..32 queries (NB = 2, round 6) double the matrix and VALU work per
// byte: ONE exact pass instead of two (68 us against 2 x 54 at d = 384, 168 against 2 x 138 at d = 1024),
// bound by instruction issue, not by HBM (section "Two B blocks" below).
//
// Same bits as ls_scan.hip. The scan kernel's fp32 order (ls_scan.hip QueryRegs::dot + group_sum;
// restated by oracle/flat_ip_ref.c ORDER_SCAN) is: lane `sub` of the L lanes sharing a row runs one
// fmaf chain over its chunks sub, sub+L, .., sub+(V-1)L (4 floats each, memory order), then a balanced
// xor tree over the L partial sums. tools/arith_probe.hip shows v_mfma_f32_16x16x4_f32 to be, bit for
// bit, acc = fmaf(a[k], b[k], acc) for k = 0, 1, 2, 3 in that order. So ONE MFMA whose K dimension is the
// four floats of chunk c advances chain (c mod L) by exactly the scan kernel's four fmafs - for 16 rows x
// 16 queries at once - and L accumulators per (row block, query block) hold the L chains; the tree is L-1
// vector adds. A query's scores, and therefore its results and their order, do not depend on whether it
// was served alone (scan) or in company (here): tests/test_mq_gpu.py, tests/test_concurrent_gpu.py and the
// zero-excuse parity checks (oracle.compare_kernel_order) assert array_equal.
//
// Work decomposition (one workgroup per CU: 4 waves, or 8 with two B blocks)
//   - a wave takes tiles of 16 consecutive rows, round-robin over all waves of the launch.
//   - MFMA operand layout: lane (i = lane % 16, kq = lane / 16) supplies A[row i][k = kq] and
//     B[k = kq][query i]. A lane loads ONE 16-byte chunk of its row per load instruction (chunk
//     cb + kq: the four lane groups cover four consecutive chunks = 64 contiguous bytes per row); a
//     4 x 4 transpose across the lane groups (2 x v_permlane32_swap + 2 x v_permlane16_swap on the four
//     registers) then leaves register m = element kq of chunk cb + m: four MFMA A operands for four
//     VALU instructions, no LDS round trip for the corpus. The loads stream through a static ring of P
//     "units" (1 KB per wave each), continuing into the wave's next tile. Round 6: P = 2 V units (6-8 KB
//     per wave in flight), HALF the scan kernel's depth - N = 200 k, 16 queries, d = 384, one box: P = 2 / 3 /
//     4 / 6 / 8 / 12 units: 68.6 / 59.9 / 55.5 / 53.1 / 54.6 / 57.2 us (d = 1024, P = 4 / 8 / 16: 150.7 / 140 /
//     142.8): 964 waves x 6 KB already cover HBM's latency, anything deeper only queues. (Requesting the wave's
//     whole first tile ahead of the query staging - 2 P units in flight during those 2.4-4 us - changed
//     nothing: 51.2 us either way.)
//   - chains are processed in groups of GC (16; 8 with two B blocks): group g = chains GC g..GC g+GC-1, their V
//     chunks each, then the group's add tree; the L / GC group sums meet on the upper levels of the same
//     tree as they complete (a binary counter of partial sums: log2 registers). The unit order inside a tile is
//     a compile-time permutation of the row's chunks; both halves of every 128-byte line are fetched by
//     consecutive units.
//   - B (the queries, normalised like the scan kernel's prologue: ls_wave_sumsq, one multiply per
//     element) lives in LDS query-major with a pitch of 2 (mod 32) floats: conflict-free staging writes
//     and fragment reads (mq_pitch).
//   - selection: the score block leaves lane (kq, query) with rows 4kq..4kq+3 of its query. Every lane
//     keeps the best M keys it has seen (branch-free compare-exchange chain; M = 3, 5 or 8, chosen by the
//     host from k / lanes so that a lane holding M of a query's top-k is a 1e-3 event). At the end the
//     four lane groups of a wave merge their lists in registers (top M of the union + the best key that
//     dropped out), the waves' lists meet in LDS (over the queries: they are dead by then), and the
//     workgroup emits its best k' of those keys (+ granule / bound exactly like the scan kernel:
//     finalize_body, the same-launch hand-off and the stand-alone finalize are shared unchanged).
//     bound = max(best key not emitted, best key dropped in a merge, every lane's M-th key): whatever the
//     workgroup saw and did not emit lies under it, so the proof in finalize_body holds; if it fails the
//     query is served again on the scan kernel, or - LS_FLAG_ASYNC-only calls - the rescue sweeps the score
//     vector S this kernel then writes like the scan does.
//
// Two B blocks (17..32 queries). Every A operand meets two 16-column B blocks: 8 MFMAs per 64-byte unit,
// 192 / 512 per tile (d = 384 / 1024) on 526 / 1265 other VALU instructions. A wave issues in order: eight
// back-to-back MFMAs hold it for 256 cycles, its lane swaps, adds and key inserts then run with the matrix
// pipe idle - one wave per SIMD took 82 / 197 us. Eight-wave workgroups (two waves per SIMD, <= 256 registers:
// 8 chains per group, no AGPR traffic; one shared copy of the queries, 131 KB at d = 1024) fill each other's
// gaps: 68 / 168 us, for 41 / 109 us of matrix time on the 224 CUs the launch uses (16 are left to the
// selection workgroups riding along). Measured and removed on the way (docs/EXPERIMENTS.md, round 6): a
// one-unit software pipeline (next unit's LDS reads and lane swaps under this unit's MFMAs: 75 vs 74 us, and
// 64 vs 55 us with one block), refills in bursts of 2 / 4 units (65 vs 59 us), 16 chains per group in 512
// registers (every tree add reads an AGPR back: 754 / 1745 VALU per tile).
#include "ls_select_dev.h"

#include <algorithm>

typedef float mq_f32x4 __attribute__((ext_vector_type(4)));

#define LS_MQ_WAVES 4        // waves per workgroup, one B block
#define LS_MQ_WAVES2 8       // ... two B blocks
#define LS_MQ_NQ 16          // query columns of one MFMA block
#define LS_MQ_LDS_MAX2 (136 * 1024)  // two-block kernel: 32 x (4 KB + 8) of queries (the key lists reuse them)
#ifndef LS_MQ_P1
#define LS_MQ_P1 0           // variant builds: ring depth in units, one B block (0: 2 V)
#endif
#ifndef LS_MQ_P2
#define LS_MQ_P2 0           // ... two B blocks (0: 2 V)
#endif

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

// floats between two queries in LDS: the stored row + 2. The LDS serves a wave's 64 lanes as two halves of 32 over
// 32 banks: the first half of a B fragment read is (query li = 0..15, k = 0..1) at li * pitch + k + 4c, so a pitch
// of 2 (mod 32) puts it on banks 2 li + k - all 32 - and the second half (k = 2, 3) likewise; a staging write
// (one query, 64 consecutive elements) is conflict-free under any pitch. PMC, 16 queries, d = 384:
// SQ_LDS_BANK_CONFLICT / SQ_INSTS_LDS = 0.019 (round 5's [chunk][k][query] layout: 0.24, all of it staging
// writes; a pitch of 4 (mod 64) - right for 64 banks - measured 3.8: half of all LDS cycles).
__host__ __device__ constexpr int mq_pitch(int chunks) { return chunks * 4 + 2; }
__host__ __device__ constexpr int mq_key_pitch(int tk) { return (tk + 14) / 16 * 16 + 1; }  // u64 between two queries' key lists

// NB = MFMA B blocks (16 query columns each) per A operand: 1 serves 2..16 queries, 2 serves 17..32.
// WPB = waves per workgroup (4, 8 with two blocks).
template <int L, int V, int M, int NB, int WPB>
__global__ __launch_bounds__(64 * WPB, WPB == 4 ? 2 : 1) void ls_mq_kernel(
    const mq_f32x4* __restrict__ corpus, long long n, const float* __restrict__ qraw, int d, int nq,
    int normalize, float* __restrict__ S, long long s_stride, u64* __restrict__ cand, long long c_stride,
    u64* __restrict__ bound, long long b_stride, int kprime, int nfin, ls_fin_batch fin,
    void* __restrict__ gran, long long g_stride, u32 tag, float* __restrict__ qkeep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    // the first `nfin` workgroups run selection jobs (of the previous launch, or - same-launch hand-off -
    // of this launch's own queries), exactly as in ls_scan_kernel
    if ((int)blockIdx.x < nfin) {
        for (int j = blockIdx.x; j < fin.njobs; j += nfin) {  // (nfin workgroups share the fin.njobs jobs)
            if (j != (int)blockIdx.x) __syncthreads();
            finalize_body<64 * WPB>(ls_fin_job(fin, j), smem_dyn, threadIdx.x);
        }
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
    constexpr int GC = NB == 1 ? 16 : 8;   // chains per accumulator group (32 for 4 KB rows - 512 contiguous bytes per row and
                                           // round instead of 256, 226 registers - is exact and slower: 137.3 vs 134.4 us)
    constexpr int NG = L / GC;             // accumulator groups
    constexpr int UPG = V * GC / 4;        // units per group
    constexpr int PREQ = NB == 1 ? LS_MQ_P1 : LS_MQ_P2;
    constexpr int P = (PREQ > 0 && NU % PREQ == 0) ? PREQ : 2 * V;  // units in flight per lane (1 KB per wave each)
    constexpr int DP = mq_pitch(CH);       // floats between two queries in LDS
    constexpr int NQT = NB * LS_MQ_NQ;     // query columns of the launch
    static_assert(L % GC == 0 && NU % P == 0 && NG * UPG == NU && (NG & (NG - 1)) == 0, "geometry");
    const int bid = (int)blockIdx.x - nfin;
    const int nblk = (int)gridDim.x - nfin;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 15, kq = lane >> 4;

    float* Bs = reinterpret_cast<float*>(smem_dyn);   // [NQT queries][DP]
    // [NQT queries][WPB waves][M] key lists, then the bounds. Two blocks: over the queries, once every wave is through
    // its tiles (131 KB of queries at d = 1024 leave no room beside them); one block: behind them (no barrier needed)
    u64* Ks = reinterpret_cast<u64*>(smem_dyn + (NB == 1 ? ((size_t)NQT * DP * 4 + 15) / 16 * 16 : 0));

    // Tiles of 16 rows are dealt round-robin to the waves of the launch (adjacent tiles go to different
    // workgroups): a run of adjacent, similar rows spreads over many workgroups.
    // Measured alternatives (tools/multiq_time.py, N = 200 k, 2 / 16 queries per pass, d = 384 | d = 1024):
    //   this form with 448 workgroups of 4 waves (1.75 per CU; 256 are ~10 % faster) 62.0 / 66.1 | 150.9 / 165.6 us
    //   256 workgroups of 8 waves, the tiles of a workgroup dealt to its waves
    //     by an LDS counter (every CU the same load, tools/mq_lifetimes.py)     63.1 / 69.1 | 149.4 / 172.3 us
    //   the same with (tile, 16-chain group) tasks of 12-16 KB, the group sums
    //     met in LDS by the last arriver (uniform work items for every d)       67.5 / 82.2 | 154.3 / 192.0 us
    //   (round 6) 8-wave workgroups with this static deal, one B block          62.2 | 149.7 us at 16 queries (55 | 139)
    // The eight-wave forms balance the CUs but pay a workgroup-wide barrier at the end (the slowest of 8
    // waves), a per-task LDS round trip, and the riding selection workgroups then displace whole scan
    // workgroups (one workgroup per CU leaves no second slot).
    const long long W = (long long)nblk * WPB;
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

    // ---- queries -> LDS, faiss.normalize_L2 fused (reference engine.py:242) exactly as in ls_scan_kernel:
    // canonical wave sum of squares (ls_wave_sumsq's order: lane l sums x[l], x[l+64], .. by fused
    // multiply-adds, then the xor tree 32..1), one correctly rounded 1/sqrt, one multiply per element. Wave w
    // stages queries w, w + WPB, ..: only the launch's REAL queries are loaded (unused columns are written as
    // zeros: what the LDS held before may be NaNs or denormals; a live query's row padding is zero too). All
    // loads of a wave are issued before the first is used (one memory round trip instead of one per query),
    // and in FRONT of the corpus loads - vector memory returns in order, and the queries (L2 hits for all but
    // the first workgroup) would otherwise arrive behind a cold HBM round trip.
    constexpr int EPL = CH * 4 / 64;       // elements per lane and query
    constexpr int QPW = NQT / WPB;         // queries per wave at most
    float xq[QPW][EPL];
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        const int qi = WPB * j + wave;
        if (qi < nq) {  // (wave-uniform)
            const float* src = qraw + (long long)qi * d;
#pragma unroll
            for (int i = 0; i < EPL; ++i) {
                const int e = lane + 64 * i;
                xq[j][i] = src[e < d ? e : d - 1];  // (unconditional loads; masked below)
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPL; ++i) xq[j][i] = 0.0f;
        }
    }
    // (a launch without score vectors keeps its raw queries for the repair: workgroup b copies query b)
    if (qkeep)
        for (int qq = bid; qq < nq; qq += nblk)
            for (int e = threadIdx.x; e < d; e += 64 * WPB) qkeep[(long long)qq * d + e] = qraw[(long long)qq * d + e];
    __builtin_amdgcn_sched_barrier(0);
    mq_f32x4 ring[P];
    {
        const mq_f32x4* p0 = tile_ptr(t);
#pragma unroll
        for (int u = 0; u < P; ++u) ring[u] = __builtin_nontemporal_load(p0 + unit_chunk(u));
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < QPW; ++j) {
        const int qi = WPB * j + wave;
        if (qi >= nq) {  // (wave-uniform) an unused column
#pragma unroll
            for (int i = 0; i < EPL; ++i) Bs[qi * DP + lane + 64 * i] = 0.0f;
            continue;
        }
#pragma unroll
        for (int i = 0; i < EPL; ++i)
            if (lane + 64 * i >= d) xq[j][i] = 0.0f;
        float inv = 1.0f;
        if (normalize) {
            float ss = 0.0f;
#pragma unroll
            for (int i = 0; i < EPL; ++i) ss = fmaf(xq[j][i], xq[j][i], ss);  // (zeros past d add nothing)
            ss = ls_wave_xor_sum(ss);
            if (ss > 0.0f) inv = 1.0f / sqrtf(ss);
        }
#pragma unroll
        for (int i = 0; i < EPL; ++i) Bs[qi * DP + lane + 64 * i] = xq[j][i] * inv;
    }
    __syncthreads();

    LS_MQSTAMP(1);
    // this lane's best rows (queries li, 16 + li; rows 4kq.. of the wave's tiles), best first - as (score, row) pairs
    // while the tiles stream (round 6): a lane meets its rows in increasing order, so "key greater" is "score
    // greater" (an equal score loses to the earlier row) and NaN / <= -FLT_MAX scores never pass `s > -FLT_MAX`:
    // one 32-bit compare and four selects per list step, no key built per row. The 64-bit keys are made once, below.
    float bs[NB][M];
    u32 br[NB][M];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int i = 0; i < M; ++i) {
            bs[b][i] = -FLT_MAX;
            br[b][i] = 0u;
        }

    while (t < NT) {
        const mq_f32x4* pcur = tile_ptr(t);
        const mq_f32x4* pnext = tile_ptr(t + W);
        constexpr int LV = NG > 1 ? 31 - __builtin_clz(NG) : 1;  // levels of the tree above the groups
        [[maybe_unused]] mq_f32x4 pend[NB][LV];   // partial sums of the groups seen so far, one per tree level (a binary counter)
        mq_f32x4 acc[NB][GC];
        mq_f32x4 sc[NB];
        // (the B fragments do not change from tile to tile: left alone, the compiler hoists all CH reads
        // out of this loop - 96 to 256 registers, spilled. An opaque copy of the lane offset per tile keeps
        // them where they are: one ds_read_b32 in front of its MFMA. The opaque value is the OFFSET, not the
        // pointer: an opaque pointer loses its LDS address space, the reads become flat_load_dword, and a
        // pending flat load forces every wait to vmcnt(0).)
        // (one base per B block: block 1 lies 16 x DP x 4 bytes up - past the 64 KB a ds_read offset field reaches
        // for 4 KB rows, and a base + constant the compiler forms per read costs a v_add each)
        const float* bf[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            int boff = (b * LS_MQ_NQ + li) * DP + kq;
            asm volatile("" : "+v"(boff));
            bf[b] = Bs + boff;
        }
        // (an explicit count: a bare `#pragma unroll` is a request the unroller declines past 16 K instructions - the
        // 64 units x 8 MFMAs of 4 KB rows with two B blocks - and a rolled loop indexes ring / acc dynamically)
        // (two waves share a SIMD's matrix pipe in the two-block form: the one streaming its tile's MFMAs goes first, the
        // other's list inserts fill the issue slots behind them - 32 queries, d = 384: 66.3 -> 64.5 us; one block: no change)
        if (NB == 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll NU
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
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const float bv = bf[b][4 * (cb + m)];
                    mq_f32x4 c;
                    if (v == 0) {
                        c[0] = 0.0f; c[1] = 0.0f; c[2] = 0.0f; c[3] = 0.0f;
                    } else {
                        c = acc[b][4 * j + m];
                    }
#ifdef LS_MQ_ABL_NOMFMA  // (ablation: everything but the matrix instruction - wrong results)
                    c[0] = fmaf(a[m], bv, c[0]);
                    acc[b][4 * j + m] = c;
#else
                    acc[b][4 * j + m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv, c, 0, 0, 0);
#endif
                }
            }
            // (the unit's B fragment reads go first: their LDS round trip then runs under the lane swaps. The
            // scheduler did that on its own for the [chunk][k][query] layout of round 5 and stopped doing it for
            // this one - every read sat right in front of its MFMA: 3.53 instead of 3.36 us per tile)
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * NB, 0);  // LDS reads
            __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);       // 2 moves + 4 lane swaps
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NB, 0);  // the MFMAs
            if (u % UPG == UPG - 1) {  // the group's chains are complete: xor tree 1, 2, 4, 8 ...
#pragma unroll
                for (int b = 0; b < NB; ++b) {
#pragma unroll
                    for (int o = 1; o < GC; o <<= 1)
#pragma unroll
                        for (int i = 0; i < GC; i += 2 * o) acc[b][i] = acc[b][i] + acc[b][i + o];
                    // ... and the levels above the groups (16, 32): group g's sum meets the partial sums of the same
                    // size as they complete - the same balanced tree as adding all NG group sums at the end, with
                    // log2(NG) live registers instead of NG
                    mq_f32x4 vsum = acc[b][0];
                    // (pinned here: the second block's sums are first USED behind the first block's score-vector
                    // branch at the end of the tile, and LLVM's sink pass moves a whole add tree down to its use -
                    // every chain sum of the tile then waits in registers, 149-214 of them spilled)
                    asm volatile("" : "+v"(vsum));
                    bool parked = false;
#pragma unroll
                    for (int lv = 0; lv < LV; ++lv) {
                        if (parked || NG == 1) continue;
                        if ((grp >> lv) & 1) {
                            vsum = pend[b][lv] + vsum;
                        } else {
                            pend[b][lv] = vsum;
                            parked = true;
                        }
                    }
                    if (grp == NG - 1) sc[b] = vsum;  // rows t*16 + 4kq + 0..3 of query 16 b + li
                }
            }
        }
        if (NB == 2) __builtin_amdgcn_s_setprio(0);
#ifdef LS_SCAN_TIMING
        if (tiles_done == 0) LS_MQSTAMP(2);
#endif

        const long long row0 = t * 16 + 4 * kq;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const bool live_q = LS_MQ_NQ * b + li < nq;
            // the score vector (what the selection's rescue sweeps). S == nullptr: the caller repairs a query whose
            // workgroup keys cannot be proven complete by serving it again on the single-query path (ls_api.hip).
            // What these stores cost is paid in the memory system, per write request, once the corpus no longer
            // fits the Infinity Cache: N = 200 k, d = 1024: 136 us without them for any query count, 140 / 144 /
            // 155 / 167 us with 2 / 4 / 8 / 16 queries (200 k half-line writes at 16: TCC_EA0_WRREQ_64B,
            // tools/mq_pmc.sh); d = 384 (307 MB): 54 -> 57.5 us. Tried, same times: a fifth wave that only stores
            // (fed through LDS: the scanning waves' in-order vmcnt never sees a store), quad-coalesced stores,
            // adjacent tiles paired into whole 128-byte lines per query (docs/EXPERIMENTS.md, round 5).
            if (live_q && S) {
                float* sp = S + (long long)(LS_MQ_NQ * b + li) * s_stride + row0;
                if (row0 + 3 < n) {
                    *reinterpret_cast<mq_f32x4*>(sp) = sc[b];  // s_stride is a multiple of 64 floats
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (row0 + r < n) sp[r] = sc[b][r];
                }
            }
            // (unused query columns keep lists of whatever their zero columns score: dropped when the keys are made)
            const bool ragged = t * 16 + 16 > n;  // (wave-uniform: only the launch's last tile holds rows >= n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float xs = (ragged && row0 + r >= n) ? -FLT_MAX : sc[b][r];
                u32 xr = (u32)(row0 + r);
#pragma unroll
                for (int i = 0; i < M; ++i) {  // branch-free insert: the better row stays, the other moves on
                    const bool gt = xs > bs[b][i];
                    const float hs = gt ? xs : bs[b][i];
                    const u32 hr = gt ? xr : br[b][i];
                    xs = gt ? bs[b][i] : xs;
                    xr = gt ? br[b][i] : xr;
                    bs[b][i] = hs;
                    br[b][i] = hr;
                }
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
    constexpr int TK = WPB * M;          // keys per query (k' + 1 <= TK)
    // (a query's keys / bounds start TKP / WPB + 1 u64 apart - 2 (mod 32) dwords: the 16 lanes that write one key slot of
    // 16 queries at once fall on 32 different banks; with the plain pitch of 32 u64 at M = 8 they all shared one pair)
    constexpr int TKP = mq_key_pitch(TK);
    u64* Kb = Ks + NQT * TKP;            // [NQT queries][WPB waves] bounds
    if (NB == 2) __syncthreads();        // (the key lists overwrite the queries: every wave is through its tiles)
    u64 lst[NB][M];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const bool live_q = LS_MQ_NQ * b + li < nq;
#pragma unroll
        for (int i = 0; i < M; ++i) lst[b][i] = live_q ? ls_make_key(bs[b][i], br[b][i]) : 0ull;  // (-FLT_MAX -> 0: no row)
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        u64 bnd = lst[b][M - 1];
#pragma unroll
        for (int mask = 16; mask <= 32; mask <<= 1) {
            u64 other[M];
#pragma unroll
            for (int i = 0; i < M; ++i) other[i] = xor_lanes64(lst[b][i], mask);
            const u64 obnd = xor_lanes64(bnd, mask);
            bnd = bnd > obnd ? bnd : obnd;
#pragma unroll
            for (int i = 0; i < M; ++i) {
                const u64 a = lst[b][i], o = other[M - 1 - i];
                const u64 lo = a < o ? a : o;
                lst[b][i] = a < o ? o : a;
                bnd = bnd > lo ? bnd : lo;
            }
#pragma unroll
            for (int pass = 0; pass < M; ++pass)  // odd-even transposition sort, descending (M <= 8)
#pragma unroll
                for (int i = pass & 1; i + 1 < M; i += 2) {
                    const u64 a = lst[b][i], o = lst[b][i + 1];
                    lst[b][i] = a > o ? a : o;
                    lst[b][i + 1] = a > o ? o : a;
                }
        }
        // Across the waves through LDS: per query 4 lists of M keys + 4 bounds
        if (kq == 0) {
            const int qc = LS_MQ_NQ * b + li;
#pragma unroll
            for (int i = 0; i < M; ++i) Ks[qc * TKP + wave * M + i] = lst[b][i];
            Kb[qc * (WPB + 1) + wave] = bnd;
        }
    }
    __syncthreads();
    LS_MQSTAMP(5);
    {
        // thread (query = tid / TPQ, slot = tid % TPQ) ranks keys slot, slot + TPQ, .. of its query among the
        // 4M by counting; the best k' go out, the bound is the best key that does not, or the best of the
        // waves' bounds.
        constexpr int TPQ = 64 * WPB / NQT;           // threads per query (16)
        constexpr int SPT = (TK + TPQ - 1) / TPQ;     // keys per thread (M = 3 / 5 / 8: 12 / 20 / 32 keys)
        const int qi = threadIdx.x / TPQ, slot = threadIdx.x % TPQ;
        const u64* kk = Ks + qi * TKP;
        u64 mine[SPT];
        int rank[SPT];
#pragma unroll
        for (int c = 0; c < SPT; ++c) {
            mine[c] = slot + TPQ * c < TK ? kk[slot + TPQ * c] : 0ull;
            rank[c] = 0;
        }
#pragma unroll
        for (int i = 0; i < TK; ++i) {
            const u64 o = kk[i];
#pragma unroll
            for (int c = 0; c < SPT; ++c)
                rank[c] += (o > mine[c]) || (o == mine[c] && i < slot + TPQ * c);  // ties exist only among the zeros
        }
        u64 lb = Kb[qi * (WPB + 1)];
#pragma unroll
        for (int w = 1; w < WPB; ++w) lb = Kb[qi * (WPB + 1) + w] > lb ? Kb[qi * (WPB + 1) + w] : lb;
#pragma unroll
        for (int c = 0; c < SPT; ++c) {
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
    if (threadIdx.x == 0 && nq <= 7 && S) {  // every workgroup's start / end tick: score vector 7 is unused
        unsigned long long* life = reinterpret_cast<unsigned long long*>(S + 7 * s_stride);
        life[2 * bid] = stamp[0];
        life[2 * bid + 1] = stamp[6];
    }
#endif
}

#ifndef LS_MQ_KERNEL_ONLY  // (scratch builds that instantiate one kernel and look at its ISA)
// ---- host side ------------------------------------------------------------------------------------
// workgroups of one launch: one per CU at most, at least two tiles per wave. A launch of 17..32 queries runs
// eight-wave workgroups that fill their CU (two waves per SIMD at up to 256 registers, up to 131 KB of LDS): it
// leaves LS_FIN_WG_MAX CUs to the selection workgroups that ride along (they would otherwise wait for a scan
// workgroup to end - and a same-launch selection for scan workgroups that cannot start before it ends).
static inline int mq_wpb(int nq) { return nq > LS_MQ_NQ ? LS_MQ_WAVES2 : LS_MQ_WAVES; }  // waves per workgroup
int ls_mq_blocks(int64_t n, int32_t n_cu, int nq, int chunks) {
    (void)chunks;
    const int wpb = mq_wpb(nq);
    const int64_t NT = (n + 15) / 16;
    constexpr int tpw = 2;  // at least this many tiles per wave on small shards
    const int64_t b = (NT + wpb * tpw - 1) / (wpb * tpw);
    const int64_t cap = nq > LS_MQ_NQ ? std::max(8, n_cu - LS_FIN_WG_MAX) : n_cu;
    // (big shards: ONE workgroup per CU. tools/mq_blocks_sweep.py, N = 200 k, 16 queries, d = 384 / 768 / 1024:
    //  256 workgroups 60.3 / 128.6 / 167.0 us, 448: 67.2 / 140.9 / 179.7, 512: 65.0 / 137.0 / 176.9,
    //  768: 65.9 / 140.3 / 181.0 - four waves per CU with 12-16 KB in flight each already carry the HBM
    //  stream; more streams only add DRAM page conflicts and a longer tail)
    if (b <= cap) return (int)std::max<int64_t>(b, 1);
    // ... and among the counts in [0.92, 1] x CUs the one whose last round of tiles is the fullest (12 500 tiles
    // over 256 x 4 waves are 12.2 rounds: a fifth of the waves then runs a 13th tile alone; 241 workgroups make
    // it 12.97). Same box, 16 queries: 256 -> 241-250 workgroups 61.6 -> 59.4 us (d = 384), 142.3 -> 133.9 (d = 768),
    // 166.8 -> 167.7-170.5 (d = 1024: within the noise).
    int64_t best = cap;
    double best_fill = -1.0;
    for (int64_t c = cap; c >= cap * 92 / 100; --c) {
        const double rounds = (double)NT / (double)(c * wpb);
        double fill = rounds - (double)(int64_t)rounds;
        if (fill == 0.0) fill = 1.0;
        if (fill > best_fill + 0.02) {  // near-ties go to the larger count
            best_fill = fill;
            best = c;
        }
    }
    return (int)best;
}

// keys a lane - and, after the in-register merge of its four lane groups, a WAVE - keeps per query: the smallest
// of {3, 5, 8} for which "some wave of the launch holds more than that many of one query's top-k" is rarer than
// 2e-3 per query (the wave-level list is the binding one: a wave sees n / waves rows of the query, a Poisson(k /
// waves) number of them in the top-k; whatever it drops raises the workgroup's bound past the k-th key and the
// query is served again - 140 us at d = 1024). Round 5 priced the per-lane lists only; with two B blocks at
// k = 1000 (1792 waves, 0.56 top-k rows per wave) that chose 5 keys and 4 % of the queries were served twice.
// 0 = this kernel is the wrong tool (k too large for the shard: the scan path's groups take the call).
int ls_mq_waves(int nq) { return mq_wpb(nq); }
int ls_mq_lane_keys(int blocks, int keff, int nq) {
    const double waves = (double)mq_wpb(nq) * blocks;
    const double mu = (double)keff / waves;
    for (int m : {3, 5, 8}) {
        double term = __builtin_exp(-mu), tail = 1.0 - term;  // P(X >= 1)
        for (int j = 1; j <= m; ++j) {
            term *= mu / j;
            tail -= term;  // ... P(X >= m + 1)
        }
        if (tail < 0.0) tail = 0.0;
        if (tail * waves < 2e-3) return m;
    }
    return 0;
}

// LDS of a scan workgroup: the queries, later overwritten by the waves' key lists + bounds
static size_t mq_lds_bytes(int chunks, int lane_keys, int nb, int wpb) {
    const size_t nqt = (size_t)nb * LS_MQ_NQ;
    const size_t qb = (nqt * mq_pitch(chunks) * sizeof(float) + 15) / 16 * 16;
    const size_t kb = nqt * (mq_key_pitch(wpb * lane_keys) + wpb + 1) * sizeof(u64);
    return nb == 1 ? qb + kb : std::max(qb, kb);
}

template <int L, int V, int M, int NB>
static int mq_launch(const void* corpus, int64_t n, const ls_geom& g, const ls_scan_args& a, hipStream_t s) {
    constexpr int WPB = NB == 1 ? LS_MQ_WAVES : LS_MQ_WAVES2;
    size_t smem = mq_lds_bytes(g.chunks, M, NB, WPB);
    if (a.nfin > 0) {
        const ls_fin_params& fp = a.fin.p0;
        const int keff = (int)((long long)fp.k < fp.n ? fp.k : fp.n);
        smem = std::max(smem, ls_fin_lds_bytes(fp.keys_cap, keff));
    }
    const int nfw = std::min(a.nfin, LS_FIN_WG_MAX);
    auto kern = ls_mq_kernel<L, V, M, NB, WPB>;
    static ls_attr_once once;
    // (one block: a riding selection job's LDS stays under LS_PIGGY_LDS_MAX and so do 16 queries of 4 KB rows;
    // two blocks take up to 132 KB - one workgroup per CU, which the block count allows for)
    if (smem > (size_t)(NB == 1 ? LS_PIGGY_LDS_MAX : LS_MQ_LDS_MAX2)) {
        ls_set_error("ls_launch_mq: %zu bytes of LDS for %d-chunk rows, %d query blocks", smem, g.chunks, NB);
        return LS_ERR_INVALID_ARG;
    }
    if (int rc = ls_set_max_dynamic_lds(once, (const void*)kern, NB == 1 ? LS_PIGGY_LDS_MAX : LS_MQ_LDS_MAX2)) return rc;
    hipLaunchKernelGGL(kern, dim3(a.blocks + nfw), dim3(64 * WPB), smem, s, (const mq_f32x4*)corpus,
                       (long long)n, a.d_q, g.d, a.nq, a.normalize ? 1 : 0, a.d_S, (long long)a.s_stride,
                       a.d_cand, (long long)a.c_stride, a.d_bound, (long long)a.b_stride, a.kprime, nfw,
                       a.fin, a.d_gran, (long long)a.g_stride, a.tag, a.d_qkeep);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// a.nq = the real query count (2..32); a.mq_keys = ls_mq_lane_keys(a.blocks, k, a.nq): 3, 5 or 8
int ls_launch_mq(const void* d_corpus, int64_t n, const ls_geom& g, const ls_scan_args& a, hipStream_t s) {
    if (n <= 0) return LS_OK;
    if (g.elem != 4 || a.nq < 1 || a.nq > 2 * LS_MQ_NQ || a.kprime < 1 || a.kprime + 1 > LS_MQ_KP_MAX ||
        (a.mq_keys != 3 && a.mq_keys != 5 && a.mq_keys != 8)) {
        ls_set_error("ls_launch_mq: bad arguments (elem %d nq %d kprime %d keys %d)", g.elem, a.nq, a.kprime, a.mq_keys);
        return LS_ERR_INVALID_ARG;
    }
#define LS_CASE_NB(LL, VV, NB)                                                       \
    return a.mq_keys == 3 ? mq_launch<LL, VV, 3, NB>(d_corpus, n, g, a, s)           \
         : a.mq_keys == 5 ? mq_launch<LL, VV, 5, NB>(d_corpus, n, g, a, s)           \
                          : mq_launch<LL, VV, 8, NB>(d_corpus, n, g, a, s);
#define LS_CASE(LL, VV)                                  \
    if (g.L == LL && g.V == VV) {                        \
        if (a.nq <= LS_MQ_NQ) { LS_CASE_NB(LL, VV, 1) }  \
        LS_CASE_NB(LL, VV, 2)                            \
    }
    LS_CASE(16, 1) LS_CASE(16, 2) LS_CASE(16, 3) LS_CASE(16, 4)
    LS_CASE(32, 3) LS_CASE(32, 4)
    LS_CASE(64, 3) LS_CASE(64, 4)
#undef LS_CASE
#undef LS_CASE_NB
    ls_set_error("ls_launch_mq: unsupported row geometry L=%d V=%d", g.L, g.V);
    return LS_ERR_INVALID_ARG;
}
#endif  // LS_MQ_KERNEL_ONLY
