// ls_scan.hip — the HBM-bound scan: one query against every corpus row (batch-1 path).
//
// Replaces the hot loop inside faiss `index.search(x, k)` for nq = 1, the only shape the
// reference ever issues (reference src/lean_explore/search/engine.py:250, nq = 1 at :238).
//
// Roofline: 2 flops per corpus element read, 0.5 FLOP/B for fp32 -> HBM bound. Algorithmic
// bytes per query = n * d * elem (corpus) + d*4 (query) + n*4 (score vector S) + candidates.
//
// Work decomposition (gfx950: 256 CUs, 64-lane waves)
//   - a row is `chunks` 16-byte chunks; L lanes share a row (lane `sub` takes chunks
//     sub, sub+L, ..), so one wave load instruction fetches 64/L whole rows as 64/L
//     contiguous segments of L*16 bytes: fully coalesced global_load_dwordx4.
//   - rows are dealt round-robin to waves in groups of R = 64/L rows; the four waves of a
//     workgroup take groups that are gridDim.x groups apart, so a run of adjacent, similar
//     rows (typical for a corpus ordered by module) is spread over many workgroups.
//   - U row groups are in flight per wave (U*V 16-byte loads per lane outstanding).
//   - dot product: per-lane fp32 FMA chain over its chunks, then an xor-butterfly over the L
//     lanes sharing the row.
//   - selection is fused: each wave keeps its best KP = k'+1 keys in lanes 0..KP-1 (sorted);
//     a wave-uniform threshold rejects almost every row with one 64-bit compare. The workgroup
//     merges its 4 lists and emits its best k' keys plus the (k'+1)-th as a bound. The finalize
//     kernel (ls_select.hip) proves the global top-k is contained in the emitted keys, or falls
//     back to an exact selection over the score vector S, which this kernel also writes.
#include "ls_select_dev.h"
#ifdef LS_SCAN_ABL_NOS  // timing ablation: no score vectors (results of unproven queries are wrong)
#define LS_SCAN_S(x) ((float*)nullptr)
#else
#define LS_SCAN_S(x) (x)
#endif

#include <hip/hip_fp16.h>

#include <algorithm>

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // one 16-byte chunk

// NOTE: never __builtin_bit_cast an ext-vector ELEMENT expression (x.y): clang reads the first
// lane. Copy the element to a scalar first.
__device__ __forceinline__ h2_t as_h2(float f) { return __builtin_bit_cast(h2_t, f); }

template <bool F16, int V>
struct QueryRegs;

template <int V>
struct QueryRegs<false, V> {
    float4 q[V];
    // raw query (d floats, any alignment) -> zero-padded registers; returns this lane's sum of squares
    __device__ __forceinline__ float load(const float* qp, int d, int sub, int L) {
        float ss = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int e = 4 * (sub + L * v);
            q[v].x = e + 0 < d ? qp[e + 0] : 0.0f;
            q[v].y = e + 1 < d ? qp[e + 1] : 0.0f;
            q[v].z = e + 2 < d ? qp[e + 2] : 0.0f;
            q[v].w = e + 3 < d ? qp[e + 3] : 0.0f;
            ss = fmaf(q[v].x, q[v].x, ss);
            ss = fmaf(q[v].y, q[v].y, ss);
            ss = fmaf(q[v].z, q[v].z, ss);
            ss = fmaf(q[v].w, q[v].w, ss);
        }
        return ss;
    }
    __device__ __forceinline__ void scale(float f) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            q[v].x *= f; q[v].y *= f; q[v].z *= f; q[v].w *= f;
        }
    }
    __device__ __forceinline__ float dot(const f32x4 (&x)[V]) const {
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float a = x[v].x, b = x[v].y, c = x[v].z, e = x[v].w;
            acc = fmaf(a, q[v].x, acc);
            acc = fmaf(b, q[v].y, acc);
            acc = fmaf(c, q[v].z, acc);
            acc = fmaf(e, q[v].w, acc);
        }
        return acc;
    }
};

template <int V>
struct QueryRegs<true, V> {
    h2_t q[V][4];
    float f[V][8];  // fp32 staging, dead after scale()
    __device__ __forceinline__ float load(const float* qp, int d, int sub, int L) {
        float ss = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int e = 8 * (sub + L * v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                f[v][j] = e + j < d ? qp[e + j] : 0.0f;
                ss = fmaf(f[v][j], f[v][j], ss);
            }
        }
        return ss;
    }
    // scale, then round the query to fp16 (LS_DTYPE_F16 semantics: both operands are fp16)
    __device__ __forceinline__ void scale(float s) {
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                q[v][j] = h2_t{(_Float16)(f[v][2 * j] * s), (_Float16)(f[v][2 * j + 1] * s)};
    }
    __device__ __forceinline__ float dot(const f32x4 (&x)[V]) const {
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float a = x[v].x, b = x[v].y, c = x[v].z, e = x[v].w;
            acc = __builtin_amdgcn_fdot2(as_h2(a), q[v][0], acc, false);
            acc = __builtin_amdgcn_fdot2(as_h2(b), q[v][1], acc, false);
            acc = __builtin_amdgcn_fdot2(as_h2(c), q[v][2], acc, false);
            acc = __builtin_amdgcn_fdot2(as_h2(e), q[v][3], acc, false);
        }
        return acc;
    }
};

// Sum over the L lanes that share a row, result in every lane of the group. Pure VALU: DPP
// inside 16-lane rows (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror), then gfx950's
// v_permlane16_swap / v_permlane32_swap across rows. (ds_bpermute-based shuffles go through the
// LDS crossbar, which becomes the bottleneck when 8 queries share one corpus pass.)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return v + __builtin_bit_cast(float, moved);
}
template <int L>
__device__ __forceinline__ float group_sum(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]: + lane ^ 1
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]: + lane ^ 2
    v = dpp_add<0x141>(v);  // row_half_mirror: + the other quad of each 8
    v = dpp_add<0x140>(v);  // row_mirror: + the other half of each 16
    if (L >= 32) {          // rows 0<->1, 2<->3
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    if (L >= 64) {          // lanes 0-31 <-> 32-63
        const unsigned u = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    return v;
}

// ---- several queries per corpus pass: reduce-SCATTER instead of U*NQ full reductions ------------
// A lane holds P = U*NQ partial dot products (row step u, query qi), pair index j = u*NQ + qi. The
// single-query path sums each of them over the L lanes of a row with its own butterfly (5-6
// cross-lane adds per pair, every lane ends with every sum) and then picks the lane that keeps it:
// ~13 VALU instructions per pair, more than the 12 FMAs that produced it. Here each butterfly step
// HALVES the live pairs instead: at bit b a lane keeps the pairs whose bit b equals its own and
// hands the others to its partner (lane ^ (1 << b)), so P pairs cost P - 1 cross-lane adds in
// total and lane `sub` ends with pair j = sub (P == L). The steps run over the same bits in the
// same order (1, 2, 4, .., L/2) and add the same two operands as group_sum, so every sum is
// BIT-IDENTICAL to the one the single-query kernel computes: a query's scores do not depend on
// the group it rides in.
template <int B>
__device__ __forceinline__ float xor_lane(float v) {  // value of lane ^ B
    if constexpr (B == 1)
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    else if constexpr (B == 2)
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    else
        return __shfl_xor(v, B, 64);  // few of these per tile (the live pairs are down to <= P/4)
}
template <int L, int B, int LIVE, int P>
struct rs_step {
    static __device__ __forceinline__ void run(float (&p)[P], int sub) {
        if constexpr (B < L) {
            if constexpr (LIVE > 1) {
                const bool hi = (sub & B) != 0;
#pragma unroll
                for (int c = 0; c < LIVE / 2; ++c) {
                    const float lo_v = p[2 * c], hi_v = p[2 * c + 1];
                    const float keep = hi ? hi_v : lo_v, send = hi ? lo_v : hi_v;
                    p[c] = keep + xor_lane<B>(send);
                }
                rs_step<L, 2 * B, LIVE / 2, P>::run(p, sub);
            } else {  // one pair left but lanes to spare: plain butterfly, both partners keep the sum
                p[0] = p[0] + xor_lane<B>(p[0]);
                rs_step<L, 2 * B, 1, P>::run(p, sub);
            }
        }
    }
};
__host__ __device__ constexpr int ls_ilog2(int v) { return v <= 1 ? 0 : 1 + ls_ilog2(v / 2); }

// value of group (lane % R) delivered to every lane: R scalar reads + a select chain, no LDS
template <int L>
__device__ __forceinline__ float pick_group(float s, int lane) {
    constexpr int R = LS_WAVE / L;
    float out = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 0));
#pragma unroll
    for (int r = 1; r < R; ++r) {
        const float vr =
            __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), r * L));
        out = (lane % R) == r ? vr : out;
    }
    return out;
}

// Waves per SIMD the register allocation must leave room for. The 8-query fp16 kernel of 3-chunk
// lanes (d = 384 fp16) would take 271 VGPRs, i.e. ONE wave per SIMD; held to 256 it spills 19
// registers outside the tile loop and runs two. (4-chunk lanes at 8 queries would spill 80-180:
// they stay at one wave.)
__host__ __device__ constexpr int scan_min_waves(bool f16, int V, int NQ) {
    return (f16 && V == 3 && NQ == 8) ? 2 : 1;
}

// SMALL: a wave sees at most 64 rows in the whole launch (small shards: a few tiles per wave).
// The running sorted list is then the wrong tool - nearly every row of a wave's first tiles
// enters it, one serial insert (~0.2 us) per row: 3.5 of the 10.3 us of a scan-only launch at
// N = 10 k - so the wave parks tile i's TR scores in lanes i*TR .. and ranks its <= 64 keys
// once, by counting, after the last tile.
// SMALL also keeps PF = 4 tiles in flight per wave: with a handful of tiles per wave each HBM round
// trip would otherwise be paid in sequence (1.0-1.4 us per tile of a 4-tile wave).
template <bool F16, int L, int V, int U, int NQ, bool SMALL>
__global__ __launch_bounds__(LS_SCAN_THREADS, scan_min_waves(F16, V, NQ)) void ls_scan_kernel(
    const f32x4* __restrict__ corpus, long long n, int chunks, const float* __restrict__ qraw,
    int d, int normalize, int reverse, float* __restrict__ S, long long s_stride,
    u64* __restrict__ cand, long long c_stride, u64* __restrict__ bound, long long b_stride,
    int kprime, int nfin, ls_fin_batch fin, void* __restrict__ gran, long long g_stride, u32 tag,
    float* __restrict__ qkeep) {
    // The first `nfin` workgroups of a launch run the PREVIOUS launch's selection jobs
    // (finalize_body, ls_select_dev.h) while every other workgroup scans for the current queries:
    // selection costs neither a launch nor a kernel boundary and hides under the scan.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dyn[];
    if ((int)blockIdx.x < nfin) {
        // (a job of this launch's OWN queries - ls_fin_params::wait - sweeps the tagged granules the
        // scan workgroups below are writing; they never wait for anything. Should they not get to
        // run while this workgroup holds its slot (a CU-masked stream, a partitioned device) the
        // sweep gives up after 200 ms and asks the host for a retry.)
        // (a single-query launch without a score vector keeps its raw query for the repair, ls_api.hip
        // mq_repair: the riding selection workgroup has the time, the scan workgroups do not)
        if (NQ == 1 && qkeep && blockIdx.x == 0)
            for (int e = threadIdx.x; e < d; e += LS_SCAN_THREADS) qkeep[e] = qraw[e];
        for (int j = blockIdx.x; j < fin.njobs; j += nfin) {  // (nfin workgroups share the fin.njobs jobs)
            if (j != (int)blockIdx.x) __syncthreads();        // the previous job's LDS is free again
            finalize_body<LS_SCAN_THREADS>(ls_fin_job(fin, j), smem_dyn, threadIdx.x);
        }
        return;
    }
    const int bid = (int)blockIdx.x - nfin;
    const int nblk = (int)gridDim.x - nfin;
    if (NQ == 1 && qkeep && nfin == 0 && bid == 0)  // (nothing rides on this launch: scan workgroup 0 copies)
        for (int e = threadIdx.x; e < d; e += LS_SCAN_THREADS) qkeep[e] = qraw[e];
#ifdef LS_HANDOFF_TIMING
    if (threadIdx.x == 0) atomicMax(&g_ho[0], ~wall_clock64());
#endif
#ifdef LS_SCAN_TIMING  // developer instrumentation: phase stamps (100 MHz ticks) of one scan workgroup
    unsigned long long stamp[6];
#define LS_SSTAMP(i) stamp[i] = wall_clock64()
#else
#define LS_SSTAMP(i) do {} while (0)
#endif
    LS_SSTAMP(0);
    constexpr int R = LS_WAVE / L;  // rows per wave load step
    constexpr int TR = U * R;       // rows per tile: one tile = U steps = TR contiguous rows
    static_assert(TR <= LS_WAVE, "a tile's scores must fit one per lane");
    const int lane = threadIdx.x & (LS_WAVE - 1);
    const int wave = threadIdx.x / LS_WAVE;
    const int sub = lane & (L - 1);
    const int grp = lane / L;
    const int kp = kprime + 1;

    // Tiles are dealt round-robin to waves; the 4 waves of a workgroup take 4 adjacent tiles,
    // so one workgroup iteration covers 4*TR contiguous rows and its S stores fill whole lines.
    const long long W = (long long)nblk * LS_SCAN_WAVES;
    const long long gw = (long long)bid * LS_SCAN_WAVES + wave;
    const long long NT = (n + TR - 1) / TR;

    constexpr int PF = SMALL ? 4 : 1;  // tile buffers (statically indexed: the loop body is unrolled PF times)
    f32x4 xb[PF][U][V];
    auto issue_loads = [&](f32x4 (&x)[U][V], long long t) {
        if (reverse) t = NT - 1 - t;  // optional back-to-front sweep (see ls_api.hip)
        const long long r0 = t * TR + grp;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long r = r0 + u * R;
            r = r < n ? r : n - 1;  // ragged last tile: re-read the last row, masked out below
            const f32x4* p = corpus + r * chunks + sub;
#pragma unroll
            for (int v = 0; v < V; ++v) x[u][v] = __builtin_nontemporal_load(p + L * v);
        }
    };
    long long t = gw;
    // the first tile's loads fly while the queries are prepared (SMALL: the query goes first -
    // loads return in order, it must not queue behind four tiles - and the tiles follow it)
    if (!SMALL && t < NT) issue_loads(xb[0], t);

    // NQ queries -> registers, with faiss.normalize_L2 (reference search/engine.py:242) fused
    // in: x *= 1/sqrt(sum x^2), rows of zero norm untouched. One pass over the corpus then
    // serves all NQ queries (the reference sends one query at a time; small batches share the
    // HBM traffic this way).
    QueryRegs<F16, V> qr[NQ];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        (void)qr[qi].load(qraw + (long long)qi * d, d, sub, L);
        float inv = 1.0f;
        if (normalize) {  // the library's canonical summation order (ls_common.h)
            const float ss = ls_wave_sumsq(qraw + (long long)qi * d, d, lane);
            if (ss > 0.0f) inv = 1.0f / sqrtf(ss);
        }
        qr[qi].scale(inv);
    }

    if constexpr (SMALL) {
        // hipcc waits with vmcnt(0) for the query (the normalise branch merges in front of its
        // first use), i.e. for every load issued before: the four tiles go out behind it
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int pf = 0; pf < PF; ++pf)
            if (t + pf * W < NT) issue_loads(xb[pf], t + pf * W);
    }
    LS_SSTAMP(1);
    u64 lst[NQ];  // per query, lanes 0..kp-1: this wave's best keys, descending
    u64 thr[NQ];  // key in lane kp-1 (wave-uniform): a row must beat it to matter
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        lst[qi] = 0;
        thr[qi] = 0;
    }

    int ti = 0;  // SMALL: tiles this wave has seen
    auto tile_step = [&](f32x4 (&x)[U][V]) {  // tile t sits in buffer x
        if (t >= NT) return;
        if constexpr (NQ > 1 && LS_SCAN_MQ_SCATTER) {
            // ---- several queries: all U*NQ partial sums, one reduce-scatter, one score per lane --
            constexpr int P = U * NQ;
            constexpr int NSC = ls_ilog2(L) < ls_ilog2(P) ? ls_ilog2(L) : ls_ilog2(P);  // scatter steps
            constexpr int NRES = P >> NSC;                                               // results per lane
            float p[P];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) p[u * NQ + qi] = qr[qi].dot(x[u]);
            const long long tt = reverse ? NT - 1 - t : t;
            t += W;
            if (t + (PF - 1) * W < NT) issue_loads(x, t + (PF - 1) * W);  // overlaps everything below
            rs_step<L, 1, P, P>::run(p, sub);
            // lane `sub` now holds pairs j = (c << NSC) | (sub & (2^NSC - 1)), c < NRES; with lanes
            // to spare (L > P) the copies in lanes sub >= P are ignored
            const bool holder = (sub >> NSC) == 0 || NSC == ls_ilog2(L);
            u64 key[NRES];
            int myq[NRES];
#pragma unroll
            for (int c = 0; c < NRES; ++c) {
                const int j = (c << NSC) | (sub & ((1 << NSC) - 1));
                const int qi = j % NQ, u = j / NQ;
                const long long row = tt * TR + u * R + grp;
                const bool valid = holder && row < n;
                if (valid && S) S[qi * s_stride + row] = p[c];
                key[c] = valid ? ls_make_key(p[c], (u32)row) : 0ull;
                myq[c] = qi;
            }
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
#pragma unroll
                for (int c = 0; c < NRES; ++c) {
                    const u64 kq = myq[c] == qi ? key[c] : 0ull;
                    u64 mask = __ballot(kq > thr[qi]);
                    while (mask) {  // rare once the threshold has warmed up
                        const int j = __ffsll((long long)mask) - 1;
                        mask &= mask - 1;
                        const u64 v = readlane64(kq, j);
                        if (v <= thr[qi]) continue;  // the ballot is older than the threshold
                        wave_insert(lst[qi], v, lane, kp);
                        thr[qi] = readlane64(lst[qi], kp - 1);
                    }
                }
            }
            ++ti;
            return;
        }
        // lane i < TR collects the score of tile row i, per query (SMALL: lane ti*TR + i)
        float sc[NQ];
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) sc[qi] = 0.0f;
        const int lane0 = SMALL ? ti * TR : 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                const float s = group_sum<L>(qr[qi].dot(x[u]));   // valid in all L lanes of a group
                const float sel = pick_group<L>(s, lane);         // lane i <- group (i % R)
                if (lane / R == (SMALL ? ti * U : 0) + u) sc[qi] = sel;
            }
        }
        const long long tt = reverse ? NT - 1 - t : t;
        const long long row = tt * TR + (lane - lane0);
        t += W;
        if (t + (PF - 1) * W < NT) issue_loads(x, t + (PF - 1) * W);  // overlaps the selection below
        const bool valid = lane >= lane0 && lane < lane0 + TR && row < n;
        ++ti;
        if constexpr (SMALL) {
#pragma unroll
            for (int qi = 0; qi < NQ; ++qi) {
                if (valid) {
                    if (S) S[qi * s_stride + row] = sc[qi];
                    lst[qi] = ls_make_key(sc[qi], (u32)row);  // this lane's one key of the launch
                }
            }
            return;
        }
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            if (valid && S) S[qi * s_stride + row] = sc[qi];  // TR contiguous floats (S == nullptr: ls_api.hip mq_repair)
            const u64 key = valid ? ls_make_key(sc[qi], (u32)row) : 0ull;
            u64 mask = __ballot(key > thr[qi]);
            while (mask) {  // rare once the threshold has warmed up
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const u64 v = readlane64(key, j);
                wave_insert(lst[qi], v, lane, kp);
                thr[qi] = readlane64(lst[qi], kp - 1);
            }
        }
    };
    while (t < NT) {  // buffers are named statically: PF calls per round
        tile_step(xb[0]);
        if constexpr (PF == 4) {
            tile_step(xb[1]);
            tile_step(xb[2]);
            tile_step(xb[3]);
        }
    }

    LS_SSTAMP(2);
    // merge the 4 wave lists of every query -> this workgroup's best kprime keys + bound
    __shared__ u64 sm[NQ][LS_SCAN_WAVES * LS_KP_MAX];
    if constexpr (SMALL) {
        // every lane holds at most one key: its rank among the wave's keys by counting (non-zero
        // keys are unique; the "no row" lanes hold 0 and rank behind every real key)
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi) {
            const u64 mine = lst[qi];
            int rank = 0;
            for (int t2 = 0; t2 < ti; ++t2) {  // only the lanes that can hold a key: ti tiles of TR rows
#pragma unroll
                for (int r = 0; r < TR; ++r) rank += readlane64(mine, t2 * TR + r) > mine;
            }
            if (lane < LS_KP_MAX) sm[qi][wave * LS_KP_MAX + lane] = 0ull;
            if (mine != 0ull && rank < kp) sm[qi][wave * LS_KP_MAX + rank] = mine;
        }
    } else {
#pragma unroll
        for (int qi = 0; qi < NQ; ++qi)
            if (lane < LS_KP_MAX) sm[qi][wave * LS_KP_MAX + lane] = (lane < kp) ? lst[qi] : 0ull;
    }
    __syncthreads();
    LS_SSTAMP(3);
    for (int qi = wave; qi < NQ; qi += LS_SCAN_WAVES) {  // wave w ranks queries w, w+4, ..
        const u64 mine = sm[qi][lane];  // LS_SCAN_WAVES * LS_KP_MAX == 64 slots
        int rank = 0;
        if constexpr (SMALL || LS_SCAN_MERGE_FILLED) {  // only the kp slots each wave filled (the others hold 0)
            for (int w = 0; w < LS_SCAN_WAVES; ++w)
                for (int j = 0; j < kp; ++j) {
                    const int i = w * LS_KP_MAX + j;
                    const u64 o = sm[qi][i];
                    rank += (o > mine) || (o == mine && i < lane);
                }
        } else {
#pragma unroll 8
            for (int i = 0; i < LS_SCAN_WAVES * LS_KP_MAX; ++i) {
                const u64 o = sm[qi][i];
                rank += (o > mine) || (o == mine && i < lane);
            }
        }
        if (gran) {
            // same-launch selection: the keys are the whole hand-off - ONE 16-byte write-through
            // (sc1) store per key, {key, tag}: the selection workgroup recognises this launch's data
            // by the tag, so there is nothing to drain, no barrier and no counter behind the stores
            // (ls_fin_params::gran; rank-major: granule [rank][workgroup], plane kprime = bounds)
            if (rank <= kprime) {
                __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (char*)gran + (long long)qi * g_stride * 16, 0, nblk * (kprime + 1) * 16, LS_BUF_RSRC_FLAGS);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{(u32)mine, (u32)(mine >> 32), tag, 0u}, rsrc,
                                                       (rank * nblk + bid) * 16, 0, LS_AUX_SC1);
            }
        } else {
            if (rank < kprime) cand[qi * c_stride + (long long)bid * kprime + rank] = mine;
            if (rank == kprime) bound[qi * b_stride + bid] = mine;
        }
    }
#ifdef LS_HANDOFF_TIMING
    if (threadIdx.x == 0) atomicMax(&g_ho[1], wall_clock64());
#endif
#ifdef LS_SCAN_TIMING
    LS_SSTAMP(4);
    if (bid == nblk / 2 && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) cand[c_stride - 8 + i] = stamp[i + 1] - stamp[i];
    if (threadIdx.x == 0 && NQ == 1) {  // every workgroup's start / end tick: slot 7 of S is unused
        unsigned long long* life = reinterpret_cast<unsigned long long*>(S + 7 * s_stride);
        life[2 * bid] = stamp[0];
        life[2 * bid + 1] = stamp[4];
    }
#endif
}

// ------------------------------------------------------------------------------------------
#ifndef LS_UNROLL_V3
#define LS_UNROLL_V3 4
#endif
static constexpr int scan_unroll(int V) { return (V >= 3) ? LS_UNROLL_V3 : 8; }  // >= 8 loads in flight

int ls_scan_blocks(int64_t n, const ls_geom& g, int32_t n_cu) {
    constexpr int bpc = 2;  // scan workgroups per CU
    const int64_t TR = (int64_t)scan_unroll(g.V) * (LS_WAVE / g.L);
    const int64_t NT = (n + TR - 1) / TR;
    // minimum tiles per wave. Small shards: fewer blocks -> fewer candidate keys for the
    // piggy-backed finalize, which bounds the launch there (N=25k: 18.6 -> 11.7 us/step)
    constexpr int tpw = 4;
    int64_t b = (NT + LS_SCAN_WAVES * tpw - 1) / (LS_SCAN_WAVES * tpw);
    const int64_t cap = (int64_t)n_cu * bpc;
    if (b > cap) {
        // Big shards: tiles are dealt round-robin to 4*b waves, so the launch ends with a partial
        // round in which only frac(NT / 4b) of the waves still have a tile - too few to keep HBM
        // busy. Measured on config 2 (25 000 tiles): 512 workgroups (12.2 rounds) 47.55 us,
        // 448 (13.95 rounds) 47.03 us; config 2' 123.2 vs 120.4 us (tools/scan_blocks_sweep.py).
        // Pick the count in [1.5, 2] workgroups per CU (multiples of the 8 XCDs) whose last round
        // is the fullest.
        int64_t best = cap;
        double best_fill = -1.0;
        for (int64_t c = cap; c >= cap * 3 / 4; c -= 8) {
            const double rounds = (double)NT / (double)(c * LS_SCAN_WAVES);
            double fill = rounds - (double)(int64_t)rounds;
            if (fill == 0.0) fill = 1.0;
            if (fill > best_fill + 0.02) {  // near-ties go to the larger count
                best_fill = fill;
                best = c;
            }
        }
        b = best;
    }
    if (b < 1) b = 1;
    return (int)b;
}

template <bool F16, int L, int V, int NQ>
static int launch_lvq(const void* corpus, int64_t n, const ls_geom& g, const ls_scan_args& a,
                      hipStream_t s) {
    constexpr int U = scan_unroll(V);
    size_t smem = 0;
    if (a.nfin > 0) {
        const ls_fin_params& fp = a.fin.p0;
        const int keff = (int)((long long)fp.k < fp.n ? fp.k : fp.n);
        smem = ls_fin_lds_bytes(fp.keys_cap, keff);
    }
    const int nfw = std::min(a.nfin, LS_FIN_WG_MAX);  // selection workgroups (jobs nfw.. are second rounds)
    // single-query launches in which no wave sees more than 64 rows rank once instead of inserting
    constexpr int TR = U * (LS_WAVE / L);
    const long long waves = (long long)a.blocks * LS_SCAN_WAVES;
    const long long tiles_per_wave = ((n + TR - 1) / TR + waves - 1) / waves;
    // (with two or more workgroups per CU the other one hides the latency: measured neutral at
    // N = 50 k, 3 % slower at 100 k, 6-11 % faster at 25 k and 10 k)
    const bool small = LS_SCAN_SMALL && NQ == 1 && tiles_per_wave * TR <= LS_SCAN_SMALL_ROWS &&
                       a.blocks <= LS_SCAN_SMALL_MAX_BLOCKS;
#define LS_SCAN_LAUNCH(SM)                                                                         \
    {                                                                                              \
        auto kern = ls_scan_kernel<F16, L, V, U, NQ, SM>;                                          \
        static ls_attr_once once;                                                                  \
        if (int rc = ls_set_max_dynamic_lds(once, (const void*)kern, LS_PIGGY_LDS_MAX)) return rc; \
        hipLaunchKernelGGL(kern, dim3(a.blocks + nfw), dim3(LS_SCAN_THREADS), smem, s,          \
                           (const f32x4*)corpus, (long long)n, g.chunks, a.d_q, g.d,               \
                           a.normalize ? 1 : 0, a.reverse ? 1 : 0, LS_SCAN_S(a.d_S), (long long)a.s_stride,   \
                           a.d_cand, (long long)a.c_stride, a.d_bound, (long long)a.b_stride,      \
                           a.kprime, nfw, a.fin, a.d_gran, (long long)a.g_stride, a.tag, a.d_qkeep);                          \
    }
    if constexpr (NQ == 1) {
        if (small) LS_SCAN_LAUNCH(true) else LS_SCAN_LAUNCH(false)
    } else {
        LS_SCAN_LAUNCH(false)
    }
#undef LS_SCAN_LAUNCH
    LS_HIP(hipGetLastError());
    return LS_OK;
}

template <bool F16, int L, int V>
static int launch_lv(const void* corpus, int64_t n, const ls_geom& g, const ls_scan_args& a,
                     hipStream_t s) {
    switch (a.nq) {
        case 1: return launch_lvq<F16, L, V, 1>(corpus, n, g, a, s);
        case 4: return launch_lvq<F16, L, V, 4>(corpus, n, g, a, s);
        case 8: return launch_lvq<F16, L, V, 8>(corpus, n, g, a, s);
    }
    ls_set_error("ls_launch_scan: unsupported queries per launch %d", a.nq);
    return LS_ERR_INVALID_ARG;
}

template <bool F16>
static int launch_dt(const void* corpus, int64_t n, const ls_geom& g, const ls_scan_args& a,
                     hipStream_t s) {
#define LS_CASE(LL, VV) \
    if (g.L == LL && g.V == VV) return launch_lv<F16, LL, VV>(corpus, n, g, a, s);
    LS_CASE(16, 1) LS_CASE(16, 2) LS_CASE(16, 3) LS_CASE(16, 4)
    LS_CASE(32, 3) LS_CASE(32, 4)
    LS_CASE(64, 3) LS_CASE(64, 4)
#undef LS_CASE
    ls_set_error("ls_launch_scan: unsupported row geometry L=%d V=%d", g.L, g.V);
    return LS_ERR_INVALID_ARG;
}

int ls_launch_scan(const void* d_corpus, int64_t n, const ls_geom& g, const ls_scan_args& a,
                   hipStream_t s) {
    if (n <= 0) return LS_OK;
    if (a.kprime < 1 || a.kprime + 1 > LS_KP_MAX) {
        ls_set_error("ls_launch_scan: kprime %d out of range", a.kprime);
        return LS_ERR_INVALID_ARG;
    }
    return g.elem == 2 ? launch_dt<true>(d_corpus, n, g, a, s) : launch_dt<false>(d_corpus, n, g, a, s);
}
