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
#include "ls_common.h"

#include <hip/hip_fp16.h>

typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // one 16-byte chunk

template <bool F16, int V>
struct QueryRegs;

template <int V>
struct QueryRegs<false, V> {
    float4 q[V];
    __device__ __forceinline__ void load(const float* qp, int sub, int L) {
#pragma unroll
        for (int v = 0; v < V; ++v) q[v] = *reinterpret_cast<const float4*>(qp + 4 * (sub + L * v));
    }
    __device__ __forceinline__ float dot(const u32x4 (&x)[V]) const {
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            acc = fmaf(__builtin_bit_cast(float, x[v].x), q[v].x, acc);
            acc = fmaf(__builtin_bit_cast(float, x[v].y), q[v].y, acc);
            acc = fmaf(__builtin_bit_cast(float, x[v].z), q[v].z, acc);
            acc = fmaf(__builtin_bit_cast(float, x[v].w), q[v].w, acc);
        }
        return acc;
    }
};

template <int V>
struct QueryRegs<true, V> {
    h2_t q[V][4];
    __device__ __forceinline__ void load(const float* qp, int sub, int L) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float* p = qp + 8 * (sub + L * v);
            const float4 a = *reinterpret_cast<const float4*>(p);
            const float4 b = *reinterpret_cast<const float4*>(p + 4);
            // the prepared query already holds fp16-representable values: exact narrowing
            q[v][0] = h2_t{(_Float16)a.x, (_Float16)a.y};
            q[v][1] = h2_t{(_Float16)a.z, (_Float16)a.w};
            q[v][2] = h2_t{(_Float16)b.x, (_Float16)b.y};
            q[v][3] = h2_t{(_Float16)b.z, (_Float16)b.w};
        }
    }
    __device__ __forceinline__ float dot(const u32x4 (&x)[V]) const {
        float acc = 0.0f;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, x[v].x), q[v][0], acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, x[v].y), q[v][1], acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, x[v].z), q[v][2], acc, false);
            acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, x[v].w), q[v][3], acc, false);
        }
        return acc;
    }
};

template <int L>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int m = L / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Insert key `v` (wave-uniform) into the wave's sorted list (lanes 0..kp-1, descending).
__device__ __forceinline__ void wave_insert(u64& lst, u64 v, int lane, int kp) {
    const int cnt = __popcll(__ballot(lst > v));  // lanes >= kp hold 0 and never count
    if (cnt < kp) {
        const u64 up = __shfl_up(lst, 1, 64);
        lst = (lane > cnt) ? up : (lane == cnt ? v : lst);
        if (lane >= kp) lst = 0;
    }
}

template <bool F16, int L, int V, int U, bool NT>
__global__ __launch_bounds__(LS_SCAN_THREADS) void ls_scan_kernel(
    const u32x4* __restrict__ corpus, long long n, int chunks, const float* __restrict__ qprep,
    float* __restrict__ S, u64* __restrict__ cand, u64* __restrict__ bound, int kprime) {
    constexpr int R = LS_WAVE / L;  // rows per wave step
    const int lane = threadIdx.x & (LS_WAVE - 1);
    const int wave = threadIdx.x / LS_WAVE;
    const int sub = lane & (L - 1);
    const int grp = lane / L;
    const int kp = kprime + 1;

    QueryRegs<F16, V> qr;
    qr.load(qprep, sub, L);

    const long long W = (long long)gridDim.x * LS_SCAN_WAVES;
    const long long gw = (long long)wave * gridDim.x + blockIdx.x;
    const long long NG = (n + R - 1) / R;

    u64 lst = 0;  // lanes 0..kp-1: this wave's best keys, descending
    u64 thr = 0;  // key in lane kp-1 (wave-uniform): a row must beat it to matter

    for (long long g0 = gw; g0 < NG; g0 += W * U) {
        u32x4 x[U][V];
        long long row[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = (g0 + (long long)u * W) * R + grp;
            const long long rc = row[u] < n ? row[u] : n - 1;  // clamp: tail re-reads last row
            const u32x4* p = corpus + rc * chunks + sub;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (NT)
                    x[u][v] = __builtin_nontemporal_load(p + L * v);
                else
                    x[u][v] = p[L * v];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float s = group_sum<L>(qr.dot(x[u]));
            const bool valid = row[u] < n;
            if (valid && sub == 0) S[row[u]] = s;
            const u64 key = valid ? ls_make_key(s, (u32)row[u]) : 0ull;
            u64 mask = __ballot(sub == 0 && key > thr);
            while (mask) {  // rare once the threshold has warmed up
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const u64 v = __shfl(key, j, 64);
                wave_insert(lst, v, lane, kp);
                thr = __shfl(lst, kp - 1, 64);
            }
        }
    }

    // merge the 4 wave lists -> this workgroup's best kprime keys + bound
    __shared__ u64 sm[LS_SCAN_WAVES * LS_KP_MAX];
    if (lane < LS_KP_MAX) sm[wave * LS_KP_MAX + lane] = (lane < kp) ? lst : 0ull;
    __syncthreads();
    if (wave == 0) {
        const u64 mine = sm[lane];  // LS_SCAN_WAVES * LS_KP_MAX == 64 slots
        int rank = 0;
#pragma unroll 8
        for (int i = 0; i < LS_SCAN_WAVES * LS_KP_MAX; ++i) {
            const u64 o = sm[i];
            rank += (o > mine) || (o == mine && i < lane);
        }
        if (rank < kprime) cand[(long long)blockIdx.x * kprime + rank] = mine;
        if (rank == kprime) bound[blockIdx.x] = mine;
    }
}

// ------------------------------------------------------------------------------------------
int ls_scan_blocks(int64_t n, const ls_geom& g, int32_t n_cu) {
    static int bpc = -1;
    if (bpc < 0) {
        const char* e = getenv("LS_SCAN_BPC");
        bpc = e ? atoi(e) : 2;
        if (bpc < 1) bpc = 1;
    }
    const int64_t R = LS_WAVE / g.L;
    const int64_t NG = (n + R - 1) / R;
    int64_t b = (NG + LS_SCAN_WAVES * 4 - 1) / (LS_SCAN_WAVES * 4);  // >= 4 row groups per wave
    const int64_t cap = (int64_t)n_cu * bpc;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

template <bool F16, int L, int V>
static int launch_lv(const void* corpus, int64_t n, const ls_geom& g, const float* q, float* S,
                     u64* cand, u64* bound, int blocks, int kprime, hipStream_t s) {
    static int nt = -1;
    if (nt < 0) {
        const char* e = getenv("LS_SCAN_NT");
        nt = e ? atoi(e) : 1;
    }
    constexpr int U = (V >= 3) ? 4 : 8;  // >= 8 loads of 16 B per lane in flight
    if (nt)
        hipLaunchKernelGGL((ls_scan_kernel<F16, L, V, U, true>), dim3(blocks),
                           dim3(LS_SCAN_THREADS), 0, s, (const u32x4*)corpus, (long long)n,
                           g.chunks, q, S, cand, bound, kprime);
    else
        hipLaunchKernelGGL((ls_scan_kernel<F16, L, V, U, false>), dim3(blocks),
                           dim3(LS_SCAN_THREADS), 0, s, (const u32x4*)corpus, (long long)n,
                           g.chunks, q, S, cand, bound, kprime);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

template <bool F16>
static int launch_dt(const void* corpus, int64_t n, const ls_geom& g, const float* q, float* S,
                     u64* cand, u64* bound, int blocks, int kprime, hipStream_t s) {
#define LS_CASE(LL, VV)                                                                    \
    if (g.L == LL && g.V == VV)                                                            \
        return launch_lv<F16, LL, VV>(corpus, n, g, q, S, cand, bound, blocks, kprime, s);
    LS_CASE(16, 1) LS_CASE(16, 2) LS_CASE(16, 3) LS_CASE(16, 4)
    LS_CASE(32, 3) LS_CASE(32, 4)
    LS_CASE(64, 3) LS_CASE(64, 4)
#undef LS_CASE
    ls_set_error("ls_launch_scan: unsupported row geometry L=%d V=%d", g.L, g.V);
    return LS_ERR_INVALID_ARG;
}

int ls_launch_scan(const void* d_corpus, int64_t n, const ls_geom& g, const float* d_q, float* d_S,
                   u64* d_cand, u64* d_bound, int32_t blocks, int32_t kprime, hipStream_t s) {
    if (n <= 0) return LS_OK;
    if (kprime < 1 || kprime + 1 > LS_KP_MAX) {
        ls_set_error("ls_launch_scan: kprime %d out of range", kprime);
        return LS_ERR_INVALID_ARG;
    }
    return g.elem == 2 ? launch_dt<true>(d_corpus, n, g, d_q, d_S, d_cand, d_bound, blocks, kprime, s)
                       : launch_dt<false>(d_corpus, n, g, d_q, d_S, d_cand, d_bound, blocks, kprime, s);
}
