// ls_wsel.hip — the batched path's select step as ONE WAVE per query in at most 48 VGPRs.
//
// Why: the MFMA pass (ls_gemm.hip) runs two 229-register waves per SIMD - 464 of the SIMD's 512
// VGPRs - and 96 KB of the CU's 160 KB LDS. A kernel whose waves need <= 48 registers and a few
// KB of LDS each is co-resident with it: tools/coresidency_probe.hip measured 1024 such waves
// running entirely INSIDE a pass-shaped kernel (3..15 us after its start), while 80-register
// waves waited for its end. So under LS_FLAG_PIPELINE the select of batch i runs as filler waves
// under the pass of batch i+1 instead of taking every CU for ~19 us between two passes
// (ls_batch_select_kernel: 256 threads, 88 registers). Same job, same results: the exact top-k of
// one query's candidate queues (faiss index.search's heap + reorder half, reference
// src/lean_explore/search/engine.py:250), with the same verification flags.
//
// One wave per query, no workgroup barrier anywhere; LDS (wave-private): keys[1024] | surv[128] |
// hist[256]. Shapes it serves: k <= 128, at most 1024 candidate keys (the plan's +5 sigma
// estimate) and at most 128 corpus slices (512 queues per query); everything else keeps
// ls_batch_select_kernel.
#include "ls_select_dev.h"

#include <hip/hip_ext.h>

#define LS_WSEL_KEYS 1024
#define LS_WSEL_MAXK 128

namespace {

// 8-bit radix passes from bit `hb` down over one 32-bit half of the keys that satisfy `match`
// (HI: the score half of every non-zero key; !HI: the row half of the keys whose score half is
// `eq_hi`). Returns the selected prefix; krem / neq follow the selected bin. Stops as soon as the
// whole bin is needed (its prefix alone is then a threshold that admits exactly the wanted keys).
template <bool HI>
__device__ __forceinline__ u32 wave_radix(const u64* keys, int cnt, u32* hist, int hb, u32 vmax,
                                          u32 eq_hi, u32& krem, u32& neq, int lane) {
    u32 pref = hb == 31 ? 0u : (vmax >> (hb + 1) << (hb + 1));  // the bits all values share
    u32 pmask = hb == 31 ? 0u : ~((2u << hb) - 1u);
    const int npass = (hb + 8) / 8;
    for (int pass = 0; pass < npass; ++pass) {
        const int top = hb - 8 * pass;
        const int shift = top >= 7 ? top - 7 : 0;
        const u32 dmask = top >= 7 ? 255u : ((2u << top) - 1u);
        reinterpret_cast<uint4*>(hist)[lane] = make_uint4(0, 0, 0, 0);
        wave_lds_fence();
        for (int i = lane; i < ((cnt + 63) & ~63); i += 64) {
            const u64 key = i < cnt ? keys[i] : 0ull;
            const u32 v = HI ? (u32)(key >> 32) : (u32)key;
            const bool act = key != 0ull && (HI || (u32)(key >> 32) == eq_hi) && (v & pmask) == pref;
            wave_hist_add(hist, (v >> shift) & dmask, act, lane);
        }
        wave_lds_fence();
        const u32 bin = wave_find_bin(hist, krem, &neq, lane);
        pref |= bin << shift;
        pmask |= dmask << shift;
        if (neq == krem) break;
    }
    return pref;
}

}  // namespace

// QPL: queues per lane = ceil(4 * nsplits / 64)
template <int QPL>
__global__ __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(48))) void ls_wave_select_kernel(
    const uint2* __restrict__ queues, const u32* __restrict__ counts, int nsplits, int k,
    long long base, long long n, long long rows_per_split, u32* __restrict__ overflow,
    float* __restrict__ out_scores, long long* __restrict__ out_indices) {
    __shared__ __attribute__((aligned(16))) u64 keys[LS_WSEL_KEYS];
    __shared__ __attribute__((aligned(16))) u64 surv[LS_WSEL_MAXK];
    __shared__ __attribute__((aligned(16))) u32 hist[256];
    constexpr int cap = LS_GEMM_QCAP;
    const int q = blockIdx.x, lane = threadIdx.x;
    const int nqueues = nsplits * 4;
    const uint2* qbase = queues + (long long)q * nqueues * cap;
    if (overflow[q]) return;  // a queue overflowed in the MFMA pass: the exact scan path re-runs the query

    // ---- gather: queue lengths -> slot ranges (wave prefix sum) -> keys in LDS ------------------
    u32 c[QPL];
    u32 ct = 0;
#pragma unroll
    for (int h = 0; h < QPL; ++h) {
        const int qi = lane + h * 64;  // == split * 4 + quarter
        c[h] = qi < nqueues ? counts[(long long)q * nqueues + qi] : 0u;
        ct += c[h];
    }
    u32 inc = ct;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = (u32)__shfl_up((int)inc, o, 64);
        if (lane >= o) inc += t;
    }
    const int cnt = __builtin_amdgcn_readlane((int)inc, 63);
    if (cnt > LS_WSEL_KEYS) {  // more candidates than this kernel holds: the exact scan path handles the query
        if (lane == 0) overflow[q] = 1u;
        return;
    }
    u32 start = inc - ct;
#pragma unroll
    for (int h = 0; h < QPL; ++h) {
        const int qi = lane + h * 64;
        const long long rb = (long long)(qi >> 2) * rows_per_split;
        const uint4* src = reinterpret_cast<const uint4*>(qbase + (long long)qi * cap);
        for (u32 e0 = 0; e0 < c[h]; e0 += 2) {  // cap is even: the 16-byte loads stay inside the queue
            const uint4 a = src[e0 >> 1];
            {
                const long long row = rb + (long long)a.y;
                keys[start + e0] = row < n ? ls_make_key(__uint_as_float(a.x), (u32)row) : 0ull;
            }
            if (e0 + 1 < c[h]) {
                const long long row = rb + (long long)a.w;
                keys[start + e0 + 1] = row < n ? ls_make_key(__uint_as_float(a.z), (u32)row) : 0ull;
            }
        }
        start += c[h];
    }
    wave_lds_fence();

    // ---- exact k-th largest key: radix select on the score half, then (ties) on the row half ------
    u32 vmax = 0, vmin = 0xffffffffu, nnz = 0;
    for (int i = lane; i < cnt; i += 64) {
        const u64 key = keys[i];
        if (key != 0ull) {
            const u32 hi = (u32)(key >> 32);
            vmax = hi > vmax ? hi : vmax;
            vmin = hi < vmin ? hi : vmin;
            ++nnz;
        }
    }
    vmax = wave_max(vmax);
    vmin = wave_min(vmin);
    nnz = wave_sum(nnz);
    const int kk = (u32)k < nnz ? k : (int)nnz;  // min(k, #valid candidates)
    u64 T = 0;
    if (kk > 0) {
        u32 krem = (u32)kk, neq = nnz;
        u32 T_hi = vmax, T_lo = 0;
        const u32 diff = vmax ^ vmin;
        if (diff) T_hi = wave_radix<true>(keys, cnt, hist, 31 - __clz((int)diff), vmax, 0u, krem, neq, lane);
        if (neq > krem) {  // several candidates share the k-th score: the lowest rows win
            u32 lmax = 0, lmin = 0xffffffffu;
            for (int i = lane; i < cnt; i += 64) {
                const u64 key = keys[i];
                if (key != 0ull && (u32)(key >> 32) == T_hi) {
                    const u32 lo = (u32)key;
                    lmax = lo > lmax ? lo : lmax;
                    lmin = lo < lmin ? lo : lmin;
                }
            }
            lmax = wave_max(lmax);
            lmin = wave_min(lmin);
            const u32 ldiff = lmax ^ lmin;  // non-zero: neq > krem >= 1 distinct rows
            T_lo = wave_radix<false>(keys, cnt, hist, 31 - __clz((int)ldiff), lmax, T_hi, krem, neq, lane);
        }
        T = ((u64)T_hi << 32) | (u64)T_lo;  // exactly kk non-zero keys are >= T

        // ---- survivors -> surv[] (ballot prefix), then rank by counting ---------------------------
        u32 nsurv = 0;  // wave-uniform
        for (int i = lane; i < ((cnt + 63) & ~63); i += 64) {
            const u64 key = i < cnt ? keys[i] : 0ull;
            const bool keep = key != 0ull && key >= T;
            const u64 bal = __ballot(keep);
            if (keep) surv[nsurv + __popcll(bal & ((1ull << lane) - 1ull))] = key;
            nsurv += (u32)__popcll(bal);
        }
        wave_lds_fence();
        u64 mine0 = lane < kk ? surv[lane] : 0ull, mine1 = lane + 64 < kk ? surv[lane + 64] : 0ull;
        int r0 = 0, r1 = 0;
        for (int j = 0; j < kk; ++j) {
            const u64 o = surv[j];  // LDS broadcast read
            r0 += o > mine0;
            r1 += o > mine1;
        }
        wave_lds_fence();
        // keys[] is dead: the ordered result goes to its first kk slots
        if (lane < kk) keys[r0] = mine0;
        if (lane + 64 < kk) keys[r1] = mine1;
        wave_lds_fence();
    }
    if (kk < k && lane == 0) overflow[q] = 2u;  // the speculative tau let < k rows through
    for (int i = lane; i < k; i += 64) {
        const u64 key = i < kk ? keys[i] : 0ull;
        out_scores[(long long)q * k + i] = ls_key_score(key);
        out_indices[(long long)q * k + i] = ls_key_index(key, base);
    }
}

bool ls_wave_select_ok(int nsplits, int k, int keys_need) {
    // (up to 8 queues per lane: the 16-per-lane instantiation needs 58 registers and would not fit beside a pass)
    return k <= LS_WSEL_MAXK && keys_need <= LS_WSEL_KEYS && nsplits <= 128;
}

int ls_launch_wave_select(const ls_gemm_bufs& b, int nsplits, int64_t nq, int k, int64_t base, int64_t n,
                          int64_t rows_per_split, float* d_out_scores, int64_t* d_out_indices,
                          hipStream_t s, hipEvent_t done_event) {
    if (!ls_wave_select_ok(nsplits, k, 0)) {
        ls_set_error("batched path: shape outside the one-wave select kernel");
        return LS_ERR_INVALID_ARG;
    }
#define LS_WSEL_LAUNCH(P)                                                                              \
    hipExtLaunchKernelGGL(ls_wave_select_kernel<P>, dim3((unsigned)nq), dim3(64), 0, s, nullptr,       \
                          done_event, 0, (const uint2*)b.d_queues, (const u32*)b.d_counts, nsplits, k, \
                          (long long)base, (long long)n, (long long)rows_per_split, b.d_overflow,      \
                          d_out_scores, (long long*)d_out_indices)
    if (nsplits <= 64) LS_WSEL_LAUNCH(4);
    else LS_WSEL_LAUNCH(8);
#undef LS_WSEL_LAUNCH
    LS_HIP(hipGetLastError());
    return LS_OK;
}
