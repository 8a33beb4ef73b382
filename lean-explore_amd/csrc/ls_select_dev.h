// ls_select_dev.h — workgroup-level exact k-selection on 64-bit keys (device code shared by
// the finalize kernel, the shard-merge kernel and the scan kernel's piggy-backed finalize).
//
// All selection is on the keys of ls_common.h, so "top-k under (score desc, row asc)" is
// "k largest unsigned integers", bit-exact by construction. Non-zero keys are unique (one per
// row); key 0 means "no result".
//
// lds_topk: given up to `cnt` keys in LDS it finds the exact k-th largest non-zero key T by
// MSB-first 8-bit radix select (4 passes over the score half; 4 more over the row half only
// when several candidates share the k-th score), compacts the keys >= T and orders them
// (rank-by-counting for k <= 256, bitonic sort above). No pass touches HBM.
//
// finalize_body: the heap + reorder half of faiss `index.search`
// (reference src/lean_explore/search/engine.py:250) for one query:
//   fast path : the blocks*k' keys emitted by the scan go to LDS (all loads in flight
//               together) -> lds_topk. The k-th best emitted key T is a lower bound of the true
//               k-th best. Every row the scan did NOT emit is <= its workgroup's bound, so if
//               max(bound) < T (or every bound is 0: nothing was withheld) the result is the
//               global top-k. Traffic: blocks*(k'+1)*8 B.
//   rescue    : otherwise one sweep over the score vector S[n] collects every row with
//               key >= T (a superset of the answer) into LDS -> lds_topk.
//   general   : if even that overflows LDS (e.g. all scores equal) or too few keys were
//               emitted: 4-pass radix select over S itself.
//   Every path is exact; only their cost differs.
#pragma once
#include "ls_common.h"

#include <type_traits>

#define LS_RES_CAP 2048  // == LS_MAX_K

// ---- a wave's running best-keys list (used by the scan and the BM25 candidate kernels) -----------
__device__ __forceinline__ u64 readlane64(u64 v, int l) {  // l must be wave-uniform
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
    return ((u64)hi << 32) | lo;
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


// ---- workgroup-wide bitonic sort, descending, m a power of two, keys in LDS ------------------
// Pair p of a stage is handled by thread p % nthreads. For strides j <= 64 both elements of
// pair p lie in the 128-element block p >> 6, and all lanes of a wave share p >> 6, so those
// stages need only wave-level ordering (LDS is in-order per wave); only strides >= 128 cross
// waves and take a workgroup barrier: 10 barriers instead of 55 for m = 1024.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void block_bitonic_desc(u64* a, int m, int tid, int nthreads) {
    for (int k2 = 2; k2 <= m; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int p = tid; p < (m >> 1); p += nthreads) {
                const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1));
                const int x = i | j;
                const u64 ai = a[i], ax = a[x];
                const bool desc = (i & k2) == 0;
                if ((ai < ax) == desc) {
                    a[i] = ax;
                    a[x] = ai;
                }
            }
            const int jn = j > 1 ? (j >> 1) : k2;  // stride of the next stage
            if (j > 64 || jn > 64)
                __syncthreads();
            else
                wave_lds_fence();
        }
    }
    __syncthreads();
}

// ---- the same network with the keys in registers ---------------------------------------------
// NT threads hold E keys each (element index = e*NT + tid; NT*E >= m, missing elements are 0 and
// sink to the end). Strides >= NT are compare-exchanges inside a thread, strides < 64 are lane
// shuffles, and only strides 64 .. NT/2 go through LDS (a[] doubles as the exchange buffer and
// receives the result). 1024 keys on 1024 threads: 10 LDS exchanges instead of 55 LDS stages.
template <int NT, int E>
__device__ __forceinline__ void block_bitonic_desc_regs(u64* a, int m, int tid) {
    // E == 1 also serves m < NT: threads >= m idle (they still reach every barrier)
    const bool active = E > 1 || tid < m;
    u64 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = active ? a[e * NT + tid] : 0ull;
    __syncthreads();
    for (int k2 = 2; k2 <= m; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            if (j >= NT) {  // partner in the same thread (compile-time register indices only)
#pragma unroll
                for (int JE = E / 2; JE >= 1; JE >>= 1) {
                    if (j != JE * NT) continue;
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        if ((e & JE) != 0) continue;
                        const int idx = e * NT + tid;
                        const bool desc = (idx & k2) == 0;
                        const u64 lo = v[e], hi = v[e | JE];
                        const bool swap = (lo < hi) == desc;
                        v[e] = swap ? hi : lo;
                        v[e | JE] = swap ? lo : hi;
                    }
                }
            } else if (j < 64) {  // partner lane
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int idx = e * NT + tid;
                    const u64 other = __shfl_xor(v[e], j, 64);
                    const bool take_max = ((idx & k2) == 0) == ((idx & j) == 0);
                    v[e] = ((other > v[e]) == take_max) ? other : v[e];
                }
            } else {  // partner in another wave: through LDS
                if (active) {
#pragma unroll
                    for (int e = 0; e < E; ++e) a[e * NT + tid] = v[e];
                }
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        const int idx = e * NT + tid;
                        const u64 other = a[idx ^ j];
                        const bool take_max = ((idx & k2) == 0) == ((idx & j) == 0);
                        v[e] = ((other > v[e]) == take_max) ? other : v[e];
                    }
                }
                __syncthreads();
            }
        }
    }
    if (active) {
#pragma unroll
        for (int e = 0; e < E; ++e) a[e * NT + tid] = v[e];
    }
    __syncthreads();
}
// m: power of two. Dispatch on the keys-per-thread count; anything else takes the LDS network.
template <int NT>
__device__ __forceinline__ void lds_sort_desc(u64* a, int m, int tid) {
    if (m <= NT) block_bitonic_desc_regs<NT, 1>(a, m, tid);
    else if (m == 2 * NT) block_bitonic_desc_regs<NT, 2>(a, m, tid);
    else if (m == 4 * NT) block_bitonic_desc_regs<NT, 4>(a, m, tid);
    else if (m == 8 * NT) block_bitonic_desc_regs<NT, 8>(a, m, tid);
    else block_bitonic_desc(a, m, tid, NT);
}

__device__ __forceinline__ int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Wave 0 finds the histogram bin that holds the krem-th largest element (krem is clamped to
// the number of elements counted). out[0] = bin, out[1] = krem - (#elements in higher bins),
// out[2] = hist[bin], out[3] = total, out[4] = clamped krem.
// Must be followed by __syncthreads() before `out` is read.
__device__ __forceinline__ void find_bin(const u32* hist, u32 krem, u32* out, int tid) {
    if (tid < 64) {
        const u32 h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2],
                  h3 = hist[4 * tid + 3];
        const u32 mine = h0 + h1 + h2 + h3;
        u32 suf = mine;  // inclusive suffix sum over lanes >= tid
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_down(suf, o, 64);
            if (tid + o < 64) suf += t;
        }
        const u32 above = suf - mine;
        const u32 total = __shfl(suf, 0, 64);
        if (krem > total) krem = total;
        if (tid == 0) {
            out[3] = total;
            out[4] = krem;
        }
        if (krem > above && krem <= above + mine) {  // exactly one lane when krem >= 1
            u32 cum = above;
            int b;
            u32 hb;
            if (cum + h3 >= krem) { b = 4 * tid + 3; hb = h3; }
            else {
                cum += h3;
                if (cum + h2 >= krem) { b = 4 * tid + 2; hb = h2; }
                else {
                    cum += h2;
                    if (cum + h1 >= krem) { b = 4 * tid + 1; hb = h1; }
                    else { cum += h1; b = 4 * tid; hb = h0; }
                }
            }
            out[0] = (u32)b;
            out[1] = krem - cum;
            out[2] = hb;
        }
    }
}

// hist[digit] += 1 for every active lane. Candidates for one query have nearly equal scores, so
// in the leading passes most lanes share one digit and plain LDS atomics would serialise on it:
// the first active lane's digit is counted once per wave by ballot, the other lanes (spread
// over many digits, few conflicts) use ordinary LDS atomics.
__device__ __forceinline__ void wave_hist_add(u32* hist, u32 digit, bool active, int lane) {
    // (several leader rounds, meant for BM25's discrete scores, measured slower than the atomics
    // they save: score passes 14.9 vs 10.6 us on the 3-token / 200 k-name query)
    const u64 act = __ballot(active);
    if (act) {
        const int leader = __ffsll((long long)act) - 1;
        const u32 dsel = (u32)__builtin_amdgcn_readlane((int)digit, leader);
        const u64 same = __ballot(active && digit == dsel);
        if (lane == leader) atomicAdd(&hist[dsel], (u32)__popcll(same));
        if (active && digit != dsel) atomicAdd(&hist[digit], 1u);
    }
}

// ---- single-wave helpers (ls_wsel.hip, the in-launch tau of ls_gemm.hip): no workgroup barrier ------
// (whole wave active; pure VALU: DPP inside the 16-lane rows, v_permlane16_swap / v_permlane32_swap across
// them - a ds_bpermute butterfly is six LDS round trips of ~120 cycles for a wave that is alone on its SIMD)
template <class OP>
__device__ __forceinline__ u32 wave_allreduce_dpp(u32 v, OP op) {
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));   // lane ^ 1
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));   // lane ^ 2
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true));  // row_half_mirror
    v = op(v, (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true));  // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = op((u32)r[0], (u32)r[1]);
    r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return op((u32)r[0], (u32)r[1]);
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
    return wave_allreduce_dpp(v, [](u32 a, u32 b) { return a + b; });
}
__device__ __forceinline__ u32 wave_max(u32 v) {
    return wave_allreduce_dpp(v, [](u32 a, u32 b) { return a > b ? a : b; });
}
__device__ __forceinline__ u32 wave_min(u32 v) {
    return wave_allreduce_dpp(v, [](u32 a, u32 b) { return a < b ? a : b; });
}

// OR over the 64 lanes, in every lane, without the LDS crossbar: DPP inside 16-lane rows, then gfx950's
// v_permlane16_swap / v_permlane32_swap across rows (whole wave active).
template <int CTRL>
__device__ __forceinline__ u32 dpp_or(u32 v) {
    return v | (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
__device__ __forceinline__ u32 wave_max_dpp(u32 v) { return wave_max(v); }
__device__ __forceinline__ u32 wave_or_dpp(u32 v) {
    v = dpp_or<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_or<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_or<0x141>(v);  // row_half_mirror
    v = dpp_or<0x140>(v);  // row_mirror
    auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = (u32)r[0] | (u32)r[1];
    r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (u32)r[0] | (u32)r[1];
}

// Inclusive prefix sum over the 64 lanes in pure VALU (whole wave active): Hillis-Steele inside the
// 16-lane rows (row_shr 1, 2, 4, 8; lanes shifted in from outside a row read 0), then lane 15 of a
// row to the next row (row_bcast15 into rows 1 and 3), then lane 31 to the upper half (row_bcast31).
__device__ __forceinline__ u32 wave_incl_scan_dpp(u32 v) {
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);  // row_bcast15 -> rows 1, 3
    v += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);  // row_bcast31 -> rows 2, 3
    return v;
}

// The wave's histogram (256 bins, lane t holds bins 4t..4t+3 in LDS) -> the bin that holds the
// krem-th largest counted element; krem becomes the rank inside that bin, *nbin its population.
__device__ __forceinline__ u32 wave_find_bin(const u32* hist, u32& krem, u32* nbin, int lane) {
    const uint4 h = reinterpret_cast<const uint4*>(hist)[lane];
    const u32 mine = h.x + h.y + h.z + h.w;
    u32 suf = mine;  // inclusive suffix sum over lanes >= lane
    for (int o = 1; o < 64; o <<= 1) {
        const u32 t = (u32)__shfl_down((int)suf, o, 64);
        if (lane + o < 64) suf += t;
    }
    const u32 above = suf - mine;
    const bool here = krem > above && krem <= above + mine;  // exactly one lane (1 <= krem <= total)
    u32 b = 0, kr = 0, hb = 0;
    if (here) {
        u32 cum = above;
        if (cum + h.w >= krem) { b = 4 * lane + 3; hb = h.w; }
        else {
            cum += h.w;
            if (cum + h.z >= krem) { b = 4 * lane + 2; hb = h.z; }
            else {
                cum += h.z;
                if (cum + h.y >= krem) { b = 4 * lane + 1; hb = h.y; }
                else { cum += h.y; b = 4 * lane; hb = h.x; }
            }
        }
        kr = krem - cum;
    }
    const int src = __ffsll((long long)__ballot(here)) - 1;
    krem = (u32)__builtin_amdgcn_readlane((int)kr, src);
    *nbin = (u32)__builtin_amdgcn_readlane((int)hb, src);
    return (u32)__builtin_amdgcn_readlane((int)b, src);
}

// Exact top-k of the non-zero keys in keys[0..cnt) (LDS, left untouched) -> res[0..kk) sorted
// descending, kk = min(k, #non-zero). tmp: LDS scratch of >= 256 keys. k <= LS_RES_CAP.
// hist: 8 * 256 counters (one histogram per radix pass), misc: 8 * 8 words.
// The caller must have synchronised the workgroup after writing keys.
//
// Candidate keys of one query share their leading score bits (they all passed one threshold), so
// a fixed MSB-first digit schedule wastes its first passes on digits that separate nothing. One
// reduction pass finds the highest bit in which the score halves differ (and the number of
// non-zero keys); the 8-bit radix passes start there: typically 3 instead of 4, and the
// selected bin thins out at once. Survivors are compacted with one wave-aggregated LDS atomic
// per wave and step instead of one per key.
#ifdef LS_FIN_TIMING  // developer instrumentation: phase stamps (100 MHz ticks) of one finalize
__device__ unsigned long long g_fin_stamp[8];
#define LS_STAMP(i) do { if (tid == 0) g_fin_stamp[i] = wall_clock64(); } while (0)
#else
#define LS_STAMP(i) do {} while (0)
#endif
#ifdef LS_HANDOFF_TIMING  // developer instrumentation: timeline of a same-launch selection (tools/handoff_timeline.py)
static __device__ unsigned long long g_ho[2];  // [0] ~(earliest scan workgroup start), [1] latest scan workgroup end
#define LS_HO(i) do { if (tid == 0) ho[i] = wall_clock64(); } while (0)
#else
#define LS_HO(i) do {} while (0)
#endif
// splitter buckets (lds_topk): the sorted sample's every (LS_SS_SAMPLE/64)-th key is a splitter; bucket of a
// key = number of splitters (j = 0..62, descending) above it
#define LS_SS_SAMPLE 128
__device__ __forceinline__ int ss_bucket(const u64* ssort, u64 key) {
    constexpr int S = LS_SS_SAMPLE / 64;
    int lo = 0;
#pragma unroll
    for (int step = 32; step >= 1; step >>= 1)
        if (ssort[S * (lo + step - 1) + (S - 1)] > key) lo += step;
    return lo;
}
static __device__ __forceinline__ int lds_topk(const u64* keys, int cnt, int k, u64* res, u64* tmp, u32* hist, u32* misc,
                        int tid, int nt) {
    if (k > cnt) k = cnt;
    if (k <= 0) return 0;
    if (cnt <= 256 && cnt <= nt) {
        // A handful of keys (the pre-filtered candidates of a small k, a merge of short lists): rank
        // every key by counting and keep the ranks below k - one barrier instead of the ~15 of the
        // radix passes. Keys of value 0 ("no result") rank behind every real key.
        const int G = (cnt * 4 <= nt) ? 4 : 1;
        const int me = tid / G, part = tid % G;
        u64 mine = 0;
        int rank = 0;
        if (me < cnt) {
            mine = keys[me];
#pragma unroll 8
            for (int j = part; j < cnt; j += G) rank += keys[j] > mine;
        }
        if (G >= 2) rank += __shfl_xor(rank, 1, 64);
        if (G == 4) rank += __shfl_xor(rank, 2, 64);
        const int nnz = __syncthreads_count(me < cnt && part == 0 && mine != 0ull);
        if (me < cnt && part == 0 && mine != 0ull && rank < k) res[rank] = mine;
        __syncthreads();
        LS_STAMP(3);
        LS_STAMP(4);
        LS_STAMP(5);
        return k < nnz ? k : nnz;
    }
    if (cnt <= LS_SS_MAX_KEYS && (nt == 256 || nt == 512 || nt == 1024)) {  // (nt / LS_SS_SAMPLE: a power of two <= 64)
        // ---- splitter buckets (round 4): one bucketing by DATA QUANTILES instead of radix digits ----------
        // 128 of the keys (an even stride through the list) are ranked by counting; every 2nd of them is
        // a splitter, which cuts the key space into 64 buckets of ~cnt/64 keys WHATEVER the values look
        // like - real keys are distinct (distinct rows), so BM25's thousands of equal scores spread over
        // the buckets like anything else, where the radix digits of the score half put them all in one
        // bin (k-th key 8.4 + row passes and compaction 5.0 + sorting network 9.2 us at k = 1000). A key's
        // bucket is a 6-step binary search; a count per bucket and one prefix scan say which bucket holds
        // the k-th key; the buckets above it are scattered to their final slot ranges in res[], the
        // straddling one to tmp[]; a key's rank is its bucket's first slot + the larger keys of its own
        // bucket. 7 barriers, no radix pass, no sorting network; falls through to the radix code below if a
        // bucket turns out long (> 256: a stride that resonates with the list's structure).
        constexpr int NS = LS_SS_SAMPLE;
        const int G = nt / NS;  // threads per sample key
        u64* const ssort = reinterpret_cast<u64*>(hist);  // NS u64 <= hist[0..512)
        u32* const bcnt = hist + 512;
        u32* const bstart = hist + 576;
        u32* const bcur = hist + 640;
        unsigned char* const bid = reinterpret_cast<unsigned char*>(hist + 704);  // bucket of keys[i], <= 4096 B
        unsigned char* const posb = reinterpret_cast<unsigned char*>(hist);       // bucket of res[i] (ssort is dead by then)
        if (tid < NS) {
            // (a "no result" key in the sample becomes a distinct value below every real key - real keys carry
            // a non-zero score half - so that plain > ranks the sample as a permutation)
            const u64 sk = keys[(int)(((long long)tid * cnt) / NS)];
            tmp[tid] = sk != 0ull ? sk : (u64)(tid + 1);
        }
        if (tid < 64) bcnt[tid] = 0u;
        __syncthreads();
        {
            const int me = tid / G, part = tid % G;
            const u64 mine = tmp[me];
            int rank = 0;
#pragma unroll 8
            for (int j = part; j < NS; j += G) rank += tmp[j] > mine;
            // sum over the G = 2, 4 or 8 neighbouring lanes that share a sample key: DPP, no LDS crossbar
            if (G >= 2) rank += __builtin_amdgcn_update_dpp(0, rank, 0xB1, 0xf, 0xf, true);   // + lane ^ 1
            if (G >= 4) rank += __builtin_amdgcn_update_dpp(0, rank, 0x4E, 0xf, 0xf, true);   // + lane ^ 2
            if (G >= 8) rank += __builtin_amdgcn_update_dpp(0, rank, 0x141, 0xf, 0xf, true);  // + the other quad of each 8
            if (part == 0) ssort[rank] = mine;
        }
        __syncthreads();
        for (int i = tid; i < cnt; i += nt) {
            const u64 key = keys[i];
            if (key != 0ull) {
                const int b = ss_bucket(ssort, key);
                bid[i] = (unsigned char)b;
                atomicAdd(&bcnt[b], 1u);
            }
        }
        __syncthreads();
        if (tid < 64) {
            const u32 c = bcnt[tid];
            const u32 inc = wave_incl_scan_dpp(c);
            const u32 total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            const u32 kq = (u32)k < total ? (u32)k : total;  // min(k, #non-zero keys)
            const u32 start = inc - c;
            bstart[tid] = start;
            bcur[tid] = 0u;
            u32 big = start < kq ? c : 0u;  // only the buckets that reach into the top kq are walked
            big = wave_max_dpp(big);
            if (kq > 0 && start < kq && kq <= inc) {  // exactly one lane: the bucket that holds the kq-th key
                misc[0] = (u32)tid;
                misc[1] = start;
                misc[2] = c;
            }
            if (tid == 0) {
                misc[3] = kq;
                misc[4] = big;
            }
        }
        __syncthreads();
        const int kk = (int)misc[3];
        if (kk == 0) return 0;
        if (misc[4] <= 256u) {
            const int bk = (int)misc[0], nres = (int)misc[1], cbk = (int)misc[2];
            LS_STAMP(3);
            for (int i = tid; i < cnt; i += nt) {
                const u64 key = keys[i];
                if (key == 0ull) continue;
                const int b = (int)bid[i];
                if (b < bk) {
                    const u32 at = bstart[b] + atomicAdd(&bcur[b], 1u);
                    res[at] = key;
                    posb[at] = (unsigned char)b;
                } else if (b == bk) {
                    tmp[atomicAdd(&bcur[b], 1u)] = key;
                }
            }
            __syncthreads();
            LS_STAMP(4);
            constexpr int E = LS_RES_CAP / 256;  // keys of res[] per thread at most (256 threads, k = 2048)
            u64 mine[E];
            int rank[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = tid + e * nt;
                mine[e] = 0ull;
                rank[e] = 0;
                if (i < nres) {
                    const u64 key = res[i];
                    const int b = (int)posb[i];
                    const int lo = (int)bstart[b], n = (int)bcnt[b];
                    int r = lo;
#pragma unroll 8
                    for (int j = lo; j < lo + n; ++j) r += res[j] > key;
                    mine[e] = key;
                    rank[e] = r;
                }
            }
            u64 tk = 0ull;  // the straddling bucket: only its ranks below kk are results
            int tr = kk;
            if (tid < cbk) {
                tk = tmp[tid];
                int r = nres;
#pragma unroll 8
                for (int j = 0; j < cbk; ++j) r += tmp[j] > tk;
                tr = r;
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (tid + e * nt < nres) res[rank[e]] = mine[e];
            if (tr < kk) res[tr] = tk;
            __syncthreads();
            LS_STAMP(5);
            return kk;
        }
        __syncthreads();  // a long bucket: the radix code below starts over (it re-initialises hist and misc)
    }
    const int lane = tid & 63, wv = tid >> 6, nw = nt >> 6;
    const int cnt_pad = (cnt + 63) & ~63;  // whole waves take part in the ballots
    // ---- range of the score halves + number of non-zero keys (one pass, no atomics) ---------
    u32 vmax = 0, vmin = 0xffffffffu, nnz = 0;
    for (int i = tid; i < cnt; i += nt) {
        const u64 key = keys[i];
        if (key != 0ull) {
            const u32 hi = (u32)(key >> 32);
            vmax = hi > vmax ? hi : vmax;
            vmin = hi < vmin ? hi : vmin;
            ++nnz;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        const u32 a = (u32)__shfl_xor((int)vmax, o, 64), b = (u32)__shfl_xor((int)vmin, o, 64);
        vmax = a > vmax ? a : vmax;
        vmin = b < vmin ? b : vmin;
        nnz += (u32)__shfl_xor((int)nnz, o, 64);
    }
    // hist is free until the first pass: [0..nw) max, [16..16+nw) min, [32..32+nw) counts
    if (lane == 0) {
        hist[wv] = vmax;
        hist[16 + wv] = vmin;
        hist[32 + wv] = nnz;
    }
    __syncthreads();
    vmax = 0; vmin = 0xffffffffu; nnz = 0;
    for (int w = 0; w < nw; ++w) {
        vmax = hist[w] > vmax ? hist[w] : vmax;
        vmin = hist[16 + w] < vmin ? hist[16 + w] : vmin;
        nnz += hist[32 + w];
    }
    __syncthreads();
    int kk = (u32)k < nnz ? k : (int)nnz;  // min(k, #non-zero keys)
    if (kk == 0) return 0;
    const u32 diff = vmax ^ vmin;
    const int hb = diff ? 31 - __clz((int)diff) : -1;  // highest bit that separates any two scores
    const int npass = (hb + 8) / 8;                      // 0 when every score is the same
    for (int i = tid; i < 8 * 256; i += nt) hist[i] = 0;
    if (tid == 0) misc[7 * 8 + 7] = 0;  // survivor counter
    __syncthreads();
    u32 pref = hb >= 0 ? (vmax >> (hb + 1) << (hb + 1)) : vmax;  // the bits all scores share
    if (hb == 31) pref = 0;
    u32 pmask = hb >= 0 ? (hb == 31 ? 0u : ~((2u << hb) - 1u)) : 0xffffffffu;
    u32 krem = (u32)kk, neq = nnz;
    for (int pass = 0; pass < npass; ++pass) {  // score half, 8 bits at a time from bit hb down
        const int top = hb - 8 * pass;           // highest bit of this digit
        const int shift = top >= 7 ? top - 7 : 0;
        const u32 dmask = top >= 7 ? 255u : ((2u << top) - 1u);
        u32* h = hist + pass * 256;
        u32* ms = misc + pass * 8;
        for (int i = tid; i < cnt_pad; i += nt) {
            const u64 key = i < cnt ? keys[i] : 0ull;
            const u32 hi = (u32)(key >> 32);
            wave_hist_add(h, (hi >> shift) & dmask, key != 0ull && (hi & pmask) == pref, lane);
        }
        __syncthreads();
        find_bin(h, krem, ms, tid);
        __syncthreads();
        pref |= ms[0] << shift;
        pmask |= dmask << shift;
        krem = ms[1];
        neq = ms[2];
        // every key of the selected bin is needed: the prefix alone (low bits 0) already is a
        // threshold that admits exactly kk keys, so the remaining passes are skipped
        if (neq == krem) break;
    }
    LS_STAMP(3);
    const u32 T_hi = pref;
    u32 T_lo = 0;
    if (neq > krem) {  // several candidates share the k-th score: split them on the row half
        // the tied keys' row halves share their leading bits (0xffffffff - row, rows < n): one
        // reduction finds the highest bit in which they differ, the 8-bit passes start there
        u32 lmax = 0, lmin = 0xffffffffu;
        for (int i = tid; i < cnt; i += nt) {
            const u64 key = keys[i];
            if (key != 0ull && (u32)(key >> 32) == T_hi) {
                const u32 lo = (u32)key;
                lmax = lo > lmax ? lo : lmax;
                lmin = lo < lmin ? lo : lmin;
            }
        }
        for (int o = 32; o >= 1; o >>= 1) {
            const u32 a = (u32)__shfl_xor((int)lmax, o, 64), b = (u32)__shfl_xor((int)lmin, o, 64);
            lmax = a > lmax ? a : lmax;
            lmin = b < lmin ? b : lmin;
        }
        u32* red2 = hist + 7 * 256;  // the last row-pass histogram is free until pass 3 (never, now)
        if (lane == 0) {
            red2[wv] = lmax;
            red2[16 + wv] = lmin;
        }
        __syncthreads();
        lmax = 0; lmin = 0xffffffffu;
        for (int w = 0; w < nw; ++w) {
            lmax = red2[w] > lmax ? red2[w] : lmax;
            lmin = red2[16 + w] < lmin ? red2[16 + w] : lmin;
        }
        __syncthreads();
        if (lane < 32 && wv == 0) red2[lane] = 0;  // hand the words back as zeros
        __syncthreads();
        const u32 ldiff = lmax ^ lmin;             // non-zero: neq > krem >= 1 distinct rows
        const int lhb = 31 - __clz((int)ldiff);
        const int lpasses = (lhb + 8) / 8;         // <= 3 for shards below 2^24 rows
        u32 lpref = lhb == 31 ? 0u : (lmax >> (lhb + 1) << (lhb + 1));
        u32 lmask = lhb == 31 ? 0u : ~((2u << lhb) - 1u);
        for (int pass = 0; pass < lpasses; ++pass) {
            const int top = lhb - 8 * pass;
            const int shift = top >= 7 ? top - 7 : 0;
            const u32 dmask = top >= 7 ? 255u : ((2u << top) - 1u);
            u32* h = hist + (4 + pass) * 256;
            u32* ms = misc + (4 + pass) * 8;
            for (int i = tid; i < cnt_pad; i += nt) {
                const u64 key = i < cnt ? keys[i] : 0ull;
                const u32 lo = (u32)key;
                wave_hist_add(h, (lo >> shift) & dmask,
                              key != 0ull && (u32)(key >> 32) == T_hi && (lo & lmask) == lpref,
                              lane);
            }
            __syncthreads();
            find_bin(h, krem, ms, tid);
            __syncthreads();
            lpref |= ms[0] << shift;
            lmask |= dmask << shift;
            krem = ms[1];
            if (ms[2] == krem) break;  // the whole bin is needed: its prefix is the threshold
        }
        T_lo = lpref;
    }
    const u64 T = ((u64)T_hi << 32) | (u64)T_lo;  // exactly kk non-zero keys are >= T
    // ---- kk > 256: bucket the survivors by the FIRST radix digit, rank inside the buckets ----------
    // The first pass's histogram already says how many survivors each of its 256 bins holds (every
    // key of a bin above the selected one; the selected bin's share is that pass's remaining rank),
    // so a suffix sum gives every bucket its slot range, one LDS atomic per survivor scatters it
    // there, and a survivor's final rank is its bucket's first slot + the number of LARGER keys in
    // its own bucket - tens of LDS reads instead of the 55 stages of a 1024-key bitonic network
    // (10 us of the 24.8 us stand-alone selection at k = 1000). Buckets that would be walked for
    // too long (a few distinct scores: BM25, all-equal corpora) fall back to the sorting network.
    bool bucketed = false;
    u32* const bstart = hist + 1 * 256;  // the later passes' histograms are dead once T is known
    u32* const bcur = hist + 2 * 256;
    const int shift0 = hb >= 7 ? hb - 7 : 0;
    const u32 dmask0 = hb >= 7 ? 255u : (hb >= 0 ? ((2u << hb) - 1u) : 0u);
    if (kk > 256 && npass > 0) {
        if (tid < 64) {
            const u32 bin0 = misc[0], need0 = misc[1];  // pass 0: selected bin, survivors it contributes
            u32 sz[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const u32 b = 4u * tid + e;
                sz[e] = b > bin0 ? hist[b] : (b == bin0 ? need0 : 0u);
            }
            const u32 mine = sz[0] + sz[1] + sz[2] + sz[3];
            u32 suf = mine;  // inclusive suffix sum over lanes >= tid
            for (int o = 1; o < 64; o <<= 1) {
                const u32 t = (u32)__shfl_down((int)suf, o, 64);
                if (tid + o < 64) suf += t;
            }
            u32 at = suf - mine;  // survivors in higher bins
            u32 big = 0;
#pragma unroll
            for (int e = 3; e >= 0; --e) {
                bstart[4 * tid + e] = at;
                bcur[4 * tid + e] = 0u;
                at += sz[e];
                big = sz[e] > big ? sz[e] : big;
            }
            for (int o = 32; o >= 1; o >>= 1) {
                const u32 t = (u32)__shfl_xor((int)big, o, 64);
                big = t > big ? t : big;
            }
            if (tid == 0) misc[7 * 8 + 6] = big;
        }
        __syncthreads();
        bucketed = misc[7 * 8 + 6] <= 192u;
    }
    u64* dst = (kk <= 256) ? tmp : res;
    for (int i = tid; i < cnt_pad; i += nt) {  // one LDS atomic per wave and step
        const u64 key = i < cnt ? keys[i] : 0ull;
        const bool keep = key != 0ull && key >= T;
        if (bucketed) {
            if (keep) {
                const u32 b = ((u32)(key >> 32) >> shift0) & dmask0;
                res[bstart[b] + atomicAdd(&bcur[b], 1u)] = key;
            }
            continue;
        }
        const u64 bal = __ballot(keep);
        if (bal) {
            u32 base = 0;
            if (lane == 0) base = atomicAdd(&misc[7 * 8 + 7], (u32)__popcll(bal));
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            if (keep) dst[base + __popcll(bal & ((1ull << lane) - 1ull))] = key;
        }
    }
    __syncthreads();
    LS_STAMP(4);
    if (kk <= 256) {  // order by counting: rank = number of larger survivors
        // with threads to spare, 4 of them share one survivor (each counts a slice of the others;
        // the slices are summed by lane shuffles): the serial LDS walk is kk / 4 long
        const int G = (kk * 4 <= nt) ? 4 : 1;  // (2-way sharing measured slower than none)
        const int me = tid / G, part = tid % G;
        u64 mine = 0;
        int rank = 0;
        if (me < kk) {
            mine = tmp[me];
#pragma unroll 8
            for (int j = part; j < kk; j += G) rank += tmp[j] > mine;
        }
        if (G >= 2) rank += __shfl_xor(rank, 1, 64);
        if (G == 4) rank += __shfl_xor(rank, 2, 64);
        if (me < kk && part == 0) res[rank] = mine;
        __syncthreads();
    } else if (bucketed) {
        constexpr int E = LS_RES_CAP / 256;  // survivors per thread at most (256 threads, k = 2048)
        u64 mine[E];
        int rank[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int i = tid + e * nt;
            mine[e] = 0ull;
            rank[e] = 0;
            if (i < kk) {
                const u64 key = res[i];
                const u32 b = ((u32)(key >> 32) >> shift0) & dmask0;
                const int lo = (int)bstart[b], n = (int)bcur[b];
                int r = lo;
#pragma unroll 8
                for (int j = lo; j < lo + n; ++j) r += res[j] > key;  // (independent LDS reads: keep 8 in flight)
                mine[e] = key;
                rank[e] = r;
            }
        }
        __syncthreads();  // every bucket has been read: the ordered keys go back in place
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (tid + e * nt < kk) res[rank[e]] = mine[e];
        __syncthreads();
    } else {
        const int m = next_pow2(kk);
        for (int i = kk + tid; i < m; i += nt) res[i] = 0ull;
        __syncthreads();
        if (nt == 1024) lds_sort_desc<1024>(res, m, tid);
        else if (nt == 256) lds_sort_desc<256>(res, m, tid);
        else block_bitonic_desc(res, m, tid, nt);
    }
    LS_STAMP(5);
    return kk;
}

// exclusive prefix sum of one flag per thread over the workgroup; returns total via *total
__device__ __forceinline__ int block_excl_scan_flag(bool flag, int tid, int nthreads, u32* wsum,
                                                    int* total) {
    const int lane = tid & 63, wave = tid >> 6, nw = nthreads >> 6;
    const u64 b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = (u32)__popcll(b);
    __syncthreads();
    int off = 0, tot = 0;
    for (int w = 0; w < nw; ++w) {
        const int c = (int)wsum[w];
        if (w < wave) off += c;
        tot += c;
    }
    __syncthreads();
    *total = tot;
    return off + within;
}

// General exact top-k of S[0..n) -> res[0..) sorted descending; returns the number of valid keys.
// 4 radix passes over S for the k-th score, one gather of everything above it, one ordered
// gather of the lowest-index rows equal to it, then a sort. k <= LS_RES_CAP.
static __device__ int general_select(const float* __restrict__ S, long long n, int k, u64* res, u32* hist,
                              u32* misc, int tid, int nthreads) {
    u32 prefix = 0, pmask = 0;
    u32 krem = (long long)k < n ? (u32)k : (u32)n;
    int keff = (int)krem;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256; i += nthreads) hist[i] = 0;
        __syncthreads();
        for (long long r = tid; r < n; r += nthreads) {
            const float s = S[r];
            if (s > -FLT_MAX) {
                const u32 o = ls_ord(s);
                if ((o & pmask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        find_bin(hist, krem, misc, tid);
        __syncthreads();
        if (pass == 0) {
            keff = (int)misc[4];  // min(k, #valid rows)
            if (keff == 0) return 0;
        }
        prefix |= misc[0] << shift;
        pmask |= 255u << shift;
        krem = misc[1];
        __syncthreads();
    }
    const u32 T = prefix;                // ord() of the k-th best score
    const int above = keff - (int)krem;  // rows strictly better than T
    if (tid == 0) misc[4] = 0;
    __syncthreads();
    for (long long r = tid; r < n; r += nthreads) {
        const float s = S[r];
        if (s > -FLT_MAX && ls_ord(s) > T) res[atomicAdd(&misc[4], 1u)] = ls_make_key(s, (u32)r);
    }
    __syncthreads();
    // rows equal to T: take the krem lowest row indices (deterministic tie-break)
    int running = 0;
    for (long long b0 = 0; b0 < n && running < (int)krem; b0 += nthreads) {
        const long long r = b0 + tid;
        float s = 0.0f;
        bool flag = false;
        if (r < n) {
            s = S[r];
            flag = (s > -FLT_MAX) && ls_ord(s) == T;
        }
        int total;
        const int pos = running + block_excl_scan_flag(flag, tid, nthreads, hist, &total);
        if (flag && pos < (int)krem) res[above + pos] = ls_make_key(s, (u32)r);
        running += total;
    }
    __syncthreads();
    const int m = next_pow2(keff);
    for (int i = keff + tid; i < m; i += nthreads) res[i] = 0;
    __syncthreads();
    block_bitonic_desc(res, m, tid, nthreads);
    return keff;
}


// ---- LDS plan of one finalize ------------------------------------------------------------------
// keys[keys_cap] | res[res_cap] | tmp[256] | red[16]   (u64)   then   hist[8*256] | misc[64] (u32)
__host__ __device__ __forceinline__ int ls_fin_res_cap(int keff) {
    int p = 256;
    while (p < keff) p <<= 1;
    return p;
}
__host__ __device__ __forceinline__ size_t ls_fin_lds_bytes(int keys_cap, int keff) {
    return ((size_t)keys_cap + (size_t)ls_fin_res_cap(keff) + 256 + 16) * sizeof(u64) +
           (8 * 256 + 64) * sizeof(u32);
}

// One workgroup of NT threads produces the final (scores, rows)[k] of one query.
// (forced inline: as a call, `p` - an element of the kernel's by-value job array - would be copied to
// scratch memory to have an address: 984 bytes per lane)
// (Round 4 also tried a REHEARSAL for same-launch jobs - the whole fast path run once on the previous
// call's granules while the scan runs, side effects off, then the same instructions again on the real
// keys - on the theory that the job's code is cold: 61.6 vs 61.7 us per call, i.e. nothing. What the
// timeline (tools/handoff_timeline.py) did show was branches: see the pivot search below.)
template <int NT>
static __device__ __forceinline__ void finalize_body(const ls_fin_params& p, unsigned char* smem, int tid) {
    const int keff = (long long)p.k < p.n ? p.k : (int)p.n;
    u64* keys = reinterpret_cast<u64*>(smem);
    u64* res = keys + p.keys_cap;
    u64* tmp = res + ls_fin_res_cap(keff);
    u64* red = tmp + 256;
    u32* hist = reinterpret_cast<u32*>(red + 16);
    u32* misc = hist + 8 * 256;

    LS_STAMP(0);
#ifdef LS_HANDOFF_TIMING
    unsigned long long ho[8] = {};
#endif
    LS_HO(0);
    const int mc = p.blocks * p.kprime;
    int nvalid = 0;
    bool done = (p.n <= 0);
    u64 T = 0;  // k-th best emitted key: a lower bound of the true k-th best key

    if (!done && !p.force_slow && mc >= keff && mc <= p.keys_cap) {
        // ---- pivot: a lower bound of the k-th best key from ONE entry per workgroup ---------------------
        // Every scan workgroup emitted its keys best first. Take the m-th key of each (m = keys a
        // workgroup must contribute on average, ceil(k / blocks): 1 for k = 50 over 448 workgroups) and
        // let P be the r-th largest of these pivots, r = ceil(k / m): at least r workgroups then hold m
        // keys >= P, i.e. >= k keys >= P, so the k-th best key is >= P and everything below P can be
        // dropped before any selection work: ~60 of the 2240 candidates survive for config 2, a third
        // for k = 1000. The same holds for ANY subset of the workgroups, so every wave searches its own
        // slice of them (blocks / waves pivots, 1-4 per lane, in registers: bit-wise search on the score
        // half, counting by ballots) for the ceil(r / waves)-th largest, and P is the smallest of the
        // waves' answers: waves * ceil(r / waves) >= r pivots are >= P. (One wave over all 448 pivots: 56
        // dependent ballot-count-add chains, 1.7 us; a slice per wave: 16.)
        const int m_need = (keff + p.blocks - 1) / p.blocks;
        const bool prefilter = p.blocks >= 64 && p.blocks <= 1024 && m_need <= p.kprime;
        if (tid == 0) misc[7 * 8 + 4] = 0u;  // survivor count (a barrier follows before it is used)
        // (at most 4 slices: with 16 waves of 28 pivots each the minimum of 16 small-sample answers let a
        // fifth more keys through at k = 1000)
        constexpr int NW = NT / 64 < 4 ? NT / 64 : 4;       // waves that search a slice
        constexpr int PW = 1024 / (64 * NW);                // pivots per lane at most (blocks <= 1024)
        const int wv = tid >> 6, ln = tid & 63;
        const int Q = (p.blocks + NW - 1) / NW;             // pivots of one wave's slice
        u32 pv[PW];
#pragma unroll
        for (int j = 0; j < PW; ++j) pv[j] = 0u;
        u64 mb = 0;  // max over workgroups of the best key each one withheld
        // candidate keys: CH per thread and round in registers (config 2: one round of 9 on 256
        // threads)
        constexpr int CH = LS_FINAL_CAP / NT < 16 ? LS_FINAL_CAP / NT : 16;
        u64 mine[CH];
        if (p.gran) {
            // ---- tagged granules (same-launch jobs and their retries, ls_fin_params::gran) ----------
            // Thread t owns granules t, t + NT, ...: it re-reads the ones that do not carry the tag yet
            // (sc1 loads: L1 bypassed, nothing to invalidate) and the workgroup votes after every sweep.
            // Before the first sweep ONE lane sleeps on one granule, so that a selection workgroup
            // does not poll 40 KB per microsecond next to scan workgroups for the whole scan.
            constexpr int CG = LS_GRAN_MAX / NT;
            static_assert(CG <= CH, "granule slots must fit the candidate registers");
            const int G = p.blocks * (p.kprime + 1);
            __amdgpu_buffer_rsrc_t rsrc =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.gran), 0, G * 16, LS_BUF_RSRC_FLAGS);
            unsigned long long t0 = 0;
            if (p.wait) {
                t0 = wall_clock64();
                if (tid == 0) {
                    while (__builtin_amdgcn_raw_buffer_load_b128(rsrc, (G - p.blocks) * 16, 0, LS_AUX_SC1)[2] != p.tag) {
                        __builtin_amdgcn_s_sleep(8);
                        if (wall_clock64() - t0 > LS_ARRIVE_TIMEOUT_TICKS) break;
                    }
                }
                __syncthreads();
            }
            LS_HO(1);
            u32x4 gv[CG];
            u32 need = 0;
            bool complete = true;
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                gv[j] = u32x4{0u, 0u, 0u, 0u};
                if (tid + j * NT < G) need |= 1u << j;
            }
            for (unsigned it = 1;; ++it) {
#pragma unroll
                for (int j = 0; j < CG; ++j)
                    if (need & (1u << j))
                        gv[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (tid + j * NT) * 16, 0, LS_AUX_SC1);
#pragma unroll
                for (int j = 0; j < CG; ++j)
                    if (gv[j].z == p.tag) need &= ~(1u << j);
                if (__syncthreads_and(need == 0u)) break;
                // (the vote makes the exit uniform; the clock is read on the same sweeps by everyone)
                if (!p.wait || ((it & 63u) == 0u &&
                                __syncthreads_or(wall_clock64() - t0 > LS_ARRIVE_TIMEOUT_TICKS))) {
                    complete = false;  // timed out (or, behind a kernel boundary, a granule is missing)
                    break;
                }
            }
            LS_HO(2);
            // planes 0 .. kprime-1 are candidates, plane kprime the bounds; the pivot plane also goes
            // to LDS for the wave that searches the pivot (hist is free until lds_topk)
            const int nc = p.blocks * p.kprime;
            const int pv0 = (m_need - 1) * p.blocks;
#pragma unroll
            for (int j = 0; j < CH; ++j) mine[j] = 0ull;
#pragma unroll
            for (int j = 0; j < CG; ++j) {
                const int i = tid + j * NT;
                const u64 key = complete ? (((u64)gv[j].y << 32) | gv[j].x) : 0ull;
                if (i < nc) mine[j] = key;
                else if (i < G) mb = key > mb ? key : mb;
                // (unconditional store, slot 2047 is nobody's: a guarded one is a branch per granule)
                hist[(prefilter && i >= pv0 && i < pv0 + p.blocks) ? i - pv0 : 2047] = (u32)(key >> 32);
            }
            if (!complete) mb = ~0ull;  // no keys, an unbeatable bound: the proof below fails
            __syncthreads();
            LS_HO(6);
            if (prefilter && wv < NW) {  // (clamped, unconditional reads: slot 2047 holds junk, masked)
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    const int o = ln + 64 * j, b = wv * Q + o;
                    const bool in = o < Q && b < p.blocks;
                    const u32 v = hist[in ? b : 2047];
                    pv[j] = in ? v : 0u;
                }
            }
        } else {
            // (loads return in order: the pivots go out first, so the wave that needs them does not
            // wait for its share of the candidates as well)
            if (prefilter && wv < NW) {
#pragma unroll
                for (int j = 0; j < PW; ++j) {
                    const int o = ln + 64 * j, b = wv * Q + o;
                    const bool in = o < Q && b < p.blocks;
                    const u32 v = (u32)(p.cand[(long long)(in ? b : 0) * p.kprime + (m_need - 1)] >> 32);
                    pv[j] = in ? v : 0u;
                }
            }
            for (int i = tid; i < p.blocks; i += NT) {
                const u64 b = p.bound[i];
                mb = b > mb ? b : mb;
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = tid + j * NT;
                mine[j] = i < mc ? p.cand[i] : 0ull;
            }
        }
        u32* const pw = reinterpret_cast<u32*>(red);  // one answer per searching wave (red[] = 16 free u64)
        if (wv < NW) {
            // No LDS shuffles on the critical path (a ds_bpermute round trip is ~120 cycles and a wave is
            // alone on its SIMD here: the max / min / sum butterflies and the bound reduction of the first
            // version were 3 of the 5.8 us between "every key is here" and "survivors in LDS").
            u32 t0 = 0;
            if (prefilter) {
                const u32 r = (u32)((keff + m_need - 1) / m_need);
                const u32 rw = (r + NW - 1) / NW;  // this wave's share
                // Straight-line on purpose: pv[j] is 0 beyond the slice and 0 counts nowhere, so no step
                // is guarded by "j < slots in use" - the guarded version compiled to two scalar branches
                // per ballot and spent 3.2 us here (tools/handoff_timeline.py).
                // ref: any real pivot; diff: every bit in which some real pivot differs from it
                const u64 has = __ballot(pv[0] != 0u);
                const u32 ref = has ? (u32)__builtin_amdgcn_readlane((int)pv[0], __ffsll((long long)has) - 1) : 0u;
                auto search = [&](auto width) -> u32 {
                    constexpr int W = decltype(width)::value;
                    u32 diff = 0, nnz = 0;
#pragma unroll
                    for (int j = 0; j < W; ++j) {
                        diff |= pv[j] ? (pv[j] ^ ref) : 0u;
                        nnz += (u32)__popcll(__ballot(pv[j] != 0u));
                    }
                    diff = wave_or_dpp(diff);
                    if (!has || nnz < rw) return 0u;
                    if (!diff) return ref;
                    // ANY value with >= rw pivots at or above it will do: the 8 bits below the highest
                    // differing one are resolved (a slightly lower pivot lets a few more keys through)
                    const int hb = 31 - __clz((int)diff);
                    u32 t = hb == 31 ? 0u : (ref >> (hb + 1) << (hb + 1));  // the bits all pivots share
#pragma unroll
                    for (int s8 = 0; s8 < 8; ++s8) {
                        const int bit = hb - s8;
                        const u32 cand = t | (bit >= 0 ? 1u << bit : 0u);  // (below bit 0: cand = t, which passes)
                        u32 c = 0;
#pragma unroll
                        for (int j = 0; j < W; ++j) c += (u32)__popcll(__ballot(pv[j] >= cand));
                        t = c >= rw ? cand : t;
                    }
                    return t;
                };
                if constexpr (PW >= 4)
                    t0 = Q <= 128 ? search(std::integral_constant<int, 2>{}) : search(std::integral_constant<int, PW>{});
                else
                    t0 = search(std::integral_constant<int, PW>{});
                LS_STAMP(7);  // (developer stamp: the pivot loads have landed)
            }
            if (ln == 0) pw[wv] = t0;
        }
        LS_HO(7);
        LS_STAMP(1);
        __syncthreads();  // the pivot and the zeroed survivor counter
        u32 T0 = pw[0];  // the smallest of the waves' answers (0 = no pre-filter)
#pragma unroll
        for (int w = 1; w < NW; ++w) T0 = pw[w] < T0 ? pw[w] : T0;
        for (int c0 = 0;; c0 += CH * NT) {  // survivors -> LDS: ONE LDS atomic per wave and round
            // (one per keeping thread was tried: fine for the ~60 survivors of k = 50, but at k = 1000
            // most of 1024 threads keep something and a thousand atomics on one LDS word took 3.8 us)
            u32 nkeep = 0;
#pragma unroll
            for (int j = 0; j < CH; ++j) nkeep += mine[j] != 0ull && (u32)(mine[j] >> 32) >= T0;
            const u32 inc = wave_incl_scan_dpp(nkeep);  // prefix over the lanes, no LDS crossbar
            const u32 wave_total = (u32)__builtin_amdgcn_readlane((int)inc, 63);
            if (wave_total) {
                u32 base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&misc[7 * 8 + 4], wave_total);
                u32 at = (u32)__builtin_amdgcn_readfirstlane((int)base) + inc - nkeep;
#pragma unroll
                for (int j = 0; j < CH; ++j) {  // (a dropped key goes to tmp[0], a slot nobody reads: no branch per key)
                    const bool keep = mine[j] != 0ull && (u32)(mine[j] >> 32) >= T0;
                    u64* dst = keep ? &keys[at] : &tmp[0];
                    *dst = mine[j];
                    at += keep;
                }
            }
            if (c0 + CH * NT >= mc) break;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = c0 + CH * NT + tid + j * NT;
                mine[j] = i < mc ? p.cand[i] : 0ull;  // (granule jobs fit one round)
            }
        }
        __syncthreads();
        const int nsurv = (int)misc[7 * 8 + 4];  // (lds_topk leaves misc[60..61] alone)
        LS_STAMP(2);
        LS_HO(3);
        nvalid = lds_topk(keys, nsurv, keff, res, tmp, hist, misc, tid, NT);
        LS_HO(4);
        T = (nvalid == keff && keff > 0) ? res[keff - 1] : 0ull;
        // proof: every withheld key is below the k-th best emitted one. mb = the largest bound THIS
        // thread saw; the vote replaces a reduction of the bounds
        done = __syncthreads_and(mb == 0ull || (T != 0ull && mb < T)) != 0;
    }
    if (!done && (p.wait || (!p.S && p.done))) {  // (no S: an ls_mq launch of a synchronous host call, ls_callers.hip)
        // same-launch job: S is not part of the hand-off (see ls_fin_params::gran). Ask the host
        // for the stand-alone finalize behind this launch instead of reading S here.
        if (tid == 0) {  // (the stand-alone finalize counts the slow path in counters[0])
            if (p.done)
                __hip_atomic_store(p.done, p.done_val | LS_DONE_RETRY, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (!done && !p.S) {  // no score vector and nobody to ask for a retry inside the stream: flag the query
        if (tid == 0 && p.repair) {
            __hip_atomic_store(p.repair, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (p.repair_any) __hip_atomic_store(p.repair_any, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    if (!done) {
        if (tid == 0 && p.counters) atomicAdd(&p.counters[0], 1u);
        bool rescued = false;
        if (T != 0ull && !p.force_slow) {
            // rescue: every row with key >= T is a candidate; the answer is among them
            if (tid == 0) misc[5] = 0;
            __syncthreads();
            for (long long r = tid; r < p.n; r += NT) {
                const u64 key = ls_make_key(p.S[r], (u32)r);
                if (key >= T) {
                    const u32 pos = atomicAdd(&misc[5], 1u);
                    if (pos < (u32)p.keys_cap) keys[pos] = key;
                }
            }
            __syncthreads();
            const int c = (int)misc[5];
            __syncthreads();
            if (c <= p.keys_cap) {
                nvalid = lds_topk(keys, c, keff, res, tmp, hist, misc, tid, NT);
                rescued = true;
            }
        }
        if (!rescued) {
            if (tid == 0 && p.counters) atomicAdd(&p.counters[1], 1u);
            nvalid = general_select(p.S, p.n, p.k, res, hist, misc, tid, NT);
        }
    }
    __syncthreads();
    if (p.out_gran) {
        // Host API: tagged result granules in pinned host memory (ls_fin_params::out_gran), one 16-byte
        // system-scope (sc0 sc1) store per result, nothing behind them. No system-scope release fence
        // either: it would write back this XCD's whole L2, which inside a scan launch holds ~100 KB of
        // freshly written score vector nobody is waiting for.
        __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.out_gran, 0, p.k * 16, LS_BUF_RSRC_FLAGS);
        for (int i = tid; i < p.k; i += NT) {
            const u64 key = (i < nvalid) ? res[i] : 0ull;
            const u32 row = key ? 0xffffffffu - (u32)key : 0xffffffffu;
            __builtin_amdgcn_raw_buffer_store_b128(
                u32x4{__builtin_bit_cast(u32, ls_key_score(key)), p.done_val, row, p.done_val}, orsrc, i * 16, 0,
                LS_AUX_SC1 | 1 /* sc0 */);
        }
    } else if (p.done) {
        // Host API, k > LS_OUT_GRAN_MAX_K: the outputs are pinned host rows and the host spins on a
        // completion word. The rows are written THROUGH at system scope, every wave drains them, the
        // workgroup meets and ONE lane publishes - the drained write-through hand-off with the host as
        // the consumer (no system-scope release fence, for the reason above).
        for (int i = tid; i < p.k; i += NT) {
            const u64 key = (i < nvalid) ? res[i] : 0ull;
            __hip_atomic_store(&p.out_scores[i], ls_key_score(key), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(&p.out_indices[i], (long long)ls_key_index(key, p.base), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(p.done, p.done_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        for (int i = tid; i < p.k; i += NT) {
            const u64 key = (i < nvalid) ? res[i] : 0ull;
            p.out_scores[i] = ls_key_score(key);
            p.out_indices[i] = ls_key_index(key, p.base);
        }
    }
    LS_STAMP(6);
    LS_HO(5);
#ifdef LS_HANDOFF_TIMING
    if (tid == 0 && p.counters && p.wait) {  // ticks after this workgroup's entry, + 10000
        p.counters[1] = (u32)(~__hip_atomic_load(&g_ho[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ho[0] + 10000);
        p.counters[2] = (u32)(__hip_atomic_load(&g_ho[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - ho[0] + 10000);
        const int which[5] = {2, 6, 7, 3, 5};  // swept | pivot plane in registers | pivot found | survivors in LDS | done
        for (int i = 0; i < 5; ++i) p.counters[3 + i] = (u32)(ho[which[i]] - ho[0] + 10000);
        __hip_atomic_store(&g_ho[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&g_ho[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#endif
#ifdef LS_FIN_TIMING
    if (tid == 0 && p.counters)
        for (int i = 0; i < 6; ++i) p.counters[2 + i] = (u32)(g_fin_stamp[i + 1] - g_fin_stamp[i]);
    if (tid == 0 && p.counters) p.counters[1] = (u32)(g_fin_stamp[7] - g_fin_stamp[0]);  // (timing builds only)
#endif
}
