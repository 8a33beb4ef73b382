// ls_common.h — shared definitions of libleansearch (gfx950 only).
//
// Result keys. Every candidate row is carried as one 64-bit key
//     key = ord(score) << 32 | (0xffffffff - row)
// where ord() is the order-preserving map float -> uint32. A larger key is a better result
// under the library's total order (score descending, row ascending), so every selection,
// merge and sort below is a plain unsigned 64-bit comparison. key == 0 means "no result"
// (score NaN or <= -FLT_MAX, padding): FAISS's IndexFlat heap never admits such rows either
// (its admission test is `score > heap_top`, heap_top initialised to -FLT_MAX).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cfloat>

#include "../../include/leansearch.h"
#include "../../include/leansearch_debug.h"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

#define LS_WAVE 64
#define LS_CORPUS_PAD_ROWS 128       // zero rows kept behind the stored corpus (whole-tile reads)
#define LS_SCAN_THREADS 256          // 4 waves per scan workgroup
#define LS_SCAN_WAVES (LS_SCAN_THREADS / LS_WAVE)
#define LS_KP_MAX 16                 // per-workgroup emitted candidates (k') + 1 bound
#define LS_MQ_KP_MAX 24              // ... of an ls_mq workgroup (it ranks waves x keys-per-lane of them; c_stride has the room)
#define LS_FINAL_THREADS 1024
#define LS_FINAL_CAP 8192            // keys the finalize workgroup sorts in LDS (64 KiB)
#ifndef LS_SS_MAX_KEYS
#define LS_SS_MAX_KEYS 4096           // lists up to this long are ordered by splitter buckets (64 buckets of <= 256)
#endif
#define LS_SCAN_PATH_MAX_NQ 16            // nq <= this: per-query HBM-bound scan path
#ifndef LS_SCAN_MQ_SCATTER
#define LS_SCAN_MQ_SCATTER 1          // multi-query scan launches: reduce-scatter of the partial sums (0: one butterfly per pair)
#endif
#ifndef LS_SCAN_SMALL
#define LS_SCAN_SMALL 1              // small shards: waves rank their <= 64 keys once instead of inserting row by row
#endif
#ifndef LS_SCAN_SMALL_ROWS
#define LS_SCAN_SMALL_ROWS 64        // ... when no wave sees more rows than this (<= 64: one key per lane)
#endif
#ifndef LS_SCAN_MERGE_FILLED
#define LS_SCAN_MERGE_FILLED 1       // the workgroup merge walks the 4*(k'+1) filled slots instead of all 64
#endif
#ifndef LS_SCAN_SMALL_MAX_BLOCKS
#define LS_SCAN_SMALL_MAX_BLOCKS 256 // ... and the launch has at most one scan workgroup per CU
#endif
#ifndef LS_GEMM_THREADS
#define LS_GEMM_THREADS 512          // batched path: 8 waves per workgroup
#endif
#define LS_GEMM_WAVES (LS_GEMM_THREADS / 64)
#define LS_GEMM_WAVES_PER_SIMD 2     // register budget: 2 waves per SIMD
#ifndef LS_GEMM_TM_SHORT
#define LS_GEMM_TM_SHORT 64          // corpus rows per LDS tile for stored rows <= 1 KiB
#endif
#define LS_GEMM_WG_PER_CU 1          // the tile ring takes most of the LDS: one workgroup per CU
#define LS_GEMM_MAX_K 1024           // batched path handles k <= this (the reference uses 1000)
#define LS_GEMM_MAX_CHUNKS 128       // ... and stored rows <= 2 KiB (d <= 1024 fp16)
#define LS_GEMM_MIN_ROWS 32768       // ... and shards at least this big,
#define LS_GEMM_MIN_ROWS_BIGNQ 8192  // or this big when the batch has >= LS_GEMM_BIGNQ queries
#define LS_GEMM_BIGNQ 128            // (small shards of a many-GPU run: ~60 us fixed vs nq/8 scans)
#define LS_GEMM_QCAP 32              // entries per private candidate queue (a multiple of 4)
#ifndef LS_GEMM_QG2_MAX_CHUNKS
#define LS_GEMM_QG2_MAX_CHUNKS 96    // stored rows up to this many chunks: 2 query groups per wave
#endif
#ifndef LS_GEMM_PF
#define LS_GEMM_PF 3                 // row-blocks-in-sequence loop: A-fragment look-ahead in k-steps
#endif
#ifndef LS_GEMM_STRAIGHT
#define LS_GEMM_STRAIGHT 1           // tile loop without previous-/next-tile branches
#endif
// timing ablations for variant builds (results are wrong with any of them set; never shipped)
#ifndef LS_ABL_NOPASS
#define LS_ABL_NOPASS 0    // the filter compares against +FLT_MAX: no append is ever taken
#endif
#ifndef LS_ABL_NOREPAIR
#define LS_ABL_NOREPAIR 0  // flagged queries are not re-run (keeps the timing of broken variants clean)
#endif
#ifndef LS_ABL_NOBARRIER
#define LS_ABL_NOBARRIER 0
#endif
#ifndef LS_ABL_NODMA
#define LS_ABL_NODMA 0     // no tile DMA after the first tile: the MFMA stream alone
#endif
#ifndef LS_ABL_LDSREADS
#define LS_ABL_LDSREADS 1  // 2 / 4: the two-accumulator pass reads only 1/2, 1/4 of its A fragments from LDS
#endif
#ifndef LS_GEMM_LEAN
#define LS_GEMM_LEAN 0               // 1: every geometry recomputes DMA offsets / queue bases (fewer registers)
#endif
#ifndef LS_GEMM_SAMPLE_TOP2
#define LS_GEMM_SAMPLE_TOP2 1        // 0: the sample pass always keeps four scores per lane
#endif
#ifndef LS_GEMM_APPEND_SC1
#define LS_GEMM_APPEND_SC1 0         // 1: candidate appends written through (sc1) instead of left dirty in L2
#endif
#ifndef LS_GEMM_RING3
#define LS_GEMM_RING3 0              // 1: three tile buffers, DMA two tiles ahead (measured 1-2 % slower than two)
#endif
#define LS_GEMM_SAMPLE_ROWS 128      // sample pass: rows per workgroup
#define LS_GEMM_MAX_SPLITS 512       // corpus slices (4 queues each; the select kernel walks 8 per thread)
#define LS_GEMM32_MIN_NQ 24           // fp32 index: batches at least this big take the f32 MFMA path (past one ls_mq pass: 33)

__host__ __device__ __forceinline__ u32 ls_ord(float f) {
    f = f + 0.0f;  // folds -0.0 into +0.0
    u32 u = __builtin_bit_cast(u32, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float ls_unord(u32 k) {
    u32 u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}
__host__ __device__ __forceinline__ u64 ls_make_key(float s, u32 row) {
    return (s > -FLT_MAX) ? (((u64)ls_ord(s) << 32) | (u64)(0xffffffffu - row)) : 0ull;
}
__host__ __device__ __forceinline__ float ls_key_score(u64 key) {
    return key ? ls_unord((u32)(key >> 32)) : -FLT_MAX;
}
__host__ __device__ __forceinline__ int64_t ls_key_index(u64 key, int64_t base) {
    return key ? base + (int64_t)(0xffffffffu - (u32)(key & 0xffffffffull)) : (int64_t)-1;
}

// ---- one step of a polite busy-wait (host code; ADVICE r5: _mm_pause alone is x86-only) ---------
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
static inline void ls_cpu_relax() { _mm_pause(); }
#elif defined(__aarch64__)
static inline void ls_cpu_relax() { __asm__ __volatile__("yield" ::: "memory"); }
#else
static inline void ls_cpu_relax() {}
#endif

// ---- error plumbing (host) -----------------------------------------------------------------
void ls_set_error(const char* fmt, ...);
#define LS_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            ls_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                         __LINE__);                                                      \
            return LS_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

// hipFuncSetAttribute applies to ONE device: remember per (kernel, device) that it was applied.
struct ls_attr_once {
    std::atomic<unsigned char> done[64];
};
static inline int ls_set_max_dynamic_lds(ls_attr_once& st, const void* fn, int bytes) {
    int dev = 0;
    LS_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !st.done[dev].load(std::memory_order_acquire)) {
        LS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        if (tracked) st.done[dev].store(1, std::memory_order_release);
    }
    return LS_OK;
}

// ---- the library's ONE summation order for faiss.normalize_L2's squared norm ------------------
// (reference search/engine.py:242; FAISS's own SIMD order is not observable here, so it is fixed
// by definition and mirrored by oracle_normalize_l2): lane l of a wave accumulates x[l], x[l+64],
// ... with fused multiply-adds in increasing index; the 64 partial sums are combined by the xor
// butterfly 32, 16, 8, 4, 2, 1 (every lane ends with the same value). Every kernel that
// normalises a query calls this, so a normalised query is bit-identical on every path.
#ifdef __HIPCC__
// v[lane] + v[lane ^ 32], then ^ 16, 8, 4, 2, 1 - the same pairs as a __shfl_xor butterfly (so the same
// bits: an IEEE add is commutative), without its six ds_bpermute round trips through the LDS crossbar
// (~120 cycles each for a wave that waits for nothing else): v_permlane32_swap / v_permlane16_swap
// across the 16-lane rows, DPP inside them (lane ^ 8 = row_ror:8; lane ^ 4 = row_shl:4 into lanes 0-3 and
// 8-11 + row_shr:4 into lanes 4-7 and 12-15; quad_perm for ^ 2 and ^ 1). Whole wave active.
__device__ __forceinline__ float ls_wave_xor_sum(float f) {
    unsigned v = __builtin_bit_cast(unsigned, f);
    auto add = [](unsigned a, unsigned b) -> unsigned {
        return __builtin_bit_cast(unsigned, __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b));
    };
    auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    v = add((unsigned)r[0], (unsigned)r[1]);
    r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = add((unsigned)r[0], (unsigned)r[1]);
    v = add(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, true));  // row_ror:8
    int x4 = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xf, 0x5, false);             // row_shl:4 -> banks 0, 2
    x4 = __builtin_amdgcn_update_dpp(x4, (int)v, 0x114, 0xf, 0xa, false);                // row_shr:4 -> banks 1, 3
    v = add(v, (unsigned)x4);
    v = add(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true));   // lane ^ 2
    v = add(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true));   // lane ^ 1
    return __builtin_bit_cast(float, v);
}
__device__ __forceinline__ float ls_wave_sumsq(const float* __restrict__ x, int d, int lane) {
    float ss = 0.0f;
    for (int j = lane; j < d; j += 64) ss = fmaf(x[j], x[j], ss);
    return ls_wave_xor_sum(ss);
}
#endif

// ---- geometry of the HBM-resident corpus -----------------------------------------------------
// A row is stored as `chunks` 16-byte chunks (4 fp32 or 8 fp16), zero padded so that
// chunks = L * V with L lanes sharing a row and V chunks per lane.
struct ls_geom {
    int32_t d;        // logical dimension
    int32_t d_pad;    // elements per stored row (multiple of 4 / 8)
    int32_t chunks;   // 16-byte chunks per stored row
    int32_t L;        // lanes per row (16, 32 or 64)
    int32_t V;        // chunks per lane (1..4)
    int32_t elem;     // bytes per element (4 or 2)
    int32_t qg4;      // fp16, 48-chunk rows: the batched pass uses the row-split, 64-queries-per-wave shape (ls_gemm.hip RS = 2)
};
int ls_pick_geom(int32_t d, int32_t dtype, ls_geom* g);

// ---- one query's selection job (finalize) -----------------------------------------------------
struct ls_fin_params {
    const float* S;        // score vector the scan wrote, n floats
    long long n;
    const u64* cand;       // blocks * kprime keys emitted by the scan
    const u64* bound;      // blocks bounds (best key each workgroup withheld)
    int blocks, kprime, k;
    int keys_cap;          // LDS key capacity of this launch (>= blocks*kprime for the fast path)
    int force_slow;
    long long base;        // added to every returned row
    float* out_scores;     // [k]
    long long* out_indices;  // [k]
    u32* counters;         // [0] left the fast path, [1] took the general path
    // Host API (synchronous calls that spin instead of sleeping in hipStreamSynchronize), k <= 256:
    // the results go to pinned host memory as k self-validating 16-byte granules {score bits, tag,
    // local row (~0 = none), tag}, one system-scope store each - the host polls the tags (both halves
    // carry one, so the pair is valid however the write is split on its way) and adds the base.
    // Nothing is drained and no completion word follows: the host sees a result one PCIe write after
    // it was stored instead of after write + acknowledge + flag write (-1.5 us per call at k = 50).
    // Larger k keeps the drained rows + completion word: a thousand granules cost the host more to
    // recognise and unpack (+3.5 us at k = 1000) than the drain costs the GPU.
    void* out_gran;        // optional: pinned host memory, k granules (out_scores / out_indices unused)
    u32* done;             // optional: pinned host word; receives done_val once out_scores / out_indices are
    u32 done_val;          //           visible to the host (out_gran null), or done_val | LS_DONE_RETRY when
                           //           a same-launch job must be re-run by the host; the granules' tag
    // Same-launch selection (synchronous host API only): the job rides on the scan launch of ITS
    // OWN query. The hand-off is the data itself: every scan workgroup writes its k' keys and its
    // bound as 16-byte granules {key, tag, 0} with ONE write-through (sc1) store each - no drain, no
    // barrier, no arrival counter - and the selection workgroup sweeps the granule array (sc1
    // loads) until every granule carries this launch's tag. Layout: rank-major,
    // granule[r * blocks + b] = r-th best key of scan workgroup b, plane r = kprime = the bounds.
    // The score vector S is NOT part of the hand-off (plain stores, possibly still dirty in another
    // XCD's L2): if the emitted keys cannot be proven complete (or the sweep times out), the job
    // does not fall back to S inside the launch: it publishes done_val | LS_DONE_RETRY and the host
    // launches the stand-alone finalize behind the scan (a kernel boundary makes S visible), which
    // reads the same granules with wait = 0.
    const void* gran;      // non-null: the candidates are granules (cand / bound unused)
    u32 tag;               // this launch's tag (never 0)
    u32 wait;              // 1: the granules are being written by this very launch: sweep for the tag
    // An ls_mq launch that wrote no score vectors (S == nullptr) for a device-output call: a job whose keys
    // cannot be proven complete raises this device word and leaves its output rows alone; ls_check (or the
    // end of a synchronous call) serves the query again on the scan kernel (ls_api.hip mq_repair).
    u32* repair;
    u32* repair_any;       // pinned host word raised with it: ls_check reads no device memory when nothing was flagged
};
struct ls_out_gran {   // host view of one result granule
    float score;
    u32 tag_lo, row, tag_hi;
};
#define LS_OUT_GRAN_MAX_K 256
#define LS_DONE_RETRY 0x80000000u        // completion word: "run the stand-alone finalize for this query"
#define LS_ARRIVE_TIMEOUT_TICKS 20000000ull  // 200 ms of the 100 MHz clock: give up waiting, ask for a retry
#define LS_GRAN_MAX 4096                 // granules per query: blocks * (kprime + 1) above this -> own launch
#define LS_BUF_RSRC_FLAGS 0x00020000     // gfx950 raw buffer descriptor, dword 3 (32-bit data format)
#define LS_AUX_SC1 16                    // buffer load / store cache policy: write-through / L1 bypass
#define LS_QUERIES_PER_LAUNCH_MAX 32  // (8 per VALU scan launch; 16 or 32 per f32 MFMA small-batch launch, ls_mq.hip)
#define LS_FIN_WG_MAX 16              // selection workgroups riding on one launch: workgroup w runs jobs w, w + 16, ..
// The selection jobs of one launch's queries. They differ by whole-query displacements only (score vector,
// candidate block, output rows, completion word ...), so the batch is ONE job + strides + the list of queries:
// 230 bytes of kernel arguments for up to 32 jobs (32 full parameter sets would be 4.8 KB, past the 4 KB a
// launch may carry). `idx` lets a retry name any subset of the group.
struct ls_fin_batch {
    ls_fin_params p0;        // the job of the group's query 0
    long long S_stride;      // floats between the queries' score vectors
    long long cand_stride;   // keys between their candidate blocks
    long long bound_stride;  // keys between their bound vectors
    long long gran_stride;   // bytes between their granule arrays (same-launch hand-off)
    int njobs;
    unsigned char idx[LS_QUERIES_PER_LAUNCH_MAX];  // job j serves query idx[j] of the group
};
__host__ __device__ __forceinline__ ls_fin_params ls_fin_job(const ls_fin_batch& b, int j) {
    ls_fin_params p = b.p0;
    const long long q = b.idx[j];
    if (p.S) p.S += q * b.S_stride;
    if (p.cand) p.cand += q * b.cand_stride;
    if (p.bound) p.bound += q * b.bound_stride;
    if (p.out_scores) p.out_scores += q * p.k;
    if (p.out_indices) p.out_indices += q * p.k;
    if (p.out_gran) p.out_gran = (char*)p.out_gran + q * p.k * 16;  // (sizeof(ls_out_gran))
    if (p.done) p.done += q;
    if (p.gran) p.gran = (const char*)p.gran + q * b.gran_stride;
    if (p.repair) p.repair += q;
    return p;
}

// ---- kernel launchers (defined in the .hip files) ---------------------------------------------
// prep: q_out[nq, d_pad] = pad(round(normalise(q_in[nq, d]))), fp32
int ls_launch_prep(const float* d_q_in, float* d_q_out, int64_t nq, const ls_geom& g,
                   bool normalize, bool round_f16, hipStream_t s);
// corpus conversion: dst[n, d_pad] (fp32 or fp16) from src fp32 [n, d]
int ls_launch_convert(const float* d_src, void* d_dst, int64_t n, const ls_geom& g,
                      hipStream_t s);
int ls_launch_unconvert(const void* d_src, float* d_dst, int64_t n, const ls_geom& g,
                        hipStream_t s);
// scan: scores S[n] for one RAW query (d floats; normalisation / fp16 rounding fused in)
// + per-workgroup best kprime keys and bound
int ls_scan_blocks(int64_t n, const ls_geom& g, int32_t n_cu);
// One scan launch: `nq` (1, 4 or 8) queries share one pass over the corpus; the launch may carry
// up to LS_QUERIES_PER_LAUNCH_MAX selection jobs of the PREVIOUS launch, executed by up to LS_FIN_WG_MAX extra
// workgroups, so that selection costs neither a launch nor a kernel boundary.
struct ls_scan_args {
    const float* d_q;      // nq raw queries, d floats apart (normalisation / fp16 rounding fused in)
    int nq;                // 1, 4 or 8
    bool normalize, reverse;
    float* d_S;            // score vectors, s_stride floats apart
    long long s_stride;
    u64* d_cand;           // blocks*kprime keys per query, c_stride keys apart
    long long c_stride;
    u64* d_bound;          // blocks bounds per query, b_stride apart
    long long b_stride;
    int blocks, kprime;
    int nfin;              // selection jobs riding on this launch (== fin.njobs)
    ls_fin_batch fin;
    void* d_gran;          // non-null: the jobs are this launch's own and the keys go out as
    long long g_stride;    //           tagged granules (ls_fin_params::gran), g_stride granules per query
    u32 tag;
    int mq_keys;           // ls_launch_mq only: keys every lane keeps (ls_mq_lane_keys)
    float* d_qkeep;        // ls_launch_mq only, optional: the launch copies its nq raw queries there (d floats apart)
};
int ls_launch_scan(const void* d_corpus, int64_t n, const ls_geom& g, const ls_scan_args& a,
                   hipStream_t s);
// Small batches on an fp32 index (ls_mq.hip): a.nq = 2..16 REAL queries share one corpus pass on the f32
// matrix cores, bit-identical to ls_launch_scan's results; same outputs, same riding selection jobs.
#define LS_MQ_MIN_ROWS 4096              // shards below this stay on the VALU scan groups
int ls_mq_blocks(int64_t n, int32_t n_cu, int nq, int chunks);
int ls_mq_lane_keys(int blocks, int keff, int nq);   // 3, 5, 8, or 0 = not for this (k, shard)
int ls_mq_waves(int nq);                             // waves per workgroup of the launch that serves nq queries
int ls_launch_mq(const void* d_corpus, int64_t n, const ls_geom& g, const ls_scan_args& a, hipStream_t s);
// LDS bytes a piggy-backed finalize may use without lowering the scan's occupancy below 2/CU
#define LS_PIGGY_LDS_MAX (72 * 1024)
// finalize: exact top-k from the scan's candidates (or, if they cannot be proven complete,
// from S itself) -> out_scores[k], out_indices[k]. Either its own launch, or carried by the
// NEXT query's scan launch as one extra workgroup (ls_launch_scan's `fin` argument).
int ls_launch_finalize(const struct ls_fin_batch& jobs, hipStream_t s);  // jobs.njobs jobs
// batched MFMA path (ls_gemm.hip)
struct ls_gemm_bufs {
    void* d_queues;     // uint2 [nq_pad][nsplits][4][LS_GEMM_QCAP]: private candidate queues
    u32* d_counts;      // [nq_pad][nsplits][4]
    u32* d_overflow;    // [nq_pad] repair flags of this call
    u32* d_sample_top;  // [nq_pad][nsplits][4][4]
};
int ls_launch_prep_f16(const float* d_q, void* d_qh, float* d_qkeep, int64_t nq, int64_t nq_pad,
                       const ls_geom& g, bool normalize, u32* d_overflow, hipStream_t s);
// a fused launch (this batch's full pass, then the NEXT batch's sample phase; ls_gemm.hip) also needs:
struct ls_gemm_fuse {
    const void* d_qh_next;     // the next batch's prepared queries (same plan as this batch)
    int64_t nq_next;
    u32* d_sample_top_next;    // where the next batch's sample scores go
    int sample_stride;
};

int ls_launch_gemm_filter(const void* d_corpus, int64_t n, const ls_geom& g, const void* d_qh,
                          int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                          int64_t rows_per_split, int tile_stride, const ls_gemm_bufs& b,
                          bool sample_top2, hipStream_t s, const ls_gemm_fuse* fz = nullptr,
                          hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
int ls_launch_tau(const u32* d_sample_top, int nsplits, int64_t nq, int64_t nq_pad, int j_rank,
                  float* d_tau, hipStream_t s);
#ifdef LS_GEMM_TIMING
int ls_gemm_read_sample_stamps(unsigned long long* out, int count);  // variant builds: sample-pass phase stamps
#endif
int ls_gemm_qg(const ls_geom& g);         // query groups of 16 per wave (2, 1 for 2 KiB rows, 4 in the row-split shape)
int ls_gemm_rs(const ls_geom& g);         // wave groups the tile's rows are split over (1, or 2: ls_geom::qg4)
int ls_gemm_qt(const ls_geom& g);         // queries per workgroup
int ls_gemm_tile_rows(const ls_geom& g);  // corpus rows per LDS tile (64, or 32 for long rows)
#define LS_BSEL_MAX_KEYS 8192         // candidate keys per query the select kernel can hold in LDS
int ls_launch_batch_select(const ls_gemm_bufs& b, int nsplits, int64_t nq, int k, int keys_need,
                           int64_t base,
                           int64_t n, int64_t rows_per_split, float* d_out_scores,
                           int64_t* d_out_indices, hipStream_t s);
// one wave per query in <= 48 VGPRs (ls_wsel.hip): co-resident with a running MFMA pass. `done_event`
// (may be null) is attached to the dispatch itself (no extra packet on the stream).
bool ls_wave_select_ok(int nsplits, int k, int keys_need);
int ls_launch_wave_select(const ls_gemm_bufs& b, int nsplits, int64_t nq, int k, int64_t base, int64_t n,
                          int64_t rows_per_split, float* d_out_scores, int64_t* d_out_indices,
                          hipStream_t s, hipEvent_t done_event);
// fp32 batched path (ls_gemm32.hip): exact f32 MFMA, shares tau / select with the fp16 path
int ls_launch_prep_f32(const float* d_q, float* d_qp, float* d_qkeep, int64_t nq, int64_t nq_pad,
                       const ls_geom& g, bool normalize, u32* d_overflow, hipStream_t s);
int ls_launch_gemm32_filter(const void* d_corpus, int64_t n, const ls_geom& g, const float* d_qp,
                            int64_t nq, int64_t nq_pad, const float* d_tau, int nsplits,
                            int64_t rows_per_split, int tile_stride, const ls_gemm_bufs& b,
                            hipStream_t s, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// merge of per-shard lists
int ls_launch_merge(const float* d_scores_in, const int64_t* d_indices_in, int64_t stride_s_bytes,
                    int64_t stride_i_bytes, int32_t n_lists, int64_t nq, int32_t k,
                    float* d_out_scores, int64_t* d_out_indices, hipStream_t s);
