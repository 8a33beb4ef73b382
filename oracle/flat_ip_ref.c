/*
 * oracle/flat_ip_ref.c — TEST INFRASTRUCTURE ONLY. CPU restatement of the dense-retrieval
 * arithmetic of LeanExplore's local backend. Nothing under lean-explore_amd/ may link, load
 * or call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * What it restates
 *   - faiss.normalize_L2(x)            reference src/lean_explore/search/engine.py:242
 *   - index.search(x, k) on an exact inner-product index
 *                                      reference src/lean_explore/search/engine.py:250
 *     (the reference's shipped index is IndexIVFFlat over an IndexFlatIP quantiser,
 *      src/lean_explore/extract/index.py:103-104; BASELINE.json's metric is FAISS-*flat*,
 *      i.e. what IVF approximates, so this file computes S = X * C^T and the per-row top-k)
 *   - the fp32 corpus layout           reference src/lean_explore/extract/index.py:71
 *
 * Where the algorithm really lives: the third-party dependency faiss-cpu, constraint ">=1.7"
 * in the reference's pyproject.toml:36, NOT pinned by any lockfile and NOT present under
 * /root/reference or in this image. Its published algorithm for IndexFlat + METRIC_INNER_PRODUCT
 * is: for every query, fp32 inner product against every stored row, keep the k largest in a
 * heap whose empty slots hold (label -1, distance -FLT_MAX), return them best-first with int64
 * labels. That is what is restated here.
 *
 * PARITY UNPINNED beyond the reference's own tests: the only fixtures the reference holds for
 * this path are the known-answer test tests/extract/index_test.py:186-205 (row 0 = e0, query e0,
 * k = 1 -> label 0) and the structural asserts at :164-183 (ntotal, d); tests/test_oracle.py
 * checks this file against those. FAISS's own summation order and tie behaviour cannot be
 * observed here (no faiss), so they are fixed by definition:
 *   summation  : scores: plain left-to-right fp32, one rounding per multiply and one per add
 *                (compiled with -ffp-contract=off so no FMA contraction sneaks in) - ORDER_STRICT,
 *                the definition; ORDER_SCAN / ORDER_FMA restate the HIP kernels' own fp32 orders
 *                (explicit fmaf calls, so -ffp-contract=off does not touch them) for the
 *                zero-excuse index checks;
 *                squared norm of normalize_L2: see oracle_normalize_l2 below
 *   total order: score descending, then row index ascending
 *   not returned: rows whose score is NaN or <= -FLT_MAX (a FAISS heap never admits them:
 *                 its test is `score > heap_top` with heap_top initialised to -FLT_MAX)
 *   padding    : label -1, score -FLT_MAX
 *
 * fp16 mode restates the build's LS_DTYPE_F16 storage: corpus AND query are rounded to IEEE
 * binary16 (round-to-nearest-even) and widened back to fp32 before the same arithmetic.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- order-preserving key: larger key == better result ------------------------------- */
static inline uint32_t f32_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float bits_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* monotone map float -> uint32 for all non-NaN values */
static inline uint32_t ord_u32(float f) {
    uint32_t u = f32_bits(f + 0.0f); /* +0.0f folds -0 into +0 */
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float unord_u32(uint32_t k) {
    return bits_f32((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
/* composite key: (score desc, row asc). 0 is reserved for "not a result". */
static inline uint64_t make_key(float s, int64_t row) {
    if (!(s > -FLT_MAX)) return 0; /* NaN, -inf and -FLT_MAX are never admitted */
    return ((uint64_t)ord_u32(s) << 32) | (uint64_t)(0xffffffffu - (uint32_t)row);
}

/* ---- IEEE binary16 rounding (gcc 11 has no _Float16 on x86) ---------------------------- */
static inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x = f32_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int32_t exp = (int32_t)((x >> 23) & 0xff);
    if (exp == 0xff) return (uint16_t)(sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0));
    exp = exp - 127 + 15;
    if (exp >= 0x1f) return (uint16_t)(sign | 0x7c00u); /* overflow -> inf */
    if (exp <= 0) {                                     /* subnormal or zero */
        if (exp < -10) return (uint16_t)sign;
        mant |= 0x00800000u;
        uint32_t shift = (uint32_t)(14 - exp);
        uint32_t half = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (half & 1u))) half++;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((uint32_t)exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) half++; /* may carry into exp: ok */
    return (uint16_t)(sign | half);
}
static inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1f;
    uint32_t mant = h & 0x3ffu;
    if (exp == 0) {
        if (mant == 0) return bits_f32(sign);
        float v = (float)mant * 5.9604644775390625e-08f; /* 2^-24 */
        return sign ? -v : v;
    }
    if (exp == 0x1f) return bits_f32(sign | 0x7f800000u | (mant << 13));
    return bits_f32(sign | ((exp - 15 + 127) << 23) | (mant << 13));
}

/* x <- fp32(fp16(x)), elementwise */
void oracle_round_f16(float* x, int64_t count) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < count; ++i) x[i] = f16_bits_to_f32(f32_to_f16_bits(x[i]));
}

/* faiss.normalize_L2 (engine.py:242): x_i *= 1/sqrt(sum x^2); zero-norm rows untouched.
 * FAISS computes the squared norm with SIMD partial sums in an order that cannot be observed
 * here, so the order is fixed by definition, the same one every HIP kernel of the build uses
 * (ls_wave_sumsq, lean-explore_amd/csrc/ls_common.h): 64 interleaved partial sums
 * p[l] = fma(x[j], x[j], p[l]) over j = l, l+64, ... in increasing j, combined by the balanced
 * xor tree with strides 32, 16, 8, 4, 2, 1. Then ONE correctly rounded 1/sqrt and one multiply
 * per element: a normalised query is bit-identical between this file and the HIP path. */
void oracle_normalize_l2(float* x, int64_t nq, int32_t d) {
    for (int64_t i = 0; i < nq; ++i) {
        float* r = x + i * (int64_t)d;
        float p[64];
        for (int l = 0; l < 64; ++l) p[l] = 0.0f;
        for (int32_t j = 0; j < d; ++j) p[j & 63] = fmaf(r[j], r[j], p[j & 63]);
        for (int o = 32; o >= 1; o >>= 1)
            for (int l = 0; l < o; ++l) p[l] = p[l] + p[l ^ o];
        const float nr = p[0];
        if (nr > 0.0f) {
            float inv = 1.0f / sqrtf(nr);
            for (int32_t j = 0; j < d; ++j) r[j] *= inv;
        }
    }
}

/* one inner product, left-to-right fp32 */
static inline float dot_f32(const float* a, const float* b, int32_t d) {
    float acc = 0.0f;
    for (int32_t j = 0; j < d; ++j) acc += a[j] * b[j];
    return acc;
}

/* ---- the HIP kernels' own fp32 summation orders (parity modes; see the header of oracle.py) ----
 * The strict left-to-right order above is the DEFINITION the 1e-5 score bound is checked against.
 * The fp32 kernels sum in two other, fully documented orders; restating them lets the parity tests
 * demand bit-identical scores AND indices (no near-tie excuse) on continuous data:
 *
 * ORDER_SCAN (lean-explore_amd/csrc/ls_scan.hip:66-77 QueryRegs::dot, :129-145 group_sum; the
 * multi-query reduce-scatter :158-186 and the f32 MFMA small-batch kernel ls_mq.hip add the same
 * operands in the same order): a stored row is `chunks` 16-byte chunks (4 floats), chunks = L*V
 * (ls_pick_geom, csrc/ls_prep.hip:123-142, restated in geom_f32 below; zero padded). Lane `sub` of the
 * L lanes sharing a row runs ONE fmaf chain, starting from +0, over its chunks sub, sub+L, ..,
 * sub+(V-1)L in that order, the four floats of a chunk in memory order:
 *     p[sub] = fmaf(x[e], q[e], p[sub]),  e = 4*(sub + L*v) + 0..3,  v = 0..V-1
 * and the L partial sums are combined by the balanced xor tree with strides 1, 2, 4, .., L/2
 * (p[i] = p[i] + p[i ^ o]; fp32 addition is commutative, so every lane holds the same value).
 *
 * ORDER_FMA (lean-explore_amd/csrc/ls_gemm32.hip: v_mfma_f32_16x16x4_f32 over k = 0..d_pad-1; measured
 * on gfx950 by tools/arith_probe.hip to be bit for bit a sequential fmaf chain in increasing k):
 *     acc = fmaf(x[k], q[k], acc),  k = 0..d-1, starting from +0.
 * (zero padding adds fmaf(0, 0, acc) = acc in both orders and is skipped here.)
 */
enum { ORDER_STRICT = 0, ORDER_SCAN = 1, ORDER_FMA = 2 };

/* fp32 row geometry: chunks padded up to one of the supported sizes, L lanes x V chunks per lane */
static int geom_f32(int32_t d, int* L, int* V) {
    static const int sizes[8][3] = {{16, 16, 1}, {32, 16, 2}, {48, 16, 3},  {64, 16, 4},
                                    {96, 32, 3}, {128, 32, 4}, {192, 64, 3}, {256, 64, 4}};
    const int raw = (d + 3) / 4;
    for (int i = 0; i < 8; ++i)
        if (sizes[i][0] >= raw) { *L = sizes[i][1]; *V = sizes[i][2]; return 0; }
    return -1;
}
int oracle_geom_f32(int32_t d, int32_t* L, int32_t* V) {
    int l = 0, v = 0;
    if (geom_f32(d, &l, &v)) return -1;
    *L = l; *V = v;
    return 0;
}

static inline float dot_scan_order(const float* a, const float* b, int32_t d, int L, int V) {
    float p[64];
    for (int sub = 0; sub < L; ++sub) {
        float acc = 0.0f;
        for (int v = 0; v < V; ++v) {
            const int32_t e0 = 4 * (sub + L * v);
            for (int j = 0; j < 4; ++j) {
                const int32_t e = e0 + j;
                const float x = e < d ? a[e] : 0.0f, y = e < d ? b[e] : 0.0f;
                acc = fmaf(x, y, acc);
            }
        }
        p[sub] = acc;
    }
    for (int o = 1; o < L; o <<= 1)
        for (int i = 0; i < L; i += 2 * o) p[i] = p[i] + p[i + o];
    return p[0];
}
static inline float dot_fma_order(const float* a, const float* b, int32_t d) {
    float acc = 0.0f;
    for (int32_t j = 0; j < d; ++j) acc = fmaf(a[j], b[j], acc);
    return acc;
}
static inline float dot_order(const float* a, const float* b, int32_t d, int order, int L, int V) {
    if (order == ORDER_SCAN) return dot_scan_order(a, b, d, L, V);
    if (order == ORDER_FMA) return dot_fma_order(a, b, d);
    return dot_f32(a, b, d);
}

/* all scores of one query: out[r] = <corpus[r], q> */
void oracle_scores(const float* corpus, int64_t n, int32_t d, const float* q, float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) out[r] = dot_f32(corpus + r * (int64_t)d, q, d);
}

/* ---- k largest keys with a binary min-heap -------------------------------------------- */
static void heap_sift_down(uint64_t* h, int32_t n, int32_t i) {
    for (;;) {
        int32_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && h[l] < h[m]) m = l;
        if (r < n && h[r] < h[m]) m = r;
        if (m == i) return;
        uint64_t t = h[i]; h[i] = h[m]; h[m] = t;
        i = m;
    }
}
static int cmp_key_desc(const void* a, const void* b) {
    uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x < y) - (x > y);
}

/* select the k best of scores[0..n) into (D, I) best-first; pads with (-FLT_MAX, -1) */
static void select_topk(const float* scores, int64_t n, int32_t k, int64_t base, float* D,
                        int64_t* I, uint64_t* heap) {
    int32_t hn = 0;
    for (int64_t r = 0; r < n; ++r) {
        uint64_t key = make_key(scores[r], r);
        if (key == 0) continue;
        if (hn < k) {
            heap[hn++] = key;
            if (hn == k)
                for (int32_t i = k / 2 - 1; i >= 0; --i) heap_sift_down(heap, k, i);
        } else if (key > heap[0]) {
            heap[0] = key;
            heap_sift_down(heap, k, 0);
        }
    }
    qsort(heap, (size_t)hn, sizeof(uint64_t), cmp_key_desc);
    for (int32_t i = 0; i < k; ++i) {
        if (i < hn) {
            D[i] = unord_u32((uint32_t)(heap[i] >> 32));
            I[i] = base + (int64_t)(0xffffffffu - (uint32_t)(heap[i] & 0xffffffffu));
        } else {
            D[i] = -FLT_MAX;
            I[i] = -1;
        }
    }
}

/*
 * index.search(x, k) for an exact inner-product index (engine.py:250).
 *   corpus f32 [n, d] row-major, q f32 [nq, d]; D f32 [nq, k]; I i64 [nq, k].
 * Parallelism is over rows for one query and over queries for a batch; neither changes any
 * result because every score is one sequential dot product.
 * Returns 0, or -1 on bad arguments / allocation failure.
 */
int oracle_flat_ip_topk_order(const float* corpus, int64_t n, int32_t d, const float* q, int64_t nq,
                              int32_t k, int64_t base, int32_t order, float* D, int64_t* I) {
    if (n < 0 || d <= 0 || nq < 0 || k <= 0 || (n > 0 && !corpus) || (nq > 0 && (!q || !D || !I)))
        return -1;
    if (order < ORDER_STRICT || order > ORDER_FMA) return -1;
    int L = 0, V = 0;
    if (order == ORDER_SCAN && geom_f32(d, &L, &V)) return -1;
    if (nq == 0) return 0;
    int fail = 0;
    if (nq == 1) {
        float* s = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        uint64_t* heap = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)k);
        if (!s || !heap) { free(s); free(heap); return -1; }
#pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < n; ++r) s[r] = dot_order(corpus + r * (int64_t)d, q, d, order, L, V);
        select_topk(s, n, k, base, D, I, heap);
        free(s); free(heap);
        return 0;
    }
#pragma omp parallel
    {
        float* s = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
        uint64_t* heap = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)k);
        if (!s || !heap) {
#pragma omp atomic write
            fail = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t i = 0; i < nq; ++i) {
                const float* qi = q + i * (int64_t)d;
                for (int64_t r = 0; r < n; ++r) s[r] = dot_order(corpus + r * (int64_t)d, qi, d, order, L, V);
                select_topk(s, n, k, base, D + i * (int64_t)k, I + i * (int64_t)k, heap);
            }
        }
        free(s); free(heap);
    }
    return fail ? -1 : 0;
}
int oracle_flat_ip_topk(const float* corpus, int64_t n, int32_t d, const float* q, int64_t nq,
                        int32_t k, int64_t base, float* D, int64_t* I) {
    return oracle_flat_ip_topk_order(corpus, n, d, q, nq, k, base, ORDER_STRICT, D, I);
}
/* all scores of one query in a given order (tests: the score vector S the scan kernels write) */
int oracle_scores_order(const float* corpus, int64_t n, int32_t d, const float* q, int32_t order, float* out) {
    int L = 0, V = 0;
    if (order < ORDER_STRICT || order > ORDER_FMA) return -1;
    if (order == ORDER_SCAN && geom_f32(d, &L, &V)) return -1;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) out[r] = dot_order(corpus + r * (int64_t)d, q, d, order, L, V);
    return 0;
}

/*
 * G-way merge of per-shard results under the same total order: inputs
 * D_in f32 [g, nq, k], I_in i64 [g, nq, k] with -1 padding; output the k best per query.
 */
int oracle_merge_topk(const float* D_in, const int64_t* I_in, int32_t g, int64_t nq, int32_t k,
                      float* D, int64_t* I) {
    if (g <= 0 || nq < 0 || k <= 0) return -1;
    int64_t m = (int64_t)g * k;
    typedef struct { float s; int64_t i; } ent;
    ent* buf = (ent*)malloc(sizeof(ent) * (size_t)m);
    if (!buf) return -1;
    for (int64_t qi = 0; qi < nq; ++qi) {
        int64_t c = 0;
        for (int32_t s = 0; s < g; ++s)
            for (int32_t j = 0; j < k; ++j) {
                int64_t off = ((int64_t)s * nq + qi) * k + j;
                if (I_in[off] >= 0) { buf[c].s = D_in[off]; buf[c].i = I_in[off]; ++c; }
            }
        /* insertion sort by (score desc, index asc): m is small */
        for (int64_t a = 1; a < c; ++a) {
            ent e = buf[a];
            int64_t b = a - 1;
            while (b >= 0 && (buf[b].s < e.s || (buf[b].s == e.s && buf[b].i > e.i))) {
                buf[b + 1] = buf[b];
                --b;
            }
            buf[b + 1] = e;
        }
        for (int32_t j = 0; j < k; ++j) {
            if (j < c) { D[qi * k + j] = buf[j].s; I[qi * k + j] = buf[j].i; }
            else { D[qi * k + j] = -FLT_MAX; I[qi * k + j] = -1; }
        }
    }
    free(buf);
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
