// ls_bm25.hip — lexical name retrieval on the GPU: the eager-sparse BM25+
// scoring of bm25s (reference src/lean_explore/search/engine.py:192-223 calls
// `bm25.retrieve([tokens], k=1000)`; indices built at src/lean_explore/extract/index.py:238-266).
//
// bm25s keeps a CSC matrix, one column per vocabulary token (rows = documents containing the
// token, data = idf * tf-part - nonoccurrence, float32), and scores a query by adding its tokens'
// columns into a zero vector in token order. The index arrives in that layout (the reference's
// on-disk files) and is transposed ONCE at creation into a document-major copy in HBM:
//     doc_ptr[n_docs + 1]  (u32)      entries[nnz] = (token id, value) pairs, 8 B each
// One launch then scores every document: a lane owns a document, walks the query's tokens IN
// QUERY ORDER and adds the value of each token the document holds (a document holds a token at
// most once; a token repeated in the query is added again, as bm25s does) — the same float32
// additions in the same order as bm25s's numpy loop, hence bit-identical scores — adds the
// query's non-occurrence sum and feeds the same per-workgroup candidates + bound + finalize_body
// selection as the dense path. Two launches per query (score + select) instead of one per token
// plus two.
//
// HBM-bound integer/byte work: bytes per query = 4 B * (n_docs + 1) + 8 B * nnz (the whole index
// once) + 4 B * n_docs (final scores, kept for the selection's rescue path): 13 MB for 200 k names
// - a latency-bound problem at this size, which is why the launch count is what matters.
#include "ls_select_dev.h"


#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

struct ls_bm25 {
    int32_t device = 0, n_cu = 256;
    int64_t n_docs = 0, n_vocab = 0, nnz = 0;
    std::vector<int64_t> h_indptr;   // host copy: column extents are launch parameters
    std::vector<float> h_nonocc;
    u32* d_doc_ptr = nullptr;  // document-major copy: entries of document r are [doc_ptr[r], doc_ptr[r+1])
    uint2* d_entries = nullptr;  // (token id, float bits)
    float* d_F = nullptr;   // scores of the last search (partial sums between the launches of a
                            // > LS_BM25_QTOK-token query; the finalize step's rescue path reads it)
    ls_out_gran* h_out_g = nullptr;  // pinned, device-visible: the finalize step writes its result granules here
    u32 out_seq = 0;                 // (ls_fin_params::out_gran) and the host polls their tags
    u64* d_cand = nullptr;
    u64* d_bound = nullptr;
    u32* d_counters = nullptr;
    int32_t blocks = 1;
    hipStream_t stream = nullptr;
    std::mutex mu;
};

#define LS_BM25_QTOK 32  // query tokens per launch (kernel arguments; longer queries chain launches)
#define LS_BM25_REG 8    // document entries kept in registers (names have a handful of tokens)
struct ls_bm25_query {
    int32_t tok[LS_BM25_QTOK];
};

// One pass over the document-major index. first: partial sums start at 0 (else at F[r]); last: add
// the non-occurrence shift, store the final score and emit this workgroup's best kprime keys +
// bound, exactly what the dense scan emits, so finalize_body can prove / complete the top-k.
//
// Row -> workgroup mapping. BM25 scores are discrete (few distinct tf / length combinations), so
// the k-th score is usually shared by thousands of documents and the total order picks the ones
// with the LOWEST row numbers: with 64-row tiles dealt to waves those all sit in a few workgroups,
// each of which may emit only k' keys, the selection's proof fails and every query pays the rescue
// sweep. Rows are therefore dealt in granules of 4 (rows 4j .. 4j+3) so that ANY run of low rows
// spreads evenly over all B workgroups (a lane quad reads 16 contiguous bytes of doc_ptr).
// Round 5: WHICH workgroup gets granule j is XCD-aware. Round 3/4 dealt granule j to workgroup j mod B;
// workgroups go round-robin over the 8 XCDs (blockIdx % 8), so the 8 granules of every 128-byte line
// of doc_ptr - and the entries runs they point at - were fetched by 8 different L2s: 19.8 MB of HBM
// traffic per launch for 7.4 MB of data (profiles/pmc_bm25.json, round 4). Now runs of 8 consecutive
// granules (32 rows = one doc_ptr line) go to ONE XCD (run r -> XCD r mod 8) and are dealt there to that
// XCD's workgroups (blockIdx = xcd + 8 m) round-robin in run order: the tie-spreading is unchanged (any
// prefix of the rows is spread over all B workgroups to within one granule), every line has one L2.
#define LS_BM25_U 4  // documents per lane in flight
#ifndef LS_BM25_ABL
#define LS_BM25_ABL 0  // timing ablations (wrong results): 1 no candidate emission, 2 also no entry loads
#endif
__global__ __launch_bounds__(256) void bm25_score_kernel(
    const u32* __restrict__ doc_ptr, const uint2* __restrict__ entries, long long n,
    ls_bm25_query q, int ntok, int first, int last, float shift, float* __restrict__ F,
    u64* __restrict__ cand, u64* __restrict__ bound, int kprime) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kp = kprime + 1;
    const long long B = gridDim.x;
    const long long NG = (n + 3) / 4;                      // granules
    const bool xcd_aware = (B & 7) == 0;                   // (the host launches a multiple of 8 workgroups)
    const long long Bx = xcd_aware ? B >> 3 : B;           // workgroups that share this one's granule sequence
    const long long xcd = blockIdx.x & 7, m = xcd_aware ? blockIdx.x >> 3 : blockIdx.x;
    // granules of one sequence: an XCD owns every 8th run of 8 granules
    const long long NGx = xcd_aware ? ((NG + 63) / 64) * 8 : NG;
    const long long steps = (NGx + 64 * Bx - 1) / (64 * Bx);  // 64 granules per workgroup and step
    u64 lst = 0, thr = 0;
    for (long long s0 = 0; s0 < steps; s0 += LS_BM25_U) {
        long long row[LS_BM25_U];
        u32 a[LS_BM25_U], b[LS_BM25_U];
        float sc[LS_BM25_U];
        bool valid[LS_BM25_U];
#pragma unroll
        for (int u = 0; u < LS_BM25_U; ++u) {
            const long long l = m + Bx * ((s0 + u) * 64 + (threadIdx.x >> 2));  // number in the sequence
            const long long j = xcd_aware ? ((l >> 3) * 8 + xcd) * 8 + (l & 7) : l;
            row[u] = 4 * j + (threadIdx.x & 3);
            valid[u] = (s0 + u) < steps && row[u] < n;
            a[u] = b[u] = 0;
            if (valid[u]) {
                a[u] = doc_ptr[row[u]];
                b[u] = doc_ptr[row[u] + 1];
            }
        }
        uint2 e[LS_BM25_U][LS_BM25_REG];
#pragma unroll
        for (int u = 0; u < LS_BM25_U; ++u) {
            sc[u] = (valid[u] && !first) ? F[row[u]] : 0.0f;
#pragma unroll
            for (int j = 0; j < LS_BM25_REG; ++j)
                e[u][j] = (LS_BM25_ABL < 2 && a[u] + j < b[u]) ? entries[a[u] + j] : make_uint2(0xffffffffu, 0u);
        }
#pragma unroll
        for (int u = 0; u < LS_BM25_U; ++u) {
            float s = sc[u];
            for (int qi = 0; qi < ntok; ++qi) {  // query order: the float32 sum is order-sensitive
                const u32 tok = (u32)q.tok[qi];
                bool hit = false;
                u32 bits = 0;
#pragma unroll
                for (int j = 0; j < LS_BM25_REG; ++j)
                    if (e[u][j].x == tok) {
                        hit = true;
                        bits = e[u][j].y;
                    }
                for (u32 j = a[u] + LS_BM25_REG; j < b[u]; ++j) {  // long documents: the tail from memory
                    const uint2 x = entries[j];
                    if (x.x == tok) {
                        hit = true;
                        bits = x.y;
                    }
                }
                if (hit) s = s + __uint_as_float(bits);
            }
            if (last) s = s + shift;
            if (valid[u]) F[row[u]] = s;
            sc[u] = s;
        }
        if (!last || LS_BM25_ABL >= 1) continue;
#pragma unroll
        for (int u = 0; u < LS_BM25_U; ++u) {
            const u64 key = valid[u] ? ls_make_key(sc[u], (u32)row[u]) : 0ull;
            u64 mask = __ballot(key > thr);
            while (mask) {
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                const u64 v = readlane64(key, j);
                if (v <= thr) continue;  // the ballot is older than the threshold: most of a first tile
                wave_insert(lst, v, lane, kp);
                thr = readlane64(lst, kp - 1);
            }
        }
    }
    if (!last) return;
    __shared__ u64 sm[4 * LS_KP_MAX];
    if (lane < LS_KP_MAX) sm[wave * LS_KP_MAX + lane] = (lane < kp) ? lst : 0ull;
    __syncthreads();
    if (wave == 0) {
        const u64 mine = sm[lane];
        int rank = 0;
#pragma unroll 8
        for (int i = 0; i < 4 * LS_KP_MAX; ++i) {
            const u64 o = sm[i];
            rank += (o > mine) || (o == mine && i < lane);
        }
        if (rank < kprime) cand[(long long)blockIdx.x * kprime + rank] = mine;
        if (rank == kprime) bound[blockIdx.x] = mine;
    }
}

extern "C" {

void ls_bm25_destroy(ls_bm25* ix) {
    if (!ix) return;
    (void)hipSetDevice(ix->device);
    if (ix->stream) (void)hipStreamSynchronize(ix->stream);
    (void)hipFree(ix->d_doc_ptr);
    (void)hipFree(ix->d_entries);
    (void)hipFree(ix->d_F);
    if (ix->h_out_g) (void)hipHostFree(ix->h_out_g);
    (void)hipFree(ix->d_cand);
    (void)hipFree(ix->d_bound);
    (void)hipFree(ix->d_counters);
    if (ix->stream) (void)hipStreamDestroy(ix->stream);
    delete ix;
}

int ls_bm25_create(ls_bm25** out, const int64_t* indptr, const int32_t* indices, const float* data,
                   const float* nonocc, int64_t n_docs, int64_t n_vocab, int32_t device) {
    if (!out) return LS_ERR_INVALID_ARG;
    *out = nullptr;
    if (n_docs < 0 || n_vocab < 0 || !indptr || (n_vocab > 0 && !nonocc) || n_docs >= 0xffffffffll) {
        ls_set_error("ls_bm25_create: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    const int64_t nnz = indptr[n_vocab];
    if (nnz < 0 || (nnz > 0 && (!indices || !data))) {
        ls_set_error("ls_bm25_create: bad CSC arrays");
        return LS_ERR_INVALID_ARG;
    }
    for (int64_t t = 0; t < n_vocab; ++t)
        if (indptr[t + 1] < indptr[t]) {
            ls_set_error("ls_bm25_create: indptr is not monotone");
            return LS_ERR_INVALID_ARG;
        }
    for (int64_t i = 0; i < nnz; ++i)
        if (indices[i] < 0 || indices[i] >= n_docs) {
            ls_set_error("ls_bm25_create: document index out of range");
            return LS_ERR_INVALID_ARG;
        }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) {
        ls_set_error("no HIP device available; libleansearch has no CPU path");
        return LS_ERR_NO_DEVICE;
    }
    LS_HIP(hipSetDevice(device));
    ls_bm25* ix = new (std::nothrow) ls_bm25();
    if (!ix) return LS_ERR_INVALID_ARG;
    ix->device = device;
    ix->n_docs = n_docs;
    ix->n_vocab = n_vocab;
    ix->nnz = nnz;
    ix->h_indptr.assign(indptr, indptr + n_vocab + 1);
    ix->h_nonocc.assign(nonocc, nonocc + n_vocab);
    int cu = 0;
    if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cu > 0)
        ix->n_cu = cu;
    // one workgroup per CU: with k' <= 15 keys each the selection step sees <= 4 k candidate keys
    ix->blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n_docs + 255) / 256, ix->n_cu));
    if (ix->blocks > 8) ix->blocks &= ~7;  // a multiple of the 8 XCDs: the score kernel's XCD-aware row deal
    auto fail = [&](const char* what) {
        ls_set_error("ls_bm25_create: %s failed", what);
        ls_bm25_destroy(ix);
        return LS_ERR_HIP;
    };
    if (nnz >= 0xffffffffll) {
        ls_set_error("ls_bm25_create: %lld postings exceed the 32-bit document offsets", (long long)nnz);
        ls_bm25_destroy(ix);
        return LS_ERR_INVALID_ARG;
    }
    // transpose CSC -> document-major (counting sort on the document id; columns are visited in
    // token order, so a document's entries end up sorted by token id)
    std::vector<u32> doc_ptr((size_t)n_docs + 1, 0u);
    for (int64_t i = 0; i < nnz; ++i) doc_ptr[(size_t)indices[i] + 1]++;
    for (int64_t r = 0; r < n_docs; ++r) doc_ptr[(size_t)r + 1] += doc_ptr[(size_t)r];
    std::vector<uint2> entries((size_t)std::max<int64_t>(nnz, 1));
    {
        std::vector<u32> fill(doc_ptr.begin(), doc_ptr.end() - 1);
        for (int64_t t = 0; t < n_vocab; ++t)
            for (int64_t i = indptr[t]; i < indptr[t + 1]; ++i) {
                const size_t r = (size_t)indices[i];
                u32 bits;
                memcpy(&bits, &data[i], 4);
                if (fill[r] > doc_ptr[r] && entries[fill[r] - 1].x == (u32)t) {
                    ls_set_error("ls_bm25_create: column %lld lists document %zu twice", (long long)t, r);
                    ls_bm25_destroy(ix);
                    return LS_ERR_INVALID_ARG;
                }
                entries[fill[r]++] = make_uint2((u32)t, bits);
            }
    }
    const size_t nz = (size_t)std::max<int64_t>(nnz, 1), nd = (size_t)std::max<int64_t>(n_docs, 1);
    if (hipMalloc((void**)&ix->d_doc_ptr, (nd + 1) * 4) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_entries, nz * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_F, nd * 4) != hipSuccess) return fail("hipMalloc");
    if (hipHostMalloc((void**)&ix->h_out_g, LS_MAX_K * sizeof(ls_out_gran), hipHostMallocDefault) != hipSuccess)
        return fail("hipHostMalloc");
    memset(ix->h_out_g, 0, LS_MAX_K * sizeof(ls_out_gran));
    if (hipMalloc((void**)&ix->d_cand, (size_t)ix->blocks * LS_KP_MAX * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_bound, (size_t)ix->blocks * 8) != hipSuccess) return fail("hipMalloc");
    if (hipMalloc((void**)&ix->d_counters, 32) != hipSuccess) return fail("hipMalloc");
    if (hipMemset(ix->d_counters, 0, 32) != hipSuccess) return fail("hipMemset");
    if (hipMemcpy(ix->d_doc_ptr, doc_ptr.data(), ((size_t)n_docs + 1) * 4, hipMemcpyHostToDevice) !=
        hipSuccess)
        return fail("upload");
    if (nnz > 0 && hipMemcpy(ix->d_entries, entries.data(), (size_t)nnz * 8, hipMemcpyHostToDevice) !=
                       hipSuccess)
        return fail("upload");
    if (hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking) != hipSuccess) return fail("stream");
    *out = ix;
    return LS_OK;
}

int64_t ls_bm25_ntotal(const ls_bm25* ix) { return ix ? ix->n_docs : -1; }

// counter 0: searches whose selection step left the fast path (rescue sweep or general select);
// counter 1: those that took the general path.
int64_t ls_bm25_debug_counter(ls_bm25* ix, int32_t which) {
    if (!ix || which < 0 || which > 7) return -1;  // 2..6: phase ticks of a -DLS_FIN_TIMING build
    std::lock_guard<std::mutex> lk(ix->mu);
    if (hipSetDevice(ix->device) != hipSuccess) return -1;
    u32 v = 0;
    if (hipMemcpy(&v, ix->d_counters + which, sizeof(u32), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int64_t)v;
}

// token_ids: host int32 [n_tokens], ids of the query's tokens in query order (duplicates count
// twice, like bm25s); out_scores host f32 [k], out_docs host i64 [k], (-FLT_MAX, -1) padded.
int ls_bm25_search(ls_bm25* ix, const int32_t* token_ids, int32_t n_tokens, int32_t k,
                   float* out_scores, int64_t* out_docs) {
    if (!ix || n_tokens < 0 || k <= 0 || (n_tokens > 0 && !token_ids) || !out_scores || !out_docs) {
        ls_set_error("ls_bm25_search: bad argument");
        return LS_ERR_INVALID_ARG;
    }
    if (std::min<int64_t>(k, ix->n_docs) > LS_MAX_K) {
        ls_set_error("ls_bm25_search: min(k, n_docs) exceeds LS_MAX_K");
        return LS_ERR_K_TOO_LARGE;
    }
    for (int i = 0; i < n_tokens; ++i)
        if (token_ids[i] < 0 || token_ids[i] >= ix->n_vocab) {
            ls_set_error("ls_bm25_search: token id out of range");
            return LS_ERR_INVALID_ARG;
        }
    std::lock_guard<std::mutex> lk(ix->mu);
    LS_HIP(hipSetDevice(ix->device));
    hipStream_t s = ix->stream;
    const long long n = ix->n_docs;
    // No query token has a posting (e.g. the raw-token index asked about a multi-word query, whose
    // single token is never a name): every document scores exactly `shift`, so under the total
    // order the answer is documents 0..k-1. The GPU path would produce the same bits the slow way
    // (200k tied keys defeat the candidate proof), so this one case is answered right here.
    int64_t postings = 0;
    for (int i = 0; i < n_tokens; ++i)
        postings += ix->h_indptr[token_ids[i] + 1] - ix->h_indptr[token_ids[i]];
    if (n > 0 && postings == 0) {
        float shift = 0.0f;
        for (int i = 0; i < n_tokens; ++i) shift = shift + ix->h_nonocc[token_ids[i]];
        shift = 0.0f + shift;              // what the sweep computes: S[row] (= 0) + shift
        const bool ok = shift > -FLT_MAX;  // (NaN / -inf rows are never returned)
        for (int i = 0; i < k; ++i) {
            const bool live = ok && i < n;
            out_scores[i] = live ? shift : -FLT_MAX;
            out_docs[i] = live ? i : -1;
        }
        return LS_OK;
    }
    if (n > 0) {
        float shift = 0.0f;
        for (int i = 0; i < n_tokens; ++i)
            shift = shift + ix->h_nonocc[token_ids[i]];  // float32, query order (as bm25s sums)
        const int keff = (int)std::min<int64_t>(k, n);
        const double lam = (double)keff / ix->blocks;
        int kprime = (int)(lam + 5.0 * __builtin_sqrt(lam) + 3.0);
        kprime = std::max(2, std::min(kprime, LS_KP_MAX - 1));
        // LS_BM25_QTOK tokens per launch (kernel arguments); only the last launch of a longer
        // query adds the shift and emits candidates, the ones before leave partial sums in F
        for (int t0 = 0; t0 < n_tokens || t0 == 0; t0 += LS_BM25_QTOK) {
            ls_bm25_query q;
            const int m = std::max(0, std::min(LS_BM25_QTOK, n_tokens - t0));
            for (int i = 0; i < LS_BM25_QTOK; ++i) q.tok[i] = i < m ? token_ids[t0 + i] : -1;
            const int last = t0 + LS_BM25_QTOK >= n_tokens;
            hipLaunchKernelGGL(bm25_score_kernel, dim3(ix->blocks), dim3(256), 0, s, ix->d_doc_ptr,
                               ix->d_entries, n, q, m, (int)(t0 == 0), last, shift, ix->d_F, ix->d_cand,
                               ix->d_bound, kprime);
        }
        LS_HIP(hipGetLastError());
        ls_fin_batch jobs{};
        ls_fin_params& p = jobs.p0;
        jobs.njobs = 1;
        p.S = ix->d_F;
        p.n = n;
        p.cand = ix->d_cand;
        p.bound = ix->d_bound;
        p.blocks = ix->blocks;
        p.kprime = kprime;
        p.k = k;
        p.keys_cap = LS_FINAL_CAP;
        p.force_slow = 0;
        p.base = 0;
        // results: tagged granules in pinned host memory, written by the kernel over PCIe; the host
        // polls the tags instead of sleeping in hipStreamSynchronize (its wake-up alone costs more than
        // the selection kernel) and falls back to the stream sync after 2 ms
        // (the name indices are always asked for their top 1000: the granule form's host cost at that k
        // is paid here too, but it replaces hipStreamSynchronize, not a completion word)
        if (++ix->out_seq >= LS_DONE_RETRY) ix->out_seq = 1;
        p.out_gran = ix->h_out_g;
        p.done_val = ix->out_seq;
        p.counters = ix->d_counters;
        if (k > LS_MAX_K) {  // only possible when k > n_docs: select LS_MAX_K >= n_docs, pad on host
            p.k = LS_MAX_K;
        }
        int rc = ls_launch_finalize(jobs, s);
        if (rc != LS_OK) return rc;
        const int kk = std::min(k, LS_MAX_K);
        {
            const auto t0 = std::chrono::steady_clock::now();
            const u32 seq = ix->out_seq;
            int j = 0;
            for (unsigned it = 0; j < kk; ++it) {
                for (; j < kk; ++j)
                    if (__atomic_load_n(&ix->h_out_g[j].tag_lo, __ATOMIC_ACQUIRE) != seq ||
                        __atomic_load_n(&ix->h_out_g[j].tag_hi, __ATOMIC_ACQUIRE) != seq)
                        break;
                if (j == kk) break;
                ls_cpu_relax();
                if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    LS_HIP(hipStreamSynchronize(s));
                    std::atomic_thread_fence(std::memory_order_acquire);
                    break;
                }
            }
        }
        for (int i = 0; i < kk; ++i) {
            out_scores[i] = ix->h_out_g[i].score;
            out_docs[i] = ix->h_out_g[i].row == 0xffffffffu ? (int64_t)-1 : (int64_t)ix->h_out_g[i].row;
        }
        for (int i = kk; i < k; ++i) {
            out_scores[i] = -FLT_MAX;
            out_docs[i] = -1;
        }
    } else {
        for (int i = 0; i < k; ++i) {
            out_scores[i] = -FLT_MAX;
            out_docs[i] = -1;
        }
    }
    return LS_OK;
}

}  // extern "C"
