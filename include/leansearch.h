/*
 * leansearch.h — C ABI of libleansearch.so, the MI355X (gfx950) exact inner-product
 * top-k search that replaces the FAISS lookup of LeanExplore's local search backend.
 *
 * The reference has no FFI of its own for this path: it reaches FAISS through the faiss
 * Python module, duck-typed on one attribute (reference src/lean_explore/search/engine.py:99).
 * Each entry point below therefore cites the faiss call (reference file:line) it stands in for.
 *
 *   reference call                                                    replaced by
 *   ---------------------------------------------------------------  -------------------------
 *   faiss.read_index(path)            search/engine.py:159            ls_create (+ Python loader)
 *   faiss.IndexFlatIP(d); index.add(x)   extract/index.py:103,116     ls_create / ls_add
 *   faiss.normalize_L2(x)             search/engine.py:242            ls_normalize_l2
 *   index.search(x, k) -> (D, I)      search/engine.py:250            ls_search
 *   index.ntotal / index.d            tests/extract/index_test.py:172-173   ls_ntotal / ls_dim
 *
 * Conventions
 *   - every function returns LS_OK (0) or a negative LS_ERR_* code and never throws; the message
 *     for the last failure on the calling thread is available from ls_last_error().
 *   - the caller owns every buffer it passes. ls_create copies the corpus into HBM and does
 *     not keep the host pointer.
 *   - result order is the total order (score descending, row index ascending). Slots past the
 *     number of valid rows hold index -1 and score -FLT_MAX (IndexFlat's heap-neutral padding,
 *     which the reference relies on at search/engine.py:254). Rows whose score is NaN or
 *     <= -FLT_MAX are never returned.
 *   - ls_search / ls_search_device may be called concurrently on one handle, from several threads and on
 *     several streams (serialised inside; scratch shared across streams is fenced by events). Concurrent
 *     ls_search calls are not queued one behind the other: whichever thread is serving takes every waiting
 *     request of the same k and flags (up to 32 queries on an fp32 index, 16 otherwise) into ONE corpus pass,
 *     bit-identical to the separate calls ("Concurrency").
 *   - stream lifetime: a hipStream_t handed to ls_search_device must stay alive until the next ls_check (or
 *     synchronous call) on that handle has returned, or until ls_destroy.
 *   - there is no CPU fallback: with no usable HIP device every compute entry point fails with
 *     LS_ERR_NO_DEVICE.
 */
#ifndef LEANSEARCH_H
#define LEANSEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS_OK 0
#define LS_ERR_INVALID_ARG (-1)   /* null pointer, bad shape, unknown dtype/flag              */
#define LS_ERR_NO_DEVICE (-2)     /* no HIP device / device id out of range                   */
#define LS_ERR_HIP (-3)           /* a HIP runtime call failed (message has the HIP string)   */
#define LS_ERR_K_TOO_LARGE (-4)   /* min(k, ntotal) exceeds LS_MAX_K                          */
#define LS_ERR_OVERFLOW (-5)      /* reserved (ls_check repairs flagged queries itself)       */

#define LS_DTYPE_F32 0 /* corpus stored in HBM as fp32 (what the reference stores)           */
#define LS_DTYPE_F16 1 /* corpus rounded to fp16 in HBM; queries are rounded to fp16 as well, */
                       /* products are exact, accumulation is fp32                            */

#define LS_FLAG_NORMALIZE 1u /* L2-normalise a private copy of the queries first (fuses          */
                             /* faiss.normalize_L2, search/engine.py:242, into the search)     */
#define LS_FLAG_ASYNC 2u     /* ls_search_device only: queue and return; results are ordered  */
                             /* on `stream` like any other work queued there                  */
#define LS_FLAG_PIPELINE 4u  /* ls_search_device only: queue on the index's internal streams so that       */
                             /* consecutive calls overlap (scan path: the selection of one query runs    */
                             /* under the scan of the next; batched MFMA path: two internal lanes);      */
                             /* results are NOT ordered on `stream` - they are valid after ls_check()    */

#define LS_FLAG_INORDER 8u   /* ls_search_device with LS_FLAG_PIPELINE: the caller consumes scan-path    */
                             /* results on the GPU before ls_check (in the lanes' order, e.g. a sharded */
                             /* exchange): every launch keeps its score vectors and repairs in-kernel   */

#define LS_MAX_K 2048 /* same ceiling as FAISS's GPU k-selection; reference uses k = 1000     */

typedef struct ls_index ls_index; /* opaque */

/* Build a flat inner-product index over `n` rows of dimension `d`.
 * `corpus` is host memory, row-major float32 [n, d] (the layout index.add receives at
 * reference extract/index.py:71,116). dtype selects the HBM storage (LS_DTYPE_*).
 * `device` is the HIP device ordinal. n == 0 is allowed (every search returns padding). */
int ls_create(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
              int32_t device);

/* Row-sharded index over several GPUs of one node, in ONE process and behind the SAME handle type: every
 * function of this header accepts the handle it returns (the reference's backend is
 * one process that owns one index, reference src/lean_explore/mcp/server.py:147-151 -> search/engine.py:250).
 * Shard g holds the contiguous row block [g*ceil(n/G), min(n, (g+1)*ceil(n/G))) on device_ids[g]; a search runs
 * the local exact top-k on every shard, exchanges the packed results [scores | rows | flags] with ONE RCCL
 * all-gather over xGMI and merges the G sorted lists on device_ids[0]: bit-identical to the unsharded index
 * for every G. Queries / outputs of ls_search_device live on device_ids[0]. Duplicate ids
 * (G shards rehearsed on one GPU) exchange by device-to-device copies. n_devices == 0 fails with
 * LS_ERR_NO_DEVICE. ls_add appends to the last shard; ls_export_flags is not available on a sharded handle. */
int ls_create_sharded(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
                      const int32_t* device_ids, int32_t n_devices);

/* REPLICAS instead of row shards: every device holds the whole corpus and the synchronous host calls (ls_search)
 * are dealt round-robin to the replicas, each with its own queue of concurrent callers - the shape that scales
 * queries/s with the device count for a corpus that fits one GPU (the reference's 200 k x 1024 fp32 = 0.8 GB).
 * ls_add appends to every replica; results are those of a single-device index. */
int ls_create_replicated(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
                         const int32_t* device_ids, int32_t n_devices);

/* As ls_create_sharded with the row blocks already in HBM: d_blocks[g] is device memory on
 * device_ids[g], row-major float32 [rows[g], d]; global rows are numbered block after block. */
int ls_create_sharded_from_device(ls_index** out, const void* const* d_blocks, const int64_t* rows,
                                  int32_t d, int32_t dtype, const int32_t* device_ids,
                                  int32_t n_devices);

/* Number of shards of a handle (0 for a plain single-device handle) and one shard's placement. */
int32_t ls_shard_count(const ls_index* index);
int ls_shard_info(const ls_index* index, int32_t shard, int32_t* device, int64_t* row0,
                  int64_t* rows);

/* What the exchange step of a sharded handle does on THIS node, as one JSON object in `buf` (NUL-terminated,
 * truncated to `cap`): "exchange" (rccl all-gather | peer copies | ...), "rccl_version", "rccl_error",
 * "enqueue_workers", "devices", "peer_access". Returns the length of the full text, or a negative LS_ERR_* code. */
int32_t ls_shard_exchange_info(ls_index* index, char* buf, int32_t cap);

/* As ls_create, but `d_corpus` is device memory on `device`, row-major float32 [n, d]
 * (used to build multi-GB synthetic shards without a host round trip). */
int ls_create_from_device(ls_index** out, const void* d_corpus, int64_t n, int32_t d,
                          int32_t dtype, int32_t device);

/* index.add(x) on an existing index (reference extract/index.py:116): append `n_add` host
 * float32 rows [n_add, d]. The rows already stored stay in HBM (device-to-device carry-over); only
 * the new rows cross PCIe. Synchronises the handle's outstanding work first. */
int ls_add(ls_index* index, const float* rows, int64_t n_add);

/* index.reconstruct_n(row0, count): copy stored rows back to host float32 [count, d]. An fp16
 * index returns the rounded values. (faiss.write_index needs the rows; the Python wrapper keeps no
 * host copy of the corpus.) */
int ls_reconstruct(ls_index* index, int64_t row0, int64_t count, float* out);

void ls_destroy(ls_index* index);

int64_t ls_ntotal(const ls_index* index); /* index.ntotal */
int32_t ls_dim(const ls_index* index);    /* index.d      */
int32_t ls_dtype(const ls_index* index);
int32_t ls_device(const ls_index* index);

/* Offset added to every returned row index (a shard's first global row). Default 0. */
int ls_set_base(ls_index* index, int64_t base);

/* index.search(x, k): q is host float32 [nq, d]; out_scores host float32 [nq, k];
 * out_indices host int64 [nq, k]. Synchronous. */
int ls_search(ls_index* index, const float* q, int64_t nq, int32_t k, uint32_t flags,
              float* out_scores, int64_t* out_indices);

/* Same search with queries and outputs already in HBM on the index's device; work is queued on `stream` (a
 * hipStream_t; NULL = default stream). Without flags it synchronises the stream before returning; LS_FLAG_ASYNC
 * returns after queueing (results ordered on `stream`), LS_FLAG_PIPELINE after queueing on internal lanes.
 * Lifetimes of an async / pipelined call: the QUERY buffer may be reused as soon as the work queued on `stream`
 * so far has consumed it (the library keeps its own copy for repairs); the OUTPUT buffers must stay valid until
 * the ls_check that covers the call, because a repaired query is re-written in place - and should not be handed
 * to another call before that check (the library skips the repair of rows a LATER pipelined call of the same
 * handle was given in the meantime, but it cannot see any other writer).
 * Call ls_check before trusting the results of ANY batched call (the speculative, verified MFMA paths: nq > 16 on
 * an fp16 index, nq > 32 on an fp32 index, shards of at least 8192 rows) or of any pipelined search: pipelined /
 * synchronous scan-path launches write no score vectors, and a query whose selection could not prove its keys
 * complete (~1e-3 per query) is served again, in place, at ls_check / before the synchronous call returns. With
 * LS_FLAG_ASYNC alone, or with LS_FLAG_INORDER, scan-path calls are exact in stream / lane order. */
int ls_search_device(ls_index* index, const void* d_q, int64_t nq, int32_t k, uint32_t flags,
                     void* d_out_scores, void* d_out_indices, void* stream);

/* Synchronise `stream` and the index's internal lanes and make the results of every async / pipelined search
 * queued since the last ls_check final: flagged queries (batched calls whose verified threshold failed or whose
 * candidate queues overflowed; pipelined scan-path launches that raised their repair word) are re-run by the exact
 * per-query scan path from the library's own copy of the queries and their output rows re-written. Up to 1024
 * batched calls / 256 pipelined scan-path launches may be outstanding; one more triggers the same step by itself. */
int ls_check(ls_index* index, void* stream);

/* Copy the per-query verification flags of the most recent search queued on this handle into d_dst (device
 * memory, uint32 [nq]) in `stream` order: non-zero = that query's rows are provisional until ls_check (scan-path
 * searches export zeros). Lets a sharded caller ship the flags with the results instead of synchronising. */
int ls_export_flags(ls_index* index, void* d_dst, int64_t nq, void* stream);

/* faiss.normalize_L2(x): in-place row normalisation of host float32 [nq, d]; rows with zero norm are left
 * unchanged. Runs on `device`. The squared norm is summed in the library's one documented order (ls_common.h,
 * ls_wave_sumsq), the same the fused LS_FLAG_NORMALIZE uses: both routes give bit-identical queries. */
int ls_normalize_l2(float* x, int64_t nq, int32_t d, int32_t device);

/* Merge `n_lists` per-shard results (each [nq, k], sorted by the total order, -1 padded)
 * into the global top-k. All pointers are device memory on `device`:
 * d_scores_in float32 [n_lists, nq, k], d_indices_in int64 [n_lists, nq, k]
 * (the layout an RCCL all-gather of per-rank results produces). */
int ls_merge_topk(const void* d_scores_in, const void* d_indices_in, int32_t n_lists,
                  int64_t nq, int32_t k, void* d_out_scores, void* d_out_indices,
                  int32_t device, void* stream);

/* As ls_merge_topk for the packed exchange buffer of the sharded path: list l's scores start at
 * (char*)d_scores_in + l*list_stride_bytes, its rows at (char*)d_indices_in + l*list_stride_bytes (a multiple of 8). */
int ls_merge_topk_strided(const void* d_scores_in, const void* d_indices_in,
                          int64_t list_stride_bytes, int32_t n_lists, int64_t nq, int32_t k,
                          void* d_out_scores, void* d_out_indices, int32_t device, void* stream);

/* Kernel timing, tuning hooks and counters (ls_set_profiling, ls_last_kernel_ms, ls_debug_option,
 * ls_debug_counter, ls_debug_read_scores, ls_bm25_debug_counter): include/leansearch_debug.h. A binder of the
 * search path needs none of them. */

/* ---- lexical (BM25+) name retrieval: SURVEY section 8(f) row 3. Stands in for `bm25s.BM25.retrieve([tokens], k)`
 * (reference src/lean_explore/search/engine.py:209-214). The index is bm25s's eager-sparse CSC matrix (one column
 * per vocabulary token: document rows + float32 scores) plus its per-token non-occurrence array (bm25+), as saved
 * by reference src/lean_explore/cli/data_commands.py:42-59. All pointers are host memory. */
typedef struct ls_bm25 ls_bm25;
int ls_bm25_create(ls_bm25** out, const int64_t* indptr, const int32_t* indices, const float* data,
                   const float* nonoccurrence, int64_t n_docs, int64_t n_vocab, int32_t device);
/* token_ids: the query's token ids in query order (unknown tokens already dropped, duplicates
 * kept). out_scores float32 [k], out_docs int64 [k], best first under (score desc, doc asc),
 * padded with (-FLT_MAX, -1) when k > n_docs. Scores are bit-identical to the sequential
 * float32 accumulation bm25s performs. Synchronous. */
int ls_bm25_search(ls_bm25* index, const int32_t* token_ids, int32_t n_tokens, int32_t k,
                   float* out_scores, int64_t* out_docs);
int64_t ls_bm25_ntotal(const ls_bm25* index);
void ls_bm25_destroy(ls_bm25* index);

const char* ls_last_error(void); /* thread-local; valid until the next call on this thread */
const char* ls_version(void);
int32_t ls_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* LEANSEARCH_H */
