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
 *   - ls_search / ls_search_device may be called concurrently on one handle, from several threads
 *     and on several streams: calls are serialised inside, and device work that shares the
 *     handle's scratch is fenced across streams by events. Concurrent ls_search calls do not
 *     queue one behind the other: whichever thread is serving takes every waiting request of the
 *     same k and flags (up to 16 queries) into ONE corpus pass (fp32 index: on the f32 matrix cores,
 *     bit-identical to the separate calls); waiters sleep, they do not spin. A synchronous call's
 *     launch may be queued while the previous call still waits for its answer (two host slots).
 *   - stream lifetime: a hipStream_t handed to ls_search_device must stay alive until the next
 *     ls_check (or synchronous call) on that handle has returned, or until ls_destroy: the handle
 *     remembers the stream of its most recent calls and may synchronise it when a later call
 *     arrives on a different stream (recording an event behind every call instead would cost
 *     several microseconds of GPU time per call; the library's own lanes ARE ordered by events).
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
#define LS_FLAG_PIPELINE 4u  /* ls_search_device only: queue on the index's internal streams  */
                             /* so that consecutive calls overlap (scan path: the selection   */
                             /* step of one query runs under the scan of the next; batched    */
                             /* MFMA path: consecutive batches alternate between two internal */
                             /* lanes with four scratch sets, behind whatever `stream` has    */
                             /* queued so far; the MFMA pass of a batch may be queued only    */
                             /* when the call after next arrives, because it also computes    */
                             /* that call's sample scores, or at ls_check()); results are NOT */
                             /* ordered on `stream` - they are valid after ls_check()         */

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

/* Row-sharded index over several GPUs of one node, in ONE process and behind the SAME handle type:
 * every function of this header accepts the handle it returns (the
 * reference's backend is a single process that owns one index, reference src/lean_explore/mcp/
 * server.py:147-151, and calls index.search from it, search/engine.py:250).
 * Shard g holds the contiguous row block [g*ceil(n/G), min(n, (g+1)*ceil(n/G))) on device_ids[g].
 * A search copies the queries to every device, runs the local exact top-k on all of them
 * concurrently, exchanges the packed per-shard results [scores | rows | flags] with ONE RCCL
 * all-gather over xGMI (ncclCommInitAll communicators, bound at first use) and merges the G sorted
 * lists on device_ids[0] under the total order: results are bit-identical to the unsharded index
 * for every G. Queries / outputs of ls_search_device live on device_ids[0] (= ls_device()).
 * Duplicate ids (e.g. {0,0,0}: G shards rehearsed on one GPU) are allowed; they exchange by
 * device-to-device copies because RCCL needs distinct devices. n_devices == 0 fails with
 * LS_ERR_NO_DEVICE (there is no CPU backend). LS_FLAG_PIPELINE reaches the shards' batched path
 * only; on the per-query scan path a sharded handle treats it as LS_FLAG_ASYNC. ls_add appends to
 * the last shard. ls_export_flags and ls_debug_read_scores are not available on a sharded handle. */
int ls_create_sharded(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
                      const int32_t* device_ids, int32_t n_devices);

/* REPLICAS instead of row shards: every device of device_ids holds the whole corpus, and the synchronous
 * host calls (ls_search - the reference's call, search/engine.py:250) are dealt round-robin to the
 * replicas, each with its own queue of concurrent callers: G callers run on G devices at once. For a
 * corpus that fits one GPU (the reference's ~200 k x 1024 fp32 = 0.8 GB) this is the shape that
 * scales queries/s with the device count; row shards only pay for corpora that do not fit (each query
 * touches every shard, plus the exchange). The handle is an ordinary ls_index*: ls_add appends to every
 * replica, ls_reconstruct reads replica 0, ls_search_device (queries resident on device_ids[0]) is
 * served by the replica there, ls_shard_count / ls_shard_info list the replicas (each covering all
 * rows). Results are those of a single-device index. debug counter 21: host calls dispatched. */
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

/* What the exchange step of a sharded handle does on THIS node, as one JSON object in `buf`
 * (NUL-terminated, truncated to `cap`): "exchange" ("rccl all-gather" | "peer-copy (RCCL failed)" |
 * "device-to-device copies (shards share a device)" ...), "rccl_version", "rccl_error" (why RCCL
 * could not be used: the handle then falls back to peer copies instead of failing every search),
 * "rccl_communicators", "enqueue_workers" (one host thread per shard queues that shard's work),
 * "devices" and the hipDeviceCanAccessPeer matrix "peer_access". Returns the length of the full
 * text, or a negative LS_ERR_* code. */
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

/* Same search with queries and outputs already in HBM on the index's device; work is queued
 * on `stream` (a hipStream_t; NULL = default stream). Without LS_FLAG_ASYNC it synchronises
 * the stream before returning. With LS_FLAG_ASYNC it returns after queueing (results ordered on
 * `stream`); with LS_FLAG_PIPELINE it returns after queueing on internal lanes (see the flag).
 * Lifetimes of an async / pipelined call: the QUERY buffer may be reused as soon as the work
 * queued on `stream` so far has consumed it (stream order; the library keeps its own copy for
 * repairs); the OUTPUT buffers must stay valid until the ls_check that covers the call, because a
 * repaired query is re-written in place. Call ls_check before trusting the results of ANY batched
 * call (the speculative MFMA paths: nq > 16 on an fp16 index, nq >= 24 on an fp32 index, on shards
 * of at least 8192 rows) or of any pipelined search; per-query scan-path calls (everything else)
 * are exact in stream order. (Single queries and small fp32 batches - ls_scan.hip, ls_mq.hip - write no score
 * vectors when the call is pipelined or synchronous: a query whose selection could not prove its keys
 * complete is served again, in place, at ls_check / before the synchronous call returns. With LS_FLAG_ASYNC
 * alone, or with LS_FLAG_INORDER, they keep the score vectors and are exact in stream / lane order.)
 * ls_debug_counter(index, 10) names the path the last call took. */
int ls_search_device(ls_index* index, const void* d_q, int64_t nq, int32_t k, uint32_t flags,
                     void* d_out_scores, void* d_out_indices, void* stream);

/* Synchronise `stream` and the index's internal lanes and make the results of every async /
 * pipelined search queued since the last ls_check final: queries of batched calls whose
 * speculative threshold let fewer than k rows through, or whose candidate queues overflowed, are
 * re-run here by the exact per-query scan path (from the library's own copy of the queries) and
 * their output rows re-written; so are the queries of pipelined scan-path launches (single queries and small
 * fp32 batches, up to 256 launches between two checks) that raised their repair word. Returns LS_OK once everything is exact. Up to 1024 batched calls
 * (fewer for batches of more than 4096 queries) may be outstanding; one more triggers the same repair step on its own. */
int ls_check(ls_index* index, void* stream);

/* Copy the per-query verification flags of the most recent search queued on this handle into
 * d_dst (device memory on the index's device, uint32 [nq]) in `stream` order: non-zero = that
 * query's output rows are provisional until ls_check. Scan-path searches export zeros: a caller that
 * ships scan-path results before ls_check passes LS_FLAG_INORDER (or LS_FLAG_ASYNC alone), which makes them
 * exact in order. Lets a sharded caller ship the flags with the results instead of synchronising before
 * the exchange. */
int ls_export_flags(ls_index* index, void* d_dst, int64_t nq, void* stream);

/* faiss.normalize_L2(x): in-place row normalisation of host float32 [nq, d];
 * rows with zero norm are left unchanged. Runs on `device` (cached pinned staging buffers that the
 * kernel reads and writes directly; no allocation per call). The squared norm is summed in the
 * library's one documented order (ls_common.h, ls_wave_sumsq), the same the fused
 * LS_FLAG_NORMALIZE uses, so both routes give bit-identical queries. */
int ls_normalize_l2(float* x, int64_t nq, int32_t d, int32_t device);

/* Merge `n_lists` per-shard results (each [nq, k], sorted by the total order, -1 padded)
 * into the global top-k. All pointers are device memory on `device`:
 * d_scores_in float32 [n_lists, nq, k], d_indices_in int64 [n_lists, nq, k]
 * (the layout an RCCL all-gather of per-rank results produces). */
int ls_merge_topk(const void* d_scores_in, const void* d_indices_in, int32_t n_lists,
                  int64_t nq, int32_t k, void* d_out_scores, void* d_out_indices,
                  int32_t device, void* stream);

/* As ls_merge_topk for the packed exchange buffer of the sharded path: list l's scores start at
 * (char*)d_scores_in + l*list_stride_bytes and its rows at (char*)d_indices_in +
 * l*list_stride_bytes (one all-gather of a per-rank block [scores | rows] yields this layout).
 * list_stride_bytes must be a multiple of 8. */
int ls_merge_topk_strided(const void* d_scores_in, const void* d_indices_in,
                          int64_t list_stride_bytes, int32_t n_lists, int64_t nq, int32_t k,
                          void* d_out_scores, void* d_out_indices, int32_t device, void* stream);

/* Kernel timing for bench.py. While profiling is on, every scan launch (up to 4096) is
 * bracketed by hipEvents on the stream it runs on. ls_last_kernel_ms returns the MEAN duration
 * of the scan kernel and of scan + selection over the launches recorded since profiling was
 * switched on (or since the last read), and clears the record. */
int ls_set_profiling(ls_index* index, int32_t enabled);
int ls_last_kernel_ms(ls_index* index, float* scan_ms, float* total_ms);

/* Test / tuning hooks.
 * option 0: force the number of keys k' each scan workgroup emits (0 = automatic);
 * option 1: force the finalize step's general exact path; option 2: alternate the sweep
 * direction of consecutive scans (default off); option 3: piggy-back the finalize of a query
 * group on the next scan launch (default on); option 4: allow the batched MFMA path (default on);
 * option 5: speculative, verified sample threshold on the batched path (default on; off = the
 * certified k-th sample score); option 6: several queries per corpus pass on the scan path
 * (default on); option 7: force the number of scan workgroups per launch (0 = automatic);
 * option 9: synchronous host searches (ls_search, nq <= 16) run the selection step inside the scan
 * launch of its own query, sweeping the tagged 16-byte granules the scan workgroups write their keys as
 * (no drain, no counter, no fence; a query whose keys cannot be proven complete answers "retry" in its
 * completion word and the host launches the stand-alone selection): 0 off, 1 on (default);
 * option 15: synchronous host searches let the scan workgroups read the pinned host copy of the query
 * over PCIe (0, default: the read hides under the first corpus tile) or bring it to device memory with a
 * copy command in front of the launch (1: measured 1.6-2 us slower per call);
 * option 10: synchronous host searches (ls_search) that arrive while another one is running are
 * served together, up to 16 queries of equal k and flags per corpus pass (default on);
 * option 13: pipelined fp16 batches of stored rows of up to 768 bytes let the sample phase of the batch
 * two calls ahead ride on the MFMA pass launch: 0 off, 1 on (default); option 14: select
 * step of the batched path as one wave per query in <= 48 registers where the shape allows (k <= 128,
 * <= 128 corpus slices; runs inside a resident MFMA pass): default on;
 * option 16: fp32 index, 2..16 queries per corpus pass on the f32 matrix cores (ls_mq.hip): default on
 * (0: the VALU scan groups of 8 / 4 / 1 - same bits); option 17: synchronous host calls may overlap two
 * deep (default on); option 18: fp16 index with 768-byte stored rows, batched pass in the row-split,
 * 64-queries-per-wave shape (measured slower, profiles/ab/r05_tile_shape.txt: default off);
 * option 20: concurrent ls_search callers are gathered into ONE pass - a leader with nothing in flight waits up
 * to a third of a call, at most 60 us, for the callers seen lately - instead of two passes at once on the two
 * host slots (2 default; 1: only calls longer than 110 us, e.g. d = 1024; 0 off: option 17's two-deep overlap
 * decides); option 21: callers up to which a second batch may go early when option 20 allows it (default 8);
 * option 19: launches of synchronous host calls (ls_scan and ls_mq) and ls_mq launches of pipelined /
 * synchronous device calls write no score vectors; an unproven query is served again on the scan kernel - same bits (default on; 0: every
 * launch writes them and the selection repairs from them);
 * option 8 (sharded handles): exchange step 0 = RCCL all-gather between distinct devices (default),
 * 1 = peer copies into the primary device's gather buffer, 2 = RCCL gather-to-root (ncclSend / ncclRecv:
 * only the primary, which merges, receives the blocks); option 11 (sharded handles): one host
 * thread per shard queues that shard's work: -1 automatic (on when the device ids are distinct,
 * default), 0 off, 1 on; option 12 (sharded handles, test hook): make the next RCCL exchange fail
 * (the handle must fall back to peer copies and keep answering).
 * counter 9: kernel launches the most recent batched call queued (counted per launch);
 * counter 10: path of the most recent search (1 per-query scan, 2 fp16 MFMA, 3 fp32 MFMA);
 * counter 11: kernel launches queued by searches on this handle so far; counter 12: batched calls
 * that were cut into sub-batches because the candidate queues could not hold the whole batch;
 * sharded handles: counter 13 exchange steps run, 14 re-exchanges after a shard repaired a
 * query, 15 exchange transport (0 copies, 1 RCCL selected, 2 RCCL communicators initialised, 3 RCCL failed
 * on this node: fell back to peer copies), 19 calls whose shards were queued by the enqueue workers;
 * counter 20: synchronous host calls that had to launch the stand-alone selection (option 9's retry);
 * counter 22: checks of pending batched calls the library ran on its own (slots exhausted or re-sliced; summed);
 * counter 23: ls_mq launches (small fp32 batches on the f32 matrix cores); counter 24: synchronous host calls
 * that were queued while another one was still in flight; counter 25: queries of ls_mq launches without
 * score vectors that were served again on the scan kernel;
 * counters 0, 1, 8, 11, 12 are summed over the shards, 9 and 10 are the primary shard's;
 * counter 16: combined batches ls_search served, 17: the requests they carried; counter 18 (sharded
 * handles): mean host nanoseconds spent queueing one search (every device's work + exchange + merge).
 * counter 0: searches whose finalize step left the fast path (rescue or general); counter 1:
 * those that took the general path; counter 8: queries of batched calls that were repaired by
 * the exact scan path. */
int ls_debug_option(ls_index* index, int32_t which, int32_t value);
int64_t ls_debug_counter(ls_index* index, int32_t which);
/* Copy the score vector S[0..count) of the most recent per-query scan to host memory. */
int ls_debug_read_scores(ls_index* index, float* out, int64_t count);

/* ---- lexical (BM25+) name retrieval: SURVEY §8(f) row 3 -----------------------------------------
 * Stands in for `bm25s.BM25.retrieve([tokens], k)` (reference src/lean_explore/search/engine.py:
 * 209-214). The index is bm25s's eager-sparse CSC matrix (one column per vocabulary token:
 * document rows + float32 scores) plus its per-token non-occurrence array (bm25+), i.e. the
 * arrays bm25s saves as data/indices/indptr.csc.index.npy and nonoccurrence_array.index.npy
 * (reference src/lean_explore/cli/data_commands.py:42-59). All pointers are host memory. */
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
/* Test hook. counter 0: searches whose selection step left its fast path; counter 1: those that
 * needed the general select over the score vector. */
int64_t ls_bm25_debug_counter(ls_bm25* index, int32_t which);
void ls_bm25_destroy(ls_bm25* index);

const char* ls_last_error(void); /* thread-local; valid until the next call on this thread */
const char* ls_version(void);
int32_t ls_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* LEANSEARCH_H */
