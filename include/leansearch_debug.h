/*
 * leansearch_debug.h — kernel timing, tuning hooks and counters of libleansearch.so (include/leansearch.h):
 * what bench.py, the tests and the A/B tools use. Nothing here is part of the drop-in surface; a binder of
 * the search path does not need this header.
 *
 * Measured losers do not ship: the row-split fp16 pass (option 18) builds only with
 * `make variant NAME=rs2 VFLAGS=-DLS_VARIANT_RS2`; the query copy command of synchronous host calls
 * (option 15) and "gather long passes only" (option 20 = 1) were removed in round 6.
 */
#ifndef LEANSEARCH_DEBUG_H
#define LEANSEARCH_DEBUG_H

#include "leansearch.h"

#ifdef __cplusplus
extern "C" {
#endif

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
 * option 10: synchronous host searches (ls_search) that arrive while another one is running are
 * served together, up to 32 queries (fp32 index; 16 otherwise) of equal k and flags per corpus pass (default on);
 * option 13: pipelined fp16 batches of stored rows of up to 768 bytes let the sample phase of the batch
 * two calls ahead ride on the MFMA pass launch: 0 off, 1 on (default); option 14: select
 * step of the batched path as one wave per query in <= 48 registers where the shape allows (k <= 128,
 * <= 128 corpus slices; runs inside a resident MFMA pass): default on;
 * option 16: fp32 index, 2..32 queries per corpus pass on the f32 matrix cores (ls_mq.hip): default on
 * (0: the VALU scan groups of 8 / 4 / 1 - same bits); option 17: synchronous host calls may overlap two
 * deep (default on); option 18 (variant builds only, -DLS_VARIANT_RS2): fp16 index with 768-byte stored
 * rows, batched pass in the row-split, 64-queries-per-wave shape;
 * option 20: concurrent ls_search callers are gathered into ONE pass - a leader with nothing in flight waits up
 * to a third of a call, at most 60 us, for the callers seen lately - instead of two passes at once on the two
 * host slots (default on; 0 off: option 17's two-deep overlap decides); option 21: callers up to which a
 * second batch may go early when option 20 allows it (default 8);
 * option 23: concurrent ls_search callers - a queue that alone fills a pass (32 requests on an fp32 index) is launched at once
 * behind the call in flight, on the other host slot (default on);
 * option 22: fp32 index, one ls_mq pass carries up to 32 queries (two MFMA B blocks per A operand; default on;
 * 0: 16 per pass - same bits);
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
 * score vectors that were served again on the scan kernel; counter 26: such repairs skipped because a later
 * pipelined call had been given the same output rows; counter 27: synchronous host calls whose 2 ms poll for the
 * results expired (they slept in hipStreamSynchronize instead);
 * counters 28-32, the phase clocks of ls_search's batch leaders, cumulative nanoseconds: 28 waiting for the call in
 * flight + gathering, 29 begin..finish of their batch, 30 re-taking the queue's mutex, 31 of 29: the enqueue
 * (staging + launch), 32 of 29: the wait for the results and handing them out; counter 33: waiters that went to
 * sleep on their request - at most (CPUs of the process: affinity mask, capped by the cgroup's CPU quota) - 3
 * waiters poll, environment LS_SPIN_CPUS overrides the CPU count (tools/callers_c.c prints all of them with
 * CALLERS_COUNTERS=1; profiles/ab/r06_open_loop.txt);
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

/* BM25 test hook. counter 0: searches whose selection step left its fast path; counter 1: those that
 * needed the general select over the score vector. */
int64_t ls_bm25_debug_counter(ls_bm25* index, int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* LEANSEARCH_DEBUG_H */
