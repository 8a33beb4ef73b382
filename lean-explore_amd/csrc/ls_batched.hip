// ls_batched.hip - host side of the batched MFMA path (ls_gemm.hip, ls_gemm32.hip, ls_wsel.hip): the geometry of one
// batched call (corpus slices, sample thinning, speculative rank), its launches - prepare, sample, tau, pass + select
// chained over the handle's lanes - and the deferred work between pipelined calls. ls_api.hip decides which calls
// come here (ls_i_batched_eligible) and repairs overflowed queries on the exact scan path (ls_i_batched_repair).
#include "ls_index.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

// Geometry of one batched call: corpus slices, sample thinning, speculative rank, expected passes.
struct bc_plan {
    int QT, TM;               // queries per workgroup, corpus rows per LDS tile
    int64_t nq_pad, rps;      // padded queries, rows per slice
    int nsplits, sample_stride, jrank, keys_need;
    double per_queue;         // expected entries of one private candidate queue (capacity LS_GEMM_QCAP)
};
static bc_plan bc_make_plan(const ls_index* ix, int64_t nq, int32_t k) {
    const ls_geom& g = ix->g;
    const bool f32 = ix->dtype == LS_DTYPE_F32;
    const int QT = f32 ? 64 : ls_gemm_qt(g);  // queries per workgroup
    const int TM = f32 ? 64 : ls_gemm_tile_rows(g);  // rows a slice contributes to one tile
    const int64_t nq_pad = (nq + QT - 1) / QT * QT;
    const int nqt = (int)(nq_pad / QT);
    // corpus slices: one 8-wave workgroup per CU in total, a multiple of the 8 XCDs. The fp32
    // kernel is light on registers (two workgroups share a CU) and each workgroup walks two slices.
    int nsplits = (LS_GEMM_WG_PER_CU * ix->n_cu / nqt) / 8 * 8;
    nsplits = std::max(8, std::min(nsplits, 256));
    if (f32) nsplits = 2 * std::max(8, std::min(LS_GEMM_MAX_SPLITS / 2, (2 * ix->n_cu / nqt) / 8 * 8));
    else nsplits *= ls_gemm_rs(g);  // the row-split shape: every workgroup walks two slices
    int64_t rps = (ix->n + nsplits - 1) / nsplits;
    rps = (rps + TM - 1) / TM * TM;
    const int tiles_per_split = (int)(rps / TM);
    // the sample is a fixed FRACTION of the corpus (~1/24 of every slice, at least
    // LS_GEMM_SAMPLE_ROWS rows; ~1/48 for slices of more than 1536 tiles, where the sample pass
    // itself is what costs): the expected number of rows passing tau, ~j*N/M0, then does not
    // grow with N
    // The fraction thins out on long slices (about 16 tiles per slice up to 1/96 of the rows:
    // config 4's 12.5 M-row shard spent 7 % of its batch in a 1/24 sample): a thinner sample
    // passes more rows per query (~j*N/M0 +- that over sqrt(j)), which the select kernel's key
    // buffer must hold; the fraction is halved until the +5 sigma count fits LS_BSEL_MAX_KEYS.
    int frac = std::max(24, std::min(96, tiles_per_split / 16));
    int sample_tiles, sample_stride, jrank, keys_need;
    double pass_mean = 0.0;
    for (;; frac /= 2) {
        sample_tiles = std::max(std::max(1, LS_GEMM_SAMPLE_ROWS / TM),
                                (tiles_per_split + frac - 1) / frac);
        sample_stride = std::max(1, (tiles_per_split + sample_tiles - 1) / sample_tiles);
        // Speculative threshold. The k-th best SAMPLE score is a certified lower bound of the
        // final k-th best but passes ~k*N/M0 rows per query. The j-th best sample score (j < k)
        // passes only ~j*N/M0 rows; it is not certified, so the select kernel verifies that at
        // least k rows passed and flags the query for the exact scan path otherwise. j is the
        // smallest rank whose expected pass count exceeds k by 4.5 standard deviations (relative
        // sd of an order statistic ~ 1/sqrt(j)): a flag is a ~1e-5 event per query on
        // exchangeable rows.
        jrank = k;
        const int visited = (tiles_per_split + sample_stride - 1) / sample_stride;
        const double m0 = (double)nsplits * visited * TM;
        const double r = (double)ix->n / std::max(1.0, m0);
        if (ix->opt_spec_tau) {
            for (int j = 1; j <= k; ++j) {
                if ((double)j * r * (1.0 - 4.5 / __builtin_sqrt((double)j)) >= (double)k) {
                    jrank = j;
                    break;
                }
            }
        }
        pass_mean = (double)jrank * r;
        const double expect = pass_mean * (1.0 + 5.0 / __builtin_sqrt((double)jrank));
        keys_need = (int)std::min(expect, 1e9);
        if (keys_need <= LS_BSEL_MAX_KEYS || frac <= 24) break;
    }
    bc_plan p;
    p.QT = QT;
    p.TM = TM;
    p.nq_pad = nq_pad;
    p.rps = rps;
    p.nsplits = nsplits;
    p.sample_stride = sample_stride;
    p.jrank = jrank;
    p.keys_need = keys_need;
    p.per_queue = pass_mean / ((double)nsplits * 4.0);
    return p;
}

// The candidate queues are private per (query, slice, quarter) and hold LS_GEMM_QCAP entries: a
// batch fits when a queue's expected length leaves 5 sigma of Poisson headroom. nsplits shrinks
// as n_cu / query tiles, so big batches at big k (nq = 4096, k = 1000: 16 slices) would overflow
// every queue and send every query to the repair path; such calls are cut into sub-batches.
static bool bc_plan_fits(const bc_plan& p) {
    return p.per_queue + 5.0 * __builtin_sqrt(p.per_queue) <= (double)LS_GEMM_QCAP &&
           p.keys_need <= LS_BSEL_MAX_KEYS;
}
// Largest sub-batch (a multiple of the query tile) whose plan fits; 0 = not even one tile does.
int64_t ls_i_bc_chunk(const ls_index* ix, int64_t nq, int32_t k) {
    bc_plan p = bc_make_plan(ix, nq, k);
    if (bc_plan_fits(p)) return nq;
    for (int64_t tiles = p.nq_pad / p.QT / 2; tiles >= 1; tiles /= 2) {
        p = bc_make_plan(ix, tiles * p.QT, k);
        if (bc_plan_fits(p)) return tiles * p.QT;
    }
    return 0;
}

// ---- one batched call ------------------------------------------------------------------------------
// Kernels of a batch: query prep -> sample pass -> tau -> MFMA pass -> select. Plain calls queue all
// of it on the caller's stream.
//
// LS_FLAG_PIPELINE calls go through the handle's CHAIN (round 4): four streams, four scratch sets.
//   prep stream   : prep(i)        behind the caller's stream (chain_in) and behind the previous readers
//                                  of the scratch set's prepared queries; a one-wave, 37-register
//                                  kernel that runs INSIDE whatever pass is resident
//   two main lanes: lane i % 2 runs  ... L(i-2) -> tau(i) -> L(i) -> tau(i+2) -> L(i+2) ...
//                                  where L(i) = ONE launch: the MFMA pass of batch i, then the sample
//                                  phase of batch i+2 (LS_GEMM_FUSED). Nothing orders the two lanes:
//                                  the workgroups of L(i+1) move onto the CUs as those of L(i) retire,
//                                  so the kernel boundaries around a pass (5 us behind it, 7 us in
//                                  front), the tau kernel and the lane's event waits all hide under
//                                  the OTHER lane's pass. For that, batch i's pass is queued when call
//                                  i+2 arrives (or at the next flush: ls_check, a call of another shape,
//                                  ls_export_flags, ls_add ...): results of pipelined calls are
//                                  defined to be valid after ls_check anyway.
//   select stream : select(i)      behind L(i): the one-wave select kernel (<= 48 VGPRs, ls_wsel.hip)
//                                  runs INSIDE the passes that follow, in the registers and LDS a
//                                  pass leaves free, instead of between two passes
// The event the select stream waits for is attached to the pass's dispatch (hipExtLaunchKernelGGL):
// no extra packet behind a pass. Measured on the way here (tools/c3_timeline.sh, tools/
// coresidency_probe.hip, profiles/ab/r04_c3_chain.txt): a cross-stream dependency takes ~15-20 us to
// resolve on this runtime, a wait packet in front of a pass ~8 us, the boundaries around a pass 5 + 7
// us - with ONE main stream those sat between two passes (36 us per batch), hence two lanes. Round
// 3 rotated whole batches over two "lanes" as well, but each lane then ran its own sample pass - the
// same 8-wave, 96 KB kernel, which could only start on CUs the other lane's pass had left (~25 us
// per batch).
static int bc_launch_select(ls_index* ix, const ls_index::bc_stage& b, hipStream_t ss) {
    ls_index::bc_set& st = ix->bc_sets[b.set_id];
    ls_gemm_bufs bufs;
    bufs.d_queues = st.d_queues;
    bufs.d_counts = st.d_counts;
    bufs.d_overflow = b.d_flags;
    bufs.d_sample_top = st.d_sample_top;
    if (ix->opt_wave_select && !b.f32 && ls_wave_select_ok(b.nsplits, b.k, b.keys_need))
        return ls_launch_wave_select(bufs, b.nsplits, b.nq, b.k, ix->base, ix->n, b.rps, b.d_out_s,
                                     b.d_out_i, ss, nullptr);
    return ls_launch_batch_select(bufs, b.nsplits, b.nq, b.k, b.keys_need, ix->base, ix->n, b.rps,
                                  b.d_out_s, b.d_out_i, ss);
}

// Queue the MFMA pass and the select of batch `b` (its sample pass and tau are already queued on
// `sm`). `next` non-null: the pass launch also runs the sample phase of that batch (same plan).
static int bc_launch_pass_select(ls_index* ix, const ls_index::bc_stage& b, const ls_index::bc_stage* next,
                                 hipStream_t sm, hipStream_t ss, bool tau_next = false) {
    ls_index::bc_set& st = ix->bc_sets[b.set_id];
    const ls_geom& g = ix->g;
    int rc;
    ls_gemm_bufs bufs;
    bufs.d_queues = st.d_queues;
    bufs.d_counts = st.d_counts;
    bufs.d_overflow = b.d_flags;
    bufs.d_sample_top = st.d_sample_top;
    const bool prof = ix->profiling && ix->prof_n < LS_PROF_MAX;
    hipEvent_t* pe = nullptr;
    if (prof) {
        while (ix->prof_ev.size() < 2 * (ix->prof_n + 1)) {
            hipEvent_t e;
            LS_HIP(hipEventCreate(&e));
            ix->prof_ev.push_back(e);
        }
        pe = &ix->prof_ev[2 * ix->prof_n];
    }
    // The select of a chain batch runs on the chain's select stream, behind an event attached to the
    // pass's dispatch (no extra packet in the lane). While profiling, a timing pair rides there
    // instead and brackets exactly the kernel.
    const bool xsel = ss != sm;
    hipEvent_t const ev_start = prof ? pe[0] : nullptr;
    hipEvent_t const ev_stop = prof ? pe[1] : (xsel ? st.ev_pass : nullptr);
    // the set's previous select (four batches ago) has read the queues this pass refills
    if (xsel && st.sel_recorded) LS_HIP(hipStreamWaitEvent(sm, st.ev_sel, 0));
    if (b.f32) {
        rc = ls_launch_gemm32_filter(ix->d_corpus, ix->n, g, (const float*)st.d_qh, b.nq, b.nq_pad,
                                     st.d_tau, b.nsplits, b.rps, 1, bufs, sm, ev_start, ev_stop);
    } else if (next) {
        ls_index::bc_set& sn = ix->bc_sets[next->set_id];
        ls_gemm_fuse fz;
        fz.d_qh_next = sn.d_qh;
        fz.nq_next = next->nq;
        fz.d_sample_top_next = sn.d_sample_top;
        fz.sample_stride = next->sample_stride;
        rc = ls_launch_gemm_filter(ix->d_corpus, ix->n, g, st.d_qh, b.nq, b.nq_pad, st.d_tau, b.nsplits,
                                   b.rps, 1, bufs, b.top2, sm, &fz, ev_start, ev_stop);
    } else {
        rc = ls_launch_gemm_filter(ix->d_corpus, ix->n, g, st.d_qh, b.nq, b.nq_pad, st.d_tau, b.nsplits,
                                   b.rps, 1, bufs, b.top2, sm, nullptr, ev_start, ev_stop);
    }
    if (rc != LS_OK) return rc;
    ix->n_launches_total++;
    if (prof) {
        if (xsel) LS_HIP(hipEventRecord(st.ev_pass, sm));  // (profiling pass only: its own packet)
        ix->prof_n++;
    }
    if (tau_next) {  // the rider's tau goes in front of this batch's select: it is the lane's critical kernel
        ls_index::bc_set& sn = ix->bc_sets[next->set_id];
        if ((rc = ls_launch_tau(sn.d_sample_top, next->nsplits, next->nq, next->nq_pad, next->jrank,
                                sn.d_tau, sm)) != LS_OK)
            return rc;
        ix->n_launches_total++;
    }
    if (xsel) LS_HIP(hipStreamWaitEvent(ss, st.ev_pass, 0));
    if ((rc = bc_launch_select(ix, b, ss)) != LS_OK) return rc;
    ix->n_launches_total++;
    if (xsel) {
        LS_HIP(hipEventRecord(st.ev_sel, ss));
        st.sel_recorded = true;
    }
    if (st.multi_stream) LS_HIP(hipEventRecord(st.done, ss));
    return LS_OK;
}

// The pipelined batches whose passes are still held back (see above): queue their passes and selects now.
int ls_i_flush_deferred(ls_index* ix) {
    while (!ix->held.empty()) {
        const ls_index::bc_stage h = ix->held.front();
        ix->held.pop_front();
        if (int rc = bc_launch_pass_select(ix, h, nullptr, ix->chain_main[h.lane], ix->chain_sel)) return rc;
    }
    return LS_OK;
}

int ls_i_batched_search_on_stream(ls_index* ix, const float* d_q, int64_t nq, int32_t k,
                                    uint32_t flags, float* d_out_s, int64_t* d_out_i,
                                    hipStream_t s) {
    int rc = ls_i_flush_pending(ix);
    if (rc != LS_OK) return rc;
    const ls_geom& g = ix->g;
    const bool f32 = ix->dtype == LS_DTYPE_F32;
    const int QT = f32 ? 64 : ls_gemm_qt(g);  // queries per workgroup
    const int64_t nq_pad = (nq + QT - 1) / QT * QT;
    const int64_t qkeep_need = nq * g.d;
    if ((int)ix->bc_pending.size() >= ix->bc_slots() ||
        (!ix->bc_pending.empty() &&
         (nq_pad > ix->bc_slot_stride || qkeep_need > ix->bc_qkeep_stride))) {
        rc = ls_i_batched_repair(ix);  // slots exhausted (or too small): check what is pending
        if (rc != LS_OK) return rc;
        ix->n_forced_checks++;
    }
    const bool chain = (flags & LS_FLAG_PIPELINE) != 0;
    if (!chain && (rc = ls_i_flush_deferred(ix)) != LS_OK) return rc;
    const uint64_t seq = chain ? ix->bc_lane_rr++ : 0;
    const int lane = (int)(seq & 1);  // LS_BC_LANES is even: a scratch set always belongs to one lane
    const int set_id = chain ? 1 + (int)(seq % LS_BC_LANES) : 0;
    ls_index::bc_set& st = ix->bc_sets[set_id];
    hipStream_t const caller = s;
    hipStream_t sp = s, sm = s, ss = s;  // prep / sample, tau, pass / select
    if (chain) {
        if (!ix->chain_main[0]) {
            // (two streams of one priority class are given two hardware queues, in creation order;
            // kernel traces show them as queues 3 and 4)
            int least = 0, greatest = 0;
            LS_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
            LS_HIP(hipStreamCreateWithPriority(&ix->chain_main[0], hipStreamNonBlocking, greatest));
            LS_HIP(hipStreamCreateWithPriority(&ix->chain_main[1], hipStreamNonBlocking, greatest));
            // the selects' own stream, in the same class: a lower class is starved for as long as
            // workgroups of a pass are waiting for CUs, which with two lanes is always
            LS_HIP(hipStreamCreateWithPriority(&ix->chain_sel, hipStreamNonBlocking, greatest));
            LS_HIP(hipEventCreateWithFlags(&ix->chain_in, hipEventDisableTiming));
        }
        if (!st.ev_prep) {
            LS_HIP(hipEventCreateWithFlags(&st.ev_prep, hipEventDisableTiming));
            LS_HIP(hipEventCreateWithFlags(&st.ev_pass, hipEventDisableTiming));
            LS_HIP(hipEventCreateWithFlags(&st.ev_sel, hipEventDisableTiming));
        }
        sp = sm = ix->chain_main[lane];
        ss = ix->chain_sel;
    } else if (st.used && st.last_stream != s) {
        // set 0 is shared by plain calls: a call on another stream waits for the previous one
        if (!st.multi_stream) {
            LS_HIP(hipStreamSynchronize(st.last_stream));  // once: no event was recorded yet
            LS_HIP(hipEventCreateWithFlags(&st.done, hipEventDisableTiming));
            st.multi_stream = true;
        } else {
            LS_HIP(hipStreamWaitEvent(s, st.done, 0));
        }
    }
    const bc_plan plan = bc_make_plan(ix, nq, k);
    const int nsplits = plan.nsplits;
    const size_t nrec = (size_t)nq_pad * nsplits;

    // (growing a buffer frees it first: hipFree drains the device, whatever stream still uses it)
    size_t c;
    c = st.qh_cap;
    if ((rc = ls_grow((unsigned char**)&st.d_qh, &c, (size_t)nq_pad * g.d_pad * (f32 ? 4 : 2))) != LS_OK)
        return rc;
    st.qh_cap = c;
    if ((rc = ls_grow(&st.d_queues, &st.queues_cap, nrec * 4 * LS_GEMM_QCAP)) != LS_OK) return rc;
    if ((rc = ls_grow(&st.d_counts, &st.counts_cap, nrec * 4)) != LS_OK) return rc;
    if ((rc = ls_grow(&st.d_tau, &st.tau_cap, (size_t)nq_pad)) != LS_OK) return rc;
    if (ix->bc_pending.empty()) {
        if (nq_pad > ix->bc_slot_stride) ix->bc_slot_stride = nq_pad;
        if (qkeep_need > ix->bc_qkeep_stride) {  // the query copies are re-sliced: drop the old blocks
            ix->bc_qkeep_stride = qkeep_need;
            for (float*& blk : ix->d_qkeep_blk) {
                if (blk) LS_HIP(hipFree(blk));
                blk = nullptr;
            }
        }
    }
    if ((rc = ls_grow(&ix->d_overflow, &ix->overflow_cap,
                   (size_t)ix->bc_slot_stride * ix->bc_slots())) != LS_OK)
        return rc;
    const int slot = (int)ix->bc_pending.size();
    float*& qblk = ix->d_qkeep_blk[slot / LS_BC_QKEEP_BLOCK];
    if (!qblk) {
        // a block of LS_BC_QKEEP_BLOCK slots is 4 GB for nq = 16384, d = 1024: if it does not
        // fit, the backlog is checked (every slot freed) and the call retried by the caller
        if (hipMalloc((void**)&qblk, sizeof(float) * (size_t)ix->bc_qkeep_stride * LS_BC_QKEEP_BLOCK) != hipSuccess) {
            (void)hipGetLastError();
            qblk = nullptr;
            ls_set_error("batched call: out of device memory for the repair copies of %d queued batches "
                         "(%zu bytes per batch): check the pending calls (ls_check) and retry",
                         LS_BC_QKEEP_BLOCK, sizeof(float) * (size_t)ix->bc_qkeep_stride);
            return LS_ERR_HIP;
        }
    }
    u32* d_flags = ix->d_overflow + (size_t)slot * ix->bc_slot_stride;
    float* d_qkeep = qblk + (size_t)(slot % LS_BC_QKEEP_BLOCK) * ix->bc_qkeep_stride;
    if ((rc = ls_grow(&st.d_sample_top, &st.sample_top_cap, nrec * 16)) != LS_OK) return rc;
    if ((rc = ls_grow_pinned(&ix->h_overflow, &ix->h_overflow_cap,
                          (size_t)ix->bc_slot_stride * ix->bc_slots())) != LS_OK)
        return rc;

    ls_index::bc_stage b;
    b.active = true;
    b.lane = lane;
    b.set_id = set_id;
    b.f32 = f32;
    b.nq = nq;
    b.nq_pad = nq_pad;
    b.k = k;
    b.nsplits = nsplits;
    b.sample_stride = plan.sample_stride;
    b.jrank = plan.jrank;
    b.keys_need = plan.keys_need;
    b.rps = plan.rps;
    // two kept sample scores per lane are enough when a query has >= 4 j lanes (two of its best
    // j sample scores then share a lane with probability ~1/8 each)
    b.top2 = LS_GEMM_SAMPLE_TOP2 && (long long)nsplits * 4 >= 4ll * plan.jrank;
    b.d_flags = d_flags;
    b.d_out_s = d_out_s;
    b.d_out_i = d_out_i;
    ls_gemm_bufs bufs;
    bufs.d_queues = st.d_queues;
    bufs.d_counts = st.d_counts;
    bufs.d_overflow = d_flags;
    bufs.d_sample_top = st.d_sample_top;

    // ---- prep -------------------------------------------------------------------------------------
    if (chain) {
        // behind everything the caller has queued so far (its queries) ...
        LS_HIP(hipEventRecord(ix->chain_in, caller));
        LS_HIP(hipStreamWaitEvent(sp, ix->chain_in, 0));
        // (everything else that touches the set is earlier work of this very lane: stream order)
    }
    rc = f32 ? ls_launch_prep_f32(d_q, (float*)st.d_qh, d_qkeep, nq, nq_pad, g,
                                  (flags & LS_FLAG_NORMALIZE) != 0, d_flags, sp)
             : ls_launch_prep_f16(d_q, st.d_qh, d_qkeep, nq, nq_pad, g,
                                  (flags & LS_FLAG_NORMALIZE) != 0, d_flags, sp);
    if (rc != LS_OK) return rc;
    int launches = 1;  // counted, not assumed: debug counter 9
    if (chain) {
        // the prep kernel was the only reader of the caller's query buffer (it also took the
        // repair copy): work the caller queues on its stream from here on may overwrite it
        LS_HIP(hipEventRecord(st.ev_prep, sp));
        LS_HIP(hipStreamWaitEvent(caller, st.ev_prep, 0));
    }
    // ---- sample pass: a few tiles of every slice, spread over the slice ------------------------------
    // Pipelined fp16 batches of one plan: the sample phase rides on the pass launch of the batch TWO
    // calls back, which is held until now for that (the register-starved geometries - config 4's
    // 1.5 KiB rows - spend ~1 % of a multi-millisecond batch between passes and keep their own
    // sample launch)
    const bool fuse_ok = chain && !f32 && ix->opt_fused != 0 && g.chunks <= 48 && ls_gemm_rs(g) == 1;
    bool ride = false;
    if (fuse_ok && ix->held.size() == 2) {
        const ls_index::bc_stage& d = ix->held.front();
        ride = d.lane == lane && d.nq_pad == nq_pad && d.k == k && d.nsplits == nsplits && d.rps == b.rps &&
               d.sample_stride == b.sample_stride && d.jrank == b.jrank && d.top2 == b.top2 &&
               d.set_id != set_id;
    }
    if (ride) {
        const ls_index::bc_stage d = ix->held.front();
        ix->held.pop_front();
        if ((rc = bc_launch_pass_select(ix, d, &b, sm, ss, true)) != LS_OK) return rc;  // + tau(this batch)
    } else {
        // nothing to ride on (the first two batches of a run, another shape, a geometry that does not
        // fuse): a full pipeline is flushed, and the sample pass gets its own launch
        if (!fuse_ok || ix->held.size() == 2) {
            if ((rc = ls_i_flush_deferred(ix)) != LS_OK) return rc;
        }
        rc = f32 ? ls_launch_gemm32_filter(ix->d_corpus, ix->n, g, (const float*)st.d_qh, nq, nq_pad,
                                           nullptr, nsplits, b.rps, b.sample_stride, bufs, sm)
                 : ls_launch_gemm_filter(ix->d_corpus, ix->n, g, st.d_qh, nq, nq_pad, nullptr, nsplits,
                                         b.rps, b.sample_stride, bufs, b.top2, sm);
        if (rc != LS_OK) return rc;
        ++launches;
        if ((rc = ls_launch_tau(st.d_sample_top, nsplits, nq, nq_pad, b.jrank, st.d_tau, sm)) != LS_OK)
            return rc;
        ++launches;
    }
    // ---- pass + select: now, or (pipelined fp16 batches) two calls from now / at the next flush -----
    const bool hold = fuse_ok;
    if (hold) {
        ix->held.push_back(b);
    } else {
        const uint64_t before = ix->n_launches_total;
        if ((rc = bc_launch_pass_select(ix, b, nullptr, sm, ss)) != LS_OK) return rc;
        launches += (int)(ix->n_launches_total - before);
        ix->n_launches_total = before;
    }
    st.used = true;
    st.chain = chain;
    st.last_stream = ss;
    ix->bc_last_set = set_id;
    // kernels of THIS batch (a held-back batch: its pass and select follow; a rider's tau was queued
    // with the carrying pass)
    ix->n_batched_launches = launches + (hold ? 2 : 0) + (ride ? 1 : 0);
    ix->n_launches_total += (uint64_t)launches;
    ix->last_path = f32 ? 3 : 2;
    ix->d_last_flags = d_flags;
    ix->last_flags_n = nq;
    ls_index::batched_call bc;
    bc.nq = nq;
    bc.k = k;
    bc.flags = flags;
    bc.d_out_s = d_out_s;
    bc.d_out_i = d_out_i;
    bc.stream = ss;
    bc.slot = slot;
    ix->bc_pending.push_back(bc);
    if (!(flags & (LS_FLAG_ASYNC | LS_FLAG_PIPELINE))) return ls_i_batched_repair(ix);
    return LS_OK;
}
