// ls_shard.hip — the row-sharded index behind the ordinary handle.
//
// The reference's backend is ONE process that owns one index object and calls index.search from
// its event loop (reference src/lean_explore/mcp/server.py:147-151, search/engine.py:250), so the
// drop-in has to reach every GPU of the node from that one process, with no launcher:
//
//   ls_create_sharded(corpus, n, d, dtype, device_ids[G])  ->  one ls_index* that every entry point
//   of leansearch.h accepts. Shard g = the contiguous row block [g*ceil(n/G), ...) resident on
//   device_ids[g] (a private single-device sub-handle whose `base` is the block's first row).
//
//   one search:  queries -> every device (peer copy over xGMI; the host API hands every device the
//                same pinned buffer)
//                local exact top-k on every device concurrently (one stream per device), written
//                into the shard's packed block [scores f32 | rows i64 | flags u32]
//                ONE exchange step: ncclAllGather of the packed blocks (RCCL over xGMI, one
//                communicator per device from ncclCommInitAll, ncclGroupStart/End)
//                G-way merge under the library's total order on the primary device
//                (device_ids[0]) -> bit-identical to the unsharded answer for every G
//   ls_check:    the shards' verification flags travelled with the results; if any shard had to
//                repair a query of a batched (speculative) call, that call's blocks are exchanged
//                and merged again into the same output buffers.
//
// Duplicate device ids (a rehearsal of G shards on fewer GPUs, e.g. {0,0,0}) cannot form an RCCL
// communicator; they, and handles switched with ls_debug_option(8, 1), exchange by plain
// device-to-device / peer copies into the primary's gather buffer instead. The same copies are the
// FALLBACK when RCCL cannot be bound or ncclCommInitAll / the all-gather fails on the node: the
// handle keeps answering (ls_shard_exchange_info says why), it does not fail every search.
//
// Host side: queueing one shard's work costs ~11-12 us of host time (stream waits, query copy, the
// sub-search's launches); from one thread that is ~90 us for 8 shards, more than the whole 1-GPU
// call. With distinct devices every shard therefore has an ENQUEUE WORKER (a thread bound to that
// device; it spins briefly between back-to-back calls, then sleeps) and the caller's thread only
// posts the jobs, waits, and queues the exchange + merge. Shards that share a device (rehearsals)
// are queued from the caller's thread: one device's queues are serialised by the runtime anyway
// (measured: no gain, profiles/ab/r03_group_host_overhead.txt). ls_debug_option(11, 0/1) forces it.
#include "ls_index.h"

#include <dlfcn.h>
#include <rccl/rccl.h>


#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <thread>

#define LS_SH_SLOTS 8  // batched calls that may be outstanding before the group checks itself
#define LS_SH_SCAN_SLOT LS_SH_SLOTS  // exact (scan-path) calls are ordered by events: one slot

// ---- RCCL, bound at first use ----------------------------------------------------------------------
// librccl.so is half a gigabyte of code objects; single-GPU users of libleansearch never map it.
// dlopen by SONAME: a process that already holds an RCCL (PyTorch bundles one under the same
// SONAME) shares that copy, otherwise the loader follows this library's RUNPATH to /opt/rocm.
namespace {
struct rccl_api {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
rccl_api g_rccl;
std::mutex g_rccl_mu;

int rccl_bind() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.lib) return LS_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        ls_set_error("sharded index: RCCL not found (%s); it is required to exchange shard results "
                     "between distinct devices", dlerror());
        return LS_ERR_NO_DEVICE;
    }
    rccl_api a;
    a.lib = lib;
#define LS_SYM(field, name)                                                  \
    a.field = reinterpret_cast<decltype(a.field)>(dlsym(lib, name));        \
    if (!a.field) {                                                          \
        ls_set_error("sharded index: librccl lacks %s", name);               \
        return LS_ERR_NO_DEVICE;                                             \
    }
    LS_SYM(CommInitAll, "ncclCommInitAll")
    LS_SYM(CommDestroy, "ncclCommDestroy")
    LS_SYM(GroupStart, "ncclGroupStart")
    LS_SYM(GroupEnd, "ncclGroupEnd")
    LS_SYM(AllGather, "ncclAllGather")
    LS_SYM(Send, "ncclSend")
    LS_SYM(Recv, "ncclRecv")
    LS_SYM(GetErrorString, "ncclGetErrorString")
    LS_SYM(GetVersion, "ncclGetVersion")
#undef LS_SYM
    g_rccl = a;
    return LS_OK;
}
}  // namespace

#define LS_NCCL(call)                                                                        \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            ls_set_error("%s failed: %s (%s:%d)", #call, g_rccl.GetErrorString(r_), __FILE__, \
                         __LINE__);                                                          \
            return LS_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

// Restores the calling thread's current HIP device on every exit path: the group entry points hop
// over the shards' devices, and PyTorch shares this per-thread state with us.
struct ls_device_guard {
    int prev = -1;
    ls_device_guard() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~ls_device_guard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// One enqueue worker per shard (see the file header). post() hands it a job; wait() returns the
// job's return code (the worker's thread-local error text is copied into `err`).
struct ls_shard_worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    std::atomic<uint32_t> posted{0}, done{0};
    bool stop = false;
    int device = 0;
    int rc = LS_OK;
    char err[256] = "";

    void run() {
        (void)hipSetDevice(device);
        uint32_t seen = 0;
        for (;;) {
            bool got = false;
            for (int i = 0; i < 3000 && !got; ++i) {  // calls come back to back: spin ~50 us first
                if (posted.load(std::memory_order_acquire) != seen) got = true;
                else ls_cpu_relax();
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || posted.load(std::memory_order_acquire) != seen; });
            }
            if (posted.load(std::memory_order_acquire) == seen) {
                if (stop) return;
                continue;
            }
            seen = posted.load(std::memory_order_acquire);
            rc = job();
            if (rc != LS_OK) snprintf(err, sizeof(err), "%s", ls_last_error());
            done.store(seen, std::memory_order_release);
        }
    }
    void post(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j);
            posted.fetch_add(1, std::memory_order_release);
        }
        cv.notify_one();
    }
    int wait() {  // the job is ~10 us of launches: spinning is cheaper than a wake-up
        const uint32_t want = posted.load(std::memory_order_relaxed);
        for (unsigned it = 0; done.load(std::memory_order_acquire) != want; ++it) {
            if ((it & 4095) == 4095) std::this_thread::yield();
            else ls_cpu_relax();
        }
        return rc;
    }
    void shutdown() {
        if (!th.joinable()) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
        }
        cv.notify_one();
        th.join();
    }
};

struct ls_shard_group {
    int G = 0;
    std::vector<ls_index*> sub;  // one single-device handle per shard
    std::vector<int> dev;        // device ordinal of shard g
    std::vector<int64_t> lo;     // first global row of shard g (relative to the group's base)
    bool replicated = false;     // every sub-handle holds the WHOLE corpus (ls_create_replicated)
    std::atomic<bool> failed{false};  // replicated: an add reached some replicas only - the handle refuses further work
    std::atomic<uint32_t> rr{0}; // replicated: the next synchronous host call goes to replica rr % G
    bool distinct = false;       // all device ids differ (an RCCL communicator can be formed)
    int exchange_mode = 0;       // 0: RCCL all-gather when `distinct`; 1: peer copies to the primary;
                                 // 2: RCCL gather-to-root (ncclSend / ncclRecv in one group) when `distinct`
    std::vector<ncclComm_t> comms;
    bool comms_ready = false;
    bool debug_fail_rccl = false;  // test hook (debug option 12): pretend the collective failed
    bool rccl_failed = false;    // RCCL could not be used on this node: the handle fell back to copies
    char rccl_error[256] = "";
    int rccl_version = 0;
    int opt_workers = -1;        // -1 auto (on when the device ids are distinct), 0 off, 1 on
    std::vector<ls_shard_worker*> workers;
    std::vector<int> peer;       // G x G: hipDeviceCanAccessPeer(dev[a], dev[b]) as seen at creation (-1: same device)

    struct buf {
        char* p = nullptr;
        size_t cap = 0;
    };
    struct shard {
        hipStream_t stream = nullptr;
        hipEvent_t ev_x = nullptr;        // this shard's block has reached the primary (copy mode)
        float* d_q = nullptr;             // this device's copy of the queries
        size_t q_cap = 0;
        buf packed[LS_SH_SLOTS + 1];      // [scores | rows | flags] of this shard, per slot
        buf gathered[LS_SH_SLOTS + 1];    // G blocks (RCCL: on every device; copy mode: primary only)
    };
    std::vector<shard> sh;
    hipEvent_t ev_in = nullptr;   // the caller's stream has produced the queries
    hipEvent_t ev_out = nullptr;  // the primary's stream has merged the previous call
    bool have_out = false;
    buf tmp_s[2], tmp_i[2];       // tree merge (n_shards * k beyond one merge launch)

    struct call {
        int slot;
        int64_t nq;
        int32_t k;
        size_t block, sbytes, rbytes;
        float* dst_s;
        int64_t* dst_i;
    };
    std::vector<call> pending;  // batched calls whose flags have not been looked at yet

    // host API staging (portable pinned: every device reads the same query buffer)
    float* h_q = nullptr;      size_t h_q_cap = 0;
    float* h_out_s = nullptr;  size_t h_out_s_cap = 0;
    int64_t* h_out_i = nullptr; size_t h_out_i_cap = 0;
    float* d_out_s = nullptr;  size_t d_out_s_cap = 0;
    int64_t* d_out_i = nullptr; size_t d_out_i_cap = 0;

    uint64_t repairs_seen = 0;  // sum of the shards' repair counters when the last check finished
    uint64_t n_exchanges = 0, n_reexchanges = 0, n_worker_calls = 0;
    uint64_t enqueue_ns = 0, enqueue_calls = 0;  // host time spent queueing searches (debug counter 18)
};

static inline bool group_uses_rccl(const ls_shard_group* G) {
    return G->distinct && (G->exchange_mode == 0 || G->exchange_mode == 2);
}

static int group_grow(ls_shard_group::buf* b, size_t need) { return ls_grow(&b->p, &b->cap, need); }

static int group_init_comms(ls_shard_group* G) {
    if (G->comms_ready) return LS_OK;
    int rc = rccl_bind();
    if (rc != LS_OK) return rc;
    (void)g_rccl.GetVersion(&G->rccl_version);
    G->comms.assign(G->G, nullptr);
    LS_NCCL(g_rccl.CommInitAll(G->comms.data(), G->G, G->dev.data()));
    G->comms_ready = true;
    return LS_OK;
}

// One exchange step of slot `slot`: afterwards (in the order of the primary's stream) the primary's
// gather buffer holds the G packed blocks, block g at offset g * block.
static int group_exchange_rccl(ls_shard_group* G, int slot, size_t block) {
    int rc;
    if ((rc = group_init_comms(G)) != LS_OK) return rc;
    for (int g = 0; g < G->G; ++g) {
        LS_HIP(hipSetDevice(G->dev[g]));
        if ((rc = group_grow(&G->sh[g].gathered[slot], block * G->G)) != LS_OK) return rc;
    }
    LS_NCCL(g_rccl.GroupStart());
    for (int g = 0; g < G->G; ++g) {
        ncclResult_t r = g_rccl.AllGather(G->sh[g].packed[slot].p, G->sh[g].gathered[slot].p,
                                          block, ncclUint8, G->comms[g], G->sh[g].stream);
        if (r != ncclSuccess) {
            (void)g_rccl.GroupEnd();
            ls_set_error("ncclAllGather failed: %s", g_rccl.GetErrorString(r));
            return LS_ERR_HIP;
        }
    }
    LS_NCCL(g_rccl.GroupEnd());
    return LS_OK;
}

// The same exchange as a GATHER TO ROOT: only the primary merges (one process, one merge - unlike the
// one-process-per-GPU model of sharded.py, where every rank needs every block), so only the primary needs
// the G blocks. ncclSend / ncclRecv in one group: every other shard sends its block once, nothing grows
// on the non-primary devices (the all-gather keeps G x the buffer and moves G x the bytes: 98 MB per rank
// and call at nq = 1024, k = 1000, G = 8). north_star names the all-gather, which stays the default;
// ls_debug_option(8, 2) selects this one, ls_shard_exchange_info names the collective in use.
static int group_exchange_rccl_gather(ls_shard_group* G, int slot, size_t block) {
    int rc;
    if ((rc = group_init_comms(G)) != LS_OK) return rc;
    LS_HIP(hipSetDevice(G->dev[0]));
    if ((rc = group_grow(&G->sh[0].gathered[slot], block * G->G)) != LS_OK) return rc;
    char* dst = G->sh[0].gathered[slot].p;
    LS_HIP(hipMemcpyAsync(dst, G->sh[0].packed[slot].p, block, hipMemcpyDeviceToDevice, G->sh[0].stream));
    if (G->G == 1) return LS_OK;
    LS_NCCL(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int g = 1; g < G->G && r == ncclSuccess; ++g)
        r = g_rccl.Recv(dst + (size_t)g * block, block, ncclUint8, g, G->comms[0], G->sh[0].stream);
    for (int g = 1; g < G->G && r == ncclSuccess; ++g)
        r = g_rccl.Send(G->sh[g].packed[slot].p, block, ncclUint8, 0, G->comms[g], G->sh[g].stream);
    if (r != ncclSuccess) {
        (void)g_rccl.GroupEnd();
        ls_set_error("ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(r));
        return LS_ERR_HIP;
    }
    LS_NCCL(g_rccl.GroupEnd());
    return LS_OK;
}

static int group_exchange(ls_shard_group* G, int slot, size_t block) {
    int rc;
    G->n_exchanges++;
    if (group_uses_rccl(G)) {
        if (G->debug_fail_rccl) {
            ls_set_error("RCCL failure injected by ls_debug_option(12, 1)");
            rc = LS_ERR_HIP;
        } else {
            rc = G->exchange_mode == 2 ? group_exchange_rccl_gather(G, slot, block)
                                       : group_exchange_rccl(G, slot, block);
        }
        if (rc == LS_OK) return LS_OK;
        // RCCL is not usable here (library missing, ncclCommInitAll or the collective failed): the
        // peer-copy exchange below gives the same gather buffer on the primary. Keep answering;
        // ls_shard_exchange_info / debug counter 15 (= 3) report what happened.
        G->rccl_failed = true;
        snprintf(G->rccl_error, sizeof(G->rccl_error), "%s", ls_last_error());
        G->exchange_mode = 1;
    }
    // copy mode: every shard writes its block into the primary's gather buffer (a peer write over
    // xGMI, or a plain device-to-device copy when the shards share a device)
    LS_HIP(hipSetDevice(G->dev[0]));
    if ((rc = group_grow(&G->sh[0].gathered[slot], block * G->G)) != LS_OK) return rc;
    char* dst = G->sh[0].gathered[slot].p;
    for (int g = 0; g < G->G; ++g) {
        LS_HIP(hipSetDevice(G->dev[g]));
        ls_shard_group::shard& S = G->sh[g];
        if (G->dev[g] == G->dev[0])
            LS_HIP(hipMemcpyAsync(dst + (size_t)g * block, S.packed[slot].p, block,
                                  hipMemcpyDeviceToDevice, S.stream));
        else
            LS_HIP(hipMemcpyPeerAsync(dst + (size_t)g * block, G->dev[0], S.packed[slot].p, G->dev[g],
                                      block, S.stream));
        if (g > 0) LS_HIP(hipEventRecord(S.ev_x, S.stream));
    }
    LS_HIP(hipSetDevice(G->dev[0]));
    for (int g = 1; g < G->G; ++g) LS_HIP(hipStreamWaitEvent(G->sh[0].stream, G->sh[g].ev_x, 0));
    return LS_OK;
}

// G-way merge of the gathered blocks on the primary's stream. One launch holds n_lists * k keys in
// LDS; beyond that (8 shards x k = 2048) the lists are merged in rounds of `fan`.
static int group_merge(ls_shard_group* G, int slot, size_t block, size_t sbytes, int64_t nq,
                       int32_t k, float* dst_s, int64_t* dst_i) {
    hipStream_t s0 = G->sh[0].stream;
    const char* base = G->sh[0].gathered[slot].p;
    int lists = G->G;
    if ((long long)lists * k <= LS_FINAL_CAP)
        return ls_launch_merge((const float*)base, (const int64_t*)(base + sbytes), (int64_t)block,
                               (int64_t)block, lists, nq, k, dst_s, dst_i, s0);
    const int fan = std::max(2, LS_FINAL_CAP / k);
    const float* in_s = (const float*)base;
    const int64_t* in_i = (const int64_t*)(base + sbytes);
    int64_t stride_s = (int64_t)block, stride_i = (int64_t)block;
    int rc;
    for (int round = 0; lists > 1; ++round) {
        const int out_lists = (lists + fan - 1) / fan;
        float* o_s = dst_s;
        int64_t* o_i = dst_i;
        if (out_lists > 1) {
            ls_shard_group::buf& bs = G->tmp_s[round & 1];
            ls_shard_group::buf& bi = G->tmp_i[round & 1];
            if ((rc = group_grow(&bs, (size_t)out_lists * nq * k * sizeof(float))) != LS_OK) return rc;
            if ((rc = group_grow(&bi, (size_t)out_lists * nq * k * sizeof(int64_t))) != LS_OK) return rc;
            o_s = (float*)bs.p;
            o_i = (int64_t*)bi.p;
        }
        for (int l = 0; l < out_lists; ++l) {
            const int first = l * fan, cnt = std::min(fan, lists - first);
            rc = ls_launch_merge((const float*)((const char*)in_s + (size_t)first * stride_s),
                                 (const int64_t*)((const char*)in_i + (size_t)first * stride_i),
                                 stride_s, stride_i, cnt, nq, k, o_s + (size_t)l * nq * k,
                                 o_i + (size_t)l * nq * k, s0);
            if (rc != LS_OK) return rc;
        }
        in_s = o_s;
        in_i = o_i;
        stride_s = nq * k * (int64_t)sizeof(float);
        stride_i = nq * k * (int64_t)sizeof(int64_t);
        lists = out_lists;
    }
    return LS_OK;
}

static int group_start_workers(ls_shard_group* G) {
    if (!G->workers.empty()) return LS_OK;
    G->workers.assign(G->G, nullptr);
    for (int g = 1; g < G->G; ++g) {  // shard 0 is queued from the calling thread
        ls_shard_worker* w = new (std::nothrow) ls_shard_worker();
        bool ok = w != nullptr;
        if (ok) {
            w->device = G->dev[g];
            try {  // (std::thread may throw std::system_error: nothing crosses the C ABI)
                w->th = std::thread([w] { w->run(); });
            } catch (...) {
                ok = false;
            }
        }
        if (!ok) {  // no half-built worker set: the next call starts over
            delete w;
            for (ls_shard_worker* o : G->workers) {
                if (!o) continue;
                o->shutdown();
                delete o;
            }
            G->workers.clear();
            ls_set_error("sharded index: cannot start the enqueue worker of shard %d", g);
            return LS_ERR_INVALID_ARG;
        }
        G->workers[g] = w;
    }
    return LS_OK;
}
static void group_stop_workers(ls_shard_group* G) {
    for (ls_shard_worker* w : G->workers) {
        if (!w) continue;
        w->shutdown();
        delete w;
    }
    G->workers.clear();
}

static uint64_t group_repairs(const ls_shard_group* G) {
    uint64_t t = 0;
    for (const ls_index* s : G->sub) t += s->n_batched_fallback;
    return t;
}

// Make every outstanding call final (see the file header). Called with the group's mutex held.
static int group_check_locked(ls_index* ix) {
    ls_shard_group* G = ix->group;
    int rc;
    // the exchanges read the packed blocks that a repair rewrites in place: drain them first
    for (int g = 0; g < G->G; ++g) {
        LS_HIP(hipSetDevice(G->dev[g]));
        LS_HIP(hipStreamSynchronize(G->sh[g].stream));
    }
    // (compared with the count at the END of the previous check, not with "now": a sub-handle may
    // already have repaired calls that are still unchecked here, e.g. when a bigger batch made it
    // re-slice its flag slots while earlier asynchronous calls were outstanding)
    const uint64_t before = G->repairs_seen;
    for (int g = 0; g < G->G; ++g) {
        LS_HIP(hipSetDevice(G->dev[g]));
        if ((rc = ls_i_flush_pending(G->sub[g])) != LS_OK) return rc;
        if ((rc = ls_i_batched_repair(G->sub[g])) != LS_OK) return rc;  // synchronises if it repaired
    }
    if (group_repairs(G) != before && !G->pending.empty()) {
        // some shard re-ran a query through its exact path: its packed rows changed. Which call it
        // belonged to is the sub-handle's business; re-merging every unchecked call is cheap
        // (tens of microseconds each) next to how rare this is.
        G->n_reexchanges++;
        for (const ls_shard_group::call& c : G->pending) {
            for (int g = 0; g < G->G; ++g) {
                LS_HIP(hipSetDevice(G->dev[g]));
                LS_HIP(hipMemsetAsync(G->sh[g].packed[c.slot].p + c.sbytes + c.rbytes, 0,
                                      c.block - c.sbytes - c.rbytes, G->sh[g].stream));
            }
            if ((rc = group_exchange(G, c.slot, c.block)) != LS_OK) return rc;
            LS_HIP(hipSetDevice(G->dev[0]));
            if ((rc = group_merge(G, c.slot, c.block, c.sbytes, c.nq, c.k, c.dst_s, c.dst_i)) != LS_OK)
                return rc;
        }
        for (int g = 0; g < G->G; ++g) {
            LS_HIP(hipSetDevice(G->dev[g]));
            LS_HIP(hipStreamSynchronize(G->sh[g].stream));
        }
    }
    G->pending.clear();
    G->repairs_seen = group_repairs(G);
    LS_HIP(hipSetDevice(G->dev[0]));
    return LS_OK;
}

int ls_group_check(ls_index* ix, hipStream_t s) {
    ls_device_guard guard;
    if (ix->group->replicated) return ls_check(ix->group->sub[0], s);
    int rc = group_check_locked(ix);
    if (rc != LS_OK) return rc;
    LS_HIP(hipStreamSynchronize(s));
    return LS_OK;
}

int ls_group_search(ls_index* ix, const float* q, bool q_on_host, int64_t nq, int32_t k,
                    uint32_t flags, float* out_s, int64_t* out_i, hipStream_t s) {
    ls_device_guard guard;  // every exit (errors included) leaves the caller's device current
    ls_shard_group* G = ix->group;
    if (G->replicated) {
        // device-resident queries live on device_ids[0]: the replica there serves them (the other
        // replicas take synchronous host calls, ls_replica_search)
        ls_index* sub = G->sub[0];
        if (q_on_host) return ls_search(sub, q, nq, k, flags, out_s, out_i);
        return ls_search_device(sub, q, nq, k, flags, out_s, out_i, s);
    }
    const int P = G->dev[0];
    const int32_t d = ix->g.d;
    const size_t qn = (size_t)nq * d, on = (size_t)nq * k;
    int rc;
    const auto t_enq = std::chrono::steady_clock::now();
    bool batched = false;  // does any shard answer speculatively (flags may come back non-zero)?
    for (int g = 0; g < G->G; ++g) batched = batched || ls_i_batched_eligible(G->sub[g], nq, k);
    if (batched && (int)G->pending.size() >= LS_SH_SLOTS) {
        if ((rc = group_check_locked(ix)) != LS_OK) return rc;
    }
    const int slot = batched ? (int)G->pending.size() : LS_SH_SCAN_SLOT;
    // one packed block per shard: [scores f32 nq*k | pad to 8 B][rows i64 nq*k][flags u32 nq | pad]
    const size_t sbytes = (on * 4 + 7) & ~(size_t)7, rbytes = on * 8;
    const size_t block = sbytes + rbytes + (((size_t)nq * 4 + 7) & ~(size_t)7);

    LS_HIP(hipSetDevice(P));
    hipStream_t s0 = G->sh[0].stream;
    // kernels read the pinned buffer themselves - single queries only: every scan workgroup reads the whole
    // query block, 256 x 16 x 4 KB over PCIe per shard otherwise (copied with one command instead)
    const bool q_direct = q_on_host && nq == 1;
    const bool out_direct = q_on_host && on <= (size_t)(1 << 16);
    float* dst_s = out_s;
    int64_t* dst_i = out_i;
    if (q_on_host) {
        if ((rc = ls_grow_pinned(&G->h_q, &G->h_q_cap, qn, hipHostMallocPortable)) != LS_OK) return rc;
        if ((rc = ls_grow_pinned(&G->h_out_s, &G->h_out_s_cap, on, hipHostMallocPortable)) != LS_OK)
            return rc;
        if ((rc = ls_grow_pinned(&G->h_out_i, &G->h_out_i_cap, on, hipHostMallocPortable)) != LS_OK)
            return rc;
        memcpy(G->h_q, q, qn * sizeof(float));
        if (out_direct) {
            dst_s = G->h_out_s;
            dst_i = G->h_out_i;
        } else {
            if ((rc = ls_grow(&G->d_out_s, &G->d_out_s_cap, on)) != LS_OK) return rc;
            if ((rc = ls_grow(&G->d_out_i, &G->d_out_i_cap, on)) != LS_OK) return rc;
            dst_s = G->d_out_s;
            dst_i = G->d_out_i;
        }
    } else {
        LS_HIP(hipEventRecord(G->ev_in, s));
    }
    // one shard's share of the call, queued on that shard's stream (runs on the shard's enqueue
    // worker when the handle has them, else right here)
    auto enqueue_shard = [&](int g) -> int {
        int rc2;
        ls_shard_group::shard& S = G->sh[g];
        ls_index* sub = G->sub[g];
        LS_HIP(hipSetDevice(G->dev[g]));
        if (!q_on_host) LS_HIP(hipStreamWaitEvent(S.stream, G->ev_in, 0));
        // the previous call's merge has read the gather buffer this call's exchange overwrites
        if (G->have_out && g > 0) LS_HIP(hipStreamWaitEvent(S.stream, G->ev_out, 0));
        const float* qg = q;
        if (q_direct) {
            qg = G->h_q;
        } else if (q_on_host) {
            if ((rc2 = ls_grow(&S.d_q, &S.q_cap, qn)) != LS_OK) return rc2;
            LS_HIP(hipMemcpyAsync(S.d_q, G->h_q, qn * sizeof(float), hipMemcpyHostToDevice, S.stream));
            qg = S.d_q;
        } else if (G->dev[g] != P) {
            if ((rc2 = ls_grow(&S.d_q, &S.q_cap, qn)) != LS_OK) return rc2;
            LS_HIP(hipMemcpyPeerAsync(S.d_q, G->dev[g], q, P, qn * sizeof(float), S.stream));
            qg = S.d_q;
        }
        if ((rc2 = group_grow(&S.packed[slot], block)) != LS_OK) return rc2;
        char* pk = S.packed[slot].p;
        // Sub-searches are always queued asynchronously. LS_FLAG_PIPELINE reaches the batched path
        // only (its internal streams); on the scan path it would leave the last selection step
        // pending in the sub-handle, and the exchange needs the shard's rows now.
        const bool sub_batched = ls_i_batched_eligible(sub, nq, k);
        const uint32_t f = (flags & LS_FLAG_NORMALIZE) | LS_FLAG_ASYNC |
                           ((sub_batched && !q_on_host) ? (flags & LS_FLAG_PIPELINE) : 0u);
        rc2 = ls_i_search_on_stream(sub, qg, nq, k, f, (float*)pk, (int64_t*)(pk + sbytes), S.stream,
                                    false);
        if (rc2 != LS_OK) return rc2;
        // orders S.stream behind the sub-handle's internal streams (and ships the flags: whether a
        // call is re-merged is decided from the shards' repair counters, group_check_locked)
        if (batched && (rc2 = ls_i_export_flags(sub, pk + sbytes + rbytes, nq, S.stream)) != LS_OK)
            return rc2;
        return LS_OK;
    };
    const bool use_workers = G->G > 1 && (G->opt_workers == 1 || (G->opt_workers < 0 && G->distinct));
    if (use_workers) {
        if ((rc = group_start_workers(G)) != LS_OK) return rc;
        for (int g = 1; g < G->G; ++g) G->workers[g]->post([&enqueue_shard, g] { return enqueue_shard(g); });
        rc = enqueue_shard(0);  // the primary's share from this thread, concurrently with the workers
        for (int g = 1; g < G->G; ++g) {
            const int rg = G->workers[g]->wait();
            if (rg != LS_OK && rc == LS_OK) {
                rc = rg;
                ls_set_error("%s", G->workers[g]->err);
            }
        }
        if (rc != LS_OK) return rc;
        G->n_worker_calls++;
    } else {
        for (int g = 0; g < G->G; ++g)
            if ((rc = enqueue_shard(g)) != LS_OK) return rc;
    }
    if ((rc = group_exchange(G, slot, block)) != LS_OK) return rc;
    LS_HIP(hipSetDevice(P));
    if ((rc = group_merge(G, slot, block, sbytes, nq, k, dst_s, dst_i)) != LS_OK) return rc;
    LS_HIP(hipEventRecord(G->ev_out, s0));
    G->have_out = true;
    if (batched) G->pending.push_back({slot, nq, k, block, sbytes, rbytes, dst_s, dst_i});
    G->enqueue_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                         std::chrono::steady_clock::now() - t_enq).count();
    G->enqueue_calls++;
    if (q_on_host) {
        if (batched) {
            if ((rc = group_check_locked(ix)) != LS_OK) return rc;  // drains, repairs, re-merges
        } else {
            LS_HIP(hipStreamSynchronize(s0));
        }
        if (!out_direct) {
            LS_HIP(hipMemcpy(G->h_out_s, G->d_out_s, on * sizeof(float), hipMemcpyDeviceToHost));
            LS_HIP(hipMemcpy(G->h_out_i, G->d_out_i, on * sizeof(int64_t), hipMemcpyDeviceToHost));
        }
        memcpy(out_s, G->h_out_s, on * sizeof(float));
        memcpy(out_i, G->h_out_i, on * sizeof(int64_t));
        return LS_OK;
    }
    // results are ordered on the caller's stream like any other work queued there
    LS_HIP(hipStreamWaitEvent(s, G->ev_out, 0));
    if (!(flags & (LS_FLAG_ASYNC | LS_FLAG_PIPELINE))) {
        if (batched) {
            if ((rc = group_check_locked(ix)) != LS_OK) return rc;
        }
        LS_HIP(hipStreamSynchronize(s));
    }
    return LS_OK;
}

void ls_group_destroy(ls_index* ix) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    if (!G) return;
    group_stop_workers(G);
    for (int g = 0; g < (int)G->sh.size(); ++g) {
        (void)hipSetDevice(G->dev[g]);
        ls_shard_group::shard& S = G->sh[g];
        if (S.stream) (void)hipStreamSynchronize(S.stream);
    }
    if (G->comms_ready)
        for (ncclComm_t c : G->comms)
            if (c) (void)g_rccl.CommDestroy(c);
    for (int g = 0; g < (int)G->sh.size(); ++g) {
        (void)hipSetDevice(G->dev[g]);
        ls_shard_group::shard& S = G->sh[g];
        (void)hipFree(S.d_q);
        for (auto& b : S.packed) (void)hipFree(b.p);
        for (auto& b : S.gathered) (void)hipFree(b.p);
        if (S.ev_x) (void)hipEventDestroy(S.ev_x);
        if (S.stream) (void)hipStreamDestroy(S.stream);
    }
    for (ls_index* s : G->sub) ls_destroy(s);
    if (!G->dev.empty()) (void)hipSetDevice(G->dev[0]);
    for (auto& b : G->tmp_s) (void)hipFree(b.p);
    for (auto& b : G->tmp_i) (void)hipFree(b.p);
    (void)hipFree(G->d_out_s);
    (void)hipFree(G->d_out_i);
    if (G->h_q) (void)hipHostFree(G->h_q);
    if (G->h_out_s) (void)hipHostFree(G->h_out_s);
    if (G->h_out_i) (void)hipHostFree(G->h_out_i);
    if (G->ev_in) (void)hipEventDestroy(G->ev_in);
    if (G->ev_out) (void)hipEventDestroy(G->ev_out);
    delete G;
    ix->group = nullptr;
}

int ls_group_set_base(ls_index* ix, int64_t base) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    ix->base = base;
    for (int g = 0; g < G->G; ++g) {
        std::lock_guard<std::mutex> lk(G->sub[g]->mu);
        G->sub[g]->base = base + G->lo[g];
    }
    return LS_OK;
}

// index.add on a sharded index: the new rows extend the LAST shard (row blocks stay contiguous and
// every global row keeps its number; rebuild the index to rebalance).
int ls_group_add(ls_index* ix, const float* rows, int64_t n_add) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    if (ix->n + n_add >= 0xffffffffll) {  // result keys and the merge carry 32-bit GLOBAL rows
        ls_set_error("ls_add: %lld rows exceed the 2^32-1 rows the result keys can index",
                     (long long)(ix->n + n_add));
        return LS_ERR_INVALID_ARG;
    }
    if (G->replicated) {
        // every replica appends the same rows. A failure part-way would leave replicas of different lengths
        // behind a round-robin dispatcher: the handle is then marked failed and refuses searches
        // and further adds instead of answering differently from call to call.
        if (G->failed) {
            ls_set_error("ls_add: an earlier add left the replicas inconsistent; recreate the index");
            return LS_ERR_INVALID_ARG;
        }
        for (size_t r = 0; r < G->sub.size(); ++r) {
            if (int rc = ls_add(G->sub[r], rows, n_add)) {
                if (r > 0) G->failed = true;
                return rc;
            }
        }
        ix->n += n_add;
        return LS_OK;
    }
    int rc = group_check_locked(ix);
    if (rc != LS_OK) return rc;
    rc = ls_add(G->sub[G->G - 1], rows, n_add);
    if (rc == LS_OK) ix->n += n_add;
    return rc;
}

int ls_group_reconstruct(ls_index* ix, int64_t row0, int64_t count, float* out) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    if (G->replicated) return ls_reconstruct(G->sub[0], row0, count, out);
    for (int g = 0; g < G->G && count > 0; ++g) {
        const int64_t lo = G->lo[g], hi = lo + G->sub[g]->n;
        if (row0 >= hi || row0 + count <= lo) continue;
        const int64_t a = std::max(row0, lo), b = std::min(row0 + count, hi);
        int rc = ls_reconstruct(G->sub[g], a - lo, b - a, out + (a - row0) * (int64_t)ix->g.d);
        if (rc != LS_OK) return rc;
    }
    return LS_OK;
}

int ls_group_debug_option(ls_index* ix, int32_t which, int32_t value) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    if (G->replicated && (which == 8 || which == 11 || which == 12)) {
        ls_set_error("ls_debug_option(%d): a replicated handle has no exchange step", which);
        return LS_ERR_INVALID_ARG;
    }
    if (which == 8) {  // 0: RCCL all-gather between distinct devices (default); 1: peer copies; 2: RCCL gather to the primary
        if (value < 0 || value > 2) {
            ls_set_error("ls_debug_option(8): exchange mode must be 0 (RCCL all-gather), 1 (peer copies) or 2 (RCCL gather-to-root)");
            return LS_ERR_INVALID_ARG;
        }
        int rc = group_check_locked(ix);
        if (rc != LS_OK) return rc;
        G->exchange_mode = value;
        return LS_OK;
    }
    if (which == 11) {  // per-shard enqueue workers: -1 auto (distinct devices), 0 off, 1 on
        G->opt_workers = value < 0 ? -1 : (value ? 1 : 0);
        return LS_OK;
    }
    if (which == 12) {  // test hook: the next RCCL exchange fails (exercises the copy fallback)
        G->debug_fail_rccl = value != 0;
        return LS_OK;
    }
    for (ls_index* s : G->sub) {
        int rc = ls_debug_option(s, which, value);
        if (rc != LS_OK) return rc;
    }
    return LS_OK;
}

int64_t ls_group_debug_counter(ls_index* ix, int32_t which) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    if (which == 21) return (int64_t)G->rr.load();  // replicated: host calls dispatched so far
    if (which == 18) return G->enqueue_calls ? (int64_t)(G->enqueue_ns / G->enqueue_calls) : 0;
    if (which == 13) return (int64_t)G->n_exchanges;
    if (which == 14) return (int64_t)G->n_reexchanges;
    // 0: copies (shards share a device, or debug option 8); 1: RCCL selected, not yet used;
    // 2: RCCL communicators initialised and in use; 3: RCCL failed on this node, fell back to copies
    if (which == 15) return G->rccl_failed ? 3 : (group_uses_rccl(G) ? (G->comms_ready ? 2 : 1) : 0);
    if (which == 19) return (int64_t)G->n_worker_calls;
    if (which == 9 || which == 10) return ls_debug_counter(G->sub[0], which);
    int64_t t = 0;
    for (ls_index* s : G->sub) {
        const int64_t v = ls_debug_counter(s, which);
        if (v < 0) return v;
        t += v;
    }
    return t;
}

// Kernel timing is the primary shard's (every shard runs the same kernels on an equal row block).
int ls_group_set_profiling(ls_index* ix, int32_t enabled) {
    ls_device_guard guard;
    return ls_set_profiling(ix->group->sub[0], enabled);
}
int ls_group_last_kernel_ms(ls_index* ix, float* scan_ms, float* total_ms) {
    ls_device_guard guard;
    return ls_last_kernel_ms(ix->group->sub[0], scan_ms, total_ms);
}

// One synchronous host call on a replicated handle: the next replica in turn takes it (its own
// combining queue serves callers that arrive together on that replica). No group lock is taken.
int ls_replica_search(ls_index* ix, const float* q, int64_t nq, int32_t k, uint32_t flags,
                      float* out_s, int64_t* out_i) {
    ls_device_guard guard;
    ls_shard_group* G = ix->group;
    if (G->failed.load(std::memory_order_acquire)) {
        ls_set_error("ls_search: an add left the replicas inconsistent; recreate the index");
        return LS_ERR_INVALID_ARG;
    }
    const uint32_t r = G->rr.fetch_add(1, std::memory_order_relaxed) % (uint32_t)G->G;
    return ls_search(G->sub[r], q, nq, k, flags, out_s, out_i);
}
bool ls_group_is_replicated(const ls_index* ix) { return ix->group && ix->group->replicated; }

// ---- construction ----------------------------------------------------------------------------------
static int group_begin(ls_index** out, int64_t n, int32_t d, int32_t dtype, const int32_t* device_ids,
                       int32_t n_devices, const char* who, ls_index** pix) {
    if (!out) {
        ls_set_error("%s: out is null", who);
        return LS_ERR_INVALID_ARG;
    }
    *out = nullptr;
    if (n_devices <= 0 || !device_ids) {
        ls_set_error("%s: needs at least one device id (libleansearch has no CPU path)", who);
        return n_devices == 0 ? LS_ERR_NO_DEVICE : LS_ERR_INVALID_ARG;
    }
    if (n_devices > 64) {
        ls_set_error("%s: at most 64 shards", who);
        return LS_ERR_INVALID_ARG;
    }
    if (n < 0 || d <= 0) {
        ls_set_error("%s: bad shape n=%lld d=%d", who, (long long)n, d);
        return LS_ERR_INVALID_ARG;
    }
    if (n >= 0xffffffffll) {
        // result keys carry 32-bit rows; the merge works on GLOBAL rows
        ls_set_error("%s: n=%lld exceeds the 2^32-1 rows the result keys can index", who, (long long)n);
        return LS_ERR_INVALID_ARG;
    }
    ls_geom geom;
    if (ls_pick_geom(d, dtype, &geom) != LS_OK) {
        ls_set_error("%s: unsupported d=%d / dtype=%d (max stored row is 4096 bytes)", who, d, dtype);
        return LS_ERR_INVALID_ARG;
    }
    for (int g = 0; g < n_devices; ++g) {
        int rc = ls_i_check_device(device_ids[g]);
        if (rc != LS_OK) return rc;
    }
    ls_index* ix = new (std::nothrow) ls_index();
    ls_shard_group* G = new (std::nothrow) ls_shard_group();
    if (!ix || !G) {
        delete ix;
        delete G;
        ls_set_error("%s: out of host memory", who);
        return LS_ERR_INVALID_ARG;
    }
    ix->group = G;
    ix->device = device_ids[0];
    ix->n = n;
    ix->dtype = dtype;
    ix->g = geom;
    G->G = n_devices;
    G->dev.assign(device_ids, device_ids + n_devices);
    G->distinct = true;
    for (int a = 0; a < n_devices; ++a)
        for (int b = a + 1; b < n_devices; ++b)
            if (device_ids[a] == device_ids[b]) G->distinct = false;
    *pix = ix;
    return LS_OK;
}

static int group_finish(ls_index* ix) {
    ls_shard_group* G = ix->group;
    G->sh.resize(G->G);
    G->peer.assign((size_t)G->G * G->G, -1);
    for (int g = 0; g < G->G; ++g) {
        LS_HIP(hipSetDevice(G->dev[g]));
        LS_HIP(hipStreamCreateWithFlags(&G->sh[g].stream, hipStreamNonBlocking));
        LS_HIP(hipEventCreateWithFlags(&G->sh[g].ev_x, hipEventDisableTiming));
        // peer copies (queries out, blocks back) go over xGMI directly when peer access is on
        for (int h = 0; h < G->G; ++h) {
            if (G->dev[h] == G->dev[g]) continue;
            int can = 0;
            const bool asked = hipDeviceCanAccessPeer(&can, G->dev[g], G->dev[h]) == hipSuccess;
            G->peer[(size_t)g * G->G + h] = asked ? (can ? 1 : 0) : -2;
            if (asked && can) {
                hipError_t e = hipDeviceEnablePeerAccess(G->dev[h], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
                if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    LS_HIP(hipSetDevice(G->dev[0]));
    LS_HIP(hipEventCreateWithFlags(&G->ev_in, hipEventDisableTiming));
    LS_HIP(hipEventCreateWithFlags(&G->ev_out, hipEventDisableTiming));
    return LS_OK;
}

extern "C" {

int ls_create_sharded(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
                      const int32_t* device_ids, int32_t n_devices) {
    if (n > 0 && !corpus) {
        ls_set_error("ls_create_sharded: corpus is null");
        return LS_ERR_INVALID_ARG;
    }
    ls_index* ix = nullptr;
    int rc = group_begin(out, n, d, dtype, device_ids, n_devices, "ls_create_sharded", &ix);
    if (rc != LS_OK) return rc;
    ls_shard_group* G = ix->group;
    const int64_t per = n ? (n + n_devices - 1) / n_devices : 0;
    for (int g = 0; g < n_devices && rc == LS_OK; ++g) {
        const int64_t lo = std::min(n, (int64_t)g * per), hi = std::min(n, lo + per);
        ls_index* sub = nullptr;
        rc = ls_create(&sub, hi > lo ? corpus + lo * (int64_t)d : nullptr, hi - lo, d, dtype,
                       device_ids[g]);
        if (rc != LS_OK) break;
        sub->base = lo;
        G->sub.push_back(sub);
        G->lo.push_back(lo);
    }
    if (rc == LS_OK) rc = group_finish(ix);
    if (rc != LS_OK) {
        ls_destroy(ix);
        return rc;
    }
    *out = ix;
    return LS_OK;
}

// A corpus that fits one GPU (the reference's real case: ~200 k x 1024 fp32 = 0.8 GB) gains nothing
// from row shards - every query would still touch every device, plus an exchange. REPLICAS scale
// the reference's workload instead: one full copy per device, and the synchronous host calls
// (ls_search: what the reference issues, search/engine.py:250) are dealt round-robin to the
// replicas, each with its own queue of concurrent callers, stream and pinned buffers, so G callers
// run on G devices at once. Everything else behaves like one index: ls_add appends to every replica,
// ls_reconstruct reads replica 0, device-resident queries (ls_search_device: they live on
// device_ids[0]) are served by the replica there. Results are those of a single-device index.
int ls_create_replicated(ls_index** out, const float* corpus, int64_t n, int32_t d, int32_t dtype,
                         const int32_t* device_ids, int32_t n_devices) {
    if (n > 0 && !corpus) {
        ls_set_error("ls_create_replicated: corpus is null");
        return LS_ERR_INVALID_ARG;
    }
    ls_device_guard guard;
    ls_index* ix = nullptr;
    int rc = group_begin(out, n, d, dtype, device_ids, n_devices, "ls_create_replicated", &ix);
    if (rc != LS_OK) return rc;
    ls_shard_group* G = ix->group;
    G->replicated = true;
    for (int g = 0; g < n_devices && rc == LS_OK; ++g) {
        ls_index* sub = nullptr;
        rc = ls_create(&sub, corpus, n, d, dtype, device_ids[g]);
        if (rc != LS_OK) break;
        G->sub.push_back(sub);
        G->lo.push_back(0);
    }
    if (rc == LS_OK) rc = group_finish(ix);
    if (rc != LS_OK) {
        ls_destroy(ix);
        return rc;
    }
    *out = ix;
    return LS_OK;
}

int ls_create_sharded_from_device(ls_index** out, const void* const* d_blocks, const int64_t* rows,
                                  int32_t d, int32_t dtype, const int32_t* device_ids,
                                  int32_t n_devices) {
    if (n_devices > 0 && (!d_blocks || !rows)) {
        ls_set_error("ls_create_sharded_from_device: null block / row-count array");
        return LS_ERR_INVALID_ARG;
    }
    int64_t n = 0;
    for (int g = 0; g < n_devices; ++g) {
        if (rows[g] < 0 || (rows[g] > 0 && !d_blocks[g])) {
            ls_set_error("ls_create_sharded_from_device: bad block %d", g);
            return LS_ERR_INVALID_ARG;
        }
        n += rows[g];
    }
    ls_index* ix = nullptr;
    int rc = group_begin(out, n, d, dtype, device_ids, n_devices, "ls_create_sharded_from_device", &ix);
    if (rc != LS_OK) return rc;
    ls_shard_group* G = ix->group;
    int64_t lo = 0;
    for (int g = 0; g < n_devices && rc == LS_OK; ++g) {
        ls_index* sub = nullptr;
        rc = ls_create_from_device(&sub, d_blocks[g], rows[g], d, dtype, device_ids[g]);
        if (rc != LS_OK) break;
        sub->base = lo;
        G->sub.push_back(sub);
        G->lo.push_back(lo);
        lo += rows[g];
    }
    if (rc == LS_OK) rc = group_finish(ix);
    if (rc != LS_OK) {
        ls_destroy(ix);
        return rc;
    }
    *out = ix;
    return LS_OK;
}

// What the exchange of a sharded handle really does on this node, as one JSON object (for bench
// records and bug reports): exchange mode, RCCL version / failure text, enqueue workers, and the
// hipDeviceCanAccessPeer matrix seen at creation (-1 on the diagonal and between shards that share a
// device). Returns the number of bytes the full text needs (excluding the terminator).
int32_t ls_shard_exchange_info(ls_index* ix, char* buf, int32_t cap) {
    if (!ix || !ix->group) {
        ls_set_error("ls_shard_exchange_info: not a sharded handle");
        return LS_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(ix->mu);
    const ls_shard_group* G = ix->group;
    std::string o = "{\"exchange\": \"";
    o += G->replicated ? "none (replicas: every device holds the whole corpus)"
         : G->rccl_failed ? "peer-copy (RCCL failed)"
         : group_uses_rccl(G) ? (G->exchange_mode == 2
                                     ? (G->comms_ready ? "rccl gather-to-root (ncclSend/ncclRecv)" : "rccl gather-to-root (not used yet)")
                                     : (G->comms_ready ? "rccl all-gather" : "rccl all-gather (not used yet)"))
         : (G->distinct ? "peer-copy (selected)" : "device-to-device copies (shards share a device)");
    o += "\", \"rccl_version\": " + std::to_string(G->rccl_version);
    std::string err = G->rccl_error;
    for (char& c : err)
        if (c == '"' || c == '\\' || (unsigned char)c < 32) c = ' ';
    o += ", \"rccl_error\": \"" + err + "\"";
    o += ", \"rccl_communicators\": " + std::to_string(G->comms_ready ? G->G : 0);
    const bool workers = G->G > 1 && (G->opt_workers == 1 || (G->opt_workers < 0 && G->distinct));
    o += std::string(", \"enqueue_workers\": ") + (workers ? "true" : "false");
    o += ", \"worker_calls\": " + std::to_string(G->n_worker_calls);
    o += ", \"devices\": [";
    for (int g = 0; g < G->G; ++g) o += (g ? ", " : "") + std::to_string(G->dev[g]);
    o += "], \"peer_access\": [";
    for (int a = 0; a < G->G; ++a) {
        o += a ? ", [" : "[";
        for (int b = 0; b < G->G; ++b)
            o += (b ? ", " : "") + std::to_string(G->peer.empty() ? -1 : G->peer[(size_t)a * G->G + b]);
        o += "]";
    }
    o += "]}";
    if (buf && cap > 0) {
        const size_t m = std::min<size_t>(o.size(), (size_t)cap - 1);
        memcpy(buf, o.data(), m);
        buf[m] = 0;
    }
    return (int32_t)o.size();
}

int32_t ls_shard_count(const ls_index* ix) { return ix ? (ix->group ? ix->group->G : 0) : -1; }

int ls_shard_info(const ls_index* ix, int32_t shard, int32_t* device, int64_t* row0, int64_t* rows) {
    if (!ix || !ix->group || shard < 0 || shard >= ix->group->G) {
        ls_set_error("ls_shard_info: not a sharded handle, or shard out of range");
        return LS_ERR_INVALID_ARG;
    }
    const ls_shard_group* G = ix->group;
    if (device) *device = G->dev[shard];
    if (row0) *row0 = ix->base + G->lo[shard];
    if (rows) *rows = G->sub[shard]->n;
    return LS_OK;
}

}  // extern "C"
