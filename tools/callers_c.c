/* callers_c.c - T threads making synchronous single-query ls_search calls on ONE handle (the reference's call,
 * search/engine.py:250, issued by several MCP clients, mcp/server.py:147-151), timed from C: what the library's
 * caller combining delivers without the Python threads' GIL hand-offs that tools/concurrent_callers.py includes.
 *   gcc -O2 tools/callers_c.c -o /tmp/callers_c -ldl -lm -lpthread && /tmp/callers_c lean-explore_amd/libleansearch.so [overlap [gather [reps]]]
 * Round 6, OPEN LOOP (verdict item 4):  /tmp/callers_c lib open [seconds per point]
 *   16 client threads, each a Poisson process of rate lambda / 16 (arrival times drawn ahead, independent of the
 *   answers: a thread that is still in a call when its next arrival is due starts late, and the latency of that
 *   request counts from its ARRIVAL time), lambda = 5 / 10 / 20 / 40 / 80 k requests/s, with the caller gather
 *   (debug option 20) on and off: achieved q/s, p50, p99 - next to the lone caller's closed-loop p50 / p99. The clients sleep
 *   while their next arrival is > 150 us away (16 busy-waiting clients would be the whole CPU quota of the GPU box's container).
 * Closed-loop switches (environment): CALLERS_ONLY=<T> one caller count (64 and 128 only this way) | CALLERS_SHAPES=1 the d = 384 shape only |
 *   CALLERS_NQ=<n> queries per call | CALLERS_COUNTERS=1 print the handle's cumulative counters (batches, requests, launches, the
 *   leaders' phase clocks 28-33; with a -DLS_LEAD_TRACE variant also counters 40-47) | CALLERS_VERIFY=1 every call's rows memcmp'd
 *   with the lone call's | CALLERS_PIN=<cpu> caller t on cpu + t | CALLERS_BURNERS=<n> n more threads that only spin |
 *   CALLERS_GAP_US=<us> think time between two calls | CALLERS_OPT22 / CALLERS_OPT23 = debug options 22 / 23. */
#define _GNU_SOURCE
#include <sched.h>
#include <sys/prctl.h>
#include <string.h>
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef struct ls_index ls_index;
static int64_t (*g_counter)(ls_index*, int32_t);
static int (*search)(ls_index*, const float*, int64_t, int32_t, uint32_t, float*, int64_t*);
static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static float gauss(uint64_t* s) {
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    const double u = ((*s >> 11) + 1.0) / 9007199254740993.0;
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    const double v = ((*s >> 11) + 1.0) / 9007199254740993.0;
    return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
}
static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
struct job { ls_index* ix; const float* q; int d, k; double stop; long calls; double* lat; long cap; int nq; const float* wantD; const int64_t* wantI; long bad; };
static void* worker(void* p) {
    struct job* j = p;
    float* D = malloc(sizeof(float) * j->k * j->nq);
    int64_t* I = malloc(sizeof(int64_t) * j->k * j->nq);
    while (now_us() < j->stop) {
        const double t0 = now_us();
        if (search(j->ix, j->q, j->nq, j->k, 1u, D, I)) break;
        if (j->calls < j->cap) j->lat[j->calls] = now_us() - t0;
        j->calls++;
        /* CALLERS_VERIFY=1: every call's rows against the lone call's, bit for bit */
        if (j->wantD && (memcmp(D, j->wantD, sizeof(float) * j->k * j->nq) || memcmp(I, j->wantI, sizeof(int64_t) * j->k * j->nq))) j->bad++;
        if (getenv("CALLERS_GAP_US")) {  /* think time between two calls (busy wait): what an idle GPU costs the next call */
            const double until = now_us() + atof(getenv("CALLERS_GAP_US"));
            while (now_us() < until) { }
        }
    }
    free(D); free(I);
    return NULL;
}
static void* burner(void* p) {  /* CALLERS_BURNERS=<n>: n more threads that only spin (what busy cores alone cost the callers) */
    const double stop = *(double*)p;
    while (now_us() < stop) { }
    return NULL;
}
/* ---- open loop ---------------------------------------------------------------------------------------------- */
struct ojob { ls_index* ix; const float* q; int d, k; double t0, stop, rate_per_us; uint64_t seed; long calls; double* lat; long cap; };
static void* open_worker(void* p) {
    struct ojob* j = p;
    prctl(PR_SET_TIMERSLACK, 1000UL);  /* (1 us instead of the default 50 us of slack on this thread's sleeps) */
    float* D = malloc(sizeof(float) * j->k);
    int64_t* I = malloc(sizeof(int64_t) * j->k);
    uint64_t s = j->seed;
    double arrival = j->t0;
    for (;;) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = ((s >> 11) + 1.0) / 9007199254740993.0;
        arrival += -log(u) / j->rate_per_us;          /* exponential inter-arrival time */
        if (arrival >= j->stop) break;
        /* idle until the request arrives: asleep while it is more than 150 us away (16 busy-waiting clients are the whole
         * 16-CPU quota of the GPU box's container - every row of the first record had a 7-8 ms maximum, the throttle),
         * polling the clock for the rest (a wake-up is tens of us late) */
        for (double now = now_us(); now < arrival; now = now_us()) {
            if (arrival - now > 150.0) {
                const double us = arrival - now - 100.0;
                struct timespec ts = {(time_t)(us * 1e-6), (long)(fmod(us, 1e6) * 1e3)};
                nanosleep(&ts, NULL);
            }
        }
        if (search(j->ix, j->q, 1, j->k, 1u, D, I)) break;
        if (j->calls < j->cap) j->lat[j->calls] = now_us() - arrival;   /* from ARRIVAL, not from the call's start */
        j->calls++;
    }
    free(D); free(I);
    return NULL;
}
static int open_loop(ls_index* ix, int (*option)(ls_index*, int32_t, int32_t), const float* q, int64_t n, int d, int k, double secs) {
    enum { T = 16 };
    float* D = malloc(sizeof(float) * k); int64_t* I = malloc(sizeof(int64_t) * k);
    for (int i = 0; i < 200; ++i) search(ix, q, 1, k, 1u, D, I);
    double lone[2000];
    for (int i = 0; i < 2000; ++i) { const double t0 = now_us(); search(ix, q, 1, k, 1u, D, I); lone[i] = now_us() - t0; }
    qsort(lone, 2000, sizeof(double), cmp);
    printf("open loop N=%lld d=%d k=%d: lone caller (closed loop) p50 %.1f us, p99 %.1f us\n", (long long)n, d, k, lone[1000], lone[1980]);
    free(D); free(I);
    const double lambdas[5] = {5e3, 10e3, 20e3, 40e3, 80e3};
    for (int li = 0; li < 5; ++li) {
        for (int gather = 1; gather >= 0; --gather) {
            option(ix, 20, gather ? 2 : 0);
            pthread_t th[T];
            struct ojob jobs[T];
            const double t0 = now_us() + 2000.0, stop = t0 + secs * 1e6;
            for (int t = 0; t < T; ++t) {
                jobs[t] = (struct ojob){ix, q + (size_t)t * d, d, k, t0, stop, lambdas[li] / T * 1e-6, 0x9E3779B97F4A7C15ull * (t + 1) + li, 0,
                                        malloc(sizeof(double) * 400000), 400000};
                pthread_create(&th[t], NULL, open_worker, &jobs[t]);
            }
            long total = 0;
            for (int t = 0; t < T; ++t) { pthread_join(th[t], NULL); total += jobs[t].calls; }
            const double dt = now_us() - t0;
            double* all = malloc(sizeof(double) * (size_t)(total + 1));
            long m = 0;
            for (int t = 0; t < T; ++t) { for (long i = 0; i < jobs[t].calls && i < jobs[t].cap; ++i) all[m++] = jobs[t].lat[i]; free(jobs[t].lat); }
            qsort(all, m, sizeof(double), cmp);
            printf("open loop N=%lld d=%d k=%d lambda %6.0f/s gather %s: achieved %7.0f q/s, p50 %6.1f us, p99 %6.1f us, p99.9 %7.1f us, max %8.1f us (from arrival)",
                   (long long)n, d, k, lambdas[li], gather ? "on " : "off", total / (dt * 1e-6), m ? all[m / 2] : 0.0, m ? all[(long)(m * 0.99)] : 0.0,
                   m ? all[(long)(m * 0.999)] : 0.0, m ? all[m - 1] : 0.0);
            if (g_counter) printf("  [2 ms poll timeouts so far %lld, retries %lld, waiters put to sleep %lld]", (long long)g_counter(ix, 27), (long long)g_counter(ix, 20), (long long)g_counter(ix, 33));
            printf("\n");
            fflush(stdout);
            free(all);
        }
    }
    option(ix, 20, 2);
    return 0;
}

int main(int argc, char** argv) {
    void* lib = dlopen(argc > 1 ? argv[1] : "lean-explore_amd/libleansearch.so", RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    int (*create)(ls_index**, const float*, int64_t, int32_t, int32_t, int32_t) = dlsym(lib, "ls_create");
    search = dlsym(lib, "ls_search");
    void (*destroy)(ls_index*) = dlsym(lib, "ls_destroy");
    int (*option)(ls_index*, int32_t, int32_t) = dlsym(lib, "ls_debug_option");
    const char* (*lasterr)(void) = dlsym(lib, "ls_last_error");
    int64_t (*counter)(ls_index*, int32_t) = dlsym(lib, "ls_debug_counter");
    g_counter = counter;
    const int open_mode = argc > 2 && !strcmp(argv[2], "open");
    const double open_secs = argc > 3 && open_mode ? atof(argv[3]) : 2.0;
    const int overlap = argc > 2 && !open_mode ? atoi(argv[2]) : 1;  /* debug option 17: synchronous calls overlap two deep */
    const int gather = argc > 3 && !open_mode ? atoi(argv[3]) : -1;  /* debug option 20: 0 off, 1 long passes only, 2 always (default) */
    const int shapes[2][3] = {{200000, 384, 50}, {200000, 1024, 1000}};
    for (int c = 0; c < (getenv("CALLERS_SHAPES") ? atoi(getenv("CALLERS_SHAPES")) : 2); ++c) {
        const int64_t n = shapes[c][0];
        const int d = shapes[c][1], k = shapes[c][2];
        float* corpus = malloc(sizeof(float) * n * d);
        float* q = malloc(sizeof(float) * 32 * d);
        uint64_t s = 1234;
        for (int64_t i = 0; i < n * d; ++i) corpus[i] = gauss(&s);
        for (int i = 0; i < 32 * d; ++i) q[i] = gauss(&s);
        ls_index* ix = NULL;
        if (create(&ix, corpus, n, d, 0, 0)) { printf("ls_create: %s\n", lasterr()); return 1; }
        if (open_mode) {
            open_loop(ix, option, q, n, d, k, open_secs);
            destroy(ix);
            free(corpus); free(q);
            continue;
        }
        option(ix, 17, overlap);
        if (gather >= 0) option(ix, 20, gather);
        if (getenv("CALLERS_OPT22")) option(ix, 22, atoi(getenv("CALLERS_OPT22")));  /* debug option 22: 32 (1, default) or 16 (0) queries per ls_mq pass */
        if (getenv("CALLERS_OPT23")) option(ix, 23, atoi(getenv("CALLERS_OPT23")));  /* debug option 23: a queue that fills a pass goes early (default 1) */
        if (argc > 5) option(ix, 21, atoi(argv[5]));  /* debug option 21: callers up to which a second batch goes early */
        {
            float* D = malloc(sizeof(float) * k); int64_t* I = malloc(sizeof(int64_t) * k);
            for (int i = 0; i < 50; ++i) search(ix, q, 1, k, 1u, D, I);
            free(D); free(I);
        }
        const int nq_each = getenv("CALLERS_NQ") ? atoi(getenv("CALLERS_NQ")) : 1;  /* queries per call (<= 32; 1 = the reference's call) */
        float* wantD = NULL; int64_t* wantI = NULL;
        if (getenv("CALLERS_VERIFY")) {  /* the lone calls' answers for the 32 queries */
            wantD = malloc(sizeof(float) * 64 * k); wantI = malloc(sizeof(int64_t) * 64 * k);
            for (int i = 0; i < 32; ++i) search(ix, q + (size_t)i * d, 1, k, 1u, wantD + (size_t)i * k, wantI + (size_t)i * k);
            memcpy(wantD + (size_t)32 * k, wantD, sizeof(float) * 32 * k); memcpy(wantI + (size_t)32 * k, wantI, sizeof(int64_t) * 32 * k);
        }
        const int Ts[8] = {1, 2, 4, 8, 16, 32, 64, 128};  /* (32: one two-block ls_mq pass carries them all, round 6) */
        for (int ti = 0; ti < (getenv("CALLERS_ONLY") ? 8 : 6); ++ti) {  /* (64 / 128 callers: only when asked for) */
            const int T = Ts[ti];
            if (getenv("CALLERS_ONLY") && atoi(getenv("CALLERS_ONLY")) != T) continue;  /* (one caller count: for a profile) */
            for (int rep = 0; rep < (argc > 4 ? atoi(argv[4]) : 2); ++rep) {
                pthread_t th[128];
                struct job jobs[128];
                const double t0 = now_us(), stop = t0 + 0.8e6;
                for (int t = 0; t < T; ++t) {
                    jobs[t] = (struct job){ix, q + (size_t)(t * nq_each % 32) * d, d, k, stop, 0, malloc(sizeof(double) * 100000), 100000, nq_each,
                                             wantD ? wantD + (size_t)(t * nq_each % 32) * k : NULL, wantI ? wantI + (size_t)(t * nq_each % 32) * k : NULL, 0};
                    pthread_create(&th[t], NULL, worker, &jobs[t]);
                    if (getenv("CALLERS_PIN")) {  /* CALLERS_PIN=<first cpu>: caller t on cpu first + t (one socket, one thread per core) */
                        cpu_set_t cs; CPU_ZERO(&cs); CPU_SET(atoi(getenv("CALLERS_PIN")) + t, &cs);
                        pthread_setaffinity_np(th[t], sizeof(cs), &cs);
                    }
                }
                pthread_t bt[64];
                double bstop = stop;
                const int nb = getenv("CALLERS_BURNERS") ? atoi(getenv("CALLERS_BURNERS")) : 0;
                for (int b = 0; b < nb && b < 64; ++b) {
                    pthread_create(&bt[b], NULL, burner, &bstop);
                    if (getenv("CALLERS_PIN")) {
                        cpu_set_t cs; CPU_ZERO(&cs); CPU_SET(atoi(getenv("CALLERS_PIN")) + T + b, &cs);
                        pthread_setaffinity_np(bt[b], sizeof(cs), &cs);
                    }
                }
                long total = 0, bad = 0;
                for (int t = 0; t < T; ++t) { pthread_join(th[t], NULL); total += jobs[t].calls; bad += jobs[t].bad; }
                const double dt = now_us() - t0;
                for (int b = 0; b < nb && b < 64; ++b) pthread_join(bt[b], NULL);
                double* all = malloc(sizeof(double) * (size_t)total);
                long m = 0;
                for (int t = 0; t < T; ++t) { for (long i = 0; i < jobs[t].calls && i < jobs[t].cap; ++i) all[m++] = jobs[t].lat[i]; free(jobs[t].lat); }
                qsort(all, m, sizeof(double), cmp);
                printf("C threads%s N=%lld d=%d k=%d, %2d callers: %8.0f q/s, p50 %.1f us", overlap ? "" : " (no overlap)", (long long)n, d, k, T, total * nq_each / (dt * 1e-6), m ? all[m / 2] : 0.0);
                if (nq_each > 1) printf(" (%d queries per call)", nq_each);
                if (wantD) printf(" [calls that differ from the lone call: %ld of %ld]", bad, total);
                if (getenv("CALLERS_COUNTERS"))  /* cumulative: combined batches, their requests, launches, ls_mq launches, retries, second serves */
                    printf("   [batches %lld requests %lld launches %lld mq %lld retries %lld reserved %lld | leaders, cumulative us: wait+gather %lld, begin..finish %lld (begin %lld, finish %lld), relock %lld | waiters put to sleep %lld]",
                           (long long)counter(ix, 16), (long long)counter(ix, 17), (long long)counter(ix, 11), (long long)counter(ix, 23), (long long)counter(ix, 20),
                           (long long)counter(ix, 25), (long long)counter(ix, 28) / 1000, (long long)counter(ix, 29) / 1000, (long long)counter(ix, 31) / 1000, (long long)counter(ix, 32) / 1000, (long long)counter(ix, 30) / 1000, (long long)counter(ix, 33));
                if (getenv("CALLERS_COUNTERS") && counter(ix, 40) >= 0)  /* (a -DLS_LEAD_TRACE variant of the library) */
                    printf("\n      leader trace, cumulative us: stage requests %lld | host_call_begin %lld (of it: to the query copy %lld, copy + H2D %lld, launch %lld) | wait for results %lld | hand results out %lld",
                           (long long)counter(ix, 40) / 1000, (long long)counter(ix, 41) / 1000, (long long)counter(ix, 44) / 1000, (long long)counter(ix, 45) / 1000,
                           (long long)counter(ix, 46) / 1000, (long long)counter(ix, 42) / 1000, (long long)counter(ix, 43) / 1000);
                printf("\n");
                fflush(stdout);
                free(all);
            }
        }
        destroy(ix);
        free(corpus); free(q);
    }
    return 0;
}
