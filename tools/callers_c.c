/* callers_c.c - T threads making synchronous single-query ls_search calls on ONE handle (the reference's call,
 * search/engine.py:250, issued by several MCP clients, mcp/server.py:147-151), timed from C: what the library's
 * caller combining delivers without the Python threads' GIL hand-offs that tools/concurrent_callers.py includes.
 *   gcc -O2 tools/callers_c.c -o scratch/callers_c -ldl -lm -lpthread && ./scratch/callers_c lean-explore_amd/libleansearch.so [overlap [gather [reps]]] */
#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef struct ls_index ls_index;
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
struct job { ls_index* ix; const float* q; int d, k; double stop; long calls; double* lat; long cap; };
static void* worker(void* p) {
    struct job* j = p;
    float* D = malloc(sizeof(float) * j->k);
    int64_t* I = malloc(sizeof(int64_t) * j->k);
    while (now_us() < j->stop) {
        const double t0 = now_us();
        if (search(j->ix, j->q, 1, j->k, 1u, D, I)) break;
        if (j->calls < j->cap) j->lat[j->calls] = now_us() - t0;
        j->calls++;
    }
    free(D); free(I);
    return NULL;
}
int main(int argc, char** argv) {
    void* lib = dlopen(argc > 1 ? argv[1] : "lean-explore_amd/libleansearch.so", RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    int (*create)(ls_index**, const float*, int64_t, int32_t, int32_t, int32_t) = dlsym(lib, "ls_create");
    search = dlsym(lib, "ls_search");
    void (*destroy)(ls_index*) = dlsym(lib, "ls_destroy");
    int (*option)(ls_index*, int32_t, int32_t) = dlsym(lib, "ls_debug_option");
    const char* (*lasterr)(void) = dlsym(lib, "ls_last_error");
    const int overlap = argc > 2 ? atoi(argv[2]) : 1;  /* debug option 17: synchronous calls overlap two deep */
    const int gather = argc > 3 ? atoi(argv[3]) : -1;  /* debug option 20: 0 off, 1 long passes only, 2 always (default) */
    const int shapes[2][3] = {{200000, 384, 50}, {200000, 1024, 1000}};
    for (int c = 0; c < 2; ++c) {
        const int64_t n = shapes[c][0];
        const int d = shapes[c][1], k = shapes[c][2];
        float* corpus = malloc(sizeof(float) * n * d);
        float* q = malloc(sizeof(float) * 16 * d);
        uint64_t s = 1234;
        for (int64_t i = 0; i < n * d; ++i) corpus[i] = gauss(&s);
        for (int i = 0; i < 16 * d; ++i) q[i] = gauss(&s);
        ls_index* ix = NULL;
        if (create(&ix, corpus, n, d, 0, 0)) { printf("ls_create: %s\n", lasterr()); return 1; }
        option(ix, 17, overlap);
        if (gather >= 0) option(ix, 20, gather);
        if (argc > 5) option(ix, 21, atoi(argv[5]));  /* debug option 21: callers up to which a second batch goes early */
        {
            float* D = malloc(sizeof(float) * k); int64_t* I = malloc(sizeof(int64_t) * k);
            for (int i = 0; i < 50; ++i) search(ix, q, 1, k, 1u, D, I);
            free(D); free(I);
        }
        const int Ts[5] = {1, 2, 4, 8, 16};
        for (int ti = 0; ti < 5; ++ti) {
            const int T = Ts[ti];
            for (int rep = 0; rep < (argc > 4 ? atoi(argv[4]) : 2); ++rep) {
                pthread_t th[16];
                struct job jobs[16];
                const double t0 = now_us(), stop = t0 + 0.8e6;
                for (int t = 0; t < T; ++t) {
                    jobs[t] = (struct job){ix, q + (size_t)t * d, d, k, stop, 0, malloc(sizeof(double) * 100000), 100000};
                    pthread_create(&th[t], NULL, worker, &jobs[t]);
                }
                long total = 0;
                for (int t = 0; t < T; ++t) { pthread_join(th[t], NULL); total += jobs[t].calls; }
                const double dt = now_us() - t0;
                double* all = malloc(sizeof(double) * (size_t)total);
                long m = 0;
                for (int t = 0; t < T; ++t) { for (long i = 0; i < jobs[t].calls && i < jobs[t].cap; ++i) all[m++] = jobs[t].lat[i]; free(jobs[t].lat); }
                qsort(all, m, sizeof(double), cmp);
                printf("C threads%s N=%lld d=%d k=%d, %2d callers: %8.0f q/s, p50 %.1f us\n", overlap ? "" : " (no overlap)", (long long)n, d, k, T, total / (dt * 1e-6), m ? all[m / 2] : 0.0);
                fflush(stdout);
                free(all);
            }
        }
        destroy(ix);
        free(corpus); free(q);
    }
    return 0;
}
