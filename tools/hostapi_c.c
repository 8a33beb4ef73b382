/* hostapi_c.c - the reference's call (index.search on host arrays, search/engine.py:250) timed from C, no
 * Python / ctypes in the loop: separates library time from interpreter time for host_api.c2 / c2p.
 *   gcc -O2 tools/hostapi_c.c -o /tmp/hostapi_c -ldl -lm && /tmp/hostapi_c lean-explore_amd/libleansearch.so */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef struct ls_index ls_index;
static double now_us(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e6 + t.tv_nsec * 1e-3;
}
static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }
static float gauss(uint64_t* s) {  /* xorshift + Box-Muller: any continuous rows do */
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    const double u = ((*s >> 11) + 1.0) / 9007199254740993.0;
    *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
    const double v = ((*s >> 11) + 1.0) / 9007199254740993.0;
    return (float)(sqrt(-2.0 * log(u)) * cos(6.283185307179586 * v));
}
int main(int argc, char** argv) {
    void* lib = dlopen(argc > 1 ? argv[1] : "lean-explore_amd/libleansearch.so", RTLD_NOW);
    if (!lib) { printf("dlopen: %s\n", dlerror()); return 1; }
    int (*create)(ls_index**, const float*, int64_t, int32_t, int32_t, int32_t) = dlsym(lib, "ls_create");
    int (*search)(ls_index*, const float*, int64_t, int32_t, uint32_t, float*, int64_t*) = dlsym(lib, "ls_search");
    int (*option)(ls_index*, int32_t, int32_t) = dlsym(lib, "ls_debug_option");
    int64_t (*counter)(ls_index*, int32_t) = dlsym(lib, "ls_debug_counter");
    void (*destroy)(ls_index*) = dlsym(lib, "ls_destroy");
    const char* (*lasterr)(void) = dlsym(lib, "ls_last_error");
    const int shapes[2][3] = {{200000, 384, 50}, {200000, 1024, 1000}};
    for (int c = 0; c < 2; ++c) {
        const int64_t n = shapes[c][0];
        const int d = shapes[c][1], k = shapes[c][2], calls = 300;
        float* corpus = malloc(sizeof(float) * n * d);
        float* q = malloc(sizeof(float) * d);
        float* D = malloc(sizeof(float) * k);
        int64_t* I = malloc(sizeof(int64_t) * k);
        uint64_t s = 1234;
        for (int64_t i = 0; i < n * d; ++i) corpus[i] = gauss(&s);
        for (int i = 0; i < d; ++i) q[i] = gauss(&s);
        ls_index* ix = NULL;
        if (create(&ix, corpus, n, d, 0, 0)) { printf("ls_create: %s\n", lasterr()); return 1; }
        for (int mode = 2; mode >= 0; --mode) {  /* 2: defaults, 1: query by copy command, 0: selection as its own launch */
            option(ix, 15, mode == 1);       /* (libraries older than option 15 answer "unknown option": ignored) */
            option(ix, 9, mode >= 1);
            double lat[300];
            for (int i = 0; i < 30; ++i) search(ix, q, 1, k, 1u, D, I);
            for (int i = 0; i < calls; ++i) {
                const double t0 = now_us();
                if (search(ix, q, 1, k, 1u, D, I)) { printf("ls_search: %s\n", lasterr()); return 1; }
                lat[i] = now_us() - t0;
            }
            qsort(lat, calls, sizeof(double), cmp);
            printf("C harness N=%lld d=%d k=%d nq=1 ls_search(host arrays, normalize): selection %s: p50 %.1f us, p10 %.1f, p90 %.1f "
                   "(same-launch retries so far %lld; top row %lld)\n", (long long)n, d, k,
                   mode == 2 ? "inside the scan launch, query read from pinned host memory (defaults)"
                             : (mode ? "inside the scan launch, query by copy command" : "as its own launch, query read from pinned host memory"), lat[calls / 2], lat[calls / 10], lat[9 * calls / 10],
                   (long long)counter(ix, 20), (long long)I[0]);
        }
        destroy(ix);
        free(corpus); free(q); free(D); free(I);
    }
    return 0;
}
