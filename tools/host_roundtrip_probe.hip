// host_roundtrip_probe.hip - the floor under a synchronous single-query host call: launch -> kernel -> a completion
// word in pinned host memory -> the polling core. Stand-alone (no library):
//   hipcc -O3 --offload-arch=gfx950 tools/host_roundtrip_probe.hip -o scratch/host_roundtrip_probe && ./scratch/host_roundtrip_probe
// E1 one workgroup writes the word | E2 448 workgroups of 256 threads read a 1.5 KB query first, from pinned host
// memory (what ls_search does: no copy command), from device memory, or from the kernel arguments | E3 the same
// with a 45 us spin in every workgroup (a scan-sized kernel: is the launch latency hidden or added?).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#include <immintrin.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct qarg { float v[384]; };
__device__ __forceinline__ void spin_us(unsigned us) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 100ull * us) __builtin_amdgcn_s_sleep(2);
}
__global__ void k_word(unsigned* done, unsigned seq, unsigned us) {
    if (us) spin_us(us);
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_query(const float* q, float* sink, unsigned* done, unsigned seq, unsigned us) {
    float s = 0.f;
    for (int i = threadIdx.x & 63; i < 384; i += 64) s += q[i];
    if (s == 12345.678f) sink[blockIdx.x] = s;
    if (us) spin_us(us);
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_karg(qarg q, float* sink, unsigned* done, unsigned seq, unsigned us) {
    float s = 0.f;
    for (int i = threadIdx.x & 63; i < 384; i += 64) s += q.v[i];
    if (s == 12345.678f) sink[blockIdx.x] = s;
    if (us) spin_us(us);
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <class F> static double p50(F&& f, volatile unsigned* done, unsigned& seq) {
    std::vector<double> t;
    for (int i = 0; i < 400; ++i) {
        ++seq;
        const auto t0 = std::chrono::steady_clock::now();
        f(seq);
        while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != seq) _mm_pause();
        const auto t1 = std::chrono::steady_clock::now();
        if (i >= 100) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned* done; CK(hipHostMalloc((void**)&done, 64, hipHostMallocDefault)); *done = 0;
    float* hq; CK(hipHostMalloc((void**)&hq, 1536, hipHostMallocDefault));
    float *dq, *sink; CK(hipMalloc((void**)&dq, 1536)); CK(hipMalloc((void**)&sink, 4096));
    qarg qa; for (int i = 0; i < 384; ++i) qa.v[i] = hq[i] = 0.001f * i;
    CK(hipMemcpy(dq, hq, 1536, hipMemcpyHostToDevice));
    unsigned seq = 0;
    for (unsigned us : {0u, 45u}) {
        printf("kernel body spins %u us:\n", us);
        printf("  E1 one workgroup, completion word only              : %.1f us\n", p50([&](unsigned q) { hipLaunchKernelGGL(k_word, dim3(1), dim3(256), 0, s, done, q, us); }, done, seq));
        printf("  E1 448 workgroups, completion word only             : %.1f us\n", p50([&](unsigned q) { hipLaunchKernelGGL(k_word, dim3(448), dim3(256), 0, s, done, q, us); }, done, seq));
        printf("  E2 448 workgroups read the query from pinned host   : %.1f us\n", p50([&](unsigned q) { memcpy(hq, qa.v, 1536); hipLaunchKernelGGL(k_query, dim3(448), dim3(256), 0, s, (const float*)hq, sink, done, q, us); }, done, seq));
        printf("  E2 ... from device memory (no copy: lower bound)    : %.1f us\n", p50([&](unsigned q) { hipLaunchKernelGGL(k_query, dim3(448), dim3(256), 0, s, (const float*)dq, sink, done, q, us); }, done, seq));
        printf("  E2 ... hipMemcpyAsync H2D + device memory           : %.1f us\n", p50([&](unsigned q) { memcpy(hq, qa.v, 1536); (void)hipMemcpyAsync(dq, hq, 1536, hipMemcpyHostToDevice, s); hipLaunchKernelGGL(k_query, dim3(448), dim3(256), 0, s, (const float*)dq, sink, done, q, us); }, done, seq));
        printf("  E2 ... from the kernel arguments (1.5 KB by value)  : %.1f us\n", p50([&](unsigned q) { hipLaunchKernelGGL(k_karg, dim3(448), dim3(256), 0, s, qa, sink, done, q, us); }, done, seq));
    }
    return 0;
}
