// coresidency_probe.hip — what a stream of back-to-back "MFMA pass"-shaped kernels pays for the things a
// three-stream pipeline would add (round 4, config 3):
//   E1  per-kernel cost on the pass stream of: a completion event attached to the dispatch
//       (hipExtLaunchKernelGGL stopEvent), hipEventRecord (default / hipEventReleaseToDevice), and a
//       hipStreamWaitEvent on an event that completed long ago;
//   E2  does a 64-thread, <= 48-VGPR kernel on a second stream run INSIDE a resident pass-shaped kernel
//       (512 threads, 229 VGPRs -> 2 waves per SIMD use 464 of the 512 registers, 96 KB LDS)?
//   E3  does hipExtAnyOrderLaunch lift the in-stream barrier on gfx950?
//   E5  the chain's main-stream pattern: [waits on events of two other streams that completed long ago but
//       that the HOST has never observed] big kernel (stop event, waited for by a third stream); small
//       kernel - what do the wait packets cost when the runtime cannot elide them?
//   E4  is a hipStreamWaitEvent on a dispatch-attached stop event a real dependency (the waiter must see
//       what the kernel wrote at its very end)?
// hipcc --offload-arch=gfx950 -O2 tools/coresidency_probe.hip -o /tmp/coresidency_probe && /tmp/coresidency_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// pass-shaped: 512 threads, >= 229 VGPRs, dynamic LDS; spins for `ticks` of the 100 MHz clock
__global__ __launch_bounds__(512, 2) void big_kernel(unsigned long long ticks, unsigned long long* stamps) {
    extern __shared__ unsigned char smem[];
    asm volatile("v_mov_b32 v228, 0" ::: "v228");
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && stamps) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}
// select-shaped: one wave, <= 48 VGPRs, `lds` bytes of LDS; spins for `ticks`
template <int VG>
__global__ __launch_bounds__(64) void small_kernel(unsigned long long ticks, unsigned long long* stamps) {
    extern __shared__ unsigned char smem[];
    if (VG == 40) asm volatile("v_mov_b32 v40, 0" ::: "v40");
    if (VG == 80) asm volatile("v_mov_b32 v80, 0" ::: "v80");
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t0;
        stamps[2 * blockIdx.x + 1] = wall_clock64();
    }
}

__global__ void late_writer(unsigned long long ticks, unsigned* flag, unsigned val) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0 && blockIdx.x == 0) *flag = val;
}
__global__ void early_reader(const unsigned* flag, unsigned* seen) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *seen = *flag;
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    CK(hipSetDevice(0));
    hipStream_t M, S;
    CK(hipStreamCreateWithFlags(&M, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int NB = 256, REP = 60;
    const unsigned long long T = 10000;  // 100 us
    unsigned long long *d_big, *d_small;
    CK(hipMalloc(&d_big, sizeof(unsigned long long) * 2 * NB * 4));
    CK(hipMalloc(&d_small, sizeof(unsigned long long) * 2 * 4096));
    std::vector<hipEvent_t> ev(REP), evd(REP);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : evd) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventReleaseToDevice));
    hipEvent_t old_ev;
    CK(hipEventCreateWithFlags(&old_ev, hipEventDisableTiming));
    CK(hipEventRecord(old_ev, S));
    CK(hipStreamSynchronize(S));
    hipEvent_t t0, t1;
    CK(hipEventCreate(&t0));
    CK(hipEventCreate(&t1));

    // ---- E1 --------------------------------------------------------------------------------------
    const char* names[] = {"plain launches", "stopEvent attached (hipExtLaunchKernelGGL)",
                           "hipEventRecord after each (default flags)",
                           "hipEventRecord after each (ReleaseToDevice)",
                           "hipStreamWaitEvent(old event) before each",
                           "stopEvent + a second stream waiting on it and running a small kernel"};
    for (int variant = 0; variant < 6; ++variant) {
        for (int round = 0; round < 2; ++round) {  // round 0 warms up
            CK(hipEventRecord(t0, M));
            const double h0 = now_us();
            for (int i = 0; i < REP; ++i) {
                if (variant == 4) CK(hipStreamWaitEvent(M, old_ev, 0));
                if (variant == 1 || variant == 5)
                    hipExtLaunchKernelGGL(big_kernel, dim3(NB), dim3(512), 96 * 1024, M, nullptr, ev[i], 0, T, (unsigned long long*)nullptr);
                else
                    hipLaunchKernelGGL(big_kernel, dim3(NB), dim3(512), 96 * 1024, M, T, (unsigned long long*)nullptr);
                if (variant == 2) CK(hipEventRecord(ev[i], M));
                if (variant == 3) CK(hipEventRecord(evd[i], M));
                if (variant == 5) {
                    CK(hipStreamWaitEvent(S, ev[i], 0));
                    hipLaunchKernelGGL(small_kernel<40>, dim3(1024), dim3(64), 8 * 1024, S, 1000ull, d_small);
                }
            }
            const double h1 = now_us();
            CK(hipEventRecord(t1, M));
            CK(hipStreamSynchronize(M));
            CK(hipStreamSynchronize(S));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, t0, t1));
            if (round == 1)
                printf("E1 %-70s %.2f us per kernel (kernel spins 100.0), host enqueue %.1f us each\n", names[variant],
                       ms * 1e3 / REP, (h1 - h0) / REP);
        }
    }
    CK(hipGetLastError());

    // ---- E2: co-residency -------------------------------------------------------------------------
    for (int vg : {40, 80}) {
        for (int lds_kb : {8, 20}) {
            std::vector<unsigned long long> hb(2 * NB), hs(2 * 1024);
            CK(hipMemset(d_big, 0, sizeof(unsigned long long) * 2 * NB));
            hipLaunchKernelGGL(big_kernel, dim3(NB), dim3(512), 96 * 1024, M, 20000ull, d_big);  // 200 us
            if (vg == 40) hipLaunchKernelGGL(small_kernel<40>, dim3(1024), dim3(64), lds_kb * 1024, S, 1000ull, d_small);
            else hipLaunchKernelGGL(small_kernel<80>, dim3(1024), dim3(64), lds_kb * 1024, S, 1000ull, d_small);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(hb.data(), d_big, sizeof(unsigned long long) * 2 * NB, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hs.data(), d_small, sizeof(unsigned long long) * 2 * 1024, hipMemcpyDeviceToHost));
            unsigned long long b0 = ~0ull, b1 = 0, s0 = ~0ull, s1 = 0;
            for (int i = 0; i < NB; ++i) { b0 = std::min(b0, hb[2 * i]); b1 = std::max(b1, hb[2 * i + 1]); }
            int inside = 0;
            for (int i = 0; i < 1024; ++i) {
                s0 = std::min(s0, hs[2 * i]); s1 = std::max(s1, hs[2 * i + 1]);
                if (hs[2 * i + 1] < b1 && hs[2 * i] > b0) ++inside;
            }
            printf("E2 small kernel %d VGPRs, %d KB LDS, 1024 x 64 threads x 10 us: %d of 1024 waves ran entirely inside the "
                   "200 us pass-shaped kernel; small kernel spans %.1f .. %.1f us after the big one's first stamp (big ends at %.1f)\n",
                   vg, lds_kb, inside, ((double)s0 - (double)b0) / 100.0, ((double)s1 - (double)b0) / 100.0, (b1 - b0) / 100.0);
        }
    }

    // ---- E3: any-order launch -----------------------------------------------------------------------
    {
        std::vector<unsigned long long> hb(2 * NB), hs(2 * 1024);
        hipLaunchKernelGGL(big_kernel, dim3(NB), dim3(512), 96 * 1024, M, 20000ull, d_big);
        hipExtLaunchKernelGGL(small_kernel<40>, dim3(1024), dim3(64), 8 * 1024, M, nullptr, nullptr, hipExtAnyOrderLaunch, 1000ull, d_small);
        hipError_t e = hipGetLastError();
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hb.data(), d_big, sizeof(unsigned long long) * 2 * NB, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hs.data(), d_small, sizeof(unsigned long long) * 2 * 1024, hipMemcpyDeviceToHost));
        unsigned long long b1 = 0, s0 = ~0ull;
        for (int i = 0; i < NB; ++i) b1 = std::max(b1, hb[2 * i + 1]);
        for (int i = 0; i < 1024; ++i) s0 = std::min(s0, hs[2 * i]);
        printf("E3 hipExtAnyOrderLaunch in the SAME stream (%s): small kernel's first wave starts %.1f us %s the big kernel's end\n",
               hipGetErrorString(e), s0 < b1 ? (b1 - s0) / 100.0 : (s0 - b1) / 100.0, s0 < b1 ? "BEFORE" : "after");
    }
    // ---- E5: barrier packets the runtime cannot elide ---------------------------------------------------
    {
        hipStream_t Pm, Mm, Sm;
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CK(hipStreamCreateWithPriority(&Mm, hipStreamNonBlocking, greatest));
        CK(hipStreamCreateWithPriority(&Pm, hipStreamNonBlocking, (least + greatest) / 2));
        CK(hipStreamCreateWithPriority(&Sm, hipStreamNonBlocking, least));
        const int NE = 64;
        std::vector<hipEvent_t> ep(NE), es(NE), ek(NE);
        for (int i = 0; i < NE; ++i) {
            CK(hipEventCreateWithFlags(&ep[i], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&es[i], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ek[i], hipEventDisableTiming));
        }
        const char* vn[] = {"no waits on the main stream", "waits on 2 old events (unobserved by the host) before each big kernel",
                            "same, plus the select-like kernel on a third stream behind the big kernel's stop event"};
        for (int variant = 0; variant < 3; ++variant) {
            for (int round = 0; round < 2; ++round) {
                // "old" events: tiny kernels on P and S, recorded, never synchronised by the host
                for (int i = 0; i < NE; ++i) {
                    hipLaunchKernelGGL(late_writer, dim3(1), dim3(64), 0, Pm, 1ull, (unsigned*)d_small, 0u);
                    CK(hipEventRecord(ep[i], Pm));
                    hipLaunchKernelGGL(late_writer, dim3(1), dim3(64), 0, Sm, 1ull, (unsigned*)d_small + 8, 0u);
                    CK(hipEventRecord(es[i], Sm));
                }
                hipLaunchKernelGGL(big_kernel, dim3(NB), dim3(512), 96 * 1024, Mm, 30000ull, (unsigned long long*)nullptr);  // lets P / S finish first
                CK(hipEventRecord(t0, Mm));
                for (int i = 0; i < NE; ++i) {
                    if (variant >= 1) {
                        CK(hipStreamWaitEvent(Mm, ep[i], 0));
                        CK(hipStreamWaitEvent(Mm, es[i], 0));
                    }
                    hipExtLaunchKernelGGL(big_kernel, dim3(NB), dim3(512), 96 * 1024, Mm, nullptr, ek[i], 0, T, (unsigned long long*)nullptr);
                    hipLaunchKernelGGL(small_kernel<40>, dim3(1024), dim3(64), 0, Mm, 300ull, d_small + 64);
                    if (variant == 2) {
                        CK(hipStreamWaitEvent(Sm, ek[i], 0));
                        hipLaunchKernelGGL(small_kernel<40>, dim3(1024), dim3(64), 8 * 1024, Sm, 1500ull, d_small + 4096 - 2048);
                    }
                }
                CK(hipEventRecord(t1, Mm));
                CK(hipDeviceSynchronize());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, t0, t1));
                if (round == 1)
                    printf("E5 %-100s %.2f us per (100 us kernel + 3 us kernel)\n", vn[variant], ms * 1e3 / NE);
            }
        }
    }
    // ---- E4: the stop event as a dependency ------------------------------------------------------------
    {
        unsigned *d_flag, *d_seen;
        CK(hipMalloc(&d_flag, 8));
        CK(hipMalloc(&d_seen, 4 * 64));
        CK(hipMemset(d_flag, 0, 8));
        int bad = 0;
        for (int i = 1; i <= 64; ++i) {
            hipExtLaunchKernelGGL(late_writer, dim3(256), dim3(64), 0, M, nullptr, ev[i % REP], 0, 3000ull, d_flag, (unsigned)i);
            CK(hipStreamWaitEvent(S, ev[i % REP], 0));
            hipLaunchKernelGGL(early_reader, dim3(1), dim3(64), 0, S, (const unsigned*)d_flag, d_seen + (i - 1));
        }
        CK(hipDeviceSynchronize());
        unsigned seen[64];
        CK(hipMemcpy(seen, d_seen, sizeof(seen), hipMemcpyDeviceToHost));
        for (int i = 1; i <= 64; ++i) bad += seen[i - 1] != (unsigned)i;
        printf("E4 reader on stream S behind hipStreamWaitEvent(stopEvent of the 30 us writer on M): %d of 64 readers saw a stale flag\n", bad);
    }
    return 0;
}
