// tools/tile_counter_probe.hip — what does a GLOBAL tile counter cost a scan-shaped launch? (verdict r4 item 6:
// "one atomicAdd per wave per tile, issued a tile ahead", instead of the static round-robin deal.)
//   hipcc --offload-arch=gfx950 -O2 tools/tile_counter_probe.hip -o scratch/tile_counter_probe && scratch/tile_counter_probe
// 448 workgroups x 4 waves take 25 000 "tiles" (config 2: N = 200 k rows, 8-row tiles); a tile is emulated by
// ~3.3 us of s_sleep (the HBM-bound scan's time per tile and wave). Variants:
//   static : tile = wave id + k * waves (what ls_scan_kernel does)
//   agent  : one counter, agent-scope atomicAdd per wave and tile, fetched one tile ahead
//   xcd    : eight counters indexed by the hardware XCC_ID, a static eighth of the tiles each, workgroup-scope atomics
// Printed: kernel time, tiles processed (must be 25 000), the spread of the waves' finishing times.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__device__ __forceinline__ void fake_tile(int ticks) {  // ticks of the 100 MHz clock
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(256) void k_static(int nt, int ticks, unsigned* done, unsigned long long* endt) {
    const int wave = threadIdx.x >> 6, W = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
    unsigned cnt = 0;
    for (int t = gw; t < nt; t += W) {
        fake_tile(ticks);
        ++cnt;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(done, cnt);
        endt[gw] = wall_clock64();
    }
}

template <bool XCD>
__global__ __launch_bounds__(256) void k_dynamic(int nt, int ticks, unsigned* ctr, unsigned* done, unsigned long long* endt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, gw = blockIdx.x * 4 + wave;
    unsigned xcc = 0;
    if (XCD) xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 7;  // HW_REG_XCC_ID, bits 3:0
    const int lo = XCD ? (int)((long long)nt * xcc / 8) : 0, hi = XCD ? (int)((long long)nt * (xcc + 1) / 8) : nt;
    auto fetch = [&]() -> int {
        int v = 0;
        if (lane == 0) {
            if (XCD) v = (int)__hip_atomic_fetch_add(&ctr[xcc * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else v = (int)__hip_atomic_fetch_add(&ctr[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return lo + __builtin_amdgcn_readfirstlane(v);
    };
    unsigned cnt = 0;
    int cur = fetch(), nxt = fetch();
    while (cur < hi) {
        fake_tile(ticks);
        ++cnt;
        cur = nxt;
        nxt = fetch();
    }
    if (lane == 0) {
        atomicAdd(done, cnt);
        endt[gw] = wall_clock64();
    }
}

int main() {
    const int nt = 25000, blocks = 448, waves = blocks * 4;
    unsigned *ctr, *done;
    unsigned long long* endt;
    hipMalloc(&ctr, 4096);
    hipMalloc(&done, 4);
    hipMalloc(&endt, waves * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<unsigned long long> h(waves);
    for (int ticks : {0, 330}) {
        for (int variant = 0; variant < 3; ++variant) {
            float best = 1e9f;
            unsigned got = 0;
            double spread = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(ctr, 0, 4096);
                hipMemset(done, 0, 4);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                if (variant == 0) hipLaunchKernelGGL(k_static, dim3(blocks), dim3(256), 0, 0, nt, ticks, done, endt);
                if (variant == 1) hipLaunchKernelGGL(k_dynamic<false>, dim3(blocks), dim3(256), 0, 0, nt, ticks, ctr, done, endt);
                if (variant == 2) hipLaunchKernelGGL(k_dynamic<true>, dim3(blocks), dim3(256), 0, 0, nt, ticks, ctr, done, endt);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms = 0;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(&got, done, 4, hipMemcpyDeviceToHost);
                hipMemcpy(h.data(), endt, waves * 8, hipMemcpyDeviceToHost);
                unsigned long long mn = ~0ull, mx = 0;
                for (auto v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
                if (ms < best) { best = ms; spread = (mx - mn) / 100.0; }
            }
            printf("tile = %4.1f us, %-6s: kernel %.1f us, %u tiles, last - first wave end %.1f us\n", ticks / 100.0,
                   variant == 0 ? "static" : variant == 1 ? "agent" : "xcd", best * 1e3, got, spread);
        }
    }
    return 0;
}
