// Stress test of lds_topk (csrc/ls_select_dev.h) against std::sort on the host: random list lengths, k, key
// distributions (distinct random keys, a few distinct score halves, planted clusters, "no result" zeros,
// sorted runs), 256 and 1024 threads. Build + run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ilean-explore_amd/csrc tools/topk_stress.hip -o scratch/topk_stress && scratch/topk_stress 3000
#include "ls_select_dev.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

template <int NT>
__global__ __launch_bounds__(NT) void topk_kernel(const u64* in, const int* cnts, const int* ks, int stride,
                                                 u64* out, int* nvalid, int keys_cap, int res_cap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u64* keys = reinterpret_cast<u64*>(smem);
    u64* res = keys + keys_cap;
    u64* tmp = res + res_cap;
    u64* red = tmp + 256;
    u32* hist = reinterpret_cast<u32*>(red + 16);
    u32* misc = hist + 8 * 256;
    const int c = blockIdx.x, tid = threadIdx.x, cnt = cnts[c], k = ks[c];
    for (int i = tid; i < cnt; i += NT) keys[i] = in[(size_t)c * stride + i];
    for (int i = tid; i < res_cap; i += NT) res[i] = 0xdeadbeefdeadbeefull;
    __syncthreads();
    const int nv = lds_topk(keys, cnt, k, res, tmp, hist, misc, tid, NT);
    __syncthreads();
    for (int i = tid; i < k; i += NT) out[(size_t)c * LS_RES_CAP + i] = i < nv ? res[i] : 0ull;
    if (tid == 0) nvalid[c] = nv;
}

int main(int argc, char** argv) {
    const int ncase = argc > 1 ? atoi(argv[1]) : 2000;
    const int keys_cap = LS_FINAL_CAP, res_cap = LS_RES_CAP;
    std::mt19937_64 rng(argc > 2 ? atoll(argv[2]) : 1);
    std::vector<u64> in((size_t)ncase * keys_cap, 0);
    std::vector<int> cnts(ncase), ks(ncase), kinds(ncase);
    for (int c = 0; c < ncase; ++c) {
        const int kind = c % 6;
        int cnt = 1 + (int)(rng() % (c % 7 == 0 ? 8192 : 4096));
        if (c % 11 == 0) cnt = 257 + (int)(rng() % 64);
        int k = 1 + (int)(rng() % std::min(cnt, LS_RES_CAP));
        if (c % 5 == 0) k = std::min(cnt, std::min(LS_RES_CAP, 50 + (int)(rng() % 1000)));
        cnts[c] = cnt; ks[c] = k; kinds[c] = kind;
        u64* a = &in[(size_t)c * keys_cap];
        std::vector<u32> rows(cnt);
        for (int i = 0; i < cnt; ++i) rows[i] = (u32)i * 7u + 3u;  // distinct
        std::shuffle(rows.begin(), rows.end(), rng);
        for (int i = 0; i < cnt; ++i) {
            u32 hi;
            switch (kind) {
                case 0: hi = (u32)(rng() >> 32) | 1u; break;                         // random
                case 1: hi = 0x40000000u + (u32)(rng() % 5) * 0x1000u; break;        // 5 distinct scores
                case 2: hi = 0x3f000000u; break;                                     // all equal
                case 3: hi = (i % 10 == 0) ? 0x3f800000u + (u32)(rng() % 1000) : 0x3d000000u + (u32)(rng() % 100000); break;  // cluster on top
                case 4: hi = 0x3f000000u + (u32)(rng() % 300); break;                // few hundred scores
                default: hi = 0x3f000000u + (u32)((cnt - i) * 16); break;            // sorted run
            }
            a[i] = ((u64)hi << 32) | (u64)(~rows[i]);
            if ((kind == 0 || kind == 4) && rng() % 13 == 0) a[i] = 0ull;           // "no result"
        }
    }
    u64 *d_in, *d_out; int *d_c, *d_k, *d_nv;
    hipMalloc(&d_in, in.size() * 8); hipMalloc(&d_out, (size_t)ncase * LS_RES_CAP * 8);
    hipMalloc(&d_c, ncase * 4); hipMalloc(&d_k, ncase * 4); hipMalloc(&d_nv, ncase * 4);
    hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_c, cnts.data(), ncase * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_k, ks.data(), ncase * 4, hipMemcpyHostToDevice);
    const size_t smem = ((size_t)keys_cap + res_cap + 256 + 16) * 8 + (8 * 256 + 64) * 4;
    hipFuncSetAttribute((const void*)topk_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipFuncSetAttribute((const void*)topk_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    std::vector<u64> out((size_t)ncase * LS_RES_CAP);
    std::vector<int> nv(ncase);
    int bad = 0;
    for (int pass = 0; pass < 2; ++pass) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(d_out, 0xff, out.size() * 8);
            if (pass == 0) topk_kernel<256><<<ncase, 256, smem>>>(d_in, d_c, d_k, keys_cap, d_out, d_nv, keys_cap, res_cap);
            else topk_kernel<1024><<<ncase, 1024, smem>>>(d_in, d_c, d_k, keys_cap, d_out, d_nv, keys_cap, res_cap);
            if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
            hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
            hipMemcpy(nv.data(), d_nv, ncase * 4, hipMemcpyDeviceToHost);
            for (int c = 0; c < ncase; ++c) {
                std::vector<u64> ref(&in[(size_t)c * keys_cap], &in[(size_t)c * keys_cap] + cnts[c]);
                ref.erase(std::remove(ref.begin(), ref.end(), 0ull), ref.end());
                std::sort(ref.begin(), ref.end(), std::greater<u64>());
                const int want = std::min<int>(ks[c], (int)ref.size());
                bool ok = nv[c] == want;
                for (int i = 0; ok && i < want; ++i) ok = out[(size_t)c * LS_RES_CAP + i] == ref[i];
                if (!ok && bad++ < 10)
                    printf("MISMATCH threads=%d rep=%d case=%d kind=%d cnt=%d k=%d nvalid=%d want=%d\n", pass ? 1024 : 256, rep, c,
                           kinds[c], cnts[c], ks[c], nv[c], want);
            }
        }
    }
    printf("%d cases x 2 thread counts x 3 repeats: %d mismatches\n", ncase, bad);
    return bad ? 1 : 0;
}
