// tools/mfma16_probe.hip — how does v_mfma_f32_16x16x32_f16 round? (round 5: could the fp16 MFMA paths get a
// zero-excuse kernel-order oracle like the fp32 ones?)  D[i][j] = C[i][j] + sum_{k<32} A[i][k] * B[k][j], fp16 inputs.
// Candidate CPU models (products of two fp16 values are exact in fp32; partial sums taken in __float128 = exact):
//   fma_seq   : acc = fmaf(a_k, b_k, acc), k = 0..31
//   once      : round(C + exact sum of the 32 products)                       (one rounding)
//   blk8/4/2  : for each block of 8 / 4 / 2 consecutive k: acc = round(acc + exact block sum)
//   blk8_first: round(C + round-to-f32 of each block's exact sum, blocks added left to right) variants are covered by blkN
//   hipcc --offload-arch=gfx950 -O2 tools/mfma16_probe.hip -o scratch/mfma16_probe && scratch/mfma16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ void k16(const _Float16* A, const _Float16* B, const float* C, float* D) {
    // per tile: A [16][32] row-major, B [32][16] row-major, C/D [16][16]
    const int t = blockIdx.x, lane = threadIdx.x, li = lane & 15, qd = lane >> 4;
    const _Float16* a = A + t * 512; const _Float16* b = B + t * 512;
    half8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = a[li * 32 + 8 * qd + e]; bv[e] = b[(8 * qd + e) * 16 + li]; }
    f32x4v acc;
    for (int r = 0; r < 4; ++r) acc[r] = C[t * 256 + (4 * qd + r) * 16 + li];
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[t * 256 + (4 * qd + r) * 16 + li] = acc[r];
}
static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
int main() {
    std::mt19937_64 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int mode = 0; mode < 3; ++mode) {
        const int tiles = 2048;
        std::vector<_Float16> A(tiles * 512), B(tiles * 512);
        std::vector<float> C(tiles * 256), D(tiles * 256);
        auto rh = [&]() { float v = nd(rng); if (mode == 1) v = ldexpf(v, (int)(rng() % 16) - 8); if (mode == 2) v *= 0.05f; return (_Float16)v; };
        for (auto& v : A) v = rh(); for (auto& v : B) v = rh();
        for (auto& v : C) v = mode == 2 ? 0.0f : nd(rng);
        _Float16 *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k16, dim3(tiles), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        long bad[6] = {0, 0, 0, 0, 0, 0}, tot = 0;
        for (int t = 0; t < tiles; ++t) for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float a[32], b[32];
            for (int k = 0; k < 32; ++k) { a[k] = (float)A[t * 512 + i * 32 + k]; b[k] = (float)B[t * 512 + k * 16 + j]; }
            const float c = C[t * 256 + i * 16 + j], got = D[t * 256 + i * 16 + j];
            float m0 = c; for (int k = 0; k < 32; ++k) m0 = fmaf(a[k], b[k], m0);
            __float128 e = c; for (int k = 0; k < 32; ++k) e += (__float128)a[k] * b[k];
            const float m1 = (float)e;
            float mb[3]; int bs[3] = {8, 4, 2};
            for (int v = 0; v < 3; ++v) { float acc = c; for (int k0 = 0; k0 < 32; k0 += bs[v]) { __float128 s = acc; for (int k = k0; k < k0 + bs[v]; ++k) s += (__float128)a[k] * b[k]; acc = (float)s; } mb[v] = acc; }
            // interleaved lanes: block of the 4 k-quarters' e-th elements (k = e, 8+e, 16+e, 24+e)
            float mi = c; for (int ee = 0; ee < 8; ++ee) { __float128 s = mi; for (int q = 0; q < 4; ++q) s += (__float128)a[8 * q + ee] * b[8 * q + ee]; mi = (float)s; }
            ++tot; bad[0] += bits(got) != bits(m0); bad[1] += bits(got) != bits(m1); bad[2] += bits(got) != bits(mb[0]);
            bad[3] += bits(got) != bits(mb[1]); bad[4] += bits(got) != bits(mb[2]); bad[5] += bits(got) != bits(mi);
        }
        printf("mfma_f32_16x16x32_f16 mode %d: %ld outputs; mismatches fma_seq %ld once %ld blk8 %ld blk4 %ld blk2 %ld interleaved4 %ld\n",
               mode, tot, bad[0], bad[1], bad[2], bad[3], bad[4], bad[5]);
    }
    return 0;
}
