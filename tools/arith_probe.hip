// tools/arith_probe.hip — which CPU formula reproduces gfx950's v_mfma_f32_16x16x4_f32 and
// v_dot2_f32_f16 bit for bit? (developer probe: the kernel-order oracle mode of
// oracle/flat_ip_ref.c restates whatever this program reports.)
//
//   hipcc --offload-arch=gfx950 -O2 tools/arith_probe.hip -o /tmp/arith_probe && /tmp/arith_probe
//
// MFMA 16x16x4 f32: D[i][j] = C[i][j] + sum_k A[i][k] * B[k][j], k = 0..3. Candidate models:
//   fma_fwd   : acc = fmaf(A[i][k], B[k][j], acc) for k = 0, 1, 2, 3
//   fma_rev   : the same for k = 3, 2, 1, 0
//   mul_add   : acc = acc + round(A*B), k ascending (no fusion)
//   exact_once: round(C + exact sum of the four exact products) (one rounding, __float128)
//   pair_tree : fma(a0,b0, fma(a1,b1,0))... two halves summed, then + C
// dot2: d = a.x*b.x + a.y*b.y + c (fp16 operands). Candidates:
//   fma_xy    : fmaf(a.y, b.y, fmaf(a.x, b.x, c))
//   fma_yx    : fmaf(a.x, b.x, fmaf(a.y, b.y, c))
//   exact_once: round(c + a.x*b.x + a.y*b.y) with the products and the sum exact (double is enough:
//               11-bit x 11-bit products, a 24-bit addend... checked with __float128 anyway)
//   prod_first: round(round(a.x*b.x + a.y*b.y) + c)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));

__global__ void mfma_kernel(const float* A, const float* B, const float* C, float* D, int tiles) {
    // one wave per tile: A [16][4], B [4][16], C/D [16][16] row-major
    const int t = blockIdx.x, lane = threadIdx.x;
    const float* a = A + t * 64;
    const float* b = B + t * 64;
    const float* c = C + t * 256;
    float* d = D + t * 256;
    const int li = lane & 15, qd = lane >> 4;
    f32x4v acc;
    for (int r = 0; r < 4; ++r) acc[r] = c[(4 * qd + r) * 16 + li];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[li * 4 + qd], b[qd * 16 + li], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[(4 * qd + r) * 16 + li] = acc[r];
}

__global__ void dot2_kernel(const unsigned* a, const unsigned* b, const float* c, float* d, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    d[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_t, a[i]), __builtin_bit_cast(h2_t, b[i]), c[i], false);
}

static unsigned bits(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static bool same(float x, float y) { return bits(x) == bits(y) || (x != x && y != y); }

static float h2f(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }

int main() {
    std::mt19937_64 rng(12345);
    auto rnd_f = [&](int mode) {
        // mode 0: normal-ish values of similar magnitude (what a corpus looks like)
        // mode 1: wide exponent range (exposes intermediate rounding)
        std::normal_distribution<float> nd(0.0f, 1.0f);
        float v = nd(rng);
        if (mode == 1) v = ldexpf(v, (int)(rng() % 40) - 20);
        return v;
    };
    for (int mode = 0; mode < 2; ++mode) {
        const int tiles = 4096;
        std::vector<float> A(tiles * 64), B(tiles * 64), C(tiles * 256), D(tiles * 256);
        for (auto& v : A) v = rnd_f(mode);
        for (auto& v : B) v = rnd_f(mode);
        for (auto& v : C) v = rnd_f(mode);
        float *dA, *dB, *dC, *dD;
        hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
        hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_kernel, dim3(tiles), dim3(64), 0, 0, dA, dB, dC, dD, tiles);
        hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
        long bad[5] = {0, 0, 0, 0, 0}, total = 0;
        for (int t = 0; t < tiles; ++t)
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    const float* a = &A[t * 64 + i * 4];
                    float bk[4];
                    for (int k = 0; k < 4; ++k) bk[k] = B[t * 64 + k * 16 + j];
                    const float c = C[t * 256 + i * 16 + j], got = D[t * 256 + i * 16 + j];
                    float m0 = c, m1 = c, m2 = c;
                    for (int k = 0; k < 4; ++k) m0 = fmaf(a[k], bk[k], m0);
                    for (int k = 3; k >= 0; --k) m1 = fmaf(a[k], bk[k], m1);
                    for (int k = 0; k < 4; ++k) { volatile float p = a[k] * bk[k]; m2 = m2 + p; }
                    __float128 e = c;
                    for (int k = 0; k < 4; ++k) e += (__float128)a[k] * (__float128)bk[k];
                    const float m3 = (float)e;
                    const float m4 = (fmaf(a[1], bk[1], a[0] * bk[0]) + fmaf(a[3], bk[3], a[2] * bk[2])) + c;
                    ++total;
                    bad[0] += !same(got, m0); bad[1] += !same(got, m1); bad[2] += !same(got, m2);
                    bad[3] += !same(got, m3); bad[4] += !same(got, m4);
                }
        printf("mfma_f32_16x16x4 mode %d: %ld outputs; mismatches fma_fwd %ld fma_rev %ld mul_add %ld exact_once %ld pair_tree %ld\n",
               mode, total, bad[0], bad[1], bad[2], bad[3], bad[4]);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    }
    // a chain of MFMAs over a long K (what the kernels do): C carried across instructions
    {
        const int K = 384;
        std::vector<float> a(16 * K), b(K * 16);
        std::normal_distribution<float> nd(0.0f, 1.0f);
        for (auto& v : a) v = nd(rng);
        for (auto& v : b) v = nd(rng);
        // reuse the one-tile kernel K/4 times on the host side of the chain
        std::vector<float> C(256, 0.0f), D(256);
        float *dA, *dB, *dC, *dD;
        hipMalloc(&dA, 64 * 4); hipMalloc(&dB, 64 * 4); hipMalloc(&dC, 256 * 4); hipMalloc(&dD, 256 * 4);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        for (int ks = 0; ks < K / 4; ++ks) {
            float ta[64], tb[64];
            for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) ta[i * 4 + k] = a[i * K + ks * 4 + k];
            for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) tb[k * 16 + j] = b[(ks * 4 + k) * 16 + j];
            hipMemcpy(dA, ta, 256, hipMemcpyHostToDevice);
            hipMemcpy(dB, tb, 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, 1);
            hipMemcpy(dC, dD, 1024, hipMemcpyDeviceToDevice);
        }
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        long badc = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            float m = 0.0f;
            for (int k = 0; k < K; ++k) m = fmaf(a[i * K + k], b[k * 16 + j], m);
            badc += !same(m, D[i * 16 + j]);
        }
        printf("mfma chain over K=%d vs sequential fmaf chain: %ld of 256 differ\n", K, badc);
    }
    for (int mode = 0; mode < 3; ++mode) {
        const int n = 1 << 20;
        std::vector<unsigned> a(n), b(n);
        std::vector<float> c(n), d(n);
        std::normal_distribution<float> nd(0.0f, 1.0f);
        auto rnd_h = [&]() -> unsigned short {
            float v = nd(rng);
            if (mode == 1) v = ldexpf(v, (int)(rng() % 20) - 10);
            if (mode == 2) v = ldexpf(v, -(int)(rng() % 16));  // small values: products in the fp32-normal, fp16-subnormal range
            return __builtin_bit_cast(unsigned short, (_Float16)v);
        };
        for (int i = 0; i < n; ++i) {
            a[i] = rnd_h() | ((unsigned)rnd_h() << 16);
            b[i] = rnd_h() | ((unsigned)rnd_h() << 16);
            c[i] = rnd_f(mode == 1 ? 1 : 0);
        }
        unsigned *da, *db; float *dc, *dd;
        hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
        hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(dot2_kernel, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dd, n);
        hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
        long bad[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const float ax = h2f(a[i] & 0xffff), ay = h2f(a[i] >> 16), bx = h2f(b[i] & 0xffff), by = h2f(b[i] >> 16);
            const float m0 = fmaf(ay, by, fmaf(ax, bx, c[i]));
            const float m1 = fmaf(ax, bx, fmaf(ay, by, c[i]));
            const float m2 = (float)((__float128)ax * bx + (__float128)ay * by + (__float128)c[i]);
            const float pp = (float)((double)ax * bx + (double)ay * by);  // exact in double, one rounding to fp32
            volatile float m3 = pp + c[i];
            // products summed exactly, the sum TRUNCATED/rounded with the addend in one go but fp16 subnormal
            // inputs flushed: a fifth model
            auto ftz = [](float v) { return fabsf(v) < 6.103515625e-05f ? copysignf(0.0f, v) : v; };
            const float m4 = (float)((__float128)ftz(ax) * ftz(bx) + (__float128)ftz(ay) * ftz(by) + (__float128)c[i]);
            bad[0] += !same(d[i], m0); bad[1] += !same(d[i], m1); bad[2] += !same(d[i], m2);
            bad[3] += !same(d[i], m3); bad[4] += !same(d[i], m4);
        }
        printf("v_dot2_f32_f16 mode %d: %d outputs; mismatches fma_xy %ld fma_yx %ld exact_once %ld prod_first %ld exact_once_ftz16 %ld\n",
               mode, n, bad[0], bad[1], bad[2], bad[3], bad[4]);
        hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    }
    return 0;
}
