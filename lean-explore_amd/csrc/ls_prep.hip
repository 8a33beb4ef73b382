// ls_prep.hip — query preparation and corpus layout conversion.
//
//   ls_prep_kernel    : faiss.normalize_L2 (reference src/lean_explore/search/engine.py:242)
//                       fused with zero padding to the stored row length and, for an fp16
//                       index, rounding of the query to fp16-representable values.
//   ls_convert_kernel : fp32 [n, d] (the array index.add receives, reference
//                       src/lean_explore/extract/index.py:71,116) -> HBM layout [n, d_pad]
//                       in fp32 or fp16, zero padded to whole 16-byte chunks.
#include "ls_common.h"

// One wave per query: the squared norm in the library's canonical order (ls_wave_sumsq).
__global__ __launch_bounds__(64) void ls_prep_kernel(const float* __restrict__ qin,
                                                     float* __restrict__ qout, int d, int d_pad,
                                                     int normalize, int round_f16) {
    const long long qi = blockIdx.x;
    const float* src = qin + qi * d;
    float* dst = qout + qi * d_pad;
    float inv = 1.0f;
    if (normalize) {
        const float ss = ls_wave_sumsq(src, d, threadIdx.x);
        if (ss > 0.0f) inv = 1.0f / sqrtf(ss);  // zero-norm rows stay untouched (fvec_renorm_L2)
    }
    for (int j = threadIdx.x; j < d_pad; j += 64) {
        float v = 0.0f;
        if (j < d) {
            v = normalize ? src[j] * inv : src[j];
            if (round_f16) v = (float)(_Float16)v;
        }
        dst[j] = v;
    }
}

int ls_launch_prep(const float* d_q_in, float* d_q_out, int64_t nq, const ls_geom& g,
                   bool normalize, bool round_f16, hipStream_t s) {
    if (nq <= 0) return LS_OK;
    hipLaunchKernelGGL(ls_prep_kernel, dim3((unsigned)nq), dim3(64), 0, s, d_q_in, d_q_out, g.d,
                       g.d_pad, normalize ? 1 : 0, round_f16 ? 1 : 0);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

template <bool F16>
__global__ __launch_bounds__(256) void ls_convert_kernel(const float* __restrict__ src,
                                                         uint4* __restrict__ dst, long long n,
                                                         int d, int chunks) {
    constexpr int E = F16 ? 8 : 4;  // elements per 16-byte chunk
    const long long total = n * chunks;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (long long)gridDim.x * 256) {
        const long long r = i / chunks;
        const int c = (int)(i - r * chunks);
        const float* p = src + r * d;
        float v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = c * E + e;
            v[e] = j < d ? p[j] : 0.0f;
        }
        uint4 o;
        if (F16) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            o.x = __builtin_bit_cast(u32, h2{(_Float16)v[0], (_Float16)v[1]});
            o.y = __builtin_bit_cast(u32, h2{(_Float16)v[2], (_Float16)v[3]});
            o.z = __builtin_bit_cast(u32, h2{(_Float16)v[4 % E], (_Float16)v[5 % E]});
            o.w = __builtin_bit_cast(u32, h2{(_Float16)v[6 % E], (_Float16)v[7 % E]});
        } else {
            o.x = __builtin_bit_cast(u32, v[0]);
            o.y = __builtin_bit_cast(u32, v[1]);
            o.z = __builtin_bit_cast(u32, v[2]);
            o.w = __builtin_bit_cast(u32, v[3]);
        }
        dst[i] = o;
    }
}

int ls_launch_convert(const float* d_src, void* d_dst, int64_t n, const ls_geom& g,
                      hipStream_t s) {
    if (n <= 0) return LS_OK;
    const long long total = (long long)n * g.chunks;
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (g.elem == 2)
        hipLaunchKernelGGL((ls_convert_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s,
                           d_src, (uint4*)d_dst, (long long)n, g.d, g.chunks);
    else
        hipLaunchKernelGGL((ls_convert_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s,
                           d_src, (uint4*)d_dst, (long long)n, g.d, g.chunks);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

// stored rows -> fp32 [n, d] (index.reconstruct_n): fp16 storage yields the rounded values
template <bool F16>
__global__ __launch_bounds__(256) void ls_unconvert_kernel(const unsigned char* __restrict__ src,
                                                           float* __restrict__ dst, long long n,
                                                           int d, int chunks) {
    const long long total = n * d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (long long)gridDim.x * 256) {
        const long long r = i / d;
        const int j = (int)(i - r * d);
        const unsigned char* row = src + r * (long long)chunks * 16;
        dst[i] = F16 ? (float)reinterpret_cast<const _Float16*>(row)[j]
                     : reinterpret_cast<const float*>(row)[j];
    }
}

int ls_launch_unconvert(const void* d_src, float* d_dst, int64_t n, const ls_geom& g,
                        hipStream_t s) {
    if (n <= 0) return LS_OK;
    long long blocks = ((long long)n * g.d + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    if (g.elem == 2)
        hipLaunchKernelGGL((ls_unconvert_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s,
                           (const unsigned char*)d_src, d_dst, (long long)n, g.d, g.chunks);
    else
        hipLaunchKernelGGL((ls_unconvert_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s,
                           (const unsigned char*)d_src, d_dst, (long long)n, g.d, g.chunks);
    LS_HIP(hipGetLastError());
    return LS_OK;
}

int ls_pick_geom(int32_t d, int32_t dtype, ls_geom* g) {
    if (d <= 0 || (dtype != LS_DTYPE_F32 && dtype != LS_DTYPE_F16)) return LS_ERR_INVALID_ARG;
    const int elem = dtype == LS_DTYPE_F16 ? 2 : 4;
    const int per = 16 / elem;
    const int raw = (d + per - 1) / per;
    static const int sizes[8][3] = {{16, 16, 1}, {32, 16, 2}, {48, 16, 3},  {64, 16, 4},
                                    {96, 32, 3}, {128, 32, 4}, {192, 64, 3}, {256, 64, 4}};
    for (int i = 0; i < 8; ++i) {
        if (sizes[i][0] >= raw) {
            g->d = d;
            g->chunks = sizes[i][0];
            g->L = sizes[i][1];
            g->V = sizes[i][2];
            g->elem = elem;
            g->qg4 = 0;
            g->d_pad = g->chunks * per;
            return LS_OK;
        }
    }
    return LS_ERR_INVALID_ARG;
}
