// csrc/kernels_q8_pack.h — block-int8 quantiser and interchange-block de-interleave (format: see kernels_q8.h).
#pragma once
#include "kernels_common.h"

namespace lh {

// Quantiser: one thread per block of 32 weights of an fp32 matrix [rows][K] -> planes.
__global__ __launch_bounds__(256) void k_quantize_q8(const float* __restrict__ src, signed char* __restrict__ q, float* __restrict__ scales, uint64_t nblocks) {
    uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (; b < nblocks; b += stride) {
        f4 v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = ((const f4*)src)[b * 8 + i];
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(fmaxf(fmaxf(m, fabsf(v[i].x)), fabsf(v[i].y)), fmaxf(fabsf(v[i].z), fabsf(v[i].w)));
        const float d = __fdiv_rn(m, 127.0f);
        unsigned int packed[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int e[4];
            const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t = d > 0.f ? rintf(__fdiv_rn(f[k], d)) : 0.f;
                t = fminf(fmaxf(t, -127.f), 127.f);
                e[k] = (int)t;
            }
            packed[i] = (unsigned)(e[0] & 255) | ((unsigned)(e[1] & 255) << 8) | ((unsigned)(e[2] & 255) << 16) | ((unsigned)(e[3] & 255) << 24);
        }
        u4* dst = (u4*)(q + b * 32);
        dst[0] = u4{packed[0], packed[1], packed[2], packed[3]};
        dst[1] = u4{packed[4], packed[5], packed[6], packed[7]};
        scales[b] = d;
    }
}

// Registration from the 36-byte interchange blocks { float d; int8 q[32] } -> planes.
__global__ __launch_bounds__(256) void k_q8_deinterleave(const unsigned int* __restrict__ blocks, signed char* __restrict__ q, float* __restrict__ scales, uint64_t nblocks) {
    uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (; b < nblocks; b += stride) {
        const unsigned int* s = blocks + b * 9;
        scales[b] = __builtin_bit_cast(float, s[0]);
        unsigned int* dst = (unsigned int*)(q + b * 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i] = s[1 + i];
    }
}

}  // namespace lh
