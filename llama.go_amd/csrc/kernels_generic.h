// csrc/kernels_generic.h — one gfx950 kernel per implemented ml.optype (ml.go:1532-1702), full stride
// semantics of the reference.  Used when a graph is not a recognised LLaMA plan (or LH_GRAPH_NO_FUSION),
// so ml.GraphCompute stays a drop-in for arbitrary graphs built from the reference's operator set.
// Strides are in FLOATS here (the host divides Tensor.NB by 4 as the reference does, e.g. ml.go:1786).
#pragma once
#include "kernels_common.h"

namespace lh {

struct TView {  // device view of an ml.Tensor
    float* p;
    uint32_t ne[4];
    uint64_t ns[4];  // strides in floats
};

// GetRows ml.go:1711-1750: dst[i,:] = src0[uint32(src1[i]),:], row pitch = NE[0] on both sides (ml.go:1748)
__global__ __launch_bounds__(256) void g_get_rows(TView src0, TView src1, TView dst) {
    const uint32_t i = blockIdx.x;
    const uint32_t r = (uint32_t)src1.p[i];
    const uint32_t nc = src0.ne[0];
    for (uint32_t c = threadIdx.x; c < nc; c += 256) dst.p[(size_t)i * dst.ne[0] + c] = src0.p[(size_t)r * src0.ne[0] + c];
}

// RMSNorm ml.go:1753-1812 (no weight): one workgroup per row (i01, i02, i03)
__global__ __launch_bounds__(256) void g_rms_norm(TView src0, TView dst) {
    __shared__ double sred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t row = blockIdx.x;
    const uint32_t i01 = row % src0.ne[1]; row /= src0.ne[1];
    const uint32_t i02 = row % src0.ne[2];
    const uint32_t i03 = row / src0.ne[2];
    const float* x = src0.p + i01 * src0.ns[1] + i02 * src0.ns[2] + i03 * src0.ns[3];
    float* y = dst.p + i01 * dst.ns[1] + i02 * dst.ns[2] + i03 * dst.ns[3];
    const uint32_t n = src0.ne[0];
    double s = 0.0;
    for (uint32_t i = tid; i < n; i += 256) s += (double)__fmul_rn(x[i], x[i]);
    s = wave_sum_f64(s);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    const double mean = (((sred[0] + sred[1]) + sred[2]) + sred[3]) / (double)n;
    const float scale = (float)(1.0 / sqrt(mean + 1e-5));
    for (uint32_t i = tid; i < n; i += 256) y[i] = __fmul_rn(x[i], scale);
}

// Repeat ml.go:1822-1868 (2-D broadcast)
__global__ __launch_bounds__(256) void g_repeat(TView src0, TView dst) {
    const uint32_t nc = dst.ne[0], nr = dst.ne[1], nc0 = src0.ne[0], nr0 = src0.ne[1];
    const uint64_t total = (uint64_t)nc * nr;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint32_t col = (uint32_t)(e % nc), row = (uint32_t)(e / nc);
        const uint32_t k = row % nr0, c = col % nc0;
        dst.p[(size_t)row * dst.ns[1] + (size_t)col * dst.ns[0]] = src0.p[(size_t)k * src0.ns[1] + c];
    }
}

// Mul ml.go:1877-1914: rows addressed by NE[0] on all three tensors (flat when contiguous)
__global__ __launch_bounds__(256) void g_mul(TView src0, TView src1, TView dst) {
    const uint64_t total = (uint64_t)src0.ne[0] * src0.ne[1] * src0.ne[2] * src0.ne[3];
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256)
        dst.p[e] = __fmul_rn(src0.p[e], src1.p[e]);
}

// Add ml.go:2515-2584 (src1 contiguous in dim 0): rows by nb1 strides
__global__ __launch_bounds__(256) void g_add(TView src0, TView src1, TView dst) {
    const uint32_t nc = src0.ne[0];
    const uint64_t n = (uint64_t)src0.ne[1] * src0.ne[2] * src0.ne[3];
    const uint64_t total = n * nc;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint64_t j = e / nc;
        const uint32_t i = (uint32_t)(e % nc);
        dst.p[j * dst.ns[1] + i] = __fadd_rn(src0.p[j * src0.ns[1] + i], src1.p[j * src1.ns[1] + i]);
    }
}

// Silu ml.go:2599-2644
__global__ __launch_bounds__(256) void g_silu(TView src0, TView dst) {
    const uint32_t nc = src0.ne[0];
    const uint64_t total = (uint64_t)src0.ne[1] * src0.ne[2] * src0.ne[3] * nc;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint64_t j = e / nc;
        const uint32_t i = (uint32_t)(e % nc);
        dst.p[j * dst.ns[1] + i] = silu_ref(src0.p[j * src0.ns[1] + i]);
    }
}

// Scale ml.go:2331-2374: in place, factor read from src1.Data[0]
__global__ __launch_bounds__(256) void g_scale(TView dst, const float* v) {
    const float f = *v;
    const uint32_t nc = dst.ne[0];
    const uint64_t total = (uint64_t)dst.ne[1] * dst.ne[2] * dst.ne[3] * nc;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint64_t j = e / nc;
        const uint32_t i = (uint32_t)(e % nc);
        float* p = dst.p + j * dst.ns[1] + i;
        *p = __fmul_rn(*p, f);
    }
}

// Cpy / Dup ml.go:2110-2240: destination contiguous, source gathered through its strides.  The three host
// branches (bulk copy, row copy, element gather) all enumerate (i03,i02,i01,i00) into consecutive dst slots.
__global__ __launch_bounds__(256) void g_cpy(TView src0, float* dst, uint64_t total) {
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        uint64_t r = e;
        const uint32_t i00 = (uint32_t)(r % src0.ne[0]); r /= src0.ne[0];
        const uint32_t i01 = (uint32_t)(r % src0.ne[1]); r /= src0.ne[1];
        const uint32_t i02 = (uint32_t)(r % src0.ne[2]);
        const uint32_t i03 = (uint32_t)(r / src0.ne[2]);
        dst[e] = src0.p[i00 * src0.ns[0] + i01 * src0.ns[1] + i02 * src0.ns[2] + i03 * src0.ns[3]];
    }
}

// DiagMaskInf ml.go:2377-2414: dst[k,j,i] = -inf for i > past + j (in place)
__global__ __launch_bounds__(256) void g_diag_mask_inf(TView dst, const float* past_f) {
    const uint32_t past = (uint32_t)*past_f;
    const uint32_t nc = dst.ne[0], nr = dst.ne[1];
    const uint32_t nz = dst.ne[2] * dst.ne[3];
    const uint64_t total = (uint64_t)nz * nr * nc;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        const uint32_t i = (uint32_t)(e % nc);
        const uint32_t j = (uint32_t)((e / nc) % nr);
        const uint32_t k = (uint32_t)(e / ((uint64_t)nc * nr));
        if (i >= past && i > past + j) dst.p[(size_t)k * dst.ns[2] + (size_t)j * dst.ns[1] + (size_t)i * dst.ns[0]] = -INFINITY;
    }
}

// SoftMax ml.go:2432-2505: one workgroup per contiguous row, in place
__global__ __launch_bounds__(256) void g_soft_max(TView dst) {
    __shared__ float scratch[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* p = dst.p + (size_t)blockIdx.x * dst.ns[1];
    const uint32_t nc = dst.ne[0];
    float m = -INFINITY;
    for (uint32_t i = tid; i < nc; i += 256) m = fmaxf(m, p[i]);
    m = wave_max(m);
    if (lane == 0) scratch[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    float s = 0.f;
    for (uint32_t i = tid; i < nc; i += 256) {
        float v = p[i];
        if (v == -INFINITY) v = 0.f;
        else { v = (float)exp((double)__fsub_rn(v, m)); s += v; }
        p[i] = v;
    }
    s = wave_sum(s);
    if (lane == 0) scratch[wave] = s;
    __syncthreads();
    const float inv = __fdiv_rn(1.0f, (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]));
    for (uint32_t i = tid; i < nc; i += 256) p[i] = __fmul_rn(p[i], inv);
}

// Rope ml.go:2253-2328: in place on adjacent pairs; cos/sin from the f64 table (ensure_rope_table)
__global__ __launch_bounds__(256) void g_rope(TView t, const double2* __restrict__ rope, uint32_t past, uint32_t dims, uint32_t mode) {
    const uint32_t half = dims >> 1;
    const uint32_t mode_count = mode == 0 ? 0 : past;
    if (t.ne[2] <= mode_count) return;
    const uint32_t n2 = t.ne[2] - mode_count;
    const uint64_t total = (uint64_t)t.ne[3] * n2 * t.ne[1] * half;
    for (uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (uint64_t)gridDim.x * 256) {
        uint64_t r = e;
        const uint32_t ih = (uint32_t)(r % half); r /= half;
        const uint32_t i1 = (uint32_t)(r % t.ne[1]); r /= t.ne[1];
        const uint32_t i2 = (uint32_t)(r % n2) + mode_count;
        const uint32_t i3 = (uint32_t)(r / n2);
        const uint32_t p = mode == 0 ? past + i2 : i2;
        float* x = t.p + i3 * t.ns[3] + i2 * t.ns[2] + i1 * t.ns[1] + (uint64_t)(2 * ih) * t.ns[0];
        float o0, o1;
        rope_rotate(x[0], x[1], rope[(size_t)p * half + ih], &o0, &o1);
        x[0] = o0;
        x[1] = o1;
    }
}

// MulMat ml.go:1976-2098, general strided case: one wave per output element dst[i3,i2,ic,i01]
// (K-contiguous operands as the reference requires, ml.go:1950, 1967).
__global__ __launch_bounds__(256) void g_mul_mat(TView src0, TView src1, TView dst) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave_id = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint32_t ne00 = src0.ne[0], ne01 = src0.ne[1], ne02 = src0.ne[2], ne11 = src1.ne[1];
    const uint64_t nr = (uint64_t)ne01 * ne02 * src0.ne[3];
    if (wave_id >= nr * ne11) return;
    const uint32_t ic = (uint32_t)(wave_id % ne11);
    uint64_t ir = wave_id / ne11;
    const uint32_t i01 = (uint32_t)(ir % ne01); ir /= ne01;
    const uint32_t i02 = (uint32_t)(ir % ne02);
    const uint32_t i03 = (uint32_t)(ir / ne02);
    const float* a = src0.p + i01 * src0.ns[1] + i02 * src0.ns[2] + i03 * src0.ns[3];
    const float* b = src1.p + ic * src1.ns[1] + i02 * src1.ns[2] + i03 * src1.ns[3];
    float s = 0.f;
    for (uint32_t k = lane; k < ne00; k += 64) s = fmaf(a[k], b[k], s);
    s = wave_sum(s);
    if (lane == 0) dst.p[i01 * dst.ns[0] + ic * dst.ns[1] + i02 * dst.ns[2] + i03 * dst.ns[3]] = s;
}

}  // namespace lh
