// csrc/kernels_common.h — device helpers shared by every kernel (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lh {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// Device-resident per-step parameters of a decode graph: kernels read `past`/`token` from here so a
// captured hipGraph can be replayed for every position without node updates.
struct StepParams {
    uint32_t token;  // token id evaluated by this step
    uint32_t past;   // llama.Eval's pastCount for this step
    uint32_t step;   // index into the resident loop's output array
    uint32_t pad;
};

// One row of a BATCHED Eval (lh_batch: the pods of a rank evaluated in one weight pass, server.go:88-101 runs them as independent
// llama.Contexts): the row's own KV cache (llama.go:173-178; base of the stage's first layer slot) and its position = pastCount.
// The table lives in device memory so that one captured hipGraph serves every tick; k_batch_argmax / k_batch_advance move `pos`.
struct BatchRow {
    float* kc;       // this pod's K cache [layers of the stage][ctx][d]
    float* vc;
    uint32_t pos;    // position of the row's token (= keys already cached for its stream)
    uint32_t step;   // ticks since the row was last set: index into the row's output list
};

// Weights are read exactly once per token and shared by no other CU: stream them with the
// non-temporal policy (MI355X_MICROARCH "nt-weights": +10-20 % on this access pattern, see
// profiles/r01_gemv_probe.txt).
__device__ __forceinline__ f4 ld_nt(const f4* p) { return __builtin_nontemporal_load(p); }

// Sum over each row of 16 lanes with DPP (no LDS traffic); every lane ends with its row's sum.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}
__device__ __forceinline__ float rdlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// Full 64-lane sum, result uniform in every lane.  Fixed association -> deterministic.
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}
// Full 64-lane sum in 6 DPP adds; the total is valid in LANE 63 only (row_bcast15 / row_bcast31 are gfx9 DPP controls).
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = row16_sum(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast15 -> rows 1,3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast31 -> rows 2,3
    return v;
}
// Sum over aligned groups of 32 lanes (two DPP rows); result valid in every lane of the group.
__device__ __forceinline__ float half_wave_sum(float v) {
    v = row16_sum(v);
    return v + __shfl_xor(v, 16, 64);
}
// Maximum over the 64 lanes, uniform in every lane: four DPP steps inside the rows of 16 + four lane reads (the butterfly of six ds_bpermute shuffles it
// replaces, round 6, stood on the decode attention's critical path - ~0.2 us of a 2 us kernel body; a maximum does not depend on the order it is taken in).
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false)));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false)));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false)));  // row_half_mirror
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false)));  // row_mirror
    return fmaxf(fmaxf(rdlane(v, 0), rdlane(v, 16)), fmaxf(rdlane(v, 32), rdlane(v, 48)));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The block of weight rows [r0, r1) of workgroup b in the weight-stream kernels: the M / 2 row PAIRS dealt as evenly as possible, the first
// `wg_r` workgroups one pair more, the last workgroup also an odd last row.  wg_q / wg_r = quotient and remainder of (M / 2) / #workgroups, computed
// by the HOST (launch_* in plan.hip).  Until round 6 every kernel computed floor(b * pairs / #workgroups) itself: two 64-bit divisions by a run-time
// value - ~250 scalar and vector instructions in front of the first weight load of a launch that lasts 5..60 us (ISA of k_gemv_q8s; the decode step
// has 160 such launches).  wg_q == wg_r == 0 (a caller that does not fill them: the probes under tools/) keeps the old formula.  Which rows a
// workgroup owns changes no sum.
__device__ __forceinline__ void wg_row_block(uint32_t M, uint32_t wg_q, uint32_t wg_r, uint32_t* r0, uint32_t* r1) {
    const uint32_t b = blockIdx.x, nwg = gridDim.x;
    if (wg_q | wg_r) {
        *r0 = 2u * (b * wg_q + (b < wg_r ? b : wg_r));
        *r1 = (b + 1 == nwg) ? M : 2u * ((b + 1) * wg_q + (b + 1 < wg_r ? b + 1 : wg_r));
    } else {
        const uint32_t npairs = M >> 1;
        *r0 = 2u * (uint32_t)(((uint64_t)b * npairs) / nwg);
        *r1 = (b + 1 == nwg) ? M : 2u * (uint32_t)(((uint64_t)(b + 1) * npairs) / nwg);
    }
}

// Kernel arguments are read with scalar loads where the code first needs them; the compiler sinks those loads into the branches that use a field, so a
// kernel that tests one argument, then follows another, then a third pays a scalar-cache miss per 64-byte line of its argument block ONE AFTER THE OTHER
// (k_attention: rows -> sp -> the position: three dependent misses in front of its first cache row).  Naming one field of every line in an empty asm at
// the top of the kernel makes all lines arrive behind one wait; the later loads hit the scalar cache.
#define LH_TOUCH_ARGS(...) do { lh::touch_args_(__VA_ARGS__); } while (0)
template <typename A, typename B> __device__ __forceinline__ void touch_args_(A a, B b) { asm volatile("" ::"s"(a), "s"(b)); }
template <typename A, typename B, typename C> __device__ __forceinline__ void touch_args_(A a, B b, C c) { asm volatile("" ::"s"(a), "s"(b), "s"(c)); }
template <typename A, typename B, typename C, typename D> __device__ __forceinline__ void touch_args_(A a, B b, C c, D d) { asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d)); }
template <typename A, typename B, typename C, typename D, typename E, typename F> __device__ __forceinline__ void touch_args_(A a, B b, C c, D d, E e, F f) {
    asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d), "s"(e), "s"(f));
}

// A pointer pinned into a scalar register pair and made opaque to the optimiser.
__device__ __forceinline__ const char* sgpr_ptr(const void* p) {
    uint32_t lo = (uint32_t)(uintptr_t)p, hi = (uint32_t)((uintptr_t)p >> 32);
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (const char*)(((uintptr_t)hi << 32) | lo);
}

// fp32 helpers with the reference's rounding sequence (no contraction):
//   SiLU  ml.go:2587-2589   x / float32(1 + exp(float64(-x)))
__device__ __forceinline__ float silu_ref(float x) {
    const float den = (float)(1.0 + exp((double)(-x)));
    return __fdiv_rn(x, den);
}
//   RoPE  ml.go:2318-2322   rotation in f64, one rounding per output
__device__ __forceinline__ void rope_rotate(float x0f, float x1f, double2 cs, float* o0, float* o1) {
    const double x0 = (double)x0f, x1 = (double)x1f;
    *o0 = (float)__dsub_rn(__dmul_rn(x0, cs.x), __dmul_rn(x1, cs.y));
    *o1 = (float)__dadd_rn(__dmul_rn(x0, cs.y), __dmul_rn(x1, cs.x));
}

}  // namespace lh
