// tools/valu_mfma_probe.hip — what can a wave do while ANOTHER wave of its SIMD issues MFMAs back to back?
//
// Why: k_stream_mm2 (csrc/kernels_stream.h) pairs a loader wave and an MFMA wave on every SIMD.  Its traces (profiles/
// r02d_stream_traffic_probe.txt) show the loader's ~20 loads per chunk taking as long as the MFMA wave's burst, with or without memory
// traffic, and a probe whose loads need no vector ALU instruction for their address takes a third of that.  This probe isolates the
// effect: one workgroup of 8 waves per CU, waves 4-7 run `mfma` back to back (or idle), waves 0-3 time a loop of ONE kind of
// instruction.  Output: shader clocks per side instruction with the MFMA waves idle / busy, and the MFMA waves' own clocks per MFMA.
//   kinds: 0 v_add_u32 (dependent chain)   1 v_lshl_add_u64 (independent)   2 global_load_dwordx4, VGPR address, L1-resident
//          3 ds_write_b128                 4 s_add_u32 chain                 5 buffer_load_dwordx4, SGPR offset, L1-resident
//          6 v_lshl_add_u64 + global_load_dwordx4 pairs (the shipped loader's instruction mix)
// usage: valu_mfma_probe [side_iters [mfma_iters [nop_lo [nop_hi]]]]   (nop range: s_nop arguments tried behind every MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

struct Args { const float* buf; unsigned long long* out; unsigned side_iters, mfma_iters, kind, mfma_on, pace; };

template <int KIND>
__device__ __forceinline__ void side_loop(const Args& a, unsigned long long* clocks) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lane = threadIdx.x & 63;
    unsigned v0 = lane, v1 = lane * 3 + 1;
    unsigned long long p0 = (unsigned long long)(uintptr_t)a.buf + lane * 16u, p1 = p0, p2 = p0, p3 = p0;
    const unsigned long long step = 0;                       // uniform addend of the 64-bit vector adds (kept in SGPRs)
    f4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    const unsigned ldsoff = (threadIdx.x & 255) * 16u;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.buf, 0, 0x7fffffff, 0x00020000);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (unsigned it = 0; it < a.side_iters; ++it) {
        if constexpr (KIND == 0) {
            asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                         "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(v0) : "v"(v1));
        } else if constexpr (KIND == 1) {
            asm volatile("v_lshl_add_u64 %0, %4, 0, %8\n v_lshl_add_u64 %1, %5, 0, %8\n v_lshl_add_u64 %2, %6, 0, %8\n v_lshl_add_u64 %3, %7, 0, %8\n"
                         "v_lshl_add_u64 %0, %4, 0, %8\n v_lshl_add_u64 %1, %5, 0, %8\n v_lshl_add_u64 %2, %6, 0, %8\n v_lshl_add_u64 %3, %7, 0, %8"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "s"(step));
        } else if constexpr (KIND == 2) {
            asm volatile("global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:1024\n global_load_dwordx4 %2, %4, off offset:2048\n"
                         "global_load_dwordx4 %3, %4, off offset:3072\n global_load_dwordx4 %0, %4, off\n global_load_dwordx4 %1, %4, off offset:1024\n"
                         "global_load_dwordx4 %2, %4, off offset:2048\n global_load_dwordx4 %3, %4, off offset:3072\n s_waitcnt vmcnt(0)"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(p0) : "memory");
        } else if constexpr (KIND == 3) {
            asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %1 offset:12288\n"
                         "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:4096\n ds_write_b128 %0, %1 offset:8192\n ds_write_b128 %0, %1 offset:12288\n"
                         "s_waitcnt lgkmcnt(0)" : : "v"(ldsoff), "v"(d0) : "memory");
        } else if constexpr (KIND == 4) {
            unsigned s = it;
            asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n"
                         "s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(s));
            v0 += s & 1;
        } else if constexpr (KIND == 5) {
            const unsigned so = (it & 1) * 1024u;
            u4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u, so, 0), r1 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 1024u, so, 0);
            u4 r2 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 2048u, so, 0), r3 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 3072u, so, 0);
            u4 r4 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 4096u, so, 0), r5 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 5120u, so, 0);
            u4 r6 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 6144u, so, 0), r7 = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16u + 7168u, so, 0);
            v0 += (r0.x ^ r1.x ^ r2.x ^ r3.x ^ r4.x ^ r5.x ^ r6.x ^ r7.x) & 1u;
        } else {
            asm volatile("v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %1, %0, off\n v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %2, %0, off offset:1024\n"
                         "v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %3, %0, off offset:2048\n v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %1, %0, off offset:3072\n"
                         "v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %2, %0, off\n v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %3, %0, off offset:1024\n"
                         "v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %1, %0, off offset:2048\n v_lshl_add_u64 %0, %4, 0, %5\n global_load_dwordx4 %2, %0, off offset:3072\n"
                         "s_waitcnt vmcnt(0)" : "=&v"(p1), "=&v"(d0), "=&v"(d1), "=&v"(d2) : "v"(p0), "s"(step) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    *clocks = t1 - t0;
    // keep every result alive
    if (v0 == 0xdeadbeefu || p1 == 1 || p2 == 1 || p3 == 1 || d0.x == 1.2345f || d1.x == 1.2345f || d2.x == 1.2345f || d3.x == 1.2345f) a.out[1023] = v0;
}

template <int NOP>
__device__ __forceinline__ void mfma_loop(const Args& a, f4 (&acc)[16], float x, float y) {
    for (unsigned it = 0; it < a.mfma_iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
            if constexpr (NOP >= 0) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop %0" : : "n"(NOP)); __builtin_amdgcn_sched_barrier(0); }
        }
    }
}

__global__ __launch_bounds__(512) void k_probe(const Args a) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned long long clocks = 0;
    if (wave < 4) {
        switch (a.kind) {
            case 0: side_loop<0>(a, &clocks); break;
            case 1: side_loop<1>(a, &clocks); break;
            case 2: side_loop<2>(a, &clocks); break;
            case 3: side_loop<3>(a, &clocks); break;
            case 4: side_loop<4>(a, &clocks); break;
            case 5: side_loop<5>(a, &clocks); break;
            default: side_loop<6>(a, &clocks); break;
        }
    } else if (a.mfma_on) {
        f4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
        const float x = (float)threadIdx.x, y = 1.0f;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        // pace = N > 0: s_nop N - 1 behind every MFMA - the MFMA wave idles instead of presenting its next MFMA (which cannot start before
        // the pipe frees, 32 clocks) to the issue arbiter right away.  Measured (profiles/r02d_valu_mfma_probe.txt): s_nop 15 lets the
        // other wave issue at full speed but costs 38 clocks (MFMA every 70.6): the length to find is the longest that keeps ~32.
        switch (a.pace) {
            case 0: mfma_loop<-1>(a, acc, x, y); break;
#define PACE_CASE(N) case N + 1: mfma_loop<N>(a, acc, x, y); break;
            PACE_CASE(0) PACE_CASE(1) PACE_CASE(2) PACE_CASE(3) PACE_CASE(4) PACE_CASE(5) PACE_CASE(6) PACE_CASE(7)
            PACE_CASE(8) PACE_CASE(9) PACE_CASE(10) PACE_CASE(11) PACE_CASE(12) PACE_CASE(13) PACE_CASE(14) PACE_CASE(15)
#undef PACE_CASE
            default: mfma_loop<-1>(a, acc, x, y); break;
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        clocks = t1 - t0;
        float s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i].x;
        if (s == 1.2345f) a.out[1022] = 1;
    }
    if (blockIdx.x == gridDim.x / 2 && (threadIdx.x & 63) == 0) a.out[wave] = clocks;
}

int main(int argc, char** argv) {
    const unsigned side_iters = argc > 1 ? atoi(argv[1]) : 2000, mfma_iters = argc > 2 ? atoi(argv[2]) : 4000;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    float* buf; unsigned long long* out;
    CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20)); CK(hipMalloc(&out, 1024 * 8)); CK(hipMemset(out, 0, 1024 * 8));
    const size_t lds = 82 * 1024;      // one workgroup per CU
    CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const char* names[7] = {"v_add_u32 chain", "v_lshl_add_u64", "global_load_dwordx4 (VGPR address)", "ds_write_b128", "s_add_u32 chain", "buffer_load_dwordx4 (SGPR offset)",
                            "v_lshl_add_u64 + global_load pairs"};
    const double per_iter[7] = {8, 12 /* 8 adds + the 4 v_mov_b64 the loop-carried values cost */, 8, 8, 8, 8, 16};
    printf("%-38s %22s %22s %18s\n", "side instruction (waves 0-3)", "clocks/instr, MFMA idle", "clocks/instr, MFMA busy", "clocks/MFMA (busy)");
    // argv[3], argv[4]: range of s_nop arguments to try behind every MFMA (e.g. "6 12"); none by default
    const int nop_lo = argc > 3 ? atoi(argv[3]) : -1, nop_hi = argc > 4 ? atoi(argv[4]) : nop_lo;
    for (int nop = nop_lo < 0 ? -1 : nop_lo - 1; nop <= nop_hi; ++nop) {
    const unsigned pace = (nop < nop_lo) ? 0u : (unsigned)nop + 1u;
    if (pace) printf("-- MFMA waves paced: s_nop %d behind every MFMA\n", nop);
    for (unsigned kind = 0; kind < 7; ++kind) {
        double res[2] = {0, 0}, mf = 0;
        for (unsigned on = 0; on < 2; ++on) {
            if (pace && !on) continue;
            Args a = {buf, out, side_iters, mfma_iters, kind, on, pace};
            hipLaunchKernelGGL(k_probe, dim3(p.multiProcessorCount), dim3(512), lds, 0, a);
            CK(hipDeviceSynchronize());
            unsigned long long h[8]; CK(hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost));
            res[on] = (double)h[0] / (side_iters * per_iter[kind]);
            if (on) mf = (double)h[4] / ((double)mfma_iters * 16);
        }
        printf("%-38s %22.1f %22.1f %18.1f\n", names[kind], res[0], res[1], mf);
    }
    }
    printf("(the MFMA waves must outlast the side loop for the 'busy' column to mean anything: raise mfma_iters if clocks/MFMA x mfma_iters x 16 < side clocks)\n");
    return 0;
}
