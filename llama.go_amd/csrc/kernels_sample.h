// csrc/kernels_sample.h — SampleTopPTopK (llama.go:455-707) as ONE workgroup on the device, so the resident decode loop never
// ships the 128 KB logits row to the host.
//
// What the reference does per sampled token (all fp32 unless noted):
//   1. v_i = logits_i * (1/temp); if token i occurs anywhere in the lastNTokens ring (the WHOLE ring of CtxSize ids is scanned,
//      llama.go:509 — lastNTokensSize only feeds debug output; the ring starts as CtxSize zeros, server.go:127-138):
//      v_i = logits_i < 0 ? v_i * penalty : v_i / penalty                                        (llama.go:497-525)
//   2. sort all V pairs by value, descending, keep the first topK (llama.go:548-567).  sort.Slice is not stable, so the order of
//      equal values is unspecified there; here (and in the checker): value descending, then token id ascending.
//   3. p_j = fl32(exp_f64(fl32(v_j - v_0))), sum of the f64 exps in rank order, p_j /= fl32(sum)   (llama.go:581-609)
//   4. topP < 1: fp32 running sum in rank order, cut after the first rank where it reaches topP, rescale by 1/cumsum (llama.go:623-639)
//   5. w_j = ((p_j*p_j)*f_j)*f_j with one uniform f_j per kept rank, result = token of the first maximum  (llama.go:661-673)
// The reference seeds math/rand from the wall clock on every call (llama.go:658), so its draws cannot be reproduced by anyone;
// f_j here comes from a counter-based generator over (seed, sampling call, rank): f = (mix64(mix64(seed ^ (call+1)·C) + j) >> 40)·2⁻²⁴.
//
// Device algorithm (1024 threads, logits read once, strided so loads coalesce; every thread keeps EPT order-preserving keys in
// registers; bitmap of ring members in LDS).  Selection is bitwise bisection on the key: T = largest value with
// count(key >= T) >= K, built from the top bit down — register compares only, no atomics, indifferent to how clustered the
// logits are; ties at the threshold are resolved by a second bisection on the token id (only when there are more ties than
// places).
//   k_sample_small (topK <= 64, the reference's default is 40): no bisection in the common case.  A lower bound P with a
//     GUARANTEED count(key >= P) >= K comes from per-thread maxima (the q-th largest thread maximum of every wave, q = ceil(K/16),
//     minimum over the waves); one pass compacts the elements >= P into LDS (about 2K of them for unclustered logits), they are
//     ranked by counting, ranks < K are the winners in order, and wave 0 runs steps 3-5 in registers (sequential sums in the
//     reference's order over v_readlane broadcasts).  More than 1024 survivors (mass ties) fall back to per-wave bisection
//     with ballot/popcount counts.  (Measured at V = 32000: bisection over all elements costs 32 rounds x 32 elements x 16
//     waves of VALU work on ONE CU = 30 us; this path does one such round.)
//   k_sample (topK <= 1024): block-wide counts (one barrier per bisection round), LDS-resident candidates.
#pragma once
#include "kernels_common.h"

namespace lh {

struct SampleState {
    uint32_t top_k;
    float top_p, temp, repeat_penalty;
    uint64_t seed, draw;           // draw = index of the next sampling call
    uint32_t ring_size, ring_pos;  // ring_pos = ids appended so far (next slot = ring_pos % ring_size)
};

constexpr uint32_t SAMPLE_MAX_K = 1024;

__host__ __device__ __forceinline__ uint64_t smp_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ float sample_uniform(uint64_t seed, uint64_t draw, uint32_t j) {
    const uint64_t key = smp_mix64(seed ^ ((draw + 1) * 0xA24BAED4963EE407ull));
    return (float)(uint32_t)(smp_mix64(key + j) >> 40) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ uint32_t f32_order_key(float v) {  // larger float <=> larger key; key 0 is below every real value
    if (v == 0.f) v = 0.f;                                     // -0 and +0 compare equal in the reference
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_key(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t c) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    return c;
}

// sum of one counter per thread over the 1024 threads; alternating LDS rows make one barrier per call enough
__device__ __forceinline__ uint32_t block_count(uint32_t c, uint32_t* cnt, int& phase, int lane, int wave) {
    c = wave_sum_u32(c);
    if (lane == 0) cnt[phase * 16 + wave] = c;
    __syncthreads();
    uint32_t t = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += cnt[phase * 16 + w];
    phase ^= 1;
    return t;
}

// Penalised, temperature-scaled logits as order keys; element e of thread t is token id t + 1024 e (coalesced).  All EPT loads
// are issued back to back and unconditionally (clamped index): a load inside an `if (i < V)` waits out a full memory latency
// before the next one is issued (32 x ~1.6 us measured).
template <int EPT>
__device__ __forceinline__ void load_keys(uint32_t (&key)[EPT], const float* __restrict__ logits, uint32_t V, const uint32_t* bitmap, float scale, float pen, int tid) {
    float l[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = tid + e * 1024;
        l[e] = logits[i < V ? i : V - 1];
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = tid + e * 1024, ic = i < V ? i : V - 1;
        float v = __fmul_rn(l[e], scale);
        const float pv = l[e] < 0.0f ? __fmul_rn(v, pen) : __fdiv_rn(v, pen);
        if ((bitmap[ic >> 5] >> (ic & 31)) & 1u) v = pv;
        key[e] = i < V ? f32_order_key(v) : 0u;
    }
}

template <int EPT>
__global__ __launch_bounds__(1024) void k_sample(const float* __restrict__ logits, uint32_t V, SampleState* st, uint32_t* __restrict__ ring, StepParams* sp,
                                                 uint32_t* __restrict__ out_tokens, uint32_t* __restrict__ token_out, uint32_t* __restrict__ dbg_ids,
                                                 float* __restrict__ dbg_probs, uint32_t* __restrict__ dbg_keep, int advance) {
    __shared__ uint32_t bitmap[EPT * 32];  // V <= EPT * 1024 bits
    __shared__ uint32_t cnt[32];
    __shared__ uint32_t cand_key[SAMPLE_MAX_K], cand_idx[SAMPLE_MAX_K], s_idx[SAMPLE_MAX_K];
    __shared__ float s_val[SAMPLE_MAX_K], s_prob[SAMPLE_MAX_K];
    __shared__ double s_p64[SAMPLE_MAX_K];
    __shared__ uint32_t ncand, n_keep_s;
    __shared__ float fsum_s, inv_s;
    __shared__ float rv[16];
    __shared__ uint32_t ri[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K = st->top_k;
    const float top_p = st->top_p, pen = st->repeat_penalty;
    const float scale = __fdiv_rn(1.0f, st->temp);  // llama.go:497: float32(1.0 / temp) with temp float32 = one fp32 divide
    const uint32_t ring_size = st->ring_size;
    const uint64_t seed = st->seed, draw = st->draw;
    int phase = 0;

    for (uint32_t w = tid; w < EPT * 32; w += 1024) bitmap[w] = 0;
    if (tid == 0) ncand = 0;
    __syncthreads();
    for (uint32_t r = tid; r < ring_size; r += 1024) {
        const uint32_t t = ring[r];
        if (t < V) atomicOr(&bitmap[t >> 5], 1u << (t & 31));
    }
    __syncthreads();

    uint32_t key[EPT];
    load_keys<EPT>(key, logits, V, bitmap, scale, pen, tid);

    // T = K-th largest key: the largest T with count(key >= T) >= K, built from the top bit down
    uint32_t T = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t c = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) c += key[e] >= cand ? 1u : 0u;
        if (block_count(c, cnt, phase, lane, wave) >= K) T = cand;
    }
    uint32_t cg = 0, ce = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) { cg += key[e] > T ? 1u : 0u; ce += key[e] == T ? 1u : 0u; }
    const uint32_t above = block_count(cg, cnt, phase, lane, wave);
    const uint32_t ties = block_count(ce, cnt, phase, lane, wave);
    const uint32_t need = K - above;  // >= 1 by construction of T
    uint32_t C = 0xFFFFFFFFu;         // ties with token id <= C are taken
    if (ties > need) {                // C = id of the need-th tie in ascending id order
        C = 0;
        for (int b = 16; b >= 0; --b) {
            const uint32_t cand = C | (1u << b);
            uint32_t c = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) c += (key[e] == T && (uint32_t)(tid + e * 1024) < cand) ? 1u : 0u;
            if (block_count(c, cnt, phase, lane, wave) < need) C = cand;
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = tid + e * 1024;
        if (key[e] > T || (key[e] == T && i <= C)) {
            const uint32_t s = atomicAdd(&ncand, 1u);
            if (s < SAMPLE_MAX_K) { cand_key[s] = key[e]; cand_idx[s] = i; }
        }
    }
    __syncthreads();
    // rank by counting: value descending, token id ascending
    if ((uint32_t)tid < K) {
        const uint32_t mk = cand_key[tid], mi = cand_idx[tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < K; ++j) {
            const uint32_t ok = cand_key[j], oi = cand_idx[j];
            rank += (ok > mk || (ok == mk && oi < mi)) ? 1u : 0u;
        }
        s_val[rank] = f32_from_key(mk);
        s_idx[rank] = mi;
    }
    __syncthreads();
    if ((uint32_t)tid < K) s_p64[tid] = exp((double)__fsub_rn(s_val[tid], s_val[0]));
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        for (uint32_t j = 0; j < K; ++j) sum += s_p64[j];
        fsum_s = (float)sum;
    }
    __syncthreads();
    if ((uint32_t)tid < K) s_prob[tid] = __fdiv_rn((float)s_p64[tid], fsum_s);
    __syncthreads();
    if (tid == 0) {
        uint32_t keep = K;
        float inv = 1.0f;
        if (top_p < 1.0f) {
            float cumsum = 0.0f;
            for (uint32_t j = 0; j < K; ++j) {
                cumsum = __fadd_rn(cumsum, s_prob[j]);
                if (cumsum >= top_p) { keep = j + 1; break; }
            }
            inv = __fdiv_rn(1.0f, cumsum);
        }
        n_keep_s = keep;
        inv_s = inv;
    }
    __syncthreads();
    const uint32_t keep = n_keep_s;
    float wv = -1.0f;  // weights are >= 0: idle lanes never win
    uint32_t wi = 0xFFFFFFFFu;
    if ((uint32_t)tid < keep) {
        float p = s_prob[tid];
        if (top_p < 1.0f) p = __fmul_rn(p, inv_s);
        if (dbg_probs) dbg_probs[tid] = p;
        if (dbg_ids) dbg_ids[tid] = s_idx[tid];
        const float f = sample_uniform(seed, draw, (uint32_t)tid);
        wv = __fmul_rn(__fmul_rn(__fmul_rn(p, p), f), f);
        wi = (uint32_t)tid;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(wv, o, 64);
        const uint32_t oi = __shfl_xor(wi, o, 64);
        if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    if (lane == 0) { rv[wave] = wv; ri[wave] = wi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (rv[w] > wv || (rv[w] == wv && ri[w] < wi)) { wv = rv[w]; wi = ri[w]; }
        const uint32_t tok = s_idx[wi];
        if (token_out) *token_out = tok;
        if (dbg_keep) *dbg_keep = keep;
        if (advance) {  // appendToken + the loop bookkeeping of server.go:205-213
            if (ring_size) ring[st->ring_pos % ring_size] = tok;
            st->ring_pos += 1;
            st->draw = draw + 1;
            out_tokens[sp->step] = tok;
            sp->token = tok;
            sp->past += 1;
            sp->step += 1;
        }
    }
}


// ---- topK <= 64 -----------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lanes_below(uint64_t m) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ float bcast_f32(float v, uint32_t l) { return __uint_as_float(bcast_u32(__float_as_uint(v), l)); }
__device__ __forceinline__ double bcast_f64(double v, uint32_t l) {
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    return __longlong_as_double((long long)(((uint64_t)bcast_u32((uint32_t)(u >> 32), l) << 32) | bcast_u32((uint32_t)u, l)));
}

// Wave-level selection of the `want` largest (key desc, id asc) among NE register elements per lane; sel bit e of the result
// marks element e of this lane.  Elements with key 0 are padding.  Everything here is wave-uniform control flow.
template <int NE>
__device__ __forceinline__ uint64_t wave_select(const uint32_t (&key)[NE], const uint32_t (&id)[NE], uint32_t want) {
    uint32_t T = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t cand = T | (1u << b);
        uint32_t c = 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) c += (uint32_t)__popcll(__ballot(key[e] >= cand));
        if (c >= want) T = cand;
    }
    uint32_t above = 0, ties = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        above += (uint32_t)__popcll(__ballot(key[e] > T));
        ties += (uint32_t)__popcll(__ballot(key[e] == T));
    }
    const uint32_t need = want - above;
    uint32_t C = 0xFFFFFFFFu;
    if (ties > need) {
        C = 0;
        for (int b = 16; b >= 0; --b) {
            const uint32_t cand = C | (1u << b);
            uint32_t c = 0;
#pragma unroll
            for (int e = 0; e < NE; ++e) c += (uint32_t)__popcll(__ballot(key[e] == T && id[e] < cand));
            if (c < need) C = cand;
        }
    }
    uint64_t sel = 0;
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if (key[e] > T || (key[e] == T && id[e] <= C)) sel |= 1ull << e;
    return sel;
}

// 64-lane maximum, uniform result: 4 DPP steps inside each row of 16, then the four row results through v_readlane
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));  // row_half_mirror
    v = mx(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));  // row_mirror
    return mx(mx(bcast_u32(v, 0), bcast_u32(v, 16)), mx(bcast_u32(v, 32), bcast_u32(v, 48)));
}

// Element e of thread t is token id 4 (t + 1024 (e / 4)) + e % 4: four consecutive ids per 16-byte load.
__device__ __forceinline__ uint32_t small_id(int tid, int e) { return 4u * ((uint32_t)tid + 1024u * (uint32_t)(e >> 2)) + (uint32_t)(e & 3); }

template <int EPT>
__global__ __launch_bounds__(1024) void k_sample_small(const float* __restrict__ logits, uint32_t V, SampleState* st, uint32_t* __restrict__ ring, StepParams* sp,
                                                       uint32_t* __restrict__ out_tokens, uint32_t* __restrict__ token_out, uint32_t* __restrict__ dbg_ids,
                                                       float* __restrict__ dbg_probs, uint32_t* __restrict__ dbg_keep, int advance) {
    __shared__ uint32_t bitmap[EPT * 32];
    __shared__ __attribute__((aligned(16))) uint64_t surv[1024 + 8];  // a superset of the top-K, unordered: key << 32 | ~id (bigger = better)
    __shared__ uint32_t win_key[64], win_idx[64];        // the K winners in rank order
    __shared__ uint32_t wave_pivot[16];
    __shared__ uint32_t n_surv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K = st->top_k;  // <= 64 (host-checked)
    const float top_p = st->top_p, pen = st->repeat_penalty;
    const float scale = __fdiv_rn(1.0f, st->temp);
    const uint32_t ring_size = st->ring_size;

    // all loads first (unconditional, clamped: see load_keys), the bitmap is built under their latency
    f4 lv[EPT / 4];
    const bool vec = (V & 3u) == 0 && ((uintptr_t)logits & 15u) == 0;
#pragma unroll
    for (int c = 0; c < EPT / 4; ++c) {
        const uint32_t i0 = small_id(tid, 4 * c);
        if (vec) {
            lv[c] = *(const f4*)(logits + (i0 + 3 < V ? i0 : V - 4));
        } else {
            lv[c].x = logits[i0 + 0 < V ? i0 + 0 : V - 1];
            lv[c].y = logits[i0 + 1 < V ? i0 + 1 : V - 1];
            lv[c].z = logits[i0 + 2 < V ? i0 + 2 : V - 1];
            lv[c].w = logits[i0 + 3 < V ? i0 + 3 : V - 1];
        }
    }
    for (uint32_t w = tid; w < EPT * 32; w += 1024) bitmap[w] = 0;
    if (tid == 0) n_surv = 0;
    __syncthreads();
    for (uint32_t r = tid; r < ring_size; r += 1024) {
        const uint32_t t = ring[r];
        if (t < V) atomicOr(&bitmap[t >> 5], 1u << (t & 31));
    }
    __syncthreads();

    uint32_t key[EPT];
    uint32_t tmax = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const uint32_t i = small_id(tid, e);
        const float l = lv[e >> 2][e & 3];
        float v = __fmul_rn(l, scale);
        if (i < V && ((bitmap[i >> 5] >> (i & 31)) & 1u)) v = l < 0.0f ? __fmul_rn(v, pen) : __fdiv_rn(v, pen);  // rare: a divide only where needed
        key[e] = i < V ? f32_order_key(v) : 0u;
        tmax = tmax > key[e] ? tmax : key[e];
    }

    // Pivot: P = min over waves of the wave's q-th largest per-thread maximum, q = ceil(K / 16).  The q largest thread maxima of a
    // wave are q distinct elements >= P, so at least 16 q >= K elements are >= P (a wave with fewer than q non-empty threads
    // yields P = 0: everything survives).  For unclustered logits about 2K elements survive.
    {
        const uint32_t q = (K + 15) / 16;
        uint32_t cur = tmax, a = 0;
        for (uint32_t r = 0; r < q; ++r) {
            const uint32_t m = wave_max_u32(cur);
            a = m;
            const uint64_t b = __ballot(cur == m);
            if (lane == (int)__ffsll((unsigned long long)b) - 1) cur = 0;  // retire ONE instance of the maximum
        }
        if (lane == 0) wave_pivot[wave] = a;
    }
    __syncthreads();
    uint32_t P = wave_pivot[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) P = P < wave_pivot[w] ? P : wave_pivot[w];

    // survivors -> LDS: they are rare (about 2K of V), so every thread that owns any reserves its slots with one LDS atomic
    uint32_t mine = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) mine += (key[e] >= P && key[e] != 0u) ? 1u : 0u;
    if (mine) {
        uint32_t base = atomicAdd(&n_surv, mine);
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (key[e] >= P && key[e] != 0u) {
                if (base < 1024u) surv[base] = ((uint64_t)key[e] << 32) | (uint64_t)(~small_id(tid, e));
                ++base;
            }
    }
    __syncthreads();
    uint32_t n = n_surv;
    if (n > 1024u) {
        // Overflow (mass ties around the K-th place, e.g. constant logits): exact per-wave selection by bisection instead.
        // Every wave leaves its own top-K in its 64-slot segment; unused slots (0) lose against everything.
        __syncthreads();
        surv[tid] = 0ull;
        uint32_t id[EPT];
        uint32_t nvalid = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            id[e] = small_id(tid, e);
            nvalid += (uint32_t)__popcll(__ballot(id[e] < V));
        }
        const uint32_t Kw = nvalid < K ? nvalid : K;
        if (Kw) {
            const uint64_t sel = wave_select<EPT>(key, id, Kw);
            uint32_t b2 = 0;
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const bool s = (sel >> e) & 1ull;
                const uint64_t m = __ballot(s);
                if (s) surv[wave * 64 + b2 + lanes_below(m)] = ((uint64_t)key[e] << 32) | (uint64_t)(~id[e]);
                b2 += (uint32_t)__popcll(m);
            }
        }
        n = 1024u;
        __syncthreads();
    }
    // rank by counting among the survivors (value descending, token id ascending = packed word descending); ranks < K are the
    // winners, already in order.  The list is zero-padded to a multiple of 4 so the loop reads 2 x 16 bytes per step.
    if ((uint32_t)tid >= n && (uint32_t)tid < ((n + 3u) & ~3u)) surv[tid] = 0ull;
    __syncthreads();
    if ((uint32_t)tid < n) {
        const uint64_t me = surv[tid];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < n; j += 4) {
            const ulonglong2 a = *(const ulonglong2*)&surv[j], b = *(const ulonglong2*)&surv[j + 2];
            rank += (a.x > me ? 1u : 0u) + (a.y > me ? 1u : 0u) + (b.x > me ? 1u : 0u) + (b.y > me ? 1u : 0u);
        }
        if (rank < K && me != 0ull) { win_key[rank] = (uint32_t)(me >> 32); win_idx[rank] = ~(uint32_t)me; }
    }
    __syncthreads();
    if (wave != 0) return;

    // steps 3-5 on wave 0, lane j = rank j; sequential sums in the reference's order over constant-lane broadcasts
    const bool live = (uint32_t)lane < K;
    const uint32_t si = live ? win_idx[lane] : 0xFFFFFFFFu;
    const float sv = live ? f32_from_key(win_key[lane]) : 0.0f;
    const float v0 = bcast_f32(sv, 0);
    const double p64 = live ? exp((double)__fsub_rn(sv, v0)) : 0.0;  // idle lanes add +0: no effect on either sum
    double sum = 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) sum += bcast_f64(p64, j);
    float p = __fdiv_rn((float)p64, (float)sum);
    uint32_t keep = K;
    if (top_p < 1.0f) {
        float run = 0.0f, cumsum = 0.0f;
        bool found = false;
#pragma unroll
        for (int j = 0; j < 64; ++j) {
            run = __fadd_rn(run, bcast_f32(p, j));
            if (!found && ((uint32_t)j + 1 == K || run >= top_p)) {  // the cut, or the end of the list without reaching topP
                found = true;
                cumsum = run;
                if (run >= top_p) keep = (uint32_t)j + 1;
            }
        }
        p = __fmul_rn(p, __fdiv_rn(1.0f, cumsum));
    }
    float wv = -1.0f;
    uint32_t wi = 0xFFFFFFFFu;
    if ((uint32_t)lane < keep) {
        if (dbg_probs) dbg_probs[lane] = p;
        if (dbg_ids) dbg_ids[lane] = si;
        const float f = sample_uniform(st->seed, st->draw, (uint32_t)lane);
        wv = __fmul_rn(__fmul_rn(__fmul_rn(p, p), f), f);
        wi = (uint32_t)lane;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(wv, o, 64);
        const uint32_t oi = __shfl_xor(wi, o, 64);
        if (ov > wv || (ov == wv && oi < wi)) { wv = ov; wi = oi; }
    }
    const uint32_t tok = bcast_u32(si, wi);
    if (lane == 0) {
        if (token_out) *token_out = tok;
        if (dbg_keep) *dbg_keep = keep;
        if (advance) {
            if (ring_size) ring[st->ring_pos % ring_size] = tok;
            st->ring_pos += 1;
            st->draw += 1;
            out_tokens[sp->step] = tok;
            sp->token = tok;
            sp->past += 1;
            sp->step += 1;
        }
    }
}

}  // namespace lh
