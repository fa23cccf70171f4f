// csrc/sample.hip — host side of the device sampler (kernels_sample.h): parameter checks, the launch helper the resident
// decode loop uses, and a one-shot entry point for op-level parity tests on arbitrary logits.
#include "plan.h"
#include "kernels_sample.h"

namespace lh {

int sample_check(lh_ctx* ctx, const lh_sample_params* sp, uint32_t V) {
    if (!sp) LH_FAIL(ctx, LH_EINVAL, "sampler: null parameters");
    if (V == 0 || V > 64u * 1024u) LH_FAIL(ctx, LH_ESHAPE, "sampler: vocabulary of %u ids outside the supported 1..65536", V);
    // the reference slices logitsID[:topK] (llama.go:567) and panics when topK > len(logits)
    if (sp->top_k == 0 || sp->top_k > V) LH_FAIL(ctx, LH_EINVAL, "sampler: topK = %u outside 1..%u", sp->top_k, V);
    if (sp->top_k > SAMPLE_MAX_K) LH_FAIL(ctx, LH_EUNSUPPORTED, "sampler: topK = %u above the device limit of %u", sp->top_k, SAMPLE_MAX_K);
    if (!(sp->temp > 0.0f)) LH_FAIL(ctx, LH_EINVAL, "sampler: temp must be > 0 (main.go:379-381 replaces 0 by 0.5)");
    if (!(sp->repeat_penalty > 0.0f)) LH_FAIL(ctx, LH_EINVAL, "sampler: repeatPenalty must be > 0");
    return 0;
}

int sample_launch(lh_ctx* ctx, const float* logits, uint32_t V, SampleState* st, uint32_t* ring, StepParams* sp, uint32_t* out_tokens, uint32_t* token_out,
                  uint32_t* dbg_ids, float* dbg_probs, uint32_t* dbg_keep, int advance, uint32_t topk_hint) {
    const bool small_k = topk_hint && topk_hint <= 64;
    if (small_k && V <= 32u * 1024u)
        LH_LAUNCH(k_sample_small<32>, dim3(1), dim3(1024), 0, ctx->stream, logits, V, st, ring, sp, out_tokens, token_out, dbg_ids, dbg_probs, dbg_keep, advance);
    else if (small_k)
        LH_LAUNCH(k_sample_small<64>, dim3(1), dim3(1024), 0, ctx->stream, logits, V, st, ring, sp, out_tokens, token_out, dbg_ids, dbg_probs, dbg_keep, advance);
    else if (V <= 32u * 1024u)
        LH_LAUNCH(k_sample<32>, dim3(1), dim3(1024), 0, ctx->stream, logits, V, st, ring, sp, out_tokens, token_out, dbg_ids, dbg_probs, dbg_keep, advance);
    else
        LH_LAUNCH(k_sample<64>, dim3(1), dim3(1024), 0, ctx->stream, logits, V, st, ring, sp, out_tokens, token_out, dbg_ids, dbg_probs, dbg_keep, advance);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace lh

using namespace lh;

extern "C" int lh_sample_top_p_top_k(lh_ctx* ctx, const float* logits, uint32_t n_logits, const uint32_t* last_n_tokens, uint32_t n_last, const lh_sample_params* sp,
                                     uint64_t draw, uint32_t* token_out, uint32_t* cand_ids, float* cand_probs, uint32_t* n_keep) {
    if (!ctx) return LH_EINVAL;
    if (!logits || !token_out || (n_last && !last_n_tokens)) LH_FAIL(ctx, LH_EINVAL, "lh_sample_top_p_top_k: null argument");
    int rc;
    if ((rc = sample_check(ctx, sp, n_logits))) return rc;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    // one allocation: logits | ring | state | token, keep | ids | probs
    const size_t o_ring = (size_t)n_logits * 4, o_st = o_ring + (size_t)(n_last ? n_last : 1) * 4, o_tok = o_st + sizeof(SampleState), o_ids = o_tok + 8,
                 o_pr = o_ids + (size_t)sp->top_k * 4, total = o_pr + (size_t)sp->top_k * 4;
    char* dev = nullptr;
    LH_HIP(ctx, hipMalloc((void**)&dev, total));
    SampleState st = {sp->top_k, sp->top_p, sp->temp, sp->repeat_penalty, sp->seed, draw, n_last, 0};
    hipError_t e = hipMemcpyAsync(dev, logits, (size_t)n_logits * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && n_last) e = hipMemcpyAsync(dev + o_ring, last_n_tokens, (size_t)n_last * 4, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dev + o_st, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // st and the caller's arrays are pageable host memory
    if (e == hipSuccess) {
        rc = sample_launch(ctx, (const float*)dev, n_logits, (SampleState*)(dev + o_st), (uint32_t*)(dev + o_ring), nullptr, nullptr, (uint32_t*)(dev + o_tok),
                           (uint32_t*)(dev + o_ids), (float*)(dev + o_pr), (uint32_t*)(dev + o_tok + 4), 0, sp->top_k);
        if (rc) { hipFree(dev); return rc; }
        uint32_t tk[2] = {0, 0};
        e = hipMemcpyAsync(tk, dev + o_tok, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) {
            *token_out = tk[0];
            if (n_keep) *n_keep = tk[1];
            if (cand_ids) e = hipMemcpy(cand_ids, dev + o_ids, (size_t)tk[1] * 4, hipMemcpyDeviceToHost);
            if (e == hipSuccess && cand_probs) e = hipMemcpy(cand_probs, dev + o_pr, (size_t)tk[1] * 4, hipMemcpyDeviceToHost);
        }
    }
    hipFree(dev);
    if (e != hipSuccess) LH_FAIL(ctx, LH_EHIP, "lh_sample_top_p_top_k: %s", hipGetErrorString(e));
    return LH_OK;
}
