// csrc/attn_worklist.h — work decomposition of the single-pass prefill attention (kernels_attn.h, "Balance"): plain C++ shared by the
// host (plan.hip), the kernels (constexpr: callable on the device) and a host-only unit test (tests/test_attn_worklist.py).
#pragma once
#include <stdint.h>
#include <algorithm>
#include <vector>

namespace lh {

constexpr int FA_BQ = 64;                          // queries per block
constexpr int FA_MAXW = 160, FA_MAXPARTS = 16;     // entries of the work list; parts of one block (4 bits of an entry)

// steps (of 64 keys) of query block qb, and the parts it is cut into: the ONE definition the host, the kernel and the combine pass share
constexpr uint32_t fa_steps(uint32_t past, uint32_t n, uint32_t qb) {
    return ((past + ((qb + 1) * FA_BQ < n ? (qb + 1) * FA_BQ : n) + 31) / 32 + 1) / 2;
}
constexpr uint32_t fa_parts(uint32_t steps, uint32_t chunk) { return chunk && steps > chunk ? (steps + chunk - 1) / chunk : 1u; }
// steps of part pt of np: [pt * steps / np, (pt + 1) * steps / np)
constexpr uint32_t fa_part_begin(uint32_t steps, uint32_t np, uint32_t pt) { return pt * steps / np; }

// The list one (n, past) runs with; every layer of an Eval shares it (Plan::fa_work).
struct FaWork {
    uint32_t n = ~0u, past = 0;
    uint32_t chunk = 0;      // a block of more than `chunk` steps is cut into ceil(steps / chunk) parts of near-equal length; 0 = no block is cut
    uint32_t qb_cut = 0;     // first block that is cut (the step count grows with the block index); = number of blocks when none is
    uint32_t pmax = 1;       // parts of the longest block
    uint32_t nwork = 0;      // entries of work[]; 0 = no list: blocks in descending order, uncut
    uint16_t work[FA_MAXW] = {};   // (block << 4 | part), longest first; every head runs the same list
    float cost = 0.f;        // modelled length of the longest workgroup chain, in steps (what the choice minimised)
};

// Chosen by pricing every candidate chunk length with the dealing rule of the kernel itself (items longest first, every head, back and
// forth over the workgroups): the longest chain of steps, + FA_ITEM_STEPS per item for its prologue / merge, + the combine launch when
// anything is cut.
constexpr float FA_ITEM_STEPS = 0.6f, FA_COMBINE_STEPS = 1.0f;
inline void flash_work_list(FaWork& w, uint32_t n, uint32_t past, uint32_t H, uint32_t slots) {
    const uint32_t nqb = (n + FA_BQ - 1) / FA_BQ;
    w = FaWork();
    w.n = n; w.past = past; w.qb_cut = nqb;
    if (n == 0 || H == 0 || slots == 0 || nqb > (uint32_t)FA_MAXW) return;   // no list: blocks in descending order, uncut
    struct Ent { uint32_t steps, code; };
    std::vector<Ent> best, cur;
    std::vector<float> load(slots);
    uint32_t best_chunk = 0;
    const uint32_t longest = fa_steps(past, n, nqb - 1);
    // candidates: uncut (0), then chunk lengths from the longest block's half down to 4 steps (shorter parts are all prologue)
    std::vector<uint32_t> cands = {0};
    for (uint32_t c = (longest + 1) / 2; c >= 4; c -= (c + 7) / 8) cands.push_back(c);
    for (const uint32_t chunk : cands) {
        cur.clear();
        bool fits = true;
        for (uint32_t qb = 0; qb < nqb && fits; ++qb) {
            const uint32_t st = fa_steps(past, n, qb), np = fa_parts(st, chunk);
            if (np > (uint32_t)FA_MAXPARTS) { fits = false; break; }
            for (uint32_t pt = 0; pt < np; ++pt) cur.push_back({fa_part_begin(st, np, pt + 1) - fa_part_begin(st, np, pt), qb << 4 | pt});
        }
        if (!fits || cur.size() > (size_t)FA_MAXW) continue;
        std::stable_sort(cur.begin(), cur.end(), [](const Ent& x, const Ent& y) { return x.steps > y.steps; });
        std::fill(load.begin(), load.end(), 0.f);
        uint64_t item = 0;
        for (const Ent& e : cur)
            for (uint32_t h = 0; h < H; ++h, ++item) {
                const uint64_t round = item / slots, b = item % slots;
                load[(round & 1) ? slots - 1 - b : b] += (float)e.steps + FA_ITEM_STEPS;
            }
        const float cost = *std::max_element(load.begin(), load.end()) + (chunk ? FA_COMBINE_STEPS : 0.f);
        if (best.empty() || cost < w.cost) { best = cur; w.cost = cost; best_chunk = chunk; }
    }
    if (best.empty()) return;
    w.chunk = best_chunk;
    w.nwork = (uint32_t)best.size();
    for (uint32_t i = 0; i < w.nwork; ++i) w.work[i] = (uint16_t)best[i].code;
    for (uint32_t qb = 0; qb < nqb; ++qb) {
        const uint32_t np = fa_parts(fa_steps(past, n, qb), best_chunk);
        if (np > 1 && qb < w.qb_cut) w.qb_cut = qb;
        w.pmax = std::max(w.pmax, np);
    }
}

}  // namespace lh
