// tests/attn_worklist_shim.cpp — host-only view of llama.go_amd/csrc/attn_worklist.h for tests/test_attn_worklist.py (g++, no HIP).
#include "attn_worklist.h"
#include <string.h>

extern "C" {
// returns the number of entries; out[0..5] = chunk, qb_cut, pmax, nwork, cost (float bits), nqb; work[] = the list
int worklist(unsigned n, unsigned past, unsigned H, unsigned slots, unsigned* out, unsigned short* work) {
    lh::FaWork w;
    lh::flash_work_list(w, n, past, H, slots);
    out[0] = w.chunk; out[1] = w.qb_cut; out[2] = w.pmax; out[3] = w.nwork;
    memcpy(&out[4], &w.cost, 4);
    out[5] = (n + lh::FA_BQ - 1) / lh::FA_BQ;
    memcpy(work, w.work, sizeof(w.work));
    return (int)w.nwork;
}
unsigned steps(unsigned past, unsigned n, unsigned qb) { return lh::fa_steps(past, n, qb); }
unsigned parts(unsigned st, unsigned chunk) { return lh::fa_parts(st, chunk); }
unsigned part_begin(unsigned st, unsigned np, unsigned pt) { return lh::fa_part_begin(st, np, pt); }
}
