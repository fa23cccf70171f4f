"""Summarise a rocprofv3 rocpd database (kernel trace): per-kernel stats + the timed decode region's busy/gap split.
usage: python tools/prof_summary.py gpurun_out/prof_x/x_results.db [n_timed_steps]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r'void lh::', '', n)
    n = re.sub(r'\(.*$', '', n)
    pros = {"0": "plain", "1": "rmsnorm"}
    epis = {"0": "store", "1": "resid", "2": "qkv_rope", "3": "silu_mul"}
    m = re.match(r'(k_gemv|k_gemv_sa)<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', n)
    if m:
        return f"{m[1]}<KI{m[2]},U{m[3]},TH{m[4]},{pros[m[5]]},{epis[m[6]]}>"
    m = re.match(r'(k_gemv_q8|k_gemv_q8s)<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', n)
    if m:
        return f"{m[1]}<KI{m[2]},U{m[3]},TPR{m[4]},{pros[m[5]]},{epis[m[6]]}>"
    m = re.match(r'k_skinny<(\d+), (\d+), (\d+), (\d+)>', n)
    if m:
        return f"k_skinny<NP{m[1]},{pros[m[2]]},{epis[m[3]]},map{m[4]}>"
    return n[:70]


def main():
    db = sys.argv[1]
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, vgpr_count, lds_size from kernels order by start").fetchall()
    st = collections.defaultdict(list)
    meta = {}
    for n, s, e, v, l in rows:
        st[short(n)].append((e - s) / 1e3)
        meta[short(n)] = (v, l)
    print(f"ALL DISPATCHES ({len(rows)})")
    print(f"{'kernel':52s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'total_ms':>9s} {'vgpr':>5s} {'lds':>7s}")
    for k, v in sorted(st.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k:52s} {len(v):6d} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {sum(v)/1e3:9.2f} {meta[k][0]:5d} {meta[k][1]:7d}")
    idx = [i for i, r in enumerate(rows) if 'argmax' in r[0]]
    if len(idx) >= nsteps:
        # the timed region = nsteps consecutive graph replays (argmax dispatches equally spaced in dispatch count)
        best = None
        for j in range(len(idx) - nsteps + 1):
            d = [idx[j + i + 1] - idx[j + i] for i in range(nsteps - 1)]
            if len(set(d)) == 1:
                best = j
                break
        if best is not None:
            per = idx[best + 1] - idx[best]
            seg = rows[idx[best] - per + 1: idx[best + nsteps - 1] + 1]
            busy = sum(e - s for _, s, e, _, _ in seg) / 1e3
            span = (seg[-1][2] - seg[0][1]) / 1e3
            print(f"\nTIMED REGION: {nsteps} replayed steps, {len(seg)} kernels: span {span:.1f} us = {span/nsteps:.1f} us/step, kernels busy {busy:.1f} us, "
                  f"gaps {span-busy:.1f} us ({(span-busy)/len(seg):.2f} us per boundary, {100*(span-busy)/span:.1f} %)")
            st2 = collections.defaultdict(list)
            for n, s, e, _, _ in seg:
                st2[short(n)].append((e - s) / 1e3)
            print(f"{'kernel':52s} {'calls':>6s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'us/step':>9s}")
            for k, v in sorted(st2.items(), key=lambda kv: -sum(kv[1])):
                print(f"{k:52s} {len(v):6d} {sum(v)/len(v):9.2f} {min(v):9.2f} {max(v):9.2f} {sum(v)/nsteps:9.1f}")


main()
