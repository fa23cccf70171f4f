"""Per-kernel HBM traffic from a rocprofv3 --pmc FETCH_SIZE run (rocpd database).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly HALF the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B) -> doubled here before comparing with the algorithmic byte count.
usage: python tools/pmc_summary.py gpurun_out/pmc_x/pmc_results.db"""
import collections
import re
import sqlite3
import sys

ALG = {  # algorithmic bytes per launch for the 7B decode kernels
    "qkv_rope": 3 * 4096 * 4096 * 4, "silu_mul": 2 * 11008 * 4096 * 4, "KI3": 4096 * 11008 * 4, "plain,resid": 4096 * 4096 * 4, "rmsnorm,store": 32000 * 4096 * 4,
}


def short(n):
    n = re.sub(r'void lh::', '', n)
    n = re.sub(r'\(.*$', '', n)
    m = re.match(r'k_gemv<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', n)
    if m:
        pro = {"0": "plain", "1": "rmsnorm"}[m[4]]
        epi = {"0": "store", "1": "resid", "2": "qkv_rope", "3": "silu_mul"}[m[5]]
        return f"k_gemv<KI{m[1]},U{m[2]},TH{m[3]},{pro},{epi}>"
    return n[:60]


c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, value, duration from counters_collection where counter_name='FETCH_SIZE'").fetchall()
st = collections.defaultdict(list)
for n, v, d in rows:
    st[short(n)].append((v, d))
print(f"{'kernel':48s} {'calls':>6s} {'FETCH_SIZE_KB(raw)':>19s} {'HBM_MB(x2 corr.)':>17s} {'algorithmic_MB':>15s} {'traffic/alg':>11s}")
for k, v in sorted(st.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    if not k.startswith("k_gemv") and "attention" not in k:
        continue
    avg = sum(x[0] for x in v) / len(v)
    hbm = avg * 1024 * 2 / 1e6
    alg = None
    for key, b in ALG.items():
        if key in k and (key != "plain,resid" or "KI1" in k):
            alg = b / 1e6
    print(f"{k:48s} {len(v):6d} {avg:19.1f} {hbm:17.1f} {alg if alg else float('nan'):15.1f} {hbm/alg if alg else float('nan'):11.3f}")
