"""Per-kernel HBM traffic from a rocprofv3 --pmc FETCH_SIZE run (rocpd database).
gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports exactly HALF the bytes of a wide coalesced streaming read
(128-B requests tallied at 64 B) -> doubled here before comparing with the algorithmic byte count.
usage: python tools/pmc_summary.py gpurun_out/pmc_x/pmc_results.db [traffic.json: bytes per launch keyed by bench.py's kernel labels]"""
import collections
import re
import sqlite3
import sys

import json

# algorithmic bytes per launch for the 7B decode kernels and bench.py's label of each
ALG = {"qkv_rope": (3 * 4096 * 4096 * 4, "gemv_qkv_rope"), "silu_mul": (2 * 11008 * 4096 * 4, "gemv_w1w3_silu"), "w2": (4096 * 11008 * 4, "gemv_w2_resid"),
       "wo": (4096 * 4096 * 4, "gemv_wo_resid"), "rmsnorm,store": (32000 * 4096 * 4, "gemv_lmhead")}


def alg_key(k):
    if "plain,resid" in k:
        return "w2" if ("KI3" in k or "KI6" in k) else "wo"   # K = 11008: 3 float4 per thread at 1024 threads, 6 at 512
    for key in ("qkv_rope", "silu_mul", "rmsnorm,store"):
        if key in k:
            return key
    return None


def short(n):
    n = re.sub(r'void lh::', '', n)
    n = re.sub(r'\(.*$', '', n)
    m = re.match(r'(k_gemv|k_gemv_sa)<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>', n)
    if m:
        pro = {"0": "plain", "1": "rmsnorm"}[m[5]]
        epi = {"0": "store", "1": "resid", "2": "qkv_rope", "3": "silu_mul"}[m[6]]
        return f"{m[1]}<KI{m[2]},U{m[3]},TH{m[4]},{pro},{epi}>"
    return n[:60]


c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, value, duration from counters_collection where counter_name='FETCH_SIZE'").fetchall()
st = collections.defaultdict(list)
for n, v, d in rows:
    st[short(n)].append((v, d))
traffic = {}
print(f"{'kernel':48s} {'calls':>6s} {'FETCH_SIZE_KB(raw)':>19s} {'HBM_MB(x2 corr.)':>17s} {'algorithmic_MB':>15s} {'traffic/alg':>11s}")
for k, v in sorted(st.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    if not k.startswith("k_gemv") and "attention" not in k:
        continue
    avg = sum(x[0] for x in v) / len(v)
    hbm = avg * 1024 * 2 / 1e6
    alg = None
    ak = alg_key(k) if k.startswith("k_gemv") and "cols" not in k else None
    if ak:
        alg = ALG[ak][0] / 1e6
        traffic[ALG[ak][1]] = round(hbm * 1e6, -5)
    print(f"{k:48s} {len(v):6d} {avg:19.1f} {hbm:17.1f} {alg if alg else float('nan'):15.1f} {hbm/alg if alg else float('nan'):11.3f}")

if len(sys.argv) > 2:
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from source_hash import kernel_source_hash
    json.dump({"kernel_source_sha256": kernel_source_hash(), "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python bench.py --no-cpu-baseline --no-prefill, x2 gfx950 correction per MI355X_MICROARCH.md "
                         "(a separate profiling run of the same kernels, not a measurement of this bench run)", "bytes_per_launch": traffic}, open(sys.argv[2], "w"), indent=1)
