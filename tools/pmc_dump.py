"""Per-kernel averages of every counter in a rocprofv3 --pmc run (rocpd database): python tools/pmc_dump.py <results.db> [name-filter]"""
import collections
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = c.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
st = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for n, cn, v, d in rows:
    n = re.sub(r'void lh::', '', n)
    n = re.sub(r'\(.*$', '', n)[:70]
    if flt and flt not in n:
        continue
    st[n][cn].append(v)
    dur[n].append(d)
for n in sorted(st, key=lambda k: -sum(dur[k])):
    print(f"{n}  calls={len(dur[n]) // max(1, len(st[n]))} avg_dur_us={sum(dur[n]) / len(dur[n]) / 1e3:.2f}")
    for cn, v in sorted(st[n].items()):
        print(f"    {cn:32s} {sum(v) / len(v):16.1f}")
