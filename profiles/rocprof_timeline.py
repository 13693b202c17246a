#!/usr/bin/env python3
"""Per-dispatch timeline out of a rocprofv3 --kernel-trace rocpd database: for every group of dispatches separated by
more than `gap_us` of idle time (= one trace_rays call in profiles/c5_once.py), each kernel's start / end relative to the
group's first start, and the group's span.   python profiles/rocprof_timeline.py <results.db> [gap_us]"""
import sqlite3, sys

db = sys.argv[1]; gap = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 2e6
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name = "kernel_name" if "kernel_name" in cols else "name"
rows = con.execute(f"select {name}, start, end from kernels order by start").fetchall()
groups, cur = [], []
for n, s, e in rows:
    if cur and s - max(x[2] for x in cur) > gap: groups.append(cur); cur = []
    cur.append((n, s, e))
if cur: groups.append(cur)
for gi, g in enumerate(groups):
    if not any("k_trace_walk" in n for n, _, _ in g): continue
    t0 = min(s for _, s, _ in g); t1 = max(e for _, _, e in g)
    print(f"# call {gi}: span {(t1 - t0) / 1e3:.1f} us")
    for n, s, e in g:
        if "tn::" not in n: continue
        short = n.split("(")[0].replace("tn::", "").replace("void ", "")
        print(f"   {short:<28} {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:8.1f} us)")
