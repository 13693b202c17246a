#!/usr/bin/env python3
"""Condense rocprofv3 (rocpd sqlite) outputs into the small text summaries committed under
profiles/.

    python profiles/rocprof_summary.py stats  <results.db>  > profiles/rNN_<what>_kernel_stats.txt
    python profiles/rocprof_summary.py pmc    <results.db>  > profiles/rNN_<what>_pmc.txt

`stats` = what `rocprofv3 --kernel-trace --stats` reports (per kernel: calls, total, average
duration); `pmc` = per-kernel average of each collected counter (one --pmc pass per counter
group, never combined with tracing domains other than --kernel-trace).
"""
import sqlite3
import sys


def short(name: str, n: int = 110) -> str:
    return name if len(name) <= n else name[: n - 3] + "..."


def stats(db):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    print(f"# rocprofv3 --kernel-trace --stats summary of {db}")
    print(f"{'kernel':<112} {'calls':>6} {'total_us':>12} {'avg_us':>12} {'%':>7}")
    for name, calls, total, avg, pct in rows:
        print(f"{short(name):<112} {calls:>6} {total:>12.1f} {avg:>12.2f} {pct:>7.2f}")
    try:
        rows = con.execute(
            "select kernel_name, min(vgpr_count), min(accum_vgpr_count), min(sgpr_count), min(lds_block_size), "
            "min(scratch_size), min(workgroup_size), max(grid_size) from kernels group by kernel_name").fetchall()
        print("\n# resources per kernel (vgpr, agpr, sgpr, lds bytes, scratch bytes, workgroup, max grid)")
        for r in rows:
            if "tn::" in r[0]:
                print(f"{short(r[0], 80):<82} " + " ".join(str(x) for x in r[1:]))
    except sqlite3.Error:
        pass


def pmc(db):
    con = sqlite3.connect(db)
    rows = con.execute(
        "select kernel_name, counter_name, avg(value), min(value), max(value), count(*) "
        "from counters_collection group by kernel_name, counter_name").fetchall()
    print(f"# rocprofv3 --pmc summary of {db} (FETCH_SIZE / WRITE_SIZE are in KiB per dispatch)")
    print(f"{'kernel':<82} {'counter':<28} {'avg':>16} {'min':>16} {'max':>16} {'n':>4}")
    for name, ctr, avg, mn, mx, n in rows:
        if "tn::" in name:
            print(f"{short(name, 80):<82} {ctr:<28} {avg:>16.1f} {mn:>16.1f} {mx:>16.1f} {n:>4}")


def dispatches(db):
    """every tn:: kernel dispatch in launch order: name, grid, duration (us)"""
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)").fetchall()]
    namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    gridcol = "grid_size" if "grid_size" in cols else ("grid" if "grid" in cols else "0")
    if not {"start", "end"} <= set(cols) or namecol is None:
        print("kernels view columns:", cols)
        print("views:", [r[0] for r in con.execute("select name from sqlite_master where type in ('view','table')").fetchall()][:60])
        return
    rows = con.execute(f"select {namecol}, {gridcol}, start, end from kernels order by start").fetchall()
    t0 = rows[0][2] if rows else 0
    for name, grid, st, en in rows:
        if "tn::" in name:
            print(f"{(st - t0) / 1e3:>12.1f} us  +{(en - st) / 1e3:>10.1f} us  grid {grid:>9}  {short(name, 70)}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "dispatches": dispatches}[sys.argv[1]](sys.argv[2])
