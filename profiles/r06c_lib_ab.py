"""Interleaved A/B of SEVERAL builds of libtetranerf_hip.so on ONE box (boxes differ by +-12 %, far above the differences
measured): runs profiles/r06c_trace_times.py in a fresh process per build, round-robin, `rounds` times; medians over the rounds
of the call's ms as scheduled and of the serialised per-kernel breakdown; outputs must be bit-identical (checksums).
Usage: python profiles/r06c_lib_ab.py rounds configs name=lib.so[:option=value...] [...] [-- option=value ...]
(options after a library apply to that variant only, options after -- to all)"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rounds, configs = int(sys.argv[1]), sys.argv[2]
rest = sys.argv[3:]
opts = rest[rest.index("--") + 1:] if "--" in rest else []
libs, vopts = [], {}
for a_ in (rest[:rest.index("--")] if "--" in rest else rest):
    n_, spec = a_.split("=", 1)
    parts = spec.split(":")
    libs.append([n_, parts[0]])
    vopts[n_] = parts[1:]
acc = {n: {} for n, _ in libs}
for r in range(rounds):
    for n, lib in libs:
        env = dict(os.environ, TETRANERF_HIP_LIB=str(Path(lib).resolve()))
        p = subprocess.run([sys.executable, str(ROOT / "profiles" / "r06c_trace_times.py"), configs] + opts + vopts[n], env=env, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(n, "FAILED", p.stdout[-500:], p.stderr[-1500:])
            continue
        for k, v in json.loads(line[-1][7:]).items():
            acc[n].setdefault(k, []).append(v)
med = lambda xs: sorted(xs)[len(xs) // 2]
base = libs[0][0]
for k in acc[base]:
    b = med([x["ms"] for x in acc[base][k]])
    print(f"== {k}")
    for n, _ in libs:
        xs = acc[n].get(k, [])
        if not xs:
            continue
        ms = [x["ms"] for x in xs]
        same = {x["checksum"] for x in xs} == {x["checksum"] for x in acc[base][k]}
        bd = {kk: med([x["breakdown"][kk] for x in xs]) for kk in xs[0]["breakdown"]}
        print(f"  {n:10s} {med(ms):7.3f} ms ({', '.join(f'{x:.3f}' for x in sorted(ms))})  {100 * (med(ms) / b - 1):+5.1f} %  identical: {same}  | "
              + " ".join(f"{kk[:9]} {v:.3f}" for kk, v in bd.items()))
