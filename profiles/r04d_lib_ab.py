"""Interleaved A/B of two builds of libtetranerf_hip.so on ONE box (box-to-box variation is +-10 %, far above the
differences being measured): runs profiles/r04d_trace_times.py in a fresh process per build, alternating, `rounds` times.
Usage: python profiles/r04d_lib_ab.py <base.so> <new.so> [rounds] [configs]"""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
base, new = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
configs = sys.argv[4] if len(sys.argv) > 4 else "C2,C4,C5"
acc = {"base": {}, "new": {}}
for r in range(rounds):
    for tag, lib in (("base", base), ("new", new)):
        env = dict(os.environ, TETRANERF_HIP_LIB=str(Path(lib).resolve()))
        p = subprocess.run([sys.executable, str(ROOT / "profiles" / "r04d_trace_times.py"), configs], env=env, capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not line:
            print(tag, "FAILED", p.stdout[-500:], p.stderr[-1500:])
            continue
        for k, v in json.loads(line[-1][7:]).items():
            acc[tag].setdefault(k, []).append(v)
for k in acc["base"]:
    b = sorted(x["ms"] for x in acc["base"][k])
    n = sorted(x["ms"] for x in acc["new"].get(k, []))
    same = {x["checksum"] for x in acc["base"][k]} == {x["checksum"] for x in acc["new"].get(k, [])}
    if n:
        print(f"{k:10s} base {b[len(b) // 2]:.3f} ms ({', '.join(f'{x:.3f}' for x in b)})   new {n[len(n) // 2]:.3f} ms ({', '.join(f'{x:.3f}' for x in n)})"
              f"   {100 * (n[len(n) // 2] / b[len(b) // 2] - 1):+.1f} %   outputs identical: {same}")
