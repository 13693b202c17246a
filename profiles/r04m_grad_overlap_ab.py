"""Round 4: the two memory-bound tails of the training backward -- the gather adjoint (atomics) and the rgb-head gradient
pass (one read of h4) -- BESIDE the MFMA-bound weight-gradient GEMMs (side streams) against everything on one stream
(TETRANERF_HIP_GRAD_OVERLAP=0).  Whole iterations of both shipped configurations on the C4 batch, a fresh process per
setting, alternating.  Usage: python profiles/r04m_grad_overlap_ab.py [rounds]"""
import os
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
acc = {}
for r in range(rounds):
    for flag in ("00", "10", "01"):
        for name, script in (("tetra-nerf-original", "r03l_train_once.py"), ("tetra-nerf", "r03ae_train_tn.py")):
            env = dict(os.environ, TETRANERF_HIP_GRAD_OVERLAP=flag[0], TETRANERF_HIP_GATHER_OVERLAP=flag[1])
            p = subprocess.run([sys.executable, str(ROOT / "profiles" / script)], env=env, capture_output=True, text=True)
            m = re.findall(r"([0-9.]+) ms per iteration", p.stdout)
            if not m:
                print("FAILED", name, flag, p.stdout[-300:], p.stderr[-800:])
                continue
            acc.setdefault((name, flag), []).append(float(m[-1]))
for name in ("tetra-nerf-original", "tetra-nerf"):
    a = sorted(acc.get((name, "00"), []))
    for flag, what in (("10", "rgb-head pass beside the GEMMs"), ("01", "gather adjoint beside the GEMMs")):
        b = sorted(acc.get((name, flag), []))
        if a and b:
            ma, mb = a[len(a) // 2], b[len(b) // 2]
            print(f"{name:20s} one stream {ma:.3f} ms ({', '.join(f'{x:.2f}' for x in a)})   {what} {mb:.3f} ms ({', '.join(f'{x:.2f}' for x in b)})   {100 * (mb / ma - 1):+.1f} %")
