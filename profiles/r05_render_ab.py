"""One-launch render (tn_render_rays) vs the kernel chain, interleaved in ONE process on one box (round 5).
    python profiles/r05_render_ab.py [reps]
TETRANERF_HIP_RENDER_PROFILE=1 additionally prints the persistent kernel's per-phase time per block (stderr)."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
bench = importlib.import_module("bench")

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4)
field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
R, M = o.shape[0], 512
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 65536      # rays per render() call (nerfstudio's eval default: 4096)
mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"             # round 6: "bf16x3" runs both forms in that arithmetic
print(f"chunks of {chunk} rays, mlp_mode {mode}")


def frame(rd):
    for s in range(0, R, chunk):
        rd.render(o[s:s + chunk], d[s:s + chunk])


for name, (s_c, s_f, biased) in (("coarse-256", (256, 0, False)), ("tetra-nerf-original", (256, 256, False)), ("tetra-nerf", (128, 128, True))):
    rds = {k: render.TetraRenderer(tr, field, mlp, s_c, M, fused=True, num_fine_samples=s_f, biased=biased, fused_pass=fp, mlp_mode=mode)
           for k, fp in (("one_launch", True), ("chain", False))}
    for rd in rds.values():
        frame(rd)
    torch.cuda.synchronize()
    tot = {k: [] for k in rds}
    for _ in range(reps):
        for k, rd in rds.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            frame(rd)
            torch.cuda.synchronize()
            tot[k].append((time.perf_counter() - t0) * 1e3)
    a, b = np.median(tot["one_launch"]), np.median(tot["chain"])
    print(f"{name}: one launch {a:.2f} ms  chain {b:.2f} ms  ratio {a / b:.4f}   all: {[round(x, 2) for x in tot['one_launch']]} vs {[round(x, 2) for x in tot['chain']]}", flush=True)
