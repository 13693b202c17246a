"""Cost of the certification cross-check (round 5): blind sample stride x risk-class band, interleaved in one process.
    python profiles/r05_risk_sweep.py"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
bench = importlib.import_module("bench")
dev = torch.device("cuda:0")
M = 512
variants = [("stride 256, no risk classes (round 4)", dict(verify_stride=256, verify_risk=0)),
            ("stride 256 + band 2", dict(verify_stride=256, verify_risk=1, risk_band=2)),
            ("stride 512 + band 2", dict(verify_stride=512, verify_risk=1, risk_band=2)),
            ("stride 1024 + band 2", dict(verify_stride=1024, verify_risk=1, risk_band=2)),
            ("stride 1024 + band 4", dict(verify_stride=1024, verify_risk=1, risk_band=4)),
            ("stride 1024 + band 8", dict(verify_stride=1024, verify_risk=1, risk_band=8)),
            ("off", dict(verify_stride=0, verify_risk=0))]
for cfg, npts, seed, rays in (("C2 frame", 15000, 0, "frame"), ("C4 frame", 45000, 2, "frame"), ("C5 2^20 rays", 150000, 3, "random")):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if rays == "frame" else scenes.outside_in_rays(1 << 20, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    times = {k: [] for k, _ in variants}
    info = {}
    for rep in range(4):
        for name, opts in variants:
            for k, v in opts.items():
                tr.set_option(k, v)
            out = tr.trace_rays(o, d, M); del out
            if rep == 0:
                info[name] = tr.cross_check()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                out = tr.trace_rays(o, d, M); del out
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / 3)
    base = np.median(times["off"])
    print(cfg)
    for name, _ in variants:
        t = np.median(times[name])
        x = info[name]
        print(f"   {name:40s} {t:7.3f} ms  {100 * (t / base - 1):+5.1f} % vs off   checked {x['checked']} + {x['risk']['checked']} "
              f"(classes: hull {x['risk']['hull_near_miss_rays']}, thin {x['risk']['thin_neighbourhood_rays']}), mismatches {x['mismatches']} + {x['risk']['mismatches']}", flush=True)
    del tr
    torch.cuda.empty_cache()
