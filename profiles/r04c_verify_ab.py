"""Round 4: what the always-on certification cross-check costs.  verify_stride = 0 / 64 (the new default: late form on the
aux stream beside the writer and the fill) / 16, interleaved in ONE process on the C2 and C4 frames and the C5 rays;
bit-identical outputs are asserted.  Usage: python profiles/r04c_verify_ab.py"""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
M = 512
for name, npts, seed, rays in (("C2", 15000, 0, "frame"), ("C4", 45000, 2, "frame"), ("C5", 150000, 3, "random")):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if rays == "frame" else scenes.outside_in_rays(1 << 20, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    ref = None
    times = {0: [], 64: [], 16: []}
    for rnd in range(4):
        for stride in (0, 64, 16):
            tr.set_option("verify_stride", stride)
            out = tr.trace_rays(o, d, M)
            if ref is None:
                ref = {k: v.clone() for k, v in out.items()}
            elif rnd == 0:
                for k in ref:
                    assert torch.equal(ref[k].view(torch.int32), out[k].view(torch.int32)), (name, stride, k)
            why = tr.flag_reasons()
            del out
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                tr.trace_rays(o, d, M)
            e1.record()
            torch.cuda.synchronize()
            times[stride].append(e0.elapsed_time(e1) / 5)
            if rnd == 0:
                print(f"{name} stride {stride}: checked {why.get(15, 0)} mismatches {why.get(14, 0)}")
    base = sorted(times[0])[len(times[0]) // 2]
    for stride in (0, 64, 16):
        t = sorted(times[stride])
        print(f"{name} ({len(cells)} tets, {o.shape[0]} rays) verify_stride {stride:3d}: median {t[len(t) // 2]:.3f} ms  (min {t[0]:.3f}, max {t[-1]:.3f})"
              f"  {100 * (t[len(t) // 2] / base - 1):+.1f} % vs off")
    del tr, ref
    torch.cuda.empty_cache()
