"""Round 4: a THROTTLED share of the post-writer tail fill beside the segment writer.  Round 3 measured the full-rate fill
beside the writer 5-9 % slower (the latency-bound writer suffers under a saturated write queue).  Here both fills pull
16-row chunks from one atomic dispenser, so a small grid (64 ... 512 blocks) can start beside the writer and the full grid
finishes whatever is left.  Interleaved in one process, bit-identical outputs asserted.
Usage: python profiles/r04l_fill_overlap.py"""
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
VARIANTS = [0, 32, 64, 128, 256, 512]
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    ref = {k: v.clone() for k, v in tr.trace_rays(o, d, M).items()}
    times = {v: [] for v in VARIANTS}
    for rnd in range(4):
        for v in VARIANTS:
            tr.set_option("fill_overlap_blocks", v)
            out = tr.trace_rays(o, d, M)
            if rnd == 0:
                for k in ref:
                    assert torch.equal(ref[k].view(torch.int32), out[k].view(torch.int32)), (name, v, k)
            del out
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                tr.trace_rays(o, d, M)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 5)
    base = sorted(times[0])[2]
    for v in VARIANTS:
        t = sorted(times[v])
        print(f"{name} fill_overlap_blocks {v:4d}: median {t[len(t) // 2]:.3f} ms (min {t[0]:.3f})  {100 * (t[len(t) // 2] / base - 1):+5.1f} %")
    del tr, ref
    torch.cuda.empty_cache()
