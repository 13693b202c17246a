"""Round 4: can the speculative tail fill really OVERLAP the walk?  Rounds 2-3 launched it with 2048 blocks = 8 blocks = 32
waves per CU, i.e. every wave slot of the chip, in front of a walk that itself wants all 32 slots (64 VGPRs): the two
kernels shared the time instead of overlapping (walk alone 0.47 ms + 4.26 GB alone 0.65 ms -> 1.05 ms together).  Here: the
fill with 2 or 1 blocks per CU (8 / 4 waves hold the write ceiling) and the walk limited to 6 / 7 blocks per CU through a
dynamic-LDS reservation, so that both are resident for the walk's whole duration.  Interleaved in one process.
Usage: python profiles/r04f_overlap_sweep.py [configs]"""
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
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["C2", "C4"]
VARIANTS = [("default (2048 blocks, walk unlimited)", {}),
            ("no speculative fill", {"spec_fill": 0}),
            ("fill 512", {"spec_blocks": 512}),
            ("fill 512, walk 7/CU", {"spec_blocks": 512, "walk_lds_kb": 22}),
            ("fill 512, walk 6/CU", {"spec_blocks": 512, "walk_lds_kb": 26}),
            ("fill 512, walk 5/CU", {"spec_blocks": 512, "walk_lds_kb": 32}),
            ("fill 256, walk 7/CU", {"spec_blocks": 256, "walk_lds_kb": 22}),
            ("fill 256, walk 6/CU", {"spec_blocks": 256, "walk_lds_kb": 26}),
            ("fill 512, walk 6/CU, K0 320", {"spec_blocks": 512, "walk_lds_kb": 26, "spec_k0": 320}),
            ("fill 512, walk 6/CU, K0 256", {"spec_blocks": 512, "walk_lds_kb": 26, "spec_k0": 256}),
            ("fill 1024, walk 6/CU", {"spec_blocks": 1024, "walk_lds_kb": 26})]
DEFAULTS = {"spec_fill": 1, "spec_blocks": 0, "walk_lds_kb": 0, "spec_k0": 0}
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2)):
    if name not in which:
        continue
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    ref = tr.trace_rays(o, d, M)
    ref = {k: v.clone() for k, v in ref.items()}
    times = {v[0]: [] for v in VARIANTS}
    for rnd in range(4):
        for label, opts in VARIANTS:
            for k, v in {**DEFAULTS, **opts}.items():
                tr.set_option(k, v)
            out = tr.trace_rays(o, d, M)
            if rnd == 0:
                for k in ref:
                    assert torch.equal(ref[k].view(torch.int32), out[k].view(torch.int32)), (name, label, k)
            del out
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                tr.trace_rays(o, d, M)
            e1.record()
            torch.cuda.synchronize()
            times[label].append(e0.elapsed_time(e1) / 5)
    base = sorted(times[VARIANTS[0][0]])[2]
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    for label, _ in VARIANTS:
        t = sorted(times[label])
        print(f"{name} {label:42s} median {t[len(t) // 2]:.3f} ms (min {t[0]:.3f})  {100 * (t[len(t) // 2] / base - 1):+5.1f} %   {100 * gb / t[len(t) // 2] / 8:.1f} % of 8 TB/s")
    del tr, ref
    torch.cuda.empty_cache()
