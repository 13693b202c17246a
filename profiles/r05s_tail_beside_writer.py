"""Round 5: the tail fill BESIDE the segment writer once more.  The fill needs the walk's counts only (k_trace_walk writes
num_visited) and the two kernels never share a line.  Round 3 measured this 4.7-8.6 % SLOWER (profiles/r03ag_fill_ab.txt) -- before the
fill had a nontemporal variant and before round 4's finding that a fill must leave wave slots to the kernel beside it.  Here: option
fill_beside_writer = 1 (plain stores) / 2 (nontemporal), with the fill's default grid (2 blocks per CU) and with 1 block per CU;
interleaved in one process, outputs compared bit for bit.
Usage: python profiles/r05s_tail_beside_writer.py [configs]"""
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
VARIANTS = [("default (tail fill after the writer)", {}),
            ("beside the writer, plain stores", {"fill_beside_writer": 1}),
            ("beside the writer, nontemporal", {"fill_beside_writer": 2}),
            ("beside the writer, nontemporal, 1 block per CU", {"fill_beside_writer": 2, "fill_blocks": 256}),
            ("beside the writer, plain, 1 block per CU", {"fill_beside_writer": 1, "fill_blocks": 256}),
            ("beside the writer, nontemporal, 4 blocks per CU", {"fill_beside_writer": 2, "fill_blocks": 1024}),
            ("writer with nontemporal stores", {"writer_nt": 1}),
            ("writer nontemporal + tail beside it, nontemporal, 1 block per CU", {"writer_nt": 1, "fill_beside_writer": 2, "fill_blocks": 256})]
DEFAULTS = {"fill_beside_writer": 0, "fill_blocks": 0, "writer_nt": 0}
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
