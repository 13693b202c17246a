"""Which rule of the walk's order test makes a ray "literal" (diagnostic build of the library: -DTN_WALK_DIAG=1, named by
TETRANERF_HIP_LIB).  Prints the first violated rule per literal ray for the C2 / C4 frames and the C5 rays."""
import ctypes as C
import importlib
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
lib = importlib.import_module("tetra-nerf_amd._lib").load()
dev = torch.device("cuda:0")
# (first run of the round, rules of round 5: gpurun_out/r06b_literal_reasons.txt had [2] = two short gaps at the entry face, [4] = short
#  inverted as the first pair, [8] = inverted pair at the very end -- the three patterns rules A-C of round 6 certify)
NAMES = ["literal rays", "joined a cluster while the entry look-ahead was pending", "eps or more below an earlier member of its cluster",
         "inverted cluster of more than three hits", "inverted cluster within eps of the previous cluster", "inverted first cluster (beyond the pair of rule C)",
         "entry look-ahead unsettled at the end of the chain", "inverted last cluster (not a pair / chain shorter than 4)", "(round 5 rules: any violation)",
         "rays with exactly one violation", "violations in total", "-", "hits of literal rays"]
for name, npts, seed in (("C2", 15000, 0), ("C4", 45000, 2), ("C5", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if name != "C5" else scenes.outside_in_rays(1 << 20, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    arr = (C.c_ulonglong * 16)()
    torch.cuda.synchronize()
    lib.tn_debug_walk_diag(arr, 1)
    out = tr.trace_rays(o, d, 512)
    torch.cuda.synchronize()
    lib.tn_debug_walk_diag(arr, 1)
    hits = int(out["num_visited_cells"].sum())
    print(f"== {name}: {len(cells)} tets, {len(o)} rays, {hits} segments, paths {tr.trace_stats()}, reasons {tr.flag_reasons()}")
    for k, nm in enumerate(NAMES):
        print(f"   [{k:2d}] {nm:48s} {arr[k]}")
    del out, tr
    torch.cuda.empty_cache()
