#!/usr/bin/env python3
"""what do the saves of the training forward cost, tensor by tensor?  Times tn_mlp_forward_gather_train alone on
the training batch's shape (4096 rays x 513 samples of the C4 mesh, placement by find_visited_cells) with the library named by
TETRANERF_HIP_LIB -- the variants are builds of THIS tree in which the stores of one saved tensor are compiled out (bit mask:
1 x0, 2 h1, 4 h2, 8 h3, 16 h4, 32 masks; `profiles/r05m_save_ablate.sh` builds them in a scratch copy, the product sources have
no such switch) -- beside the non-saving forward kernel on the same samples."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
cpp = tn.cpp
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(4096, 1); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 0.5)
w = [x.detach() for x in render.mlp_weights(mlp)]
out = tr.trace_rays(o, d, 512)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 513
nv = out["num_visited_cells"]
near = out["hit_distances"][:, 0, 0]
far = out["hit_distances"][torch.arange(len(o), device=dev), (nv.long() - 1).clamp_min(0), 1]
tt = (torch.arange(S, device=dev) + 0.5) / S
dist = (near[:, None] + (far - near)[:, None] * tt[None, :]).contiguous()
m = tr.find_visited_cells(nv, out["visited_cells"], out["barycentric_coordinates"], out["hit_distances"], out["vertex_indices"], dist)
vi, bc = m["vertex_indices"], m["barycentric_coordinates"]

def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

n = vi.numel() // 4
a = timed(lambda: cpp.mlp_forward_gather_train(vi, bc, field, d, w, S))
b = timed(lambda: cpp.mlp_forward_gather(vi, bc, field, d, w, S))
print(f"n = {n} samples ({S} per ray): forward with saves {a:.4f} ms, plain forward {b:.4f} ms, difference {a - b:+.4f} ms")
