#!/usr/bin/env python3
"""Where does the persistent render launch lose its 1.5-3 % to the kernel chain?  (experiment build: TETRANERF_HIP_RENDER_SKIP bit
0/1/2 skips ray phase 1/2/3 of k_render_rays -- the scratch then holds the previous launch's sample placement, so the MLP phases do
the same work.)  Per 65,536-ray chunk of the bench frame, `tetra-nerf-original`:
  one launch, all phases | one launch, MLP phases only | the chain's two k_mlp_forward launches alone on the same samples"""
import importlib, os, sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); render = importlib.import_module("tetra-nerf_amd.render")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
cpp = tn.cpp
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
c = np.array([0.5, 0.5, 0.5], np.float32); eye = c + 2.0 * np.array([0.0, 1.0, 0.0], np.float32)
o, d = scenes.pinhole_rays(800, 800, eye=tuple(eye), lookat=tuple(c), up=(0.0, 0.0, 1.0), fov_y=45.0)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4); field[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
w = [x.detach() for x in render.mlp_weights(mlp)]
chunk = 65536
for name, (S, Sf_, biased) in (("tetra-nerf-original", (256, 256, False)), ("tetra-nerf", (128, 128, True)), ("coarse-256", (256, 0, False))):
    rd = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=Sf_, biased=biased)
    ch = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=Sf_, biased=biased, fused_pass=False)
    s0 = 4 * chunk                     # a chunk from the middle of the frame
    oo, dd = o[s0:s0 + chunk].contiguous(), d[s0:s0 + chunk].contiguous()
    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    os.environ["TETRANERF_HIP_RENDER_SKIP"] = "0"
    full = timed(lambda: rd.render(oo, dd))
    chain = timed(lambda: ch.render(oo, dd))
    os.environ["TETRANERF_HIP_RENDER_SKIP"] = "7"
    only = timed(lambda: rd.render(oo, dd))
    os.environ["TETRANERF_HIP_RENDER_SKIP"] = "0"
    # the trace + compaction both forms share
    tr_only = timed(lambda: cpp.compact_hits(tr.trace_rays(oo, dd, 512)["num_visited_cells"]))
    # the chain's MLP launches alone: sample placement of the hitting rays
    out = tr.trace_rays(oo, dd, 512)
    nv = out["num_visited_cells"]
    idx = torch.nonzero(nv > 0)[:, 0].to(torch.int32)
    r = len(idx)
    lists = [out[k] for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances", "vertex_indices")]
    def place(Sx):
        near = out["hit_distances"][idx.long(), 0, 0]
        far = out["hit_distances"][idx.long(), (nv[idx.long()].long() - 1), 1]
        tt = (torch.arange(Sx, device=dev) + 0.5) / Sx
        dist = (near[:, None] + (far - near)[:, None] * tt[None, :]).contiguous()
        m = tr.find_visited_cells(*lists, dist, ray_index=idx)
        return m["vertex_indices"], m["barycentric_coordinates"]
    dirs = dd[idx.long()].contiguous()
    Sfull = S + Sf_ + 1 if Sf_ else S
    vif, bcf = place(Sfull)
    t_full = timed(lambda: cpp.mlp_forward_gather(vif, bcf, field, dirs, w, Sfull))
    t_den = 0.0
    if Sf_:
        vic, bcc = place(S)
        t_den = timed(lambda: cpp.mlp_forward_gather(vic, bcc, field, None, w, S))
    print(f"{name} ({r} hitting rays of the chunk): one launch {full:.3f} ms | chain {chain:.3f} ms | trace + compaction {tr_only:.3f} ms | "
          f"one launch, MLP phases only {only:.3f} ms -> {only - tr_only:.3f} without the trace | k_mlp_forward alone: density "
          f"{t_den:.3f} + full {t_full:.3f} = {t_den + t_full:.3f} ms (with dir encoding / head term launches)")
