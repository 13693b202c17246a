#!/usr/bin/env python3
"""Round-3 debugging aid (run with TETRANERF_HIP_POISON=1): (1) the PDF sampler kernel against the PyTorch statement with
sortedness / collision diagnostics, (2) chunked vs whole render: which rays differ, (3) the training step with poisoned
scratch: does any gradient pick up an unwritten byte."""
import importlib, sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(5000, 41)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(6000, 42)
out = tr.trace_rays(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 256)
nv = out["num_visited_cells"]; idx = torch.nonzero(nv > 0)[:, 0]; ridx = idx.to(torch.int32)
near = out["hit_distances"][idx, 0, 0][:, None]; far = out["hit_distances"][idx, nv[idx].long() - 1, 1][:, None]
print("rays", len(idx))
for S, SF, biased in ((64, 32, False), (64, 64, True), (256, 256, False)):
    edges, nf = tn.cpp.sample_coarse(nv, out["hit_distances"], ridx, S, biased=biased)
    torch.manual_seed(1)
    w = torch.rand(len(idx), S, device=dev) ** 8; w = (w / w.sum(-1, keepdim=True).clamp_min(1.0)).contiguous()
    got = tn.cpp.sample_pdf(edges, w, nf, SF)
    spacing = (edges - near) / (far - near)
    want = render.pdf_sample_bins(spacing, w, SF, near, far)
    unsorted_sp = int((spacing[:, 1:] < spacing[:, :-1]).any(1).sum())
    nan_rows = int(torch.isnan(got).any(1).sum())
    unsorted = int((got[:, 1:] < got[:, :-1]).any(1).sum())
    err = ((got - want).abs() / (far - near))
    print(f"S={S} SF={SF} biased={biased}: rows with unsorted spacing edges {unsorted_sp}, rows with NaN (= unwritten) {nan_rows}, unsorted rows {unsorted}, "
          f"max err {float(torch.nan_to_num(err, nan=9.0).max()):.3e}, median row-max err {float(torch.nan_to_num(err, nan=9.0).max(1).values.median()):.3e}")
    if nan_rows or unsorted:
        r = int(torch.nonzero(torch.isnan(got).any(1) | (got[:, 1:] < got[:, :-1]).any(1))[0])
        print("  first bad row", r, "near/far", float(near[r]), float(far[r]), "got", got[r, :8].tolist(), "want", want[r, :8].tolist())
        sp = spacing[r].cpu().numpy(); print("  spacing diffs min", np.diff(sp).min())

# (2) chunked vs whole render
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev); field = torch.randn(64, len(pts), device=dev) * 0.5
rd = render.TetraRenderer(tr, field, mlp, 64, 256, fused=True, num_fine_samples=32)
po, pd = scenes.pinhole_rays(200, 150, eye=(0.5, 2.4, 0.6), lookat=(0.5, 0.5, 0.5))
to, td = torch.from_numpy(po).to(dev), torch.from_numpy(pd).to(dev)
a = rd.render(to, td); b = rd.render(to, td)
parts = [rd.render(to[s:s + 8192].contiguous(), td[s:s + 8192].contiguous()) for s in range(0, len(po), 8192)]
c = {k: torch.cat([p[k] for p in parts]) for k in ("rgb", "accumulation", "depth")}
for k in ("rgb", "accumulation", "depth"):
    ab = ~torch.isclose(a[k], b[k], rtol=0, atol=0, equal_nan=True); ac = ~torch.isclose(a[k], c[k], rtol=0, atol=0, equal_nan=True)
    print(f"render {k}: whole vs whole differing entries {int(ab.sum())}, whole vs chunked {int(ac.sum())}, max |diff| {float(torch.nan_to_num(a[k] - c[k]).abs().max()):.3e}; hitting rays {int(a['ray_mask'].sum())}")
for ds in (True, False):
    rd2 = render.TetraRenderer(tr, field, mlp, 64, 256, fused=True, num_fine_samples=32, device_samplers=ds)
    a2 = rd2.render(to, td)
    parts = [rd2.render(to[s:s + 8192].contiguous(), td[s:s + 8192].contiguous()) for s in range(0, len(po), 8192)]
    c2 = torch.cat([p["rgb"] for p in parts])
    print(f"device_samplers={ds}: whole vs chunked differing rgb entries {int((~torch.isclose(a2['rgb'], c2, rtol=0, atol=0, equal_nan=True)).sum())}")

# (3) training step, poisoned scratch
to2, td2 = scenes.outside_in_rays(512, 6); to2, td2 = torch.from_numpy(to2).to(dev), torch.from_numpy(td2).to(dev)
pts2, cells2 = scenes.random_mesh(4000, 5)
tr2 = tn.TetrahedraTracer(dev); tr2.load_tetrahedra(torch.from_numpy(pts2).to(dev), torch.from_numpy(cells2).to(dev))
torch.manual_seed(0)
mlp2 = render.TetraMLP().to(dev); field2 = ((torch.rand(64, len(pts2), device=dev) * 2 - 1) * 0.5).requires_grad_(True)
rd3 = render.TetraRenderer(tr2, field2, mlp2, 48, 256, fused=True)
target = torch.rand(512, 3, device=dev)
res = []
for rep in range(3):
    field2.grad = None; mlp2.zero_grad()
    torch.manual_seed(5)
    o_ = rd3.render_train(to2, td2, fused=True)
    (((o_["rgb"] - target) ** 2).mean() + 0.1 * o_["accumulation"].mean()).backward()
    g = [field2.grad.clone()] + [p.grad.clone() for p in render.mlp_weights(mlp2)]
    res.append(g)
    print("train rep", rep, "NaN in grads:", [bool(torch.isnan(x).any()) for x in g][:4], "field grad absmax", float(g[0].abs().max()))
for i in range(len(res[0])):
    d01 = float((res[0][i] - res[1][i]).abs().max() / res[0][i].abs().max().clamp_min(1e-30))
    if d01 > 1e-6: print("  run-to-run relative difference of gradient", i, d01)
