"""Diagnostic (round 4): per-tensor error of the fused training gradients vs float64, next to float32 autograd's own error,
for the configuration of tests/test_train_gpu.py::test_render_train_fused_equals_autograd_statement that sits at its bound
((24, 24, biased, scaling): w1 1.06e-4 vs the 1e-4 bound on one box).  Usage: python profiles/r04b_grad_diag.py"""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
tn = importlib.import_module("tetra-nerf_amd")
render = importlib.import_module("tetra-nerf_amd.render")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


names = ["field", "w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]
for mesh_seed in (5, 6):
    pts, cells = scenes.random_mesh(4000, mesh_seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = scenes.outside_in_rays(512, 6)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    torch.manual_seed(123)
    target = torch.rand(len(o), 3, device=dev)
    loss_of = lambda rgb, acc: ((rgb - target.to(rgb.dtype)) ** 2).mean() + 0.1 * acc.mean()   # noqa: E731
    for S, S_fine, biased, scaling in ((24, 24, True, True), (24, 24, True, False), (24, 24, False, True), (32, 32, False, False)):
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(dev)
        field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 0.5).requires_grad_(True)
        rd = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased)
        hit = int((tr.trace_rays(to, td, 256)["num_visited_cells"] > 0).sum())
        rand = {"coarse": torch.rand(hit, S + 1, device=dev), "fine": torch.rand(hit, S_fine + 1, device=dev)}
        grads, cap = [], {}
        for fused in (True, False):
            field.grad = None
            mlp.zero_grad()
            out = rd.render_train(to, td, gradient_scaling=scaling, rand=rand, fused=fused, capture=cap)
            loss_of(out["rgb"], out["accumulation"]).backward()
            grads.append([field.grad.clone()] + [p.grad.clone() for p in render.mlp_weights(mlp)])
        dt = torch.float64
        m64 = render.TetraMLP().to(dev).to(dt)
        m64.load_state_dict({k: v.to(dt) for k, v in mlp.state_dict().items()})
        f64 = field.detach().to(dt).requires_grad_(True)
        vi, bc, edges, S2 = cap["vertex_indices"], cap["barycentric_coordinates"].to(dt), cap["edges"].to(dt), cap["samples_per_ray"]
        wts = torch.cat([1 - bc.sum(-1, keepdim=True), bc], -1)
        wts = torch.where(vi < 0, torch.zeros_like(wts), wts)
        feats = (f64.t()[vi.long().clamp_min(0)] * wts[..., None]).sum(-2)
        sg, col = m64(feats, cap["dirs"].to(dt)[:, None, :].expand(-1, S2, -1))
        if scaling:
            spacing = (edges - cap["near"].to(dt)) / (cap["far"].to(dt) - cap["near"].to(dt))
            col, sg, _ = render.GradientScaler.apply(col, sg, (spacing[:, 1:] + spacing[:, :-1])[..., None])
        rgb_r, acc_r, _, _ = render.composite(sg, col, edges[:, :-1, None], edges[:, 1:, None])
        rgb = torch.ones(len(o), 3, dtype=dt, device=dev).index_copy(0, cap["idx"], rgb_r)
        acc = torch.zeros(len(o), 1, dtype=dt, device=dev).index_copy(0, cap["idx"], acc_r)
        loss_of(rgb, acc).backward()
        want = [f64.grad] + [p.grad for p in render.mlp_weights(m64)]
        print(f"mesh seed {mesh_seed} tets {len(cells)} config {(S, S_fine, biased, scaling)} samples {hit * S2}")
        for n, a, b, w in zip(names, grads[0], grads[1], want):
            print(f"   {n:6s} fused {rel(a, w):.3e}   torch32 {rel(b, w):.3e}   |max| {float(w.abs().max()):.3e}")
