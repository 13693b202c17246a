"""Bisect of the one batch of tests/test_train_gpu.py::test_render_train_fused_equals_autograd_statement on which the fused
training gradients sit 1e-3 from float64 although NO ReLU decision differs (config (24, 24, biased, gradient scaling), mesh seed 5):
which stage makes the difference -- sample placement, composite adjoint, gradient scaler, MLP adjoint?"""
import importlib
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
import test_train_gpu as T  # noqa: E402

device = torch.device("cuda:0")
cpp = tn.cpp
pts, cells = scenes.random_mesh(4000, 5)
tr = tn.TetrahedraTracer(device)
tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
o, d = scenes.outside_in_rays(512, 6)
to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
torch.manual_seed(123)
target = torch.rand(len(o), 3, device=device)
names = ["field", "w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]


def loss_of(rgb, acc):
    return ((rgb - target.to(rgb.dtype)) ** 2).mean() + 0.1 * acc.mean()


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())


for S, S_fine, biased, scaling in ((24, 24, True, True), (24, 24, True, False), (24, 24, False, True)):
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(device)
    field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
    rd = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased)
    hit = int((tr.trace_rays(to, td, 256)["num_visited_cells"] > 0).sum())
    rand = {"coarse": torch.rand(hit, S + 1, device=device), "fine": torch.rand(hit, S_fine + 1, device=device)}
    grads, caps = {}, {}
    for fused in (True, False):
        field.grad = None
        mlp.zero_grad()
        caps[fused] = {}
        out = rd.render_train(to, td, gradient_scaling=scaling, rand=rand, fused=fused, capture=caps[fused])
        loss_of(out["rgb"], out["accumulation"]).backward()
        grads[fused] = [field.grad.clone()] + [p.grad.clone() for p in render.mlp_weights(mlp)]
    ca, cb = caps[True], caps[False]
    print(f"== config {(S, S_fine, biased, scaling)}: placement fused vs unfused: edges max |diff| {float((ca['edges'] - cb['edges']).abs().max()):.3e}, "
          f"vertex ids differ in {int((ca['vertex_indices'] != cb['vertex_indices']).sum())} entries, "
          f"bary max |diff| {float((ca['barycentric_coordinates'] - cb['barycentric_coordinates']).abs().max()):.3e}")
    for which, cap in (("fused run's placement", ca), ("unfused run's placement", cb)):
        dt = torch.float64
        m64 = render.TetraMLP().to(device).to(dt)
        m64.load_state_dict({k: v.to(dt) for k, v in mlp.state_dict().items()})
        f64 = field.detach().to(dt).requires_grad_(True)
        vi, bc, edges, S2 = cap["vertex_indices"], cap["barycentric_coordinates"].to(dt), cap["edges"].to(dt), cap["samples_per_ray"]
        wts = torch.cat([1 - bc.sum(-1, keepdim=True), bc], -1)
        wts = torch.where(vi < 0, torch.zeros_like(wts), wts)
        feats = (f64.t()[vi.long().clamp_min(0)] * wts[..., None]).sum(-2)
        sg, col = m64(feats, cap["dirs"].to(dt)[:, None, :].expand(-1, S2, -1))
        sg.retain_grad(); col.retain_grad()
        sg2, col2 = sg, col
        if scaling:
            spacing = (edges - cap["near"].to(dt)) / (cap["far"].to(dt) - cap["near"].to(dt))
            col2, sg2, _ = render.GradientScaler.apply(col, sg, (spacing[:, 1:] + spacing[:, :-1])[..., None])
        rgb_r, acc_r, _, _ = render.composite(sg2, col2, edges[:, :-1, None], edges[:, 1:, None])
        rgb = torch.ones(len(o), 3, dtype=dt, device=device).index_copy(0, cap["idx"], rgb_r)
        acc = torch.zeros(len(o), 1, dtype=dt, device=device).index_copy(0, cap["idx"], acc_r)
        loss_of(rgb, acc).backward()
        want = [f64.grad] + [p.grad for p in render.mlp_weights(m64)]
        print(f"  float64 on the {which}: fused " + " ".join(f"{n} {rel(a, w):.1e}" for n, a, w in zip(names, grads[True], want)))
        print(f"  {'':>{len(which) + 12}}  torch32 " + " ".join(f"{n} {rel(a, w):.1e}" for n, a, w in zip(names, grads[False], want)))
        if cap is ca:
            # the MLP adjoint kernels alone on float64's upstream gradients (d sigma, d rgb at the MLP's outputs)
            n_s = vi.numel() // 4
            w = [x.detach() for x in render.mlp_weights(mlp)]
            sigma, rgbk, saved = cpp.mlp_forward_gather_train(vi, cap["barycentric_coordinates"], field.detach(), cap["dirs"], w, S2)
            gf, gw = cpp.mlp_backward(saved, vi, cap["barycentric_coordinates"], field.detach(), cap["dirs"], w, sigma, rgbk,
                                      sg.grad.reshape(-1).float().contiguous(), col.grad.reshape(-1, 3).float().contiguous())
            print("  MLP adjoint kernels on float64's d sigma / d rgb: " + " ".join(f"{n} {rel(a, wv):.1e}" for n, a, wv in zip(names, [gf] + list(gw), want)))
            with torch.no_grad():
                nat = T._statement(render, device, mlp, field, vi.reshape(n_s, 4), cap["barycentric_coordinates"].reshape(n_s, 3), cap["dirs"], S2, None, torch.float64)[3]
            print(f"  ReLU decisions differing from float64 per layer: {(T._decode_relu_masks(saved.masks.clone(), n_s) != nat).sum(dim=(1, 2)).tolist()}")
