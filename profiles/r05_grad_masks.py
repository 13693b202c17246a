"""Round 5 diagnostic of the training-gradient gap: the fused adjoint kernels against float64 autograd
WITH THE ReLU MASKS THE FUSED FORWARD SAVED.

Round 4 found the fused gradients of the tensors upstream of a hidden layer up to 1e-3 (relative to the largest entry) away from
float64 autograd while float32 autograd sat at 1e-6, and attributed it to ReLU units whose pre-activation rounds to the other
side of 0 than in float64 -- asserted, never shown.  Here the float64 (and float32) statement is evaluated with the SAME masks
the fused forward kernel saved (tn_mlp_forward_gather_train: [4, n, 2] u64, bit j of word (layer, sample, half h) = slot j =
feature 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h), on the same sample placement and the same upstream gradients, so that
what remains is the arithmetic of k_mlp_backward / k_dw_gemm / k_interp_bwd alone; the number of mask bits that differ from
float64's own signs is printed beside it.
    python profiles/r05_grad_masks.py"""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
tn = importlib.import_module("tetra-nerf_amd")
render = importlib.import_module("tetra-nerf_amd.render")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
cpp = tn.cpp
dev = torch.device("cuda:0")


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def decode_masks(masks, n):
    """[4, n, 2] int64 -> bool [4, n, 128] in nn.Linear feature order"""
    j = torch.arange(64, device=masks.device)
    bits = ((masks[..., None] >> j) & 1).bool()                      # [4, n, 2, 64]
    out = torch.empty(4, n, 128, dtype=torch.bool, device=masks.device)
    for h in range(2):
        feat = 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h
        out[:, :, feat] = bits[:, :, h, :]
    return out


def masked_statement(mlp, field, vi, bc, dirs, S, masks, dtype):
    """gather + MLP + heads in `dtype`, ReLU replaced by the given masks (None: the natural ones).  Returns sigma [n], rgb [n,3],
    the leaves (field, 12 weights) and the natural masks it would have used."""
    m = render.TetraMLP().to(dev).to(dtype)
    m.load_state_dict({k: v.to(dtype) for k, v in mlp.state_dict().items()})
    f = field.detach().to(dtype).requires_grad_(True)
    b = bc.to(dtype)
    wts = torch.cat([1 - b.sum(-1, keepdim=True), b], -1)
    wts = torch.where(vi < 0, torch.zeros_like(wts), wts)
    x = (f.t()[vi.long().clamp_min(0)] * wts[..., None]).sum(-2)        # [n, 64]
    natural = []
    for l, lin in enumerate(m.base):
        pre = lin(x)
        natural.append(pre > 0)
        x = pre * (masks[l] if masks is not None else natural[-1]).to(dtype)
    sigma = torch.nn.functional.softplus(m.density(x))[..., 0]
    enc = render.direction_encoding(dirs.to(dtype))[:, None, :].expand(-1, S, -1).reshape(-1, 27)
    pre = m.head(torch.cat([enc, x], -1))
    natural.append(pre > 0)
    h = pre * (masks[3] if masks is not None else natural[-1]).to(dtype)
    rgb = torch.sigmoid(m.rgb(h))
    return sigma, rgb, [f] + render.mlp_weights(m), torch.stack(natural)


names = ["field", "w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]
worst = {}
for mesh_seed in (5, 6):
    pts, cells = scenes.random_mesh(4000, mesh_seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = scenes.outside_in_rays(512, 6)
    to, td = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    torch.manual_seed(123)
    target = torch.rand(len(o), 3, device=dev)
    for S, S_fine, biased, scaling in ((24, 24, True, True), (24, 24, True, False), (24, 24, False, True), (32, 32, False, False)):
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(dev)
        field = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 0.5).requires_grad_(True)
        rd = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased)
        hit = int((tr.trace_rays(to, td, 256)["num_visited_cells"] > 0).sum())
        rand = {"coarse": torch.rand(hit, S + 1, device=dev), "fine": torch.rand(hit, S_fine + 1, device=dev)}
        cap = {}
        with torch.no_grad():
            rd.render_train(to, td, gradient_scaling=scaling, rand=rand, fused=True, capture=cap)
        vi, bc, edges, S2, dirs = cap["vertex_indices"], cap["barycentric_coordinates"], cap["edges"], cap["samples_per_ray"], cap["dirs"]
        n = vi.numel() // 4
        w = render.mlp_weights(mlp)
        # fused forward (saves the masks) -> upstream gradients from the float64 composite + loss at ITS outputs
        sigma, rgb, saved = cpp.mlp_forward_gather_train(vi, bc, field.detach(), dirs, [x.detach() for x in w], S2)
        masks = decode_masks(saved.masks.clone(), n)
        dt = torch.float64
        sg = sigma.detach().to(dt).view(-1, S2, 1).requires_grad_(True)
        cl = rgb.detach().to(dt).view(-1, S2, 3).requires_grad_(True)
        e64 = edges.to(dt)
        sg2, cl2 = sg, cl
        if scaling:
            spacing = (e64 - cap["near"].to(dt)) / (cap["far"].to(dt) - cap["near"].to(dt))
            cl2, sg2, _ = render.GradientScaler.apply(cl, sg, (spacing[:, 1:] + spacing[:, :-1])[..., None])
        rgb_r, acc_r, _, _ = render.composite(sg2, cl2, e64[:, :-1, None], e64[:, 1:, None])
        full_rgb = torch.ones(len(o), 3, dtype=dt, device=dev).index_copy(0, cap["idx"], rgb_r)
        full_acc = torch.zeros(len(o), 1, dtype=dt, device=dev).index_copy(0, cap["idx"], acc_r)
        (((full_rgb - target.to(dt)) ** 2).mean() + 0.1 * full_acc.mean()).backward()
        d_sigma, d_rgb = sg.grad.reshape(-1), cl.grad.reshape(-1, 3)
        # fused adjoint
        gf, gw = cpp.mlp_backward(saved, vi, bc, field.detach(), dirs, [x.detach() for x in w], sigma, rgb, d_sigma.float().contiguous(),
                                  d_rgb.float().contiguous())
        fused = [gf] + list(gw)
        res = {}
        for label, dtype, mk in (("f64 fused masks", torch.float64, masks), ("f32 fused masks", torch.float32, masks),
                                 ("f64 own masks", torch.float64, None), ("f32 own masks", torch.float32, None)):
            s_, c_, leaves, natural = masked_statement(mlp, field, vi.reshape(n, 4), bc.reshape(n, 3), dirs, S2, mk, dtype)
            ((s_ * d_sigma.to(dtype)).sum() + (c_ * d_rgb.to(dtype)).sum()).backward()
            res[label] = ([x.grad for x in leaves], natural)
        flipped = (masks != res["f64 own masks"][1]).sum(dim=(1, 2)).tolist()
        flipped32 = (res["f32 own masks"][1] != res["f64 own masks"][1]).sum(dim=(1, 2)).tolist()
        print(f"mesh seed {mesh_seed} config {(S, S_fine, biased, scaling)} samples {n}: mask bits that differ from float64's signs, per "
              f"layer (of {n * 128} each): fused {flipped}, float32 autograd {flipped32}")
        want_m, want_o = res["f64 fused masks"][0], res["f64 own masks"][0]
        for k, name in enumerate(names):
            a = rel(fused[k], want_m[k]); b = rel(res["f32 fused masks"][0][k], want_m[k])
            c = rel(fused[k], want_o[k]); e = rel(res["f32 own masks"][0][k], want_o[k])
            worst[name] = max(worst.get(name, 0.0), a)
            print(f"   {name:6s} vs float64 WITH THE FUSED MASKS: fused {a:.3e}  torch32 {b:.3e}   |  vs float64 with its own masks: fused {c:.3e}  "
                  f"torch32 {e:.3e}   |max| {float(want_m[k].abs().max()):.3e}")
print("worst fused error under the masked comparison per tensor:", {k: f"{v:.2e}" for k, v in worst.items()})
