"""Training adjoints on the GPU: the fused MLP node (tn_mlp_forward_gather / tn_mlp_backward /
tn_mlp_param_grads / tn_interpolate_values_backward), the composite node (tn_composite / tn_composite_backward) and
GradientScaler (model.py:195-205), against PyTorch autograd of the plain statement in render.py -- in float64 as the
ground truth, with the float32 autograd result as the yardstick for what fp32 can deliver."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def test_composite_backward_matches_autograd(tn, device):
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    torch.manual_seed(3)
    R, S = 300, 200
    sigma = (torch.rand(R, S, device=device) * 4).requires_grad_(True)
    rgb = torch.rand(R, S, 3, device=device).requires_grad_(True)
    edges = (1.0 + torch.cumsum(torch.rand(R, S + 1, device=device) * 0.02, -1)).contiguous()
    g_rgb, g_acc = torch.randn(R, 3, device=device), torch.randn(R, 1, device=device)
    out_rgb, acc, _, _ = render.composite(sigma.double()[..., None], rgb.double(), edges.double()[:, :-1, None], edges.double()[:, 1:, None])
    (out_rgb * g_rgb.double()).sum().add((acc * g_acc.double()).sum()).backward()
    want_s, want_c = sigma.grad.clone(), rgb.grad.clone()
    got_s, got_c = tn.cpp.composite_backward(sigma.detach(), rgb.detach(), edges, g_rgb, g_acc.reshape(-1))
    assert _rel(got_s, want_s) < 1e-5 and _rel(got_c, want_c) < 1e-5, (_rel(got_s, want_s), _rel(got_c, want_c))
    # and as an autograd node
    s2, c2 = sigma.detach().clone().requires_grad_(True), rgb.detach().clone().requires_grad_(True)
    o, a, d = render._FusedCompositeFunction.apply(s2, c2, edges, 1.0)
    np.testing.assert_allclose(o.detach().cpu().numpy(), out_rgb.detach().float().cpu().numpy(), rtol=0, atol=1e-5)
    ((o * g_rgb).sum() + (a.reshape(-1, 1) * g_acc).sum()).backward()
    assert _rel(s2.grad, want_s) < 1e-5 and _rel(c2.grad, want_c) < 1e-5


@pytest.mark.parametrize("R,S,V", [(3, 7, 200), (300, 64, 5000), (300, 97, 5000), (4096, 513, 45000)])
def test_mlp_backward_matches_autograd(tn, device, R, S, V):
    """Gradients of the fused gather + MLP + heads node w.r.t. the field and all 12 weight tensors; the first case is
    smaller than one 32-sample wave tile (21 samples), the last one the C4 training batch itself (4096 rays x 513 fine
    samples = 2.1 M samples, V = 45k): 512 slices of 4128 samples per weight-gradient GEMM, summed in a fixed order."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    torch.manual_seed(1)
    n = R * S
    mlp = render.TetraMLP().to(device)
    for p in mlp.parameters():      # larger weights than the default init: every ReLU / softplus / sigmoid branch is live
        p.data.mul_(1.5)
    field = (torch.randn(64, V, device=device) * 0.7).requires_grad_(True)
    vi = torch.randint(0, V, (n, 4), dtype=torch.int32, device=device)
    vi[::17, 2] = -1                # EMPTY vertices are skipped by the gather
    bc = (torch.rand(n, 3, device=device) / 3).contiguous()
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=device), dim=-1)
    g_sigma, g_rgb = torch.randn(n, device=device) / R, torch.randn(n, 3, device=device) / R

    def reference(dtype):
        m = render.TetraMLP().to(device).to(dtype)
        m.load_state_dict({k: v.to(dtype) for k, v in mlp.state_dict().items()})
        f = field.detach().to(dtype).requires_grad_(True)
        ft = f.t()                                                   # [V, 64]
        idx = vi.long().clamp_min(0)
        wts = torch.cat([1 - bc.to(dtype).sum(-1, keepdim=True), bc.to(dtype)], -1)   # weights of (v0, v1, v2, v3)
        wts = torch.where(vi < 0, torch.zeros_like(wts), wts)
        feats = (ft[idx] * wts[..., None]).sum(1)                    # [n, 64]
        sg, col = m(feats, dirs.to(dtype)[:, None, :].expand(-1, S, -1).reshape(n, 3))
        ((sg[:, 0] * g_sigma.to(dtype)).sum() + (col * g_rgb.to(dtype)).sum()).backward()
        return sg[:, 0].detach(), col.detach(), f.grad, [p.grad for p in render.mlp_weights(m)]

    s64, c64, gf64, gw64 = reference(torch.float64)
    _, _, gf32, gw32 = reference(torch.float32)
    w = render.mlp_weights(mlp)
    sigma, col = render._FusedMlpFunction.apply(vi, bc, field, dirs, S, None, *w)
    np.testing.assert_allclose(sigma.detach().cpu().numpy(), s64.float().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(col.detach().cpu().numpy(), c64.float().cpu().numpy(), rtol=0, atol=1e-5)
    ((sigma * g_sigma).sum() + (col * g_rgb).sum()).backward()
    names = ["w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]
    errs = {"field": (_rel(field.grad, gf64), _rel(gf32, gf64))}
    for name, p, g64, g32 in zip(names, w, gw64, gw32):
        errs[name] = (_rel(p.grad, g64), _rel(g32, g64))
    for name, (ours, torch32) in errs.items():
        # within 1e-5 of the float64 gradient (relative to its largest entry), or at least as good as twice what
        # float32 autograd itself achieves on sums over ~2e4 samples
        assert ours < max(1e-5, 2.0 * torch32), (name, ours, torch32, errs)


def test_render_train_fused_equals_autograd_statement(tn, device, scenes):
    """Whole training forward/backward of both shipped configurations on a real mesh: fused nodes and the plain float32
    PyTorch statement, with the same random draws (stratified coarse samples, stratified PDF samples, gradient
    scaling), each measured against the FLOAT64 statement on the same sample placement."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(4000, 5)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(512, 6)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    torch.manual_seed(123)     # (the target used to depend on the RNG state the preceding tests left behind)
    target = torch.rand(len(o), 3, device=device)

    def loss_of(rgb, acc):
        return ((rgb - target.to(rgb.dtype)) ** 2).mean() + 0.1 * acc.mean()

    for S, S_fine, biased, scaling in ((48, 0, False, False), (32, 32, False, False), (24, 24, True, True)):
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(device)
        field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
        rd = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased)
        hit = int((tr.trace_rays(to, td, 256)["num_visited_cells"] > 0).sum())
        rand = {"coarse": torch.rand(hit, S + 1, device=device), "fine": torch.rand(hit, S_fine + 1, device=device)}
        # Each path is compared with float64 autograd of the plain statement ON ITS OWN SAMPLE PLACEMENT (round 6).  Until round 5
        # one capture dictionary served both runs, so the fused gradients were compared with float64 on the UNFUSED run's
        # placement: the two placements differ by an ulp of the bin edges (5e-5 in the barycentrics of short segments), which
        # moves samples across ReLU boundaries -- that, not the kernels, was most of the 1e-4 .. 2e-3 this test used to allow
        # (profiles/r06h_grad_bisect.txt: on its own placement the fused path is within 5e-7 of float64 on every tensor
        # whenever no ReLU decision differs).
        def float64_statement(cap):
            dt = torch.float64
            m64 = render.TetraMLP().to(device).to(dt)
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
            rgb = torch.ones(len(o), 3, dtype=dt, device=device).index_copy(0, cap["idx"], rgb_r)
            acc = torch.zeros(len(o), 1, dtype=dt, device=device).index_copy(0, cap["idx"], acc_r)
            loss_of(rgb, acc).backward()
            assert float(f64.grad.abs().max()) > 0
            return rgb.detach().float(), [f64.grad] + [p.grad for p in render.mlp_weights(m64)]

        def flipped_decisions(cap, fused):
            """ReLU decisions of the fp32 forward on this placement (fused: the masks tn_mlp_forward_gather_train saves; unfused:
            the float32 statement's own signs) that differ from float64's, per layer"""
            vi, bc, S2 = cap["vertex_indices"], cap["barycentric_coordinates"], cap["samples_per_ray"]
            n_s = vi.numel() // 4
            with torch.no_grad():
                natural64 = _statement(render, device, mlp, field, vi.reshape(n_s, 4), bc.reshape(n_s, 3), cap["dirs"], S2, None, torch.float64)[3]
                if fused:
                    _, _, saved = tn.cpp.mlp_forward_gather_train(vi, bc, field.detach(), cap["dirs"], [x.detach() for x in render.mlp_weights(mlp)], S2)
                    ours = _decode_relu_masks(saved.masks.clone(), n_s)
                else:
                    ours = _statement(render, device, mlp, field, vi.reshape(n_s, 4), bc.reshape(n_s, 3), cap["dirs"], S2, None, torch.float32)[3]
            return n_s, (ours != natural64).sum(dim=(1, 2)).tolist()

        names = ["field", "w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]
        for fused in (True, False):
            field.grad = None
            mlp.zero_grad()
            cap = {}
            out = rd.render_train(to, td, gradient_scaling=scaling, rand=rand, fused=fused, capture=cap)
            loss_of(out["rgb"], out["accumulation"]).backward()
            got = [field.grad.clone()] + [p.grad.clone() for p in render.mlp_weights(mlp)]
            want_rgb, want = float64_statement(cap)
            np.testing.assert_allclose(out["rgb"].detach().cpu().numpy(), want_rgb.cpu().numpy(), rtol=0, atol=1e-5)
            n_s, flips = flipped_decisions(cap, fused)
            errs = [(nm, _rel(a, w)) for nm, a, w in zip(names, got, want)]
            print(f"config {(S, S_fine, biased, scaling)} {'fused' if fused else 'float32 autograd'}: {n_s} samples, ReLU decisions differing "
                  f"from float64 per layer {flips}; " + " ".join(f"{nm} {e:.1e}" for nm, e in errs))
            for nm, e in errs:
                # within 1e-5 of float64 on every tensor -- unless a ReLU decision differs: a unit whose pre-activation is within an
                # ulp of 0 contributes its whole (sample, unit) term or nothing, and one term can be 1e-3 of the largest entry
                # of a gradient that is a sum over 25k samples (the masked test below shows that this is ALL that differs)
                assert e < 1e-5 or sum(flips) > 0, ((S, S_fine, biased, scaling), "fused" if fused else "float32", nm, e, flips, errs)
                assert e < (4e-3 if nm == "field" else 3e-3), ((S, S_fine, biased, scaling), nm, e, flips)


def _decode_relu_masks(masks, n):
    """[4, n, 2] int64 as saved by tn_mlp_forward_gather_train -> bool [4, n, 128] in nn.Linear feature order: bit j of word
    (layer, sample, half h) = accumulator slot j = feature 32 (j >> 4) + (j & 3) + 8 ((j & 15) >> 2) + 4 h (tn_mlp_common.h)."""
    import torch

    j = torch.arange(64, device=masks.device)
    bits = ((masks[..., None] >> j) & 1).bool()
    out = torch.empty(4, n, 128, dtype=torch.bool, device=masks.device)
    for h in range(2):
        out[:, :, 32 * (j >> 4) + (j & 3) + 8 * ((j & 15) >> 2) + 4 * h] = bits[:, :, h, :]
    return out


def _statement(render, device, mlp, field, vi, bc, dirs, S, masks, dtype):
    """gather + TetraMLP in `dtype` on a given sample placement, the ReLU decisions either its own (masks = None) or the ones
    given; returns sigma, rgb, the leaves [field] + 12 weight tensors, and its own decisions [4, n, 128]."""
    import torch

    m = render.TetraMLP().to(device).to(dtype)
    m.load_state_dict({k: v.to(dtype) for k, v in mlp.state_dict().items()})
    f = field.detach().to(dtype).requires_grad_(True)
    b = bc.to(dtype)
    wts = torch.cat([1 - b.sum(-1, keepdim=True), b], -1)
    wts = torch.where(vi < 0, torch.zeros_like(wts), wts)
    x = (f.t()[vi.long().clamp_min(0)] * wts[..., None]).sum(-2)
    natural = []
    for l, lin in enumerate(m.base):
        pre = lin(x)
        natural.append(pre > 0)
        x = pre * (masks[l] if masks is not None else natural[-1]).to(dtype)
    sigma = torch.nn.functional.softplus(m.density(x))[..., 0]
    enc = render.direction_encoding(dirs.to(dtype))[:, None, :].expand(-1, S, -1).reshape(-1, 27)
    pre = m.head(torch.cat([enc, x], -1))
    natural.append(pre > 0)
    rgb = torch.sigmoid(m.rgb(pre * (masks[3] if masks is not None else natural[-1]).to(dtype)))
    return sigma, rgb, [f] + render.mlp_weights(m), torch.stack(natural)


@pytest.mark.parametrize("mesh_seed", [5, 6])
def test_training_gradients_match_float64_under_the_saved_relu_masks(tn, device, scenes, mesh_seed):
    """WHY the gradients of the tensors upstream of a hidden layer can sit 1e-4 .. 1e-3 (of the largest entry) away from float64
    autograd while the kernels are exact to 1e-7: a handful of ReLU units (1-5 of 3-4 million) whose pre-activation rounds to the
    other side of 0 in fp32 than in float64 -- float32 autograd flips as many, not always the same ones.  Shown, not asserted
    (profiles/r05e_grad_masks.txt): the float64 statement evaluated WITH THE MASKS THE FUSED FORWARD SAVED
    (read back from tn_mlp_forward_gather_train's buffer), on the same sample placement and upstream gradients, leaves only the
    arithmetic of k_mlp_backward / k_dw_gemm / k_interp_bwd -- and there the fused gradients are within 5e-6 on EVERY tensor incl.
    the field (measured: <= 2.3e-7 field, <= 1.4e-6 wh, at or below float32 autograd under the same masks), for all four
    configurations of the round-4 diagnostic on both mesh seeds.  The number of flipped bits is reported and bounded; a deviation
    from the unmasked float64 gradient beyond 1e-5 may only occur where bits flipped."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    cpp = tn.cpp
    pts, cells = scenes.random_mesh(4000, mesh_seed)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(512, 6)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    torch.manual_seed(123)
    target = torch.rand(len(o), 3, device=device)
    names = ["field", "w1", "b1", "w2", "b2", "w3", "b3", "wd", "bd", "wh", "bh", "wr", "br"]

    for S, S_fine, biased, scaling in ((24, 24, True, True), (24, 24, True, False), (24, 24, False, True), (32, 32, False, False)):
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(device)
        field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5)
        rd = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased)
        hit = int((tr.trace_rays(to, td, 256)["num_visited_cells"] > 0).sum())
        rand = {"coarse": torch.rand(hit, S + 1, device=device), "fine": torch.rand(hit, S_fine + 1, device=device)}
        cap = {}
        with torch.no_grad():
            rd.render_train(to, td, gradient_scaling=scaling, rand=rand, fused=True, capture=cap)
        vi, bc, edges, S2, dirs = cap["vertex_indices"], cap["barycentric_coordinates"], cap["edges"], cap["samples_per_ray"], cap["dirs"]
        n = vi.numel() // 4
        w = [x.detach() for x in render.mlp_weights(mlp)]
        sigma, rgb, saved = cpp.mlp_forward_gather_train(vi, bc, field, dirs, w, S2)
        masks = _decode_relu_masks(saved.masks.clone(), n)
        # upstream gradients: the float64 composite + loss at the fused forward's outputs
        dt = torch.float64
        sg = sigma.detach().to(dt).view(-1, S2, 1).requires_grad_(True)
        cl = rgb.detach().to(dt).view(-1, S2, 3).requires_grad_(True)
        e64, sg2, cl2 = edges.to(dt), sg, cl
        if scaling:
            spacing = (e64 - cap["near"].to(dt)) / (cap["far"].to(dt) - cap["near"].to(dt))
            cl2, sg2, _ = render.GradientScaler.apply(cl, sg, (spacing[:, 1:] + spacing[:, :-1])[..., None])
        rgb_r, acc_r, _, _ = render.composite(sg2, cl2, e64[:, :-1, None], e64[:, 1:, None])
        full_rgb = torch.ones(len(o), 3, dtype=dt, device=device).index_copy(0, cap["idx"], rgb_r)
        full_acc = torch.zeros(len(o), 1, dtype=dt, device=device).index_copy(0, cap["idx"], acc_r)
        (((full_rgb - target.to(dt)) ** 2).mean() + 0.1 * full_acc.mean()).backward()
        d_sigma, d_rgb = sg.grad.reshape(-1), cl.grad.reshape(-1, 3)
        gf, gw = cpp.mlp_backward(saved, vi, bc, field, dirs, w, sigma, rgb, d_sigma.float().contiguous(), d_rgb.float().contiguous())
        fused = [gf] + list(gw)
        res = {}
        for label, dtype, mk in (("f64 masked", torch.float64, masks), ("f32 masked", torch.float32, masks), ("f64 own", torch.float64, None)):
            s_, c_, leaves, natural = _statement(render, device, mlp, field, vi.reshape(n, 4), bc.reshape(n, 3), dirs, S2, mk, dtype)
            ((s_ * d_sigma.to(dtype)).sum() + (c_ * d_rgb.to(dtype)).sum()).backward()
            res[label] = ([x.grad for x in leaves], natural)
        flipped = (masks != res["f64 own"][1]).sum(dim=(1, 2)).tolist()
        print(f"mesh seed {mesh_seed} config {(S, S_fine, biased, scaling)}: {n} samples, flipped ReLU bits per layer (of {n * 128}): {flipped}")
        assert max(flipped) <= 32, flipped                      # a handful of 3-4 million
        for k, name in enumerate(names):
            ours, t32 = _rel(fused[k], res["f64 masked"][0][k]), _rel(res["f32 masked"][0][k], res["f64 masked"][0][k])
            assert ours < max(5.0 * t32, 5e-6), ((S, S_fine, biased, scaling), name, ours, t32)
            unmasked = _rel(fused[k], res["f64 own"][0][k])
            assert unmasked < 1e-5 or sum(flipped) > 0, ((S, S_fine, biased, scaling), name, unmasked, flipped)


def test_training_gradients_are_bit_reproducible_without_atomics(tn, device, scenes):
    """With cpp.DETERMINISTIC_FIELD_GRADIENT (or torch.use_deterministic_algorithms) the gather's adjoint runs without float
    atomics: ALL thirteen gradients of a training step -- the field's included -- are the same bits run after run (4096-ray
    batch of the 300k-tet mesh, both shipped configurations); against the atomic form the field gradient differs by rounding."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(45000, 2)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(4096, 1)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    torch.manual_seed(5)
    target = torch.rand(len(o), 3, device=device)
    for S, S_fine, biased, scaling in ((256, 256, False, False), (128, 128, True, True)):
        torch.manual_seed(1)
        mlp = render.TetraMLP().to(device)
        field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
        rd = render.TetraRenderer(tr, field, mlp, S, 512, fused=True, num_fine_samples=S_fine, biased=biased)

        def grads():
            field.grad = None
            mlp.zero_grad()
            torch.manual_seed(9)
            out = rd.render_train(to, td, gradient_scaling=scaling)
            ((out["rgb"] - target) ** 2).mean().backward()
            return [field.grad.clone()] + [p.grad.clone() for p in render.mlp_weights(mlp)]

        atomic = grads()
        tn.cpp.DETERMINISTIC_FIELD_GRADIENT = True
        try:
            a, b = grads(), grads()
        finally:
            tn.cpp.DETERMINISTIC_FIELD_GRADIENT = False
        assert len(a) == 13
        for x, y in zip(a, b):
            assert torch.equal(x.view(torch.int32), y.view(torch.int32))
        for x, y in zip(a[1:], atomic[1:]):                     # the weight gradients never came from atomics
            assert torch.equal(x, y)
        assert float(a[0].abs().max()) > 0
        assert float((a[0] - atomic[0]).abs().max()) <= 2e-5 * float(atomic[0].abs().max())


def test_render_train_in_several_autograd_nodes(tn, device, scenes):
    """Batches beyond `train_node_samples` go through several fused-MLP nodes (one per block of rays): outputs identical,
    gradients equal up to the order in which the blocks' parameter gradients are summed."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(4000, 5)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(700, 8)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    torch.manual_seed(5)
    target = torch.rand(len(o), 3, device=device)
    mlp = render.TetraMLP().to(device)
    field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
    S, S_fine = 32, 32
    rd = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine)
    hit = int((tr.trace_rays(to, td, 256)["num_visited_cells"] > 0).sum())
    rand = {"coarse": torch.rand(hit, S + 1, device=device), "fine": torch.rand(hit, S_fine + 1, device=device)}
    res = []
    for node_samples in (1 << 22, 100 * (S + S_fine + 1)):      # one node; nodes of 100 rays
        rd.train_node_samples = node_samples
        field.grad = None
        mlp.zero_grad()
        out = rd.render_train(to, td, rand=rand)
        ((out["rgb"] - target) ** 2).mean().backward()
        res.append((out["rgb"].detach().clone(), field.grad.clone(), [p.grad.clone() for p in render.mlp_weights(mlp)]))
    assert hit > 300
    assert torch.equal(res[0][0], res[1][0])
    for a, b in [(res[0][1], res[1][1])] + list(zip(res[0][2], res[1][2])):
        scale = float(a.abs().max()) + 1e-30
        assert float((a - b).abs().max()) <= 2e-5 * scale, float((a - b).abs().max()) / scale


def test_training_iterations_do_not_leak(tn, device, scenes):
    """The fused MLP node saves 2.3 KB per sample for its backward pass; nothing of it may outlive the iteration (a node that
    references its own outputs from Python forms a cycle through the C++ graph that no collector breaks)."""
    import gc
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(4000, 5)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(1024, 9)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    mlp = render.TetraMLP().to(device)
    field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
    opt = torch.optim.SGD([field] + list(mlp.parameters()), lr=1e-3)
    rd = render.TetraRenderer(tr, field, mlp, 64, 256, fused=True, num_fine_samples=64)

    def step():
        opt.zero_grad(set_to_none=True)
        out = rd.render_train(to, td)
        (out["rgb"] ** 2).mean().backward()
        opt.step()

    gc.disable()        # reference counting alone must release everything
    try:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        before = torch.cuda.memory_allocated(device)
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        after = torch.cuda.memory_allocated(device)
    finally:
        gc.enable()
    assert after - before < 8 << 20, f"{(after - before) / 2**20:.1f} MiB more allocated after 5 further iterations"


def test_render_train_sync_free(tn, device, scenes):
    """render_train's default form never synchronises the host with the device (torch.cuda.set_sync_debug_mode("error")
    around forward + backward + an optimiser step): the batch is processed at its full size, hitting rays first, padded
    entries masked out.  Against the compacting form (torch.nonzero): identical when every ray hits (same draws, same
    kernels on the same rows); with a third of the rays missing -- and with ALL rays missing -- the outputs of the missing
    rays are the miss values, the hitting rays' outputs are finite renders, and the gradients are finite (the padded
    entries contribute exact zeros)."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(4000, 5)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    o, d = scenes.outside_in_rays(1024, 6)
    for S, S_fine, biased, scaling in ((32, 32, False, False), (24, 24, True, True)):
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(device)
        field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
        params = [field] + list(mlp.parameters())
        opt = torch.optim.SGD(params, lr=1e-4)
        free = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased,
                                    sync_free_min_hits=0.0)     # (never fall back to compaction: see the adaptive test below)
        sync = render.TetraRenderer(tr, field, mlp, S, 256, fused=True, num_fine_samples=S_fine, biased=biased, sync_free_train=False)
        assert free.sync_free_train and not sync.sync_free_train
        for case in ("all hit", "a third misses", "all miss"):
            dd = d.copy()
            if case == "a third misses":
                dd[::3] = -dd[::3]
            elif case == "all miss":
                dd = -dd
            to, td = torch.from_numpy(o).to(device), torch.from_numpy(np.ascontiguousarray(dd)).to(device)
            target = torch.rand(len(o), 3, device=device)

            def step(rd, guard):
                opt.zero_grad(set_to_none=True)
                torch.manual_seed(3)
                if guard:
                    torch.cuda.set_sync_debug_mode("error")
                try:
                    out = rd.render_train(to, td, gradient_scaling=scaling)
                    loss = ((out["rgb"] - target) ** 2).mean() + 0.1 * out["accumulation"].mean()
                    loss.backward()
                finally:
                    torch.cuda.set_sync_debug_mode("default")
                return out, [None if p.grad is None else p.grad.clone() for p in params]

            step(free, False)                      # first call: allocations, weight packing
            out_f, g_f = step(free, True)          # ... then not one synchronisation
            if case == "all miss":      # (the compacting form returns constants without a graph then, like the reference)
                with torch.no_grad():
                    out_s, g_s = sync.render_train(to, td, gradient_scaling=scaling), None
            else:
                out_s, g_s = step(sync, False)
            assert torch.equal(out_f["ray_mask"], out_s["ray_mask"])
            miss = ~out_f["ray_mask"]
            assert bool((out_f["rgb"][miss] == 1.0).all()) and bool((out_f["accumulation"][miss] == 0.0).all())
            assert bool((out_f["depth"][miss] == 1000.0).all())
            assert bool(torch.isfinite(out_f["rgb"]).all())
            if case == "all hit":
                assert int(miss.sum()) == 0
                for k in ("rgb", "accumulation", "depth"):
                    assert torch.equal(out_f[k], out_s[k]), k
                for a, b in zip(g_f[1:], g_s[1:]):        # the 12 weight gradients are reproducible; the field's is atomic
                    assert torch.equal(a, b)
                assert float((g_f[0] - g_s[0]).abs().max()) <= 1e-5 * float(g_s[0].abs().max())
            elif case == "a third misses":
                assert 0.25 < float(miss.float().mean()) < 0.45
                hit = out_f["ray_mask"]
                assert float(out_f["accumulation"][hit].mean()) > 0.05
                for g in g_f:
                    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
                # the same distribution, another random stream: the mean colour of the hitting rays agrees
                assert abs(float(out_f["rgb"][hit].mean()) - float(out_s["rgb"][hit].mean())) < 0.02
            else:
                assert int(out_f["ray_mask"].sum()) == 0
                for g in g_f:
                    assert g is None or (bool(torch.isfinite(g).all()) and float(g.abs().max()) == 0.0)


def test_render_train_falls_back_to_compaction_when_rays_miss(tn, device, scenes):
    """The sync-free form computes (and discards) padded entries for the rays that miss, so it only pays while nearly all rays
    hit: the hit fraction of a batch is read back asynchronously and a later batch whose last known fraction is below
    SYNC_FREE_MIN_HITS (0.85) is compacted with torch.nonzero instead -- and the sync-free form resumes when it recovers."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(3000, 5)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(device)
    field = ((torch.rand(64, len(pts), device=device) * 2 - 1) * 0.5).requires_grad_(True)
    rd = render.TetraRenderer(tr, field, mlp, 32, 256, fused=True, num_fine_samples=32)
    o, d = scenes.outside_in_rays(512, 6)
    dm = d.copy()
    dm[::2] = -dm[::2]                                 # half of the rays miss
    to = torch.from_numpy(o).to(device)
    hit_all, hit_half = torch.from_numpy(d).to(device), torch.from_numpy(np.ascontiguousarray(dm)).to(device)

    def call(dirs, guard):
        torch.cuda.set_sync_debug_mode("error" if guard else "default")
        try:
            out = rd.render_train(to, dirs)
            out["rgb"].mean().backward()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        torch.cuda.synchronize()                       # (the asynchronous read-back of this batch's hit count has landed)
        return out

    call(hit_all, False)
    call(hit_all, True)                                # every ray hits: sync-free
    assert rd._hit_fraction == 1.0
    call(hit_half, True)                               # still decided on the previous batch: sync-free, with padded entries
    call(hit_half, False)                              # now the 0.5 is known
    assert 0.45 < rd._hit_fraction < 0.55
    with pytest.raises(RuntimeError, match="synchroniz"):
        call(hit_half, True)                           # compaction: torch.nonzero synchronises
    call(hit_all, False)                               # (compacting call; its read-back says 1.0 again)
    call(hit_all, False)
    call(hit_all, True)                                # sync-free resumed
