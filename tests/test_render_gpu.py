"""GPU tests of the fused fp32-MFMA MLP, the composite kernel and the whole render path
(config C3 of BASELINE.json: 100k-tet mesh, shallow-MLP forward, fp32, tolerance 1e-5) against the
plain-PyTorch fp32 statement of the same arithmetic (render.py) evaluated on the CPU with the oracle."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def render():
    return importlib.import_module("tetra-nerf_amd.render")


@pytest.fixture(params=["fp32", "bf16x3"])
def mlp_mode(request):
    """Both arithmetic modes of the fused MLP kernels (a per-call argument) against the same 1e-5 bar."""
    return request.param


def _model(render, seed=0, field_scale=1.0):
    import torch

    torch.manual_seed(seed)
    return render.TetraMLP()


def test_mlp_forward_matches_torch(tn, device, render, mlp_mode):
    import torch

    mlp = _model(render)
    torch.manual_seed(1)
    for R, S in ((8, 256), (5, 37), (1, 1), (300, 64)):
        n = R * S
        feats = torch.randn(n, 64) * 0.7
        dirs = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
        with torch.no_grad():
            ws, wc = mlp(feats.double().float(), dirs[:, None, :].expand(R, S, 3).reshape(n, 3))
        gm = mlp.to(device)
        feats_fm = feats.t().contiguous().to(device)
        sigma, rgb = tn.cpp.mlp_forward(feats_fm, dirs.to(device), render.mlp_weights(gm), S, mode=mlp_mode)
        mlp.cpu()
        np.testing.assert_allclose(sigma.cpu().numpy(), ws[:, 0].numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(rgb.cpu().numpy(), wc.numpy(), rtol=0, atol=1e-5)


def test_mlp_forward_gather_equals_two_step(tn, device, render, mlp_mode):
    """Fused gather+MLP must produce exactly the bits of interpolate_values followed by mlp_forward up to
    the K order of layer 1 (different, so compare at fp32 round-off) and match the torch statement."""
    import torch

    mlp = _model(render, 5).to(device)
    rng = np.random.default_rng(6)
    V, R, S = 3000, 40, 96
    vi = rng.integers(0, V, (R, S, 4)).astype(np.int32)
    vi[rng.random((R, S)) < 0.25] = -1
    bc = (rng.random((R, S, 3)).astype(np.float32)) / 4
    bc[vi[..., 0] < 0] = 0
    field = torch.randn(64, V, device=device)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=device), dim=-1)
    tvi, tbc = torch.from_numpy(vi).to(device), torch.from_numpy(bc).to(device)
    w = render.mlp_weights(mlp)
    feats = tn.cpp.interpolate_values(tvi, tbc, field)
    s2, c2 = tn.cpp.mlp_forward(feats.moveaxis(-1, 0).reshape(64, -1), dirs, w, S, mode=mlp_mode)
    s1, c1 = tn.cpp.mlp_forward_gather(tvi, tbc, field, dirs, w, S, mode=mlp_mode)
    np.testing.assert_allclose(s1.cpu().numpy(), s2.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(c1.cpu().numpy(), c2.cpu().numpy(), rtol=0, atol=2e-6)
    with torch.no_grad():
        ws, wc = mlp(feats, dirs[:, None, :].expand(R, S, 3))
    np.testing.assert_allclose(s1.cpu().numpy(), ws.reshape(-1).cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c1.cpu().numpy(), wc.reshape(-1, 3).cpu().numpy(), rtol=0, atol=1e-5)


def test_density_only_and_weights_only(tn, device, render, mlp_mode):
    """Coarse pass of the model (model.py:577-582): mlp_base + density head only, then get_weights."""
    import torch

    mlp = _model(render, 8).to(device)
    rng = np.random.default_rng(9)
    V, R, S = 2000, 33, 128
    vi = rng.integers(0, V, (R, S, 4)).astype(np.int32)
    vi[rng.random((R, S)) < 0.3] = -1
    bc = (rng.random((R, S, 3)).astype(np.float32)) / 4
    field = torch.randn(64, V, device=device)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=device), dim=-1)
    tvi, tbc = torch.from_numpy(vi).to(device), torch.from_numpy(bc).to(device)
    w = render.mlp_weights(mlp)
    s_full, _ = tn.cpp.mlp_forward_gather(tvi, tbc, field, dirs, w, S, mode=mlp_mode)
    s_only = tn.cpp.mlp_forward_gather(tvi, tbc, field, None, w, S, mode=mlp_mode)
    # the density head alone accumulates the same products in the same order as the 5th tile of the head layer
    np.testing.assert_allclose(s_only.cpu().numpy(), s_full.cpu().numpy(), rtol=1e-6, atol=1e-6)
    edges = (1 + torch.cumsum(torch.rand(R, S + 1, device=device) * 0.02, -1)).contiguous()
    w_only = tn.cpp.composite(s_only.view(R, S), None, edges)
    want = render.ray_weights(s_only.view(R, S), edges)
    np.testing.assert_allclose(w_only.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-6)


def test_mlp_forward_vs_float64(tn, device, render, mlp_mode):
    """fp32 MFMA = exact fp32 fma chain; bf16x3 = six exact partial products per multiply with the dropped
    terms below 2^-24: in both modes the error vs a float64 evaluation stays at fp32 round-off."""
    import torch

    mlp = _model(render, 3)
    torch.manual_seed(4)
    R, S = 64, 128
    feats = torch.randn(R * S, 64)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3), dim=-1)
    m64 = render.TetraMLP().double()
    m64.load_state_dict({k: v.double() for k, v in mlp.state_dict().items()})
    with torch.no_grad():
        ws, wc = m64(feats.double(), dirs.double()[:, None, :].expand(R, S, 3).reshape(-1, 3))
    gm = mlp.to(device)
    sigma, rgb = tn.cpp.mlp_forward(feats.t().contiguous().to(device), dirs.to(device), render.mlp_weights(gm), S, mode=mlp_mode)
    assert float((sigma.cpu().double() - ws[:, 0]).abs().max()) < 5e-6
    assert float((rgb.cpu().double() - wc).abs().max()) < 2e-6


def test_composite_matches_torch(tn, device, render):
    import torch

    torch.manual_seed(2)
    for R, S in ((16, 256), (7, 65), (3, 2), (100, 128)):
        sigma = torch.rand(R, S) * 20
        sigma[0] = 0            # empty ray: background only, depth = last sample
        if R > 1:
            sigma[1] = 1e4      # opaque at the first sample
        rgb = torch.rand(R, S, 3)
        near = torch.rand(R, 1) + 0.5
        edges = near + torch.cumsum(torch.rand(R, S + 1) * 0.02, -1)
        w_rgb, w_acc, w_depth, w_w = render.composite(sigma[..., None], rgb, edges[:, :-1, None], edges[:, 1:, None])
        g_rgb, g_acc, g_depth, g_w = tn.cpp.composite(sigma.to(device), rgb.to(device), edges.to(device),
                                                      return_weights=True)
        np.testing.assert_allclose(g_w.cpu().numpy(), w_w[..., 0].numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(g_rgb.cpu().numpy(), w_rgb.numpy(), rtol=0, atol=1e-5)
        np.testing.assert_allclose(g_acc.cpu().numpy(), w_acc.numpy(), rtol=0, atol=1e-5)
        # median depth: the tolerance of north_star (1e-5) on every ray whose median is DECIDED -- the chosen sample
        # may only differ where a cumulative weight of the reference sits within round-off of 0.5 (then the two
        # summation orders legitimately pick neighbouring samples), and in that case it must be the neighbour
        same = np.isclose(g_depth.cpu().numpy(), w_depth.numpy(), rtol=0, atol=1e-5)[:, 0]
        cum = torch.cumsum(w_w[..., 0].double(), -1).numpy()
        undecided = (np.abs(cum - 0.5) < 1e-5 * S).any(-1)
        assert np.all(same | undecided), f"{R}x{S}: median depth differs on a decided ray"
        mid = ((edges[:, :-1] + edges[:, 1:]) / 2).numpy()
        for r in np.nonzero(~same)[0]:
            i_ref = int(np.argmin(np.abs(mid[r] - float(w_depth[r, 0]))))
            i_got = int(np.argmin(np.abs(mid[r] - float(g_depth[r, 0]))))
            assert abs(i_ref - i_got) <= 1, (r, i_ref, i_got)
        assert same.mean() >= 0.9


@pytest.mark.parametrize("cfg", ["coarse", "tetra-nerf-original", "tetra-nerf"])
def test_render_c3(tn, device, oracle, scenes, render, cfg, mlp_mode):
    """Config C3: 100k-tet mesh (seed 1), field U(-1e-4,1e-4) with colour rows, default-init MLP
    (torch.manual_seed(0)), 4096 rays: sigma-weights-RGB within 1e-5 of the CPU fp32 statement (oracle
    tracer + torch MLP).  coarse: 256 uniform samples; tetra-nerf-original: 256 uniform + 256 PDF
    (513 fine samples); tetra-nerf: biased 128 + 128 PDF (registration.py:55-57)."""
    import torch

    S, S_fine, biased = {"coarse": (256, 0, False), "tetra-nerf-original": (256, 256, False),
                         "tetra-nerf": (128, 128, True)}[cfg]

    pts, cells = scenes.random_mesh(15000, 1)
    torch.manual_seed(0)
    mlp = render.TetraMLP()
    field = (torch.rand(64, len(pts)) * 2 - 1) * 1e-4
    field[1:4] = torch.rand(3, len(pts)) * 2 - 1
    if cfg != "coarse":
        field[0] = torch.rand(len(pts)) * 6 - 3    # a density-driving row so that the PDF pass has structure
    o, d = scenes.outside_in_rays(4096, 7)   # BASELINE configs[2]: 4096 rays, all three configurations
    M = 512

    class CpuTracer:
        def __init__(self):
            self.t = oracle.OracleTracer(use_bvh=True)
            self.t.load_tetrahedra(pts, cells)

        def trace_rays(self, o_, d_, M_):
            return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.t.trace_rays(o_.numpy(), d_.numpy(), M_).items()}

        def find_visited_cells(self, *a):
            r = self.t.find_visited_cells(*[x.numpy() for x in a])
            return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in r.items()}

    def cpu_interp(vi, bc, f):
        return torch.from_numpy(np.ascontiguousarray(oracle.interpolate_values(vi.numpy(), bc.numpy(), f.numpy())))

    with torch.no_grad():
        want = render.render_reference(CpuTracer(), cpu_interp, field, mlp, torch.from_numpy(o), torch.from_numpy(d), S, M,
                                       num_fine_samples=S_fine, biased=biased)

    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    gm = render.TetraMLP()
    gm.load_state_dict(mlp.state_dict())
    gm = gm.to(device)
    # fused_pass: everything after the trace ONE persistent launch (tn_render_rays); fused without it: separate
    # match / fused gather+MLP / composite kernels; fused=False: the HIP ops + the PyTorch MLP
    for fused, fused_pass in ((True, True), (True, False), (False, False)):
        rd = render.TetraRenderer(tr, field.to(device), gm, S, M, fused=fused, num_fine_samples=S_fine, biased=biased,
                                  fused_pass=fused_pass, mlp_mode=mlp_mode)
        got = rd.render(torch.from_numpy(o).to(device), torch.from_numpy(d).to(device))
        assert torch.equal(got["ray_mask"].cpu(), want["ray_mask"])
        np.testing.assert_allclose(got["rgb"].cpu().numpy(), want["rgb"].numpy(), rtol=0, atol=1e-5,
                                   err_msg=f"fused={fused} fused_pass={fused_pass}")
        np.testing.assert_allclose(got["accumulation"].cpu().numpy(), want["accumulation"].numpy(), rtol=0, atol=1e-5)
        # median depth: asserted at 1e-5 on every DECIDED ray: cumulative weights further from the threshold 0.5 at
        # every sample than two fp32 evaluations of the weights can drift apart over the ray (4e-6 per sample) -- they
        # cannot disagree about the median bin then; undecided rays may land in a neighbouring bin.  (In the coarse
        # configuration the field is ~0, sigma = softplus(0) = ln 2 and a ray of length 1 accumulates exactly 0.5: a
        # third of the rays is undecided by construction.)
        n_final = S + S_fine + 1 if S_fine else S
        decided = (want["depth_margin"].numpy() > 4e-6 * n_final)[:, 0] & want["ray_mask"].numpy()
        assert decided.sum() > 200, decided.sum()
        # and overall (round-2 form): at most 2 % of the rays land in a neighbouring bin
        assert np.isclose(got["depth"].cpu().numpy(), want["depth"].numpy(), rtol=0, atol=1e-4).mean() > 0.98
        np.testing.assert_allclose(got["depth"].cpu().numpy()[decided], want["depth"].numpy()[decided], rtol=0, atol=1e-5,
                                   err_msg=f"depth fused={fused} fused_pass={fused_pass}")


@pytest.mark.parametrize("scale_w,scale_x", [(1.0, 1.0), (1e-4, 1e4), (1e4, 1e-4), (1e-3, 1e-3), (30.0, 30.0)])
def test_bf16x3_error_bound_per_layer(tn, device, render, scale_w, scale_x):
    """Error bound of the split-operand mode (tn_mlp_forward mode 1: every fp32 operand = three bf16 pieces, six partial
    products per multiply, fp32 accumulation) on ONE layer, against float64, over adversarial operand scales and
    cancelling sums.  Layers 2 and 3 are the identity and the density head reads a single feature, so
    softplus^-1(sigma) IS one output of layer 1: y_k = sum_i W1[k,i] x_i + b1[k].  Claim: |y - y64| <= 2^-21 * sum_i
    |W1[k,i] x_i| (dropped split terms < 2^-24 per product, plus the fp32 accumulation of 64 terms) -- the same bound
    the exact-fp32 MFMA mode meets; both modes are checked."""
    import torch

    torch.manual_seed(11)
    n, S = 4096, 64
    w1 = torch.randn(128, 64, dtype=torch.float64) * scale_w
    x = torch.randn(n, 64, dtype=torch.float64) * scale_x
    x[:, 1::2] = -x[:, 0::2] * (w1[0, 0::2] / w1[0, 1::2]) * (1 + 1e-3 * torch.randn(n, 32, dtype=torch.float64))   # row 0 cancels to 1e-3
    x[::7] *= 1e-30 / scale_x            # denormal-range activations
    b1 = torch.zeros(128, dtype=torch.float64)
    shift = float((w1.abs() @ x.abs().t()).max()) * 1.5 + 1.0       # keeps every pre-activation positive through the ReLUs
    b1 += shift
    eye = torch.eye(128, dtype=torch.float64)
    zeros = lambda *s: torch.zeros(*s, dtype=torch.float64)   # noqa: E731
    for k in (0, 5, 127):
        wd = zeros(1, 128); wd[0, k] = 1.0
        ws = [w1, b1, eye, zeros(128), eye, zeros(128), wd, torch.tensor([-shift], dtype=torch.float64), zeros(128, 155), zeros(128), zeros(3, 128), zeros(3)]
        w32 = [t.float().contiguous().to(device) for t in ws]
        w64 = [t.double().cpu() for t in w32]                      # the fp32-representable weights, in float64
        x32 = x.float().contiguous().to(device)
        x64 = x32.double().cpu()
        y64 = (x64 @ w64[0][k]) + w64[1][k] - float(w64[7][0].neg())  # = sum_i W x_i + b1 - shift
        bound = 2.0 ** -21 * ((x64.abs() @ w64[0][k].abs()) + abs(float(w64[1][k])) + shift)
        dirs = torch.nn.functional.normalize(torch.randn(n // S, 3), dim=-1).to(device)
        for mode in ("fp32", "bf16x3"):
            sigma, _ = tn.cpp.mlp_forward(x32.t().contiguous(), dirs, w32, S, mode=mode)
            s = sigma.double().cpu()
            y = torch.where(s > 20, s, torch.log(torch.expm1(s.clamp_min(1e-30))))      # softplus^-1
            ok = s > 1e-3                                                                # (below: softplus^-1 is ill-conditioned)
            err = (y - y64).abs()
            assert int(ok.sum()) > n // 4
            assert bool((err[ok] <= bound[ok] + 4e-7 * y64.abs()[ok]).all()), (mode, k, float((err[ok] / bound[ok]).max()))


def test_ray_head_bias_on_every_forward_kernel(tn, device, scenes):
    """The per-ray head bias (appearance embedding) is added by four kernels -- the one-launch render pass,
    the fp32 gather + MLP kernel of the kernel chain, its bf16x3 form, and the training forward (against the reference body:
    tests/test_reference_model_gpu.py) -- which must agree with each other, and a zero bias must reproduce the call without
    one bit for bit."""
    import torch

    render = importlib.import_module("tetra-nerf_amd.render")
    pts, cells = scenes.random_mesh(5000, 9)
    tr = tn.TetrahedraTracer(device)
    tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
    torch.manual_seed(2)
    mlp = render.TetraMLP().to(device)
    field = torch.randn(64, len(pts), device=device) * 0.5
    o, d = scenes.outside_in_rays(3000, 4)
    to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
    bias = torch.randn(len(o), 128, device=device) * 0.7
    outs = {}
    for name, kw in (("one launch", dict(fused_pass=True)), ("chain fp32", dict(fused_pass=False)),
                     ("chain bf16x3", dict(fused_pass=False, mlp_mode="bf16x3"))):
        rd = render.TetraRenderer(tr, field, mlp, 64, 256, fused=True, num_fine_samples=64, **kw)
        plain = rd.render(to, td)
        zero = rd.render(to, td, ray_head_bias=torch.zeros_like(bias))
        for k in ("rgb", "accumulation", "depth"):
            assert torch.equal(plain[k], zero[k]), (name, k)
        outs[name] = rd.render(to, td, ray_head_bias=bias)
        assert float((outs[name]["rgb"] - plain["rgb"]).abs().max()) > 1e-2, name        # the bias acts
        assert torch.equal(outs[name]["accumulation"], plain["accumulation"]), name      # ... on the colours only
    ref = outs["chain fp32"]
    np.testing.assert_allclose(outs["one launch"]["rgb"].cpu().numpy(), ref["rgb"].cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs["chain bf16x3"]["rgb"].cpu().numpy(), ref["rgb"].cpu().numpy(), rtol=0, atol=1e-5)
