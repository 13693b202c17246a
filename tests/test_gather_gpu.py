"""GPU parity tests of interpolate_values forward / backward through the C-ABI.

Forward: bit-exact against the oracle (same summation order), and against the reference's own
definition -- einsum('jrbi,rbi->rbj') -- within assert_allclose defaults
(tests/test_tetrahedra_tracer.py:400-416).  Backward: atomics reorder the sum, so it is compared
with the oracle / autograd of the einsum at rtol 1e-4 (the reference test uses
assert_allclose defaults on integer-valued fields, :436-453)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(rng, V, shape, D, empty_frac=0.2):
    vi = rng.integers(0, V, shape + (D,)).astype(np.int32)
    empty = rng.random(shape) < empty_frac
    vi[empty] = -1
    bc = (rng.random(shape + (D - 1,)).astype(np.float32)) / D
    bc[empty] = 0
    return vi, bc


@pytest.mark.parametrize("D", [2, 3, 4, 6])
@pytest.mark.parametrize("Fd", [64, 7, 130])
def test_forward_bit_exact(tn, device, oracle, D, Fd):
    import torch

    rng = np.random.default_rng(D * 100 + Fd)
    V = 777
    vi, bc = _inputs(rng, V, (33, 19), D)
    field = rng.standard_normal((Fd, V)).astype(np.float32)
    want = oracle.interpolate_values(vi, bc, field)
    got = tn.cpp.interpolate_values(torch.from_numpy(vi).to(device), torch.from_numpy(bc).to(device),
                                    torch.from_numpy(field).to(device))
    assert tuple(got.shape) == (33, 19, Fd)
    # same memory layout as the reference: a moveaxis(0,-1) view of a contiguous [Fd, n] buffer
    assert got.moveaxis(-1, 0).is_contiguous()
    np.testing.assert_array_equal(got.cpu().numpy().view(np.uint32), np.ascontiguousarray(want).view(np.uint32))


def test_forward_backward_reference_definition(tn, device):
    """Restates test_tetrahedra_interpolate_values (tests/test_tetrahedra_tracer.py:346-456):
    256 rays x 256 samples, field [64,V].random_()."""
    import torch

    rng = np.random.default_rng(3)
    V, R, S = 2549, 256, 256
    vi_np, bc_np = _inputs(rng, V, (R, S), 4, empty_frac=0.35)
    vi = torch.from_numpy(vi_np).to(device)
    bc = torch.from_numpy(bc_np).to(device)
    torch.manual_seed(0)
    field = torch.empty((64, V), dtype=torch.float32, device=device).random_(0, 1000)

    def get_field_safe(f):
        safe_vi = vi.long().clamp_min(0)
        g = f[:, safe_vi]
        return torch.where(vi >= 0, g, torch.zeros_like(g))

    w = torch.cat((1 - bc.sum(-1, keepdim=True), bc), -1)
    val = tn.interpolate_values(vi, bc, field)
    assert val.shape == (R, S, 64)
    gt = torch.einsum("jrbi,rbi->rbj", get_field_safe(field), w)
    torch.testing.assert_close(val, gt, rtol=1.3e-6, atol=1e-2)  # values up to ~1e3: a few fp32 ulps

    field.requires_grad_(True)
    val = tn.interpolate_values(vi, bc, field)
    val.sum().backward()
    grad = field.grad
    assert grad.shape == (64, V)
    field2 = field.detach().clone().requires_grad_(True)
    gt = torch.einsum("jrbi,rbi->rj", get_field_safe(field2), w)
    gt.sum().backward()
    torch.testing.assert_close(grad, field2.grad, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("D", [3, 4])
def test_backward_vs_oracle(tn, device, oracle, D):
    import torch

    rng = np.random.default_rng(17 + D)
    V, Fd = 321, 64
    vi, bc = _inputs(rng, V, (2000,), D)
    # runs of identical vertex tuples (consecutive samples in one tetrahedron)
    vi[100:140] = vi[100]
    bc[100:140] = bc[100]
    field = rng.standard_normal((Fd, V)).astype(np.float32)
    g = rng.standard_normal((2000, Fd)).astype(np.float32)
    want = oracle.interpolate_values_backward(vi, bc, field, g)
    got = tn.cpp.interpolate_values_backward(torch.from_numpy(vi).to(device), torch.from_numpy(bc).to(device),
                                             torch.from_numpy(field).to(device), torch.from_numpy(g).to(device))
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,Fd", [(4, 64), (3, 64), (4, 37), (2, 130)])
def test_backward_without_atomics_is_reproducible_and_agrees(tn, device, oracle, D, Fd):
    """tn_interpolate_values_backward_vm_det (cpp.DETERMINISTIC_FIELD_GRADIENT / torch.use_deterministic_algorithms): one writer
    per gradient element, fixed summation order -- the same bits on every run (the atomic kernel's sums depend on the order
    the hardware retires its atomics in), the oracle's values, EMPTY slots skipped, vertices nobody samples left at zero."""
    import torch

    rng = np.random.default_rng(23 + D + Fd)
    V = 501
    vi, bc = _inputs(rng, V, (6000,), D)
    vi[200:900] = vi[200]; bc[200:900] = bc[200]          # a long run of one tuple: vertices with hundreds of pairs
    vi[vi == 77] = 78                                      # a vertex nobody samples
    vi[1000:1100, 0] = -1                                  # unmatched slots (TN_EMPTY)
    field = rng.standard_normal((Fd, V)).astype(np.float32)
    g = rng.standard_normal((6000, Fd)).astype(np.float32)
    want = oracle.interpolate_values_backward(vi, bc, field, g)
    args = (torch.from_numpy(vi).to(device), torch.from_numpy(bc).to(device), torch.from_numpy(field).to(device),
            torch.from_numpy(g).to(device))
    atomic = tn.cpp.interpolate_values_backward(*args)
    assert not tn.cpp.deterministic_gradients()
    tn.cpp.DETERMINISTIC_FIELD_GRADIENT = True
    try:
        assert tn.cpp.deterministic_gradients()
        runs = [tn.cpp.interpolate_values_backward(*args) for _ in range(3)]
    finally:
        tn.cpp.DETERMINISTIC_FIELD_GRADIENT = False
    for r in runs[1:]:
        assert torch.equal(r.view(torch.int32), runs[0].view(torch.int32))
    np.testing.assert_allclose(runs[0].cpu().numpy(), want, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(runs[0].cpu().numpy(), atomic.cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert float(runs[0][:, 77].abs().max()) == 0.0
    # PyTorch's own switch selects it too
    prev = torch.are_deterministic_algorithms_enabled()
    torch.use_deterministic_algorithms(True, warn_only=True)
    try:
        assert tn.cpp.deterministic_gradients()
    finally:
        torch.use_deterministic_algorithms(prev)


def test_gather_errors(tn, device):
    import torch

    vi = torch.zeros((8, 5), dtype=torch.int32, device=device)
    bc = torch.zeros((8, 4), dtype=torch.float32, device=device)
    field = torch.zeros((64, 10), dtype=torch.float32, device=device)
    with pytest.raises(RuntimeError, match="Unsupported interpolation dimension with value 5"):
        tn.cpp.interpolate_values(vi, bc, field)
    with pytest.raises(RuntimeError, match="int32"):
        tn.cpp.interpolate_values(vi.long(), bc, field)
    with pytest.raises(RuntimeError, match="same last dimension"):
        tn.cpp.interpolate_values(vi, bc[:, :3].contiguous(), field)
    # empty batch
    out = tn.cpp.interpolate_values(torch.zeros((0, 4), dtype=torch.int32, device=device),
                                    torch.zeros((0, 3), device=device), field)
    assert tuple(out.shape) == (0, 64)


def test_gather_uint32(tn, device):
    """Mirrors tests/test_uint32.py::test_gather_uint32 of the reference."""
    import torch

    torch.manual_seed(0)
    for dt in (torch.float32, torch.float64):
        vals = torch.rand((5,), dtype=dt, device=device)
        indices = torch.randint(0, 5, (12,), dtype=torch.int32, device=device)
        res = tn.gather_uint32(vals, 0, indices)
        torch.testing.assert_close(res, vals[indices.long()], rtol=0, atol=0)
    with pytest.raises(Exception):
        tn.gather_uint32(torch.rand((5, 3), device=device), 0, torch.randint(0, 3, (5, 8), device=device))


def test_scatter_ema_uint32(tn, device):
    """Mirrors tests/test_uint32.py::test_scatter_ema_uint32 of the reference."""
    import torch

    torch.manual_seed(0)
    for dt in (torch.float32, torch.float64):
        tensor = torch.rand((10,), dtype=dt, device=device)
        indices = torch.tensor([4, 3, 5, 8, 2, 1, 0], dtype=torch.int32, device=device)
        vals = torch.rand((7,), dtype=dt, device=device)
        res = tensor.clone()
        decay = 0.5
        tn.scatter_ema_uint32_(res, 0, indices, decay, vals)
        gt = torch.scatter(tensor, 0, indices.long(), tensor[indices.long()] * decay + (1 - decay) * vals)
        torch.testing.assert_close(res, gt)
    # repeated index: the updates are applied one after the other (atomic), in some order
    x = torch.zeros((1,), device=device)
    tn.scatter_ema_uint32_(x, 0, torch.zeros((3,), dtype=torch.int32, device=device), 0.5, torch.ones((3,), device=device))
    torch.testing.assert_close(x, torch.tensor([0.875], device=device))
    with pytest.raises(Exception):
        t2 = torch.rand((5, 3), device=device)
        tn.scatter_ema_uint32_(t2, 0, torch.randint(0, 3, (5, 8), dtype=torch.int32, device=device), 0.5,
                               torch.rand((5, 8), device=device))


def test_stale_field_cache_is_caught_by_the_debug_check(tn, device):
    """A registered field written through `.data` keeps its cached vertex-major shadow (no version counter moves): the
    documented contract is invalidate_field_cache(); TETRANERF_HIP_CHECK_CACHES=1 turns the silent staleness into an error."""
    import torch

    cpp = tn.cpp
    field = torch.randn(64, 500, device=device)
    vi = torch.randint(0, 500, (1000, 4), dtype=torch.int32, device=device)
    bc = torch.rand(1000, 3, device=device) * 0.3
    cpp.register_field(field)
    try:
        a = cpp.interpolate_values(vi, bc, field).clone()
        field.data[3] += 1.0                                    # no version bump
        stale = cpp.interpolate_values(vi, bc, field)
        assert torch.equal(a, stale)                            # the documented hazard
        old = cpp._CHECK_CACHES
        cpp._CHECK_CACHES = True
        try:
            with pytest.raises(RuntimeError, match="STALE"):
                cpp.interpolate_values(vi, bc, field)
            cpp.invalidate_field_cache(field)
            fresh = cpp.interpolate_values(vi, bc, field)
            assert not torch.equal(a, fresh)
        finally:
            cpp._CHECK_CACHES = old
    finally:
        cpp.unregister_field(field)


@pytest.mark.parametrize("D,Fd", [(4, 64), (3, 7), (2, 1)])
def test_backward_rows_entry_point_equals_the_feature_major_one(tn, device, D, Fd):
    """tn_interpolate_values_backward_rows (the gradient in autograd's own [n, F] layout, straight through the C-ABI: no Python
    wrapper routes to it) against tn_interpolate_values_backward on the transposed copy the reference makes
    (py_binding.cpp:369): the same sums up to the order of the atomic additions."""
    import ctypes as C
    import importlib

    import torch

    lib = importlib.import_module("tetra-nerf_amd._lib").load()
    torch.manual_seed(D * 100 + Fd)
    V, n = 500, 6000
    vi = torch.randint(0, V, (n, D), dtype=torch.int32, device=device)
    vi[::17, 0] = -1                                   # empty slots (0xFFFFFFFF) are skipped
    bc = torch.rand(n, D - 1, device=device) / D
    g_rows = torch.randn(n, Fd, device=device)
    g_fm = g_rows.t().contiguous()
    out_rows = torch.full((Fd, V), float("nan"), device=device)
    out_fm = torch.full((Fd, V), float("nan"), device=device)
    st = torch.cuda.current_stream(device).cuda_stream
    assert lib.tn_interpolate_values_backward_rows(D, V, n, Fd, vi.data_ptr(), bc.data_ptr(), g_rows.data_ptr(), out_rows.data_ptr(), st) == 0
    assert lib.tn_interpolate_values_backward(D, V, n, Fd, vi.data_ptr(), bc.data_ptr(), g_fm.data_ptr(), out_fm.data_ptr(), st) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out_rows).all()) and bool(torch.isfinite(out_fm).all())      # fully written
    assert float((out_rows - out_fm).abs().max()) <= 2e-5 * max(1.0, float(out_fm.abs().max()))
    # and the definition: scatter-add of w_k * g over the vertices
    w = torch.cat([1.0 - bc.sum(-1, keepdim=True), bc], -1).double()
    want = torch.zeros(V, Fd, dtype=torch.float64, device=device)
    for k in range(D):
        ok = vi[:, k] >= 0
        want.index_add_(0, vi[ok, k].long(), w[ok, k:k + 1] * g_rows[ok].double())
    assert float((out_rows.double() - want.t()).abs().max()) <= 2e-5 * max(1.0, float(want.abs().max()))
