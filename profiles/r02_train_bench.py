#!/usr/bin/env python3
"""fwd + bwd of the fused MLP node at the C4 training-batch size (4096 rays x 256 coarse + 513 fine samples) vs PyTorch
autograd of the same statement: ms and fp32-MFMA fraction (122,624 FLOP per sample forward; the backward does the
same again twice -- dX chain and weight gradients -- plus the recompute)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0"); torch.manual_seed(0)
FLOP = 2 * (64 * 128 + 128 * 128 * 2 + 128 * 1 + 155 * 128 + 128 * 3)
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
V = 45000
mlp = render.TetraMLP().to(dev); w = render.mlp_weights(mlp)
field = torch.randn(64, V, device=dev, requires_grad=True)
for R, S in ((4096, 256), (4096, 513)):
    n = R * S
    vi = torch.randint(0, V, (n, 4), dtype=torch.int32, device=dev); bc = torch.rand(n, 3, device=dev) / 4
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    gs, gc = torch.randn(n, device=dev), torch.randn(n, 3, device=dev)
    def fused():
        field.grad = None; mlp.zero_grad()
        s, c = render._FusedMlpFunction.apply(vi, bc, field, dirs, S, None, *w)
        ((s * gs).sum() + (c * gc).sum()).backward()
    wd = [x.detach() for x in w]
    def fwd_only():
        with torch.no_grad(): return tn.cpp.mlp_forward_gather_train(vi, bc, field.detach(), dirs, wd, S)
    s0, c0, saved = fwd_only()
    def bwd_only():
        tn.cpp.mlp_backward(saved, vi, bc, field.detach(), dirs, wd, s0, c0, gs, gc)
    def autograd():
        field.grad = None; mlp.zero_grad()
        feats = tn.interpolate_values(vi, bc, field)
        s, c = mlp(feats, dirs[:, None, :].expand(-1, S, -1).reshape(n, 3))
        ((s[:, 0] * gs).sum() + (c * gc).sum()).backward()
    ms_f, ms_b, ms_fb = timeit(fwd_only), timeit(bwd_only), timeit(fused)
    ms_t = timeit(autograd, 3)
    useful = 3 * n * FLOP      # forward + dX + dW
    print(f"{R}x{S}: fused fwd {ms_f:.2f} ms ({n*FLOP/ms_f/1e9:.1f} TFLOP/s), bwd {ms_b:.2f} ms, fwd+bwd {ms_fb:.2f} ms = "
          f"{useful/ms_fb/1e9:.1f} TFLOP/s useful = {useful/ms_fb/1e9/157.3*100:.0f} % of the fp32 MFMA peak "
          f"; PyTorch autograd {ms_t:.2f} ms ({ms_t/ms_fb:.1f}x)", flush=True)
