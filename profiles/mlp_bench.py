#!/usr/bin/env python3
"""Throughput of the fused fp32-MFMA MLP + heads (122,624 FLOP per sample) vs the PyTorch chain,
and the composite kernel, at the C3 size (4096 rays x 256 samples)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
FLOP = 2 * (64 * 128 + 128 * 128 * 2 + 128 * 1 + 155 * 128 + 128 * 3)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for R, S in ((4096, 256), (4096, 513), (16384, 256)):
    n = R * S
    feats_fm = torch.randn(64, n, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=-1)
    w = render.mlp_weights(mlp)
    ms = timeit(lambda: tn.cpp.mlp_forward(feats_fm, dirs, w, S))
    tn.cpp.mlp_set_mode("bf16x3")
    ms3 = timeit(lambda: tn.cpp.mlp_forward(feats_fm, dirs, w, S))
    s3, c3 = tn.cpp.mlp_forward(feats_fm, dirs, w, S)
    # the production variant: barycentric gather fused in (no [64,n] buffer)
    V = 45000
    vi = torch.randint(0, V, (n, 4), dtype=torch.int32, device=dev); bcw = torch.rand(n, 3, device=dev) / 4
    field = torch.randn(64, V, device=dev)
    msg3 = timeit(lambda: tn.cpp.mlp_forward_gather(vi, bcw, field, dirs, w, S))
    tn.cpp.mlp_set_mode("fp32")
    msg = timeit(lambda: tn.cpp.mlp_forward_gather(vi, bcw, field, dirs, w, S))
    print(f"MLP n={n} gather-fused: fp32 {msg:.3f} ms = {n*FLOP/msg/1e9:.1f} TFLOP/s, bf16x3 {msg3:.3f} ms = {n*FLOP/msg3/1e9:.1f} fp32-equivalent TFLOP/s")
    del vi, bcw, field
    feats = feats_fm.t().contiguous(); dd = dirs[:, None, :].expand(R, S, 3).reshape(n, 3)
    with torch.no_grad():
        ms_t = timeit(lambda: mlp(feats, dd), 5)
    sigma, rgb = tn.cpp.mlp_forward(feats_fm, dirs, w, S)
    edges = (torch.rand(R, 1, device=dev) + torch.cumsum(torch.rand(R, S + 1, device=dev) * 0.01, -1)).contiguous()
    ms_c = timeit(lambda: tn.cpp.composite(sigma.view(R, S), rgb.view(R, S, 3), edges))
    print(f"MLP n={n} bf16x3 mode: {ms3:.3f} ms = {n*FLOP/ms3/1e9:.1f} fp32-equivalent TFLOP/s ({ms/ms3:.2f}x the fp32 MFMA kernel); "
          f"max |sigma diff| {float((s3 - sigma).abs().max()):.2e}, max |rgb diff| {float((c3 - rgb).abs().max()):.2e} vs the fp32 kernel")
    print(f"MLP n={n} ({R}x{S}): fused {ms:.3f} ms = {n*FLOP/ms/1e9:.1f} TFLOP/s ({n*FLOP/ms/1e9/157.3*100:.1f}% of 157.3 fp32 MFMA peak); "
          f"torch {ms_t:.3f} ms = {n*FLOP/ms_t/1e9:.1f} TFLOP/s; composite {ms_c*1e3:.1f} us ({n*20/ms_c/1e6:.0f} GB/s of 20 B/sample)")
