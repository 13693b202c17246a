#!/usr/bin/env python3
"""The bench's MLP-kernel leg (16.7 M samples, fp32 and bf16x3 forward without the gather) with the package of the tree named on the
command line: python profiles/r05p_x3_tree.py <root containing tetra-nerf_amd/>   (A/B of round 4's library against this tree's)."""
import importlib, sys
import torch
root = sys.argv[1]
sys.path.insert(0, root)
tn = importlib.import_module("tetra-nerf_amd"); render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev)
samples = 256
n = 65536 * samples
feats = torch.randn(64, n, device=dev)
dirs = torch.nn.functional.normalize(torch.randn(n // samples, 3, device=dev), dim=-1)
w = render.mlp_weights(mlp)
def timed(**kw):
    for _ in range(3): tn.cpp.mlp_forward(feats, dirs, w, samples, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): tn.cpp.mlp_forward(feats, dirs, w, samples, **kw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
print(f"{root}: fp32 {timed():.3f} ms   bf16x3 {timed(mode='bf16x3'):.3f} ms   fp32 {timed():.3f} ms   bf16x3 {timed(mode='bf16x3'):.3f} ms")
