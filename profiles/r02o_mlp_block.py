#!/usr/bin/env python3
"""fp32 forward MLP kernel: one 8-wave block per CU (512 threads) vs two 4-wave blocks per CU (256 threads, head layer
staged in two halves).  16.7 M samples, all four kernel variants; TFLOP/s of the 122,624 (82,176 density-only) FLOP per
sample; results compared bit for bit; then the render frame with either shape."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
render = importlib.import_module("tetra-nerf_amd.render")
dev = torch.device("cuda:0")
torch.manual_seed(0)
mlp = render.TetraMLP().to(dev); w = render.mlp_weights(mlp)
S, n = 256, 65536 * 256
feats = torch.randn(64, n, device=dev); dirs = torch.nn.functional.normalize(torch.randn(n // S, 3, device=dev), dim=-1)
V = 15000
field = torch.randn(64, V, device=dev)
vi = torch.randint(0, V, (n, 4), device=dev, dtype=torch.int32); bc = torch.rand(n, 3, device=dev) / 3
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
FULL, DENS = 122624, 2 * (64 * 128 + 128 * 128 * 2 + 128)
cases = (("feats, full", lambda: tn.cpp.mlp_forward(feats, dirs, w, S), FULL),
         ("gather, full", lambda: tn.cpp.mlp_forward_gather(vi, bc, field, dirs, w, S), FULL),
         ("gather, density only", lambda: tn.cpp.mlp_forward_gather(vi, bc, field, None, w, S), DENS))
for name, fn, flop in cases:
    res, outs = {}, {}
    for blk in (512, 256, 512, 256):
        tn.cpp.mlp_set_block(blk)
        o = fn(); outs[blk] = o if isinstance(o, torch.Tensor) else o[1]
        res.setdefault(blk, []).append(timed(fn))
    assert torch.equal(outs[256], outs[512]), name
    a, b = min(res[512]), min(res[256])
    print(f"{name}: 8-wave blocks {a:.2f} ms ({n*flop/a/1e9:.1f} TFLOP/s = {n*flop/a/1e9/157.3*100:.1f} %), 4-wave x2 {b:.2f} ms ({n*flop/b/1e9:.1f} TFLOP/s = {n*flop/b/1e9/157.3*100:.1f} %)", flush=True)
del feats, vi, bc
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
fld = ((torch.rand(64, len(pts), device=dev) * 2 - 1) * 1e-4); fld[1:4] = torch.rand(3, len(pts), device=dev) * 2 - 1
o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for name, (s_c, s_f, biased) in (("coarse-256", (256, 0, False)), ("tetra-nerf-original", (256, 256, False))):
    rd = render.TetraRenderer(tr, fld, mlp, s_c, 512, fused=True, num_fine_samples=s_f, biased=biased)
    def frame():
        for s in range(0, o.shape[0], 65536): rd.render(o[s:s + 65536], d[s:s + 65536])
    res = {}
    for blk in (512, 256, 512, 256):
        tn.cpp.mlp_set_block(blk)
        frame(); torch.cuda.synchronize(); t0 = time.perf_counter(); frame(); frame(); torch.cuda.synchronize()
        res.setdefault(blk, []).append((time.perf_counter() - t0) / 2)
    print(f"frame {name}: 8-wave blocks {min(res[512])*1e3:.1f} ms ({o.shape[0]/min(res[512])/1e6:.2f} M rays/s), 4-wave x2 {min(res[256])*1e3:.1f} ms ({o.shape[0]/min(res[256])/1e6:.2f} M rays/s)", flush=True)
tn.cpp.mlp_set_block(0)
