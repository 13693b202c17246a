#!/usr/bin/env python3
"""Config C4 (300k-tet stand-in, 4096-ray training batches): per-call time of the drop-in ops."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
M, S = 512, 256
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (o, d) in (("outside_in", scenes.outside_in_rays(4096, 1)), ("inside_out", scenes.inside_out_rays(4096, 2))):
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    for walk in (1, 0, 2):   # 2 = adjacency walk forced at this size
        tr.set_option("walk", walk)
        us = timeit(lambda: tr.trace_rays(o, d, M))
        out = tr.trace_rays(o, d, M)
        print(f"C4 {name} R=4096 M={M} walk={walk}: trace_rays {us:.1f} us, segs/ray {float(out['num_visited_cells'].float().mean()):.1f}, stats {tr.trace_stats()}")
    nv = out["num_visited_cells"]
    near = out["hit_distances"][:, 0, 0]
    far = torch.gather(out["hit_distances"][:, :, 1], 1, (nv[:, None].long() - 1).clamp_min(0))[:, 0]
    ts = torch.linspace(0, 1, S, device=dev)[None]
    samples = (near[:, None] * (1 - ts) + far[:, None] * ts).contiguous()
    f = lambda: tr.find_visited_cells(nv, out["visited_cells"], out["barycentric_coordinates"], out["hit_distances"], out["vertex_indices"], samples)
    us = timeit(f); inter = f()
    print(f"C4 {name} find_visited_cells S={S}: {us:.1f} us ({4096*S*37/us/1e3:.1f} GB/s of 37 B/sample)")
    field = torch.randn((64, len(pts)), device=dev, requires_grad=True)
    g = lambda: tn.cpp.interpolate_values(inter["vertex_indices"], inter["barycentric_coordinates"], field)
    us = timeit(g); feats = g()
    print(f"C4 {name} interpolate_values n={4096*S}: {us:.1f} us ({4096*S*284/us/1e3:.1f} GB/s of 284 B/sample)")
    go = torch.randn(feats.shape, device=dev)
    h = lambda: tn.cpp.interpolate_values_backward(inter["vertex_indices"], inter["barycentric_coordinates"], field, go)
    us = timeit(h)
    print(f"C4 {name} interpolate_values_backward (grad rows [n,64], autograd layout): {us:.1f} us ({4096*S*(256+28)/us/1e3:.1f} GB/s of 284 B/sample read)")
    # the reference's native layout through the C-ABI directly (the Python surface takes contiguous [..., 64])
    import ctypes as C
    lib = importlib.import_module("tetra-nerf_amd._lib").load()
    go_fm = go.reshape(-1, 64).t().contiguous()
    gout = torch.empty((64, len(pts)), device=dev)
    vi_, bc_ = inter["vertex_indices"], inter["barycentric_coordinates"]
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    h2 = lambda: lib.tn_interpolate_values_backward(4, len(pts), 4096 * S, 64, C.c_void_p(vi_.data_ptr()), C.c_void_p(bc_.data_ptr()),
                                                    C.c_void_p(go_fm.data_ptr()), C.c_void_p(gout.data_ptr()), st)
    us = timeit(h2)
    assert torch.allclose(gout, h(), rtol=1e-4, atol=1e-4)
    print(f"C4 {name} interpolate_values_backward (grad [64,n], reference layout): {us:.1f} us")
