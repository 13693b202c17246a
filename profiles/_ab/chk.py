import importlib, sys, torch
sys.path.insert(0, '/root/repo')
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for v in (0, 256):
    tr.set_option("fill_overlap_blocks", v)
    out = tr.trace_rays(o, d, 512)
    n = out["num_visited_cells"]
    valid = torch.arange(512, device=dev)[None] < n[:, None]
    vc = out["visited_cells"]
    print(v, "tail ids all -1:", bool((vc[~valid] == -1).all()), "tail dist all 0:", bool((out["hit_distances"][~valid] == 0).all()), "nan anywhere:", bool(torch.isnan(out["barycentric_coordinates"]).any()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): tr.trace_rays(o, d, 512)
    e1.record(); torch.cuda.synchronize()
    print(v, e0.elapsed_time(e1)/5, "ms")
