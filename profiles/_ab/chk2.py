import importlib, sys, torch
sys.path.insert(0, '/root/repo')
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for opts in ({}, {"spec_fill": 0}, {"verify_stride": 0}, {"spec_fill": 0, "verify_stride": 0}):
    for k, v in {"spec_fill": 1, "verify_stride": 256, **opts}.items(): tr.set_option(k, v)
    out = tr.trace_rays(o, d, 512)
    n = out["num_visited_cells"]
    vc = out["visited_cells"]
    slot = torch.arange(512, device=dev)[None].expand_as(vc)
    bad = (slot >= n[:, None]) & (vc != -1)
    rows = bad.any(1)
    print(opts, "bad rows", int(rows.sum()), "of", len(n))
    if rows.any():
        r = int(torch.nonzero(rows)[0])
        b = torch.nonzero(bad[r])[:, 0]
        print("  first bad row", r, "n", int(n[r]), "bad slots", int(b.min()), "..", int(b.max()), "count", len(b))
        r = int(torch.nonzero(rows)[-1])
        b = torch.nonzero(bad[r])[:, 0]
        print("  last bad row", r, "n", int(n[r]), "bad slots", int(b.min()), "..", int(b.max()), "count", len(b))
