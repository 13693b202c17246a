import importlib, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from oracle import tn_oracle
tn_oracle.build()
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(5000, 21)
ot = tn_oracle.OracleTracer(use_bvh=True); ot.load_tetrahedra(pts, cells)
tr = tn.TetrahedraTracer(dev); tr.set_option("walk", 1); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.pinhole_rays(256, 256, eye=(0.5, 2.3, 0.6), lookat=(0.5, 0.5, 0.5))
want = ot.trace_rays(o, d, 256)
got = tr.trace_rays(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 256)
print(tr.trace_stats(), tr.flag_reasons())
for k in ("num_visited_cells", "visited_cells", "hit_distances"):
    g = got[k].cpu().numpy(); w = np.ascontiguousarray(want[k])
    bad = np.nonzero((g.view(np.uint32) != w.view(np.uint32)).reshape(len(o), -1).any(1))[0]
    print(k, "bad rays", len(bad), bad[:10])
    if len(bad) and k == "visited_cells":
        r = bad[0]; n = want["num_visited_cells"][r]
        slots = np.nonzero(g[r] != w[r])[0]
        print(" ray", r, "n", n, "got n", int(got["num_visited_cells"][r]), "bad slots", slots[:10], "..", slots[-5:], "got", g[r][slots[:5]], "want", w[r][slots[:5]])
