import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, "/root/repo")
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
pts, cells = scenes.random_mesh(15000, 0)
o, d = bench.frame_rays(scenes, 0, 800, 800)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for rep in range(2):
    for fb in (0, 512, 384, 256, 192, 128, 64):
        for rw in (4096,):
            tr.set_option("fill_blocks", fb); tr.set_option("rewalk_min", rw)
            for _ in range(2): out = tr.trace_rays(o, d, 512); del out
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
            e0.record()
            for _ in range(5): out = tr.trace_rays(o, d, 512); del out
            e1.record(); torch.cuda.synchronize()
            print(f"fill_blocks={fb} rewalk_min={rw}: {e0.elapsed_time(e1)/5:.3f} ms")
