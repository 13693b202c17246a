"""N trace_rays calls of a frame with given options (for rocprofv3 --kernel-trace --stats: in-schedule kernel durations)."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
npts, seed, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pts, cells = scenes.random_mesh(npts, seed)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
for kv in sys.argv[4:]:
    k, v = kv.split("="); tr.set_option(k, int(v))
for _ in range(n):
    out = tr.trace_rays(o, d, 512); del out
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    out = tr.trace_rays(o, d, 512); del out
e1.record(); torch.cuda.synchronize()
print(f"{sys.argv[4:]}: {e0.elapsed_time(e1) / n:.3f} ms per call")
