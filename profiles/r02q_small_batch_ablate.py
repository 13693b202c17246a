import importlib, sys, torch
sys.path.insert(0, "/root/repo")
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
pts, cells = scenes.random_mesh(45000, 2)
tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = scenes.outside_in_rays(4096, 1); o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
def call():
    x = tr.trace_rays(o, d, M); del x
for gd, what in ((0, "everything"), (1, "traversal only (no hits kept: sort/pairing of empty lists, rows still written)"), (2, "no sort")):
    tr.set_option("gdebug", gd)
    print(f"C4 4096 rays outside-in, gdebug {gd} ({what}): {timed(call):.3f} ms", flush=True)
tr.set_option("gdebug", 8); call(); torch.cuda.synchronize(); tr.set_option("gdebug", 0)
