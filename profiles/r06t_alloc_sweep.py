"""In-process option sweep repeated over fresh ALLOCATIONS of the output rows (torch.cuda.empty_cache() between: same virtual
addresses, new physical pages; the same fill kernel moves by 25 % with that placement, profiles/r06s_placement.txt).
Usage: python profiles/r06t_alloc_sweep.py allocs config name:opt=v,... ..."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench
tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
allocs, name = int(sys.argv[1]), sys.argv[2]
variants = []
for a in sys.argv[3:]:
    n, _, spec = a.partition(":")
    variants.append((n, [kv.split("=") for kv in spec.split(",") if kv]))
DEFAULTS = {"cert_ends": 2, "verify_stride": 256, "fill_blocks": -1, "spec_blocks": 512, "spec_fill": 0, "spec_k0": 0, "writer_blocks": 0, "hull_flat": 1}
CFG = {"C2": (15000, 0, 512), "C4": (45000, 2, 512), "C5": (150000, 3, 512)}
npts, seed, M = CFG[name]
pts, cells = scenes.random_mesh(npts, seed)
tr = tn.TetrahedraTracer(dev)
tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
o, d = bench.frame_rays(scenes, 0, 800, 800) if name != "C5" else scenes.outside_in_rays(1 << 20, 4)
o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
print(f"== {name}: {len(cells)} tets, {len(o)} rays; ms per call (median of 5 x 3 calls), per fresh allocation")
tot = {n: [] for n, _ in variants}
for a in range(allocs):
    torch.cuda.empty_cache()
    res = {n: [] for n, _ in variants}
    for rep in range(5):
        for n, opts in variants:
            for k, v in DEFAULTS.items():
                tr.set_option(k, v)
            for k, v in opts:
                tr.set_option(k, int(v))
            tr.trace_rays(o, d, M)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                tr.trace_rays(o, d, M)
            e1.record()
            torch.cuda.synchronize()
            res[n].append(e0.elapsed_time(e1) / 3)
    line = []
    for n, _ in variants:
        med = sorted(res[n])[2]
        tot[n].append(med)
        line.append(f"{n} {med:.3f}")
    print(f"alloc {a}: " + "  ".join(line), flush=True)
base = tot[variants[0][0]]
print("mean : " + "  ".join(f"{n} {sum(v) / len(v):.3f} ({100 * (sum(v) / sum(base) - 1):+.1f} %)" for n, v in tot.items()))
