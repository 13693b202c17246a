"""In-process interleaved sweep of trace_rays OPTIONS on one tracer, one set of buffers (round 6: separate processes on one box
differed by up to 20 % for the SAME kernel -- physical placement of the 17 GB of rows -- far above what is being measured).
Per config: every variant is warmed up, then `rounds` x (for each variant: 3 timed calls).  Prints median / min ms per variant,
the serialised breakdown of the first variant, path statistics, and checks that all variants write identical outputs.
Usage: python profiles/r06e_sweep.py rounds configs name:opt=v,opt=v [name:... ...]"""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402

tn = importlib.import_module("tetra-nerf_amd")
scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
rounds, which = int(sys.argv[1]), sys.argv[2].split(",")
variants = []
for a in sys.argv[3:]:
    n, _, spec = a.partition(":")
    variants.append((n, [kv.split("=") for kv in spec.split(",") if kv]))
DEFAULTS = {"cert_ends": 2, "verify_stride": 256}
CFG = {"C2": (15000, 0, 512), "C4": (45000, 2, 512), "C5": (150000, 3, 512)}


def checksum(out):
    chk = 0
    for k in sorted(out):
        chk = (chk * 1000003 + int(out[k].view(torch.int32).to(torch.int64).sum())) & ((1 << 62) - 1)
    return chk


for name in which:
    npts, seed, M = CFG[name]
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if name != "C5" else scenes.outside_in_rays(1 << 20, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

    def apply(opts):
        for k, v in DEFAULTS.items():
            tr.set_option(k, v)
        for k, v in opts:
            tr.set_option(k, int(v))

    info = {}
    for n, opts in variants:
        apply(opts)
        out = tr.trace_rays(o, d, M)
        info[n] = {"chk": checksum(out), "paths": tr.trace_stats(), "reasons": tr.flag_reasons(), "ms": []}
        del out
        info[n]["breakdown"] = bench.trace_breakdown(tr, o, d, M)
    for _ in range(rounds):
        for n, opts in variants:
            apply(opts)
            tr.trace_rays(o, d, M)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                tr.trace_rays(o, d, M)
            e1.record()
            torch.cuda.synchronize()
            info[n]["ms"].append(e0.elapsed_time(e1) / 3)
    base = sorted(info[variants[0][0]]["ms"])
    bmed = base[len(base) // 2]
    print(f"== {name}: {len(cells)} tets, {len(o)} rays")
    for n, _ in variants:
        ms = sorted(info[n]["ms"])
        med = ms[len(ms) // 2]
        bd = info[n]["breakdown"]
        print(f"  {n:14s} median {med:7.3f}  min {ms[0]:7.3f}  max {ms[-1]:7.3f}  {100 * (med / bmed - 1):+5.1f} %  identical: {info[n]['chk'] == info[variants[0][0]]['chk']}"
              f"  general {info[n]['paths']['general']} literal {info[n]['reasons'].get(13, 0)} | "
              + " ".join(f"{k[:7]} {v:.3f}" for k, v in bd.items()))
    del tr
    torch.cuda.empty_cache()
