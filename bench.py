#!/usr/bin/env python3
"""Benchmark of the ray -> tetrahedra hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: re-executes itself under
                                                           torch.distributed.run with N ranks, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W          (the driver's form; WORLD_SIZE must equal --gpus)

Environment (tests on a one-GPU box): TETRANERF_BENCH_BACKEND=gloo + TETRANERF_BENCH_ONE_DEVICE=1 put all N ranks on
cuda:0 with gloo collectives (RCCL refuses two ranks on one device), so that the N > 1 code runs on device tensors.

A "step" is one pass of the hot path over one batch of synthetic input: trace_rays over one
800x800 frame (640,000 rays, M = 512) through the 100k-tetrahedra stand-in of BASELINE.json's
configs[1] ("lego sparse 100k"; dataset unavailable).  Inputs are resident in
HBM before the timed region.  Multi-GPU: rays are independent units -- every rank traces its
own frame (camera rotated per rank), no data-path collective; scaling = weak.

Prints ONE JSON line on rank 0 (see the driver contract): metric/value = whole-job ray-tet
intersections per second, plus `roofline` (dense-output HBM bytes of the dominant kernel over
its HIP-event duration), `cpu_baseline` (the CPU oracle's BVH path, timing build -O3
-march=native, on a bounded sample of the same rays, all host cores), and -- single-GPU runs --
`configs`: the other sizes BASELINE.json names (C4 300k-tet frame and both 4096-ray training
batches, C5 1M-tet 2^20-ray stress), each with ms, rays/s, intersections/s and its roofline
fraction, and `C4_train_4096`: one training iteration (forward + backward + SGD) of both shipped
configurations on the C4 batch, fused HIP nodes beside PyTorch autograd.  `sharded_render` (every N): the 800x800 frame rendered in ray shards over the N ranks
and all-gathered as ONE RCCL collective (north_star's multi-GPU flow), with the time of the
collective.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X spec (6290 GB/s measured copy ceiling), MI355X_MICROARCH.md:35


def mesh_sha256(pts, cells) -> str:
    return importlib.import_module("tetra-nerf_amd.scenes").mesh_sha256(pts, cells)


def frame_rays(scenes, rank: int, width: int, height: int):
    ang = rank * math.pi / 4.0
    c = np.array([0.5, 0.5, 0.5], np.float32)
    eye = c + 2.0 * np.array([math.sin(ang), math.cos(ang), 0.0], np.float32)
    return scenes.pinhole_rays(width, height, eye=tuple(eye), lookat=tuple(c), up=(0.0, 0.0, 1.0), fov_y=45.0)


def pmc_traffic(args, R, M):
    """HBM bytes per trace_rays launch from the committed rocprofv3 PMC passes of this same
    command (profiles/traffic.json: WRITE_SIZE + 2 x FETCH_SIZE summed over the kernels of one
    launch, per MI355X_MICROARCH.md's gfx950 FETCH_SIZE correction).  Counters cannot be read from
    inside the timed process; null when the workload differs from the profiled one."""
    f = ROOT / "profiles" / "traffic.json"
    if not f.exists():
        return None
    t = json.loads(f.read_text())
    key = {"mesh_points": args.mesh_points, "mesh_seed": args.mesh_seed, "rays": R, "M": M}
    return t.get("hbm_bytes_per_launch") if t.get("workload") == key else None


def render_leg(tn, tracer, num_vertices, o, d, M, dev, samples=256, chunk=65536, reps=2):
    """Second half of BASELINE.json's metric: rendered rays/s for the 800x800 frame through the whole
    forward path (trace -> uniform samples -> match -> gather -> fused fp32-MFMA MLP -> composite),
    coarse pass of the `tetra-nerf-original` evaluation config, nerfstudio-style chunks."""
    render = importlib.import_module("tetra-nerf_amd.render")
    torch.manual_seed(0)
    mlp = render.TetraMLP().to(dev)
    field = ((torch.rand(64, num_vertices, device=dev) * 2 - 1) * 1e-4)
    field[1:4] = torch.rand(3, num_vertices, device=dev) * 2 - 1
    rd = render.TetraRenderer(tracer, field, mlp, samples, M, fused=True)
    R = o.shape[0]

    def frame(rd=rd):
        hit = torch.zeros((), dtype=torch.int64, device=dev)
        for s in range(0, R, chunk):
            out = rd.render(o[s:s + chunk], d[s:s + chunk])
            hit += out["ray_mask"].sum()          # stays on the device: the render path itself never synchronises (round 5)
        return hit

    def timed(rd):
        frame(rd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            frame(rd)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    hit = int(frame())
    dt = timed(rd)
    # the two shipped evaluation configs with the PDF fine pass (registration.py:55-57, model.py:78-80).  Per config:
    #   "fp32"               the DEFAULT renderer = everything after the trace as ONE persistent launch (tn_render_rays)
    #   "fp32_kernel_chain"  the same stages as separate kernels (sampler, matcher, gather + MLP, composite per pass): bit-identical frame
    #   "bf16x3"             the one launch with the split-operand bf16 MFMA arithmetic in its MLP phases (opt-in; round 6)
    #   "bf16x3_kernel_chain" the kernel chain in that arithmetic: bit-identical frame
    # none of them synchronises with the host (device-side compaction of the hitting rays)
    full = {}
    configs = (("coarse-256", (samples, 0, False)), ("tetra-nerf-original", (256, 256, False)), ("tetra-nerf", (128, 128, True)))
    for name, (s_c, s_f, biased) in configs:
        full[name] = {"samples_per_ray": f"{s_c} coarse" + (f" (density only) + {s_c + s_f + 1} fine" if s_f else "")}
        for key, kw in (("fp32", {}), ("fp32_kernel_chain", dict(fused_pass=False)), ("bf16x3", dict(mlp_mode="bf16x3")),
                        ("bf16x3_kernel_chain", dict(mlp_mode="bf16x3", fused_pass=False))):
            dtf = timed(render.TetraRenderer(tracer, field, mlp, s_c, M, fused=True, num_fine_samples=s_f, biased=biased, **kw))
            full[name][key] = {"rendered_rays_per_s": R / dtf, "ms_per_frame": dtf * 1e3}
        full[name]["one_launch_over_chain"] = full[name]["fp32"]["ms_per_frame"] / full[name]["fp32_kernel_chain"]["ms_per_frame"]
        full[name]["one_launch_over_chain_bf16x3"] = full[name]["bf16x3"]["ms_per_frame"] / full[name]["bf16x3_kernel_chain"]["ms_per_frame"]
    # MLP kernel alone on one chunk worth of samples of hitting rays (MFMA roofline)
    n = min(hit, chunk) * samples
    feats = torch.randn(64, n, device=dev)
    dirs = torch.nn.functional.normalize(torch.randn(n // samples, 3, device=dev), dim=-1)
    w = render.mlp_weights(mlp)
    for _ in range(2):
        tn.cpp.mlp_forward(feats, dirs, w, samples)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        tn.cpp.mlp_forward(feats, dirs, w, samples)
    e1.record()
    torch.cuda.synchronize()
    mlp_ms = e0.elapsed_time(e1) / 5
    flop = 2 * (64 * 128 + 128 * 128 * 2 + 128 + 155 * 128 + 128 * 3)
    tf = n * flop / (mlp_ms * 1e-3) / 1e12
    for _ in range(2):
        tn.cpp.mlp_forward(feats, dirs, w, samples, mode="bf16x3")
    e0.record()
    for _ in range(5):
        tn.cpp.mlp_forward(feats, dirs, w, samples, mode="bf16x3")
    e1.record()
    torch.cuda.synchronize()
    x3_ms = e0.elapsed_time(e1) / 5
    return {"rendered_rays_per_s": R / dt, "ms_per_frame": dt * 1e3, "rays": R, "hitting_rays": hit,
            "samples_per_ray": samples, "pass": "coarse only (uniform samples), 65,536-ray chunks: trace_rays + tn_compact_hits + ONE persistent launch (tn_render_rays), no host synchronisation",
            "eval_configs": full,
            "roofline_mlp": {"bound": "mfma", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                             "dtype": "f32 (v_mfma_f32_32x32x2_f32)", "samples": n, "kernel_ms": mlp_ms,
                             "flop_per_sample": flop},
            "mlp_bf16x3": {"kernel_ms": x3_ms, "fp32_equivalent_TFLOPs": n * flop / (x3_ms * 1e-3) / 1e12,
                           "note": "optional mode: operands split into 3 bf16 pieces, 6 bf16 MFMAs per fp32 product, "
                                   "fp32 accumulate; same 1e-5 parity tests; not the default"}}


def box_write_ceiling(tn, out, reps=5):
    """The write rate THIS box sustains, measured in this process: tn_fill_rows (k_fill_linear: one linear stream per array,
    one 16-byte store per thread -- since round 6 the one fill whose rate does not depend on where the driver put the
    pages, 7.0 - 7.15 TB/s, profiles/r06s_flat_fill.txt) ALONE over every slot of the dense rows of one bench launch.
    88 % of a trace_rays call's bytes are these constant tails, so this is the ceiling of the call on this box."""
    cpp = tn.cpp
    args = (out["visited_cells"], out["barycentric_coordinates"], out["hit_distances"], out["vertex_indices"])
    R, M = out["visited_cells"].shape
    cpp.fill_rows(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        cpp.fill_rows(*args)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nbytes = R * M * 52
    return {"GBps": nbytes / (ms * 1e-3) / 1e9, "bytes": nbytes, "ms": ms,
            "kernel": "k_fill_linear (tn_fill_rows) alone, all slots of the bench launch's rows, same process"}


def trace_breakdown(tracer, o, d, M):
    """Per-kernel HIP-event milliseconds of ONE extra launch with the kernels serialised on the caller's stream (option
    "timing"); the normal schedule overlaps them on four streams, so the parts do not add up to `ms`."""
    tracer.set_option("timing", 1)
    try:
        out = tracer.trace_rays(o, d, M)
        del out
        bd = tracer.trace_timings()
    finally:
        tracer.set_option("timing", 0)
    bd["sum_serialised"] = sum(bd.values())
    return bd


def cross_check(tracer):
    """blind sample (every verify_stride-th certified ray: 1 in 256 by default) + risk classes (every certified ray inside the
    wide band of a guard)"""
    return tracer.cross_check()


def trace_leg(tracer, o, d, M, reps, box_ceiling_gbps=None):
    """ms per trace_rays call (HIP events on the launch stream), intersections, path statistics."""
    def run():
        out = tracer.trace_rays(o, d, M)
        k = out["num_visited_cells"]
        del out
        return k
    inter = int(run().sum())
    stats = tracer.trace_stats()
    stats["reasons"] = {str(k): v for k, v in tracer.flag_reasons().items()}
    xc = cross_check(tracer)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    R = o.shape[0]
    gbs = R * (28 + 52 * M) / (ms * 1e-3) / 1e9
    roof = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    if box_ceiling_gbps:
        roof["frac_of_box_write_ceiling"] = gbs / box_ceiling_gbps
    leg = {"rays": R, "ms": ms, "rays_per_s": R / (ms * 1e-3), "intersections_per_s": inter / (ms * 1e-3),
           "intersections": inter, "roofline": roof, "paths": stats,
           "certification_cross_check": xc}
    if stats.get("walk", 0) and R >= 12288:
        leg["breakdown_ms_serialised"] = trace_breakdown(tracer, o, d, M)
    return leg


def config_legs(tn, scenes, dev, M, box_ceiling_gbps=None):
    """The other BASELINE.json sizes: C4 = 45,000 points seed 2 (~302k tets): the 800x800 frame and the
    two 4096-ray training batches; C5 = 150,000 points seed 3 (~1.01M tets): 2^20 outside-in rays seed 4."""
    out = {}
    for cfg, npts, seed in (("C4", 45000, 2), ("C5", 150000, 3)):
        pts, cells = scenes.random_mesh(npts, seed)
        tr = tn.TetrahedraTracer(dev)
        x_dev, c_dev = torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)
        loads = {}
        for what, opt in (("host_build", 0), ("device_build_first_call", 1), ("device_build", 1)):
            tr.set_option("gpu_build", opt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tr.load_tetrahedra(x_dev, c_dev)
            torch.cuda.synchronize()
            loads[what] = time.perf_counter() - t0
        load_s = loads["device_build"]
        sets = ((("C4_frame_800x800", frame_rays(scenes, 0, 800, 800), 5),
                 ("C4_batch_4096_outside_in", scenes.outside_in_rays(4096, 1), 20),
                 ("C4_batch_4096_inside_out", scenes.inside_out_rays(4096, 2), 20)) if cfg == "C4" else
                (("C5_2^20_outside_in", scenes.outside_in_rays(1 << 20, 4), 3),))
        for name, (o, d), reps in sets:
            leg = trace_leg(tr, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), M, reps, box_ceiling_gbps)
            leg.update(tets=int(len(cells)), mesh_sha256=mesh_sha256(pts, cells), load_tetrahedra_s=load_s,
                       load_tetrahedra_host_build_s=loads["host_build"])
            out[name] = leg
        if cfg == "C4":
            out["C4_train_4096"] = train_leg(tn, tr, len(pts), scenes, M, dev)
            out["C4_ops"] = ops_leg(tn, tr, pts, scenes, M, dev)
        del tr
        torch.cuda.empty_cache()
    return out


def start_c6_mesh(n_points=1_000_000, seed=7):
    """C6 = the reference's own upper size (tetranerf/scripts/triangulate.py:15 keeps up to 1,000,000 points): the Delaunay of
    1M uniform points = 6.7M tets.  Qhull needs 60-110 s on one host core, so it runs in a child process from the start of
    the bench, beside every other leg; `finish_c6_mesh` collects the cells."""
    import subprocess
    import tempfile

    f = Path(tempfile.gettempdir()) / f"tetranerf_bench_delaunay_{n_points}_{seed}_{os.getpid()}.npy"
    code = ("import numpy as np, sys; from scipy.spatial import Delaunay; "
            f"p = np.random.default_rng({seed}).random(({n_points}, 3)).astype(np.float32); "
            "np.save(sys.argv[1], np.ascontiguousarray(Delaunay(p.astype(np.float64)).simplices.astype(np.int32)))")
    t0 = time.perf_counter()
    proc = subprocess.Popen([sys.executable, "-c", code, str(f)], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return {"proc": proc, "file": f, "n": n_points, "seed": seed, "t0": t0}


def finish_c6_mesh(h, timeout_s=400.0):
    try:
        rc = h["proc"].wait(timeout=timeout_s)
    except Exception:
        h["proc"].kill()
        return None
    if rc != 0 or not h["file"].exists():
        return None
    cells = np.load(h["file"])
    try:
        h["file"].unlink()
    except OSError:
        pass
    pts = np.random.default_rng(h["seed"]).random((h["n"], 3)).astype(np.float32)
    return pts, cells, time.perf_counter() - h["t0"]


def c6_legs(tn, scenes, dev, mesh, box_ceiling_gbps=None, M=1024):
    """C6_ref_scale: 1M points / 6.7M tets (walk tables of 0.9 GB: 3.5x the 256 MiB Infinity Cache); rays cross 500-650 faces,
    so M = 1024.  A 512x400 frame (204,800 rays) and one 4096-ray outside-in batch, each with its dense-row roofline, and the
    time of load_tetrahedra (every structure built on the device)."""
    pts, cells, qhull_s = mesh
    tr = tn.TetrahedraTracer(dev)
    x_dev, c_dev = torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)
    load_s = []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.load_tetrahedra(x_dev, c_dev)
        torch.cuda.synchronize()
        load_s.append(time.perf_counter() - t0)
    out = {}
    for name, (o, d), reps in (("C6_frame_512x400", frame_rays(scenes, 0, 512, 400), 3),
                                ("C6_batch_4096_outside_in", scenes.outside_in_rays(4096, 31), 10)):
        leg = trace_leg(tr, torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), M, reps, box_ceiling_gbps)
        leg.update(tets=int(len(cells)), points=int(len(pts)), max_ray_triangles=M, load_tetrahedra_s=load_s[1],
                   load_tetrahedra_first_call_s=load_s[0], qhull_s_host=qhull_s)
        out[name] = leg
    del tr
    torch.cuda.empty_cache()
    return out


def _timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ops_leg(tn, tracer, pts, scenes, M, dev, cpu=True):
    """The three stand-alone ops the UNMODIFIED reference model calls every step (model.py:560-573 and autograd):
    find_visited_cells (src/tetrahedra_tracer.cu:115-193; 37 algorithmic bytes per sample: 4 read + 33 written),
    interpolate_values forward / backward (:195-290; 284 B per sample: 16 ids + 12 weights + 256 of features), on the C4
    mesh at the two sizes SURVEY.md 8(d) names: a 4096-ray training batch with 513 fine samples, and a 65,536-ray
    evaluation chunk with 256 coarse samples.  CPU baseline (rank 0, bounded): the oracle's port of the same three
    functions (timing build, all host cores) and the MLP forward in torch on the CPU."""
    cpp = tn.cpp
    V = len(pts)
    torch.manual_seed(0)
    field = torch.randn(64, V, device=dev)
    res = {}
    keep = None
    fo, fd = frame_rays(scenes, 0, 800, 800)
    bo, bd = scenes.outside_in_rays(4096, 1)
    for name, (o_np, d_np), S, want in (("4096x513", (bo, bd), 513, 4096), ("65536x256", (fo, fd), 256, 65536)):
        o, d = torch.from_numpy(o_np).to(dev), torch.from_numpy(d_np).to(dev)
        out = tracer.trace_rays(o, d, M, compact_rows=True)
        idx = torch.nonzero(out["num_visited_cells"] > 0)[:, 0][:want]
        lists = [out[k][idx].contiguous() for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates",
                                                      "hit_distances", "vertex_indices")]
        del out
        R = int(idx.numel())
        nv = lists[0].long()
        near = lists[3][:, 0, 0][:, None]
        far = torch.gather(lists[3][:, :, 1], 1, (nv[:, None] - 1).clamp_min(0))
        ts = torch.linspace(0.0, 1.0, S, device=dev)[None]
        dist = (near * (1 - ts) + far * ts).contiguous()
        n = R * S
        traced = tracer.find_visited_cells(*lists, dist)
        vi, bc = traced["vertex_indices"], traced["barycentric_coordinates"]
        ms_match = _timed(lambda: tracer.find_visited_cells(*lists, dist), 10)
        ms_fwd = _timed(lambda: cpp.interpolate_values(vi, bc, field), 10)
        g = torch.randn(R, S, 64, device=dev)
        ms_bwd = _timed(lambda: cpp.interpolate_values_backward(vi, bc, field, g), 10)

        def roof(ms, bytes_per_sample):
            gbs = n * bytes_per_sample / (ms * 1e-3) / 1e9
            return {"ms": ms, "samples_per_s": n / (ms * 1e-3), "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
                                                                                "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                                                                "algorithmic_bytes_per_sample": bytes_per_sample}}
        res[name] = {"rays": R, "samples_per_ray": S, "matched_fraction": float(traced["mask"].float().mean()),
                     "find_visited_cells": roof(ms_match, 37), "interpolate_values": roof(ms_fwd, 284),
                     "interpolate_values_backward": roof(ms_bwd, 284)}
        if name == "4096x513":
            keep = ([x.cpu().numpy() for x in lists], dist.cpu().numpy(), vi.cpu().numpy(), bc.cpu().numpy(),
                    field.cpu().numpy(), R, S)
        del lists, traced, vi, bc, g, dist
        torch.cuda.empty_cache()
    if cpu and keep is not None:
        res["cpu_baseline"] = cpu_ops_baseline(*keep)
    return res


def cpu_ops_baseline(lists, dist, vi, bc, field, R, S):
    """the oracle's find_matched_cells / interpolate_values (+ adjoint), timing build (-O3 -march=native
    -fopenmp), all host cores, on the 4096 x 513 batch; the shallow MLP forward in torch on the CPU on a bounded sample."""
    from oracle import tn_oracle

    n = R * S
    out = {"cores": tn_oracle.num_threads(), "kind": "port", "sample": f"{R} rays x {S} samples of the C4 mesh (the 4096x513 leg)"}
    tn_oracle.find_visited_cells(*[x[:64] for x in lists], dist[:64], fast=True)          # loads / builds the timing library
    t0 = time.perf_counter()
    tn_oracle.find_visited_cells(*lists, dist, fast=True)
    dt = time.perf_counter() - t0
    out["find_visited_cells"] = {"s": dt, "samples_per_s": n / dt}
    t0 = time.perf_counter()
    feats = tn_oracle.interpolate_values(vi, bc, field, fast=True)
    dt = time.perf_counter() - t0
    out["interpolate_values"] = {"s": dt, "samples_per_s": n / dt}
    g = np.ones((R, S, 64), np.float32)
    t0 = time.perf_counter()
    tn_oracle.interpolate_values_backward(vi, bc, field, g, fast=True)
    dt = time.perf_counter() - t0
    out["interpolate_values_backward"] = {"s": dt, "samples_per_s": n / dt}
    # the shallow MLP + heads (model.py:414-455, 602-621) as plain torch on the host cores
    render = importlib.import_module("tetra-nerf_amd.render")
    torch.manual_seed(0)
    mlp = render.TetraMLP()
    m = min(n, 1 << 18)
    x = torch.from_numpy(np.ascontiguousarray(feats.reshape(-1, 64)[:m]))
    dirs = torch.nn.functional.normalize(torch.randn(m, 3), dim=-1)
    with torch.no_grad():
        mlp(x[:4096], dirs[:4096])
        t0 = time.perf_counter()
        mlp(x, dirs)
        dt = time.perf_counter() - t0
    flop = 2 * (64 * 128 + 128 * 128 * 2 + 128 + 155 * 128 + 128 * 3)
    out["mlp_forward_torch_cpu"] = {"s": dt, "samples_per_s": m / dt, "TFLOPs": m * flop / dt / 1e12, "samples": m,
                                    "threads": torch.get_num_threads()}
    return out


def train_leg(tn, tracer, num_vertices, scenes, M, dev, iters=10):
    """BASELINE.json configs[3] is a TRAINING batch: one optimisation step's forward + backward on 4096 outside-in rays of
    the C4 mesh for both shipped configurations (registration.py:20-61): trace_rays + TetraRenderer.render_train (stratified
    coarse samples, PDF fine pass, gather + MLP + heads, GradientScaler, renderers) + loss.backward() down to
    tetrahedra_field and the 12 weight tensors -- through the fused HIP nodes and, beside it, through PyTorch autograd of
    the plain statement (HIP tracer / matcher / gather + nn.Linear MLP: what the unmodified reference model runs).
    useful FLOP = coarse density pass forward (82,176 / sample) + fine pass forward + dX + dW (3 x 122,624 / sample)."""
    render = importlib.import_module("tetra-nerf_amd.render")
    o, d = scenes.outside_in_rays(4096, 1)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    target = torch.rand(len(o), 3, device=dev)
    out = {}
    for name, (s_c, s_f, biased, scaling) in (("tetra-nerf-original", (256, 256, False, False)), ("tetra-nerf", (128, 128, True, True))):
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(dev)
        field = ((torch.rand(64, num_vertices, device=dev) * 2 - 1) * 1e-4)
        field[1:4] = torch.rand(3, num_vertices, device=dev) * 2 - 1
        field.requires_grad_(True)
        params = [field] + list(mlp.parameters())
        opt = torch.optim.SGD(params, lr=1e-3)
        rd = render.TetraRenderer(tracer, field, mlp, s_c, M, fused=True, num_fine_samples=s_f, biased=biased)
        res = {}
        for what, fused in (("fused", True), ("pytorch_autograd", False)):
            def step():
                opt.zero_grad(set_to_none=True)
                o_ = rd.render_train(o, d, gradient_scaling=scaling, fused=fused)
                loss = ((o_["rgb"] - target) ** 2).mean()
                loss.backward()
                opt.step()
                return o_
            hit = int(step()["ray_mask"].sum())
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                step()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            flop = hit * (s_c * 82176 + 3 * (s_c + s_f + 1) * 122624)
            res[what] = {"ms_per_iteration": ms, "useful_TFLOPs": flop / (ms * 1e-3) / 1e12,
                         "frac_of_fp32_mfma_peak_157.3": flop / (ms * 1e-3) / 1e12 / 157.3}
        res.update(rays=len(o), hitting_rays=hit, samples_per_ray=f"{s_c} coarse (density only, no gradient) + {s_c + s_f + 1} fine",
                   useful_flop_per_iteration=flop, speedup_vs_pytorch_autograd=res["pytorch_autograd"]["ms_per_iteration"] / res["fused"]["ms_per_iteration"],
                   step="trace_rays + render_train forward + backward to the field and 12 weight tensors + SGD step")
        out[name] = res
    return out


def sharded_render_leg(tn, tracer, num_vertices, scenes, width, height, M, dev, reps=2, samples=256, chunk=65536):
    """north_star's multi-GPU flow: every rank renders its share of ONE 800x800 frame (replicated mesh / field / MLP;
    4096-ray tiles dealt round-robin so that the hitting rays -- the MLP work -- are balanced) and the rendered tiles are
    all-gathered as one RCCL collective.  Returns max-over-ranks times + the per-rank render times."""
    render = importlib.import_module("tetra-nerf_amd.render")
    sharding = importlib.import_module("tetra-nerf_amd.sharding")
    torch.manual_seed(0)   # replicated parameters, as DDP keeps them
    mlp = render.TetraMLP().to(dev)
    field = ((torch.rand(64, num_vertices, device=dev) * 2 - 1) * 1e-4)
    field[1:4] = torch.rand(3, num_vertices, device=dev) * 2 - 1
    rd = render.TetraRenderer(tracer, field, mlp, samples, M, fused=True)
    o, d = frame_rays(scenes, 0, width, height)   # the SAME frame on every rank
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    sharding.render_sharded(rd.render, o, d, chunk=chunk)   # warm-up (RCCL communicator, allocator)
    tot = {"render": 0.0, "all_gather": 0.0}
    tm = {}
    for _ in range(reps):
        tm = {}
        full = sharding.render_sharded(rd.render, o, d, chunk=chunk, timings=tm)
        for k in tot:
            tot[k] += tm[k] / reps
    t_render = sharding.max_over_ranks(tot["render"], device=dev)
    t_gather = sharding.max_over_ranks(tot["all_gather"], device=dev)
    per_rank_ms = [x * 1e3 for x in sharding.gather_scalars(tot["render"], device=dev)]
    per_rank_hits = [int(x) for x in sharding.gather_scalars(float(tm.get("hitting_rays", 0)), device=dev)]
    R = o.shape[0]
    return {"rays": R, "hitting_rays": int(full["ray_mask"].sum()), "samples_per_ray": samples,
            "ms_per_frame": (t_render + t_gather) * 1e3, "render_ms": t_render * 1e3, "all_gather_ms": t_gather * 1e3,
            "render_ms_per_rank": {"min": min(per_rank_ms), "max": max(per_rank_ms), "all": per_rank_ms},
            "hitting_rays_per_rank": per_rank_hits,
            "rendered_rays_per_s": R / (t_render + t_gather),
            "partition": f"tiles of {sharding.TILE_RAYS} rays dealt round-robin to the ranks (hitting rays per rank balanced)",
            "collective": "one all_gather_into_tensor of [rays/N, 6] f32 (rgb, accumulation, depth, mask) per frame + inverse permutation",
            "pass": "coarse only (uniform samples), fused MLP + composite"}


def ddp_train_leg(tn, scenes, dev, M, rank, world, iters=5, mesh_points=45000, mesh_seed=2, rays=4096, warm=6):
    """The reference's only training-time collective (pipeline.py:53-58): the model replicated under
    DistributedDataParallel(find_unused_parameters=True), every rank training on ITS OWN 4096-ray batch of the C4 mesh
    (BASELINE.json configs[3]), gradients of tetrahedra_field (256 V bytes) + the MLP (245 KB) all-reduced over RCCL.
    The model is render.TetraNerfModule: forward = the fused autograd nodes, so DDP's reducer hooks fire from the HIP
    adjoint kernels' gradients.  Reported: ms per iteration (max over ranks) with the all-reduce and with
    `no_sync()` (no collective: the difference is the exposed all-reduce time), whole-job rays/s."""
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP

    render = importlib.import_module("tetra-nerf_amd.render")
    sharding = importlib.import_module("tetra-nerf_amd.sharding")
    pts, cells = scenes.random_mesh(mesh_points, mesh_seed)
    tr = tn.TetrahedraTracer(dev)
    tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = scenes.outside_in_rays(rays, 100 + rank)          # a different batch per rank
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    target = torch.rand(len(o), 3, device=dev)
    out = {}
    for name, (s_c, s_f, biased, scaling) in (("tetra-nerf-original", (256, 256, False, False)), ("tetra-nerf", (128, 128, True, True))):
        torch.manual_seed(0)                                   # replicated initial parameters (DDP broadcasts rank 0's anyway)
        module = render.TetraNerfModule(tr, len(pts), s_c, M, num_fine_samples=s_f, biased=biased, gradient_scaling=scaling).to(dev)
        model = DDP(module, device_ids=[dev.index], find_unused_parameters=True)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)

        def step(sync=True):
            opt.zero_grad(set_to_none=True)
            if sync:
                res = model(o, d)
                ((res["rgb"] - target) ** 2).mean().backward()
            else:
                with model.no_sync():
                    res = model(o, d)
                    ((res["rgb"] - target) ** 2).mean().backward()
            opt.step()

        res = {}
        for what, sync in (("ddp", True), ("no_sync", False)):
            for _ in range(warm):     # DDP rebuilds its buckets after the first iteration (find_unused_parameters=True): not timed
                step(sync)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                step(sync)
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            res[what] = sharding.max_over_ranks((time.perf_counter() - t0) / iters, device=dev) * 1e3
        grad_bytes = sum(p.numel() for p in module.parameters()) * 4
        out[name] = {"ms_per_iteration": res["ddp"], "ms_per_iteration_no_sync": res["no_sync"], "warmup_iterations": warm,
                     "exposed_all_reduce_ms": res["ddp"] - res["no_sync"], "rays_per_s": world * rays / (res["ddp"] * 1e-3),
                     "rays_per_rank": rays, "all_reduced_bytes_per_iteration": grad_bytes}
        del model, module, opt
    out.update(n_gpus=world, backend=dist.get_backend(), mesh_tets=int(len(cells)),
               wrapper="DistributedDataParallel(find_unused_parameters=True) as in pipeline.py:53-58",
               step="trace_rays + render_train forward + backward (gradient all-reduce) + SGD step, one 4096-ray batch per rank")
    return out


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: re-execute under torch.distributed.run, one rank
    per GPU (the same command line the driver uses), and pass its exit code on."""
    import subprocess

    one_device = os.environ.get("TETRANERF_BENCH_ONE_DEVICE") == "1"
    if not one_device and torch.cuda.device_count() < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible "
                         "(TETRANERF_BENCH_ONE_DEVICE=1 + TETRANERF_BENCH_BACKEND=gloo run all ranks on cuda:0)")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def cpu_baseline(pts, cells, o, d, M, target_s=30.0):
    """Oracle (BVH all-hits + sort + pairing, OpenMP) on a bounded sample of the bench rays."""
    from oracle import tn_oracle

    ot = tn_oracle.OracleTracer(use_bvh=True, fast=True)  # timing-only build: -O3 -march=native
    ot.load_tetrahedra(pts, cells)
    rng = np.random.default_rng(0)
    perm = rng.permutation(len(o))
    probe = perm[:65536]
    ot.trace_rays(o[probe[:256]], d[probe[:256]], M)  # thread pool warm-up
    t0 = time.perf_counter()
    ot.trace_rays(o[probe], d[probe], M)
    rate = len(probe) / max(time.perf_counter() - t0, 1e-6)
    # bounded sample: about `target_s` seconds of CPU work, at most 12 passes over the frame
    n = int(min(12 * len(o), max(8192, rate * target_s)))
    idx = np.resize(perm, n)
    os_, ds_ = np.ascontiguousarray(o[idx]), np.ascontiguousarray(d[idx])
    t0 = time.perf_counter()
    res = ot.trace_rays(os_, ds_, M)
    dt = time.perf_counter() - t0
    inter = int(res["num_visited_cells"].astype(np.int64).sum())
    return {"value": inter / dt, "unit": "ray-tet intersections/s", "cores": tn_oracle.num_threads(),
            "kind": "port", "rays_per_s": n / dt, "build": "gcc -O3 -march=native -fopenmp (timing build of oracle/tn_oracle.c)",
            "sample": f"{n} random rays of the same frame, M={M}, oracle BVH path, {dt:.1f} s"}


def _get(dct, *path):
    for k in path:
        if not isinstance(dct, dict) or k not in dct:
            return None
        dct = dct[k]
    return dct


def _r(x, nd=4):
    return round(float(x), nd) if isinstance(x, (int, float)) and x == x else x


def secondary_summary(line):
    """Every graded secondary figure of the run in one small object (`roofline.secondary` of the printed line): the driver
    keeps `roofline` whole, while the complete result is tens of KB (file named by --full-json)."""
    cfg = line.get("configs") if isinstance(line.get("configs"), dict) else {}
    sec = {}
    for key, name in (("C4_frame", "C4_frame_800x800"), ("C4_batch_4096_oi", "C4_batch_4096_outside_in"),
                      ("C4_batch_4096_io", "C4_batch_4096_inside_out"), ("C5", "C5_2^20_outside_in"),
                      ("C6_frame", "C6_frame_512x400"), ("C6_batch_4096", "C6_batch_4096_outside_in")):
        leg = cfg.get(name)
        if isinstance(leg, dict):
            e = {"frac": _r(_get(leg, "roofline", "frac")), "ms": _r(leg.get("ms")), "tets": leg.get("tets"),
                 "frac_of_box_write_ceiling": _r(_get(leg, "roofline", "frac_of_box_write_ceiling"))}
            if "load_tetrahedra_s" in leg:
                e["load_tetrahedra_s"] = _r(leg["load_tetrahedra_s"])
            sec[key] = e
    tr = cfg.get("C4_train_4096")
    if isinstance(tr, dict):
        sec["train"] = {k: {"frac": _r(_get(v, "fused", "frac_of_fp32_mfma_peak_157.3")), "ms": _r(_get(v, "fused", "ms_per_iteration")),
                            "pytorch_autograd_ms": _r(_get(v, "pytorch_autograd", "ms_per_iteration"))}
                        for k, v in tr.items() if isinstance(v, dict)}
    ops = cfg.get("C4_ops")
    if isinstance(ops, dict):
        sec["ops"] = {shape: {op: _r(_get(v, op, "roofline", "frac")) for op in ("find_visited_cells", "interpolate_values",
                                                                                "interpolate_values_backward")}
                      for shape, v in ops.items() if shape != "cpu_baseline" and isinstance(v, dict)}
    rn = line.get("render")
    if isinstance(rn, dict):
        sec["render"] = {"coarse_only_ms": _r(rn.get("ms_per_frame"), 2), "mlp_forward_frac": _r(_get(rn, "roofline_mlp", "frac"))}
        for k, v in (rn.get("eval_configs") or {}).items():
            sec["render"][k] = {"one_launch_ms": _r(_get(v, "fp32", "ms_per_frame"), 2), "chain_ms": _r(_get(v, "fp32_kernel_chain", "ms_per_frame"), 2),
                                "bf16x3_ms": _r(_get(v, "bf16x3", "ms_per_frame"), 2), "bf16x3_chain_ms": _r(_get(v, "bf16x3_kernel_chain", "ms_per_frame"), 2),
                                "one_launch_over_chain": _r(v.get("one_launch_over_chain")), "one_launch_over_chain_bf16x3": _r(v.get("one_launch_over_chain_bf16x3"))}
    if isinstance(line.get("sharded_render"), dict):
        sec["sharded_render"] = {"ms_per_frame": _r(line["sharded_render"].get("ms_per_frame"), 2),
                                 "all_gather_ms": _r(line["sharded_render"].get("all_gather_ms"))}
    if isinstance(line.get("ddp_train_4096"), dict):
        sec["ddp_train_4096"] = {k: {"ms": _r(v.get("ms_per_iteration")), "exposed_all_reduce_ms": _r(v.get("exposed_all_reduce_ms"))}
                                 for k, v in line["ddp_train_4096"].items() if isinstance(v, dict)}
    return sec


SHORT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "rays_per_s", "config", "roofline", "cpu_baseline")


def emit(line, args):
    """The complete result -> a file (and stderr); ONE short JSON line (<= 4 KB) -> stdout, the last thing printed."""
    line["roofline"]["secondary"] = secondary_summary(line)
    full = json.dumps(line)
    target = Path(args.full_json) if args.full_json else ROOT / "gpurun_out" / "bench_full.json"
    try:
        target.parent.mkdir(parents=True, exist_ok=True)
        target.write_text(full + "\n")
        where = str(target)
    except OSError:
        where = "stderr only"
    print(full, file=sys.stderr, flush=True)
    short = {k: line[k] for k in SHORT_KEYS if k in line}
    roof = dict(short["roofline"])
    roof.pop("box_write_ceiling", None)          # the scalar box_write_ceiling_GBps stays
    short["roofline"] = roof
    if isinstance(short.get("cpu_baseline"), dict):
        short["cpu_baseline"] = {k: v for k, v in short["cpu_baseline"].items() if k != "ops"}
    short["certification_mismatches"] = _get(line, "certification_cross_check", "mismatches")
    short["full_result"] = where
    print(json.dumps(short), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--max-ray-triangles", type=int, default=512)
    ap.add_argument("--mesh-points", type=int, default=15000)
    ap.add_argument("--mesh-seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-render", action="store_true", help="skip the rendered-rays/s legs")
    ap.add_argument("--no-configs", action="store_true", help="skip the C4 / C5 legs")
    ap.add_argument("--no-ddp", action="store_true", help="skip the DDP training leg of multi-rank runs")
    ap.add_argument("--no-c6", action="store_true", help="skip the reference-scale leg (1M points / 6.7M tets; 60-110 s of Qhull beside the other legs)")
    ap.add_argument("--full-json", default=None,
                    help="file for the complete result (default: gpurun_out/bench_full.json under the repository); stdout carries the "
                         "short line (<= 4 KB) whose `roofline.secondary` holds every graded fraction")
    ap.add_argument("--no-calibration", action="store_true",
                    help="skip the box write-ceiling measurement and the serialised breakdown (rocprofv3 PMC passes: only the bench launches)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)            # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: the two must agree")
    one_device = os.environ.get("TETRANERF_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("TETRANERF_BENCH_BACKEND", "nccl")
    dev = torch.device("cuda", 0 if one_device else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    tn = importlib.import_module("tetra-nerf_amd")
    scenes = importlib.import_module("tetra-nerf_amd.scenes")

    c6 = start_c6_mesh() if (world == 1 and rank == 0 and not args.no_configs and not args.no_c6) else None
    pts, cells = scenes.random_mesh(args.mesh_points, args.mesh_seed)
    o_np, d_np = frame_rays(scenes, rank, args.width, args.height)
    R, M = len(o_np), args.max_ray_triangles

    tracer = tn.TetrahedraTracer(dev)
    x_dev, c_dev = torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev)
    load_s = 0.0
    for _ in range(2):   # the second call is the steady-state figure (the first one loads the build kernels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tracer.load_tetrahedra(x_dev, c_dev)
        torch.cuda.synchronize()
        load_s = time.perf_counter() - t0
    o = torch.from_numpy(o_np).to(dev)
    d = torch.from_numpy(d_np).to(dev)

    def step():
        out = tracer.trace_rays(o, d, M)
        return out["num_visited_cells"]

    inter = 0
    for _ in range(max(args.warmup, 0)):
        inter = int(step().sum())
    if args.warmup <= 0:
        inter = int(step().sum())
    stats = tracer.trace_stats()
    reasons = {str(k): v for k, v in tracer.flag_reasons().items()}   # why the walk handed rays over (include/tetranerf_hip.h)
    xcheck = cross_check(tracer)

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    tot_inter = float(inter)
    sharding = importlib.import_module("tetra-nerf_amd.sharding")
    per_rank_ms = [x / args.steps * 1e3 for x in sharding.gather_scalars(elapsed, device=dev)]   # a straggler shows here
    per_rank_kern_ms = sharding.gather_scalars(kern_ms, device=dev)
    elapsed = sharding.max_over_ranks(elapsed, device=dev)           # MAX over ranks
    tot_inter, tot_rays = sharding.sum_over_ranks([tot_inter, float(R)], device=dev)  # whole-job units

    # the write rate this box sustains, measured in this process (every rank: its own GPU), and the serialised per-kernel
    # breakdown of one extra launch
    if args.no_calibration:
        ceiling, breakdown = {"GBps": float("nan"), "note": "--no-calibration"}, None
    else:
        full = tracer.trace_rays(o, d, M)
        ceiling = box_write_ceiling(tn, full)
        del full
        breakdown = trace_breakdown(tracer, o, d, M) if stats.get("walk", 0) else None
    per_rank_ceiling = sharding.gather_scalars(ceiling["GBps"], device=dev)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        alg_bytes = R * (28 + 52 * M)  # o,d + count + dense rows, SURVEY.md 8(d)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "ray_tet_intersections_per_s",
            "value": tot_inter * args.steps / elapsed,
            "unit": "ray-tet intersections/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "rays_per_s": tot_rays * args.steps / elapsed,
            "config": {
                "workload": f"configs[1] stand-in: {len(cells)}-tet Delaunay ({args.mesh_points} uniform points, seed {args.mesh_seed}), "
                            f"{args.width}x{args.height} frame = {R} rays/GPU, trace_rays M={M}, dense reference rows",
                "tets": int(len(cells)), "mesh_sha256": mesh_sha256(pts, cells), "rays_per_gpu": R, "max_ray_triangles": M,
                "intersections_per_frame": inter, "sharding": f"rays/{world} ranks, no collective",
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args, R, M),
                "traffic_source": "committed rocprofv3 PMC passes of this command (profiles/traffic.json), not read in this run",
                # `achieved` / `frac` are LAUNCH figures (conservative): the whole trace_rays call -- walk, segment writer, tail
                # fills, literal pairing, BVH fallback, cross-check -- over the HIP-event duration of the call.  The dominant
                # kernel by time and bytes is k_fill_rows_fine (the constant tails, 88 % of the bytes); what a pure fill
                # of the same rows reaches on this box, in this process, is `box_write_ceiling_GBps`
                "kernel": "trace_rays launch: k_hull_entry, k_trace_walk, k_write_segments, k_fill_rows_fine (dominant), "
                          "k_postprocess_log, k_trace_general, k_verify_counts",
                "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                "box_write_ceiling_GBps": ceiling["GBps"], "box_write_ceiling": ceiling,
                "frac_of_box_write_ceiling": achieved / ceiling["GBps"],
                "breakdown_ms_serialised": breakdown,
            },
            "ms_per_step_per_rank": per_rank_ms, "kernel_ms_per_rank": per_rank_kern_ms,
            "box_write_ceiling_GBps_per_rank": per_rank_ceiling,
            "trace_path_stats": stats,
            "walk_hand_over_reasons": reasons,
            # the always-on cross-check of the walk's certification (count-only BVH traversal beside the writer and the fill) of
            # every certified ray of the risk classes + a blind sample of 1 in `stride`: all `mismatches` must be 0
            "certification_cross_check": xcheck,
            "load_tetrahedra_s": load_s,
        }
        # secondary figure of SURVEY.md 8(d): the same launch without the constant tails of the dense
        # reference layout ("dense_tails" = 0; rows valid through num_visited only) -- what a
        # non-materialising consumer (find_visited_cells, the render path) needs
        def step_compact():
            return tracer.trace_rays(o, d, M, compact_rows=True)["num_visited_cells"]

        step_compact()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            step_compact()
        e1.record()
        torch.cuda.synchronize()
        seg_ms = e0.elapsed_time(e1) / args.steps
        seg_bytes = 28 * R + 52 * inter
        line["segments_only"] = {"ms_per_step": seg_ms, "rays_per_s": R / (seg_ms * 1e-3),
                                 "intersections_per_s": inter / (seg_ms * 1e-3),
                                 "algorithmic_bytes_per_launch": seg_bytes,
                                 "achieved_GBps": seg_bytes / (seg_ms * 1e-3) / 1e9,
                                 "note": "per-call flag TN_TRACE_COMPACT_ROWS (no reference counterpart): 28 B/ray + 52 B/segment; instruction-issue-bound walk"}
        if not args.no_render and world == 1:   # secondary legs: single-GPU runs only (rank 0 alone would hold the job)
            line["render"] = render_leg(tn, tracer, len(pts), o, d, M, dev)
    # the sharded render: a collective leg, every rank takes part
    shr = None
    if not args.no_render:
        shr = sharded_render_leg(tn, tracer, len(pts), scenes, args.width, args.height, M, dev)
    # the training collective: DDP gradient all-reduce, one 4096-ray batch per rank
    ddp = None
    if dist is not None and not args.no_ddp:
        ddp = ddp_train_leg(tn, scenes, dev, M, rank, world)
    if rank == 0:
        if ddp is not None:
            line["ddp_train_4096"] = ddp
        if shr is not None:
            shr["n_gpus"] = world
            line["sharded_render"] = shr
        if not args.no_configs and world == 1:
            del tracer
            torch.cuda.empty_cache()
            line["configs"] = config_legs(tn, scenes, dev, M, ceiling["GBps"])
            if c6 is not None:
                mesh6 = finish_c6_mesh(c6)
                if mesh6 is not None:
                    line["configs"].update(c6_legs(tn, scenes, dev, mesh6, ceiling["GBps"]))
                    del mesh6
                else:
                    line["configs"]["C6_ref_scale"] = "Qhull child process failed or timed out"
            line["configs"]["C2_frame_800x800"] = {"rays": R, "ms": kern_ms, "rays_per_s": R / (kern_ms * 1e-3),
                                                   "intersections_per_s": inter / (kern_ms * 1e-3), "intersections": inter,
                                                   "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                                                                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                                                "frac_of_box_write_ceiling": achieved / ceiling["GBps"]},
                                                   "breakdown_ms_serialised": breakdown,
                                                   "paths": stats, "tets": int(len(cells)), "note": "the headline workload"}
        if world > 1:
            line["configs"] = "single-GPU legs (C4 / C5 / training / stand-alone ops): see the --gpus 1 line"
        if not args.no_cpu_baseline:
            # rank 0's host cores (the other ranks are idle by now); a shorter sample in multi-rank runs
            line["cpu_baseline"] = cpu_baseline(pts, cells, o_np, d_np, M, target_s=30.0 if world == 1 else 10.0)
            ops = line.get("configs", {}).get("C4_ops") if isinstance(line.get("configs"), dict) else None
            if ops and "cpu_baseline" in ops:
                line["cpu_baseline"]["ops"] = ops["cpu_baseline"]
        emit(line, args)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
