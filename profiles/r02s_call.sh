cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_trace_gpu.py tests/test_parity_configs_gpu.py tests/test_reference_suite_gpu.py -q -m gpu -x 2>&1 | grep -v "^W2026" | grep -E "passed|failed|rror" | tail -4
python - <<'PY' 2>&1 | grep -v "^W2026" | tee gpurun_out/r02s_small_batch.txt
import importlib, sys, torch
sys.path.insert(0, ".")
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg, npts, seed in (("c2", 15000, 0), ("c4", 45000, 2), ("c5", 150000, 3)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    for name, (o, d) in (("outside-in", scenes.outside_in_rays(4096, 1)), ("inside-out", scenes.inside_out_rays(4096, 2))):
        o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
        def call():
            x = tr.trace_rays(o, d, M); del x
        print(f"{cfg} 4096 rays {name}: {min(timed(call), timed(call)):.3f} ms", flush=True)
    del tr
PY
