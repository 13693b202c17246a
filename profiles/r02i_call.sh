cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_walk_gpu.py tests/test_parity_configs_gpu.py tests/test_trace_gpu.py tests/test_build_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^W2026" | tail -8
for cfg in "45000 2 frame" "150000 3 1048576"; do
  cd /tmp; rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o s -- python $GRAFT_REPO_ROOT/profiles/c5_once.py $cfg > /tmp/log.txt 2>&1
  echo "=== $cfg"; grep -v "^W2026" /tmp/log.txt | tail -1
  python $GRAFT_REPO_ROOT/profiles/rocprof_timeline.py $(find /tmp/pp -name "*.db" | head -1) 200 | tail -6
done
cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v "^W2026"
import importlib, sys, torch
sys.path.insert(0, ".")
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0"); M = 512
def timed(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for cfg, npts, seed, rays in (("c2", 15000, 0, "frame"), ("c4", 45000, 2, "frame"), ("c5", 150000, 3, "frame"), ("c5rays", 150000, 3, 1 << 20)):
    pts, cells = scenes.random_mesh(npts, seed)
    tr = tn.TetrahedraTracer(dev); tr.load_tetrahedra(torch.from_numpy(pts).to(dev), torch.from_numpy(cells).to(dev))
    o, d = bench.frame_rays(scenes, 0, 800, 800) if rays == "frame" else scenes.outside_in_rays(rays, 4)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    def frame():
        out = tr.trace_rays(o, d, M); del out
    gb = o.shape[0] * (28 + 52 * M) / 1e9
    res = {}
    for rep in range(2):
        for lr in ((0, 1) if rep == 0 else (1, 0)):
            tr.set_option("literal_rows", lr)
            res.setdefault(lr, []).append(timed(frame, 8 if rays == "frame" else 3))
    tr.set_option("literal_rows", 0); frame(); torch.cuda.synchronize()
    print(f"{cfg}: emit-mask {min(res[0]):.3f} ms ({gb/min(res[0])/8*100:.1f} %), literal rows {min(res[1]):.3f} ms ({gb/min(res[1])/8*100:.1f} %) {tr.trace_stats()} {tr.flag_reasons()}", flush=True)
    del tr
PY
