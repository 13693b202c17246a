"""What makes the SAME fill kernel 30 % slower in one process than in the next (r06r_*: 5.1 vs 6.7 TB/s)?
One process: `trials` x (free everything, allocate the four dense row arrays of a 640000 x 512 launch, time tn_fill_rows
alone 5x) -- prints the pointers and the rate; between trials an extra allocation of `shift` MiB stays alive so the rows land
elsewhere.  Run several times in one gpurun call with clock samples beside it (profiles/r06s_call.sh)."""
import importlib, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
tn = importlib.import_module("tetra-nerf_amd")
cpp = importlib.import_module("tetra-nerf_amd.tetranerf_cpp_extension")
dev = torch.device("cuda:0")
trials, shift = int(sys.argv[1]), int(sys.argv[2])
R, M = 640000, 512
keep = []
t0 = time.time()
for t in range(trials):
    torch.cuda.empty_cache()
    vc = torch.empty(R, M, dtype=torch.int32, device=dev)
    bc = torch.empty(R, M, 2, 3, dtype=torch.float32, device=dev)
    hd = torch.empty(R, M, 2, dtype=torch.float32, device=dev)
    vi = torch.empty(R, M, 4, dtype=torch.int32, device=dev)
    nbytes = sum(x.numel() * 4 for x in (vc, bc, hd, vi))
    for _ in range(2):
        cpp.fill_rows(vc, bc, hd, vi, 0)
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); cpp.fill_rows(vc, bc, hd, vi, 0); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    # torch's own fill of the biggest array, for a second opinion that does not go through this library
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bc.zero_(); torch.cuda.synchronize()
    e0.record(); bc.zero_(); e1.record(); torch.cuda.synchronize()
    zr = bc.numel() * 4 / e0.elapsed_time(e1) / 1e6
    print(f"t={time.time() - t0:6.1f}s trial {t}: ptrs {vc.data_ptr():#x} {bc.data_ptr():#x} {hd.data_ptr():#x} {vi.data_ptr():#x}  "
          f"fill_rows median {ms[2]:.3f} ms = {nbytes / ms[2] / 1e6:7.0f} GB/s (min {ms[0]:.3f} max {ms[-1]:.3f})  torch zero_ {zr:7.0f} GB/s", flush=True)
    del vc, bc, hd, vi
    if shift:
        keep.append(torch.empty(shift << 20, dtype=torch.uint8, device=dev))
