"""Is the placement sensitivity of tn_fill_rows (5.6 ... 7.1 TB/s over fresh allocations of the same rows) a property of
writing FOUR arrays at once, or of this kernel?  Per fresh allocation: tn_fill_rows; torch's own fill kernel over the four
arrays one after the other; the same four fills on four streams at once."""
import importlib, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
cpp = importlib.import_module("tetra-nerf_amd.tetranerf_cpp_extension")
import ctypes as _C
lib = _C.CDLL(str(importlib.import_module("tetra-nerf_amd._lib").LIB_PATH))
dev = torch.device("cuda:0")
R, M = 640000, 512
streams = [torch.cuda.Stream(dev) for _ in range(4)]


def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]


for t in range(int(sys.argv[1])):
    torch.cuda.empty_cache()
    vc = torch.empty(R, M, dtype=torch.int32, device=dev)
    bc = torch.empty(R, M, 2, 3, dtype=torch.float32, device=dev)
    hd = torch.empty(R, M, 2, dtype=torch.float32, device=dev)
    vi = torch.empty(R, M, 4, dtype=torch.int32, device=dev)
    arrs = (vc, bc, hd, vi)
    nbytes = sum(x.numel() * 4 for x in arrs)

    def seq():
        for x in arrs:
            x.zero_()

    def par():
        cur = torch.cuda.current_stream(dev)
        for s, x in zip(streams, arrs):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                x.zero_()
        for s in streams:
            cur.wait_stream(s)

    a = nbytes / timed(lambda: cpp.fill_rows(vc, bc, hd, vi, 0)) / 1e6
    flat = ""
    if hasattr(lib, "tn_debug_fill_flat"):
        import ctypes as C
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for mode in (1, 2):
            f = lambda: lib.tn_debug_fill_flat(C.c_size_t(R), C.c_uint32(M), C.c_int(mode), C.c_void_p(vc.data_ptr()), C.c_void_p(bc.data_ptr()),
                                               C.c_void_p(hd.data_ptr()), C.c_void_p(vi.data_ptr()), st)
            flat += f"  flat mode {mode}: {nbytes / timed(f) / 1e6:5.0f}"
    each = [x.numel() * 4 / timed(lambda: x.zero_()) / 1e6 for x in arrs]
    b = nbytes / timed(seq) / 1e6
    c = nbytes / timed(par) / 1e6
    print(f"alloc {t}: tn_fill_rows {a:5.0f}{flat}   torch fill each array " + " ".join(f"{e:5.0f}" for e in each) + f"   four in sequence {b:5.0f}   four streams at once {c:5.0f} GB/s", flush=True)
    del vc, bc, hd, vi, arrs
