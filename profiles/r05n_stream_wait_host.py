#!/usr/bin/env python3
"""Does a cross-stream wait block the HOST on this ROCm build?  Host-side duration of wait_stream / event calls while the
device is busy for ~100 ms."""
import time, torch
dev = torch.device("cuda:0")
x = torch.randn(8192, 8192, device=dev)
main = torch.cuda.current_stream(dev)
side = torch.cuda.Stream(device=dev)
y = torch.zeros(1024, device=dev)
def busy():
    for _ in range(10): x @ x
for trial in range(3):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    busy(); t.append(time.perf_counter())
    side.wait_stream(main); t.append(time.perf_counter())
    with torch.cuda.stream(side): y.add_(1.0)
    t.append(time.perf_counter())
    main.wait_stream(side); t.append(time.perf_counter())
    y.add_(1.0); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    names = ["enqueue 10 GEMMs", "side.wait_stream(main)", "kernel on side", "main.wait_stream(side)", "kernel on main", "synchronize"]
    print("  ".join(f"{n}: {1e3 * (b - a):.3f} ms" for n, a, b in zip(names, t, t[1:])))
