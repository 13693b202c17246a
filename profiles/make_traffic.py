#!/usr/bin/env python3
"""profiles/traffic.json from the two rocprofv3 PMC passes of the bench command (profiles/rNNend_call.sh):
    python profiles/make_traffic.py <pmc_write.txt> <pmc_fetch.txt> <round>
Per dense trace_rays launch = the kernels of one call.  The bench command also runs the compact-rows leg (same kernels,
fewer bytes): per kernel name the MAX over dispatches is the dense leg; kernels launched twice per call (k_trace_general:
fallback rays + cross-check re-trace; k_verify_counts: blind sample + risk classes) count max + the mean of the rest.
WRITE_SIZE / FETCH_SIZE are KiB per dispatch; FETCH_SIZE x 2 = the gfx950 wide-read correction (MI355X_MICROARCH.md, HBM
section)."""
import json
import sys
from pathlib import Path

KERNELS = {"k_hull_entry": 1, "k_trace_walk": 1, "k_write_segments": 1, "k_fill_range<false>": 1, "k_fill_range<true>": 1,
           "k_fill_rows_fine": 1, "k_fill_linear": 1, "k_postprocess_log": 1,
           "k_trace_general": 2, "k_verify_counts": 2}


def read(path):
    out = {}
    for line in Path(path).read_text().splitlines():
        for k in KERNELS:
            if k in line and ("WRITE_SIZE" in line or "FETCH_SIZE" in line):
                f = line.split()
                avg, mn, mx, n = float(f[-4]), float(f[-3]), float(f[-2]), int(f[-1])
                out[k] = (avg, mn, mx, n)
    return out


def per_launch(stats):
    tot, parts = 0.0, {}
    for k, per_call in KERNELS.items():
        if k not in stats:
            continue
        avg, mn, mx, n = stats[k]
        b = mx if per_call == 1 else mx + (avg * per_call - mx) if avg * per_call > mx else mx
        parts[k] = b * 1024
        tot += b * 1024
    return tot, parts


w, wp = per_launch(read(sys.argv[1]))
f, fp = per_launch(read(sys.argv[2]))
R, M = 640000, 512
alg = R * (28 + 52 * M)
doc = {"workload": {"mesh_points": 15000, "mesh_seed": 0, "rays": R, "M": M},
       "hbm_bytes_per_launch": int(w + 2 * f), "algorithmic_bytes_per_launch": alg,
       "write_bytes": int(w), "fetch_bytes_counter": int(f), "ratio": (w + 2 * f) / alg,
       "write_bytes_per_kernel": {k: int(v) for k, v in wp.items()}, "fetch_bytes_per_kernel": {k: int(v) for k, v in fp.items()},
       "how": "rocprofv3 --pmc WRITE_SIZE and --pmc FETCH_SIZE in separate passes of `python bench.py --steps 3 --warmup 1 "
              "--no-cpu-baseline --no-render --no-configs --no-calibration` (profiles/r06zz_call.sh -> " + Path(sys.argv[1]).name +
              ", " + Path(sys.argv[2]).name + "); profiles/make_traffic.py: per dense launch = max over dispatches of every kernel of "
              "the call (the min is the compact-rows leg), WRITE_SIZE + 2 x FETCH_SIZE (gfx950 wide-read correction)",
       "round": sys.argv[3]}
print(json.dumps(doc, indent=1))
