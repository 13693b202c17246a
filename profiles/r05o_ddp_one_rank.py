#!/usr/bin/env python3
"""bench.ddp_train_leg on a ONE-rank "nccl" (= RCCL) group: the all-reduce of the gradients really runs as an RCCL kernel on RCCL's
own stream, ordered against the compute stream with events -- what does that ordering cost per iteration on this ROCm build?
(ddp = with the all-reduce, no_sync = without)"""
import importlib, json, os, sys
from pathlib import Path
import torch, torch.distributed as dist
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
tn = importlib.import_module("tetra-nerf_amd"); scenes = importlib.import_module("tetra-nerf_amd.scenes")
dev = torch.device("cuda:0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1, device_id=dev)
try:
    print(json.dumps(bench.ddp_train_leg(tn, scenes, dev, 512, 0, 1, iters=10), indent=1))
finally:
    dist.destroy_process_group()
