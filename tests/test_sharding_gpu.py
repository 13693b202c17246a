"""The N-GPU render flow on the one GPU a test box has: torch.distributed backend "nccl" (= RCCL on ROCm) with a
one-rank group, so that the RCCL communicator is really created and the all-gather of rendered tiles really runs as a
RCCL kernel on the device (the world_size-2 logic is covered on CPU by tests/test_sharding.py with gloo)."""
import importlib
import os
import socket

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_nccl_one_rank_sharded_render(tn, device, scenes):
    import torch
    import torch.distributed as dist

    sh = importlib.import_module("tetra-nerf_amd.sharding")
    render = importlib.import_module("tetra-nerf_amd.render")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1,
                            device_id=device)
    try:
        assert dist.get_backend() == "nccl"
        pts, cells = scenes.random_mesh(3000, 4)
        tr = tn.TetrahedraTracer(device)
        tr.load_tetrahedra(torch.from_numpy(pts).to(device), torch.from_numpy(cells).to(device))
        torch.manual_seed(0)
        mlp = render.TetraMLP().to(device)
        field = torch.randn(64, len(pts), device=device) * 0.5
        rd = render.TetraRenderer(tr, field, mlp, 64, 256, fused=True, num_fine_samples=32)
        o, d = scenes.pinhole_rays(200, 150, eye=(0.5, 2.4, 0.6), lookat=(0.5, 0.5, 0.5))
        to, td = torch.from_numpy(o).to(device), torch.from_numpy(d).to(device)
        tm = {}
        full = sh.render_sharded(rd.render, to, td, chunk=8192, timings=tm)
        ref = rd.render(to, td)
        for k in ("rgb", "accumulation", "depth"):   # (degenerate rays -- near == far -- have a NaN depth in both)
            assert torch.allclose(full[k], ref[k], rtol=0, atol=0, equal_nan=True), k
        assert torch.equal(full["ray_mask"], ref["ray_mask"])
        assert int(ref["ray_mask"].sum()) > 5000 and tm["all_gather"] > 0
        # the bench's reductions and the raw collective on device tensors: since round 5 a one-rank group runs the RCCL
        # all-reduce / all-gather too (the non-staging branch of sharding._all_reduce / _all_gather_into)
        calls = []
        real_ar, real_ag = dist.all_reduce, dist.all_gather_into_tensor
        dist.all_reduce = lambda t, *a, **k: (calls.append(("all_reduce", t.is_cuda)), real_ar(t, *a, **k))[1]
        dist.all_gather_into_tensor = lambda o_, i_, *a, **k: (calls.append(("all_gather", i_.is_cuda)), real_ag(o_, i_, *a, **k))[1]
        try:
            assert sh.max_over_ranks(1.5, device=device) == 1.5
            assert sh.sum_over_ranks([2.0, 3.0], device=device) == [2.0, 3.0]
            assert sh.gather_scalars(4.25, device=device) == [4.25]
            # dealt tiles with a short last tile (R not a multiple of the tile): deal -> all-gather -> un-deal on device rows
            rows = torch.arange(10000 * 6, dtype=torch.float32, device=device).view(10000, 6)
            mine = sh.deal_tiles(10000, 0, 1, 4096).to(device)
            back = sh.all_gather_dealt(rows.index_select(0, mine), 10000, tile=4096)
            assert torch.equal(back, rows)
        finally:
            dist.all_reduce, dist.all_gather_into_tensor = real_ar, real_ag
        assert calls == [("all_reduce", True), ("all_reduce", True), ("all_gather", True), ("all_gather", True)], calls
        # the reference's training collective on the real backend: the fused autograd nodes under DistributedDataParallel
        # over RCCL (one rank: the all-reduce still runs as an RCCL kernel on the bucketed gradients; two ranks over gloo:
        # tests/test_multirank_gpu.py) -- gradients equal the un-wrapped module's, bit for bit for the weight tensors
        from torch.nn.parallel import DistributedDataParallel as DDP

        torch.manual_seed(1)
        module = render.TetraNerfModule(tr, len(pts), 64, 256, num_fine_samples=64, biased=True, gradient_scaling=True).to(device)
        bo, bd = scenes.outside_in_rays(1024, 9)
        bo, bd = torch.from_numpy(bo).to(device), torch.from_numpy(bd).to(device)
        target = torch.rand(len(bo), 3, device=device)

        def grads(m):
            for p_ in module.parameters():
                p_.grad = None
            torch.manual_seed(5)
            ((m(bo, bd)["rgb"] - target) ** 2).mean().backward()
            return {n: p_.grad.clone() for n, p_ in module.named_parameters()}

        plain = grads(module)
        wrapped = grads(DDP(module, device_ids=[device.index], find_unused_parameters=True))
        assert set(plain) == set(wrapped) and len(plain) == 13
        for n in plain:
            if n == "tetrahedra_field":      # float atomics
                assert float((plain[n] - wrapped[n]).abs().max()) <= 1e-5 * float(plain[n].abs().max()), n
            else:
                assert torch.equal(plain[n], wrapped[n]), n
        x = torch.arange(12, dtype=torch.float32, device=device).view(4, 3)
        y = torch.empty_like(x)
        dist.all_gather_into_tensor(y, x)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    finally:
        dist.destroy_process_group()
