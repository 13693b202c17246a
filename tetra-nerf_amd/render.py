"""Forward render path of the Tetra-NeRF model on top of the HIP ops (inference / evaluation).

Mirrors TetrahedraNerf.get_outputs (/root/reference/tetranerf/nerfstudio/model.py:520-662) for
the `tetra-nerf-original` evaluation configuration restricted to the coarse pass:

    trace_rays -> nears/fars (:531-544) -> uniform samples (nerfstudio UniformSampler, eval mode:
    bins = linspace(0,1,S+1), euclidean = near + bins*(far-near)) -> find_visited_cells (:560-567)
    -> interpolate_values (:569-573) -> mlp_base 64->128->128->128 ReLU (+ReLU out) (:414-455, 602-603)
    -> density head 128->1 + softplus, direction encoding (NeRFEncoding 3->27) ++ base -> mlp_head
    155->128 ReLU -> rgb head 128->3 + sigmoid (:605-621) -> weights = alpha * transmittance
    (RaySamples.get_weights) -> rgb over white background, accumulation, median depth (:632-662).

nerfstudio is not installed in this environment: layer shapes and activations are taken from the
call sites above plus nerfstudio 0.3.4's public API as recalled (SURVEY.md 8c caveat).  The
arithmetic below is therefore *our* definition; `render_reference` is its plain-PyTorch fp32
statement (runs on CPU tensors with any tracer-like object), `render` the GPU pipeline, and
`use_fused_mlp=True` swaps the PyTorch MLP for the fp32-MFMA HIP kernel (tn_mlp.hip).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

FIELD_DIM = 64
HIDDEN = 128
DIR_ENC = 27


def direction_encoding(d: torch.Tensor) -> torch.Tensor:
    """NeRFEncoding(in_dim=3, num_frequencies=4, min_freq_exp=0, max_freq_exp=4, include_input=True)
    (model.py:428-434): sin(2*pi*x*f) and sin(2*pi*x*f + pi/2) for f = 2**linspace(0,4,4), then x."""
    freqs = 2.0 ** torch.linspace(0.0, 4.0, 4, dtype=d.dtype, device=d.device)
    scaled = (2.0 * math.pi * d)[..., None] * freqs                      # [...,3,4]
    scaled = scaled.reshape(*d.shape[:-1], 12)
    enc = torch.sin(torch.cat([scaled, scaled + math.pi / 2.0], dim=-1))  # [...,24]
    return torch.cat([enc, d], dim=-1)                                    # [...,27]


class TetraMLP(torch.nn.Module):
    """The shallow MLP + heads of the model (model.py:414-455): default-initialised nn.Linear
    layers, exactly the parameter shapes a reference checkpoint holds."""

    def __init__(self, field_dim: int = FIELD_DIM, hidden: int = HIDDEN):
        super().__init__()
        self.base = torch.nn.ModuleList([torch.nn.Linear(field_dim, hidden), torch.nn.Linear(hidden, hidden),
                                         torch.nn.Linear(hidden, hidden)])
        self.density = torch.nn.Linear(hidden, 1)
        self.head = torch.nn.Linear(DIR_ENC + hidden, hidden)
        self.rgb = torch.nn.Linear(hidden, 3)

    def forward(self, feats: torch.Tensor, dirs: torch.Tensor):
        """feats [..., 64], dirs [..., 3] (per sample) -> sigma [..., 1], rgb [..., 3]."""
        x = feats
        for lin in self.base:
            x = torch.relu(lin(x))
        sigma = torch.nn.functional.softplus(self.density(x))
        h = torch.relu(self.head(torch.cat([direction_encoding(dirs), x], dim=-1)))
        rgb = torch.sigmoid(self.rgb(h))
        return sigma, rgb


def uniform_sample_bins(nears: torch.Tensor, fars: torch.Tensor, num_samples: int) -> torch.Tensor:
    """[R,S+1] euclidean bin edges of nerfstudio's UniformSampler in eval mode."""
    bins = torch.linspace(0.0, 1.0, num_samples + 1, dtype=nears.dtype, device=nears.device)[None]
    return bins * fars + (1.0 - bins) * nears


def composite(sigma: torch.Tensor, rgb: torch.Tensor, starts: torch.Tensor, ends: torch.Tensor,
              background: float = 1.0):
    """RaySamples.get_weights + RGBRenderer(white) + AccumulationRenderer + DepthRenderer(median).
    sigma [R,S,1], rgb [R,S,3], starts/ends [R,S,1]."""
    deltas = ends - starts
    dd = deltas * sigma
    alphas = 1.0 - torch.exp(-dd)
    trans = torch.cumsum(dd[..., :-1, :], dim=-2)
    trans = torch.cat([torch.zeros_like(trans[..., :1, :]), trans], dim=-2)
    weights = torch.nan_to_num(alphas * torch.exp(-trans))
    acc = weights.sum(-2)
    out_rgb = (weights * rgb).sum(-2) + background * (1.0 - acc)
    steps = (starts + ends) / 2.0
    cum = torch.cumsum(weights[..., 0], dim=-1)
    split = torch.full_like(cum[..., :1], 0.5)
    idx = torch.searchsorted(cum.contiguous(), split, side="left").clamp(0, steps.shape[-2] - 1)
    depth = torch.gather(steps[..., 0], -1, idx)
    return out_rgb, acc, depth, weights


def render_reference(tracer, interpolate_values, field: torch.Tensor, mlp: TetraMLP, origins: torch.Tensor,
                     directions: torch.Tensor, num_samples: int = 256, max_ray_triangles: int = 512,
                     far_plane: float = 1000.0) -> Dict[str, torch.Tensor]:
    """Plain-PyTorch statement of the coarse render path; `tracer` needs trace_rays /
    find_visited_cells returning tensors, `interpolate_values(vi, bc, field)` the gather."""
    out = tracer.trace_rays(origins.contiguous(), directions.contiguous(), max_ray_triangles)
    nv = out["num_visited_cells"]
    nears = out["hit_distances"][:, 0, 0][:, None]
    fars = torch.gather(out["hit_distances"][:, :, 1], 1, (nv[:, None].long() - 1).clamp_min(0))
    ray_mask = nv > 0
    R = origins.shape[0]
    rgb = torch.ones((R, 3), dtype=torch.float32, device=origins.device)
    acc = torch.zeros((R, 1), dtype=torch.float32, device=origins.device)
    depth = torch.full((R, 1), far_plane, dtype=torch.float32, device=origins.device)
    if bool(ray_mask.any()):
        edges = uniform_sample_bins(nears[ray_mask], fars[ray_mask], num_samples)
        starts, ends = edges[:, :-1, None], edges[:, 1:, None]
        dist = ((ends + starts) / 2).squeeze(-1).contiguous()
        traced = tracer.find_visited_cells(nv[ray_mask].contiguous(), out["visited_cells"][ray_mask].contiguous(),
                                           out["barycentric_coordinates"][ray_mask].contiguous(),
                                           out["hit_distances"][ray_mask].contiguous(),
                                           out["vertex_indices"][ray_mask].contiguous(), dist)
        feats = interpolate_values(traced["vertex_indices"], traced["barycentric_coordinates"], field)
        dirs = directions[ray_mask][:, None, :].expand(-1, num_samples, -1)
        sigma, col = mlp(feats, dirs)
        rgb_r, acc_r, depth_r, _ = composite(sigma, col, starts, ends)
        rgb[ray_mask] = rgb_r
        acc[ray_mask] = acc_r
        depth[ray_mask] = depth_r
    return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}


def mlp_weights(mlp: TetraMLP):
    """The 12 tensors tn_mlp_forward takes, in its order."""
    b = mlp.base
    return [b[0].weight, b[0].bias, b[1].weight, b[1].bias, b[2].weight, b[2].bias, mlp.density.weight,
            mlp.density.bias, mlp.head.weight, mlp.head.bias, mlp.rgb.weight, mlp.rgb.bias]


class TetraRenderer:
    """GPU render path: HIP tracer/matcher/gather (+ optionally the fused fp32-MFMA MLP and the
    composite kernel).  With `fused=False` the MLP and the composite run in PyTorch on the GPU."""

    def __init__(self, tracer, field: torch.Tensor, mlp: TetraMLP, num_samples: int = 256,
                 max_ray_triangles: int = 512, fused: bool = True, far_plane: float = 1000.0):
        from . import tetranerf_cpp_extension as cpp

        self.cpp = cpp
        self.tracer, self.field, self.mlp = tracer, field, mlp
        self.S, self.M, self.fused, self.far_plane = int(num_samples), int(max_ray_triangles), fused, far_plane

    @torch.no_grad()
    def render(self, origins: torch.Tensor, directions: torch.Tensor) -> Dict[str, torch.Tensor]:
        cpp, S = self.cpp, self.S
        if not self.fused:
            return render_reference(self.tracer, cpp.interpolate_values, self.field, self.mlp, origins, directions,
                                    S, self.M, self.far_plane)
        out = self.tracer.trace_rays(origins.contiguous(), directions.contiguous(), self.M)
        nv = out["num_visited_cells"]
        nears = out["hit_distances"][:, 0, 0][:, None]
        fars = torch.gather(out["hit_distances"][:, :, 1], 1, (nv[:, None].long() - 1).clamp_min(0))
        ray_mask = nv > 0
        R, dev = origins.shape[0], origins.device
        rgb = torch.ones((R, 3), dtype=torch.float32, device=dev)
        acc = torch.zeros((R, 1), dtype=torch.float32, device=dev)
        depth = torch.full((R, 1), self.far_plane, dtype=torch.float32, device=dev)
        idx = torch.nonzero(ray_mask)[:, 0]
        if idx.numel():
            edges = uniform_sample_bins(nears[idx], fars[idx], S).contiguous()
            dist = ((edges[:, 1:] + edges[:, :-1]) / 2).contiguous()
            traced = self.tracer.find_visited_cells(nv[idx].contiguous(), out["visited_cells"][idx].contiguous(),
                                                    out["barycentric_coordinates"][idx].contiguous(),
                                                    out["hit_distances"][idx].contiguous(),
                                                    out["vertex_indices"][idx].contiguous(), dist)
            # gather + MLP + heads in one kernel (no [64, n] feature buffer)
            sigma, col = cpp.mlp_forward_gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field,
                                                directions[idx].contiguous(), mlp_weights(self.mlp), S)
            rgb_r, acc_r, depth_r = cpp.composite(sigma.view(-1, S), col.view(-1, S, 3), edges)
            rgb[idx] = rgb_r
            acc[idx] = acc_r
            depth[idx] = depth_r
        return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}
