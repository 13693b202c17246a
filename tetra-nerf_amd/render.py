"""Forward render path of the Tetra-NeRF model on top of the HIP ops (inference / evaluation).

Mirrors TetrahedraNerf.get_outputs (/root/reference/tetranerf/nerfstudio/model.py:520-662) in
evaluation mode for both shipped configurations (`tetra-nerf-original`: uniform 256 + PDF 256;
`tetra-nerf`: biased TetrahedraSampler 128 + PDF 128, registration.py:55-57):

    trace_rays -> nears/fars (:531-544) -> coarse samples: nerfstudio UniformSampler (eval:
    bins = linspace(0,1,S+1), euclidean = near + bins*(far-near)) or the biased TetrahedraSampler
    (:111-192) -> find_visited_cells (:560-567) -> interpolate_values (:569-573)
    [num_fine_samples > 0 (:575-600): mlp_base + density head -> get_weights -> PDFSampler
     (include_original: S + S_fine + 1 samples) -> find_visited_cells -> interpolate_values]
    -> mlp_base 64->128->128->128 ReLU (+ReLU out) (:414-455, 602-603)
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
SYNC_FREE_MIN_HITS = 0.85     # render_train: below this (last known) fraction of hitting rays the batch is compacted instead
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


def stratified_bins(num_samples: int, t_rand: torch.Tensor) -> torch.Tensor:
    """Train-mode spacing bins of TetrahedraSampler / nerfstudio's UniformSampler (model.py:166-175): every edge of
    linspace(0, 1, S+1) is jittered between the centres of its two neighbouring bins.  t_rand: U[0,1) [R, S+1]."""
    bins = torch.linspace(0.0, 1.0, num_samples + 1, dtype=t_rand.dtype, device=t_rand.device)[None]
    centers = (bins[..., 1:] + bins[..., :-1]) / 2.0
    upper = torch.cat([centers, bins[..., -1:]], -1)
    lower = torch.cat([bins[..., :1], centers], -1)
    return lower + (upper - lower) * t_rand


def spacing_bins(num_samples: int, t_rand: Optional[torch.Tensor], dtype, device) -> torch.Tensor:
    """[1 or R, S+1] spacing bins of the coarse samplers before the map to distances (model.py:166-174): linspace in
    evaluation mode, jittered with `t_rand` [R,S+1] in training mode."""
    if t_rand is None:
        return torch.linspace(0.0, 1.0, num_samples + 1, dtype=dtype, device=device)[None]
    return stratified_bins(num_samples, t_rand)


def uniform_sample_bins(nears: torch.Tensor, fars: torch.Tensor, num_samples: int, t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R,S+1] euclidean bin edges of nerfstudio's UniformSampler (eval mode; train mode with `t_rand` [R,S+1])."""
    bins = spacing_bins(num_samples, t_rand, nears.dtype, nears.device)
    return bins * fars + (1.0 - bins) * nears


def coarse_samples(nears, fars, num_samples, biased=False, num_visited_cells=None, hit_distances=None, t_rand=None):
    """(euclidean edges [R,S+1], spacing edges [R,S+1]) of the model's coarse sampler as the reference hands them on to
    the PDF sampler and the GradientScaler (`spacing_starts / spacing_ends` of its RaySamples): the UniformSampler keeps
    the spacing bins it drew, the TetrahedraSampler re-derives them from the re-mapped distances (model.py:182)."""
    bins = spacing_bins(num_samples, t_rand, nears.dtype, nears.device)
    edges = bins * fars + (1.0 - bins) * nears
    if biased:
        edges = map_to_biased(num_visited_cells, hit_distances, edges)
        return edges, (edges - nears) / (fars - nears)
    return edges, bins.expand(nears.shape[0], -1)


def biased_sample_bins(nears: torch.Tensor, fars: torch.Tensor, num_samples: int, num_visited_cells: torch.Tensor,
                       hit_distances: torch.Tensor, t_rand: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[R,S+1] euclidean bin edges of the biased TetrahedraSampler in eval mode (model.py:111-192):
    the uniform edges are re-mapped so that every visited tetrahedron receives the same share of the
    samples, placed proportionally inside its [t_in, t_out] segment (the mapping stacks the segment
    lengths from the first entry point; negative lengths -- the cell -1 closing segments -- count 0)."""
    return map_to_biased(num_visited_cells, hit_distances, uniform_sample_bins(nears, fars, num_samples, t_rand))


def map_to_biased(num_visited_cells: torch.Tensor, hit_distances: torch.Tensor, uni: torch.Tensor) -> torch.Tensor:
    """map_from_real_distances_to_biased_with_bounds (model.py:111-122) without the in-place clamps."""
    nb = num_visited_cells.long()
    lengths = (hit_distances[..., 1] - hit_distances[..., 0]).clamp_min(0)
    start = hit_distances[..., 0, 0]
    end = torch.gather(hit_distances[..., 1], 1, (nb[:, None] - 1).clamp_min(0)).squeeze(-1)
    rest = (uni - start[:, None]) / (end - start)[:, None] * nb[:, None]
    intervals = rest.floor().clamp_max(nb[:, None] - 1).clamp_min(0)
    rest = rest - intervals
    intervals = intervals.long()
    cum = torch.cumsum(torch.cat((start[:, None], lengths), 1), 1)
    return torch.gather(cum, 1, intervals) + torch.gather(lengths, 1, intervals) * rest


def pdf_sample_bins(spacing_edges: torch.Tensor, weights: torch.Tensor, num_fine: int, nears: torch.Tensor,
                    fars: torch.Tensor, histogram_padding: float = 0.01, eps: float = 1e-5,
                    u_rand: Optional[torch.Tensor] = None, return_spacing: bool = False):
    """[R, S + num_fine + 2] euclidean bin edges of nerfstudio's PDFSampler in eval mode with
    include_original=True (model.py:463,584): inverse-CDF samples of the padded coarse weights at the
    num_fine+1 bin-centred quantiles, merged with the coarse edges and sorted, then mapped back with
    spacing_to_euclidean (x*far + (1-x)*near, model.py:177).  spacing_edges [R,S+1] in [0,1], weights [R,S]."""
    num_bins = num_fine + 1
    w = weights + histogram_padding
    wsum = w.sum(-1, keepdim=True)
    padding = torch.relu(eps - wsum)
    w = w + padding / w.shape[-1]
    wsum = wsum + padding
    pdf = w / wsum
    cdf = torch.min(torch.ones_like(pdf), torch.cumsum(pdf, dim=-1))
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], dim=-1)
    u = torch.linspace(0.0, 1.0 - (1.0 / num_bins), steps=num_bins, dtype=cdf.dtype, device=cdf.device)
    if u_rand is None:   # eval: bin-centred quantiles
        u = (u + 1.0 / (2 * num_bins)).expand(*cdf.shape[:-1], num_bins).contiguous()
    else:                # train_stratified: one uniform draw per quantile bin, u_rand U[0,1) [R, num_fine+1]
        u = (u + u_rand / num_bins).contiguous()
    inds = torch.searchsorted(cdf.contiguous(), u, side="right")
    last = spacing_edges.shape[-1] - 1
    below, above = (inds - 1).clamp(0, last), inds.clamp(0, last)
    cdf0, cdf1 = torch.gather(cdf, -1, below), torch.gather(cdf, -1, above)
    b0, b1 = torch.gather(spacing_edges, -1, below), torch.gather(spacing_edges, -1, above)
    t = torch.clip(torch.nan_to_num((u - cdf0) / (cdf1 - cdf0), 0), 0, 1)
    bins = b0 + t * (b1 - b0)
    bins, _ = torch.sort(torch.cat([spacing_edges, bins], -1), -1)
    edges = bins * fars + (1.0 - bins) * nears
    return (edges, bins) if return_spacing else edges


def ray_weights(sigma: torch.Tensor, edges: torch.Tensor) -> torch.Tensor:
    """RaySamples.get_weights on [R,S] densities and [R,S+1] edges."""
    dd = (edges[:, 1:] - edges[:, :-1]) * sigma
    trans = torch.cumsum(dd[:, :-1], dim=-1)
    trans = torch.cat([torch.zeros_like(trans[:, :1]), trans], dim=-1)
    return torch.nan_to_num((1.0 - torch.exp(-dd)) * torch.exp(-trans))


def background_tensor(background, device=None) -> torch.Tensor:
    """[3] fp32 tensor of a grey level (float) or an (r, g, b) triple."""
    if isinstance(background, torch.Tensor):
        return background.detach().reshape(3).to(device=device, dtype=torch.float32)
    if isinstance(background, (int, float)):
        background = (background,) * 3
    return torch.tensor([float(x) for x in background], dtype=torch.float32, device=device)


def composite(sigma: torch.Tensor, rgb: torch.Tensor, starts: torch.Tensor, ends: torch.Tensor,
              background=1.0, clamp: bool = False):
    """RaySamples.get_weights + RGBRenderer(background) + AccumulationRenderer + DepthRenderer(median).
    sigma [R,S,1], rgb [R,S,3], starts/ends [R,S,1].  background: grey level or (r, g, b); clamp = the RGB renderer's
    evaluation mode (nerfstudio RGBRenderer.forward when not training: nan_to_num of the colours, result clamped to [0, 1])."""
    if clamp:
        rgb = torch.nan_to_num(rgb)
    deltas = ends - starts
    dd = deltas * sigma
    alphas = 1.0 - torch.exp(-dd)
    trans = torch.cumsum(dd[..., :-1, :], dim=-2)
    trans = torch.cat([torch.zeros_like(trans[..., :1, :]), trans], dim=-2)
    weights = torch.nan_to_num(alphas * torch.exp(-trans))
    acc = weights.sum(-2)
    out_rgb = (weights * rgb).sum(-2) + background_tensor(background, rgb.device) * (1.0 - acc)
    if clamp:
        out_rgb = out_rgb.clamp(0.0, 1.0)
    steps = (starts + ends) / 2.0
    cum = torch.cumsum(weights[..., 0], dim=-1)
    split = torch.full_like(cum[..., :1], 0.5)
    idx = torch.searchsorted(cum.contiguous(), split, side="left").clamp(0, steps.shape[-2] - 1)
    depth = torch.gather(steps[..., 0], -1, idx)
    return out_rgb, acc, depth, weights


def median_margin(weights: torch.Tensor) -> torch.Tensor:
    """[R,1] distance of a ray's cumulative weights from the median threshold 0.5: the median depth of a ray is DECIDED
    (independent of round-off in the weights) when this exceeds the accumulated rounding error of the cumulative sum."""
    cum = torch.cumsum(weights[..., 0] if weights.dim() == 3 else weights, dim=-1)
    return (cum - 0.5).abs().min(dim=-1, keepdim=True).values


def render_reference(tracer, interpolate_values, field: torch.Tensor, mlp: TetraMLP, origins: torch.Tensor,
                     directions: torch.Tensor, num_samples: int = 256, max_ray_triangles: int = 512,
                     far_plane: float = 1000.0, num_fine_samples: int = 0, biased: bool = False,
                     background=1.0) -> Dict[str, torch.Tensor]:
    """Plain-PyTorch statement of the render path in EVALUATION mode (model.py:520-662 with `self.training == False`: the
    samplers do not jitter, the RGB renderer sanitises and clamps); `tracer` needs trace_rays / find_visited_cells
    returning tensors, `interpolate_values(vi, bc, field)` the gather.  Pinned by the reference's own `get_outputs`
    executed from its file (tests/test_reference_model.py)."""
    out = tracer.trace_rays(origins.contiguous(), directions.contiguous(), max_ray_triangles)
    nv = out["num_visited_cells"]
    nears = out["hit_distances"][:, 0, 0][:, None]
    fars = torch.gather(out["hit_distances"][:, :, 1], 1, (nv[:, None].long() - 1).clamp_min(0))
    ray_mask = nv > 0
    R = origins.shape[0]
    rgb = background_tensor(background, origins.device).expand(R, 3).contiguous()   # get_background_color (model.py:504-518,642)
    acc = torch.zeros((R, 1), dtype=torch.float32, device=origins.device)
    depth = torch.full((R, 1), far_plane, dtype=torch.float32, device=origins.device)
    margin = torch.full((R, 1), 0.5, dtype=torch.float32, device=origins.device)   # test aid: see median_margin
    if bool(ray_mask.any()):
        lists = [out[k][ray_mask].contiguous() for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates",
                                                         "hit_distances", "vertex_indices")]
        near_r, far_r = nears[ray_mask], fars[ray_mask]

        def features(edges):
            dist = ((edges[:, 1:] + edges[:, :-1]) / 2).contiguous()
            traced = tracer.find_visited_cells(*lists, dist)
            return interpolate_values(traced["vertex_indices"], traced["barycentric_coordinates"], field)

        edges, spacing = coarse_samples(near_r, far_r, num_samples, biased, lists[0], lists[3])
        feats = features(edges)
        if num_fine_samples > 0:
            if hasattr(mlp, "coarse_sigma"):     # an adapter around other modules (nerfstudio_plugin.ModelMLP)
                sigma_c = mlp.coarse_sigma(feats)
            else:
                x = feats
                for lin in mlp.base:
                    x = torch.relu(lin(x))
                sigma_c = torch.nn.functional.softplus(mlp.density(x))[..., 0]
            edges = pdf_sample_bins(spacing, ray_weights(sigma_c, edges), num_fine_samples, near_r, far_r)
            feats = features(edges)
        starts, ends = edges[:, :-1, None], edges[:, 1:, None]
        dirs = directions[ray_mask][:, None, :].expand(-1, edges.shape[1] - 1, -1)
        sigma, col = mlp(feats, dirs)
        rgb_r, acc_r, depth_r, w_r = composite(sigma, col, starts, ends, background=background, clamp=True)
        rgb[ray_mask] = rgb_r
        acc[ray_mask] = acc_r
        depth[ray_mask] = depth_r
        margin[ray_mask] = median_margin(w_r)
    return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask, "depth_margin": margin}


class GradientScaler(torch.autograd.Function):
    """Radiance-field gradient scaling of the `tetra-nerf` configuration (model.py:195-205, applied :625-630):
    identity forward; the gradients of colours and densities are multiplied by clamp(ray_dist^2, 0, 1)."""

    @staticmethod
    def forward(ctx, colors, sigmas, ray_dist):
        ctx.save_for_backward(ray_dist)
        return colors, sigmas, ray_dist

    @staticmethod
    def backward(ctx, grad_colors, grad_sigmas, grad_ray_dist):
        (ray_dist,) = ctx.saved_tensors
        scaling = torch.square(ray_dist).clamp(0, 1)
        return grad_colors * scaling, grad_sigmas * scaling, grad_ray_dist


class _FusedMlpFunction(torch.autograd.Function):
    """gather + MLP + heads as ONE autograd node: forward = tn_mlp_forward_gather_train (fp32; saves the layer inputs and the
    ReLU masks, 2.3 KB per sample), backward = tn_mlp_backward + tn_mlp_param_grads + tn_interpolate_values_backward (dX
    chain and weight gradients on the fp32 matrix cores, nothing recomputed).  Gradients flow to the field and the 12
    weight tensors."""

    @staticmethod
    def forward(ctx, vertex_indices, barycentric_coordinates, field, dirs, samples_per_ray, ray_head_bias, *weights):
        from . import tetranerf_cpp_extension as cpp

        ctx.has_bias = ray_head_bias is not None
        sigma, rgb, saved = cpp.mlp_forward_gather_train(vertex_indices, barycentric_coordinates, field, dirs, list(weights),
                                                         int(samples_per_ray), ray_head_bias=ray_head_bias)
        # the outputs go through save_for_backward (which knows how to hold a node's own outputs); `saved` must not
        # reference them: node -> saved -> output -> grad_fn -> node is a cycle no collector sees through, i.e. 5 GB
        # leaked per iteration
        saved.sigma = saved.rgb = None
        ctx.save_for_backward(vertex_indices, barycentric_coordinates, field, dirs, sigma, rgb, *weights)
        ctx.saved = saved
        return sigma, rgb

    @staticmethod
    def backward(ctx, d_sigma, d_rgb):
        from . import tetranerf_cpp_extension as cpp

        vi, bc, field, dirs, sigma, rgb, *weights = ctx.saved_tensors
        saved = ctx.saved          # (kept: a second backward through a retained graph reads the same activations)
        res = cpp.mlp_backward(saved, vi, bc, field, dirs, list(weights), sigma, rgb, d_sigma.contiguous(), d_rgb.contiguous(),
                               want_ray_head_grad=ctx.has_bias)
        grad_field, grads = res[0], res[1]
        # the per-ray head bias (appearance embedding): its gradient = per-ray sums of the head pre-activation's gradient
        return (None, None, grad_field, None, None, res[2] if ctx.has_bias else None, *grads)


class _FusedCompositeFunction(torch.autograd.Function):
    """get_weights + RGB / accumulation / median-depth renderers as one node (tn_composite / tn_composite_backward)."""

    @staticmethod
    def forward(ctx, sigma, rgb, edges, background):
        from . import tetranerf_cpp_extension as cpp

        ctx.save_for_backward(sigma, rgb, edges)
        ctx.background = background     # grey level or (r, g, b); training mode: the RGB renderer does not clamp
        out_rgb, acc, depth = cpp.composite(sigma.detach(), rgb.detach(), edges, ctx.background)
        ctx.mark_non_differentiable(depth)
        return out_rgb, acc, depth

    @staticmethod
    def backward(ctx, d_rgb, d_acc, _d_depth):
        from . import tetranerf_cpp_extension as cpp

        sigma, rgb, edges = ctx.saved_tensors
        d_sigma, d_col = cpp.composite_backward(sigma, rgb, edges, d_rgb, None if d_acc is None else d_acc.reshape(-1),
                                                ctx.background)
        return d_sigma, d_col, None, None


def mlp_weights(mlp):
    """The 12 tensors the fused kernels take, in their order: of a TetraMLP, or of any object that provides
    `fused_weights()` (the nerfstudio adapter: nerfstudio_plugin.ModelMLP)."""
    if hasattr(mlp, "fused_weights"):
        return list(mlp.fused_weights())
    b = mlp.base
    return [b[0].weight, b[0].bias, b[1].weight, b[1].bias, b[2].weight, b[2].bias, mlp.density.weight,
            mlp.density.bias, mlp.head.weight, mlp.head.bias, mlp.rgb.weight, mlp.rgb.bias]


class TetraRenderer:
    """GPU render path: HIP tracer/matcher/gather (+ optionally the fused fp32-MFMA MLP and the
    composite kernel).  With `fused=False` the MLP and the composite run in PyTorch on the GPU."""

    def __init__(self, tracer, field: torch.Tensor, mlp: TetraMLP, num_samples: int = 256,
                 max_ray_triangles: int = 512, fused: bool = True, far_plane: float = 1000.0,
                 num_fine_samples: int = 0, biased: bool = False, dense_tails: bool = False, fused_pass="auto",
                 mlp_mode: str = "fp32", background=1.0, cache_field: bool = True, device_samplers: bool = True,
                 interpolate_values=None, sync_free_train: bool = True, sync_free_min_hits: float = None):
        from . import tetranerf_cpp_extension as cpp

        # render_train without a host synchronisation (see there); False: compact the hitting rays with torch.nonzero
        self.sync_free_train = bool(sync_free_train)
        # the sync-free form pays for rays that miss (padded entries are computed and discarded): the hit fraction of a batch
        # is read back ASYNCHRONOUSLY (pinned buffer + event, never waited for) and, when the last known fraction is below
        # SYNC_FREE_MIN_HITS, the next batches take the compacting form until it recovers
        self._hits_pinned, self._hits_event, self._hit_fraction = None, None, 1.0
        self._hits_pending, self._hits_host_id, self._batch_id = None, 0, 0
        self.sync_free_min_hits = SYNC_FREE_MIN_HITS if sync_free_min_hits is None else float(sync_free_min_hits)

        self.cpp = cpp
        self.tracer, self.field, self.mlp = tracer, field, mlp
        # the gather of the UNFUSED statement (render_train(fused=False)): default = the product's autograd op; the CPU
        # tests pass the reference's einsum definition so that the statement runs next to the reference model's body
        self._interpolate_values = interpolate_values
        # arithmetic of the fused forward kernels, per renderer (not process-wide): "fp32" = exact fp32 MFMA chain (what
        # the parity tests pin), "bf16x3" = split-operand bf16 MFMA (opt-in; inference only -- the training forward is
        # always fp32: tn_mlp_forward_gather_train has no bf16x3 mode)
        self.mlp_mode = mlp_mode
        self.train_node_samples = 1 << 22      # render_train: samples per autograd node of the fused MLP (see there)
        # RGBRenderer background: grey level (1.0 white = default config, 0.0 black) or an (r, g, b) triple; render() /
        # render_train() take a per-call override (nerfstudio's BACKGROUND_COLOR_OVERRIDE, model.py:504-518)
        self.background = background if isinstance(background, (int, float)) else tuple(float(x) for x in background_tensor(background).tolist())
        # samplers as device kernels on the trace rows in place (tn_sample_coarse / tn_sample_pdf): a render is then
        # trace -> [sampler -> pass] x 2 with no PyTorch operator in between (False: the PyTorch statements above, ~15
        # small kernels per pass -- the parity definition, kept for tests and A/B)
        self.device_samplers = bool(device_samplers)
        if cache_field:
            # this renderer owns `field`: cached vertex-major shadow, refreshed per tensor version.  After a write through
            # `.data` call cpp.invalidate_field_cache(field) (see tetranerf_cpp_extension.register_field)
            cpp.register_field(field)
        self.S, self.M, self.fused, self.far_plane = int(num_samples), int(max_ray_triangles), fused, far_plane
        self.S_fine, self.biased = int(num_fine_samples), bool(biased)
        # the render path only reads the trace rows through num_visited_cells, so the constant tails of the
        # dense reference layout need not be written (non-materialising trace: 52 B per segment, not 52*M per ray)
        self.dense_tails = bool(dense_tails)
        # Everything after the trace as ONE persistent launch (tn_render_rays: samplers + match + gather + MLP + composite of both
        # passes; round 5) whenever its preconditions hold (fp32 arithmetic, device samplers, the per-wave LDS regions fit:
        # max_ray_triangles <= 2048).  True / "auto": use it; False: the chain of separate kernels (sampler, matcher, gather + MLP,
        # composite per pass), which takes the same device-side ray count -- neither form synchronises with the host.  Round 2-4's
        # per-pass fusion (tn_render_pass) interleaved match / composite with the MFMA layers and lost 4-6 % to the chain on
        # 65,536-ray chunks; the persistent kernel separates the stages in time inside one launch and runs the chain's own device
        # functions, so its frame is bit-identical to the chain's.
        self.fused_pass = fused_pass if fused_pass == "auto" else bool(fused_pass)

    def _trace(self, origins, directions):
        """trace_rays; compact rows (a PER-CALL flag of the op: the tracer may be shared with other threads) unless this
        renderer was asked for the dense reference rows."""
        o, d = origins.contiguous(), directions.contiguous()
        if not self.dense_tails and getattr(self.tracer, "supports_compact_rows", False):
            return self.tracer.trace_rays(o, d, self.M, compact_rows=True)
        return self.tracer.trace_rays(o, d, self.M)

    @staticmethod
    def _background_rows(R, bg, dev):
        """[R,3] rows of the background colour (get_background_color, model.py:504-518,642) without a host->device copy."""
        if isinstance(bg, (int, float)):
            return torch.full((R, 3), float(bg), dtype=torch.float32, device=dev)
        rows = torch.empty((R, 3), dtype=torch.float32, device=dev)
        for c in range(3):
            rows[:, c].fill_(float(bg[c]))
        return rows

    def _bg(self, background):
        if background is None:
            return self.background
        return background if isinstance(background, (int, float)) else tuple(float(x) for x in background_tensor(background).tolist())

    def _one_launch_ok(self, mode):
        """tn_render_rays' preconditions: device samplers, the per-wave LDS regions of its ray phases fit (either arithmetic
        since round 6: the bf16x3 mode runs x3::forward_group in the MLP phases)."""
        region = max(max(2 * self.M, 28) + self.S + 1, (max(2 * self.M, 3 * self.S + self.S_fine + 6) + 2 * self.S + self.S_fine + 2) if self.S_fine else 0) + 4
        return (self.fused_pass is not False and mode in ("fp32", "bf16x3") and self.device_samplers and 8 * 4 * region <= 160 * 1024
                and self.S + self.S_fine + 2 <= 8192)

    @torch.no_grad()
    def render(self, origins: torch.Tensor, directions: torch.Tensor, background=None, ray_head_bias=None) -> Dict[str, torch.Tensor]:
        """Evaluation-mode render (model.py:520-662 with `self.training == False`: samplers without jitter, RGB renderer
        with nan_to_num + clamp).  background: per-call override of the renderer's colour (grey level or (r, g, b)).
        ray_head_bias f32 [R, 128] (fused path only): per-ray vector added to mlp_head's pre-activation -- the appearance
        embedding's share of the head layer, Wh[:, 155:] emb (model.py:608-620), made by the caller.
        NO HOST SYNCHRONISATION (round 5): the reference compacts the hitting rays with boolean indexing (model.py:540-567),
        rounds 2-4 with torch.nonzero -- a device -> host round trip per chunk during which the GPU idles.  Here the hitting
        rays are compacted on the device (tn_compact_hits: their number stays there) and every kernel after the trace takes the
        address of that count."""
        cpp, S = self.cpp, self.S
        bg = self._bg(background)
        if not self.fused:
            if ray_head_bias is not None:
                raise RuntimeError("ray_head_bias is an input of the fused kernels; the PyTorch statement takes the model's own modules")
            return render_reference(self.tracer, cpp.interpolate_values, self.field, self.mlp, origins, directions,
                                    S, self.M, self.far_plane, self.S_fine, self.biased, background=bg)
        if not self.device_samplers:
            return self._render_host_compaction(origins, directions, bg, ray_head_bias)
        out = self._trace(origins, directions)
        nv = out["num_visited_cells"]
        ray_mask = nv > 0
        R, dev = origins.shape[0], origins.device
        rgb = self._background_rows(R, bg, dev)
        acc = torch.zeros((R, 1), dtype=torch.float32, device=dev)
        depth = torch.full((R, 1), self.far_plane, dtype=torch.float32, device=dev)
        res = {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}
        if R == 0:
            return res
        # the 26 KB trace rows of the hitting rays are NOT compacted (model.py:546-567 copies them with boolean indexing):
        # samplers, matcher and composite read them in place through the ray index
        lists = [out[k] for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances",
                                  "vertex_indices")]
        order, count = cpp.compact_hits(nv)          # hitting rays first, in ray order; their number stays on the device
        w = mlp_weights(self.mlp)
        mode = self.mlp_mode
        d = directions.contiguous()
        if self._one_launch_ok(mode):
            cpp.render_rays(lists, order, count, self.field, d, w, S, self.S_fine, self.biased, out=(rgb, acc, depth), background=bg,
                            clamp=True, ray_head_bias=ray_head_bias, mode=mode)
            return res
        # the chain of separate kernels: each is launched over R rows and processes the first `count` of them
        order_l = order.long()
        dirs_o = d.index_select(0, order_l)
        hb = None if ray_head_bias is None else ray_head_bias.index_select(0, order_l).contiguous()

        def locate(edges):
            dist = ((edges[:, 1:] + edges[:, :-1]) / 2).contiguous()
            return self.tracer.find_visited_cells(*lists, dist, ray_index=order, count=count)

        edges, near_far = cpp.sample_coarse(lists[0], lists[3], order, S, biased=self.biased, count=count)
        traced = locate(edges)
        if self.S_fine > 0:
            # coarse pass: gather + mlp_base + density head in one kernel, weights in one more
            sigma_c = cpp.mlp_forward_gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field,
                                             None, w, S, mode=mode, count=count)
            weights_c = cpp.composite(sigma_c.view(-1, S), None, edges, count=count)
            edges = cpp.sample_pdf(edges, weights_c, near_far, self.S_fine, count=count)
            traced = locate(edges)
            S = edges.shape[1] - 1
        # gather + MLP + heads in one kernel (no [64, n] feature buffer)
        sigma, col = cpp.mlp_forward_gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field,
                                            dirs_o, w, S, mode=mode, ray_head_bias=hb, count=count)
        cpp.composite(sigma.view(-1, S), col.view(-1, S, 3), edges, background=bg, clamp=True, out=(rgb, acc, depth), ray_index=order,
                      count=count)
        return res

    @torch.no_grad()
    def _render_host_compaction(self, origins, directions, bg, ray_head_bias=None):
        """render() with the PyTorch SAMPLER statements (device_samplers=False: the parity definition of tn_sample_coarse /
        tn_sample_pdf, ~15 small operators per pass) between the HIP kernels.  Sizes its work on the host (torch.nonzero), as
        the reference does; not the production path."""
        cpp, S = self.cpp, self.S
        out = self._trace(origins, directions)
        nv = out["num_visited_cells"]
        ray_mask = nv > 0
        R, dev = origins.shape[0], origins.device
        rgb = self._background_rows(R, bg, dev)
        acc = torch.zeros((R, 1), dtype=torch.float32, device=dev)
        depth = torch.full((R, 1), self.far_plane, dtype=torch.float32, device=dev)
        idx = torch.nonzero(ray_mask)[:, 0]
        mode = self.mlp_mode
        if idx.numel():
            lists = [out[k] for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances",
                                      "vertex_indices")]
            ridx = idx.to(torch.int32)
            w = mlp_weights(self.mlp)
            hb = None if ray_head_bias is None else ray_head_bias.index_select(0, idx).contiguous()

            def locate(edges):
                dist = ((edges[:, 1:] + edges[:, :-1]) / 2).contiguous()
                return self.tracer.find_visited_cells(*lists, dist, ray_index=ridx)

            # (rows of empty rays are unwritten without dense tails: nears / fars only of the hitting rays)
            near_r = out["hit_distances"][idx, 0, 0][:, None]
            far_r = out["hit_distances"][idx, (nv[idx].long() - 1), 1][:, None]
            if self.biased:
                edges = biased_sample_bins(near_r, far_r, S, lists[0][idx], lists[3][idx]).contiguous()
            else:
                edges = uniform_sample_bins(near_r, far_r, S).contiguous()
            traced = locate(edges)
            if self.S_fine > 0:
                sigma_c = cpp.mlp_forward_gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field,
                                                 None, w, S, mode=mode)
                weights_c = cpp.composite(sigma_c.view(-1, S), None, edges)
                spacing = (edges - near_r) / (far_r - near_r)
                edges = pdf_sample_bins(spacing, weights_c, self.S_fine, near_r, far_r).contiguous()
                traced = locate(edges)
                S = edges.shape[1] - 1
            sigma, col = cpp.mlp_forward_gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field,
                                                directions[idx].contiguous(), w, S, mode=mode, ray_head_bias=hb)
            rgb_r, acc_r, depth_r = cpp.composite(sigma.view(-1, S), col.view(-1, S, 3), edges, background=bg, clamp=True)
            rgb[idx] = rgb_r
            acc[idx] = acc_r
            depth[idx] = depth_r
        return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}

    def render_train(self, origins: torch.Tensor, directions: torch.Tensor, gradient_scaling: bool = False,
                     generator: Optional[torch.Generator] = None, rand: Optional[Dict[str, torch.Tensor]] = None,
                     fused: bool = True, capture: Optional[dict] = None, background=None,
                     ray_head_bias: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        """One training forward (TetrahedraNerf.get_outputs in training mode, model.py:520-662): stratified coarse samples
        (uniform or biased), optional PDF fine pass on the detached coarse weights (nerfstudio's PDFSampler detaches
        them), gather + MLP + heads, optional GradientScaler, weights and renderers (training mode: no clamp) --
        differentiable w.r.t. the field and the MLP parameters.  fused=True: the MLP and the composite are single autograd
        nodes backed by the HIP forward / adjoint kernels (without a graph being recorded -- `torch.no_grad()` -- the
        non-saving forward kernels run instead); fused=False: the plain PyTorch statement (autograd through nn.Linear
        etc.; with device_samplers=False not one HIP kernel above the tracer's ops), which is what the parity tests compare
        against and what tests/test_reference_model.py pins with the reference's own get_outputs.  `rand` may carry the
        uniform draws ("coarse" [r,S+1], "fine" [r,S_fine+1] over the hitting rays) so that two calls see the same
        samples; without it they are drawn from `generator` / torch's global generator in the reference's order and
        shapes (coarse first, then fine), so the same seed gives the reference body and this path the same draws."""
        cpp, S = self.cpp, self.S
        R, dev = origins.shape[0], origins.device
        rand = rand or {}
        # SYNC-FREE form (default for the fused path): the reference compacts the hitting rays with boolean indexing
        # (model.py:540-567: a device -> host synchronisation per call, like torch.nonzero here).  Training batches are
        # pixels of the object: nearly all of their rays hit, so the batch is processed at its full size R instead -- the
        # hitting rays first, in ray order (stable argsort of the miss flag), the tail padded with copies of the first entry,
        # whose (finite) results and gradients are masked out -- and nothing on the host ever waits for the ray count:
        # no gap in the launch stream between the trace and the samplers.  When every ray hits (the bench batch, the
        # parity tests) the stratified draws are the reference's, element for element; with misses they are the first
        # `count` rows of an [R, S+1] draw instead of an [r, S+1] draw -- the same distribution, another stream.
        sync_free = self.sync_free_train and fused and self.device_samplers and capture is None and not rand and R > 0
        self._batch_id += 1
        if self._hits_pending is not None and self._hits_event.query():
            # (whenever the copy has landed, whatever form THIS call takes: a renderer that alternates between forms must not
            #  decide on a stale fraction -- nor let the count of an older batch override one the host has seen since)
            issued, rays = self._hits_pending
            self._hits_pending = None
            if issued > self._hits_host_id:
                self._hit_fraction = float(self._hits_pinned[0]) / max(float(rays), 1.0)   # of an EARLIER batch
        sync_free = sync_free and self._hit_fraction >= self.sync_free_min_hits
        ridx = None
        with torch.no_grad():
            out = self._trace(origins, directions)
            nv = out["num_visited_cells"]
            ray_mask = nv > 0
            if sync_free:
                # ONE small kernel pair (tn_compact_hits) instead of a stable argsort of the miss flag (13 rocprim launches) +
                # where: order = the hitting rays in ray order, then the others; padded = order with the tail naming order[0]
                order32, count, padded = self.cpp.compact_hits(nv, want_padded=True)
                order = order32.long()
                valid = torch.arange(R, device=dev) < count     # (count is a one-element device tensor: no read-back)
                idx, ridx = padded.long(), padded
                # asynchronous read-back of this batch's hit count for the decision of a later one: the count the compaction
                # left on the device goes to pinned memory as it is (one copy; rounds 4's form built it from five small kernels)
                if self._hits_pinned is None:
                    self._hits_pinned = torch.zeros(1, dtype=torch.int32).pin_memory()
                    self._hits_event = torch.cuda.Event()
                if self._hits_pending is None:     # (no copy in flight: the buffer is free)
                    self._hits_pinned.copy_(count, non_blocking=True)
                    self._hits_event.record(torch.cuda.current_stream(dev))
                    self._hits_pending = (self._batch_id, R)
            else:
                idx = torch.nonzero(ray_mask)[:, 0]
                if self.sync_free_train:           # this form knows its count on the host anyway
                    self._hit_fraction, self._hits_host_id = idx.numel() / max(R, 1), self._batch_id
        bg = self._bg(background)
        rgb = self._background_rows(R, bg, dev)
        acc = torch.zeros((R, 1), dtype=torch.float32, device=dev)
        depth = torch.full((R, 1), self.far_plane, dtype=torch.float32, device=dev)
        if idx.numel() == 0:
            return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}
        lists = [out[k] for k in ("num_visited_cells", "visited_cells", "barycentric_coordinates", "hit_distances",
                                  "vertex_indices")]
        if ridx is None:
            ridx = idx.to(torch.int32)
        r = idx.numel()
        record = fused and torch.is_grad_enabled()
        spacing = None            # spacing bins of the final samples (exact only on the PyTorch sampler path)
        # (outside the no_grad block below: an adapter may hand over differentiable VIEWS of its parameters -- the first
        # 155 columns of an mlp_head widened by an appearance embedding, nerfstudio_plugin.ModelMLP.fused_weights)
        w = mlp_weights(self.mlp)
        with torch.no_grad():
            t_rand = rand.get("coarse")
            if t_rand is None:
                t_rand = torch.rand((r, S + 1), device=dev, generator=generator)
            if self.device_samplers:
                edges, near_far = cpp.sample_coarse(lists[0], lists[3], ridx, S, biased=self.biased, t_rand=t_rand.contiguous())
                near_r, far_r = near_far[:, 0:1], near_far[:, 1:2]
            else:
                near_r = out["hit_distances"][idx, 0, 0][:, None]
                far_r = out["hit_distances"][idx, (nv[idx].long() - 1), 1][:, None]
                near_far = torch.cat([near_r, far_r], 1).contiguous()
                edges, spacing = coarse_samples(near_r, far_r, S, self.biased, lists[0][idx], lists[3][idx], t_rand)
                edges = edges.contiguous()

            def locate(e):
                dist = ((e[:, 1:] + e[:, :-1]) / 2).contiguous()
                return self.tracer.find_visited_cells(*lists, dist, ray_index=ridx)

            traced = locate(edges)
            if self.S_fine > 0:
                if fused:
                    sigma_c = cpp.mlp_forward_gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field,
                                                     None, w, S).view(-1, S)
                    weights_c = cpp.composite(sigma_c, None, edges)
                else:       # model.py:577-582 in PyTorch
                    gather = self._interpolate_values or cpp.interpolate_values
                    feats_c = gather(traced["vertex_indices"], traced["barycentric_coordinates"], self.field)
                    if hasattr(self.mlp, "coarse_sigma"):
                        sigma_c = self.mlp.coarse_sigma(feats_c)
                    else:
                        x = feats_c
                        for lin in self.mlp.base:
                            x = torch.relu(lin(x))
                        sigma_c = torch.nn.functional.softplus(self.mlp.density(x))[..., 0]
                    weights_c = ray_weights(sigma_c, edges)
                u_rand = rand.get("fine")
                if u_rand is None:
                    u_rand = torch.rand((r, self.S_fine + 1), device=dev, generator=generator)
                if self.device_samplers:
                    edges = cpp.sample_pdf(edges, weights_c, near_far, self.S_fine, u_rand=u_rand.contiguous())
                    spacing = None
                else:
                    if spacing is None:
                        spacing = (edges - near_r) / (far_r - near_r)
                    edges, spacing = pdf_sample_bins(spacing, weights_c, self.S_fine, near_r, far_r, u_rand=u_rand, return_spacing=True)
                    edges = edges.contiguous()
                traced = locate(edges)
                S = edges.shape[1] - 1
        dirs = directions[idx].contiguous()
        # per-ray bias of the head layer (appearance embedding; fused path only): differentiable w.r.t. the caller's tensor
        hb = None if ray_head_bias is None else ray_head_bias.index_select(0, idx).contiguous()
        if hb is not None and not fused:
            raise RuntimeError("ray_head_bias is an input of the fused kernels; the PyTorch statement takes the model's own modules")
        vi, bc = traced["vertex_indices"], traced["barycentric_coordinates"]
        if capture is not None:   # the (non-differentiable) sample placement, for tests that restate the rest in float64
            capture.update(idx=idx, vertex_indices=vi, barycentric_coordinates=bc, edges=edges, dirs=dirs,
                           near=near_r, far=far_r, samples_per_ray=S)
        if record:
            # the node keeps 2.3 KB per sample from forward to backward (and its backward writes as much again): batches
            # beyond 2^22 samples (nerfstudio trains on 4096 rays) go through several nodes, one per block of rays
            rays_per_node = max(1, int(self.train_node_samples) // S)
            if r <= rays_per_node:
                sigma, col = _FusedMlpFunction.apply(vi, bc, self.field, dirs, S, hb, *w)
            else:
                parts = [_FusedMlpFunction.apply(vi[a:a + rays_per_node], bc[a:a + rays_per_node], self.field,
                                                 dirs[a:a + rays_per_node], S, None if hb is None else hb[a:a + rays_per_node], *w)
                         for a in range(0, r, rays_per_node)]
                sigma, col = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
            sigma, col = sigma.view(-1, S), col.view(-1, S, 3)
        elif fused:               # no graph: the plain forward kernel, nothing saved
            sigma, col = cpp.mlp_forward_gather(vi, bc, self.field, dirs, w, S, ray_head_bias=hb)
            sigma, col = sigma.view(-1, S), col.view(-1, S, 3)
        else:
            interpolate_values = self._interpolate_values
            if interpolate_values is None:
                from . import interpolate_values

            feats = interpolate_values(vi, bc, self.field)
            sg, col = self.mlp(feats, dirs[:, None, :].expand(-1, S, -1))
            sigma = sg[..., 0]
        if gradient_scaling and torch.is_grad_enabled():
            if spacing is None:
                spacing = (edges - near_r) / (far_r - near_r)
            ray_dist = (spacing[:, 1:] + spacing[:, :-1])[..., None]      # model.py:625-630
            col, sg, _ = GradientScaler.apply(col, sigma[..., None], ray_dist)
            sigma = sg[..., 0]
        if record:
            rgb_r, acc_r, depth_r = _FusedCompositeFunction.apply(sigma, col, edges, bg)
        elif fused:
            rgb_r, acc_r, depth_r = cpp.composite(sigma.contiguous(), col.contiguous(), edges, background=bg)
        else:
            rgb_r, acc_r, depth_r, _ = composite(sigma[..., None], col, edges[:, :-1, None], edges[:, 1:, None], background=bg)
        if sync_free:     # `order` is a permutation of the rays: every row is written once, padded entries get the miss values
            v = valid[:, None]
            rgb = rgb.index_copy(0, order, torch.where(v, rgb_r, rgb))
            acc = acc.index_copy(0, order, torch.where(v, acc_r.reshape(-1, 1), acc))
            depth = depth.index_copy(0, order, torch.where(v, depth_r.reshape(-1, 1).detach(), depth))
            return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}
        rgb = rgb.index_copy(0, idx, rgb_r)
        acc = acc.index_copy(0, idx, acc_r.reshape(-1, 1))
        depth = depth.index_copy(0, idx, depth_r.reshape(-1, 1).detach())
        return {"rgb": rgb, "accumulation": acc, "depth": depth, "ray_mask": ray_mask}


class TetraNerfModule(torch.nn.Module):
    """The model's trainable state -- `tetrahedra_field` [64,V] + the MLP -- and its forward as ONE nn.Module, i.e. the
    unit the reference wraps in `DistributedDataParallel(find_unused_parameters=True)` (pipeline.py:53-58: the model is
    replicated, every rank trains on its own rays, the gradients of the field and the MLP are all-reduced).
    `forward(origins, directions)` = TetraRenderer.render_train in training mode (the fused autograd nodes: their
    gradients reach the parameters through the autograd engine, so DDP's reducer hooks see them like any other) and
    TetraRenderer.render otherwise.  With a real nerfstudio the same role is played by the reference's TetrahedraNerf
    after nerfstudio_plugin.install()."""

    def __init__(self, tracer, num_vertices: int, num_samples: int = 256, max_ray_triangles: int = 512,
                 num_fine_samples: int = 256, biased: bool = False, gradient_scaling: bool = False, **renderer_kw):
        super().__init__()
        field = (torch.rand(FIELD_DIM, num_vertices) * 2 - 1) * 1e-4      # model.py:269-271
        field[1:4] = torch.rand(3, num_vertices) * 2 - 1                  # colours, model.py:379-386
        self.tetrahedra_field = torch.nn.Parameter(field)
        self.mlp = TetraMLP()
        self.gradient_scaling = bool(gradient_scaling)
        self._tracer = tracer
        self._renderer_args = (int(num_samples), int(max_ray_triangles))
        self._renderer_kw = dict(num_fine_samples=int(num_fine_samples), biased=bool(biased), **renderer_kw)
        self._renderer = None

    def renderer(self) -> "TetraRenderer":
        rd = self._renderer
        if rd is None or rd.field is not self.tetrahedra_field:      # (.to(device) replaces the parameter's storage)
            rd = TetraRenderer(self._tracer, self.tetrahedra_field, self.mlp, *self._renderer_args, fused=True, **self._renderer_kw)
            object.__setattr__(self, "_renderer", rd)
        return rd

    def forward(self, origins: torch.Tensor, directions: torch.Tensor) -> Dict[str, torch.Tensor]:
        rd = self.renderer()
        if self.training:
            return rd.render_train(origins, directions, gradient_scaling=self.gradient_scaling)
        return rd.render(origins, directions)
