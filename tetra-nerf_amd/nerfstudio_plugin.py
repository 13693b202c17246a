"""Reference-side wiring of the fused MI355X kernels into `TetrahedraNerf` (the nerfstudio model of the reference).

With only the `tetranerf_cpp_extension` shim `ns-train tetra-nerf` runs the HIP tracer, matcher
and gather, but the MLP, the samplers and the renderers stay nerfstudio's PyTorch modules -- the fp32-MFMA MLP
(`tn_mlp_forward_gather`), the one-launch render (`tn_render_rays`) and the training adjoints (`tn_mlp_backward`,
`tn_composite_backward`) are never reached.  This module closes that gap without touching the reference's files:

    import tetranerf.nerfstudio.model as ref_model
    from tetranerf_amd.nerfstudio_plugin import install
    install(ref_model.TetrahedraNerf)            # get_outputs now runs TetraRenderer.render / render_train

or, without monkey-patching, `FusedTetrahedraNerf = make_fused_model_class(ref_model.TetrahedraNerf)` and point
`TetrahedraNerfConfig._target` at it (registration.py:20-67 builds the two method configs from that class).

What is mapped (reference: /root/reference/tetranerf/nerfstudio/model.py):

    get_outputs                                   :520-662   -> fused_get_outputs (below)
    mlp_base.layers[0..2]                         :436-441   -> w1,b1 / w2,b2 / w3,b3     [128,64] [128,128] [128,128]
    field_output_density.net                      :456       -> wd,bd                     [1,128]   (+ softplus)
    mlp_head.layers[0]  (in = [dir enc 27 | base 128]) :449-454 -> wh,bh                  [128,155]
    field_output_color.net                        :455       -> wr,br                     [3,128]   (+ sigmoid)
    tetrahedra_field                              :247-256   -> the [64,V] field (cached vertex-major shadow)
    sampler_uniform / TetrahedraSampler / sampler_pdf  :459-463,111-192 -> render.uniform/biased/pdf_sample_bins
    renderer_rgb(background) / accumulation / depth    :466-468  -> tn_composite (+ adjoint)
    GradientScaler                                :195-205,625-630 -> render.GradientScaler
    state dict keys                               :273-300   -> weights_from_state_dict

    appearance_embedding (appearance_embed_dim = E > 0) :437-447,608-620 -> the E extra columns of mlp_head collapse to a
                                                  per-RAY bias c = Wh[:, 155:] emb(camera) (training: the ray's camera; evaluation:
                                                  the mean embedding), made here in PyTorch ([R,E] x [E,128], differentiable) and
                                                  added by the kernels to the head pre-activation (`ray_head_bias`)

Fallback rule (the fused kernels hard-code the shipped architecture 64 -> 128^3 -> (27 + 128 [+ E]) -> 128 -> 3): any other
configuration -- `field_dim != 64`, `hidden_size != 128`, `num_density_layers != 3`, `num_color_layers != 1`,
`input_fourier_frequencies > 0`, a `background_color` that is not a constant on the call -- runs the
REFERENCE `get_outputs` unchanged (i.e. the HIP tracer / matcher / gather under nerfstudio's PyTorch MLP): same results as
the reference, just without the fused speed-up.  `fused_config_supported` states the rule; both method configs the
reference registers (`tetra-nerf-original`, `tetra-nerf`) are supported.

nerfstudio is not installed in this environment: the adapter is duck-typed (it only touches the attribute names listed
above) and is tested with stand-ins of nerfstudio's MLP / FieldHead / RayBundle (tests/golden/nerfstudio_standins.py).
"""
from __future__ import annotations

import sys
from typing import Dict, List, Tuple

import torch

# state-dict keys of the 12 kernel weight tensors, in kernel order (w1,b1,w2,b2,w3,b3,wd,bd,wh,bh,wr,br)
STATE_DICT_KEYS = (
    "mlp_base.layers.0.weight", "mlp_base.layers.0.bias",
    "mlp_base.layers.1.weight", "mlp_base.layers.1.bias",
    "mlp_base.layers.2.weight", "mlp_base.layers.2.bias",
    "field_output_density.net.weight", "field_output_density.net.bias",
    "mlp_head.layers.0.weight", "mlp_head.layers.0.bias",
    "field_output_color.net.weight", "field_output_color.net.bias",
)
FIELD_KEY = "tetrahedra_field"
_SHAPES = [(128, 64), (128,), (128, 128), (128,), (128, 128), (128,), (1, 128), (1,), (128, 155), (128,), (3, 128), (3,)]


def weights_from_state_dict(state_dict, prefix: str = "") -> Tuple[List[torch.Tensor], torch.Tensor]:
    """(the 12 weight tensors in kernel order, the [64,V] field) out of a reference checkpoint's model state dict
    (`prefix` = "_model." for a nerfstudio pipeline checkpoint: ckpt["pipeline"]).  Shapes are checked: a checkpoint
    of a non-default architecture raises."""
    ws = []
    for k, shp in zip(STATE_DICT_KEYS, _SHAPES):
        t = state_dict[prefix + k]
        if tuple(t.shape) != shp:
            raise RuntimeError(f"{prefix + k}: shape {tuple(t.shape)} is not the shipped architecture's {shp} "
                               "(fused kernels: 64 -> 128^3 -> (27+128) -> 128 -> 3)")
        ws.append(t)
    field = state_dict[prefix + FIELD_KEY]
    if field.dim() != 2 or field.shape[0] != 64:
        raise RuntimeError(f"{prefix + FIELD_KEY}: expected [64, V], got {tuple(field.shape)}")
    return ws, field


def weights_from_model(model) -> List[torch.Tensor]:
    """The 12 live parameters of a TetrahedraNerf (model.py:436-456), in kernel order."""
    b, h = model.mlp_base.layers, model.mlp_head.layers
    if len(b) != 3 or len(h) != 1:
        raise RuntimeError("fused kernels need num_density_layers = 3 and num_color_layers = 1")
    d, c = model.field_output_density.net, model.field_output_color.net
    return [b[0].weight, b[0].bias, b[1].weight, b[1].bias, b[2].weight, b[2].bias, d.weight, d.bias,
            h[0].weight, h[0].bias, c.weight, c.bias]


class ModelMLP:
    """What TetraRenderer needs of an MLP, served by the reference model's own modules: `fused_weights()` for the
    kernels, `__call__` (the modules themselves, model.py:602-621) for the unfused path."""

    def __init__(self, model):
        self.model = model

    def fused_weights(self):
        """The 12 kernel tensors.  With an appearance embedding mlp_head's weight is [128, 155 + E]: the kernels take its
        first 155 columns (a differentiable slice while a graph is recorded; cached per parameter version otherwise, so
        that evaluation does not re-pack the weights on every call); the other E act through `ray_head_bias`."""
        ws = weights_from_model(self.model)
        wh = ws[8]
        if wh.shape[1] != 155:
            if torch.is_grad_enabled() and wh.requires_grad:
                ws[8] = wh[:, :155].contiguous()
            else:
                hit = getattr(self, "_wh_main", None)
                if hit is None or hit[0] != (wh._version, wh.data_ptr()):
                    hit = self._wh_main = ((wh._version, wh.data_ptr()), wh.detach()[:, :155].contiguous())
                ws[8] = hit[1]
        return ws

    def ray_head_bias(self, ray_bundle):
        """[R, 128] per-ray bias of the head layer = Wh[:, 155:] applied to the ray's appearance embedding, exactly the
        embedded_appearance of model.py:608-620 (training: the embedding of the ray's camera; evaluation: the mean
        embedding for every ray); None without an appearance embedding."""
        m = self.model
        E = int(getattr(m.config, "appearance_embed_dim", 0))
        if E <= 0:
            return None
        wa = m.mlp_head.layers[0].weight[:, 155:155 + E]
        R = ray_bundle.origins.reshape(-1, 3).shape[0]
        if m.training:
            assert ray_bundle.camera_indices is not None
            emb = m.appearance_embedding(ray_bundle.camera_indices.reshape(-1))       # [R, E]
            return (emb @ wa.t()).contiguous()
        mean = m.appearance_embedding.weight.mean(dim=0)                              # [E]
        return (mean @ wa.t())[None, :].expand(R, -1).contiguous()

    def coarse_sigma(self, feats):
        """density of the coarse pass (model.py:577-581) -> [..., S]"""
        m = self.model
        return m.field_output_density(m.mlp_base(m.position_encoding(feats)))[..., 0]

    def __call__(self, feats, dirs):
        m = self.model
        x = m.mlp_base(m.position_encoding(feats))
        sigma = m.field_output_density(x)
        h = m.mlp_head(torch.cat([m.direction_encoding(dirs), x], dim=-1))
        return sigma, m.field_output_color(h)


def fused_config_supported(config) -> Tuple[bool, str]:
    """The fallback rule of the module docstring: (True, "") or (False, reason)."""
    g = lambda k, dflt: getattr(config, k, dflt)   # noqa: E731
    checks = (
        (g("field_dim", 64) == 64, "field_dim != 64"),
        (g("hidden_size", 128) == 128, "hidden_size != 128"),
        (g("num_density_layers", 3) == 3, "num_density_layers != 3"),
        (g("num_color_layers", 1) == 1, "num_color_layers != 1"),
        (g("input_fourier_frequencies", 0) == 0, "input_fourier_frequencies > 0"),
        (g("appearance_embed_dim", 0) >= 0, "appearance_embed_dim < 0"),
        (g("background_color", "white") not in ("random", "last_sample"), "background_color is not a constant"),
        (g("num_samples", 256) >= 1, "num_samples < 1"),
    )
    for ok, why in checks:
        if not ok:
            return False, why
    return True, ""


def resolve_background(model):
    """The colour nerfstudio's RGB renderer will blend with on THIS call, resolved the way the reference does it
    (`get_background_color`, model.py:504-518; `RGBRenderer.combine_rgb`): `renderers.BACKGROUND_COLOR_OVERRIDE` when a
    `background_color_override_context` is active (the viewer / exporters use it), else `renderer_rgb.background_color`,
    else `config.background_color`.  Returns a grey level (float), an (r, g, b) tuple, or None when the colour is not a
    constant ("random", "last_sample": the fused kernels do not implement them -> reference body)."""
    renderers = sys.modules.get("nerfstudio.model_components.renderers")
    bg = getattr(renderers, "BACKGROUND_COLOR_OVERRIDE", None) if renderers is not None else None
    if bg is None:
        bg = getattr(getattr(model, "renderer_rgb", None), "background_color", None)
    if bg is None:
        bg = getattr(model.config, "background_color", "white")
    if isinstance(bg, str):
        if bg in ("white", "black"):
            return 1.0 if bg == "white" else 0.0
        colors = sys.modules.get("nerfstudio.utils.colors")
        if colors is None or bg not in getattr(colors, "COLORS_DICT", {}):
            return None
        bg = colors.COLORS_DICT[bg]
    # A colour TENSOR is read back once per (object, version, storage): get_outputs runs per batch, and a device tensor
    # (nerfstudio keeps its colours on the host, an override may not) would cost a host synchronisation every time -- in the
    # middle of the sync-free training path.  The key changes when the tensor is replaced or written in place.
    key = (id(bg), getattr(bg, "_version", None), bg.data_ptr()) if isinstance(bg, torch.Tensor) else None
    if key is not None and _BG_CACHE.get("key") == key and _BG_CACHE.get("ref", lambda: None)() is bg:
        return _BG_CACHE["value"]
    t = torch.as_tensor(bg).detach().to(dtype=torch.float32, device="cpu").reshape(-1)   # colours live on the host in nerfstudio
    if t.numel() != 3:
        value = None
    else:
        r, g, b = t.tolist()
        value = r if r == g == b else (r, g, b)
    if key is not None:
        import weakref

        _BG_CACHE.update(key=key, value=value, ref=weakref.ref(bg))
    return value


_BG_CACHE: dict = {}


def _renderer_for(model, tracer):
    from . import render

    rd = getattr(model, "_tn_renderer", None)
    if rd is not None and rd.tracer is tracer and rd.field is model.tetrahedra_field:
        return rd
    cfg = model.config
    rd = render.TetraRenderer(
        tracer, model.tetrahedra_field, ModelMLP(model), num_samples=int(cfg.num_samples),
        max_ray_triangles=int(cfg.max_intersected_triangles), fused=True, far_plane=float(model.collider.far_plane),
        num_fine_samples=int(getattr(cfg, "num_fine_samples", 0)), biased=bool(getattr(cfg, "use_biased_sampler", False)),
        background=1.0)     # every call passes the colour resolve_background() found
    # plain attribute (not a registered submodule / buffer): nothing of it enters the state dict
    object.__setattr__(model, "_tn_renderer", rd)
    return rd


def fused_get_outputs(model, ray_bundle) -> Dict[str, torch.Tensor]:
    """Drop-in body of TetrahedraNerf.get_outputs (model.py:520-662) on the fused kernels: training mode = stratified
    samples + autograd through the HIP adjoints (`render_train`), evaluation = `render`; same output dictionary
    ("rgb", "accumulation", "depth", "ray_mask").  Unsupported configurations run the reference implementation."""
    ok, _why = fused_config_supported(model.config)
    ref = getattr(type(model), "_tn_reference_get_outputs", None)
    bg = resolve_background(model) if ok else None
    if ok and bg is None:
        ok, _why = False, "background colour is not a constant on this call"
    if not ok:
        if ref is None:
            raise RuntimeError(f"fused path unsupported ({_why}) and no reference get_outputs to fall back to")
        return ref(model, ray_bundle)
    if model.mlp_base is None:
        raise ValueError("populate_fields() must be called before get_outputs")
    tracer = model.get_tetrahedra_tracer()          # lazy mesh initialisation + structure build (model.py:394-407)
    rd = _renderer_for(model, tracer)
    o = ray_bundle.origins.reshape(-1, 3).contiguous()
    d = ray_bundle.directions.reshape(-1, 3).contiguous()
    hb = rd.mlp.ray_head_bias(ray_bundle)       # appearance embedding -> per-ray head bias (None without one)
    if model.training:
        # the reference's samplers stratify and its RGB renderer skips the clamp whenever `self.training` is set, with
        # or without autograd (model.py:169; nerfstudio RGBRenderer.forward); render_train skips the activation saves
        # when no graph is being recorded
        return rd.render_train(o, d, gradient_scaling=bool(getattr(model.config, "use_gradient_scaling", False)), background=bg,
                               ray_head_bias=hb)
    return rd.render(o, d, background=bg, ray_head_bias=hb)


def install(model_cls=None):
    """Monkey-patch `model_cls.get_outputs` (default: the reference's TetrahedraNerf, imported here -- needs nerfstudio)
    with `fused_get_outputs`; the original stays reachable as `_tn_reference_get_outputs` (fallback rule).  Idempotent.
    Returns the class."""
    if model_cls is None:
        from tetranerf.nerfstudio.model import TetrahedraNerf as model_cls   # noqa: N813  (reference package)
    if getattr(model_cls, "_tn_reference_get_outputs", None) is None:
        model_cls._tn_reference_get_outputs = model_cls.get_outputs
        model_cls.get_outputs = fused_get_outputs
    return model_cls


def uninstall(model_cls):
    ref = getattr(model_cls, "_tn_reference_get_outputs", None)
    if ref is not None:
        model_cls.get_outputs = ref
        model_cls._tn_reference_get_outputs = None
    return model_cls


def make_fused_model_class(base_cls):
    """Subclass of the reference model whose get_outputs runs the fused kernels (no monkey-patching): use it as
    `TetrahedraNerfConfig._target`."""

    class FusedTetrahedraNerf(base_cls):   # type: ignore[misc, valid-type]
        _tn_reference_get_outputs = base_cls.get_outputs

        def get_outputs(self, ray_bundle):
            return fused_get_outputs(self, ray_bundle)

    FusedTetrahedraNerf.__name__ = "Fused" + base_cls.__name__
    return FusedTetrahedraNerf


def load_reference_checkpoint(model, state_dict, prefix: str = ""):
    """Copy a reference checkpoint's parameters into a model through `copy_` (bumps the version counters the weight /
    field caches watch) after checking the architecture; returns the model."""
    ws, field = weights_from_state_dict(state_dict, prefix)
    with torch.no_grad():
        for dst, src in zip(weights_from_model(model), ws):
            dst.copy_(src)
        model.tetrahedra_field.copy_(field)
    return model
