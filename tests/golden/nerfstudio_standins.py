"""Duck-typed stand-ins for the nerfstudio pieces the reference model touches (nerfstudio is not installed here and is
absent from /root/reference): same attribute names and the arithmetic nerfstudio 0.3.4 documents for them.  TEST
INFRASTRUCTURE: used by tests/test_plugin*.py to drive tetra-nerf_amd/nerfstudio_plugin.py the way
tetranerf/nerfstudio/model.py would.

    MLP              nerfstudio.field_components.mlp.MLP          (.layers ModuleList, activation, out_activation)
    FieldHead        nerfstudio.field_components.field_heads      (.net Linear + activation): Density = Softplus, RGB = Sigmoid
    NeRFEncoding     nerfstudio.field_components.encodings        (sin / cos of 2*pi*x*2^f, include_input)
    RayBundle        nerfstudio.cameras.rays.RayBundle            (origins, directions [R,3])
    StandInTetrahedraNerf   the attributes of TetrahedraNerf that get_outputs reads (model.py:218-266,394-478) + a
                     `get_outputs` that restates model.py:520-662 in eval mode through render.render_reference
"""
from __future__ import annotations

import dataclasses
import importlib
import math
from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn


class MLP(nn.Module):
    def __init__(self, in_dim, num_layers, layer_width, out_dim=None, activation=nn.ReLU(), out_activation=None):
        super().__init__()
        out_dim = layer_width if out_dim is None else out_dim
        dims = [in_dim] + [layer_width] * (num_layers - 1) + [out_dim]
        self.layers = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(num_layers)])
        self.activation, self.out_activation = activation, out_activation

    def get_out_dim(self):
        return self.layers[-1].out_features

    def forward(self, x):
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if self.activation is not None and i < len(self.layers) - 1:
                x = self.activation(x)
        return x if self.out_activation is None else self.out_activation(x)


class FieldHead(nn.Module):
    def __init__(self, in_dim, out_dim, activation):
        super().__init__()
        self.net = nn.Linear(in_dim, out_dim)
        self.activation = activation

    def forward(self, x):
        return self.activation(self.net(x))


class NeRFEncoding(nn.Module):
    def __init__(self, in_dim, num_frequencies, min_freq_exp, max_freq_exp, include_input=True):
        super().__init__()
        self.in_dim, self.nf, self.lo, self.hi, self.include_input = in_dim, num_frequencies, min_freq_exp, max_freq_exp, include_input

    def get_out_dim(self):
        return self.in_dim * self.nf * 2 + (self.in_dim if self.include_input else 0)

    def forward(self, x):
        freqs = 2.0 ** torch.linspace(self.lo, self.hi, self.nf, dtype=x.dtype, device=x.device)
        scaled = ((2.0 * math.pi * x)[..., None] * freqs).reshape(*x.shape[:-1], -1)
        enc = torch.sin(torch.cat([scaled, scaled + math.pi / 2.0], dim=-1))
        return torch.cat([enc, x], dim=-1) if self.include_input else enc


@dataclasses.dataclass
class RayBundle:
    origins: torch.Tensor
    directions: torch.Tensor
    camera_indices: Optional[torch.Tensor] = None


@dataclasses.dataclass
class Config:   # the fields of TetrahedraNerfConfig (model.py:70-107) that matter here, with its defaults
    max_intersected_triangles: int = 512
    num_samples: int = 256
    num_fine_samples: int = 256
    use_biased_sampler: bool = False
    field_dim: int = 64
    num_color_layers: int = 1
    num_density_layers: int = 3
    hidden_size: int = 128
    input_fourier_frequencies: int = 0
    use_gradient_scaling: bool = False
    background_color: str = "white"
    appearance_embed_dim: int = 0


class StandInTetrahedraNerf(nn.Module):
    def __init__(self, config: Config, vertices: torch.Tensor, cells: torch.Tensor, far_plane: float = 1000.0):
        super().__init__()
        self.config = config
        self.register_buffer("tetrahedra_vertices", vertices.float().contiguous())
        self.register_buffer("tetrahedra_cells", cells.int().contiguous())
        field = torch.empty((config.field_dim, len(vertices)), dtype=torch.float32).uniform_(-1e-4, 1e-4)
        field[1:4] = torch.rand(3, len(vertices)) * 2 - 1          # model.py:379-386 (colours)
        self.tetrahedra_field = nn.Parameter(field)
        self.position_encoding = lambda x: x                        # input_fourier_frequencies = 0
        self.direction_encoding = NeRFEncoding(3, 4, 0.0, 4.0, True)
        self.mlp_base = MLP(config.field_dim, config.num_density_layers, config.hidden_size, out_activation=nn.ReLU())
        head_in = self.mlp_base.get_out_dim() + self.direction_encoding.get_out_dim() + config.appearance_embed_dim
        self.mlp_head = MLP(head_in, config.num_color_layers, config.hidden_size, out_activation=nn.ReLU())
        self.field_output_color = FieldHead(self.mlp_head.get_out_dim(), 3, nn.Sigmoid())
        self.field_output_density = FieldHead(self.mlp_base.get_out_dim(), 1, nn.Softplus())
        self.collider = SimpleNamespace(far_plane=far_plane)
        self._tetrahedra_tracer = None
        self.reference_calls = 0

    def get_tetrahedra_tracer(self):                                # model.py:394-407
        tn = importlib.import_module("tetra-nerf_amd")
        device = self.tetrahedra_field.device
        if device.type != "cuda":
            raise RuntimeError("Tetrahedra tracer is only supported on a CUDA device")
        if self._tetrahedra_tracer is not None and self._tetrahedra_tracer.device == device:
            return self._tetrahedra_tracer
        self._tetrahedra_tracer = tn.TetrahedraTracer(device)
        self._tetrahedra_tracer.load_tetrahedra(self.tetrahedra_vertices, self.tetrahedra_cells)
        return self._tetrahedra_tracer

    def get_outputs(self, ray_bundle):
        """model.py:520-662 (evaluation mode) through the ops of the shim + this model's own torch modules."""
        render = importlib.import_module("tetra-nerf_amd.render")
        plugin = importlib.import_module("tetra-nerf_amd.nerfstudio_plugin")
        tn = importlib.import_module("tetra-nerf_amd")
        self.reference_calls += 1
        cfg = self.config
        return render.render_reference(self.get_tetrahedra_tracer(), tn.interpolate_values, self.tetrahedra_field,
                                       plugin.ModelMLP(self), ray_bundle.origins, ray_bundle.directions, cfg.num_samples,
                                       cfg.max_intersected_triangles, self.collider.far_plane, cfg.num_fine_samples,
                                       cfg.use_biased_sampler, background={"white": 1.0, "black": 0.0}[cfg.background_color])
