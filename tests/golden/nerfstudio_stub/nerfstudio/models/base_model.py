"""nerfstudio.models.base_model (0.3.4), restated: ModelConfig defaults and the Model skeleton that
TetrahedraNerf.__init__ / populate_modules build on (model.py:218-229,409-411)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Type

import torch
from torch import nn

from nerfstudio.model_components.scene_colliders import NearFarCollider


@dataclass
class ModelConfig:
    _target: Type = field(default_factory=lambda: Model)
    enable_collider: bool = True
    collider_params: Dict[str, float] = field(default_factory=lambda: {"near_plane": 2.0, "far_plane": 6.0})
    loss_coefficients: Dict[str, float] = field(default_factory=lambda: {"rgb_loss_coarse": 1.0, "rgb_loss_fine": 1.0})
    eval_num_rays_per_chunk: int = 4096
    prompt: Any = None

    def setup(self, **kwargs) -> Any:
        return self._target(self, **kwargs)


class Model(nn.Module):
    config: ModelConfig

    def __init__(self, config: ModelConfig, scene_box=None, num_train_data: int = 1, **kwargs) -> None:
        super().__init__()
        self.config = config
        self.scene_box = scene_box
        self.render_aabb = None
        self.num_train_data = num_train_data
        self.kwargs = kwargs
        self.collider = None
        self.populate_modules()  # populate the modules
        self.callbacks = None
        # to keep track of which device the nn.Module is on
        self.device_indicator_param = nn.Parameter(torch.empty(0))

    @property
    def device(self):
        return self.device_indicator_param.device

    def populate_modules(self):
        if self.config.enable_collider:
            assert self.config.collider_params is not None
            self.collider = NearFarCollider(near_plane=self.config.collider_params["near_plane"],
                                            far_plane=self.config.collider_params["far_plane"])

    def forward(self, ray_bundle):
        if self.collider is not None:
            ray_bundle = self.collider(ray_bundle)
        return self.get_outputs(ray_bundle)

    def get_outputs(self, ray_bundle):
        raise NotImplementedError
