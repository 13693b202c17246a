"""nerfstudio.cameras.rays (0.3.4), restated: TensorDataclass semantics (batch shape = broadcast of every tensor field's
shape[:-1]; indexing applies to the batch dimensions), Frustums, RaySamples.get_weights, RayBundle.get_ray_samples."""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import Callable, Dict, Optional

import torch
from torch import Tensor


class TensorDataclass:
    """nerfstudio.utils.tensor_dataclass.TensorDataclass: `__post_init__` broadcasts all tensor fields to a common batch
    shape; `self[idx]` indexes the batch dimensions of every tensor field."""

    _shape: tuple

    def __post_init__(self) -> None:
        tensors = {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if isinstance(getattr(self, f.name), Tensor)}
        if not tensors:
            raise ValueError("TensorDataclass must have at least one tensor")
        batch_shape = torch.broadcast_shapes(*[v.shape[:-1] for v in tensors.values()])
        for k, v in tensors.items():
            object.__setattr__(self, k, v.broadcast_to((*batch_shape, v.shape[-1])))
        object.__setattr__(self, "_shape", tuple(batch_shape))

    def _apply(self, fn: Callable, dict_fn: Optional[Callable] = None):
        new = {}
        for f in dataclasses.fields(self):
            v = getattr(self, f.name)
            if isinstance(v, Tensor):
                new[f.name] = fn(v)
            elif isinstance(v, dict) and dict_fn is not None:
                new[f.name] = {k: dict_fn(x) if isinstance(x, Tensor) else x for k, x in v.items()}
        return dataclasses.replace(self, **new)

    def __getitem__(self, indices):
        if isinstance(indices, Tensor):
            return self._apply(lambda x: x[indices], lambda x: x[indices])
        if isinstance(indices, (int, slice, type(Ellipsis))):
            indices = (indices,)
        assert isinstance(indices, tuple)
        fn = lambda x: x[indices + (slice(None),)]   # noqa: E731
        return self._apply(fn, fn)

    def __len__(self) -> int:
        if len(self._shape) == 0:
            raise TypeError("len() of a 0-d tensor")
        return self._shape[0]

    @property
    def shape(self):
        return self._shape

    @property
    def size(self) -> int:
        n = 1
        for s in self._shape:
            n *= s
        return n

    def reshape(self, shape):
        if isinstance(shape, int):
            shape = (shape,)
        return self._apply(lambda x: x.reshape((*shape, x.shape[-1])), lambda x: x.reshape((*shape, x.shape[-1])))

    def flatten(self):
        return self.reshape((-1,))

    def to(self, device):
        return self._apply(lambda x: x.to(device), lambda x: x.to(device))


@dataclass(init=False)
class Frustums(TensorDataclass):
    origins: Tensor
    directions: Tensor
    starts: Tensor
    ends: Tensor
    pixel_area: Optional[Tensor]
    offsets: Optional[Tensor] = None

    def __init__(self, origins, directions, starts, ends, pixel_area=None, offsets=None):
        self.origins, self.directions, self.starts, self.ends = origins, directions, starts, ends
        self.pixel_area, self.offsets = pixel_area, offsets
        self.__post_init__()

    def get_positions(self) -> Tensor:
        pos = self.origins + self.directions * (self.starts + self.ends) / 2
        if self.offsets is not None:
            pos = pos + self.offsets
        return pos


@dataclass(init=False)
class RaySamples(TensorDataclass):
    frustums: Frustums
    camera_indices: Optional[Tensor] = None
    deltas: Optional[Tensor] = None
    spacing_starts: Optional[Tensor] = None
    spacing_ends: Optional[Tensor] = None
    spacing_to_euclidean_fn: Optional[Callable] = None
    metadata: Optional[Dict[str, Tensor]] = None
    times: Optional[Tensor] = None

    def __init__(self, frustums, camera_indices=None, deltas=None, spacing_starts=None, spacing_ends=None,
                 spacing_to_euclidean_fn=None, metadata=None, times=None):
        self.frustums, self.camera_indices, self.deltas = frustums, camera_indices, deltas
        self.spacing_starts, self.spacing_ends = spacing_starts, spacing_ends
        self.spacing_to_euclidean_fn, self.metadata, self.times = spacing_to_euclidean_fn, metadata, times
        # batch shape: broadcast of the frustums' batch shape and the tensor fields'
        tensors = {f.name: getattr(self, f.name) for f in dataclasses.fields(self) if isinstance(getattr(self, f.name), Tensor)}
        batch_shape = torch.broadcast_shapes(frustums.shape, *[v.shape[:-1] for v in tensors.values()])
        for k, v in tensors.items():
            object.__setattr__(self, k, v.broadcast_to((*batch_shape, v.shape[-1])))
        object.__setattr__(self, "_shape", tuple(batch_shape))

    def get_weights(self, densities: Tensor) -> Tensor:
        """weights = alpha_i * T_i  (nerfstudio/cameras/rays.py RaySamples.get_weights).  densities [..., S, 1]."""
        delta_density = self.deltas * densities
        alphas = 1 - torch.exp(-delta_density)
        transmittance = torch.cumsum(delta_density[..., :-1, :], dim=-2)
        transmittance = torch.cat(
            [torch.zeros((*transmittance.shape[:1], 1, 1), device=densities.device), transmittance], dim=-2)
        transmittance = torch.exp(-transmittance)
        weights = alphas * transmittance
        weights = torch.nan_to_num(weights)
        return weights


@dataclass
class RayBundle(TensorDataclass):
    origins: Tensor
    directions: Tensor
    pixel_area: Optional[Tensor] = None
    camera_indices: Optional[Tensor] = None
    nears: Optional[Tensor] = None
    fars: Optional[Tensor] = None
    metadata: Dict[str, Tensor] = dataclasses.field(default_factory=dict)
    times: Optional[Tensor] = None

    def get_ray_samples(self, bin_starts: Tensor, bin_ends: Tensor, spacing_starts: Optional[Tensor] = None,
                        spacing_ends: Optional[Tensor] = None, spacing_to_euclidean_fn: Optional[Callable] = None) -> RaySamples:
        deltas = bin_ends - bin_starts
        camera_indices = self.camera_indices[..., None] if self.camera_indices is not None else None
        shaped = self[..., None]
        frustums = Frustums(origins=shaped.origins, directions=shaped.directions, starts=bin_starts, ends=bin_ends,
                            pixel_area=shaped.pixel_area)
        return RaySamples(frustums=frustums, camera_indices=camera_indices, deltas=deltas, spacing_starts=spacing_starts,
                          spacing_ends=spacing_ends, spacing_to_euclidean_fn=spacing_to_euclidean_fn,
                          metadata=shaped.metadata, times=None if self.times is None else self.times[..., None])
