"""nerfstudio.model_components.renderers (0.3.4), restated: RGBRenderer (combine_rgb, get_background_color,
BACKGROUND_COLOR_OVERRIDE + its context manager), AccumulationRenderer, DepthRenderer (median / expected)."""
import contextlib
from typing import Generator, Literal, Optional, Union

import torch
from torch import Tensor, nn

from nerfstudio.utils import colors

BackgroundColor = Union[Literal["random", "last_sample", "black", "white"], Tensor]
BACKGROUND_COLOR_OVERRIDE: Optional[Tensor] = None


@contextlib.contextmanager
def background_color_override_context(mode: Tensor) -> Generator[None, None, None]:
    """Context manager for setting background mode."""
    global BACKGROUND_COLOR_OVERRIDE
    old_background_color = BACKGROUND_COLOR_OVERRIDE
    try:
        BACKGROUND_COLOR_OVERRIDE = mode
        yield
    finally:
        BACKGROUND_COLOR_OVERRIDE = old_background_color


class RGBRenderer(nn.Module):
    def __init__(self, background_color: BackgroundColor = "random") -> None:
        super().__init__()
        self.background_color: BackgroundColor = background_color

    @classmethod
    def combine_rgb(cls, rgb, weights, background_color: BackgroundColor = "random", ray_indices=None, num_rays=None):
        assert ray_indices is None and num_rays is None, "packed samples (nerfacc) are not part of this stub"
        comp_rgb = torch.sum(weights * rgb, dim=-2)
        accumulated_weight = torch.sum(weights, dim=-2)
        if BACKGROUND_COLOR_OVERRIDE is not None:
            background_color = BACKGROUND_COLOR_OVERRIDE
        if isinstance(background_color, str) and background_color == "random":
            # If background color is random, the predicted color is returned without blending,
            # as if the background color was black.
            return comp_rgb
        elif isinstance(background_color, str) and background_color == "last_sample":
            # Note, this is only supported for non-packed samples.
            background_color = rgb[..., -1, :]
        background_color = cls.get_background_color(background_color, shape=comp_rgb.shape, device=comp_rgb.device)
        assert isinstance(background_color, torch.Tensor)
        comp_rgb = comp_rgb + background_color * (1.0 - accumulated_weight)
        return comp_rgb

    @classmethod
    def get_background_color(cls, background_color: BackgroundColor, shape, device) -> Tensor:
        assert not (isinstance(background_color, str) and background_color in {"last_sample", "random"})
        assert shape[-1] == 3, "Background color must be RGB."
        if BACKGROUND_COLOR_OVERRIDE is not None:
            background_color = BACKGROUND_COLOR_OVERRIDE
        if isinstance(background_color, str) and background_color in colors.COLORS_DICT:
            background_color = colors.COLORS_DICT[background_color]
        assert isinstance(background_color, Tensor)
        # Ensure correct shape
        return background_color.expand(shape).to(device)

    def forward(self, rgb, weights, ray_indices=None, num_rays=None, background_color: Optional[BackgroundColor] = None):
        if background_color is None:
            background_color = self.background_color
        if not self.training:
            rgb = torch.nan_to_num(rgb)
        rgb = self.combine_rgb(rgb, weights, background_color=background_color, ray_indices=ray_indices, num_rays=num_rays)
        if not self.training:
            torch.clamp_(rgb, min=0.0, max=1.0)
        return rgb


class AccumulationRenderer(nn.Module):
    @classmethod
    def forward(cls, weights, ray_indices=None, num_rays=None):
        assert ray_indices is None and num_rays is None
        accumulation = torch.sum(weights, dim=-2)
        return accumulation


class DepthRenderer(nn.Module):
    def __init__(self, method: Literal["median", "expected"] = "median") -> None:
        super().__init__()
        self.method = method

    def forward(self, weights, ray_samples, ray_indices=None, num_rays=None):
        if self.method == "median":
            steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
            assert ray_indices is None and num_rays is None
            cumulative_weights = torch.cumsum(weights[..., 0], dim=-1)  # [..., num_samples]
            split = torch.ones((*weights.shape[:-2], 1), device=weights.device) * 0.5  # [..., 1]
            median_index = torch.searchsorted(cumulative_weights, split, side="left")  # [..., 1]
            median_index = torch.clamp(median_index, 0, steps.shape[-2] - 1)  # [..., 1]
            median_depth = torch.gather(steps[..., 0], dim=-1, index=median_index)  # [..., 1]
            return median_depth
        if self.method == "expected":
            eps = 1e-10
            steps = (ray_samples.frustums.starts + ray_samples.frustums.ends) / 2
            depth = torch.sum(weights * steps, dim=-2) / (torch.sum(weights, -2) + eps)
            depth = torch.clip(depth, steps.min(), steps.max())
            return depth
        raise NotImplementedError(f"Method {self.method} not implemented")
