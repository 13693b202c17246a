"""The `.th` tetrahedra file either side of the hot path.

Format (tetranerf/scripts/triangulate.py:68-75): a `torch.save`d dict
    {"cells": i32 [T,4], "vertices": f32 [V,3], "colors": u8 [V,4] (optional)}
read back by the model (tetranerf/nerfstudio/model.py:349-392): vertices go through the dataparser
transform ([x,1] @ T^T, then * scale), cells to int32, and -- with `initialize_colors` -- the field rows
1..3 are seeded from RGB (c*2/255 - 1) and row 0 from alpha."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Optional

import torch


def save_tetrahedra(path, vertices: torch.Tensor, cells: torch.Tensor, colors: Optional[torch.Tensor] = None) -> None:
    out = {"cells": cells.detach().cpu().int(), "vertices": vertices.detach().cpu().float()}
    if colors is not None:
        if colors.dtype != torch.uint8 or tuple(colors.shape) != (len(vertices), 4):
            raise ValueError("colors must be uint8 [num_vertices, 4]")
        out["colors"] = colors.detach().cpu()
    Path(path).absolute().parent.mkdir(parents=True, exist_ok=True)
    torch.save(out, str(path))


def load_tetrahedra(path, dataparser_transform: Optional[torch.Tensor] = None,
                    dataparser_scale: Optional[float] = None) -> Dict[str, torch.Tensor]:
    """-> {"vertices": f32 [V,3] (transformed like model.py:355-373), "cells": i32 [T,4], "colors": u8 [V,4]?}."""
    path = Path(path)
    if not path.exists():
        raise RuntimeError(f"Specified tetrahedra path {path} does not exist")
    t = torch.load(str(path), map_location=torch.device("cpu"))
    vertices = t["vertices"].float()
    if dataparser_transform is not None:
        hom = torch.cat((vertices, torch.ones_like(vertices[..., :1])), -1)
        vertices = hom @ dataparser_transform.T.to(vertices.dtype)
    if dataparser_scale is not None:
        vertices = vertices * dataparser_scale
    out = {"vertices": vertices.contiguous(), "cells": t["cells"].int().contiguous()}
    if "colors" in t:
        out["colors"] = t["colors"]
    return out


def init_field_from_colors(field: torch.Tensor, colors: torch.Tensor) -> None:
    """model.py:380-386: field [64,V]; rows 1..3 <- RGB in [-1,1], row 0 <- alpha in [-1,1]."""
    if colors.dtype != torch.uint8 or tuple(colors.shape) != (field.shape[1], 4):
        raise ValueError("colors must be uint8 [num_vertices, 4]")
    c = colors.float().to(field.device) * 2.0 / 255.0 - 1.0
    field.data[1:4, :] = c[:, :3].T
    field.data[0, :] = c[:, 3]
