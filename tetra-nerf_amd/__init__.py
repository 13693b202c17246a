"""tetra-nerf_amd: the ray -> tetrahedra hot path of Tetra-NeRF on AMD MI355X (gfx950).

Python surface = what `tetranerf.utils.extension` exposes in the reference
(/root/reference/tetranerf/utils/extension/__init__.py:23-26,67-73): `TetrahedraTracer`,
`triangulate`, `gather_uint32`, `scatter_ema_uint32_`, `interpolate_values`
(differentiable w.r.t. the field), `add_barycentrics_grad`; the raw module is `cpp`
(same role as `tetranerf.cpp`, tetranerf/__init__.py:1).

The directory name contains a hyphen, so import it with
    importlib.import_module("tetra-nerf_amd")
or through the alias module `tetranerf_amd` at the repository root.
"""
from __future__ import annotations

import torch

from . import tetranerf_cpp_extension as cpp

__version__ = "0.1.0"

TetrahedraTracer = cpp.TetrahedraTracer
triangulate = cpp.triangulate
gather_uint32 = cpp.gather_uint32
scatter_ema_uint32_ = cpp.scatter_ema_uint32


class _GatherField(torch.autograd.Function):
    """field[64,V] -> per-sample features; gradient flows to the field only, exactly like
    _InterpolateValuesFunction (extension/__init__.py:29-42): no gradient for indices or
    barycentrics (py_binding.cpp:354 leaves that as a TODO)."""

    @staticmethod
    def forward(ctx, vertex_indices, barycentric_coordinates, field):
        ctx.save_for_backward(vertex_indices, barycentric_coordinates, field)
        return cpp.interpolate_values(vertex_indices, barycentric_coordinates, field)

    @staticmethod
    def backward(ctx, grad_out):
        vertex_indices, barycentric_coordinates, field = ctx.saved_tensors
        grad_field = cpp.interpolate_values_backward(
            vertex_indices, barycentric_coordinates, field, grad_out.contiguous())
        return None, None, grad_field


def interpolate_values(vertex_indices, barycentric_coordinates, field):
    return _GatherField.apply(vertex_indices, barycentric_coordinates, field)


class _BarycentricsGrad(torch.autograd.Function):
    """Identity on the barycentrics that routes their gradient to the tetrahedron vertices and
    the query points (role of _BarycentricsGradFunction, extension/__init__.py:45-68; unused
    by the model).  With lambda_1..3 the non-leading barycentrics of p in (v0..v3):
        p - v0 = T^T lambda,  T = rows (v_k - v0)   =>   d lambda = T^-T (dp - sum_k w_k dv_k)
    with w = (1 - sum lambda, lambda)."""

    @staticmethod
    def forward(ctx, barycentrics, vertices, points):
        ctx.save_for_backward(barycentrics, vertices)
        return barycentrics

    @staticmethod
    def backward(ctx, grad_bary):
        barycentrics, vertices = ctx.saved_tensors
        need_v, need_p = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        grad_v = grad_p = None
        if need_v or need_p:
            edges = vertices[..., 1:, :] - vertices[..., :1, :]
            m = torch.linalg.solve(edges, grad_bary)
            if need_p:
                grad_p = m
            if need_v:
                w = torch.cat([1.0 - barycentrics.sum(-1, keepdim=True), barycentrics], -1)
                grad_v = -(w.unsqueeze(-1) * m.unsqueeze(-2))
        return grad_bary, grad_v, grad_p


def add_barycentrics_grad(barycentrics, vertices, points):
    return _BarycentricsGrad.apply(barycentrics, vertices, points)


__all__ = ["TetrahedraTracer", "triangulate", "gather_uint32", "scatter_ema_uint32_",
           "interpolate_values", "add_barycentrics_grad", "cpp"]
