"""Synthetic stand-ins for the benchmark scenes (no datasets / network here).

Mesh = Delaunay tetrahedralisation of a point cloud, as the reference's offline step
produces (tetranerf/scripts/triangulate.py:12-76 -> CGAL Delaunay_triangulation_3,
src/triangulation.cpp:34-75).  CGAL is not available, so scipy.spatial.Delaunay (Qhull)
generates the cells; the result has the same structure: int32 [T,4] cells over
float32 [V,3] vertices, convex hull boundary, each interior face shared by 2 tets.

Ray generators: `pinhole_rays` restates generate_rays of the reference's tests
(tests/test_tetrahedra_tracer.py:23-59); the other two are the ray sets SURVEY.md §8(d)
defines for configs C4/C5.
"""
from __future__ import annotations

import math

import numpy as np


def delaunay_cells(points: np.ndarray) -> np.ndarray:
    from scipy.spatial import Delaunay

    tri = Delaunay(np.asarray(points, dtype=np.float64))
    return np.ascontiguousarray(tri.simplices.astype(np.int32))


def random_mesh(num_points: int, seed: int):
    """`num_points` uniform points in [0,1]^3 (fp32) and their Delaunay cells."""
    rng = np.random.default_rng(seed)
    pts = rng.random((num_points, 3)).astype(np.float32)
    return pts, delaunay_cells(pts)


def cube_mesh():
    """The 9-vertex / 12-tet cube of tests/test_tetrahedra_tracer.py:231-256."""
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1],
                    [1, 1, 1], [0.5, 0.5, 0.5]], np.float32)
    cells = np.array([[0, 1, 2, 8], [2, 1, 3, 8], [0, 1, 4, 8], [4, 1, 5, 8], [0, 2, 4, 8],
                      [4, 2, 6, 8], [4, 5, 6, 8], [5, 6, 7, 8], [2, 3, 6, 8], [3, 6, 7, 8],
                      [1, 3, 5, 8], [3, 5, 7, 8]], np.int32)
    return pts, cells


def _normalize(x):
    return x / np.linalg.norm(x, axis=-1, keepdims=True)


def pinhole_rays(width=800, height=800, eye=(0.0, 1.0, 0.0), lookat=(0.0, 0.0, 0.0),
                 up=(0.0, 0.0, 1.0), fov_y=45.0):
    """Pinhole camera rays, row-major over a linspace(0,1) pixel grid."""
    eye = np.asarray(eye, np.float32)
    lookat = np.asarray(lookat, np.float32)
    up = np.asarray(up, np.float32)
    W = lookat - eye
    wlen = np.linalg.norm(W)
    U = _normalize(np.cross(W, up))
    V = _normalize(np.cross(U, W))
    vlen = wlen * math.tan(0.5 * fov_y * math.pi / 180.0)
    V = V * vlen
    U = U * (vlen * (width / height))
    gx, gy = np.meshgrid(np.linspace(0, 1, width, dtype=np.float32),
                         np.linspace(0, 1, height, dtype=np.float32), indexing="ij")
    d = np.stack([gy, gx], -1).reshape(-1, 2)  # reversed(meshgrid(..., "ij"))
    d = 2.0 * d - 1.0
    dirs = _normalize(d[:, :1] * U[None] + d[:, 1:] * V[None] + W[None]).astype(np.float32)
    origins = np.repeat(eye[None], len(dirs), 0).astype(np.float32)
    return np.ascontiguousarray(origins), np.ascontiguousarray(dirs)


def outside_in_rays(n: int, seed: int, center=(0.5, 0.5, 0.5), radius=1.5):
    """Origins on a sphere around the unit cube, aimed at uniform targets in [0.2,0.8]^3."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    o = (np.asarray(center) + radius * v).astype(np.float32)
    tgt = (0.2 + 0.6 * rng.random((n, 3))).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def inside_out_rays(n: int, seed: int):
    """Origins uniform in [0.3,0.7]^3, directions uniform on the sphere."""
    rng = np.random.default_rng(seed)
    o = (0.3 + 0.4 * rng.random((n, 3))).astype(np.float32)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    return np.ascontiguousarray(o), np.ascontiguousarray(v.astype(np.float32))
