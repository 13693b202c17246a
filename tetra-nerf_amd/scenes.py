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


# ---------------------------------------------------------------------------------------------------------
# Adversarial point clouds (parity tests): what real Tetra-NeRF inputs look like -- COLMAP clouds are
# clustered on surfaces, near-coplanar, and get jittered copies (tetranerf/scripts/triangulate.py:36-55) --
# and what breaks walk-style traversals: cospherical points (Qhull slivers / zero-volume tets / ties in t),
# thin shells, near-duplicate points.

def _mesh_of(pts):
    pts = np.ascontiguousarray(pts, dtype=np.float32)
    return pts, delaunay_cells(pts)


def grid_mesh(n: int = 16, jitter: float = 0.0, seed: int = 5):
    """n^3 lattice in [0,1]^3 (jitter 0: every cube's 8 corners are cospherical -> flat slivers, coplanar faces,
    rays through lattice edges/vertices tie exactly); jitter > 0 perturbs each coordinate uniformly."""
    g = np.linspace(0.0, 1.0, n)
    pts = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    if jitter > 0:
        pts = pts + np.random.default_rng(seed).uniform(-jitter, jitter, pts.shape)
    return _mesh_of(pts)


def shells_mesh(n: int = 6000, gap: float = 1e-3, seed: int = 6):
    """Points on two concentric spheres `gap` apart (plus a few interior points): the band between the shells
    is filled with needle / sliver tetrahedra whose faces are nearly parallel to tangential rays."""
    rng = np.random.default_rng(seed)
    v = rng.normal(size=(n, 3))
    v /= np.linalg.norm(v, axis=-1, keepdims=True)
    r = np.where(np.arange(n) % 2 == 0, 0.45, 0.45 - gap)[:, None]
    pts = 0.5 + r * v
    inner = 0.5 + 0.2 * (rng.random((n // 20, 3)) - 0.5)
    return _mesh_of(np.concatenate([pts, inner], 0))


def near_duplicates_mesh(n: int = 4000, spacing: float = 1e-7, seed: int = 7):
    """Uniform cloud in which every fourth point has a twin `spacing` away (a few fp32 ulps): tetrahedra with
    edges of ~1e-7, i.e. crossings far below the 1e-6 tie window of the pairing stage."""
    rng = np.random.default_rng(seed)
    pts = rng.random((n, 3)).astype(np.float32).astype(np.float64)
    twins = pts[::4] + spacing * rng.choice([-1.0, 1.0], size=(len(pts[::4]), 3))
    allp = np.concatenate([pts, twins], 0).astype(np.float32)
    allp = np.unique(allp, axis=0)  # exact fp32 duplicates would make Qhull drop vertices silently
    rng.shuffle(allp)
    return _mesh_of(allp)


def colmap_like_mesh(n: int = 12000, ratio: float = 0.5, seed: int = 8):
    """Surface-clustered cloud + jittered copies as scripts/triangulate.py:36-55 makes them: points on a few
    planes / a sphere with small noise, then `ratio`*n copies offset by |N(s, s/2)| in a random direction
    (s = a spacing estimate)."""
    rng = np.random.default_rng(seed)
    k = n // 4
    plane1 = np.c_[rng.random((k, 2)), 0.3 + 1e-4 * rng.normal(size=k)]
    plane2 = np.c_[rng.random(k), 0.7 + 1e-4 * rng.normal(size=k), rng.random(k)]
    v = rng.normal(size=(k, 3)); v /= np.linalg.norm(v, axis=-1, keepdims=True)
    sphere = 0.5 + 0.25 * v * (1 + 1e-4 * rng.normal(size=(k, 1)))
    blob = 0.5 + 0.05 * rng.normal(size=(n - 3 * k, 3))
    pts = np.concatenate([plane1, plane2, sphere, blob], 0)
    s = 1.0 / np.cbrt(n)
    m = int(n * ratio)
    base = pts[rng.choice(len(pts), m, replace=True)]
    off = rng.normal(size=(m, 3)); off /= np.linalg.norm(off, axis=-1, keepdims=True)
    off *= np.abs(rng.normal(s, 0.5 * s, size=(m, 1)))
    return _mesh_of(np.concatenate([pts, base + off], 0))


def dense_core_mesh(n: int = 16000, core_frac: float = 0.7, core_sigma: float = 0.03, seed: int = 21):
    """Uniform cloud with a Gaussian core holding `core_frac` of the points (a COLMAP cloud's dense object in a sparse
    scene): rays through the core cross several times the ~3.45 T^(1/3) faces of a uniform mesh of the same size --
    what the mesh-size heuristics of the tracer (speculative tail fill, small-batch LDS arrays) must survive.
    Defaults: 107k tets; rays aimed at the core (`core_rays`) cross 218 faces in the median, 254+ in 1 % of the cases."""
    rng = np.random.default_rng(seed)
    k = int(n * core_frac)
    core = 0.5 + core_sigma * rng.normal(size=(k, 3))
    rest = rng.random((n - k, 3))
    pts = np.unique(np.concatenate([core, rest], 0).astype(np.float32), axis=0)
    rng.shuffle(pts)
    return _mesh_of(pts)


def core_rays(n: int, seed: int, spread: float = 0.02, radius: float = 1.5):
    """Origins on a sphere around the unit cube, aimed at N(0.5, spread) targets: through the core of dense_core_mesh."""
    o, _ = outside_in_rays(n, seed, radius=radius)
    rng = np.random.default_rng(seed + 1)
    d = (0.5 + spread * rng.normal(size=(n, 3))) - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    return np.ascontiguousarray(o), np.ascontiguousarray(d)


def needle_mesh(n: int = 40000, width: float = 0.05, background: int = 2000, seed: int = 22):
    """`n` points in the needle [0,1] x (0.5 +- width/2)^2 plus a sparse background: rays along the needle
    (`needle_rays`) cross ~1000 faces at the defaults (n = 20000: ~820) -- the max_ray_triangles > 512 cases."""
    rng = np.random.default_rng(seed)
    core = np.c_[rng.random(n), 0.5 + width * (rng.random(n) - 0.5), 0.5 + width * (rng.random(n) - 0.5)]
    pts = np.unique(np.concatenate([core, rng.random((background, 3))], 0).astype(np.float32), axis=0)
    rng.shuffle(pts)
    return _mesh_of(pts)


def needle_rays(n: int, seed: int, width: float = 0.05):
    rng = np.random.default_rng(seed)
    a = np.c_[np.full(n, -0.5), 0.5 + 0.8 * width * (rng.random(n) - 0.5), 0.5 + 0.8 * width * (rng.random(n) - 0.5)]
    b = np.c_[np.full(n, 1.5), 0.5 + 0.8 * width * (rng.random(n) - 0.5), 0.5 + 0.8 * width * (rng.random(n) - 0.5)]
    d = b - a
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return np.ascontiguousarray(a.astype(np.float32)), np.ascontiguousarray(d.astype(np.float32))


def vertex_to_vertex_rays(pts: np.ndarray, n: int, seed: int, extend: float = 1.0):
    """Rays that pass exactly (up to fp32 rounding of the direction) through two mesh vertices a -> b, started
    `extend` before a: they graze edges and vertices of many tetrahedra (zero edge functions, ties)."""
    rng = np.random.default_rng(seed)
    a = pts[rng.integers(0, len(pts), n)].astype(np.float64)
    b = pts[rng.integers(0, len(pts), n)].astype(np.float64)
    d = b - a
    nz = np.linalg.norm(d, axis=-1) > 1e-12
    d[~nz] = (1.0, 0.0, 0.0)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    o = a - extend * d
    return np.ascontiguousarray(o.astype(np.float32)), np.ascontiguousarray(d.astype(np.float32))


def orbit_rays(width: int, height: int, views: int, center=(0.0, 0.0, 0.0), radius: float = 1.0, fov_y: float = 45.0):
    """`views` pinhole cameras on a tilted orbit around `center`, all looking at it."""
    os_, ds_ = [], []
    for k in range(views):
        a = 2.0 * math.pi * k / views
        eye = (center[0] + radius * math.cos(a), center[1] + radius * math.sin(a), center[2] + 0.35 * radius * math.sin(2 * a + 0.5))
        o, d = pinhole_rays(width, height, eye=eye, lookat=center, fov_y=fov_y)
        os_.append(o); ds_.append(d)
    return np.ascontiguousarray(np.concatenate(os_, 0)), np.ascontiguousarray(np.concatenate(ds_, 0))


def mesh_sha256(points, cells) -> str:
    """Fingerprint (first 16 hex digits of the sha256 of the float32 points + uint32 cells) of a stand-in mesh.  scipy /
    Qhull may triangulate the same points differently between images, so "the same configuration" is only the same mesh
    where this agrees: printed by bench.py and by the parity tests beside the tet count."""
    import hashlib

    h = hashlib.sha256()
    h.update(np.ascontiguousarray(points, dtype=np.float32).tobytes())
    h.update(np.ascontiguousarray(cells).astype(np.uint32).tobytes())
    return h.hexdigest()[:16]
