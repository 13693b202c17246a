"""Importable alias of the package directory `tetra-nerf_amd/` (a hyphen is not a valid
identifier): `import tetranerf_amd` == importlib.import_module("tetra-nerf_amd")."""
import importlib
import sys

_pkg = importlib.import_module("tetra-nerf_amd")
sys.modules[__name__] = _pkg
