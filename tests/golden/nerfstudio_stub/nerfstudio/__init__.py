"""Stub of nerfstudio 0.3.4 for running the reference's model file (see ../README.md).  TEST INFRASTRUCTURE."""
__version__ = "0.3.4+stub"
