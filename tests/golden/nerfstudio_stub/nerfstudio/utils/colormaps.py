"""nerfstudio.utils.colormaps: import-only (visualisation, model.py:681-685; off the hot path)."""


def apply_colormap(image, *args, **kwargs):
    return image.expand(*image.shape[:-1], 3)


def apply_depth_colormap(depth, accumulation=None, *args, **kwargs):
    return depth.expand(*depth.shape[:-1], 3)
