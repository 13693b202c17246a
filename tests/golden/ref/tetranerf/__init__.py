from .utils.extension import cpp
