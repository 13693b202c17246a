# tetranerf/utils/extension/tetranerf_cpp_extension.py   (ROCm / MI355X build) -- INTEGRATION.md section 2.
# The ONE file a maintainer adds to the reference instead of the CUDA/OptiX-built pybind11 module
# (src/py_binding.cpp:433-449): every name the reference binds, backed by libtetranerf_hip.so.
import importlib

_impl = importlib.import_module("tetra-nerf_amd.tetranerf_cpp_extension")
TetrahedraTracer = _impl.TetrahedraTracer                        # src/py_binding.cpp:434-440
triangulate = _impl.triangulate                                  # :442
find_average_spacing = _impl.find_average_spacing                # :443
interpolate_values = _impl.interpolate_values                    # :444
interpolate_values_backward = _impl.interpolate_values_backward  # :445
gather_uint32 = _impl.gather_uint32                              # :446
scatter_ema_uint32 = _impl.scatter_ema_uint32                    # :447
