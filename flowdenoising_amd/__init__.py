"""flowdenoising_amd -- MI355X (gfx950) implementation of FlowDenoising's hot path.

Python host code over libflowdn.so (hand-written HIP kernels behind the C ABI in
include/flowdn.h).  See DESIGN.md for the path, its boundary and the data layout.
"""
from . import _lib  # noqa: F401
from .operators import (  # noqa: F401
    FlowDenoising, GaussianDenoising, OF_filter, OF_filter_along_X, OF_filter_along_Y, OF_filter_along_Z,
    get_flow, get_flow_with_prev_flow, get_flow_without_prev_flow, get_gaussian_kernel, no_OF_filter,
    no_OF_filter_along_X, no_OF_filter_along_Y, no_OF_filter_along_Z, warp_slice,
)

from .flower import CPU_flower, GPU_flower  # noqa: F401,E402

__version__ = "0.1.0"
