"""vistrace_amd -- MI355X (gfx950) ray-tracing core for VisTrace's AccelStruct::Traverse path.

Python here is a thin ctypes layer over the C ABI (include/vistrace_hip.h); the product is
vistrace_amd/lib/libvistrace_hip.so (hand-written HIP kernels + C++ host).  No CPU tracing
path exists in this package.
"""
from .api import *  # noqa: F401,F403
from . import workloads  # noqa: F401
