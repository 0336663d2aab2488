"""vistrace_amd -- MI355X (gfx950) ray-tracing core for VisTrace's AccelStruct::Traverse path.

Python here is a thin ctypes layer over the C ABI (include/vistrace_hip.h); the product is
vistrace_amd/lib/libvistrace_hip.so (hand-written HIP kernels + C++ host).  Batches are traced on the
GPU only -- there is no CPU fallback; the one host-side tracer is the explicitly named single-ray
latency path (HostScene.trace_closest_host = vt_host_scene_trace_closest, BASELINE config 1).
"""
from .api import *  # noqa: F401,F403
from . import workloads  # noqa: F401
