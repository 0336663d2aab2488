"""torch as plumbing: device buffers, streams and (in bench.py) torch.distributed.

Device buffers are plain uint8 tensors holding the POD records of include/vistrace_hip.h;
every computation on them happens in libvistrace_hip.so.
"""
from __future__ import annotations

import numpy as np
import torch

from ._lib import HIT, HIT_ATTRS, RAY, RAY_STATS


def to_device(arr: np.ndarray, device: torch.device) -> torch.Tensor:
    a = np.ascontiguousarray(arr)
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).to(device)


def to_host(t: torch.Tensor, dtype: np.dtype) -> np.ndarray:
    return t.cpu().numpy().view(dtype)


def empty_records(n: int, dtype: np.dtype, device: torch.device) -> torch.Tensor:
    return torch.empty(n * dtype.itemsize, dtype=torch.uint8, device=device)


def current_stream_handle(device: torch.device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


def trace_closest(scene, d_rays: torch.Tensor, n: int, d_hits: torch.Tensor | None = None) -> torch.Tensor:
    if d_hits is None:
        d_hits = empty_records(n, HIT, d_rays.device)
    scene.trace_closest_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), current_stream_handle(d_rays.device))
    return d_hits


def trace_any(scene, d_rays: torch.Tensor, n: int, d_occ: torch.Tensor | None = None) -> torch.Tensor:
    if d_occ is None:
        d_occ = torch.empty(n, dtype=torch.uint8, device=d_rays.device)
    scene.trace_any_dev(d_rays.data_ptr(), n, d_occ.data_ptr(), current_stream_handle(d_rays.device))
    return d_occ


def trace_stats(scene, d_rays: torch.Tensor, n: int):
    d_hits = empty_records(n, HIT, d_rays.device)
    d_stats = empty_records(n, RAY_STATS, d_rays.device)
    scene.trace_stats_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), d_stats.data_ptr(),
                          current_stream_handle(d_rays.device))
    return d_hits, d_stats


def trace_any_stats(scene, d_rays: torch.Tensor, n: int):
    d_occ = torch.empty(n, dtype=torch.uint8, device=d_rays.device)
    d_stats = empty_records(n, RAY_STATS, d_rays.device)
    scene.trace_any_stats_dev(d_rays.data_ptr(), n, d_occ.data_ptr(), d_stats.data_ptr(),
                              current_stream_handle(d_rays.device))
    return d_occ, d_stats


def hit_attrs(scene, d_rays: torch.Tensor, d_hits: torch.Tensor, n: int) -> torch.Tensor:
    d_attrs = empty_records(n, HIT_ATTRS, d_rays.device)
    scene.hit_attrs_dev(d_rays.data_ptr(), d_hits.data_ptr(), n, d_attrs.data_ptr(),
                        current_stream_handle(d_rays.device))
    return d_attrs


def bounce_loop(scene, d_rays: torch.Tensor, n: int, depth: int, seed: int):
    """(d_hits as a (depth*n*16)-byte tensor, live-path counts per depth); see vt_bounce_loop_dev."""
    d_hits = empty_records(n * depth, HIT, d_rays.device)
    live = scene.bounce_loop_dev(d_rays.data_ptr(), n, depth, seed, d_hits.data_ptr(), current_stream_handle(d_rays.device))
    return d_hits, live


__all__ = ["to_device", "to_host", "empty_records", "current_stream_handle", "trace_closest", "trace_any",
           "trace_stats", "trace_any_stats", "hit_attrs", "bounce_loop", "RAY", "HIT"]
