"""pytest configuration: `gpu` marker, shared scene/oracle fixtures.

`-m "not gpu"` runs the oracle against its known answers and golden vectors, the host
logic (BVH build, linearise, workloads, multi-process sharding) and the C-ABI symbol check.
`-m gpu` runs the parity tests proper: HIP path (through the C ABI) vs the CPU oracle.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def O():
    from oracle import binding
    binding.lib()
    return binding


@pytest.fixture(scope="session")
def va():
    import vistrace_amd
    return vistrace_amd


class SceneBundle:
    """A scene built once by the product's host code + the oracle's view of the same tree."""

    def __init__(self, va, O, verts, flags=None):
        self.verts = np.ascontiguousarray(verts, np.float32)
        self.flags = flags
        self.tris = va.tris_setup(self.verts, flags)
        self.bvh = va.HostBvh(self.tris)
        self.host_scene = va.HostScene(self.bvh)
        self.nodes = self.bvh.nodes().view(O.NODE)
        self.prim_indices = self.bvh.prim_indices()
        self.otris = O.tris_from_tri64(self.tris)
        self._O = O

    def oracle(self, rays, any_hit=False, want_stats=False):
        hits, st, steps, tests, _ = self._O.traverse_batch(self.nodes, self.prim_indices, self.otris, rays,
                                                           any_hit=any_hit, want_stats=want_stats)
        return (hits, st) if want_stats else hits


@pytest.fixture(scope="session")
def make_bundle(va, O):
    cache = {}

    def _make(name):
        from vistrace_amd import workloads as W
        if name not in cache:
            if name == "terrain":
                verts, flags = W.make_terrain()
                cache[name] = SceneBundle(va, O, verts, flags)
            else:
                cache[name] = SceneBundle(va, O, W.make_scene(name))
        return cache[name]

    return _make


@pytest.fixture(scope="session", params=["persistent", "static"])
def engine(va, request):
    """Every GPU parity test runs against both kernels: persistent waves (LDS-DMA fetch, lane re-fill,
    coherence probe) and the one-ray-per-lane kernel that the default auto mode picks for small batches."""
    # VT_TEST_GROUP_MEMBERS=N (with VT_TEST_ALLOW_DEVICE_ALIASES=1 and the RCCL test double): the whole parity suite through the
    # root of an N-member group -- scenes, refits, skins, alpha tables and frames are then replicated to every member
    members = int(os.environ.get("VT_TEST_GROUP_MEMBERS", "0"))
    eng = va.Engine([0] * members) if members > 1 else va.Engine(0)
    eng.set_option("persistent", 1 if request.param == "persistent" else 0)
    eng.set_option("static_overflow_mb", 2048)
    return eng
