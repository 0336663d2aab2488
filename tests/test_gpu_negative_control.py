"""Negative control of the parity suite, run with it every time: a deliberately wrong traversal kernel must go red.

`__graft_entry__.build()` also builds two of the mutants of `trace_kernels.hip`'s VT_MUT list (vistrace_amd/lib/variants/, never
the product library: tests/test_abi_symbols.py): 2 = a hit accepted on `t < tmax` instead of `t <= tmax`
(source/objects/Primitives.h:189), 7 = a node accepted on `first < second` instead of `<=` (bvh v1's slab test).
Each is put in front of the golden-vector check through VISTRACE_HIP_LIB in a child process: the product library reproduces the
committed fixtures bit for bit, both mutants must not.  The whole campaign (73 mutants): scripts/mutants*.sh, profiles/r6/mutants*.txt."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHECK = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import vistrace_amd as va
bad = 0
for fx in ("s1k_golden.npz", "terrain_golden.npz"):
    g = np.load(os.path.join(%r, "tests", "golden", fx))
    tris = va.tris_setup(g["verts"], g["flags"] if "flags" in g.files else None)
    eng = va.Engine(0)
    for mode in (1, 0):
        eng.set_option("persistent", mode)
        scene = va.Scene.from_tree(eng, va.HostBvh(tris, builder="ploc"))
        got = scene.trace_closest(g["rays"].view(va.RAY).reshape(-1))
        bad += int((got.view(np.uint8).reshape(-1, 16) != g["hits"].view(va.HIT).reshape(-1).view(np.uint8).reshape(-1, 16)).any(axis=1).sum())
        scene.free()
    eng.close()
print("MISMATCHES", bad)
""" % (ROOT, ROOT)


def _mismatches(lib):
    env = dict(os.environ)
    if lib:
        env["VISTRACE_HIP_LIB"] = lib
    else:
        env.pop("VISTRACE_HIP_LIB", None)
    out = subprocess.run([sys.executable, "-c", CHECK], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("MISMATCHES")][-1]
    return int(line.split()[1]), out.stderr


@pytest.mark.gpu
def test_wrong_kernels_go_red_and_the_product_does_not():
    n, _ = _mismatches(None)
    assert n == 0, "the product library does not reproduce the golden fixtures"
    for k, what in ((2, "t < tmax"), (7, "first < second")):
        lib = os.path.join(ROOT, "vistrace_amd", "lib", "variants", f"libvistrace_hip_mut_{k}.so")
        if not os.path.exists(lib):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "vistrace_amd", "csrc"), "mutant", f"K={k}"], stdout=subprocess.DEVNULL)
        n, err = _mismatches(lib)
        assert f"MUTANT {k}" in err, "the child did not load the mutant library"
        assert n > 0, f"mutant {k} ({what}) reproduces the golden fixtures: the parity check has no teeth"
