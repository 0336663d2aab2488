"""The lane <-> ray mapping of engine option `ray_image_width` (trace_kernels.hip, TraceArgs::tile_w), restated in numpy: both
kernels must visit every ray of the batch exactly once, and a wave's 64 rays must be one 4-wide, 16-high pixel tile inside the tiled
part.  (The kernels themselves are checked on the GPU: tests/test_gpu_parity.py::test_ray_image_width_changes_no_result.)"""
import numpy as np
import pytest


def tiled_rays(n, width):
    band = width * 16
    return n // band * band if width >= 4 and width % 4 == 0 and n < (1 << 32) else 0


def static_kernel_index(j, n, width):
    """One ray per lane: thread j -> ray index (trace_body, !PERSISTENT)."""
    j = np.asarray(j, np.int64)
    t, k, tpr = j >> 6, j & 63, width >> 2
    ty, tx = t // tpr, t % tpr
    mapped = (ty * 16 + (k >> 2)) * width + tx * 4 + (k & 3)
    return np.where(j < tiled_rays(n, width), mapped, j)


def persistent_kernel_index(j, n, width, block_rays):
    """Persistent waves: tile-order index j of a ray block -> ray index (the re-fill path of trace_body)."""
    j = np.asarray(j, np.int64)
    first = j // block_rays * block_rays
    t, tpr = first >> 6, width >> 2
    ty, tx = t // tpr, t % tpr
    base = ty * 16 * width + tx * 4
    k = j - first
    mapped = base + ((k >> 6) << 2) + (k & 3) + ((k >> 2) & 15) * width
    return np.where(first + block_rays <= tiled_rays(n, width), mapped, j)


@pytest.mark.parametrize("width,height", [(1024, 1024), (256, 48), (100, 37), (36, 50), (8, 16), (4, 33), (4096, 16)])
def test_both_mappings_are_permutations_made_of_pixel_tiles(width, height):
    n = width * height
    j = np.arange(n)
    block = 128 if width % 8 == 0 else 64          # engine.hip: two tiles side by side need a row length that is a multiple of 8
    for idx in (static_kernel_index(j, n, width), persistent_kernel_index(j, n, width, block)):
        assert np.array_equal(np.sort(idx), j), "not a permutation of the batch"
        tiled = tiled_rays(n, width)
        x, y = idx[:tiled] % width, idx[:tiled] // width
        for w in range(0, tiled, 64):               # every wave-sized run of the tiled part is one 4 x 16 tile
            assert x[w:w + 64].max() - x[w:w + 64].min() == 3 and y[w:w + 64].max() - y[w:w + 64].min() == 15
        assert np.array_equal(idx[tiled:], j[tiled:]), "the rest of the batch is taken in order"


def test_widths_the_engine_ignores():
    for width in (0, 1, 2, 3, 5, 255, 1022):
        assert tiled_rays(1 << 20, width) == 0
    assert tiled_rays(1 << 32, 1024) == 0           # batches of 2^32 rays and more are taken in order
