"""A second, independent restatement of the single-ray walk for SMALL cases: pure-Python loops over numpy.float32
scalars (every operation rounds to fp32; a product followed by a sum is two roundings, i.e. unfused), written from
the pseudo-code of SURVEY.md section 3.2 and source/objects/Primitives.h:168-215 -- not from vt_oracle.c.

TEST INFRASTRUCTURE like the rest of oracle/: tests/test_oracle_pywalk.py checks that the C oracle and this walk
agree bit for bit on hits and on the two counters (same visitation order, same tie-breaks)."""
import numpy as np

F = np.float32
FLT_EPSILON = np.finfo(np.float32).eps
MISS = 0xFFFFFFFF


def _safe_inverse(x):
    if abs(x) <= FLT_EPSILON:
        return np.copysign(F(1.0) / FLT_EPSILON, x)
    return F(1.0) / x


def _tri_intersect(tri, org, d, tmin, tmax):
    """TriangleBackfaceCull::intersect without the alpha branch. tri = (p0, e1, e2, n, flags)."""
    p0, e1, e2, n, flags = tri
    ndd = (n[0] * d[0] + n[1] * d[1]) + n[2] * d[2]
    if (flags & 1) and ndd > 0:
        return None
    c = [p0[k] - org[k] for k in range(3)]
    r = [d[1] * c[2] - d[2] * c[1], d[2] * c[0] - d[0] * c[2], d[0] * c[1] - d[1] * c[0]]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = F(1.0) / ndd
        u = ((r[0] * e2[0] + r[1] * e2[1]) + r[2] * e2[2]) * inv
        v = ((r[0] * e1[0] + r[1] * e1[1]) + r[2] * e1[2]) * inv
        w = F(1.0) - u - v
        if u >= 0 and v >= 0 and w >= 0:
            t = ((n[0] * c[0] + n[1] * c[1]) + n[2] * c[2]) * inv
            if t >= tmin and t <= tmax:
                return t, u, v
    return None


def walk(nodes, prim_indices, tris, ray, any_hit=False):
    """nodes: structured array (bounds[6], prim_count, first); tris: structured (p0,e1,e2,n,flags) in original order.
    Returns (prim, t, u, v, steps, tests)."""
    org = [F(x) for x in ray["org"]]
    d = [F(x) for x in ray["dir"]]
    tmin, tmax = F(ray["tmin"]), F(ray["tmax"])
    best = [MISS, F(0), F(0), F(0)]
    steps = tests = 0
    tri_of = lambda i: (tris["p0"][i], tris["e1"][i], tris["e2"][i], tris["n"][i], int(tris["flags"][i]))

    def leaf(node):
        nonlocal tmax, tests
        for slot in range(int(node["first"]), int(node["first"]) + int(node["prim_count"])):
            idx = int(prim_indices[slot])
            tests += 1
            hit = _tri_intersect(tri_of(idx), org, d, tmin, tmax)
            if hit is not None:
                best[:] = [idx, hit[0], hit[1], hit[2]]
                if any_hit:
                    return True
                tmax = hit[0]
        return False

    if len(nodes) == 0:
        return (*best, 0, 0)
    if nodes[0]["prim_count"] != 0:
        leaf(nodes[0])
        return (*best, steps, tests)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        octant = [1 if np.signbit(x) else 0 for x in d]
        inv = [_safe_inverse(x) for x in d]
        sorg = [-org[k] * inv[k] for k in range(3)]

        def slab(node):
            b = node["bounds"]
            ent = [F(b[2 * k + octant[k]]) * inv[k] + sorg[k] for k in range(3)]
            ext = [F(b[2 * k + 1 - octant[k]]) * inv[k] + sorg[k] for k in range(3)]
            rmax = lambda a, bb: a if a > bb else bb
            rmin = lambda a, bb: a if a < bb else bb
            first = rmax(ent[0], rmax(ent[1], rmax(ent[2], tmin)))
            second = rmin(ext[0], rmin(ext[1], rmin(ext[2], tmax)))
            return first, second

        stack = []
        left = int(nodes[0]["first"])
        while True:
            right = left + 1
            steps += 1
            fl, sl = slab(nodes[left])
            fr, sr = slab(nodes[right])
            l_keep = r_keep = False
            if fl <= sl:
                if nodes[left]["prim_count"] != 0:
                    if leaf(nodes[left]):
                        break
                else:
                    l_keep = True
            if fr <= sr:
                if nodes[right]["prim_count"] != 0:
                    if leaf(nodes[right]):
                        break
                else:
                    r_keep = True
            if l_keep:
                if r_keep:
                    near, far = (right, left) if fl > fr else (left, right)
                    stack.append(int(nodes[far]["first"]))
                    left = int(nodes[near]["first"])
                else:
                    left = int(nodes[left]["first"])
            elif r_keep:
                left = int(nodes[right]["first"])
            else:
                if not stack:
                    break
                left = stack.pop()
    return (*best, steps, tests)
