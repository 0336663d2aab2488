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


# ---- the shading frame, restated a second time from the reference's text (source/objects/TraceResult.cpp:45-103, 132-186 and
# ---- source/objects/Primitives.h:93-104), not from vt_oracle.c: vector expressions as glm evaluates them component-wise -----------
def _v(x):
    return [F(x[0]), F(x[1]), F(x[2])]


def _dot(a, b):                      # glm::dot(vec3): tmp = a * b; tmp.x + tmp.y + tmp.z
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _cross(x, y):                    # glm::cross
    return [x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]]


def _normalize(v):                   # glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = F(1.0) / np.sqrt(_dot(v, v))
        return [v[0] * inv, v[1] * inv, v[2] * inv]


def _bary(uvw, a):                   # uvw[2] * a[0] + uvw[0] * a[1] + uvw[1] * a[2], per component, left to right
    return [(uvw[2] * a[0][k] + uvw[0] * a[1][k]) + uvw[1] * a[2][k] for k in range(3)]


def hit_tbn(tri, direction, distance, u, v, normals, tangents, uvs, cone_width, cone_angle):
    """tri = (p0, e1, e2, n) as fp32 triples; normals / tangents: 3 x 3, uvs: 3 x 2.  Returns (normal, tangent, binormal,
    lod_info or None) -- TraceResult's GetNormal / GetTangent / GetBinormal and textureLodInfo for a material without a normal map."""
    p0, e1, e2, n = (_v(x) for x in tri)
    u, v, distance, cone_width, cone_angle = F(u), F(v), F(distance), F(cone_width), F(cone_angle)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        wo = [-c for c in _normalize(_v(direction))]                        # AccelStruct.cpp:826, TraceResult.cpp:56
        vN, vT = [_v(x) for x in normals], [_v(x) for x in tangents]         # :58-59
        vB = [_cross(vT[i], vN[i]) for i in range(3)]                        # :60
        uvw = [u, v, F(1.0) - u - v]                                         # :70
        length = np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])          # Primitives.h:102 length(n)
        ngeo = [n[0] / length, n[1] / length, n[2] / length]                 # :104, TraceResult.cpp:71
        normal, tangent, binormal = _normalize(_bary(uvw, vN)), _normalize(_bary(uvw, vT)), _normalize(_bary(uvw, vB))   # :134-136
        k_threshold = F(0.1)                                                 # :175
        cos_theta = abs(_dot(wo, normal))
        if cos_theta <= k_threshold:
            t = min(max(cos_theta * (F(1.0) / k_threshold), F(0.0)), F(1.0))                   # saturate
            normal = _normalize([ngeo[k] * (F(1.0) - t) + normal[k] * t for k in range(3)])     # lerp(geometricNormal, normal, t)
            tn = _dot(tangent, normal)
            tangent = _normalize([tangent[k] - normal[k] * tn for k in range(3)])
            binormal = _cross(tangent, normal)
        lod = None
        if not (cone_width < 0 or cone_angle <= 0):                          # :54 mipOverride
            cw = cone_angle * distance + cone_width                          # :95
            normal_term = _dot(wo, ngeo)
            uv = [[F(a), F(b)] for a, b in uvs]
            uv10, uv20 = [uv[1][0] - uv[0][0], uv[1][1] - uv[0][1]], [uv[2][0] - uv[0][0], uv[2][1] - uv[0][1]]
            area = abs(uv10[0] * uv20[1] - uv20[0] * uv10[1])                # Primitives.h:99
            lod = (F(0.5) * np.log2(area / length), (cw * cw) / (normal_term * normal_term))    # :103, TraceResult.cpp:99-102
    return normal, tangent, binormal, lod


# ---- skinning, restated a second time from source/objects/AccelStruct.cpp:34-92 (glm mat4 * mat4, mat4 * vec4, vec4 * scalar in
# ---- their generic scalar forms: columns scaled and added left to right) ------------------------------------------------------------
def mat4_mul(a, b):
    """glm: Result[c] = A[0] * B[c][0] + A[1] * B[c][1] + A[2] * B[c][2] + A[3] * B[c][3]; a, b: 16 floats, column-major."""
    a, b = np.asarray(a, F), np.asarray(b, F)
    out = np.zeros(16, F)
    for c in range(4):
        for r in range(4):
            acc = a[0 * 4 + r] * b[c * 4 + 0]
            acc = acc + a[1 * 4 + r] * b[c * 4 + 1]
            acc = acc + a[2 * 4 + r] * b[c * 4 + 2]
            acc = acc + a[3 * 4 + r] * b[c * 4 + 3]
            out[c * 4 + r] = acc
    return out


def transform_to_bone(vec, mats, num_bones, weights, bone_ids, angle_only=False):
    """TransformToBone (:35-47): final += bones[id] * binds[id] * vertex * weight, vertex = (vec, angle_only ? 0 : 1);
    mats[k] = the product bones[k] * binds[k] (16 floats, column-major).  glm mat4 * vec4 = (m0 * x + m1 * y) + (m2 * z + m3 * w)."""
    x, y, z, w = F(vec[0]), F(vec[1]), F(vec[2]), F(0.0 if angle_only else 1.0)
    fin = [F(0.0)] * 4
    with np.errstate(invalid="ignore", over="ignore"):
        for i in range(int(num_bones)):
            m = np.asarray(mats[int(bone_ids[i])], F)
            for r in range(4):
                col = (m[0 * 4 + r] * x + m[1 * 4 + r] * y) + (m[2 * 4 + r] * z + m[3 * 4 + r] * w)
                fin[r] = fin[r] + col * F(weights[i])
    return np.array(fin[:3], F)


# ---- the TraceResult constructor's derived fields and GetPos, restated a second time (source/objects/TraceResult.cpp:56-85, 255-262) ----
def hit_attrs(tri, direction, u, v, uvs=None, alphas=None):
    """tri = (p0, e1, e2, n).  Returns dict(wo, uvw, ngeo, pos, front[, tex_uv, blend])."""
    p0, e1, e2, n = (_v(x) for x in tri)
    u, v = F(u), F(v)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        wo = [-c for c in _normalize(_v(direction))]                         # :56 with AccelStruct.cpp:826
        uvw = [u, v, F(1.0) - u - v]                                          # :70
        length = np.sqrt((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2])
        ngeo = [n[k] / length for k in range(3)]                              # :71, Primitives.h:104
        verts = [p0, [p0[k] - e1[k] for k in range(3)], [p0[k] + e2[k] for k in range(3)]]   # :65-68: p0, p1() = p0 - e1, p2() = p0 + e2
        pos = _bary(uvw, verts)                                               # :258
        out = dict(wo=wo, uvw=uvw, ngeo=ngeo, pos=pos, front=bool(_dot(wo, ngeo) >= 0))          # :85
        if uvs is not None:
            uv = [[F(a), F(b)] for a, b in uvs]
            out["tex_uv"] = [(uvw[2] * uv[0][k] + uvw[0] * uv[1][k]) + uvw[1] * uv[2][k] for k in range(2)]   # :74
            al = [F(a) for a in alphas]
            out["blend"] = (uvw[2] * al[0] + uvw[0] * al[1]) + uvw[1] * al[2]   # :73
    return out
