"""Deterministic ray batches and synthetic scenes for the parity tests (numpy only)."""
import numpy as np


def triangle_soup(n_tri: int = 19058, edge: float = 0.02, seed: int = 1):
    """n_tri small triangles scattered in the unit cube: one mesh, flat normals, no uvs."""
    rng = np.random.default_rng(seed)
    base = rng.random((n_tri, 3)).astype(np.float32)
    d1 = ((rng.random((n_tri, 3)) - 0.5) * 2 * edge).astype(np.float32)
    d2 = ((rng.random((n_tri, 3)) - 0.5) * 2 * edge).astype(np.float32)
    p = np.stack([base, base + d1, base + d2], axis=1)  # [n,3,3]
    nrm = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
    ln = np.linalg.norm(nrm, axis=1, keepdims=True)
    nrm = (nrm / np.where(ln > 0, ln, 1)).astype(np.float32)
    verts = np.zeros((n_tri * 3, 8), np.float32)
    verts[:, 0:3] = p.reshape(-1, 3)
    verts[:, 3:6] = np.repeat(nrm, 3, axis=0)
    verts[:, 6:8] = rng.random((n_tri * 3, 2)).astype(np.float32)
    idx = np.arange(n_tri * 3, dtype=np.uint32)
    return verts, idx


def scene_bounds(model):
    allv = np.concatenate([m.vertices[:, :3] for m in model.meshes if len(m.vertices)])
    return allv.min(0), allv.max(0)


def default_camera(model):
    """eye/look-at/up/yview framing the whole model from +z (used when a scene has no .rtc)."""
    lo, hi = scene_bounds(model)
    c = (lo + hi) / 2
    ext = float((hi - lo).max())
    eye = np.array([c[0], c[1], hi[2] + 1.5 * ext], np.float32)
    return eye, c.astype(np.float32), np.array([0, 1, 0], np.float32), 1.0


def primary_rays(camera12, width, height, n, seed):
    """R1: jittered primary rays of random pixels, formed as the reference forms them
    (orion/raytracer.cpp:72-80): unnormalised dir = front + xs*right + ys*up."""
    rng = np.random.default_rng(seed)
    c = np.asarray(camera12, np.float32)
    eye, front, right, up = c[0:3], c[3:6], c[6:9], c[9:12]
    j = rng.integers(0, width, n); i = rng.integers(0, height, n)
    x = (2 * (j.astype(np.float32) / np.float32(width)) - 1 + (rng.random(n) * 2 / width).astype(np.float32)).astype(np.float32)
    y = (2 * (i.astype(np.float32) / np.float32(height)) - 1 + (rng.random(n) * 2 / height).astype(np.float32)).astype(np.float32)
    d = (front[None, :] + x[:, None] * right[None, :] + (-y)[:, None] * up[None, :]).astype(np.float32)
    o = np.repeat(eye[None, :], n, axis=0).astype(np.float32)
    return o, d


def bounce_rays(org, dir, ids, tuv, normals, seed):
    """R2: incoherent rays leaving the hit points of (org, dir): origin = hit + 1e-3*n,
    direction = cosine-weighted about n (orion/raytracer.cpp:174-194).  Misses are dropped."""
    rng = np.random.default_rng(seed)
    hit = ids[:, 0] != 0xFFFFFFFF
    o = org[hit]; d = dir[hit]; t = tuv[hit, 0:1]; n = normals[hit]
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    n = n / np.where(ln > 0, ln, 1)
    p = o + d * t
    k = len(p)
    s = np.sqrt(rng.random(k)); c = np.sqrt(1 - s * s); psi = rng.random(k) * 2 * np.pi
    tang = np.cross(n, np.array([0, 1, 0.0]))
    bad = (tang * tang).sum(1) == 0
    tang[bad] = np.cross(n[bad], np.array([0, 0, 1.0]))
    bit = np.cross(n, tang)
    nd = (s * np.cos(psi))[:, None] * tang + (s * np.sin(psi))[:, None] * bit + c[:, None] * n
    return (p + 1e-3 * n).astype(np.float32), nd.astype(np.float32)


def rays_to_surface(origins, model, seed):
    """R3: shadow-style rays from the given origins towards uniform points of random triangles
    (target - origin, unnormalised, as orion/raytracer.cpp:143-147 forms them)."""
    rng = np.random.default_rng(seed)
    tris = []
    for m in model.meshes:
        if m.n_triangles:
            tris.append(m.vertices[m.indices.reshape(-1, 3), :3])
    tris = np.concatenate(tris)  # [T,3,3]
    k = len(origins)
    pick = rng.integers(0, len(tris), k)
    a = rng.random(k); b = rng.random(k)
    flip = a + b > 1
    a[flip] = 1 - a[flip]; b[flip] = 1 - b[flip]
    tp = tris[pick]
    target = tp[:, 0] + a[:, None] * (tp[:, 1] - tp[:, 0]) + b[:, None] * (tp[:, 2] - tp[:, 0])
    return origins.astype(np.float32), (target - origins).astype(np.float32)


def box_rays(model, n, seed):
    """Rays from around the scene towards random points of its bounding box."""
    lo, hi = scene_bounds(model)
    rng = np.random.default_rng(seed)
    c = (lo + hi) / 2; ext = (hi - lo).max()
    org = (c + (rng.random((n, 3)) - 0.5) * ext * 2.5).astype(np.float32)
    tgt = (lo + rng.random((n, 3)) * (hi - lo)).astype(np.float32)
    return org, (tgt - org).astype(np.float32)


def grazing_rays(model, n, seed):
    """Rays that run along a triangle's plane at a slope between 1e-8 and 1e-2, through a point of the triangle or just
    beside it, from up to a few scene sizes back, direction length 0.1 to 10: where Triangle::intersect's t, u, v
    (orion/geometry.hpp:80-136) are worst conditioned -- a determinant a few rounding errors above the 1e-6 it rejects,
    numerators that are rounding residue."""
    rng = np.random.default_rng(seed)
    tris = np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles]).astype(np.float64)
    lo = tris.reshape(-1, 3).min(0); hi = tris.reshape(-1, 3).max(0)
    tp = tris[rng.integers(0, len(tris), n)]
    e1 = tp[:, 1] - tp[:, 0]; e2 = tp[:, 2] - tp[:, 0]
    nr = np.cross(e1, e2); nr /= np.maximum(np.linalg.norm(nr, axis=1, keepdims=True), 1e-30)
    a = rng.random(n) * 1.4 - 0.2; b = rng.random(n) * 1.4 - 0.2
    p = tp[:, 0] + a[:, None] * e1 + b[:, None] * e2
    ang = rng.random(n) * 2 * np.pi
    u1 = e1 / np.maximum(np.linalg.norm(e1, axis=1, keepdims=True), 1e-30); u2 = np.cross(nr, u1)
    inplane = np.cos(ang)[:, None] * u1 + np.sin(ang)[:, None] * u2
    slope = 10.0 ** (-8.0 + 6.0 * rng.random(n)) * (rng.integers(0, 2, n) * 2 - 1)
    d = inplane + slope[:, None] * nr
    back = (hi - lo).max() * 10.0 ** (-2.0 + 2.5 * rng.random(n))
    o = p - back[:, None] * d
    d = d * (10.0 ** (-1.0 + 2.0 * rng.random(n)))[:, None]
    return o.astype(np.float32), d.astype(np.float32)


def edge_case_rays(model, seed=7):
    """Rays built to land on the predicates' edges: axis-parallel directions (zero
    components -> infinite reciprocals), origins on box planes, rays through triangle
    vertices and along edges (u = 0, v = 0, u + v = 1), rays in triangle planes, rays
    through coincident coplanar faces, zero-length directions."""
    rng = np.random.default_rng(seed)
    lo, hi = scene_bounds(model)
    c = (lo + hi) / 2
    tris = np.concatenate([m.vertices[m.indices.reshape(-1, 3), :3] for m in model.meshes if m.n_triangles])
    if len(tris) > 400:
        tris = tris[rng.choice(len(tris), 400, replace=False)]
    O, D = [], []
    axes = np.eye(3, dtype=np.float32)
    # axis-parallel rays through vertices, edge midpoints and centroids
    pts = np.concatenate([tris.reshape(-1, 3), (tris[:, 0] + tris[:, 1]) / 2, (tris[:, 1] + tris[:, 2]) / 2,
                          (tris[:, 0] + tris[:, 2]) / 2, tris.mean(1)])
    for a in range(3):
        for sgn in (1.0, -1.0):
            d = axes[a] * sgn
            O.append(pts - d * np.float32(3.0)); D.append(np.repeat(d[None], len(pts), 0))
    # from the scene centre (and an outside point) exactly at vertices / edge points
    for src in (c, hi + (hi - lo) * 0.7):
        O.append(np.repeat(src[None].astype(np.float32), len(pts), 0)); D.append(pts - src)
    # origins on the planes of the scene box, axis-parallel and diagonal directions
    k = 256
    p = (lo + rng.random((k, 3)) * (hi - lo)).astype(np.float32)
    for a in range(3):
        q = p.copy(); q[:, a] = lo[a]
        O.append(q); D.append(np.repeat(axes[(a + 1) % 3][None], k, 0))
        q = p.copy(); q[:, a] = hi[a]
        O.append(q); D.append((c - q).astype(np.float32))
    # rays lying in triangle planes (det ~ 0) and starting on the triangle (t = 0)
    e1 = tris[:, 1] - tris[:, 0]
    O.append(tris[:, 0] - e1); D.append(e1 * 3)
    O.append(tris.mean(1)); D.append(np.cross(e1, tris[:, 2] - tris[:, 0]))
    # degenerate directions
    O.append(np.repeat(c[None].astype(np.float32), 4, 0))
    D.append(np.array([[0, 0, 0], [0, 0, 1e-30], [1e30, 0, 0], [-0.0, 0.0, -1.0]], np.float32))
    return np.concatenate(O).astype(np.float32), np.concatenate(D).astype(np.float32)
