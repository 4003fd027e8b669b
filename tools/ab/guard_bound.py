#!/usr/bin/env python3
"""The error bound behind the wide walk's guard, checked by brute force on the CPU (numpy, FP32 emulated with the
reference's operation order: orion/geometry.hpp:80-136 as built, cross = a*b_yzx - a_yzx*b with the second product fused,
dot = (p0 + p1) + p2).

Claim (orion_amd/csrc/wide_walk.h): for a triangle (v0, e1, e2) and a ray (o, d) with
|d . (e1 x e2)| >= kappa |d| |e1| |e2|, the point o + t d of the computed t lies within
28.1 eps |o - v0| / kappa of the point v0 + u e1 + v e2 of the computed u, v (eps = 2^-24).
Prints the largest observed value of  |o + t d - (v0 + u e1 + v e2)| * kappa_ray / (eps |o - v0|),
kappa_ray = |d . n| / (|d| |e1| |e2|), over random pairs drawn near the threshold: it must stay below 28.1."""
import sys

import numpy as np

EPS = 2.0 ** -24
f32 = np.float32


def cross(a, b):
    def comp(i, j):   # a_i b_j - a_j b_i, second product fused
        p = (a[:, i] * b[:, j]).astype(f32)
        return (p.astype(np.float64) - a[:, j].astype(np.float64) * b[:, i].astype(np.float64)).astype(f32)
    return np.stack([comp(1, 2), comp(2, 0), comp(0, 1)], 1)


def dot(a, b):
    p = (a * b).astype(f32)
    return ((p[:, 0] + p[:, 1]).astype(f32) + p[:, 2]).astype(f32)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
    rng = np.random.default_rng(1)
    worst = 0.0
    for rnd in range(8):
        scale = 10.0 ** rng.uniform(-2, 2, (n, 1))
        v0 = (rng.standard_normal((n, 3)) * scale).astype(f32)
        e1 = (rng.standard_normal((n, 3)) * scale * 10.0 ** rng.uniform(-2, 0, (n, 1))).astype(f32)
        e2 = (rng.standard_normal((n, 3)) * scale * 10.0 ** rng.uniform(-2, 0, (n, 1))).astype(f32)
        nr = np.cross(e1.astype(np.float64), e2.astype(np.float64))
        ln = np.linalg.norm(nr, axis=1, keepdims=True)
        nh = nr / np.maximum(ln, 1e-300)
        # a point of the triangle, a direction nearly in its plane, an origin some way back
        a = rng.random((n, 1)); b = rng.random((n, 1)) * (1 - a)
        p = v0.astype(np.float64) + a * e1 + b * e2
        u1 = e1.astype(np.float64) / np.maximum(np.linalg.norm(e1.astype(np.float64), axis=1, keepdims=True), 1e-300)
        u2 = np.cross(nh, u1)
        ang = rng.random((n, 1)) * 2 * np.pi
        slope = 10.0 ** rng.uniform(-5, -1, (n, 1)) * rng.choice([-1.0, 1.0], (n, 1))
        dd = np.cos(ang) * u1 + np.sin(ang) * u2 + slope * nh
        back = np.linalg.norm(e1.astype(np.float64), axis=1, keepdims=True) * 10.0 ** rng.uniform(-1, 3, (n, 1))
        o = (p - back * dd).astype(f32)
        d = (dd * 10.0 ** rng.uniform(-1, 1, (n, 1))).astype(f32)
        # the reference's test
        pvec = cross(d, e2)
        det = dot(e1, pvec)
        with np.errstate(all="ignore"):
            inv = (f32(1.0) / det).astype(f32)
            tvec = (o - v0).astype(f32)
            u = (dot(tvec, pvec) * inv).astype(f32)
            qvec = cross(tvec, e1)
            v = (dot(d, qvec) * inv).astype(f32)
            t = (dot(e2, qvec) * inv).astype(f32)
        ok = (np.abs(det) >= 1e-6) & (u >= 0) & (u <= 1) & (v >= 0) & ((u + v).astype(f32) <= 1) & (t >= 0) & np.isfinite(t)
        D = np.einsum("ij,ij->i", d.astype(np.float64), nr)
        m = np.linalg.norm(e1.astype(np.float64), axis=1) * np.linalg.norm(e2.astype(np.float64), axis=1)
        kap = np.abs(D) / np.maximum(np.linalg.norm(d.astype(np.float64), axis=1) * m, 1e-300)
        hit = o.astype(np.float64) + t.astype(np.float64)[:, None] * d.astype(np.float64)
        X = v0.astype(np.float64) + u.astype(np.float64)[:, None] * e1 + v.astype(np.float64)[:, None] * e2
        T = np.linalg.norm(o.astype(np.float64) - v0.astype(np.float64), axis=1)
        val = np.linalg.norm(hit - X, axis=1) * kap / (EPS * np.maximum(T, 1e-300))
        sel = ok & (kap > 1e-6)
        if sel.any():
            worst = max(worst, float(val[sel].max()))
        print(f"round {rnd}: accepted {int(sel.sum())} of {n}, largest so far {worst:.2f} (bound 28.1)", flush=True)
    sys.exit(0 if worst < 28.1 else 1)


if __name__ == "__main__":
    main()
