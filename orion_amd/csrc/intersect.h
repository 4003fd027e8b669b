// Ray/box and ray/triangle tests and the two closest-hit / any-hit traversals.
//
// Arithmetic contract.  This translation unit is compiled with
// -ffp-contract=off: the only fused operations are the fmaf() calls written
// out below, and every other product and sum is rounded on its own.  The forms
// are the ones the reference's own object code uses (sbvh.cpp built with the
// reference's flags), so results are bit-identical to it:
//   cross(a,b): each component is a*b' - a'*b with the SECOND product fused
//               (g++ contracts _mm_sub_ps(_mm_mul_ps, _mm_mul_ps) of
//               orion/math.hpp:276-284 into vfnmadd);
//   dot(a,b):   (a.x*b.x + a.y*b.y) + a.z*b.z, products unfused (_mm_dp_ps,
//               orion/math.hpp:304-310).  _mm_dp_ps also adds a +0.0f lane, which
//               only ever changes the sign of a zero result; it is applied when a
//               hit is accepted (canon()) and skipped in the rejection tests,
//               whose comparisons cannot see the sign of zero;
//   divisions:  IEEE, correctly rounded (hipcc's default for FP32 '/').
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.h"

namespace orion_dev {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 splat(float s) { return mk(s, s, s); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }

// Dot product without the +0.0f lane (see the header comment).
__device__ __forceinline__ float dot_raw(V3 a, V3 b) {
    float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z;
    return (p0 + p1) + p2;
}
// The +0.0f lane of _mm_dp_ps: turns a -0 result into +0, changes nothing else.
__device__ __forceinline__ float canon(float s) { return s + 0.0f; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return canon(dot_raw(a, b)); }

__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(fmaf(-a.z, b.y, a.y * b.z), fmaf(-a.x, b.z, a.z * b.x), fmaf(-a.y, b.x, a.x * b.y));
}

// Lane rules of _mm_min_ps / _mm_max_ps (NaN or equal -> second operand) and of the
// scalar min/max templates (orion/math.hpp:336-365).  Written as compare+select on
// purpose: v_min_f32 / v_max_f32 treat NaN differently.
__device__ __forceinline__ float sse_min(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float sse_max(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float tpl_min(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float tpl_max(float a, float b) { return (a < b) ? b : a; }

// AABB::intersect (orion/AABB.hpp:79-99) with 1/dir hoisted out of the node loop
// (the reference recomputes the same quotient at every node).
__device__ __forceinline__ bool slab_hit(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 inv) {
    float t1x = (lox - o.x) * inv.x, t1y = (loy - o.y) * inv.y, t1z = (loz - o.z) * inv.z;
    float t2x = (hix - o.x) * inv.x, t2y = (hiy - o.y) * inv.y, t2z = (hiz - o.z) * inv.z;
    float lx = sse_min(t1x, t2x), ly = sse_min(t1y, t2y), lz = sse_min(t1z, t2z);
    float hx = sse_max(t1x, t2x), hy = sse_max(t1y, t2y), hz = sse_max(t1z, t2z);
    float tmin = tpl_max(lx, tpl_max(ly, lz));
    float tmax = tpl_min(hx, tpl_min(hy, hz));
    return (tmax > tmin) && (tmax > 0.0f);
}

// The same test on v_min/v_max.  For operands that are not NaN, (a<b)?a:b and min(a,b) differ at
// most in the sign of a zero, which the two final comparisons cannot see, so the result is the
// same bit.  NaN operands (0 * inf: the origin on a slab plane of an axis the ray does not move
// along) are where the select order matters; slab_safe() tells whether a ray can produce one.
// Single-instruction min/max: the builtins would first canonicalise every operand (v_max x, x).
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float vmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

__device__ __forceinline__ bool slab_hit_fast(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 inv) {
    float t1x = (lox - o.x) * inv.x, t1y = (loy - o.y) * inv.y, t1z = (loz - o.z) * inv.z;
    float t2x = (hix - o.x) * inv.x, t2y = (hiy - o.y) * inv.y, t2z = (hiz - o.z) * inv.z;
    float tmin = vmax3(vmin(t1x, t2x), vmin(t1y, t2y), vmin(t1z, t2z));
    float tmax = vmin3(vmax(t1x, t2x), vmax(t1y, t2y), vmax(t1z, t2z));
    return (tmax > tmin) && (tmax > 0.0f);
}

// True when no slab product of this ray can be NaN, whatever the (non-NaN) box: the origin is
// finite and every reciprocal is finite and non-zero, so (bound - o) * inv is finite or +-inf.
__device__ __forceinline__ bool slab_safe(V3 o, V3 inv) {
    const float big = __builtin_inff();
    const bool o_ok = fabsf(o.x) < big && fabsf(o.y) < big && fabsf(o.z) < big;
    const bool i_ok = fabsf(inv.x) < big && fabsf(inv.y) < big && fabsf(inv.z) < big &&
                      inv.x != 0.0f && inv.y != 0.0f && inv.z != 0.0f;
    return o_ok && i_ok;
}

// Triangle::intersect (orion/geometry.hpp:80-136).  Updates h on acceptance.
__device__ __forceinline__ bool tri_hit(V3 v0, V3 e1, V3 e2, uint32_t gtri, V3 o, V3 d, HitRec& h) {
    const float eps = 0.000001f;
    V3 pvec = cross(d, e2);
    float det = dot_raw(e1, pvec);
    if (det > -eps && det < eps) return false;
    float inv_det = 1.0f / det;
    V3 tvec = o - v0;
    float du = dot_raw(tvec, pvec);
    float u = du * inv_det;
    if (u < 0 || u > 1) return false;
    V3 qvec = cross(tvec, e1);
    float dv = dot_raw(d, qvec);
    float v = dv * inv_det;
    if (v < 0 || u + v > 1) return false;
    float dt = dot_raw(e2, qvec);
    float t = dt * inv_det;
    if (t > h.t || t < 0) return false;
    h.u = canon(du) * inv_det;
    h.v = canon(dv) * inv_det;
    h.t = canon(dt) * inv_det;
    h.gtri = gtri;
    return true;
}

__device__ __forceinline__ V3 ray_inverse(V3 d) { return mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// ---------------------------------------------------------------------------
// Per-lane traversal: every lane walks the threaded node array on its own.
// Scene data comes through vector loads (L1/L2 resident); the right choice when
// the lanes of a wave visit different parts of a large tree.
// ANY = true stops at the first accepted triangle (point-light shadow query,
// orion/raytracer.cpp:198-199: any hit at t >= 0 on the unbounded ray).
// ---------------------------------------------------------------------------
template <bool ANY>
__device__ __forceinline__ void traverse_lane(const SceneView& S, V3 o, V3 d, HitRec& h) {
    const V3 inv = ray_inverse(d);
    const bool fast = __all(slab_safe(o, inv));   // wave-uniform choice of the slab arithmetic
    uint32_t node = 0;
    const uint32_t n_nodes = S.n_nodes;
    while (node < n_nodes) {
        const float4 A = reinterpret_cast<const float4*>(S.nodes)[2 * node];
        const float4 B = reinterpret_cast<const float4*>(S.nodes)[2 * node + 1];
        const uint32_t a = __float_as_uint(A.w), b = __float_as_uint(B.w);
        if (b & kLeafFlag) {
            const uint32_t cnt = b & ~kLeafFlag;
            for (uint32_t s = 0; s < cnt; s++) {
                const float4* tp = reinterpret_cast<const float4*>(S.tris + (a + s));
                const float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
                bool acc = tri_hit(mk(T0.x, T0.y, T0.z), mk(T1.x, T1.y, T1.z), mk(T2.x, T2.y, T2.z), __float_as_uint(T0.w), o, d, h);
                if (ANY && acc) return;
            }
            node++;
        } else {
            const bool in = fast ? slab_hit_fast(A.x, A.y, A.z, B.x, B.y, B.z, o, inv) : slab_hit(A.x, A.y, A.z, B.x, B.y, B.z, o, inv);
            node = in ? node + 1 : a;
        }
    }
}

// ---------------------------------------------------------------------------
// Wave-synchronous traversal: the wave walks the node array once, front to
// back, and each lane joins in at the nodes on its own path (`next`).  The node
// index is wave-uniform, so nodes and triangles arrive through scalar loads into
// SGPRs and feed the VALU as broadcast operands: no vector memory traffic and no
// per-lane address arithmetic for scene data.  Cost is the union of the lanes'
// paths, which for scenes of a few dozen nodes is close to every lane's own path.
// Must be entered by all lanes that want a result; `active` = false lanes idle.
// ---------------------------------------------------------------------------
template <bool ANY>
__device__ __forceinline__ void traverse_wave(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) {
    const V3 inv = ray_inverse(d);
    const bool fast = __all(!active || slab_safe(o, inv));   // wave-uniform choice of the slab arithmetic
    uint32_t next = active ? 0u : kInvalid;
    const uint32_t n_nodes = S.n_nodes;
    for (uint32_t node = 0; node < n_nodes; node++) {
        const uint32_t un = __builtin_amdgcn_readfirstlane(node);   // the walk is wave-uniform: keep it in SGPRs
        const bool mine = (next == un);
        if (__ballot(mine) == 0ull) {
            if (ANY && !__any(next != kInvalid)) break;   // every lane has its answer
            continue;
        }
        const Node N = S.nodes[un];
        uint32_t to = un + 1;
        if (N.b & kLeafFlag) {
            const Tri* tp = S.tris + N.a;
            const Tri* const te = tp + (N.b & ~kLeafFlag);
            bool done = false;
            for (; tp != te; tp++) {
                const Tri T = *tp;
                // The lanes that are not at this node sit the test out (execution mask), so its
                // early exits fire as soon as the few lanes that are here have failed.
                if (mine && !(ANY && done)) {
                    bool acc = tri_hit(mk(T.v0[0], T.v0[1], T.v0[2]), mk(T.e1[0], T.e1[1], T.e1[2]),
                                       mk(T.e2[0], T.e2[1], T.e2[2]), T.gtri, o, d, h);
                    if (ANY && acc) done = true;
                }
            }
            if (ANY && done) to = kInvalid;
        } else {
            const bool in = fast ? slab_hit_fast(N.lo[0], N.lo[1], N.lo[2], N.hi[0], N.hi[1], N.hi[2], o, inv)
                                 : slab_hit(N.lo[0], N.lo[1], N.lo[2], N.hi[0], N.hi[1], N.hi[2], o, inv);
            if (!in) to = N.a;
        }
        next = mine ? to : next;
    }
}

} // namespace orion_dev
