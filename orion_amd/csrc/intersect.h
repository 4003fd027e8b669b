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
//   divisions:  IEEE, correctly rounded (hipcc's default for FP32 '/').  The two reciprocals of the
//               walk, 1/dir and 1/det, take a 3-instruction form (recip_quick) for rays whose
//               divisors are provably in the range where it returns the division's bits.
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

__device__ __forceinline__ V3 ray_inverse(V3 d) { return mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); }

// 1 / x as the hardware's reciprocal estimate and one Newton step.  For every float with
// 2^-126 <= |x| < 2^126 this returns the bits of the correctly rounded division (checked for each
// of them, tools/micro/rcp_exact.hip) in 3 instructions instead of 11; outside that range
// (zero, denormals, the top two binades, infinities) it does not, and the callers divide.
__device__ __forceinline__ float recip_quick(float x) {
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ V3 ray_inverse_quick(V3 d) { return mk(recip_quick(d.x), recip_quick(d.y), recip_quick(d.z)); }

// Which rays may take the quick reciprocals.  `small`: every |component| of the direction is
// below the scene's bound (SceneView::dir_bound <= 2^126), chosen at upload so that no triangle's
// determinant can then reach 2^126 -- with the |det| >= 1e-6 test of the triangle routine that
// puts every divisor in recip_quick's range.  `normal`: also no component below 2^-126, so the
// ray's own reciprocals are in range.  NaN components fail both.
__device__ __forceinline__ void dir_class(V3 d, float bound, bool& small, bool& normal) {
    small = fabsf(d.x) < bound && fabsf(d.y) < bound && fabsf(d.z) < bound;
    const int kNormal = (1 << 3) | (1 << 8);   // v_cmp_class: negative / positive normal number
    normal = small && __builtin_amdgcn_classf(d.x, kNormal) && __builtin_amdgcn_classf(d.y, kNormal) && __builtin_amdgcn_classf(d.z, kNormal);
}

// Finite (not infinite, not NaN) in one instruction per component.
__device__ __forceinline__ bool finite3(V3 a) {
    const int kFinite = 0x1F8;   // v_cmp_class: normal, denormal and zero of either sign
    return __builtin_amdgcn_classf(a.x, kFinite) && __builtin_amdgcn_classf(a.y, kFinite) && __builtin_amdgcn_classf(a.z, kFinite);
}

// Triangle::intersect (orion/geometry.hpp:80-136).  Updates h on acceptance.
// QUICK: the ray is one for which recip_quick() returns the division's bits (see dir_class below).
template <bool QUICK>
__device__ __forceinline__ bool tri_hit(V3 v0, V3 e1, V3 e2, uint32_t gtri, V3 o, V3 d, HitRec& h) {
    const float eps = 0.000001f;
    V3 pvec = cross(d, e2);
    float det = dot_raw(e1, pvec);
    if (det > -eps && det < eps) return false;
    float inv_det = QUICK ? recip_quick(det) : 1.0f / det;
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

__device__ __forceinline__ V3 ld3(const float* p) { return mk(p[0], p[1], p[2]); }

// ---------------------------------------------------------------------------
// Per-lane traversal: every lane walks the threaded node array on its own.
// Scene data comes through vector loads (L1/L2 resident); the right choice when
// the lanes of a wave visit different parts of a large tree.
// ANY = true stops at the first accepted triangle (point-light shadow query,
// orion/raytracer.cpp:198-199: any hit at t >= 0 on the unbounded ray).
// ---------------------------------------------------------------------------
// Keeps a 16-byte load whole and where it was written.  Left alone, the compiler splits a record's
// loads and sinks the parts into the branches that use them (first the leaf flag, then, inside the
// branch, the box), which turns one memory round trip per node into two dependent ones.
__device__ __forceinline__ void pin(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }

// A lane parks up to kLeafSlots leaves; the wave runs the triangle code once kLeafQuorum lanes are
// full.  Measured on two large scenes: slots 1 / 2 / 4 / 6 / 8 give 1 / 1.12 / 1.20 / 1.25 / 1.26 of
// the one-slot rate on bounce rays; quorum 8, 16, 32 are within 3 % of each other.
constexpr int kLeafSlots = 6, kLeafQuorum = 16;

// Walks nodes [begin, end) of the threaded array: the whole scene, or the subtree of the inner node at
// `begin` with end = its skip link (in depth-first preorder a subtree is a contiguous range and no link
// inside it points past its end).
template <bool ANY, bool QUICK>
__device__ __forceinline__ void traverse_lane_walk(const SceneView& S, V3 o, V3 d, V3 inv, bool fast, HitRec& h, uint32_t begin, uint32_t end) {
    uint32_t node = begin;
    const uint32_t n_nodes = end;
    constexpr int K = kLeafSlots;
    uint32_t first[K], cnt[K];   // the leaves this lane has yet to test, in visit order (registers: every index below is a constant)
#pragma unroll
    for (int k = 0; k < K; k++) { first[k] = 0; cnt[k] = 0; }
    while (node < n_nodes || cnt[0]) {
        const bool moving = cnt[K - 1] == 0 && node < n_nodes;
        if (moving) {
            float4 A = reinterpret_cast<const float4*>(S.nodes)[2 * node];
            float4 B = reinterpret_cast<const float4*>(S.nodes)[2 * node + 1];
            pin(A); pin(B);   // both halves in one round trip (see pin)
            const uint32_t a = __float_as_uint(A.w), b = __float_as_uint(B.w);
            if (b & kLeafFlag) {
                const uint32_t c = b & ~kLeafFlag;
                bool placed = false;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const bool here = !placed && cnt[k] == 0;
                    first[k] = here ? a : first[k];
                    cnt[k] = here ? c : cnt[k];
                    placed = placed || here;
                }
                node++;
            } else {
                const bool in = fast ? slab_hit_fast(A.x, A.y, A.z, B.x, B.y, B.z, o, inv) : slab_hit(A.x, A.y, A.z, B.x, B.y, B.z, o, inv);
                node = in ? node + 1 : a;
            }
        }
        const uint64_t full = __ballot(cnt[K - 1] != 0), going = __ballot(cnt[K - 1] == 0 && node < n_nodes);
        if (__popcll(full) >= kLeafQuorum || going == 0ull) {
#pragma unroll
            for (int k = 0; k < K; k++) {
                for (uint32_t s = 0; s < cnt[k]; s++) {
                    const float4* tp = reinterpret_cast<const float4*>(S.tris + (first[k] + s));
                    float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
                    pin(T0); pin(T1); pin(T2);
                    bool acc = tri_hit<QUICK>(mk(T0.x, T0.y, T0.z), mk(T1.x, T1.y, T1.z), mk(T2.x, T2.y, T2.z), __float_as_uint(T0.w), o, d, h);
                    if (ANY && acc) return;
                }
                cnt[k] = 0;
            }
        }
    }
}

template <bool ANY>
__device__ __forceinline__ void traverse_lane(const SceneView& S, V3 o, V3 d, HitRec& h) {
    bool d_small, d_normal;
    dir_class(d, S.dir_bound, d_small, d_normal);
    const bool quick_det = __all(d_small), quick_inv = __all(d_normal);   // wave-uniform choices of the arithmetic
    const V3 inv = quick_inv ? ray_inverse_quick(d) : ray_inverse(d);
    const bool fast = quick_inv ? __all(finite3(o)) : __all(slab_safe(o, inv));
    if (quick_det) traverse_lane_walk<ANY, true>(S, o, d, inv, fast, h, 0u, S.n_nodes);
    else traverse_lane_walk<ANY, false>(S, o, d, inv, fast, h, 0u, S.n_nodes);
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
template <bool QUICK_DIV, bool AHEAD>
__device__ __forceinline__ void traverse_wave_closest_asm(const SceneView& S, V3 o, V3 d, V3 inv, HitRec& h, uint32_t next, uint32_t fast);
// Large scenes: one test of the box around all the meshes' roots (the spare record behind the node array) before the
// roots are tested one after the other.  Exact: for rays without NaN products (`fast`, wave-uniform) the slab test is
// monotone in the box, so a ray that misses this box fails the reference's test at every root.  (A root that is a leaf
// has no test in the reference: a scene with one gets a record that holds all of space, orion_hip_scene_create.)  Camera
// rays that miss a model of a dozen meshes are done after one box test instead of a dozen.  Smaller scenes are walked from inside
// (nothing to skip), and there every instruction counts.
constexpr uint32_t kSceneBoxAbove = 384;   // nodes
__device__ __forceinline__ void scene_box_gate(const SceneView& S, V3 o, V3 inv, bool fast, bool& active, uint32_t& next) {
    if (S.n_nodes > kSceneBoxAbove && fast) {
        const Node* all = S.nodes + S.n_nodes;
        const bool in = active && slab_hit_fast(all->lo[0], all->lo[1], all->lo[2], all->hi[0], all->hi[1], all->hi[2], o, inv);
        if (!in) next = kInvalid;
        active = in;
    }
}
// Closest hit only.  A point-light shadow query ("is anything hit at t >= 0") is answered by the
// same walk: a triangle is accepted against t = +inf under exactly the any-hit condition, and later
// acceptances never undo a hit, so "h.gtri is set" means the same in both.  (A separate any-hit
// loop that stopped at the first acceptance was 30 % slower than this walk run to its end.)
template <bool GATE>
__device__ __forceinline__ void traverse_wave_from(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) {
    bool d_small, d_normal;
    dir_class(d, S.dir_bound, d_small, d_normal);
    const bool quick_det = __all(!active || d_small), quick_inv = __all(!active || d_normal);   // wave-uniform
    const V3 inv = quick_inv ? ray_inverse_quick(d) : ray_inverse(d);
    // wave-uniform choice of the slab arithmetic; reciprocals of normal numbers below 2^126 are finite and not zero
    const bool fast = quick_inv ? __all(!active || finite3(o)) : __all(!active || slab_safe(o, inv));
    uint32_t next = active ? 0u : kInvalid;
    const uint32_t ufast = __builtin_amdgcn_readfirstlane(fast ? 1u : 0u);
    if (GATE) scene_box_gate(S, o, inv, fast, active, next);
    if (S.n_nodes && __any(active)) {   // (the walk relies on a lane being there to follow)
        if (quick_det) traverse_wave_closest_asm<true, false>(S, o, d, inv, h, next, ufast);
        else traverse_wave_closest_asm<false, false>(S, o, d, inv, h, next, ufast);
    }
}
__device__ __forceinline__ void traverse_wave(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) { traverse_wave_from<false>(S, o, d, h, active); }
// ... for rays that may come from outside a large scene (camera rays, coherent ray queries): the scene's box first.
__device__ __forceinline__ void traverse_wave_outside(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) { traverse_wave_from<true>(S, o, d, h, active); }

// The same walk for trees of thousands of nodes, whose records come from the L2 (the scalar cache holds a few hundred):
// there the loop waits for its scalar loads more than it computes (camera rays on a 13 k-node model: the waves of the
// first bounce spend 0.7 of their cycles waiting), and its AHEAD form asks for the next record in array order while it
// tests the present one (see ORION_NODE_FETCH_AHEAD below).  Same operations, same bits.
__device__ __forceinline__ void traverse_wave_ahead(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) {
    bool d_small, d_normal;
    dir_class(d, S.dir_bound, d_small, d_normal);
    const bool quick_det = __all(!active || d_small), quick_inv = __all(!active || d_normal);   // wave-uniform
    const V3 inv = quick_inv ? ray_inverse_quick(d) : ray_inverse(d);
    const bool fast = quick_inv ? __all(!active || finite3(o)) : __all(!active || slab_safe(o, inv));
    uint32_t next = active ? 0u : kInvalid;
    const uint32_t ufast = __builtin_amdgcn_readfirstlane(fast ? 1u : 0u);
    scene_box_gate(S, o, inv, fast, active, next);
    if (S.n_nodes && __any(active)) {
        if (quick_det) traverse_wave_closest_asm<true, true>(S, o, d, inv, h, next, ufast);
        else traverse_wave_closest_asm<false, true>(S, o, d, inv, h, next, ufast);
    }
}

// ---------------------------------------------------------------------------
// One ray per wave: every lane holds the SAME ray and takes one item of the
// scene (an inner node's box or a triangle), so the whole walk is one box test
// or one triangle test deep.  For the few thousand rays of a path tracer's
// last bounces, where a launch lasts as long as one ray's serial walk, this is
// what shortens it (some 50 dependent steps -> 1).
// The serial walk's result, restated for a parallel evaluation:
//   * a triangle is visited iff every inner node above it has its box hit
//     (leaves are not box-tested, orion/avx/sbvh.cpp:50-59);
//   * a visited triangle that passes det/u/v and t >= 0 is accepted iff
//     !(t > best so far) (orion/geometry.hpp:127), so the final hit is the
//     smallest t and, among equal t, the LAST in visit order;
//   * a NaN t passes both comparisons and then lets the next candidate replace
//     it: the walk effectively restarts after the last NaN candidate.
// Requires h.t == +inf on entry (true at every call site) and full exec.
// ---------------------------------------------------------------------------
template <bool ANY>
__device__ __forceinline__ void traverse_one(const SceneView& S, V3 o, V3 d, HitRec& h) {
    const uint32_t lane = threadIdx.x & 63u;
    const float4* ip = reinterpret_cast<const float4*>(S.items + lane);
    const float4 A = ip[0], B = ip[1], C = ip[2], D = ip[3];
    const uint32_t kind = __float_as_uint(B.w);
    const uint64_t anc = (uint64_t)__float_as_uint(D.x) | ((uint64_t)__float_as_uint(D.y) << 32);
    bool d_small, d_normal;
    dir_class(d, S.dir_bound, d_small, d_normal);
    d_small = __all(d_small); d_normal = __all(d_normal);   // every lane holds the same ray: branches, not selects
    const V3 inv = d_normal ? ray_inverse_quick(d) : ray_inverse(d);
    // boxes
    const bool in = slab_safe(o, inv) ? slab_hit_fast(A.x, A.y, A.z, B.x, B.y, B.z, o, inv) : slab_hit(A.x, A.y, A.z, B.x, B.y, B.z, o, inv);
    const uint64_t box_hit = __ballot(kind == 1u && in);
    // triangles: Triangle::intersect against t = +inf, every stage computed
    const float eps = 0.000001f;
    const V3 v0 = mk(A.x, A.y, A.z), e1 = mk(B.x, B.y, B.z), e2 = mk(C.x, C.y, C.z);
    const V3 pvec = cross(d, e2);
    const float det = dot_raw(e1, pvec);
    bool ok = !(det > -eps && det < eps);
    const float inv_det = d_small ? recip_quick(det) : 1.0f / det;   // only read where ok
    const V3 tvec = o - v0;
    const float du = dot_raw(tvec, pvec);
    const float u = du * inv_det;
    ok = ok && !(u < 0 || u > 1);
    const V3 qvec = cross(tvec, e1);
    const float dv = dot_raw(d, qvec);
    const float v = dv * inv_det;
    ok = ok && !(v < 0 || u + v > 1);
    const float dt = dot_raw(e2, qvec);
    const float t = dt * inv_det;
    ok = ok && !(t < 0);
    bool cand = kind == 2u && ok && (anc & ~box_hit) == 0ull;
    uint64_t cm = __ballot(cand);
    if (cm == 0ull) return;
    if (ANY) { h.gtri = 0u; return; }   // the callers of the any-hit form only ask whether there was one
    const uint64_t nanm = __ballot(cand && t != t);
    uint32_t winner;
    bool decided = false;
    if (nanm != 0ull) {
        const uint32_t last_nan = 63u - (uint32_t)__builtin_clzll(nanm);
        const uint64_t after = last_nan == 63u ? 0ull : (cm >> (last_nan + 1u)) << (last_nan + 1u);
        if (after == 0ull) { winner = last_nan; decided = true; }
        else { cm = after; cand = cand && lane > last_nan; }
    }
    if (!decided) {
        float m = cand ? t : __builtin_inff();
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const float other = __shfl_xor(m, sft);
            m = (other < m) ? other : m;
        }
        const uint64_t best = __ballot(cand && t == m);
        winner = 63u - (uint32_t)__builtin_clzll(best);
    }
    winner = __builtin_amdgcn_readfirstlane(winner);
    h.t = __shfl(canon(dt) * inv_det, (int)winner);
    h.u = __shfl(canon(du) * inv_det, (int)winner);
    h.v = __shfl(canon(dv) * inv_det, (int)winner);
    h.gtri = __shfl(__float_as_uint(A.w), (int)winner);
}

// The same walk for a wave whose lanes hold different rays: one ray after the other.
template <bool ANY>
__device__ __forceinline__ void traverse_one_each(const SceneView& S, V3 o, V3 d, HitRec& h, bool active) {
    const uint64_t todo = __ballot(active);
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t r = 0; r < 64u; r++) {
        if (!((todo >> r) & 1ull)) continue;
        const V3 ro = mk(__shfl(o.x, (int)r), __shfl(o.y, (int)r), __shfl(o.z, (int)r));
        const V3 rd = mk(__shfl(d.x, (int)r), __shfl(d.y, (int)r), __shfl(d.z, (int)r));
        HitRec hr; hr.t = __builtin_inff(); hr.u = 0.0f; hr.v = 0.0f; hr.gtri = kInvalid;
        traverse_one<ANY>(S, ro, rd, hr);
        if (lane == r) h = hr;
    }
}

// ---------------------------------------------------------------------------
// The wave-synchronous walk, hand-scheduled.  The FP32 operations of slab_hit / slab_hit_fast and tri_hit in the same
// order (the division is the compiler's own correctly rounded sequence, or recip_quick where
// QUICK_DIV says the wave's rays allow it), but
// with what the compiler will not do for this loop: one scalar pointer that
// steps through the node array, one 48-byte scalar fetch per triangle, the
// execution mask narrowed stage by stage with a single restore per triangle,
// and no register shuffling between loop levels.  The wave spends its time in
// this loop, one instruction at a time, so its length is the traversal's speed.
// `fast` (wave-uniform): slab_safe() holds for every active lane, the min/max
// slab test may be used.  Preconditions: n_nodes > 0, and the node array and the triangle array are
// each below 4 GiB (134 M nodes, 89 M triangles): positions are 32-bit byte offsets.  The scene upload
// checks it and never selects this walk for a larger scene.  SGPRs s52-s63 hold the
// node or triangle being processed.
// ---------------------------------------------------------------------------
/* How a node's record gets into s52-s59.  PLAIN: load it and wait.  AHEAD (trees of thousands of nodes, whose records come \
   from the L2: the walk then waits for its loads more than it computes): the record behind the present one in array order \
   -- where the walk goes from a leaf and from an inner node that a lane enters -- was asked for a node ago into s64-s71 \
   and is copied over; a jump to a skip link loads afresh.  The request for the next record goes out right after the wait, \
   so it is under way while this node is tested (the node array carries one spare record at its end). */ \
#define ORION_NODE_FETCH_PLAIN \
        "s_load_dwordx8 s[52:59], %[nodes], %[noff]\n"  /* lo.xyz a | hi.xyz b */ \
        "s_waitcnt lgkmcnt(0)\n"
/* ... and a leaf's triangles into s52-s63.  AHEAD: while a triangle is tested the next one of the leaf is on its way into \
   s72-s83 (the triangle array carries one spare record as well, though the look-ahead never leaves the leaf). */ \
#define ORION_TRI_FETCH_PLAIN \
        "L_tri%=:\n" \
        "s_load_dwordx8 s[52:59], %[tris], %[toff]\n"  /* v0.xyz gtri e1.xyz */ \
        "s_load_dwordx4 s[60:63], %[tris], %[toff] offset:0x20\n"  /* e2.xyz */ \
        "s_add_u32 %[toff], %[toff], 48\n" \
        "s_waitcnt lgkmcnt(0)\n"
#define ORION_TRI_NEXT_PLAIN \
        "s_cbranch_scc1 L_tri%=\n"
#define ORION_TRI_FETCH_AHEAD \
        ORION_TRI_FETCH_PLAIN \
        "L_tri_go%=:\n"  /* s52-s63 hold the triangle, toff is the offset of the one after it */ \
        "s_cmp_lt_u32 %[toff], %[cnt]\n" \
        "s_cbranch_scc0 L_tri_test%=\n" \
        "s_load_dwordx8 s[72:79], %[tris], %[toff]\n" \
        "s_load_dwordx4 s[80:83], %[tris], %[toff] offset:0x20\n" \
        "L_tri_test%=:\n"
#define ORION_TRI_NEXT_AHEAD \
        "s_cbranch_scc0 L_leaf_done%=\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "s_mov_b64 s[52:53], s[72:73]\n s_mov_b64 s[54:55], s[74:75]\n s_mov_b64 s[56:57], s[76:77]\n" \
        "s_mov_b64 s[58:59], s[78:79]\n s_mov_b64 s[60:61], s[80:81]\n s_mov_b64 s[62:63], s[82:83]\n" \
        "s_add_u32 %[toff], %[toff], 48\n" \
        "s_branch L_tri_go%=\n"
#define ORION_NODE_INIT_PLAIN
#define ORION_NODE_INIT_AHEAD "s_mov_b32 %[pfo], -1\n"  /* the offset the record in s64-s71 was asked from */
#define ORION_NODE_FETCH_AHEAD \
        "s_cmp_eq_u32 %[pfo], %[noff]\n" \
        "s_cbranch_scc1 L_have%=\n" \
        "s_load_dwordx8 s[52:59], %[nodes], %[noff]\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "s_branch L_go%=\n" \
        "L_have%=:\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "s_mov_b64 s[52:53], s[64:65]\n s_mov_b64 s[54:55], s[66:67]\n s_mov_b64 s[56:57], s[68:69]\n s_mov_b64 s[58:59], s[70:71]\n" \
        "L_go%=:\n" \
        "s_add_u32 %[pfo], %[noff], 32\n" \
        "s_load_dwordx8 s[64:71], %[nodes], %[pfo]\n"
#define ORION_WALK_BEFORE_DIV(NODE_FETCH_INIT, NODE_FETCH, TRI_FETCH) \
        "s_mov_b64 %[EX], exec\n" \
        "s_mov_b32 %[noff], 0\n"  /* byte offset of the node, steps by 32; the lanes' `next` is kept in the same unit */ \
        NODE_FETCH_INIT \
        "s_lshl_b32 %[nend], %[nn], 5\n" \
        "v_lshlrev_b32 %[nx], 5, %[nx]\n"  /* (kInvalid stays out of range) */ \
        "s_mov_b32 %[eps], 0x358637bd\n"  /* 1e-6f */ \
        "L_node%=:\n" \
        /* the lanes at this node: never none -- the walk only moves to where the lane furthest behind waits */ \
        "v_cmp_eq_u32_e64 %[MI], %[noff], %[nx]\n" \
        NODE_FETCH \
        "s_bitcmp1_b32 s59, 31\n"                                                                           \
        "s_cbranch_scc1 L_leaf%=\n"                                                                         \
        /* ---- inner node: slab test, next = in ? node + 1 : skip */                                       \
        "v_sub_f32 %[aa], s52, %[ox]\n"                                                                     \
        "v_sub_f32 %[bb], s56, %[ox]\n"                                                                     \
        "v_sub_f32 %[cc], s53, %[oy]\n"                                                                     \
        "v_sub_f32 %[uu], s57, %[oy]\n"                                                                     \
        "v_sub_f32 %[du], s54, %[oz]\n"                                                                     \
        "v_sub_f32 %[dv], s58, %[oz]\n"                                                                     \
        "v_mul_f32 %[aa], %[aa], %[ix]\n"                                                                   \
        "v_mul_f32 %[bb], %[bb], %[ix]\n"                                                                   \
        "v_mul_f32 %[cc], %[cc], %[iy]\n"                                                                   \
        "v_mul_f32 %[uu], %[uu], %[iy]\n"                                                                   \
        "v_mul_f32 %[du], %[du], %[iz]\n"                                                                   \
        "v_mul_f32 %[dv], %[dv], %[iz]\n"                                                                   \
        "s_cmp_eq_u32 %[fast], 0\n"                                                                         \
        "s_cbranch_scc1 L_slab_exact%=\n"                                                                   \
        "v_min_f32 %[tx], %[aa], %[bb]\n"                                                                   \
        "v_min_f32 %[ty], %[cc], %[uu]\n"                                                                   \
        "v_min_f32 %[tz], %[du], %[dv]\n"                                                                   \
        "v_max_f32 %[aa], %[aa], %[bb]\n"                                                                   \
        "v_max_f32 %[cc], %[cc], %[uu]\n"                                                                   \
        "v_max_f32 %[du], %[du], %[dv]\n"                                                                   \
        "v_max3_f32 %[tx], %[tx], %[ty], %[tz]\n"  /* tmin */                                               \
        "v_min3_f32 %[aa], %[aa], %[cc], %[du]\n"  /* tmax */                                               \
        "s_branch L_slab_done%=\n"                                                                          \
        /* the reference's select order, for waves with a ray whose products can be NaN: */                 \
        /* lx = (t1<t2)?t1:t2, hx = (t1>t2)?t1:t2, tmin = (lx<m)?m:lx with m = (ly<lz)?lz:ly, */            \
        /* tmax = (m2<hx)?m2:hx with m2 = (hz<hy)?hz:hy  (orion/AABB.hpp:79-99, orion/math.hpp:336-365) */  \
        "L_slab_exact%=:\n"                                                                                 \
        "v_cmp_lt_f32 vcc, %[aa], %[bb]\n s_nop 1\n v_cndmask_b32 %[tx], %[bb], %[aa], vcc\n"               \
        "v_cmp_lt_f32 vcc, %[cc], %[uu]\n s_nop 1\n v_cndmask_b32 %[ty], %[uu], %[cc], vcc\n"               \
        "v_cmp_lt_f32 vcc, %[du], %[dv]\n s_nop 1\n v_cndmask_b32 %[tz], %[dv], %[du], vcc\n"               \
        "v_cmp_gt_f32 vcc, %[aa], %[bb]\n s_nop 1\n v_cndmask_b32 %[aa], %[bb], %[aa], vcc\n"               \
        "v_cmp_gt_f32 vcc, %[cc], %[uu]\n s_nop 1\n v_cndmask_b32 %[cc], %[uu], %[cc], vcc\n"               \
        "v_cmp_gt_f32 vcc, %[du], %[dv]\n s_nop 1\n v_cndmask_b32 %[du], %[dv], %[du], vcc\n"               \
        "v_cmp_lt_f32 vcc, %[ty], %[tz]\n s_nop 1\n v_cndmask_b32 %[ty], %[ty], %[tz], vcc\n"               \
        "v_cmp_lt_f32 vcc, %[tx], %[ty]\n s_nop 1\n v_cndmask_b32 %[tx], %[tx], %[ty], vcc\n"               \
        "v_cmp_lt_f32 vcc, %[du], %[cc]\n s_nop 1\n v_cndmask_b32 %[cc], %[cc], %[du], vcc\n"               \
        "v_cmp_lt_f32 vcc, %[cc], %[aa]\n s_nop 1\n v_cndmask_b32 %[aa], %[aa], %[cc], vcc\n"               \
        "L_slab_done%=:\n"                                                                                  \
        "v_cmp_gt_f32 vcc, %[aa], %[tx]\n"                                                                  \
        "v_cmp_lt_f32_e64 %[PM], 0, %[aa]\n"                                                                \
        "v_add_u32_e64 %[bb], %[noff], 32\n" \
        "v_lshlrev_b32 %[cc], 5, s55\n" \
        "s_and_b64 vcc, vcc, %[PM]\n" \
        "v_cndmask_b32_e32 %[bb], %[cc], %[bb], vcc\n"                                                      \
        "v_cndmask_b32_e64 %[nx], %[nx], %[bb], %[MI]\n"                                                    \
        "s_and_b64 %[PM], vcc, %[MI]\n"                                                                     \
        "s_cbranch_scc1 L_next%=\n"                                                                         \
        /* no lane enters: the walk resumes at the skip link (lanes get into a subtree through its root only) */ \
        "s_lshl_b32 %[noff], s55, 5\n"                                                                      \
        "s_branch L_test%=\n"                                                                               \
        /* ---- leaf: its triangles, for the lanes that are here */                                         \
        "L_leaf%=:\n"                                                                                       \
        "s_and_b32 %[cnt], s59, 0x7fffffff\n"  /* triangle count */ \
        "s_cbranch_scc0 L_leaf_done%=\n" \
        "s_mul_i32 %[toff], s55, 48\n"  /* byte offset of the leaf's first triangle */ \
        "s_mul_i32 %[cnt], %[cnt], 48\n" \
        "s_add_u32 %[cnt], %[cnt], %[toff]\n"  /* ... and of the end of its last */ \
        "s_and_b64 exec, %[EX], %[MI]\n" \
        TRI_FETCH \
        /* pvec = cross(d, e2) */                                                                           \
        "v_mul_f32 %[px], s62, %[dy]\n"                                                                     \
        "v_mul_f32 %[py], s60, %[dz]\n"                                                                     \
        "v_mul_f32 %[pz], s61, %[dx]\n"                                                                     \
        "v_fma_f32 %[px], -%[dz], s61, %[px]\n"                                                             \
        "v_fma_f32 %[py], -%[dx], s62, %[py]\n"                                                             \
        "v_fma_f32 %[pz], -%[dy], s60, %[pz]\n"                                                             \
        /* det = dot(e1, pvec) */                                                                           \
        "v_mul_f32 %[aa], s56, %[px]\n"                                                                     \
        "v_mul_f32 %[bb], s57, %[py]\n"                                                                     \
        "v_mul_f32 %[cc], s58, %[pz]\n"                                                                     \
        "v_add_f32 %[aa], %[aa], %[bb]\n"                                                                   \
        "v_add_f32 %[det], %[aa], %[cc]\n"                                                                  \
        "v_cmpx_nlt_f32_e64 vcc, |%[det]|, %[eps]\n"  /* the comparison narrows the execution mask itself */ \
        /* (no exit here: a whole wave parallel to a triangle is too rare to pay for a branch per test) */
#define ORION_WALK_DIV_EXACT \
        /* inv_det = 1 / det, correctly rounded (q0 = aa, r = bb, q1 = cc, e = tx, m = dt) */  \
        "v_div_scale_f32 %[aa], %[PM], %[det], %[det], 1.0\n"                                  \
        "v_rcp_f32 %[bb], %[aa]\n"                                                             \
        "v_div_scale_f32 %[cc], vcc, 1.0, %[det], 1.0\n"                                       \
        "v_fma_f32 %[tx], -%[aa], %[bb], 1.0\n"                                                \
        "v_fmac_f32 %[bb], %[tx], %[bb]\n"                                                     \
        "v_mul_f32 %[dt], %[cc], %[bb]\n"                                                      \
        "v_fma_f32 %[tx], -%[aa], %[dt], %[cc]\n"                                              \
        "v_fmac_f32 %[dt], %[tx], %[bb]\n"                                                     \
        "v_fma_f32 %[aa], -%[aa], %[dt], %[cc]\n"                                              \
        "v_div_fmas_f32 %[aa], %[aa], %[bb], %[dt]\n"                                          \
        "v_div_fixup_f32 %[idet], %[aa], %[det], 1.0\n"                                       \
        /* tvec = o - v0 */ \
        "v_subrev_f32 %[tx], s52, %[ox]\n"                                               \
        "v_subrev_f32 %[ty], s53, %[oy]\n"                                               \
        "v_subrev_f32 %[tz], s54, %[oz]\n"
#define ORION_WALK_DIV_QUICK \
        /* inv_det = 1 / det as reciprocal estimate + one Newton step: the bits of the division for */  \
        /* 2^-126 <= |det| < 2^126 (every such float checked, tools/micro/rcp_exact.hip) */             \
        "v_rcp_f32 %[bb], %[det]\n" \
        /* tvec = o - v0 (also the wait state a reciprocal needs before its result is read) */ \
        "v_subrev_f32 %[tx], s52, %[ox]\n"                                               \
        "v_subrev_f32 %[ty], s53, %[oy]\n"                                               \
        "v_subrev_f32 %[tz], s54, %[oz]\n"                                               \
        "v_fma_f32 %[aa], -%[det], %[bb], 1.0\n" \
        "v_fma_f32 %[idet], %[aa], %[bb], %[bb]\n"
#define ORION_WALK_AFTER_DIV(TRI_NEXT) \
        /* u = dot(tvec, pvec) * inv_det */ \
        "v_mul_f32 %[aa], %[tx], %[px]\n"                                                \
        "v_mul_f32 %[bb], %[ty], %[py]\n"                                                \
        "v_mul_f32 %[cc], %[tz], %[pz]\n"                                                \
        "v_add_f32 %[aa], %[aa], %[bb]\n"                                                \
        "v_add_f32 %[du], %[aa], %[cc]\n"                                                \
        "v_mul_f32 %[uu], %[du], %[idet]\n"                                              \
        "v_cmpx_ngt_f32 vcc, 0, %[uu]\n" \
        "v_cmpx_nlt_f32 vcc, 1.0, %[uu]\n" \
        "s_cbranch_execz L_tri_next%=\n" \
        /* qvec = cross(tvec, e1) (in pvec's registers), v = dot(d, qvec) * inv_det */   \
        "v_mul_f32 %[px], s58, %[ty]\n"                                                  \
        "v_mul_f32 %[py], s56, %[tz]\n"                                                  \
        "v_mul_f32 %[pz], s57, %[tx]\n"                                                  \
        "v_fma_f32 %[px], -%[tz], s57, %[px]\n"                                          \
        "v_fma_f32 %[py], -%[tx], s58, %[py]\n"                                          \
        "v_fma_f32 %[pz], -%[ty], s56, %[pz]\n"                                          \
        "v_mul_f32 %[aa], %[dx], %[px]\n"                                                \
        "v_mul_f32 %[bb], %[dy], %[py]\n"                                                \
        "v_mul_f32 %[cc], %[dz], %[pz]\n"                                                \
        "v_add_f32 %[aa], %[aa], %[bb]\n"                                                \
        "v_add_f32 %[dv], %[aa], %[cc]\n"                                                \
        "v_mul_f32 %[aa], %[dv], %[idet]\n"                                              \
        "v_add_f32 %[bb], %[uu], %[aa]\n" \
        "v_cmpx_ngt_f32 vcc, 0, %[aa]\n" \
        "v_cmpx_nlt_f32 vcc, 1.0, %[bb]\n" \
        "s_cbranch_execz L_tri_next%=\n" \
        /* t = dot(e2, qvec) * inv_det */                                                \
        "v_mul_f32 %[aa], s60, %[px]\n"                                                  \
        "v_mul_f32 %[bb], s61, %[py]\n"                                                  \
        "v_mul_f32 %[cc], s62, %[pz]\n"                                                  \
        "v_add_f32 %[aa], %[aa], %[bb]\n"                                                \
        "v_add_f32 %[dt], %[aa], %[cc]\n"                                                \
        "v_mul_f32 %[bb], %[dt], %[idet]\n"                                              \
        "v_cmpx_ngt_f32 vcc, %[bb], %[ht]\n" \
        "v_cmpx_ngt_f32 vcc, 0, %[bb]\n" \
        "s_cbranch_execz L_tri_next%=\n" \
        /* accepted: the +0.0f lane of the reference's dot products, then the record */  \
        "v_add_f32 %[aa], 0, %[du]\n"                                                    \
        "v_add_f32 %[cc], 0, %[dv]\n"                                                    \
        "v_add_f32 %[bb], 0, %[dt]\n"                                                    \
        "v_mul_f32 %[hu], %[aa], %[idet]\n"                                              \
        "v_mul_f32 %[hv], %[cc], %[idet]\n"                                              \
        "v_mul_f32 %[ht], %[bb], %[idet]\n"                                              \
        "v_mov_b32 %[hg], s55\n"                                                         \
        "L_tri_next%=:\n"                                                                \
        "s_and_b64 exec, %[EX], %[MI]\n" \
        "s_cmp_lt_u32 %[toff], %[cnt]\n" \
        TRI_NEXT \
        "L_leaf_done%=:\n"                                                               \
        "s_mov_b64 exec, %[EX]\n" \
        "v_add_u32_e64 %[bb], %[noff], 32\n" \
        "v_cndmask_b32_e64 %[nx], %[nx], %[bb], %[MI]\n" \
        "L_next%=:\n" \
        "s_add_u32 %[noff], %[noff], 32\n" \
        "L_test%=:\n" \
        "s_cmp_lt_u32 %[noff], %[nend]\n" \
        "s_cbranch_scc1 L_node%=\n" \
        "s_waitcnt lgkmcnt(0)\n"  /* (AHEAD: the last request must have landed before s64-s71 are anyone else's) */ \
        "s_mov_b64 exec, %[EX]\n"                                                       
#define ORION_WALK_OPERANDS(EXTRA_OUT) \
        : EXTRA_OUT [nx] "+v"(next), [ht] "+v"(ht), [hu] "+v"(hu), [hv] "+v"(hv), [hg] "+v"(hg),                                          \
          [px] "=&v"(px), [py] "=&v"(py), [pz] "=&v"(pz), [det] "=&v"(det), [idet] "=&v"(idet),                                 \
          [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz),                                                                       \
          [aa] "=&v"(aa), [bb] "=&v"(bb), [cc] "=&v"(cc), [uu] "=&v"(uu), [du] "=&v"(du), [dv] "=&v"(dv), [dt] "=&v"(dt),       \
          [EX] "=&s"(EX), [MI] "=&s"(MI), [PM] "=&s"(PM),                                                                       \
          [noff] "=&s"(noff), [toff] "=&s"(toff), [cnt] "=&s"(cnt), [nend] "=&s"(nend), [eps] "=&s"(eps)  \
        : [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z),                             \
          [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z),                                                                    \
          [nodes] "s"(S.nodes), [tris] "s"(S.tris), [nn] "s"(S.n_nodes), [fast] "s"(fast)                                       \
        : "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "vcc", "scc"
#define ORION_WALK_OUT_PLAIN
#define ORION_WALK_OUT_AHEAD [pfo] "=&s"(pfo),
#define ORION_WALK_CLOBBER_PLAIN );
#define ORION_WALK_CLOBBER_AHEAD , "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83");

template <bool QUICK_DIV, bool AHEAD>
__device__ __forceinline__ void traverse_wave_closest_asm(const SceneView& S, V3 o, V3 d, V3 inv, HitRec& h, uint32_t next, uint32_t fast) {
    float ht = h.t, hu = h.u, hv = h.v;
    uint32_t hg = h.gtri;
    // VGPR temporaries; the division's five and qvec share registers with values that are dead or not live yet
    float px, py, pz, det, idet, tx, ty, tz, aa, bb, cc, uu, du, dv, dt;
    // SGPR temporaries: EX = lanes that entered, MI = lanes at the current node, PM = a stage's verdict
    unsigned long long EX, MI, PM;
    uint32_t noff, nend, toff, cnt, eps, pfo;
    fast = __builtin_amdgcn_readfirstlane(fast);
    if (AHEAD) {
        if (QUICK_DIV) asm volatile(ORION_WALK_BEFORE_DIV(ORION_NODE_INIT_AHEAD, ORION_NODE_FETCH_AHEAD, ORION_TRI_FETCH_AHEAD) ORION_WALK_DIV_QUICK ORION_WALK_AFTER_DIV(ORION_TRI_NEXT_AHEAD) ORION_WALK_OPERANDS(ORION_WALK_OUT_AHEAD) ORION_WALK_CLOBBER_AHEAD
        else asm volatile(ORION_WALK_BEFORE_DIV(ORION_NODE_INIT_AHEAD, ORION_NODE_FETCH_AHEAD, ORION_TRI_FETCH_AHEAD) ORION_WALK_DIV_EXACT ORION_WALK_AFTER_DIV(ORION_TRI_NEXT_AHEAD) ORION_WALK_OPERANDS(ORION_WALK_OUT_AHEAD) ORION_WALK_CLOBBER_AHEAD
    } else {
        if (QUICK_DIV) asm volatile(ORION_WALK_BEFORE_DIV(ORION_NODE_INIT_PLAIN, ORION_NODE_FETCH_PLAIN, ORION_TRI_FETCH_PLAIN) ORION_WALK_DIV_QUICK ORION_WALK_AFTER_DIV(ORION_TRI_NEXT_PLAIN) ORION_WALK_OPERANDS(ORION_WALK_OUT_PLAIN) ORION_WALK_CLOBBER_PLAIN
        else asm volatile(ORION_WALK_BEFORE_DIV(ORION_NODE_INIT_PLAIN, ORION_NODE_FETCH_PLAIN, ORION_TRI_FETCH_PLAIN) ORION_WALK_DIV_EXACT ORION_WALK_AFTER_DIV(ORION_TRI_NEXT_PLAIN) ORION_WALK_OPERANDS(ORION_WALK_OUT_PLAIN) ORION_WALK_CLOBBER_PLAIN
    }
    (void)pfo;
    h.t = ht; h.u = hu; h.v = hv; h.gtri = hg;
}

#undef ORION_WALK_BEFORE_DIV
#undef ORION_WALK_DIV_EXACT
#undef ORION_WALK_DIV_QUICK
#undef ORION_WALK_AFTER_DIV
#undef ORION_WALK_OPERANDS
#undef ORION_WALK_CLOBBER_PLAIN
#undef ORION_WALK_OUT_PLAIN
#undef ORION_WALK_OUT_AHEAD
#undef ORION_WALK_CLOBBER_AHEAD
#undef ORION_NODE_FETCH_PLAIN
#undef ORION_NODE_FETCH_AHEAD
#undef ORION_NODE_INIT_PLAIN
#undef ORION_NODE_INIT_AHEAD
#undef ORION_TRI_FETCH_PLAIN
#undef ORION_TRI_FETCH_AHEAD
#undef ORION_TRI_NEXT_PLAIN
#undef ORION_TRI_NEXT_AHEAD

} // namespace orion_dev
