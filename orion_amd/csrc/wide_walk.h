// The walk of large scenes: every lane takes its ray through the 4-wide tree (WideNode,
// device_types.h; built by wide_bvh.h) nearest slot first, skipping what cannot matter, and still
// returns what the reference's unordered, unculled walk returns (orion/avx/sbvh.cpp:36-83,
// orion/model.hpp:52-62), bit for bit:
//
//   * Visit rule.  The reference tests a triangle iff every binary inner node above it passes
//     AABB::intersect.  A wide node carries the exact box of the binary inner node it stands for and the
//     walk tests it on entry with the reference's arithmetic; the inner nodes collapsed away above it
//     contain that box, and for a ray whose slab products cannot be NaN the test is monotone in the box,
//     so they pass whenever it does.  Leaves hang from the wide node of their own binary parent.
//   * Result rule.  The reference keeps a candidate unless t > best (orion/geometry.hpp:127), so it ends
//     with the smallest t and, among equal t, the LAST triangle in its visit order.  Triangle slots are
//     stored in that order (meshes in load order, each tree depth first), so in any visiting order
//     "t < best, or t == best and slot > best slot" selects the same triangle with the same t, u, v.
//   * Skipping.  A slot's decoded box is grown by the scene's pad (it then contains the slot's real box) and further by
//     kWideSlack times (the origin's distance to the farthest corner of the node's box + two pads).  An inner slot
//     missed by the ray would fail the reference's own test on the (smaller) real box: exact.  Two more skips need the
//     reference's computed hit point o + t*d of a triangle it accepts to lie inside the triangle's box grown by that much:
//     a LEAF whose grown box the ray misses, and ANY slot whose grown box begins beyond the best hit so far.
//   * The guard: when that holds.  Triangle::intersect (orion/geometry.hpp:96-124) computes det = e1 . (d x e2) and
//     t, u, v as three numerators over det, in FP32 (eps = 2^-24) with |error of a cross product| <= 3.2 eps |a||b| (one
//     product rounded, one fused) and |error of a dot product| <= 3 eps |a||b|.  So |det - D| <= 6.5 eps |d||e1||e2| for the
//     exact D = d . (e1 x e2), and each numerator is within 7.2 eps |o - v0| (times the other two lengths) of its exact value.
//     If |D| >= kappa |d||e1||e2| (WELL CONDITIONED, kappa = kGuardKappa = 2^-11): the error of det scales all three quotients
//     alike, which moves the point o + t*d against v0 + u*e1 + v*e2 by at most (6.5 eps / kappa) |o - v0|; the numerators'
//     errors move it along e1, e2 and d by at most (7.2 eps / kappa) |o - v0| each: 28.1 eps |o - v0| / kappa in all.
//     v0 + u*e1 + v*e2 is a point of the triangle (the accepted u, v, u + v lie in [0, 1]) and |o - v0| <= sqrt(3) (reach + pad),
//     so the computed hit lies within 48.7 eps / kappa = 0.0059 of (reach + pad) of the triangle's box: 0.76 of the slack
//     (tools/ab/guard_bound.py: the largest ratio seen on 5 * 10^6 accepted pairs near the threshold is 2.7 of the 28.1).
//     The rest of the slack covers the rounding of the slot test itself (a few 10^-4 of it).  Both skips are then sound: the
//     ray meets the triangle's grown box at the computed t.
//     If |D| < kappa |d||e1||e2| (ILL CONDITIONED: the ray lies within asin(kappa / sin(apex)), about 0.03 degrees, of the
//     triangle's plane) nothing bounds t, u, v: they are rounding residue over a tiny determinant, and the reference accepts
//     them one time in ten -- UNLESS |d||e1||e2| <= kGuardHarmless = 2.0e-3, for then |det| <= (kappa + 6.5 eps) |d||e1||e2| <
//     1e-6 and the reference itself rejects the triangle (orion/geometry.hpp:103).
//     The scene therefore carries a table (wide_guard.h; built on the device when the scene is created): for every node and
//     every cell of a grid over all ray directions, four bits -- may slot k hold a triangle that is ill conditioned for SOME
//     direction of the cell.  A slot whose bit is set for the ray's cell is DANGEROUS.  A dangerous leaf is visited whatever
//     its box says, as in the reference; a dangerous inner slot iff the ray meets its grown box anywhere (if not, the
//     reference's test of that node's own box fails too); and neither is ever dropped for beginning beyond the best hit.
//     The nodes above the meshes' roots (wide_bvh.h: virtual nodes, numbered first) have no box in the reference -- a mesh whose
//     tree is one leaf hangs from them, and the reference tests its triangles for every ray.  For a ray that needs the guard
//     their slots are therefore never dropped by distance (a slot whose grown box the ray meets anywhere is visited: every
//     mesh the reference walks into has its root's box inside these boxes), and their bits in the table mean FORCED: slot k
//     is a leaf with an ill-conditioned triangle, or a virtual node with a forced slot -- visited whatever its box says, and
//     a virtual node with a forced slot is entered whatever its own box says.  (A mesh's own inner node is never forced: the
//     reference tests its box.)
//     Rays no longer than kGuardHarmless / (the scene's largest |e1||e2|) need no guard.  With that, every triangle the
//     reference accepts is either tested here or provably cannot beat the result.
//     What it costs: one more byte loaded per node step, and the subtrees that are opened for nothing -- every closed surface
//     has a silhouette for every direction, so the top of the tree is dangerous for every ray and the walk opens what the
//     ray's LINE meets down to where the cells tell the silhouette from the rest (a 19 k-triangle model, wide nodes per ray:
//     bounce rays 17.6 -> 20.9, rays across the model 21.0 -> 56.8; a scene of small triangles: none).  Option "wide_guard" = 0
//     switches the guard off: the walk then differs from the reference on rays aimed ALONG a triangle's plane at a slope
//     below 1e-5 (349 of 2^30 such rays on that model, none of 1.6 * 10^10 natural rays: tools/ab/stress_wide.py).
//   * Rays the argument does not cover -- an origin component that is not finite or beyond the scene's
//     bound, a direction component that is zero, denormal, not finite or outside 2^-40 .. 2^40 -- and
//     lanes whose stack would overflow take the unculled walk (traverse_lane) instead.
#pragma once

#include "intersect.h"
#include "wide_guard.h"

namespace orion_dev {

constexpr int kWideBlock = 256;    // threads per block of every kernel that walks the wide tree
#ifndef ORION_WIDE_STACK
#define ORION_WIDE_STACK 16
#endif
constexpr int kWideStack = ORION_WIDE_STACK;     // stack entries per lane (LDS: 6 bytes each)

// One lane's stack entry: what to visit (4 bytes) and where its grown box begins along the ray, as the
// upper half of the float rounded down (2 bytes: a bound from below is all the entry is used for).
// [depth][thread]: a lane's entries sit in the same banks at every depth.  One row more than the stack is
// deep: a node step writes its candidates without asking which of them count (wide_node_step), and the one
// that does not count lands on the first free row.
__shared__ uint32_t g_wide_ref[kWideStack + 1][kWideBlock];
__shared__ uint16_t g_wide_t[kWideStack + 1][kWideBlock];

__device__ __forceinline__ void wide_push(uint32_t sp, uint32_t tid, uint32_t ref, float tmin) {
    // Upper 16 bits: truncation rounds a positive float down; a negative entry distance (origin inside
    // the box) becomes 0, which is below every best hit as well.
    g_wide_ref[sp][tid] = ref;
    g_wide_t[sp][tid] = (uint16_t)(__float_as_uint(vmax(tmin, 0.0f)) >> 16);
}

// (compiles to one v_cvt_f32_ubyteK)
template <int K> __device__ __forceinline__ float byte_of(uint32_t w) { return (float)((w >> (8 * K)) & 0xFFu); }

// What a lane is doing: `cur` is kWideEmpty (nothing left), a wide node's index, or a leaf reference.
struct WideState { uint32_t cur, sp, best_slot; };

__device__ __forceinline__ bool wide_at_node(const WideState& w) { return !(w.cur & kWideLeaf) && w.cur != kWideEmpty; }
__device__ __forceinline__ bool wide_at_leaf(const WideState& w) { return (w.cur & kWideLeaf) != 0u; }

// Next entry from the lane's stack; entries whose box begins beyond the best hit by now are dropped unvisited.
__device__ __forceinline__ void wide_pop(WideState& w, float best, uint32_t tid) {
    w.cur = kWideEmpty;
    while (w.sp) {
        w.sp--;
        const uint32_t ref = g_wide_ref[w.sp][tid];   // (both halves of the entry in one round trip)
        const float t = __uint_as_float((uint32_t)g_wide_t[w.sp][tid] << 16);
        if (!(t > best)) { w.cur = ref; break; }
    }
}

__device__ __forceinline__ float vmax3_abs(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// The byte words of a node's four slots, picked per axis by the direction's sign: where the ray enters the slots' slabs
// (the lower bounds if it runs up the axis, the upper bounds if it runs down) and where it leaves them.
struct WideBytes { uint32_t nx, ny, nz, fx, fy, fz; };

// The guard (see the header comment): the cell of the ray's direction in the scene's table of dangerous slots
// (wide_guard.h), or kGuardNoCell for a ray that needs no guard.
struct WideGuard { uint32_t cell; };

__device__ __forceinline__ WideGuard wide_guard_of(const SceneView& S, V3 d) {
    WideGuard g; g.cell = kGuardNoCell;
    const float l2 = fmaf(d.x, d.x, fmaf(d.y, d.y, d.z * d.z));
    // Rays no longer than wide_guard_dlen (with room for this sum's rounding) cannot be accepted by an ill-conditioned test
    // (device_types.h: kGuardHarmless).  Tame directions (wide_tame) have components between 2^-40 and 2^40: l2 is a normal number.
    if (S.wide_guard != 0u && !(l2 * 1.00001f <= S.wide_guard_dlen * S.wide_guard_dlen)) g.cell = guard_cell_of(d.x, d.y, d.z, S.wide_cells_grid);
    return g;
}

// Slot K: its entry distance, or +inf if it is not to be visited; `keep` = it may not be dropped later however near the
// best hit has come by then (it goes onto the stack with distance 0).  The slab distances of the grown box come straight
// from the bytes: byte * c + l where the ray enters the slab, byte * c + u where it leaves it (wide_node_step computes
// c, l, u per axis).  A slot that is not dangerous is skipped when it is empty, missed, behind the origin or beyond `best`.  A
// dangerous inner slot is visited iff the ray meets its grown box at all, however far away (if it does not, the
// reference's own test of the inner node's box fails too); a dangerous leaf is visited whatever its box says, as in the
// reference.  Slots visited only because they are dangerous come after the others (distance kWideLast).
constexpr float kWideLast = 3.0e38f;
// `vrule`: the lane's ray needs the guard and the node is a virtual one (see the header comment): its slots are visited iff
// the ray meets their grown box anywhere or their bit is set, and never dropped later.
template <int K>
__device__ __forceinline__ float wide_slot_guarded(const WideBytes& q, V3 c, V3 l, V3 u, float best, uint32_t ref, bool vrule, bool bit, bool& keep) {
    const float t1x = fmaf(byte_of<K>(q.nx), c.x, l.x), t2x = fmaf(byte_of<K>(q.fx), c.x, u.x);
    const float t1y = fmaf(byte_of<K>(q.ny), c.y, l.y), t2y = fmaf(byte_of<K>(q.fy), c.y, u.y);
    const float t1z = fmaf(byte_of<K>(q.nz), c.z, l.z), t2z = fmaf(byte_of<K>(q.fz), c.z, u.z);
    const float tmin = vmax3(t1x, t1y, t1z);
    const float tmax = vmin3(t2x, t2y, t2z);
    const bool met = vmax(tmin, 0.0f) <= tmax;            // the ray meets the grown box somewhere
    const bool within = met & (tmin <= best);
    const bool danger = bit | vrule;                       // not to be dropped by distance
    const bool boxless = vrule ? bit : (ref & kWideLeaf) != 0u;   // visited even if the box is missed (given `danger`)
    keep = danger;
    const bool vis = (ref != kWideEmpty) & (within | (danger & (met | boxless)));
    return vis ? (within ? tmin : kWideLast) : __builtin_inff();
}

// One wide node: the reference's test on its own box, then the four slots; the nearest one is next and the
// others go onto the stack.  Returns false, with the lane still at the node, if the stack would overflow
// (wide_overflow then deals with the node's subtree).  `guard`: wave-uniform, whether any lane's ray needs the guard.
__device__ __forceinline__ bool wide_node_step(const SceneView& S, V3 o, V3 inv, float best, WideState& w, uint32_t tid, WideGuard g, bool guard) {
    const float4* np = reinterpret_cast<const float4*>(S.wide + w.cur);
    float4 A = np[0], B = np[1], C = np[2], D = np[3];
    uint8_t cells = 0;
    if (guard && g.cell != kGuardNoCell) cells = S.wide_cells[guard_table_offset(w.cur, g.cell, S.wide_cells_grid * S.wide_cells_grid)];
    pin(A); pin(B); pin(C); pin(D);
    const float inf = __builtin_inff();
    float t0 = inf, t1 = inf, t2 = inf, t3 = inf;
    bool k0 = false, k1 = false, k2 = false, k3 = false;
    const uint32_t r0 = __float_as_uint(A.w), r1 = __float_as_uint(B.w), r2 = __float_as_uint(D.z), r3 = __float_as_uint(D.w);
    // AABB::intersect on the node's own box (orion/AABB.hpp:79-99; the form of slab_hit_fast), keeping its terms
    const V3 n0 = mk(A.x - o.x, A.y - o.y, A.z - o.z), n1 = mk(B.x - o.x, B.y - o.y, B.z - o.z);
    const V3 s0 = n0 * inv, s1 = n1 * inv;
    const float own_min = vmax3(vmin(s0.x, s1.x), vmin(s0.y, s1.y), vmin(s0.z, s1.z));
    const float own_max = vmin3(vmax(s0.x, s1.x), vmax(s0.y, s1.y), vmax(s0.z, s1.z));
    // Which slots are dangerous: four bits of the table (wide_guard.h), asked for together with the node; only in waves that
    // hold a ray that needs the guard.
    uint32_t bits = 0u;
    if (guard) bits = g.cell == kGuardNoCell ? 0u : ((uint32_t)cells >> ((w.cur & 1u) << 2)) & 15u;   // (two nodes per byte)
    // A virtual node's box is not the reference's: it is entered whenever one of its slots is forced.
    const bool vrule = w.cur < S.wide_n_virtual && g.cell != kGuardNoCell;
    if ((own_max > own_min && own_max > 0.0f) || (vrule && bits != 0u)) {
        // The grid's plane at byte q lies at lo + q * (hi - lo) / 254; its slab distance is s0 + q * (s1 - s0) / 254, and
        // that of a plane moved by g against / along the axis s0 -+ g * inv (the rounding of these forms is
        // ten-thousandths of the slack).  g = pad + slack * (reach + 2 pad), reach = the origin's distance to the
        // farthest corner of the node's box.
        const float reach = vmax(vmax3_abs(n0.x, n1.x, n0.y), vmax3_abs(n1.y, n0.z, n1.z));
        const float gr = fmaf(kWideSlack, reach, S.wide_grow0);
        // Which of a slab's two planes the ray meets first is the sign of the direction (bytes: lower <= upper, c has the
        // sign of inv, l and u below lie on the entry and on the exit side), so the entry and exit distances are
        // picked per node, one word of four slots at a time, not compared per slot.
        const V3 c = mk((s1.x - s0.x) * kWideInvCells, (s1.y - s0.y) * kWideInvCells, (s1.z - s0.z) * kWideInvCells);
        const V3 l = mk(fmaf(-gr, fabsf(inv.x), s0.x), fmaf(-gr, fabsf(inv.y), s0.y), fmaf(-gr, fabsf(inv.z), s0.z));
        const V3 u = mk(fmaf(gr, fabsf(inv.x), s0.x), fmaf(gr, fabsf(inv.y), s0.y), fmaf(gr, fabsf(inv.z), s0.z));
        const bool ux = inv.x > 0.0f, uy = inv.y > 0.0f, uz = inv.z > 0.0f;
        const uint32_t lx = __float_as_uint(C.x), ly = __float_as_uint(C.y), lz = __float_as_uint(C.z);
        const uint32_t hx = __float_as_uint(C.w), hy = __float_as_uint(D.x), hz = __float_as_uint(D.y);
        const WideBytes q = {ux ? lx : hx, uy ? ly : hy, uz ? lz : hz, ux ? hx : lx, uy ? hy : ly, uz ? hz : lz};
        const bool d0 = (bits & 1u) != 0u, d1 = (bits & 2u) != 0u, d2 = (bits & 4u) != 0u, d3 = (bits & 8u) != 0u;
        t0 = wide_slot_guarded<0>(q, c, l, u, best, r0, vrule, d0, k0);
        t1 = wide_slot_guarded<1>(q, c, l, u, best, r1, vrule, d1, k1);
        t2 = wide_slot_guarded<2>(q, c, l, u, best, r2, vrule, d2, k2);
        t3 = wide_slot_guarded<3>(q, c, l, u, best, r3, vrule, d3, k3);
    }
    const float tn = vmin(vmin3(t0, t1, t2), t3);
    if (!(tn < inf)) { wide_pop(w, best, tid); return true; }
    // The first slot at distance tn is next; the other slots that were hit go onto the stack, the farthest slot
    // number first.  No branches: every candidate is written to the row above the stack and the stack only
    // grows past it if the candidate counts (the spare row takes the last one that does not).
    const bool e0 = t0 == tn, e1 = t1 == tn, e2 = t2 == tn;
    const bool p0 = (t0 < inf) & !e0, p1 = (t1 < inf) & (e0 | !e1), p2 = (t2 < inf) & (e0 | e1 | !e2), p3 = (t3 < inf) & (e0 | e1 | e2);
    const uint32_t more = (p0 ? 1u : 0u) + (p1 ? 1u : 0u) + (p2 ? 1u : 0u) + (p3 ? 1u : 0u);
    if (w.sp + more > (uint32_t)kWideStack) return false;   // (w.cur still names the node: see wide_overflow)
    uint32_t sp = w.sp;
    // (a slot that is kept goes on with distance 0: wide_pop never drops it)
    wide_push(sp, tid, r3, k3 ? 0.0f : t3); sp += p3 ? 1u : 0u;
    wide_push(sp, tid, r2, k2 ? 0.0f : t2); sp += p2 ? 1u : 0u;
    wide_push(sp, tid, r1, k1 ? 0.0f : t1); sp += p1 ? 1u : 0u;
    wide_push(sp, tid, r0, k0 ? 0.0f : t0); sp += p0 ? 1u : 0u;
    w.sp = sp;
    w.cur = e0 ? r0 : (e1 ? r1 : (e2 ? r2 : r3));
    return true;
}

// A leaf: its triangles in slot order, each against t = +inf (Triangle::intersect's candidate test), the
// result rule of the header comment deciding between candidates.  Returns true when ANY is set and the ray
// has a hit (the any-hit callers only ask whether there is one).
template <bool ANY, bool QUICK>
__device__ __forceinline__ bool wide_leaf_step(const SceneView& S, V3 o, V3 d, HitRec& h, WideState& w, uint32_t tid) {
    const uint32_t first = w.cur & 0x01FFFFFFu, count = (w.cur >> 25) & 63u;
    for (uint32_t s = 0; s < count; s++) {
        const float4* tp = reinterpret_cast<const float4*>(S.tris + (first + s));
        float4 T0 = tp[0], T1 = tp[1], T2 = tp[2];
        pin(T0); pin(T1); pin(T2);
        HitRec c; c.t = __builtin_inff(); c.u = 0.0f; c.v = 0.0f; c.gtri = kInvalid;
        if (tri_hit<QUICK>(mk(T0.x, T0.y, T0.z), mk(T1.x, T1.y, T1.z), mk(T2.x, T2.y, T2.z), __float_as_uint(T0.w), o, d, c)) {
            if (c.t < h.t || (c.t == h.t && first + s > w.best_slot)) { h = c; w.best_slot = first + s; }
            if (ANY) { w.cur = kWideEmpty; return true; }
        }
    }
    wide_pop(w, h.t, tid);
    return false;
}

__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A lane whose stack cannot take a node's slots (`over`: wide_node_step returned false; trees deeper than the
// stack: hundreds of thousands of triangles) hands the subtree of that node to the unculled walk: the node
// stands for a binary inner node, whose subtree is a range of the threaded array, and the reference's walk of
// that range returns the subtree's own closest hit by the reference's rule (smallest t, among equal t the last
// in visit order).  The lane keeps the better of that and its best so far by the same rule and goes on with its
// stack.  Returns false for a lane that has to start its ray over with the unculled walk of the whole scene
// (the node is a virtual one above the meshes' roots: no range).  `quick`: wave-uniform among the lanes in here
// (the callers pass false: the path is rare and one copy of the unculled walk's code is enough).
template <bool ANY>
__device__ __forceinline__ bool wide_overflow(const SceneView& S, V3 o, V3 d, V3 inv, bool quick, HitRec& h, WideState& w, uint32_t tid) {
    const uint32_t b = S.wide_bin[w.cur];
    if (b == kInvalid) { w.cur = kWideEmpty; return false; }
    const uint32_t end = S.nodes[b].a;   // the inner node's skip link
    HitRec c; c.t = __builtin_inff(); c.u = 0.0f; c.v = 0.0f; c.gtri = kInvalid;
    // (rays the wide walk takes have finite origins and reciprocals: the slab test's short form applies)
    if (quick) traverse_lane_walk<ANY, true>(S, o, d, inv, true, c, b, end);
    else traverse_lane_walk<ANY, false>(S, o, d, inv, true, c, b, end);
    if (c.gtri != kInvalid) {
        const uint32_t slot = S.slot_of[c.gtri];
        if (c.t < h.t || (c.t == h.t && slot > w.best_slot)) { h = c; w.best_slot = slot; }
        if (ANY) { w.cur = kWideEmpty; return true; }
    }
    wide_pop(w, h.t, tid);
    return true;
}

// Which rays the wide walk takes (see the header comment).
__device__ __forceinline__ bool wide_tame(const SceneView& S, V3 o, V3 d) {
    const int kNormal = (1 << 3) | (1 << 8);
    const float ob = S.wide_org_bound;
    const bool o_ok = fabsf(o.x) <= ob && fabsf(o.y) <= ob && fabsf(o.z) <= ob;   // false for NaN
    const bool d_ok = __builtin_amdgcn_classf(d.x, kNormal) && __builtin_amdgcn_classf(d.y, kNormal) && __builtin_amdgcn_classf(d.z, kNormal) &&
                      fabsf(d.x) < 0x1p+40f && fabsf(d.y) < 0x1p+40f && fabsf(d.z) < 0x1p+40f &&
                      fabsf(d.x) > 0x1p-40f && fabsf(d.y) > 0x1p-40f && fabsf(d.z) > 0x1p-40f;
    return o_ok && d_ok;
}

// A whole ray, for callers whose lanes all start together (the light-sample and shadow rays of the
// shade kernel).  Returns false for a lane whose stack would have overflowed (its h is then unspecified).
template <bool ANY, bool QUICK>
__device__ __forceinline__ bool traverse_wide_walk(const SceneView& S, V3 o, V3 d, V3 inv, HitRec& h) {
    const uint32_t tid = threadIdx.x;
    WideState w; w.cur = 0u; w.sp = 0u; w.best_slot = 0u;
    bool ok = true;
    const WideGuard g = wide_guard_of(S, d);
    const bool guard = __any(g.cell != kGuardNoCell);
    // The lanes of the wave that walk do so together: node steps until nobody is at a node or kStaticQuorum lanes
    // wait at a leaf (they would idle while the others walk on), then the leaves (nanosuit frame with four
    // bounces: +13 % against "every lane to its next leaf first"; 8, 12 and 16 lanes within 2 % of each other).
    constexpr uint32_t kStaticQuorum = 12;
    while (__any(w.cur != kWideEmpty)) {
        for (;;) {
            const bool at_node = wide_at_node(w);
            if (!__any(at_node)) break;
            if ((uint32_t)__popcll(__ballot(wide_at_leaf(w))) >= kStaticQuorum) break;
            const bool over = at_node && !wide_node_step(S, o, inv, h.t, w, tid, g, guard);
            if (__any(over)) {
                if (over) ok = wide_overflow<ANY>(S, o, d, inv, false, h, w, tid) && ok;
            }
        }
        if (wide_at_leaf(w)) wide_leaf_step<ANY, QUICK>(S, o, d, h, w, tid);
    }
    return ok;
}

template <bool ANY>
__device__ __forceinline__ void traverse_wide(const SceneView& S, V3 o, V3 d, HitRec& h) {
    bool d_small, d_normal;
    dir_class(d, S.dir_bound, d_small, d_normal);
    const bool tame = wide_tame(S, o, d);
    bool redo = !tame;
    if (tame) {
        // wave-uniform choice of the arithmetic among the lanes that walk (tame directions are normal numbers)
        const bool quick_det = __all(d_small), quick_inv = __all(d_normal);
        const V3 inv = quick_inv ? ray_inverse_quick(d) : ray_inverse(d);
        const bool ok = quick_det ? traverse_wide_walk<ANY, true>(S, o, d, inv, h) : traverse_wide_walk<ANY, false>(S, o, d, inv, h);
        redo = !ok;
    }
    if (redo) {
        h.t = __builtin_inff(); h.u = 0.0f; h.v = 0.0f; h.gtri = kInvalid;
        traverse_lane<ANY>(S, o, d, h);
    }
}

} // namespace orion_dev
