// Host side of the 4-wide tree (WideNode, device_types.h): collapses the caller's binary trees
// (BuildBVHNode, orion/avx/sbvh.hpp:44-66) into wide nodes, ties the meshes together under
// virtual nodes and quantises the slot boxes.  Plain C++: included by the scene upload and by the
// CPU model the tests run against the oracle.
//
// What the collapse keeps (the reference's visit rule, orion/avx/sbvh.cpp:36-83):
//   * every wide node stands for one binary inner node and carries that node's box bit for bit;
//   * a binary inner node is collapsed into its parent's wide node only if BOTH its children are
//     inner nodes, so a leaf is always a slot of the wide node of its own binary parent;
//   * the build refuses trees in which a child box is not contained in its parent's box or a
//     triangle pokes out of its leaf's parent (the monotonicity argument needs containment): such
//     scenes keep the unculled walk.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/orion_hip.h"
#include "device_types.h"
#include "wide_guard.h"

namespace orion_wide {

using orion_dev::Tri;
using orion_dev::WideNode;
using orion_dev::kWideEmpty;
using orion_dev::kWideLeaf;
using orion_dev::kWideMaxLeaf;
using orion_dev::kWideSlack;
using orion_dev::kGuardKappa;
using orion_dev::kGuardHarmless;
using orion_dev::GuardBand;

// The grid of a wide node along one axis: 254 cells from the lower to the upper bound of the node's own box
// (byte 255 lies one cell beyond the upper bound).  The walk never decodes a byte into a coordinate: it places
// the grid's planes between the two slab distances of the node's own box (wide_walk.h).
inline double grid_decode(uint32_t q, float lo, float hi) { return (double)lo + (double)q * (((double)hi - (double)lo) * (1.0 / 254.0)); }

struct WideTree {
    std::vector<WideNode> nodes;   // nodes[0] is the root; empty: the scene is not eligible
    std::vector<uint32_t> src_mesh, src_node;   // per wide node: the caller's mesh and node it stands for (0xFFFFFFFF: a virtual node)
    float pad = 0.0f;              // every decoded slot box, grown by this, contains the real box grown by three quarters of it
    float org_bound = 0.0f;        // rays with an origin component beyond this take the unculled walk
    uint32_t stack_need = 0;       // deepest the ordered walk's stack can get
    uint32_t n_virtual = 0;        // nodes[0 .. n_virtual) are the virtual nodes above the meshes' roots (no box of theirs exists in the reference)
    float guard_dlen = 0.0f;       // rays whose direction is at most this long need no guard (no triangle is large enough)
    std::vector<GuardBand> bands;  // per triangle slot: unit normal and the half-width of its band of ill-conditioned directions (wide_guard.h)
    std::string why_not;           // set when nodes is empty
};

namespace detail {

struct Box { float lo[3], hi[3]; };

inline Box box_empty() {
    Box b;
    for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    return b;
}
inline void box_grow(Box& b, const float* p) {
    for (int a = 0; a < 3; a++) { b.lo[a] = std::fmin(b.lo[a], p[a]); b.hi[a] = std::fmax(b.hi[a], p[a]); }
}
inline void box_merge(Box& b, const Box& o) { box_grow(b, o.lo); box_grow(b, o.hi); }
inline bool box_inside(const Box& in, const Box& out) {
    for (int a = 0; a < 3; a++) if (!(in.lo[a] >= out.lo[a] && in.hi[a] <= out.hi[a])) return false;
    return true;
}
inline float box_half_area(const Box& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx * dy + dy * dz + dz * dx;
}

// A node of the combined binary forest: the meshes' own nodes plus the virtual nodes above their roots.
struct BNode {
    Box box;               // inner: the caller's box; leaf: the box of its triangles; virtual: union of the children
    bool leaf = false, virt = false;
    uint32_t child[2] = {0, 0};
    uint32_t first = 0, count = 0;   // leaf: triangle slots
};

} // namespace detail

// meshes / slot_base: as handed to orion_hip_scene_create, slot_base[m] = first triangle slot of
// mesh m in `tris` (slots in tri_order).  The trees must already have been validated (child
// indices in range, every node reachable once).
inline WideTree build_wide_tree(const OrionMeshDesc* meshes, uint32_t n_meshes, const std::vector<uint32_t>& slot_base,
                                const std::vector<Tri>& tris) {
    using namespace detail;
    WideTree out;
    auto refuse = [&](const std::string& why) { out.nodes.clear(); out.src_mesh.clear(); out.src_node.clear(); out.why_not = why; return out; };
    if (tris.empty()) return refuse("no triangles");
    if (tris.size() >= (1u << 25)) return refuse("more than 2^25 triangles");   // a leaf reference keeps 25 bits for the slot

    // ---- the binary forest ---------------------------------------------------------------
    std::vector<BNode> bn;
    std::vector<uint32_t> bn_mesh, bn_node;   // where each node of the forest comes from (virtual nodes: 0xFFFFFFFF)
    std::vector<uint32_t> roots;
    float biggest = 0.0f;
    for (uint32_t mi = 0; mi < n_meshes; mi++) {
        const OrionMeshDesc& m = meshes[mi];
        if (m.n_nodes == 0 || m.n_indices == 0) continue;
        const uint32_t base = (uint32_t)bn.size();
        bn.resize(base + m.n_nodes);
        for (uint32_t i = 0; i < m.n_nodes; i++) { bn_mesh.push_back(mi); bn_node.push_back(i); }
        for (uint32_t i = 0; i < m.n_nodes; i++) {
            const OrionBvhNode& s = m.nodes[i];
            BNode& b = bn[base + i];
            if (s.is_leaf) {
                b.leaf = true; b.first = slot_base[mi] + s.begin; b.count = s.end - s.begin;
                if (b.count > kWideMaxLeaf) return refuse("a leaf holds more than 63 triangles");
                b.box = box_empty();
                for (uint32_t k = 0; k < b.count; k++) {
                    const Tri& t = tris[b.first + k];
                    float p1[3], p2[3];
                    for (int a = 0; a < 3; a++) { p1[a] = t.v0[a] + t.e1[a]; p2[a] = t.v0[a] + t.e2[a]; }
                    box_grow(b.box, t.v0); box_grow(b.box, p1); box_grow(b.box, p2);
                }
                if (b.count == 0) for (int a = 0; a < 3; a++) b.box.lo[a] = b.box.hi[a] = s.lo[a];   // nothing to hit; any box inside the parent
            } else {
                std::memcpy(b.box.lo, s.lo, 12); std::memcpy(b.box.hi, s.hi, 12);
                b.child[0] = base + s.child[0]; b.child[1] = base + s.child[1];
            }
            for (int a = 0; a < 3; a++) {
                if (!std::isfinite(b.box.lo[a]) || !std::isfinite(b.box.hi[a])) return refuse("non-finite box or vertex");
                biggest = std::fmax(biggest, std::fmax(std::fabs(b.box.lo[a]), std::fabs(b.box.hi[a])));
            }
        }
        roots.push_back(base);
    }
    if (roots.empty()) return refuse("no trees");
    if (!(biggest > 0.0f) || biggest > 0x1p+60f) return refuse("scene coordinates out of range");
    // The pad covers what the triangle test's own rounding can move a hit point away from its
    // triangle for all but rays within a fraction of a degree of a triangle's plane (see the walk).
    out.pad = biggest * 0x1p-12f;
    out.org_bound = biggest * 4.0f;

    // containment: a child's box inside its parent's (leaf children: the triangles' box, the edge vectors'
    // rounding included -- the pad is thousands of times that)
    for (const BNode& b : bn) {
        if (b.leaf) continue;
        for (int c = 0; c < 2; c++) {
            const BNode& ch = bn[b.child[c]];
            Box grown = b.box;
            if (ch.leaf) for (int a = 0; a < 3; a++) { grown.lo[a] -= out.pad * 0.25f; grown.hi[a] += out.pad * 0.25f; }
            if (ch.count == 0 && ch.leaf) continue;
            if (!box_inside(ch.box, grown)) return refuse("a child box is not inside its parent's box");
        }
    }

    // virtual nodes above the meshes' roots: median split of the roots' centres along the widest axis
    uint32_t top;
    {
        std::vector<uint32_t> order = roots;
        auto centre = [&](uint32_t n, int a) { return bn[n].box.lo[a] + bn[n].box.hi[a]; };
        // recursive build through an explicit stack; returns the index of the subtree's node
        struct Job { uint32_t begin, end; uint32_t* out; };
        std::vector<Job> jobs;
        top = 0;
        jobs.push_back({0, (uint32_t)order.size(), &top});
        // child slots are written through pointers into bn, so reserve first: one virtual node per split + one wrapper
        bn.reserve(bn.size() + 2 * order.size() + 2);
        while (!jobs.empty()) {
            Job j = jobs.back(); jobs.pop_back();
            if (j.end - j.begin == 1) { *j.out = order[j.begin]; continue; }
            Box all = box_empty();
            for (uint32_t i = j.begin; i < j.end; i++) box_merge(all, bn[order[i]].box);
            int axis = 0;
            for (int a = 1; a < 3; a++) if (all.hi[a] - all.lo[a] > all.hi[axis] - all.lo[axis]) axis = a;
            const uint32_t mid = (j.begin + j.end) / 2;
            std::nth_element(order.begin() + j.begin, order.begin() + mid, order.begin() + j.end,
                             [&](uint32_t x, uint32_t y) { return centre(x, axis) < centre(y, axis); });
            BNode v; v.virt = true; v.box = all;
            // grown by more than the walk's slack can reach for any ray it takes (origins within org_bound)
            const float grow = out.pad + 1.25f * kWideSlack * (out.org_bound + biggest);
            for (int a = 0; a < 3; a++) { v.box.lo[a] -= grow; v.box.hi[a] += grow; }
            bn.push_back(v);
            const uint32_t me = (uint32_t)bn.size() - 1;
            *j.out = me;
            jobs.push_back({j.begin, mid, &bn[me].child[0]});
            jobs.push_back({mid, j.end, &bn[me].child[1]});
        }
    }

    // ---- collapse ---------------------------------------------------------------------------
    // A wide node stands for binary node `b` (inner or virtual).  Its slots start as b's two children;
    // the slot with the largest box that is an inner / virtual node with two non-leaf children is
    // replaced by those children until four slots are filled.
    auto expandable = [&](uint32_t n) {
        const BNode& b = bn[n];
        return !b.leaf && !bn[b.child[0]].leaf && !bn[b.child[1]].leaf;
    };
    struct Work { uint32_t bnode, index; bool wrapper; };
    std::vector<Work> work;
    out.nodes.push_back(WideNode{});
    // A lone mesh whose root is a leaf (or any root that is a leaf) still needs a node to hang from.
    work.push_back({top, 0, bn[top].leaf});
    std::vector<uint32_t> need(1, 0);          // stack entries the walk can hold below each wide node
    std::vector<uint32_t> parent(1, 0xFFFFFFFFu);
    for (size_t w = 0; w < work.size(); w++) {
        const Work job = work[w];
        const BNode& b = bn[job.bnode];
        std::vector<uint32_t> slots;
        Box own = b.box;
        if (job.wrapper) {
            slots.push_back(job.bnode);
            const float grow = out.pad + 1.25f * kWideSlack * (out.org_bound + biggest);
            for (int a = 0; a < 3; a++) { own.lo[a] -= grow; own.hi[a] += grow; }
        }
        else { slots.push_back(b.child[0]); slots.push_back(b.child[1]); }
        while (slots.size() < 4) {
            int pick = -1;
            float best = -1.0f;
            for (size_t k = 0; k < slots.size(); k++)
                if (expandable(slots[k]) && box_half_area(bn[slots[k]].box) > best) { best = box_half_area(bn[slots[k]].box); pick = (int)k; }
            if (pick < 0) break;
            const uint32_t n = slots[pick];
            slots[pick] = bn[n].child[0];
            slots.insert(slots.begin() + pick + 1, bn[n].child[1]);
        }
        WideNode wn{};
        std::memcpy(wn.lo, own.lo, 12); std::memcpy(wn.hi, own.hi, 12);
        if (out.src_mesh.size() < out.nodes.size()) { out.src_mesh.resize(out.nodes.size(), 0xFFFFFFFFu); out.src_node.resize(out.nodes.size(), 0xFFFFFFFFu); }
        if (!job.wrapper && job.bnode < bn_mesh.size()) { out.src_mesh[job.index] = bn_mesh[job.bnode]; out.src_node[job.index] = bn_node[job.bnode]; }
        uint32_t refs[4] = {kWideEmpty, kWideEmpty, kWideEmpty, kWideEmpty};
        for (size_t k = 0; k < 4; k++) {
            if (k >= slots.size() || (bn[slots[k]].leaf && bn[slots[k]].count == 0)) {
                for (int a = 0; a < 3; a++) wn.qlo[a] |= 255u << (8 * k);   // empty: lower bound above upper bound
                continue;
            }
            const BNode& c = bn[slots[k]];
            if (c.leaf) {
                refs[k] = kWideLeaf | (c.count << 25) | c.first;
            } else {
                out.nodes.push_back(WideNode{});
                refs[k] = (uint32_t)out.nodes.size() - 1;
                if (refs[k] >= kWideEmpty) return refuse("too many nodes");
                work.push_back({slots[k], refs[k], false});
                need.push_back(0); parent.push_back(job.index);
            }
            for (int a = 0; a < 3; a++) {
                const float lo = own.lo[a], hi = own.hi[a];
                int ql = 0, qh = 0;
                if (hi > lo) {
                    const double cells = 254.0 / ((double)hi - (double)lo);
                    ql = (int)std::floor(((double)c.box.lo[a] - lo) * cells);
                    qh = (int)std::ceil(((double)c.box.hi[a] - lo) * cells);
                    ql = std::min(255, std::max(0, ql)); qh = std::min(255, std::max(0, qh));
                    while (ql > 0 && grid_decode((uint32_t)ql, lo, hi) > c.box.lo[a]) ql--;
                    while (qh < 255 && grid_decode((uint32_t)qh, lo, hi) < c.box.hi[a]) qh++;
                }
                // a leaf's triangles may poke out of the parent's box by the rounding of an edge vector (a quarter
                // pad at most was accepted above): the walk grows every decoded box by the pad
                if (grid_decode((uint32_t)ql, lo, hi) > (double)c.box.lo[a] + 0.25 * out.pad || grid_decode((uint32_t)qh, lo, hi) < (double)c.box.hi[a] - 0.25 * out.pad)
                    return refuse("a slot box does not fit its node's grid");
                wn.qlo[a] |= (uint32_t)ql << (8 * k);
                wn.qhi[a] |= (uint32_t)qh << (8 * k);
            }
        }
        wn.ref0 = refs[0]; wn.ref1 = refs[1]; wn.ref2 = refs[2]; wn.ref3 = refs[3];
        out.nodes[job.index] = wn;
        need[job.index] = (uint32_t)slots.size() - 1;
    }
    out.src_mesh.resize(out.nodes.size(), 0xFFFFFFFFu); out.src_node.resize(out.nodes.size(), 0xFFFFFFFFu);
    {   // The guard's per-triangle data (wide_walk.h, wide_guard.h).  A triangle's test is ill conditioned for a direction d
        // when |d . (e1 x e2)| < kGuardKappa |d| |e1| |e2|, that is |d^ . n^| < kGuardKappa / sin(apex): a band of half-width
        // beta = asin(kGuardKappa / sin(apex)) around the great circle perpendicular to the unit normal n^ (apex: the angle
        // between the edge vectors).  A triangle without a normal (or a needle: sin(apex) <= kGuardKappa) is ill
        // conditioned for every direction.  And the largest |e1| |e2| of the scene: rays no longer than kGuardHarmless
        // over it cannot be accepted by an ill-conditioned test at all.
        double m_max = 0.0;
        out.bands.resize(tris.size());
        for (size_t i = 0; i < tris.size(); i++) {
            const Tri& T = tris[i];
            const double e1[3] = {T.e1[0], T.e1[1], T.e1[2]}, e2[3] = {T.e2[0], T.e2[1], T.e2[2]};
            const double c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
            const double m = std::sqrt((e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]) * (e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]));
            const double lc = std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
            m_max = std::fmax(m_max, m);
            GuardBand b = {{0.0f, 0.0f, 0.0f}, 2.0f};
            if (lc > 0.0 && m > 0.0 && std::isfinite(lc) && std::isfinite(m) && lc / m > 1.001 * (double)kGuardKappa) {
                for (int a = 0; a < 3; a++) b.n[a] = (float)(c[a] / lc);
                b.sin_beta = (float)(1.001 * (double)kGuardKappa * m / lc + 1e-7);    // (+ the rounding of the normal's components)
            }
            out.bands[i] = b;
        }
        out.guard_dlen = (m_max > 0.0 && std::isfinite(m_max)) ? (float)((double)kGuardHarmless / m_max * 0.999) : 0.0f;
    }
    // deepest stack: entries left behind on the way down (slots - 1 per node), children are created after their parents
    std::vector<uint32_t> total(out.nodes.size(), 0);
    for (size_t i = 0; i < out.nodes.size(); i++) {
        total[i] = need[i] + (parent[i] == 0xFFFFFFFFu ? 0 : total[parent[i]]);
        out.stack_need = std::max(out.stack_need, total[i]);
    }
    // Renumber the nodes: first the virtual nodes above the meshes' roots, then the others, each group depth first (a node,
    // then its slots' subtrees in slot order).  A mesh's subtree is then a contiguous range of the array, children still
    // stand behind their parents, the nodes a ray visits at the bottom of the tree lie close together -- which is what
    // the guard's table is laid out for (wide_guard.h: guard_table_offset) -- and "is this node virtual" is one comparison
    // of its index (the walk enters such a node, whose box the reference does not have, whenever one of its slots is
    // dangerous: a mesh whose tree is one leaf hangs from it, and the reference tests that mesh's triangles whatever the ray).
    {
        const size_t n = out.nodes.size();
        std::vector<uint32_t> new_of(n, 0xFFFFFFFFu), order;
        order.reserve(n);
        for (int pass = 0; pass < 2; pass++) {
            std::vector<uint32_t> todo(1, 0u);
            while (!todo.empty()) {
                const uint32_t i = todo.back(); todo.pop_back();
                const bool virt = out.src_mesh[i] == 0xFFFFFFFFu;
                if (virt == (pass == 0)) { new_of[i] = (uint32_t)order.size(); order.push_back(i); }
                const WideNode& w = out.nodes[i];
                const uint32_t refs[4] = {w.ref3, w.ref2, w.ref1, w.ref0};     // (pushed in reverse: slot 0's subtree comes first)
                for (uint32_t r : refs) if (r != kWideEmpty && !(r & kWideLeaf)) todo.push_back(r);
            }
            if (pass == 0) out.n_virtual = (uint32_t)order.size();
        }
        if (order.size() != n) return refuse("the wide tree does not reach all of its nodes");
        std::vector<WideNode> nodes(n);
        std::vector<uint32_t> src_mesh(n), src_node(n);
        for (size_t k = 0; k < n; k++) {
            WideNode w = out.nodes[order[k]];
            uint32_t* refs[4] = {&w.ref0, &w.ref1, &w.ref2, &w.ref3};
            for (uint32_t* r : refs) if (*r != kWideEmpty && !(*r & kWideLeaf)) *r = new_of[*r];
            nodes[k] = w; src_mesh[k] = out.src_mesh[order[k]]; src_node[k] = out.src_node[order[k]];
        }
        out.nodes.swap(nodes); out.src_mesh.swap(src_mesh); out.src_node.swap(src_node);
    }
    return out;
}

} // namespace orion_wide
