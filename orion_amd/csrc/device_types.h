// Device-side scene layout (HBM) shared by the host flattening code and the kernels.
//
// The reference walks one pointer-linked binary tree per mesh, recursively and
// in a fixed order: first child, then second child, leaves front to back, meshes
// in load order (orion/avx/sbvh.cpp:36-83, orion/model.hpp:52-62).  Because that
// order never depends on the ray, the whole scene flattens into ONE array of
// nodes in depth-first preorder with a "skip" link per inner node: a hit box
// continues at node+1, a missed box jumps to `skip`.  Traversal needs no stack
// and visits nodes and triangles in exactly the reference's order, which is what
// makes exact ties resolve the same way (the last accepted triangle wins).
#pragma once

#include <stdint.h>

namespace orion_dev {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kLeafFlag = 0x80000000u;

// 32 bytes, two 16-byte loads.
struct Node {
    float lo[3];
    uint32_t a;      // inner: skip link (index of the next node when the box is missed); leaf: first triangle slot
    float hi[3];
    uint32_t b;      // inner: 0; leaf: kLeafFlag | triangle count
};
static_assert(sizeof(Node) == 32, "Node layout");

// 48 bytes, three 16-byte loads.  Slots are stored in visit order.
struct Tri {
    float v0[3];
    uint32_t gtri;   // scene-wide triangle id = mesh.tri_base + original triangle index
    float e1[3];
    float light_bias;      // (triangle area / mesh area) * triangle count: the sampling bias of
                           // TracedMesh::randomPointOnSurface (orion/mesh.hpp:182), a per-triangle constant
    float e2[3];
    float light_intensity; // light_bias * mesh area (orion/raytracer.cpp:153)
};
static_assert(sizeof(Tri) == 48, "Tri layout");

// 64 bytes per triangle, indexed by gtri: the three vertex normals and texture
// coordinates pre-gathered (orion/mesh.hpp:124-152 reads them through indices).
struct TriAttr {
    float n0[3]; float uv0x;
    float n1[3]; float uv0y;
    float n2[3]; float uv1x;
    float uv1y, uv2x, uv2y;
    uint32_t mesh;
};
static_assert(sizeof(TriAttr) == 64, "TriAttr layout");

// One lane's share of the one-ray-per-wave walk (scenes of at most 64 inner nodes + triangles):
// item i, in visit order, is either an inner node's box or a triangle, with the set of inner
// nodes above it.  64 bytes.
struct WaveItem {
    float a[3];          // box: lo        triangle: v0
    uint32_t gtri;       //                triangle: scene-wide id
    float b[3];          // box: hi        triangle: e1
    uint32_t kind;       // 0 unused, 1 box, 2 triangle
    float c[3];          //                triangle: e2
    uint32_t pad;
    uint32_t anc_lo, anc_hi;   // bit i set: item i is an inner node whose subtree holds this item
    uint32_t pad2[2];
};
static_assert(sizeof(WaveItem) == 64, "WaveItem layout");

// One node of the 4-wide tree the ordered, distance-culled walk of large scenes uses (64 bytes,
// four 16-byte loads).  It is a collapse of the reference's binary trees that keeps the reference's
// visit rule: a triangle counts iff every binary inner node above it passes AABB::intersect
// (orion/avx/sbvh.cpp:36-83, orion/AABB.hpp:79-99).
//   * `lo`/`hi` is the exact box of the binary inner node this wide node stands for; the walk
//     tests it with the reference's arithmetic when it enters the node.  Boxes of the binary nodes
//     collapsed between a wide node and its parent contain it, and the slab test is monotone in
//     the box for rays without NaN products, so they need no test of their own.
//   * A leaf is only ever a slot of the wide node that stands for its binary parent (an inner node
//     with a leaf child is never collapsed away): entering that wide node is the leaf's condition.
//   * The four slots carry their boxes as bytes on a grid over the node's own box (254 cells from `lo` to
//     `hi` per axis); a decoded box, grown by the scene's pad, contains the slot's real box grown by
//     three quarters of the pad.  These boxes only ever REJECT: a slot is skipped when the ray misses
//     its decoded box (grown by the pad and the walk's slack) or meets it beyond the best hit so far.
//   * ref: kWideEmpty, the index of a wide node, or kWideLeaf | count << 25 | first triangle slot.
// Virtual nodes tie the meshes' roots together; their box is the union of what hangs below, grown
// by more than any slack the walk applies, so that missing it means missing everything below.
struct WideNode {
    float lo[3]; uint32_t ref0;
    float hi[3]; uint32_t ref1;
    uint32_t qlo[3];        // per axis: the four slots' lower bounds, one byte each (slot 0 in bits 0-7)
    uint32_t qhi[3];
    uint32_t ref2, ref3;
};
static_assert(sizeof(WideNode) == 64, "WideNode layout");
constexpr uint32_t kWideEmpty = 0x7FFFFFFFu;   // also "nothing left to visit" in the walk
constexpr uint32_t kWideLeaf = 0x80000000u;
constexpr uint32_t kWideMaxLeaf = 63;
// The walk grows every decoded box by the pad and by this fraction of (the ray origin's distance to the
// farthest corner of the node's box + two pads) (wide_walk.h explains what the slack is for).
constexpr float kWideSlack = 0x1p-7f;
constexpr float kWideInvCells = 1.0f / 254.0f;   // one cell of the byte grid, as a fraction of the node's box
// The guard's two constants (derived in wide_walk.h).  A triangle's test is well conditioned for a ray of direction d
// when |d . (e1 x e2)| >= kGuardKappa * |d| |e1| |e2|: the hit point it computes then lies within 0.9 of the walk's slack
// of the triangle.  And a triangle with |d| |e1| |e2| <= kGuardHarmless cannot be accepted in any other case: its
// determinant then falls below the 1e-6 the reference rejects (orion/geometry.hpp:103).
constexpr float kGuardKappa = 0x1p-11f;
constexpr float kGuardHarmless = 2.0e-3f;

struct Texture {
    uint32_t offset;   // element offset into texels_f32 (floats) or texels_u8 (bytes)
    uint32_t width, height;
    uint32_t format;   // ORION_TEX_*
};

struct Mesh {
    uint32_t tri_base;     // gtri of this mesh's triangle 0
    uint32_t n_tri;
    uint32_t tex[4];       // ambient, diffuse, specular, emissive -> textures[]
    float shininess;
    float area;            // sum of |dot(e1,e2)|/2 over triangles (orion/mesh.hpp:197-208, orion/geometry.hpp:155-157)
    // Emissive meshes whose vertex normals are all one vector: its unit form, and flat = 1.  A light sample's term
    // carries the factor max(dot(n, l) * dot(light normal, -l), 0) (orion/material.hpp:95-105); with the light's
    // normal known before the sample's ray is traced, the samples for which that factor is surely zero are not traced
    // (shade_core).  flat stays 0 when a sum of zeros could still be NaN: a light intensity or a colour texel that is
    // not finite.
    float flat_n[3];
    uint32_t flat;
    // What the shade kernel would otherwise reach through chains of dependent loads (mesh -> texture table -> texel;
    // hit -> attributes; triangle index -> slot -> triangle), each link a round trip a wave waits for:
    // the colour of a 1x1 diffuse / emissive texture (every solid colour; Texture::color maps any uv to its one texel),
    // the light's normal as its vertices hold it (one_normal: all of them hold the same), and where the light's
    // triangles stand once more in their original order behind the triangle array (light sampling picks by that order).
    float kd[3]; uint32_t kd_solid;
    float ke[3]; uint32_t ke_solid;
    float raw_n[3]; uint32_t one_normal;
    uint32_t light_tris;   // index into SceneView::tris
    uint32_t pad_[3];
};
static_assert(sizeof(Mesh) == 112, "Mesh layout");

// Everything a kernel needs, passed by value.
struct SceneView {
    const Node* nodes;
    const Tri* tris;
    const TriAttr* attrs;
    const uint32_t* slot_of;      // gtri -> triangle slot (light sampling picks triangles by original index)
    const Mesh* meshes;
    const Texture* textures;
    const float* texels_f32;
    const uint8_t* texels_u8;
    const uint32_t* emissive;     // mesh indices of emissive meshes, in mesh order
    uint32_t n_nodes;
    uint32_t n_tris;
    uint32_t n_meshes;
    uint32_t n_emissive;
    const WaveItem* items;        // one-ray-per-wave table, n_items entries (0: the scene is too large for it)
    uint32_t n_items;
    // Rays whose direction components are all below this in magnitude cannot produce a triangle
    // determinant of 2^126 or more in this scene (0 if the scene has non-finite edges).
    float dir_bound;
    // The 4-wide tree of the ordered, distance-culled walk (n_wide == 0: the scene has none).
    const WideNode* wide;
    const uint32_t* wide_bin;   // per wide node: index in `nodes` of the binary inner node it stands for (kInvalid: a virtual node)
    uint32_t n_wide;
    uint32_t wide_n_virtual;    // wide[0 .. wide_n_virtual): the virtual nodes above the meshes' roots
    float wide_pad;          // every decoded slot box, grown by this, contains its real box (wide_bvh.h)
    float wide_grow0;        // wide_pad * (1 + 2 * kWideSlack): what a slot box is grown by before the distance term
    float wide_org_bound;    // rays with an origin component beyond this take the unculled walk
    // The guard of the walk's unproven skips (wide_walk.h): on unless the caller switched it off (0), and only
    // for rays longer than wide_guard_dlen = kGuardHarmless / (the largest |e1| |e2| of the scene's triangles).
    // wide_cells: four bits per wide node and cell of the wide_cells_grid^2 grid over the ray directions (wide_guard.h:
    // guard_table_offset): bit k = slot k may hold a triangle that is ill conditioned for a direction of the cell.
    uint32_t wide_guard;
    float wide_guard_dlen;
    const uint8_t* wide_cells;
    uint32_t wide_cells_grid;
};

// Closest-hit record kept in the queues: 16 bytes.  gtri == kInvalid is a miss.
struct HitRec { float t, u, v; uint32_t gtri; };

struct Light { float position[3]; float color[3]; float intensity; };

// Frame constants for the path-tracing kernels.
struct FrameParams {
    float eye[3], front[3], right[3], up[3];
    uint32_t width, height;
    uint32_t tile_x0, tile_y0, tile_w, tile_h;
    uint32_t spp_begin;        // absolute index of the batch's first sample
    uint32_t n_samples;        // samples in this batch
    uint32_t light_samples, max_depth;
    float inv_light_samples;   // 1.0f / light_samples (orion/raytracer.cpp:158)
    uint32_t seed_lo, seed_hi;
    uint32_t n_lights;
    const Light* lights;       // device memory, n_lights entries (any number: orion/raytracer.cpp:196 loops over a vector)
    // The batch's queues come in 2^part_log2 partitions of part_stride records each, every one with counters of its
    // own: block b of a bounce launch serves partition b mod 2^part_log2 (kPartWords below).
    uint32_t part_log2, part_stride;
};

// Queue partitions.  A block appends to a queue through one returning 64-bit atomic on the queue's counter, and
// atomics on ONE address are served one after the other, 11.4 ns each on this chip whatever else the kernels do
// (tools/micro/atomic_rate.hip): 1.2 million blocks of the shade kernel per 128-spp Cornell pass cannot finish in
// less than 13 ms through one counter.  Sixteen counters 128 bytes apart take 0.5 ns per append.  A partition is a
// closed chain: the block that reads partition k appends to partition k of the next queue, so a partition's
// count never grows from bounce to bounce and its region of part_stride records, sized for the first bounce, holds.
constexpr uint32_t kPartsLog2 = 4, kParts = 1u << kPartsLog2;
constexpr uint32_t kPartWords = 16;                     // 64-bit words between two partitions' counters (128 bytes)
constexpr uint32_t kSlotWords = kParts * kPartWords;    // ... and between two bounces' counter slots

} // namespace orion_dev
