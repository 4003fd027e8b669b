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
};

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
};

// Closest-hit record kept in the queues: 16 bytes.  gtri == kInvalid is a miss.
struct HitRec { float t, u, v; uint32_t gtri; };

struct Light { float position[3]; float color[3]; float intensity; };

constexpr int kMaxLights = 16;

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
    Light lights[kMaxLights];
};

} // namespace orion_dev
