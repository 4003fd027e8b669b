/*
 * orion_hip.h -- C ABI of the MI355X (gfx950) ray-casting / path-tracing library.
 *
 * The reference renderer has no plugin or FFI seam: its hot path is three C++
 * call sites inside one binary.  This header is the seam a maintainer binds
 * instead of those call sites; every entry point names what it replaces.
 *
 *   orion_hip_scene_create   <- TracedModel / TracedMesh construction output
 *                               (orion/model.hpp:96-98, orion/mesh.hpp:27-34, 62-71)
 *   orion_hip_closest_hit    <- TracedModel::intersect            (orion/model.hpp:52-62)
 *   orion_hip_any_hit        <- the point-light shadow query      (orion/raytracer.cpp:198-199)
 *   orion_hip_render         <- the pixel/sample loop + trace()   (orion/raytracer.cpp:66-88, 105-210)
 *   orion_hip_camera_vectors <- RayTracer::calculateCameraVectors (orion/raytracer.cpp:212-238)
 *
 * Conventions
 *   - Plain C: pointers and sizes only.  Every function returns 0 on success
 *     and a non-zero code on failure; orion_hip_last_error() then describes it.
 *     Nothing throws across the boundary.
 *   - Array arguments named *_any may point to host memory or to device memory
 *     (hipMalloc / a framework's allocation on the scene's device).  The library
 *     inspects the pointer; host arrays are staged through internal buffers.
 *   - Calls on one scene handle must be serialised by the caller.  Each handle
 *     owns one HIP stream; every call returns after its work on that stream has
 *     completed.
 *   - The scene is immutable after creation (as it is in the reference while
 *     rendering); the caller keeps ownership of every host array it passes in.
 */
#ifndef ORION_HIP_H
#define ORION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OrionScene OrionScene;

#define ORION_INVALID_ID 0xFFFFFFFFu

/* One node of the host-built binary BVH (BuildBVHNode, orion/avx/sbvh.hpp:44-66).
 * nodes[0] is the root; children are indices into the same array. */
typedef struct OrionBvhNode {
    float lo[3];
    float hi[3];
    uint32_t is_leaf;
    uint32_t begin, end;   /* leaf: slots [begin,end) of tri_order */
    uint32_t child[2];     /* inner: first and second child */
} OrionBvhNode;

/* Texel formats of OrionTextureDesc.texels */
#define ORION_TEX_RGB_F32 0u   /* 3 floats per texel (solid colours are 1x1 of these) */
#define ORION_TEX_RGB_U8  1u   /* 3 bytes per texel, value = byte/255.0f (orion/texture.hpp:53-58) */
#define ORION_TEX_RGBA_U8 2u   /* 4 bytes per texel, alpha ignored (orion/texture.hpp:59-63) */
#define ORION_TEX_GRAY_U8 3u   /* 1 byte per texel replicated to rgb (orion/texture.hpp:50-52) */

/* A texture as Texture holds it (orion/texture.hpp:99-102): row-major, row 0 first. */
typedef struct OrionTextureDesc {
    uint32_t width, height;
    uint32_t format;
    const void* texels;    /* host memory */
} OrionTextureDesc;

/* Material::tex[] + SolidSurface (orion/material.hpp:12-18, 26-33, 146-147). */
typedef struct OrionMaterialDesc {
    uint32_t tex_ambient, tex_diffuse, tex_specular, tex_emissive;  /* indices into textures[] */
    float shininess;          /* SolidSurface::shininess, the Phong exponent */
    float emissive_color[3];  /* SolidSurface::color_emissive: non-zero marks an emissive mesh
                                 (Material::isEmissive, orion/material.hpp:126-128) */
} OrionMaterialDesc;

/* One TracedMesh (orion/mesh.hpp:27-34).  Mesh order is the order of
 * TracedModel::meshes: later meshes win exact ties (orion/model.hpp:54-60). */
typedef struct OrionMeshDesc {
    const float* vertices;     /* n_vertices x 8 floats: position3, normal3, texcoord2 (orion/vertex.hpp:8-21) */
    uint32_t n_vertices;
    const uint32_t* indices;   /* 3 per triangle (orion/mesh.hpp:28) */
    uint32_t n_indices;
    const OrionBvhNode* nodes; /* SBVH build tree (orion/avx/sbvh.cpp:115-153) */
    uint32_t n_nodes;
    const uint32_t* tri_order; /* n_indices/3 entries: SBVHTriangle::originalID per leaf slot (orion/avx/sbvh.hpp:40) */
    uint32_t material;         /* index into materials[] */
} OrionMeshDesc;

/* MeshIntersection + t (orion/model.hpp:16-34).  Miss: mesh = tri = ORION_INVALID_ID, t = +inf. */
typedef struct OrionHit {
    uint32_t mesh;   /* index into the meshes passed at creation */
    uint32_t tri;    /* triangle index inside that mesh (indices/3) */
    float t, u, v;
} OrionHit;

/* Light (orion/light.hpp:8-13) */
typedef struct OrionLight {
    float position[3];
    float color[3];
    float intensity;
} OrionLight;

/* Output of calculateCameraVectors plus the eye point (orion/raytracer.cpp:40-41, 80). */
typedef struct OrionCamera {
    float eye[3];
    float front[3];
    float right[3];
    float up[3];
} OrionCamera;

typedef struct OrionRenderParams {
    uint32_t width, height;        /* full image size (rtc xres, yres) */
    uint32_t spp_begin, spp_end;   /* absolute sample indices [begin,end) rendered by this call */
    uint32_t light_samples;        /* -l of the reference CLI (orion/launcher.cpp:28-29) */
    uint32_t max_depth;            /* rtc recursion level (orion/raytracer.cpp:67) */
    uint64_t seed;                 /* key of the counter-based generator */
    uint32_t n_lights;             /* 0 selects the BRDF path tracer, >0 the point-light tracer (orion/raytracer.cpp:131) */
    const OrionLight* lights;      /* host memory */
    uint32_t tile_x0, tile_y0, tile_x1, tile_y1;  /* half-open pixel rectangle; 0,0,0,0 = whole image */
} OrionRenderParams;

/* Per-kernel timing collected with HIP events on the scene's stream while
 * profiling is enabled. */
#define ORION_KERNEL_CLOSEST_HIT 0   /* closest-hit traversal (extend rays and orion_hip_closest_hit) */
#define ORION_KERNEL_SHADOW      1   /* light-sample / shadow rays: traced inside the shade launch, so only
                                        `items` (rays traced) is counted here; their time is part of SHADE */
#define ORION_KERNEL_SHADE       2   /* surface evaluation, roulette, next-ray generation */
#define ORION_KERNEL_GENERATE    3   /* primary-ray generation (part of the first closest-hit launch: no launches of its own) */
#define ORION_KERNEL_RESOLVE     4   /* per-sample accumulation into the image */
#define ORION_KERNEL_ANY_HIT     5   /* any-hit traversal */
#define ORION_KERNEL_TAIL        6   /* the last bounces of a batch, one path per wave (items: paths handed over) */
#define ORION_KERNEL_COUNT       7

typedef struct OrionKernelStats {
    uint64_t launches;
    uint64_t items;      /* rays (or path vertices / pixels) processed over all launches */
    double   ms;         /* summed launch durations */
} OrionKernelStats;

/* Flatten and upload a model.  device < 0 selects the current HIP device. */
int orion_hip_scene_create(const OrionMeshDesc* meshes, uint32_t n_meshes,
                           const OrionMaterialDesc* materials, uint32_t n_materials,
                           const OrionTextureDesc* textures, uint32_t n_textures,
                           int device, OrionScene** out);

int orion_hip_scene_destroy(OrionScene* scene);

/* n rays; org_any / dir_any: n x 3 floats; out_any: n OrionHit.
 * Same result as TracedModel::intersect(origin, dir, t = +inf). */
int orion_hip_closest_hit(OrionScene* scene, const float* org_any, const float* dir_any,
                          uint32_t n, OrionHit* out_any);

/* occluded_any[i] = 1 iff ray i hits any triangle at t >= 0 (unbounded ray), else 0. */
int orion_hip_any_hit(OrionScene* scene, const float* org_any, const float* dir_any,
                      uint32_t n, uint8_t* occluded_any);

/* orion_hip_closest_hit for callers that cannot allocate (tight serving loops, harnesses that forbid
 * hipMalloc after start-up).  It has no counterpart in the reference.  Same results as
 * orion_hip_closest_hit.  org, dir and out must be DEVICE arrays; n is at most 16 777 216 (2^24) rays per
 * call, larger batches are refused.  On large scenes the kernels' work words and chunk list (what the
 * other entry point keeps in a buffer it allocates) live in a 1 MiB array inside the library's code
 * object, one copy per GPU, held until the process ends.  The call makes no hipMalloc / hipFree and
 * records no profiling events.  It is NOT something to record into a HIP graph: it launches on the scene's
 * own stream and returns once the results are in place.  Every call on one GPU shares that GPU's array:
 * a call that finds it in use by another call (another scene, another thread) fails with code 4 and
 * touches nothing -- serialise calls per GPU. */
int orion_hip_closest_hit_noalloc(OrionScene* scene, const float* org, const float* dir,
                                  uint32_t n, OrionHit* out);

/* Adds, for every pixel of the tile, the radiance of samples [spp_begin, spp_end)
 * to accum_rgba_any (width*height*4 floats: r, g, b sums and the sample count in
 * the fourth channel), in ascending sample order.  No 1/spp scaling: the caller
 * combines shards and scales once (orion/raytracer.cpp:84-86). */
int orion_hip_render(OrionScene* scene, const OrionCamera* camera,
                     const OrionRenderParams* params, float* accum_rgba_any);

/* Camera basis from the .rtc view parameters. */
int orion_hip_camera_vectors(const float eye[3], const float look_at[3], const float up[3],
                             float y_view, float aspect_ratio, OrionCamera* out);

/* Profiling: enable != 0 starts collecting per-kernel HIP-event timings (and
 * clears earlier ones); stats must have room for ORION_KERNEL_COUNT entries. */
int orion_hip_profile_enable(OrionScene* scene, int enable);
int orion_hip_profile_read(OrionScene* scene, OrionKernelStats* stats);

/* Tuning knobs, mainly for A/B measurements.  Known names:
 *   "traversal"   0 = choose per scene and per bounce (default: wave-synchronous for scenes of at
 *                 most 384 nodes and for the primary rays of any scene, the ordered distance-culled
 *                 walk of the 4-wide tree for everything else on larger scenes; ray queries on large
 *                 scenes decide per 64 rays: rays leaving one point in nearly one direction walk together),
 *                 1 = unculled per-lane traversal (the reference's visit order, node for node),
 *                 2 = wave-synchronous traversal (scene data through scalar loads),
 *                 3 = one ray per wave at a time (scenes of at most 64 inner nodes +
 *                 triangles; slow on batches, it exists to test the walk the tail kernel uses),
 *                 4 = ordered, distance-culled traversal of the 4-wide tree (scenes above 64 nodes whose
 *                 trees keep every child box inside its parent's); what 0 picks for scenes above 384 nodes.
 *                 Returns what 1 returns, bit for bit, on every ray (see "wide_guard")
 *   "wide_guard"  1 (default): the 4-wide walk visits a slot whatever its box says when the slot may hold a triangle
 *                 whose test is ill conditioned for the ray (a ray within about 0.03 degrees of the triangle's plane,
 *                 where the reference's own t, u, v are rounding noise that can land anywhere); with it every skip of
 *                 the walk is proven and the walk returns the reference's bits.  0: the walk trusts its boxes (1.3 to
 *                 2.6 times faster on scenes of large triangles; differs from the reference on a few rays in 10^9 aimed
 *                 along triangle planes -- 349 of 2^30 on a 19 058-triangle model, none on any natural batch tested)
 *   "tail_paths"  a render batch hands its remaining paths to the one-path-per-wave
 *                 tail kernel once at most this many are left (0 = never; small scenes only)
 *   "batch_paths" upper bound on paths in flight per render batch (default 256 Mi: up to 36 GiB of queues
 *                 per stream; a render call takes fewer where the device's free memory asks for it)
 *   "overlap_below" a render call of several batches starts the next batch on a second
 *                 stream (with queues of its own) once the current batch has fewer rays
 *                 than this left (default 128 Mi; 0 = one batch at a time, half the memory)
 *   "queue_parts" 16 (default), 8, 4, 2 or 1: a render batch's ray and vertex queues come in this many partitions, each
 *                 with a counter of its own (appends through one counter are served 11 ns apart, whatever the
 *                 kernels do); the image does not depend on it.  Scenes whose bounce rays are drawn from the queue
 *                 by resident waves ("draw_rays") and frames without bounces keep one partition
 *   "draw_rays"   1 (default): on scenes that walk the 4-wide tree, the closest hits of bounce rays are found
 *                 by waves that draw rays from the queue as their lanes finish; 0: four rays per thread
 *   "wave_ahead"  1: the wave-synchronous walk of a large scene's camera rays asks for the next node and the next
 *                 triangle of a leaf while it tests the present one; 0 (default): it fetches each when it gets there
 *   "queue_budget_mib" memory a render call's queues may take, in MiB (0, the default: three quarters of
 *                 the device's free memory); the call halves its samples per batch until they fit
 * Results never depend on these, with the one exception stated under "wide_guard" = 0. */
int orion_hip_scene_set_option(OrionScene* scene, const char* name, int64_t value);

/* Message for the most recent failure on the calling thread ("" if none). */
const char* orion_hip_last_error(void);

#ifdef __cplusplus
}
#endif

#endif /* ORION_HIP_H */
