// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by or shipped with the product path.
//
// C-ABI harness around the *real* reference classes.  It is compiled against the
// reference headers and translation units where they lie under /root/reference
// (recipe: oracle/Makefile, target `ref`), with output only into oracle/_ref/.
// Nothing from the reference is copied: this file only #includes and calls it.
//
// What it exposes (all used to pin oracle/orion_oracle.cpp and to generate the
// golden fixtures under tests/golden/):
//   * TracedMesh construction + TracedMesh::intersect            (orion/mesh.hpp:62-102)
//   * the mesh loop of TracedModel::intersect, restated here       (orion/model.hpp:52-62)
//     because model.hpp needs assimp, which this image lacks
//   * a dump of the SBVH build tree and triangle permutation      (orion/avx/sbvh.cpp:115-153)
//   * normal / uv interpolation, surface area, light-point sampling (orion/mesh.hpp:124-184)
//   * Material::colorBRDF / Material::color / Texture::color      (orion/material.hpp:72-105, orion/texture.hpp:72-86)
//   * xoroshiro128 streams and libstdc++ uniform floats on top     (orion/random.hpp:38-134)
//   * parse_rtc                                                    (orion/rtc_parser.cpp:69-134)
//   * the camera basis of RayTracer::calculateCameraVectors, restated with the
//     reference's own vec3f math (orion/raytracer.cpp:212-238; raytracer.cpp itself
//     needs tqdm + assimp and is unbuildable here)
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include <omp.h>

// The build tree and the permutation are private members of SBVH; the harness
// needs to read (never write) them to pin the builder.
#define private public
#include <orion/avx/sbvh.hpp>
#undef private
#include <orion/mesh.hpp>
#include <orion/material.hpp>
#include <orion/random.hpp>
#include <orion/rtc_parser.hpp>

using namespace orion;

namespace {

struct RefMesh {
    std::unique_ptr<TracedMesh> mesh;
    // A second tree built with a non-default strategy (SBVH is not assignable,
    // so it lives beside the mesh's own MEDIAN tree).
    std::unique_ptr<SBVH> alt;
    const SBVH& tree() const { return alt ? *alt : mesh->mTree; }
    unsigned intersect(const vec3f& o, const vec3f& d, float& t, float& u, float& v) const {
        return alt ? alt->intersect(o, d, t, u, v) : mesh->intersect(o, d, t, u, v);
    }
};

static void flattenTree(const BuildBVHNode* n, std::vector<float>& boxes, std::vector<uint32_t>& meta) {
    // preorder: box lo xyz, hi xyz | isLeaf, begin, end
    vec3f lo = n->bv.lowerBound(), hi = n->bv.upperBound();
    boxes.push_back(lo.x()); boxes.push_back(lo.y()); boxes.push_back(lo.z());
    boxes.push_back(hi.x()); boxes.push_back(hi.y()); boxes.push_back(hi.z());
    meta.push_back(n->isLeaf ? 1u : 0u);
    meta.push_back(n->begin);
    meta.push_back(n->end);
    if (!n->isLeaf) {
        flattenTree(n->children[0].get(), boxes, meta);
        flattenTree(n->children[1].get(), boxes, meta);
    }
}

} // namespace

extern "C" {

// verts: n_verts x 8 floats (pos3, normal3, uv2), idx: n_idx u32 (triangles),
// solid: Ka3 Kd3 Ks3 Ke3 Ns d  (14 floats).  strategy: 0 MEDIAN, 1 MIDDLE, 2 SAH.
void* ref_mesh_create(const float* verts, uint32_t n_verts, const uint32_t* idx, uint32_t n_idx,
                      const float* solid, int strategy) {
    std::vector<Vertex> vv(n_verts);
    for (uint32_t i = 0; i < n_verts; i++) {
        const float* p = verts + 8 * i;
        vv[i].position = vec3f_compact(vec3f(p[0], p[1], p[2]));
        vv[i].normal = vec3f_compact(vec3f(p[3], p[4], p[5]));
        vv[i].texCoords = vec2f(p[6], p[7]);
    }
    std::vector<unsigned> ii(idx, idx + n_idx);
    SolidSurface s;
    s.color_ambient = vec3f(solid[0], solid[1], solid[2]);
    s.color_diffuse = vec3f(solid[3], solid[4], solid[5]);
    s.color_specular = vec3f(solid[6], solid[7], solid[8]);
    s.color_emissive = vec3f(solid[9], solid[10], solid[11]);
    s.shininess = solid[12];
    s.opacity = solid[13];
    Material m(s);
    auto* r = new RefMesh;
    r->mesh.reset(new TracedMesh(vv, ii, m));
    if (strategy != 0) {
        // TracedMesh always builds with the default strategy; build a second tree
        // with the requested one through SBVH's public constructor.
        SBVH::Strategy st(8, strategy == 1 ? SBVH::Strategy::MIDDLE : SBVH::Strategy::SAH, 12, 4);
        r->alt.reset(new SBVH(r->mesh->vertices, r->mesh->indices, st));
    }
    return r;
}

void ref_mesh_destroy(void* h) { delete static_cast<RefMesh*>(h); }

// Install an image file as one of the material's textures (type as Material::TextureType).
int ref_mesh_set_texture(void* h, int type, const char* path) {
    auto* r = static_cast<RefMesh*>(h);
    Texture t;
    if (t.loadFromFile(path) != 0) return -1;
    r->mesh->pMat->setTexture(static_cast<Material::TextureType>(type), t);
    return 0;
}

uint32_t ref_mesh_id(void* h) { return static_cast<RefMesh*>(h)->mesh->mID; }
float ref_mesh_area(void* h) { return static_cast<RefMesh*>(h)->mesh->surfaceArea(); }
uint32_t ref_mesh_triangle_count(void* h) { return static_cast<RefMesh*>(h)->mesh->triangleCount(); }

// Number of nodes of the build tree.
uint32_t ref_mesh_tree_size(void* h) {
    std::vector<float> b; std::vector<uint32_t> m;
    flattenTree(static_cast<RefMesh*>(h)->tree().mRoot.get(), b, m);
    return (uint32_t)(m.size() / 3);
}

// boxes: nodes x 6 floats, meta: nodes x 3 u32 (isLeaf, begin, end), perm: n_tri u32 (originalID per slot)
void ref_mesh_tree_dump(void* h, float* boxes, uint32_t* meta, uint32_t* perm) {
    auto* r = static_cast<RefMesh*>(h);
    std::vector<float> b; std::vector<uint32_t> m;
    flattenTree(r->tree().mRoot.get(), b, m);
    std::memcpy(boxes, b.data(), b.size() * sizeof(float));
    std::memcpy(meta, m.data(), m.size() * sizeof(uint32_t));
    const auto& tris = r->tree().mTriangles;
    for (size_t i = 0; i < tris.size(); i++) perm[i] = tris[i].originalID;
}

// Single-mesh closest hit, t in/out (pass +inf for a fresh query).
void ref_mesh_intersect(void* h, const float* org, const float* dir, uint32_t n,
                        float* t, float* u, float* v, uint32_t* tri) {
    auto* r = static_cast<RefMesh*>(h);
    for (uint32_t i = 0; i < n; i++) {
        vec3f o(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        vec3f d(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        tri[i] = r->intersect(o, d, t[i], u[i], v[i]);
    }
}

// The loop of TracedModel::intersect (orion/model.hpp:52-62) over caller-ordered meshes,
// starting from t = +inf.  out: n x {mesh index, triangle id} u32, n x {t,u,v} float.
// threads <= 1 runs serially; otherwise OpenMP over rays (used for CPU-baseline timing).
void ref_model_intersect(void** meshes, uint32_t n_meshes, const float* org, const float* dir,
                         uint32_t n, uint32_t* ids, float* tuv, int threads) {
    if (threads < 1) threads = 1;
    #pragma omp parallel for num_threads(threads) schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        vec3f o(org[3 * i], org[3 * i + 1], org[3 * i + 2]);
        vec3f d(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]);
        float t = F_INFINITY;
        float uv[2] = {0.0f, 0.0f};
        uint32_t mesh = ~0u, tri = ~0u;
        for (uint32_t m = 0; m < n_meshes; m++) {
            unsigned id = static_cast<RefMesh*>(meshes[m])->intersect(o, d, t, uv[0], uv[1]);
            if (id != INVALID_INTERSECT_ID) { mesh = m; tri = id; }
        }
        ids[2 * i] = mesh; ids[2 * i + 1] = tri;
        tuv[3 * i] = t; tuv[3 * i + 1] = uv[0]; tuv[3 * i + 2] = uv[1];
    }
}

// Interpolated attributes (orion/mesh.hpp:124-152) -> out: n x {nx,ny,nz,tu,tv}
void ref_mesh_attributes(void* h, const uint32_t* tri, const float* u, const float* v, uint32_t n, float* out) {
    auto* r = static_cast<RefMesh*>(h);
    for (uint32_t i = 0; i < n; i++) {
        vec3f nn = r->mesh->normal(tri[i], u[i], v[i]);
        vec2f tt = r->mesh->texture_uv(tri[i], u[i], v[i]);
        out[5 * i] = nn.x(); out[5 * i + 1] = nn.y(); out[5 * i + 2] = nn.z();
        out[5 * i + 3] = tt.x(); out[5 * i + 4] = tt.y();
    }
}

// normalize() as the reference does it (rsqrtps, hardware-approximate): in/out n x 3
void ref_normalize(float* xyz, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        vec3f a(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
        a = a.normalized();
        xyz[3 * i] = a.x(); xyz[3 * i + 1] = a.y(); xyz[3 * i + 2] = a.z();
    }
}

// TracedMesh::randomPointOnSurface (orion/mesh.hpp:179-184) driven by a xoroshiro128
// seeded with `seed`: out n x {px,py,pz,bias}
void ref_mesh_random_points(void* h, uint64_t seed, uint32_t n, float* out) {
    auto* r = static_cast<RefMesh*>(h);
    xoroshiro128 rng; rng.seed(seed);
    for (uint32_t i = 0; i < n; i++) {
        vec3f p; float bias;
        r->mesh->randomPointOnSurface(rng, p, bias);
        out[4 * i] = p.x(); out[4 * i + 1] = p.y(); out[4 * i + 2] = p.z(); out[4 * i + 3] = bias;
    }
}

// Material::colorBRDF (orion/material.hpp:95-105). in: n x {normal3, hit3, lightpos3, lightcol3, intensity, lightnormal3, uv2} = 18 floats
void ref_color_brdf(void* h, const float* in, uint32_t n, float* out3) {
    auto* r = static_cast<RefMesh*>(h);
    for (uint32_t i = 0; i < n; i++) {
        const float* p = in + 18 * i;
        Light l; l.position = vec3f(p[6], p[7], p[8]); l.color = vec3f(p[9], p[10], p[11]); l.intensity = p[12];
        vec3f c = r->mesh->material().colorBRDF(vec3f(p[0], p[1], p[2]), vec3f(p[3], p[4], p[5]), l,
                                                vec3f(p[13], p[14], p[15]), vec2f(p[16], p[17]));
        out3[3 * i] = c.x(); out3[3 * i + 1] = c.y(); out3[3 * i + 2] = c.z();
    }
}

// Material::color, 5-argument overload (orion/material.hpp:72-93).
// in: n x {raydir3, normal3, hit3, lightpos3, lightcol3, intensity, uv2} = 18 floats
void ref_color_phong(void* h, const float* in, uint32_t n, float* out3) {
    auto* r = static_cast<RefMesh*>(h);
    for (uint32_t i = 0; i < n; i++) {
        const float* p = in + 18 * i;
        Light l; l.position = vec3f(p[9], p[10], p[11]); l.color = vec3f(p[12], p[13], p[14]); l.intensity = p[15];
        vec3f c = r->mesh->material().color(vec3f(p[0], p[1], p[2]), vec3f(p[3], p[4], p[5]),
                                            vec3f(p[6], p[7], p[8]), l, vec2f(p[16], p[17]));
        out3[3 * i] = c.x(); out3[3 * i + 1] = c.y(); out3[3 * i + 2] = c.z();
    }
}

// Texture::color through the material's accessors: which = 0 diffuse, 1 reflectivity(specular), 2 emissivity
void ref_material_fetch(void* h, int which, const float* uv, uint32_t n, float* out3) {
    auto* r = static_cast<RefMesh*>(h);
    for (uint32_t i = 0; i < n; i++) {
        vec2f t(uv[2 * i], uv[2 * i + 1]);
        vec3f c = which == 0 ? r->mesh->material().diffuse(t)
                : which == 1 ? r->mesh->material().reflectivity(t)
                             : r->mesh->material().emissivity(t);
        out3[3 * i] = c.x(); out3[3 * i + 1] = c.y(); out3[3 * i + 2] = c.z();
    }
}

// reflect() (orion/math.hpp:321-323): in n x {I3, N3}, out n x 3
void ref_reflect(const float* in, uint32_t n, float* out3) {
    for (uint32_t i = 0; i < n; i++) {
        const float* p = in + 6 * i;
        vec3f c = reflect(vec3f(p[0], p[1], p[2]), vec3f(p[3], p[4], p[5]));
        out3[3 * i] = c.x(); out3[3 * i + 1] = c.y(); out3[3 * i + 2] = c.z();
    }
}

// xoroshiro128 raw stream after seed(sd) and `jumps` calls to jump().
void ref_xoroshiro_u64(uint64_t sd, uint32_t jumps, uint32_t n, uint64_t* out) {
    xoroshiro128 rng; rng.seed(sd);
    for (uint32_t j = 0; j < jumps; j++) rng.jump();
    for (uint32_t i = 0; i < n; i++) out[i] = rng();
}

// std::uniform_real_distribution<float>(0,1) on the same stream (libstdc++ generate_canonical).
void ref_xoroshiro_floats(uint64_t sd, uint32_t jumps, uint32_t n, float* out) {
    xoroshiro128 rng; rng.seed(sd);
    for (uint32_t j = 0; j < jumps; j++) rng.jump();
    std::uniform_real_distribution<float> d(0.0f, 1.0f);
    for (uint32_t i = 0; i < n; i++) out[i] = d(rng);
}

// parse_rtc -> ints: {recursion_level, xres, yres, n_lights}; floats: {aspect, eye3, lookat3, up3, yview} (11);
// lights: up to max_lights x {pos3, col3, intensity}; obj: path copied into buf.
int ref_parse_rtc(const char* path, int* ints, float* floats, float* lights, uint32_t max_lights, char* obj, uint32_t obj_cap) {
    rtc_data r = parse_rtc(path);
    ints[0] = r.recursion_level; ints[1] = r.xres; ints[2] = r.yres; ints[3] = (int)r.lights.size();
    floats[0] = r.aspect_ratio;
    floats[1] = r.view_point.x(); floats[2] = r.view_point.y(); floats[3] = r.view_point.z();
    floats[4] = r.look_at.x(); floats[5] = r.look_at.y(); floats[6] = r.look_at.z();
    floats[7] = r.vector_up.x(); floats[8] = r.vector_up.y(); floats[9] = r.vector_up.z();
    floats[10] = r.y_view;
    for (uint32_t i = 0; i < r.lights.size() && i < max_lights; i++) {
        const Light& l = r.lights[i];
        float* p = lights + 7 * i;
        p[0] = l.position.x(); p[1] = l.position.y(); p[2] = l.position.z();
        p[3] = l.color.x(); p[4] = l.color.y(); p[5] = l.color.z(); p[6] = l.intensity;
    }
    std::strncpy(obj, r.obj_file.c_str(), obj_cap - 1); obj[obj_cap - 1] = 0;
    return 0;
}

// Camera basis with the reference's math, following orion/raytracer.cpp:212-238.
// in: eye3, lookat3, up3, yview, aspect; out: front3, up3, right3
void ref_camera_vectors(const float* eye, const float* lookat, const float* up, float yview, float aspect, float* out9) {
    vec3f vUp(up[0], up[1], up[2]);
    vec3f vFront = vec3f(lookat[0], lookat[1], lookat[2]) - vec3f(eye[0], eye[1], eye[2]);
    vUp = orthogonalize(vFront, vUp);
    vUp.normalize();
    vFront.normalize();
    vec3f vRight = cross(vFront, vUp);
    vUp = vUp * yview * 0.5f;
    vRight = vRight * yview * aspect * 0.5f;
    out9[0] = vFront.x(); out9[1] = vFront.y(); out9[2] = vFront.z();
    out9[3] = vUp.x(); out9[4] = vUp.y(); out9[5] = vUp.z();
    out9[6] = vRight.x(); out9[7] = vRight.y(); out9[8] = vRight.z();
}

int ref_max_threads() { return omp_get_max_threads(); }

// The reference's own image decoder (its vendored stb_image, as orion/texture.hpp:34 calls it):
// 8-bit texels with the file's channel count.  out must hold width*height*4 bytes at most; returns
// 0 on success, -1 if the decoder refuses the file.
int ref_load_image(const char* path, int* width, int* height, int* channels, unsigned char* out, size_t out_cap) {
    unsigned char* raw = stbi_load(path, width, height, channels, 0);
    if (!raw) return -1;
    const size_t n = (size_t)*width * *height * *channels;
    int rc = 0;
    if (n <= out_cap) std::memcpy(out, raw, n); else rc = -1;
    stbi_image_free(raw);
    return rc;
}

} // extern "C"
