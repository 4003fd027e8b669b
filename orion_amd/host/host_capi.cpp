// C entry points over the host layer for scripting (ctypes): scene files in,
// flat arrays out.  No GPU code here.
#include <cstdlib>
#include <cstring>
#include <string>

#include "bvh_builder.hpp"
#include "obj_loader.hpp"
#include "rtc_parser.hpp"

using namespace orion_amd;

extern "C" {

struct OrionHostRtc {
    int32_t recursion_level, xres, yres, n_lights;
    float aspect_ratio;
    float view_point[3], look_at[3], vector_up[3];
    float y_view;
    char obj_file[512];
    char texture_file[512];
};

// lights: room for max_lights x 7 floats (position3, colour3, intensity)
int orion_host_parse_rtc(const char* path, OrionHostRtc* out, float* lights, uint32_t max_lights) {
    RtcData r = parse_rtc(path);
    out->recursion_level = r.recursion_level; out->xres = r.xres; out->yres = r.yres;
    out->n_lights = (int32_t)r.lights.size();
    out->aspect_ratio = r.aspect_ratio;
    std::memcpy(out->view_point, r.view_point, 12);
    std::memcpy(out->look_at, r.look_at, 12);
    std::memcpy(out->vector_up, r.vector_up, 12);
    out->y_view = r.y_view;
    std::strncpy(out->obj_file, r.obj_file.c_str(), sizeof out->obj_file - 1); out->obj_file[sizeof out->obj_file - 1] = 0;
    std::strncpy(out->texture_file, r.texture_file.c_str(), sizeof out->texture_file - 1); out->texture_file[sizeof out->texture_file - 1] = 0;
    for (uint32_t i = 0; i < r.lights.size() && i < max_lights; i++) std::memcpy(lights + 7 * i, &r.lights[i], 28);
    return 0;
}

int orion_host_write_rtc(const char* path, const OrionHostRtc* in, const float* lights) {
    RtcData r;
    r.recursion_level = in->recursion_level; r.xres = in->xres; r.yres = in->yres; r.aspect_ratio = in->aspect_ratio;
    std::memcpy(r.view_point, in->view_point, 12);
    std::memcpy(r.look_at, in->look_at, 12);
    std::memcpy(r.vector_up, in->vector_up, 12);
    r.y_view = in->y_view; r.obj_file = in->obj_file; r.texture_file = in->texture_file;
    for (int i = 0; i < in->n_lights; i++) { Light l; std::memcpy(&l, lights + 7 * i, 28); r.lights.push_back(l); }
    write_rtc(path, r);
    return 0;
}

void* orion_host_model_load(const char* obj_path, int strategy) {
    Model* m = new Model;
    if (!load_obj(obj_path, *m)) { /* empty model, like the reference after a failed import */ }
    build_bvh(*m, (SplitStrategy)strategy);
    return m;
}

// Model from caller arrays (one mesh), for synthetic scenes.
void* orion_host_model_from_arrays(const float* verts8, uint32_t n_verts, const uint32_t* idx, uint32_t n_idx, int strategy) {
    Model* m = new Model;
    m->materials.push_back(Material());
    Mesh mesh;
    mesh.vertices.resize(n_verts);
    std::memcpy(mesh.vertices.data(), verts8, (size_t)n_verts * 32);
    mesh.indices.assign(idx, idx + n_idx);
    m->meshes.push_back(std::move(mesh));
    build_bvh(*m, (SplitStrategy)strategy);
    return m;
}

void orion_host_model_free(void* h) { delete static_cast<Model*>(h); }
uint32_t orion_host_model_mesh_count(void* h) { return (uint32_t)static_cast<Model*>(h)->meshes.size(); }
uint32_t orion_host_model_material_count(void* h) { return (uint32_t)static_cast<Model*>(h)->materials.size(); }

// sizes: n_vertices, n_indices, n_nodes, material
void orion_host_mesh_sizes(void* h, uint32_t mesh, uint32_t* sizes4) {
    const Mesh& m = static_cast<Model*>(h)->meshes[mesh];
    sizes4[0] = (uint32_t)m.vertices.size(); sizes4[1] = (uint32_t)m.indices.size();
    sizes4[2] = (uint32_t)m.nodes.size(); sizes4[3] = m.material;
}

// Pointers into the model (valid until orion_host_model_free).
const float* orion_host_mesh_vertices(void* h, uint32_t mesh) { return &static_cast<Model*>(h)->meshes[mesh].vertices[0].position[0]; }
const uint32_t* orion_host_mesh_indices(void* h, uint32_t mesh) { return static_cast<Model*>(h)->meshes[mesh].indices.data(); }
const void* orion_host_mesh_nodes(void* h, uint32_t mesh) { return static_cast<Model*>(h)->meshes[mesh].nodes.data(); }
const uint32_t* orion_host_mesh_tri_order(void* h, uint32_t mesh) { return static_cast<Model*>(h)->meshes[mesh].tri_order.data(); }
const char* orion_host_mesh_name(void* h, uint32_t mesh) { return static_cast<Model*>(h)->meshes[mesh].name.c_str(); }

// floats14: Ka3 Kd3 Ks3 Ke3 Ns d
void orion_host_material_get(void* h, uint32_t mat, float* floats14) {
    const Material& m = static_cast<Model*>(h)->materials[mat];
    std::memcpy(floats14 + 0, m.ambient, 12); std::memcpy(floats14 + 3, m.diffuse, 12);
    std::memcpy(floats14 + 6, m.specular, 12); std::memcpy(floats14 + 9, m.emissive, 12);
    floats14[12] = m.shininess; floats14[13] = m.opacity;
}

// which: 0 name, 1 map_Kd, 2 map_Ks, 3 map_bump, 4 map_Ka, 5 model directory
const char* orion_host_material_string(void* h, uint32_t mat, int which) {
    Model* md = static_cast<Model*>(h);
    if (which == 5) return md->directory.c_str();
    const Material& m = md->materials[mat];
    switch (which) {
    case 0: return m.name.c_str();
    case 1: return m.map_diffuse.c_str();
    case 2: return m.map_specular.c_str();
    case 3: return m.map_bump.c_str();
    default: return m.map_ambient.c_str();
    }
}

// Texture decode (png_reader.cpp, jpeg_reader.cpp; kind 0 = PNG, 1 = JPEG).  Returns 0 and a malloc'd
// buffer of width*height*channels bytes (release with orion_host_free), or -1 with a message in
// err (up to err_len bytes).
int orion_host_load_image(int kind, const char* path, int* width, int* height, int* channels, uint8_t** texels, char* err, int err_len) {
    std::vector<uint8_t> data;
    std::string e;
    const bool ok = kind == 0 ? load_png(path, *width, *height, *channels, data, e) : load_jpeg(path, *width, *height, *channels, data, e);
    if (!ok) {
        if (err && err_len > 0) { std::strncpy(err, e.c_str(), (size_t)err_len - 1); err[err_len - 1] = 0; }
        return -1;
    }
    *texels = static_cast<uint8_t*>(std::malloc(data.size()));
    std::memcpy(*texels, data.data(), data.size());
    return 0;
}
void orion_host_free(void* p) { std::free(p); }

} // extern "C"
