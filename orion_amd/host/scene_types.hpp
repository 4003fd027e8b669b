// Host-side scene data handed to the HIP library through include/orion_hip.h.
// The layouts mirror what the reference keeps per mesh (orion/mesh.hpp:27-34,
// orion/vertex.hpp:8-21, orion/material.hpp:12-18, orion/light.hpp:8-13,
// orion/rtc_parser.hpp:11-31) so that a reference-side caller can fill the
// C-ABI descriptors straight from its own containers.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace orion_amd {

// 32-byte vertex: position, normal, texture coordinate (orion/vertex.hpp:8-21).
struct Vertex {
    float position[3];
    float normal[3];
    float uv[2];
};
static_assert(sizeof(Vertex) == 32, "Vertex must be 8 packed floats");

// One node of the host-built binary BVH (orion/avx/sbvh.hpp:44-66), children by index.
struct BvhNode {
    float lo[3];
    float hi[3];
    uint32_t is_leaf;
    uint32_t begin, end;     // leaf slots [begin, end) into tri_order
    uint32_t child[2];       // inner nodes only
};

struct Texture {
    std::string path;        // empty for a solid colour
    int width = 1, height = 1;
    std::vector<float> rgb;  // width*height*3, values in [0,1]
};

// Phong parameters read from the .mtl (orion/model.cpp:156-187).
struct Material {
    std::string name;
    float ambient[3] = {0, 0, 0};
    float diffuse[3] = {0.6f, 0.6f, 0.6f};
    float specular[3] = {0, 0, 0};
    float emissive[3] = {0, 0, 0};
    float shininess = 0.0f;
    float opacity = 1.0f;
    std::string map_diffuse, map_specular, map_bump, map_ambient;
    bool is_emissive() const { return emissive[0] != 0.0f || emissive[1] != 0.0f || emissive[2] != 0.0f; }
};

struct Mesh {
    std::string name;
    std::vector<Vertex> vertices;
    std::vector<uint32_t> indices;   // 3 per triangle
    uint32_t material = 0;
    // filled by build_bvh()
    std::vector<BvhNode> nodes;
    std::vector<uint32_t> tri_order; // originalID per leaf slot
};

struct Model {
    std::vector<Mesh> meshes;
    std::vector<Material> materials;
    std::string directory;
    uint32_t triangle_count() const {
        uint32_t n = 0;
        for (const Mesh& m : meshes) n += (uint32_t)(m.indices.size() / 3);
        return n;
    }
};

struct Light {
    float position[3];
    float color[3];
    float intensity;
};

// Contents of an .rtc scene file (orion/rtc_parser.hpp:11-31).
struct RtcData {
    int recursion_level = 0;
    int xres = 0, yres = 0;
    float aspect_ratio = 0.0f;
    float view_point[3] = {0, 0, 0};
    float look_at[3] = {0, 0, 0};
    float vector_up[3] = {0, 1, 0};
    float y_view = 1.0f;
    std::vector<Light> lights;
    std::string obj_file;
    std::string texture_file;
};

// png_reader.cpp: 8-bit texels with the file's channel count (1, 2, 3 or 4), as stbi_load(..., 0) returns them.
bool load_png(const std::string& path, int& width, int& height, int& channels, std::vector<uint8_t>& out, std::string& err);
// jpeg_reader.cpp: baseline JPEG with stb_image's arithmetic; 1 (greyscale) or 3 channels.
bool load_jpeg(const std::string& path, int& width, int& height, int& channels, std::vector<uint8_t>& out, std::string& err);

} // namespace orion_amd
