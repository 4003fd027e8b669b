#include "raytracer.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

#include <zlib.h>

#include <orion_hip.h>

#include "bvh_builder.hpp"
#include "obj_loader.hpp"
#include "rtc_parser.hpp"

namespace orion_amd {

namespace {

std::string parent_dir(const std::string& p) {
    size_t k = p.find_last_of('/');
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}

std::string extension_of(const std::string& p) {
    size_t k = p.find_last_of('.');
    size_t s = p.find_last_of('/');
    if (k == std::string::npos || (s != std::string::npos && k < s)) return "";
    return p.substr(k);
}

bool ends_with(const std::string& s, const std::string& suf) {
    return s.size() >= suf.size() && std::equal(suf.rbegin(), suf.rend(), s.rbegin(), [](char a, char b) { return std::tolower(a) == std::tolower(b); });
}

// Texture pool for scene creation: every material gets four entries (ambient, diffuse,
// specular, emissive), each an image or the 1x1 solid colour (orion/material.hpp:133-138).
struct TexturePool {
    std::vector<OrionTextureDesc> descs;
    std::vector<std::vector<float>> solid;
    std::vector<std::vector<uint8_t>> images;
    std::vector<std::pair<std::string, uint32_t>> loaded;   // path -> index into images

    uint32_t add_solid(const float rgb[3]) {
        solid.emplace_back(rgb, rgb + 3);
        descs.push_back({1, 1, ORION_TEX_RGB_F32, nullptr});
        return (uint32_t)descs.size() - 1;
    }
    // returns false when the file cannot be used; the caller falls back to the solid colour
    bool add_image(const std::string& path, uint32_t& out) {
        for (auto& kv : loaded) if (kv.first == path) { out = kv.second; return true; }
        int w = 0, h = 0, ch = 3;
        std::vector<uint8_t> texels;
        if (ends_with(path, ".ppm")) {
            if (!load_ppm(path, w, h, texels)) { std::fprintf(stderr, "texture %s: not a binary PPM; using the solid colour\n", path.c_str()); return false; }
        } else if (ends_with(path, ".png")) {
            std::string err;
            if (!load_png(path, w, h, ch, texels, err)) { std::fprintf(stderr, "texture %s: %s; using the solid colour\n", path.c_str(), err.c_str()); return false; }
            if (ch == 2) {   // the reference refuses two-channel images (orion/texture.hpp:39-42)
                std::fprintf(stderr, "Unknown texture format (2 color channels) in path: %s\n", path.c_str());
                return false;
            }
        } else if (ends_with(path, ".jpg") || ends_with(path, ".jpeg")) {
            std::string err;
            if (!load_jpeg(path, w, h, ch, texels, err)) { std::fprintf(stderr, "texture %s: %s; using the solid colour\n", path.c_str(), err.c_str()); return false; }
        } else {
            std::fprintf(stderr, "texture %s: this tool decodes PNG, baseline JPEG and binary PPM images; using the solid colour\n", path.c_str());
            return false;
        }
        images.push_back(std::move(texels));
        descs.push_back({(uint32_t)w, (uint32_t)h, uint32_t(ch == 1 ? ORION_TEX_GRAY_U8 : ch == 4 ? ORION_TEX_RGBA_U8 : ORION_TEX_RGB_U8), nullptr});
        out = (uint32_t)descs.size() - 1;
        loaded.emplace_back(path, out);
        return true;
    }
    void fix_pointers() {   // vectors may have moved while growing: patch texel pointers last
        size_t si = 0, ii = 0;
        for (OrionTextureDesc& d : descs) {
            if (d.format == ORION_TEX_RGB_F32) d.texels = solid[si++].data();
            else d.texels = images[ii++].data();
        }
    }
};

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back(uint8_t(x >> 24)); v.push_back(uint8_t(x >> 16)); v.push_back(uint8_t(x >> 8)); v.push_back(uint8_t(x));
}

void png_chunk(std::ofstream& f, const char type[4], const std::vector<uint8_t>& data) {
    std::vector<uint8_t> head;
    put_be32(head, (uint32_t)data.size());
    f.write((const char*)head.data(), 4);
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    f.write((const char*)body.data(), (std::streamsize)body.size());
    std::vector<uint8_t> tail;
    put_be32(tail, (uint32_t)crc32(0L, body.data(), (uInt)body.size()));
    f.write((const char*)tail.data(), 4);
}

} // namespace

bool load_ppm(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgb) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    std::string magic;
    f >> magic;
    if (magic != "P6") return false;
    int vals[3], got = 0;
    while (got < 3 && f) {
        int c = f.peek();
        if (c == '#') { std::string skip; std::getline(f, skip); continue; }
        if (std::isspace(c)) { f.get(); continue; }
        f >> vals[got++];
    }
    if (got < 3 || vals[2] != 255 || vals[0] <= 0 || vals[1] <= 0) return false;
    f.get();   // the single whitespace byte before the raster
    width = vals[0]; height = vals[1];
    rgb.resize((size_t)width * height * 3);
    f.read((char*)rgb.data(), (std::streamsize)rgb.size());
    return (size_t)f.gcount() == rgb.size();
}

int RayTracer::traceRTC(const std::string& rtc_file_name, const std::string& path_to_image,
                        unsigned samples, unsigned light_samples, unsigned /*threads*/) {
    int status = 0;
    RtcData rtc = parse_rtc(rtc_file_name);
    width_ = std::max(0, rtc.xres); height_ = std::max(0, rtc.yres);
    image_.assign((size_t)width_ * height_ * 3, 0.0f);

    Model model;
    load_obj(parent_dir(rtc_file_name) + "/" + rtc.obj_file, model);
    build_bvh(model, static_cast<SplitStrategy>(bvh_strategy));
    size_t n_emissive = 0;
    for (const Mesh& m : model.meshes) n_emissive += model.materials[m.material].is_emissive();
    std::printf("Number of emissive meshes: %zu\n", n_emissive);

    TexturePool pool;
    std::vector<OrionMaterialDesc> mats(model.materials.size());
    for (size_t k = 0; k < model.materials.size(); k++) {
        const Material& m = model.materials[k];
        OrionMaterialDesc& d = mats[k];
        d.tex_ambient = pool.add_solid(m.ambient);
        d.tex_diffuse = pool.add_solid(m.diffuse);
        d.tex_specular = pool.add_solid(m.specular);
        d.tex_emissive = pool.add_solid(m.emissive);
        uint32_t img;
        if (!m.map_diffuse.empty() && pool.add_image(model.directory + "/" + m.map_diffuse, img)) {
            d.tex_diffuse = img;
            // the reference installs the DIFFUSE map as specular map whenever a specular map is named (orion/model.cpp:186-187)
            if (!m.map_specular.empty()) d.tex_specular = img;
        }
        d.shininess = m.shininess;
        std::memcpy(d.emissive_color, m.emissive, 12);
    }
    pool.fix_pointers();
    std::vector<OrionMeshDesc> meshes(model.meshes.size());
    for (size_t i = 0; i < model.meshes.size(); i++) {
        const Mesh& m = model.meshes[i];
        static_assert(sizeof(BvhNode) == sizeof(OrionBvhNode), "host and ABI node layouts must agree");
        meshes[i] = {&m.vertices[0].position[0], (uint32_t)m.vertices.size(), m.indices.data(), (uint32_t)m.indices.size(),
                     reinterpret_cast<const OrionBvhNode*>(m.nodes.data()), (uint32_t)m.nodes.size(), m.tri_order.data(), m.material};
    }

    if (width_ > 0 && height_ > 0 && samples > 0) {
        OrionScene* scene = nullptr;
        if (orion_hip_scene_create(meshes.data(), (uint32_t)meshes.size(), mats.data(), (uint32_t)mats.size(),
                                   pool.descs.data(), (uint32_t)pool.descs.size(), device, &scene) != 0) {
            std::printf("ERROR::ORION_HIP:: %s\n", orion_hip_last_error());
            status = 2;
        } else {
            OrionCamera cam;
            orion_hip_camera_vectors(rtc.view_point, rtc.look_at, rtc.vector_up, rtc.y_view, rtc.aspect_ratio, &cam);
            std::vector<OrionLight> lights(rtc.lights.size());
            for (size_t i = 0; i < lights.size(); i++) std::memcpy(&lights[i], &rtc.lights[i], sizeof(OrionLight));
            OrionRenderParams p{};
            p.width = (uint32_t)width_; p.height = (uint32_t)height_;
            p.spp_begin = 0; p.spp_end = samples;
            p.light_samples = light_samples;
            p.max_depth = (uint32_t)std::max(0, rtc.recursion_level);
            p.seed = seed;
            p.n_lights = (uint32_t)lights.size(); p.lights = lights.data();
            std::vector<float> accum((size_t)width_ * height_ * 4, 0.0f);
            if (orion_hip_render(scene, &cam, &p, accum.data()) != 0) {
                std::printf("ERROR::ORION_HIP:: %s\n", orion_hip_last_error());
                status = 3;
            }
            const float inv_samples = 1.0f / samples;
            for (size_t px = 0; px < (size_t)width_ * height_; px++)
                for (int c = 0; c < 3; c++) image_[3 * px + c] = accum[4 * px + c] * inv_samples;
            orion_hip_scene_destroy(scene);
        }
    }
    std::printf("Triangles:                  %u\n", model.triangle_count());
    if (status != 0) {   // the device did not render: no image is written (a zeroed one would pass for a result)
        std::printf("ERROR::ORION_HIP:: rendering failed, %s not written\n", path_to_image.c_str());
        return status;
    }

    const std::string ext = extension_of(path_to_image);
    if (ext == ".ppm") savePPM(path_to_image);
    else if (ext == ".png") savePNG(path_to_image);
    else if (ext == ".hdr") saveHDR(path_to_image);
    else {
        std::printf("Extension not recognized \"%s\" (will be saved as .hdr), supported extensions: .hdr .png .ppm\n", ext.c_str());
        saveHDR(path_to_image);
    }
    return 0;
}

void RayTracer::savePPM(const std::string& path) const {
    std::ofstream f(path, std::ios::out | std::ios::binary);
    f << "P6\n" << width_ << " " << height_ << "\n255\n";
    for (size_t i = 0; i < image_.size(); i++) f << (unsigned char)(std::min(1.0f, image_[i]) * 255);
}

void RayTracer::savePNG(const std::string& path, bool normalize) const {
    float maxi = 1.0f;
    if (normalize) {
        maxi = 0.0f;
        for (float v : image_) maxi = std::max(maxi, v);
    }
    const float inv_scale = 1.0f / maxi;
    std::vector<uint8_t> raw;
    raw.reserve((size_t)height_ * (1 + 4 * (size_t)width_));
    for (int y = 0; y < height_; y++) {
        raw.push_back(0);   // filter type: none
        for (int x = 0; x < width_; x++) {
            const float* p = &image_[3 * ((size_t)y * width_ + x)];
            for (int c = 0; c < 3; c++) raw.push_back((uint8_t)(std::min(1.0f, p[c] * inv_scale) * 255));
            raw.push_back(255);
        }
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        std::printf("Error ocurred when saving file to %s\n", path.c_str());
        return;
    }
    z.resize(zlen);
    std::ofstream f(path, std::ios::out | std::ios::binary);
    if (!f) { std::printf("Error ocurred when saving file to %s\n", path.c_str()); return; }
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    f.write((const char*)sig, 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)width_); put_be32(ihdr, (uint32_t)height_);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGBA
    png_chunk(f, "IHDR", ihdr);
    png_chunk(f, "IDAT", z);
    png_chunk(f, "IEND", {});
}

void RayTracer::saveHDR(const std::string& path) const {
    // Radiance RGBE, flat (non run-length-encoded) scanlines
    std::ofstream f(path, std::ios::out | std::ios::binary);
    if (!f) { std::printf("Error ocurred when saving file to %s\n", path.c_str()); return; }
    f << "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y " << height_ << " +X " << width_ << "\n";
    for (size_t px = 0; px < (size_t)width_ * height_; px++) {
        const float* p = &image_[3 * px];
        const float m = std::max(p[0], std::max(p[1], p[2]));
        uint8_t out[4] = {0, 0, 0, 0};
        if (m >= 1e-32f) {
            int e;
            const float scale = std::frexp(m, &e) * 256.0f / m;
            out[0] = (uint8_t)(p[0] * scale); out[1] = (uint8_t)(p[1] * scale); out[2] = (uint8_t)(p[2] * scale);
            out[3] = (uint8_t)(e + 128);
        }
        f.write((const char*)out, 4);
    }
}

} // namespace orion_amd
