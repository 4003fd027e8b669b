// PNG decoder for texture maps (the reference decodes textures through stb_image,
// orion/texture.hpp:33-70; nanosuit's maps are PNG).  Output convention of
// stbi_load(..., 0): 8 bits per channel, the file's own channel count (grey 1, grey+alpha 2,
// RGB 3, RGBA 4), palettes expanded to RGB (RGBA with a tRNS chunk), a tRNS colour key adds an
// alpha channel, 16-bit samples keep their high byte, 1/2/4-bit grey is scaled to 0..255.
// PNG is lossless, so any correct decoder yields the same texels.  Not handled: Adam7
// interlacing (reported as a failure; the caller falls back to the solid colour).
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "scene_types.hpp"

namespace orion_amd {

namespace {

uint32_t be32(const uint8_t* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

}  // namespace

bool load_png(const std::string& path, int& width, int& height, int& channels, std::vector<uint8_t>& out, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open"; return false; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }

    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = reinterpret_cast<const char*>(&file[pos + 4]);
        if ((uint64_t)pos + 12 + len > file.size()) { err = "truncated chunk"; return false; }
        const uint8_t* data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            if (data[10] != 0 || data[11] != 0) { err = "unknown compression or filter method"; return false; }
        } else if (!std::memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
        else if (!std::memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!std::memcmp(type, "IEND", 4)) end = true;
        pos += 12 + (size_t)len;
    }
    if (ctype < 0 || w == 0 || h == 0 || w > (1u << 24) || h > (1u << 24)) { err = "missing or bad IHDR"; return false; }
    if ((uint64_t)w * h > (1u << 27)) { err = "image too large"; return false; }
    if (interlace) { err = "interlaced PNG is not supported"; return false; }
    int src_ch;
    switch (ctype) {
        case 0: src_ch = 1; break;
        case 2: src_ch = 3; break;
        case 3: src_ch = 1; break;
        case 4: src_ch = 2; break;
        case 6: src_ch = 4; break;
        default: err = "unknown colour type"; return false;
    }
    const bool depth_ok = ctype == 0 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)
                        : ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : (depth == 8 || depth == 16);
    if (!depth_ok) { err = "bit depth not allowed for the colour type"; return false; }
    if (ctype == 3 && palette.size() < 3) { err = "palette image without PLTE"; return false; }

    // inflate
    const size_t bits_per_px = (size_t)src_ch * depth;
    const size_t stride = ((size_t)w * bits_per_px + 7) / 8;
    const size_t bpp = bits_per_px >= 8 ? bits_per_px / 8 : 1;   // filter distance in bytes
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { err = "bad IDAT stream"; return false; }

    // unfilter in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* line = &raw[(stride + 1) * y + 1];
        const uint8_t* up = y ? line - (stride + 1) : zero.data();
        const int ft = line[-1];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? line[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
            int v = line[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "unknown filter type"; return false;
            }
            line[x] = (uint8_t)v;
        }
    }

    // samples -> 8-bit channels
    const bool key = !trns.empty() && (ctype == 0 || ctype == 2);
    const bool pal_alpha = ctype == 3 && !trns.empty();
    const int out_ch = ctype == 3 ? (pal_alpha ? 4 : 3) : src_ch + (key ? 1 : 0);
    out.assign((size_t)w * h * out_ch, 0);
    auto sample = [&](const uint8_t* line, size_t idx) -> uint32_t {   // raw sample value
        if (depth == 8) return line[idx];
        if (depth == 16) return (uint32_t)line[2 * idx] << 8 | line[2 * idx + 1];
        const size_t bit = idx * depth;
        return (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
    };
    const uint32_t scale = depth == 1 ? 255u : depth == 2 ? 85u : depth == 4 ? 17u : 1u;
    uint32_t key_v[3] = {0, 0, 0};
    if (key) for (int c = 0; c < (ctype == 0 ? 1 : 3) && 2 * c + 1 < (int)trns.size(); c++) key_v[c] = (uint32_t)trns[2 * c] << 8 | trns[2 * c + 1];
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* line = &raw[(stride + 1) * y + 1];
        uint8_t* dst = &out[(size_t)y * w * out_ch];
        for (uint32_t x = 0; x < w; x++, dst += out_ch) {
            if (ctype == 3) {
                const uint32_t i = sample(line, x);
                if (3 * i + 2 < palette.size()) { dst[0] = palette[3 * i]; dst[1] = palette[3 * i + 1]; dst[2] = palette[3 * i + 2]; }
                if (pal_alpha) dst[3] = i < trns.size() ? trns[i] : 255;
                continue;
            }
            bool is_key = key;
            for (int c = 0; c < src_ch; c++) {
                const uint32_t v = sample(line, (size_t)x * src_ch + c);
                dst[c] = depth == 16 ? (uint8_t)(v >> 8) : (uint8_t)(v * scale);
                if (key && c < (ctype == 0 ? 1 : 3) && v != key_v[c]) is_key = false;
            }
            if (key) dst[src_ch] = is_key ? 0 : 255;
        }
    }
    width = (int)w; height = (int)h; channels = out_ch;
    return true;
}

}  // namespace orion_amd
