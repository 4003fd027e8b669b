// Baseline JPEG decoder for texture maps (hindu's and view_test's maps are baseline JPEG).  The
// reference decodes through its vendored stb_image (orion/texture.hpp:33-70), and a JPEG's texels
// depend on the decoder's arithmetic, so this one restates stb_image's: coefficients dequantised
// into 16-bit storage, the jidctint-derived integer IDCT with its two rounding points, the
// 3:1 / 9:3:3:1 chroma upsampling filters, and the 20-bit fixed-point YCbCr -> RGB conversion.
// Output as stbi_load(..., 0) returns it: 8 bits per channel, 1 channel for greyscale files,
// 3 otherwise.  Not handled (reported as failures): progressive and arithmetic-coded files,
// 12-bit samples, CMYK/YCCK.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "scene_types.hpp"

namespace orion_amd {

namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {
    bool present = false;
    uint8_t size[256]; uint8_t value[256];
    int mincode[17], maxcode[17], first[17]; int n = 0;   // codes of length L: [mincode, maxcode), values from index first
    bool build(const uint8_t counts[16], const uint8_t* vals) {
        int k = 0;
        for (int len = 1; len <= 16; len++)
            for (int i = 0; i < counts[len - 1]; i++) { if (k >= 256) return false; size[k] = (uint8_t)len; value[k] = vals[k]; k++; }
        n = k;
        int c = 0; k = 0;
        for (int len = 1; len <= 16; len++) {
            mincode[len] = c; first[len] = k;
            c += counts[len - 1]; k += counts[len - 1];
            if (counts[len - 1] && c - 1 >= (1 << len)) return false;
            maxcode[len] = c;
            c <<= 1;
        }
        present = true;
        return true;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int w2 = 0, h2 = 0, x = 0, y = 0;   // padded size, real size in samples
    int pred = 0;
    std::vector<uint8_t> data;
};

struct BitReader {
    const uint8_t* p; const uint8_t* end;
    uint32_t acc = 0; int bits = 0;
    int marker = -1;   // a marker met inside entropy-coded data
    void fill() {
        while (bits <= 24) {
            int b = 0;
            if (marker < 0 && p < end) {
                b = *p++;
                if (b == 0xFF) {
                    int c = p < end ? *p++ : 0;
                    while (c == 0xFF && p < end) c = *p++;
                    if (c != 0) { marker = c; b = 0; }
                }
            }
            acc |= (uint32_t)b << (24 - bits);
            bits += 8;
        }
    }
    int get(int n) {   // n <= 16
        if (n == 0) return 0;
        if (bits < n) fill();
        const int v = (int)(acc >> (32 - n));
        acc <<= n; bits -= n;
        return v;
    }
    void reset() { acc = 0; bits = 0; marker = -1; }
};

int decode_symbol(BitReader& br, const Huffman& h) {
    if (br.bits < 16) br.fill();
    for (int len = 1; len <= 16; len++) {
        const int c = (int)(br.acc >> (32 - len));
        if (c >= h.mincode[len] && c < h.maxcode[len]) {
            br.acc <<= len; br.bits -= len;
            return h.value[h.first[len] + c - h.mincode[len]];
        }
    }
    return -1;
}

// JPEG's sign extension of an n-bit magnitude
inline int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

inline uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

inline int f2f(double x) { return (int)(x * 4096 + 0.5); }

// One 8-point pass.  64-bit intermediates: a valid file never leaves the 32-bit range (so the values
// are the reference decoder's), a corrupt one cannot overflow.
typedef long long i64;
struct Idct1D { i64 x0, x1, x2, x3, t0, t1, t2, t3; };
inline Idct1D idct_1d(i64 s0, i64 s1, i64 s2, i64 s3, i64 s4, i64 s5, i64 s6, i64 s7) {
    static const int c0541 = f2f(0.5411961f), cm1847 = f2f(-1.847759065f), c0765 = f2f(0.765366865f), c1175 = f2f(1.175875602f),
                     c0298 = f2f(0.298631336f), c2053 = f2f(2.053119869f), c3072 = f2f(3.072711026f), c1501 = f2f(1.501321110f),
                     cm0899 = f2f(-0.899976223f), cm2562 = f2f(-2.562915447f), cm1961 = f2f(-1.961570560f), cm0390 = f2f(-0.390180644f);
    Idct1D r;
    i64 p2 = s2, p3 = s6;
    i64 p1 = (p2 + p3) * c0541;
    i64 t2 = p1 + p3 * cm1847, t3 = p1 + p2 * c0765;
    p2 = s0; p3 = s4;
    i64 t0 = (p2 + p3) * 4096, t1 = (p2 - p3) * 4096;
    r.x0 = t0 + t3; r.x3 = t0 - t3; r.x1 = t1 + t2; r.x2 = t1 - t2;
    t0 = s7; t1 = s5; t2 = s3; t3 = s1;
    p3 = t0 + t2; i64 p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
    const i64 p5 = (p3 + p4) * c1175;
    t0 *= c0298; t1 *= c2053; t2 *= c3072; t3 *= c1501;
    p1 = p5 + p1 * cm0899; p2 = p5 + p2 * cm2562; p3 *= cm1961; p4 *= cm0390;
    r.t3 = t3 + p1 + p4; r.t2 = t2 + p2 + p3; r.t1 = t1 + p2 + p4; r.t0 = t0 + p1 + p3;
    return r;
}

inline uint8_t clamp255l(i64 x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

void idct_block(uint8_t* out, int stride, const int16_t d[64]) {
    int val[64];
    for (int i = 0; i < 8; i++) {          // columns, 2 extra bits of precision kept
        Idct1D r = idct_1d(d[i], d[8 + i], d[16 + i], d[24 + i], d[32 + i], d[40 + i], d[48 + i], d[56 + i]);
        r.x0 += 512; r.x1 += 512; r.x2 += 512; r.x3 += 512;
        val[i] = (int)((r.x0 + r.t3) >> 10); val[56 + i] = (int)((r.x0 - r.t3) >> 10);
        val[8 + i] = (int)((r.x1 + r.t2) >> 10); val[48 + i] = (int)((r.x1 - r.t2) >> 10);
        val[16 + i] = (int)((r.x2 + r.t1) >> 10); val[40 + i] = (int)((r.x2 - r.t1) >> 10);
        val[24 + i] = (int)((r.x3 + r.t0) >> 10); val[32 + i] = (int)((r.x3 - r.t0) >> 10);
    }
    for (int i = 0; i < 8; i++) {          // rows: remove 2^17, round, re-centre on 128
        const int* v = val + 8 * i;
        uint8_t* o = out + (size_t)stride * i;
        Idct1D r = idct_1d(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
        const i64 bias = 65536 + (128 << 17);
        r.x0 += bias; r.x1 += bias; r.x2 += bias; r.x3 += bias;
        o[0] = clamp255l((r.x0 + r.t3) >> 17); o[7] = clamp255l((r.x0 - r.t3) >> 17);
        o[1] = clamp255l((r.x1 + r.t2) >> 17); o[6] = clamp255l((r.x1 - r.t2) >> 17);
        o[2] = clamp255l((r.x2 + r.t1) >> 17); o[5] = clamp255l((r.x2 - r.t1) >> 17);
        o[3] = clamp255l((r.x3 + r.t0) >> 17); o[4] = clamp255l((r.x3 - r.t0) >> 17);
    }
}

// Row upsamplers: `near` is the source row closer to the output row, `far` the other one.
const uint8_t* up_row(std::vector<uint8_t>& buf, const uint8_t* near, const uint8_t* far, int w, int hs, int vs) {
    uint8_t* out = buf.data();
    if (hs == 1 && vs == 1) return near;
    if (hs == 1 && vs == 2) {
        for (int i = 0; i < w; i++) out[i] = (uint8_t)((3 * near[i] + far[i] + 2) >> 2);
        return out;
    }
    if (hs == 2 && vs == 1) {
        if (w == 1) { out[0] = out[1] = near[0]; return out; }
        out[0] = near[0];
        out[1] = (uint8_t)((near[0] * 3 + near[1] + 2) >> 2);
        int i;
        for (i = 1; i < w - 1; i++) {
            const int n = 3 * near[i] + 2;
            out[2 * i] = (uint8_t)((n + near[i - 1]) >> 2);
            out[2 * i + 1] = (uint8_t)((n + near[i + 1]) >> 2);
        }
        out[2 * i] = (uint8_t)((near[w - 2] * 3 + near[w - 1] + 2) >> 2);
        out[2 * i + 1] = near[w - 1];
        return out;
    }
    if (hs == 2 && vs == 2) {
        if (w == 1) { out[0] = out[1] = (uint8_t)((3 * near[0] + far[0] + 2) >> 2); return out; }
        int t1 = 3 * near[0] + far[0];
        out[0] = (uint8_t)((t1 + 2) >> 2);
        for (int i = 1; i < w; i++) {
            const int t0 = t1;
            t1 = 3 * near[i] + far[i];
            out[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
            out[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        out[2 * w - 1] = (uint8_t)((t1 + 2) >> 2);
        return out;
    }
    for (int i = 0; i < w; i++)            // any other factor: nearest neighbour
        for (int j = 0; j < hs; j++) out[i * hs + j] = near[i];
    return out;
}

inline int float2fixed(float x) { return ((int)(x * 4096.0f + 0.5f)) << 8; }

}  // namespace

bool load_jpeg(const std::string& path, int& width, int& height, int& channels, std::vector<uint8_t>& out, std::string& err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { err = "cannot open"; return false; }
    std::vector<uint8_t> file((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if (file.size() < 4 || file[0] != 0xFF || file[1] != 0xD8) { err = "not a JPEG file"; return false; }

    uint16_t quant[4][64] = {};
    Huffman hdc[4], hac[4];
    std::vector<Component> comp;
    int img_w = 0, img_h = 0, hmax = 1, vmax = 1, restart = 0;
    bool jfif = false; int app14 = -1;
    bool have_frame = false, done = false;
    size_t pos = 2;

    auto be16 = [&](size_t p) { return (int)file[p] << 8 | file[p + 1]; };

    while (!done && pos + 4 <= file.size()) {
        if (file[pos] != 0xFF) { pos++; continue; }
        const int m = file[pos + 1];
        if (m == 0xFF) { pos++; continue; }
        pos += 2;
        if (m == 0x00 || m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (pos + 2 > file.size()) break;
        const int len = be16(pos);
        if (len < 2 || pos + (size_t)len > file.size()) { err = "truncated segment"; return false; }
        const uint8_t* seg = &file[pos + 2];
        const int n = len - 2;
        if (m == 0xDB) {                                   // quantisation tables
            int i = 0;
            while (i < n) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                if (tq > 3 || pq > 1) { err = "bad DQT"; return false; }
                i++;
                if (i + (pq ? 128 : 64) > n) { err = "bad DQT"; return false; }
                for (int k = 0; k < 64; k++) { quant[tq][kZigzag[k]] = (uint16_t)(pq ? (seg[i] << 8 | seg[i + 1]) : seg[i]); i += pq ? 2 : 1; }
            }
        } else if (m == 0xC4) {                            // Huffman tables
            int i = 0;
            while (i + 17 <= n) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                if (tc > 1 || th > 3) { err = "bad DHT"; return false; }
                int total = 0;
                for (int k = 0; k < 16; k++) total += seg[i + 1 + k];
                if (total > 256 || i + 17 + total > n) { err = "bad DHT"; return false; }
                if (!(tc ? hac[th] : hdc[th]).build(seg + i + 1, seg + i + 17)) { err = "bad code lengths"; return false; }
                i += 17 + total;
            }
        } else if (m == 0xC0 || m == 0xC1) {               // baseline / extended sequential frame
            if (n < 6 || seg[0] != 8) { err = "only 8-bit samples are supported"; return false; }
            img_h = be16(pos + 3); img_w = be16(pos + 5);
            const int nc = seg[5];
            if (img_w <= 0 || img_h <= 0) { err = "empty image"; return false; }
            if ((uint64_t)img_w * img_h > (1u << 27)) { err = "image too large"; return false; }
            if (nc != 1 && nc != 3) { err = "only greyscale and three-component JPEGs are supported"; return false; }
            if (n < 6 + 3 * nc) { err = "bad SOF"; return false; }
            comp.assign(nc, Component());
            for (int k = 0; k < nc; k++) {
                comp[k].id = seg[6 + 3 * k]; comp[k].h = seg[7 + 3 * k] >> 4; comp[k].v = seg[7 + 3 * k] & 15; comp[k].tq = seg[8 + 3 * k];
                if (comp[k].h < 1 || comp[k].h > 4 || comp[k].v < 1 || comp[k].v > 4 || comp[k].tq > 3) { err = "bad SOF"; return false; }
                if (comp[k].h > hmax) hmax = comp[k].h;
                if (comp[k].v > vmax) vmax = comp[k].v;
            }
            for (Component& c : comp) if (hmax % c.h || vmax % c.v) { err = "fractional sampling ratios are not supported"; return false; }
            const int mcux = (img_w + 8 * hmax - 1) / (8 * hmax), mcuy = (img_h + 8 * vmax - 1) / (8 * vmax);
            for (Component& c : comp) {
                c.x = (img_w * c.h + hmax - 1) / hmax; c.y = (img_h * c.v + vmax - 1) / vmax;
                c.w2 = mcux * c.h * 8; c.h2 = mcuy * c.v * 8;
                c.data.assign((size_t)c.w2 * c.h2, 0);
            }
            have_frame = true;
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
            err = m == 0xC2 ? "progressive JPEG is not supported" : "this JPEG coding process is not supported";
            return false;
        } else if (m == 0xDD) {
            if (n >= 2) restart = seg[0] << 8 | seg[1];
        } else if (m == 0xE0) {
            if (n >= 5 && !std::memcmp(seg, "JFIF\0", 5)) jfif = true;
        } else if (m == 0xEE) {
            if (n >= 12 && !std::memcmp(seg, "Adobe\0", 6)) app14 = seg[11];
        } else if (m == 0xDA) {                            // scan
            if (!have_frame) { err = "scan before frame"; return false; }
            const int ns = seg[0];
            if (ns < 1 || ns > (int)comp.size() || n < 1 + 2 * ns + 3) { err = "bad SOS"; return false; }
            int order[3];
            for (int k = 0; k < ns; k++) {
                int which = -1;
                for (size_t c = 0; c < comp.size(); c++) if (comp[c].id == seg[1 + 2 * k]) which = (int)c;
                if (which < 0) { err = "bad SOS"; return false; }
                comp[which].td = seg[2 + 2 * k] >> 4; comp[which].ta = seg[2 + 2 * k] & 15;
                if (comp[which].td > 3 || comp[which].ta > 3 || !hdc[comp[which].td].present || !hac[comp[which].ta].present) { err = "scan uses a missing Huffman table"; return false; }
                order[k] = which;
            }
            BitReader br{&file[pos + len], file.data() + file.size()};
            for (Component& c : comp) c.pred = 0;
            int todo = restart ? restart : 0x7fffffff;
            auto block = [&](Component& c, int bx, int by) -> bool {
                int16_t data[64] = {};
                const int t = decode_symbol(br, hdc[c.td]);
                if (t < 0 || t > 15) return false;
                const int diff = t ? extend(br.get(t), t) : 0;
                c.pred = (int)((unsigned)c.pred + (unsigned)diff);
                data[0] = (int16_t)((unsigned)c.pred * quant[c.tq][0]);
                for (int k = 1; k < 64;) {
                    const int rs = decode_symbol(br, hac[c.ta]);
                    if (rs < 0) return false;
                    const int s = rs & 15, r = rs >> 4;
                    if (s == 0) { if (rs != 0xF0) break; k += 16; continue; }
                    k += r;
                    if (k > 63) return false;
                    const int z = kZigzag[k++];
                    data[z] = (int16_t)(extend(br.get(s), s) * quant[c.tq][z]);
                }
                idct_block(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, data);
                return true;
            };
            auto restart_check = [&]() -> bool {
                if (--todo > 0) return true;
                if (br.bits < 24) br.fill();
                if (br.marker < 0xD0 || br.marker > 0xD7) return true;   // no restart marker: carry on to the end of the data
                br.reset();
                for (Component& c : comp) c.pred = 0;
                todo = restart ? restart : 0x7fffffff;
                return true;
            };
            bool ok = true;
            if (ns == 1) {                                 // one component: its own block raster
                Component& c = comp[order[0]];
                const int bw = (c.x + 7) >> 3, bh = (c.y + 7) >> 3;
                for (int by = 0; by < bh && ok; by++)
                    for (int bx = 0; bx < bw && ok; bx++) { ok = block(c, bx, by); if (ok) restart_check(); }
            } else {                                       // interleaved MCUs
                const int mcux = (img_w + 8 * hmax - 1) / (8 * hmax), mcuy = (img_h + 8 * vmax - 1) / (8 * vmax);
                for (int my = 0; my < mcuy && ok; my++)
                    for (int mx = 0; mx < mcux && ok; mx++) {
                        for (int k = 0; k < ns && ok; k++) {
                            Component& c = comp[order[k]];
                            for (int y = 0; y < c.v && ok; y++)
                                for (int x = 0; x < c.h && ok; x++) ok = block(c, mx * c.h + x, my * c.v + y);
                        }
                        if (ok) restart_check();
                    }
            }
            if (!ok) { err = "corrupt entropy-coded data"; return false; }
            // continue after the scan's data: at the marker the bit reader stopped at, if any
            size_t p = (size_t)(br.p - file.data());
            if (br.marker >= 0) { pos = p - 2; continue; }
            pos = p;
            continue;
        }
        pos += (size_t)len;
    }
    if (!have_frame) { err = "no frame"; return false; }

    // upsample and colour-convert, one output row at a time
    const int nc = (int)comp.size();
    const int n_out = nc >= 3 ? 3 : 1;
    bool is_rgb = nc == 3 && ((comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') || (app14 == 0 && !jfif));
    out.assign((size_t)img_w * img_h * n_out, 0);
    struct Resample { int hs, vs, ystep, w_lores, ypos; const uint8_t *line0, *line1; std::vector<uint8_t> buf; };
    std::vector<Resample> rs(nc);
    for (int k = 0; k < nc; k++) {
        rs[k].hs = hmax / comp[k].h; rs[k].vs = vmax / comp[k].v;
        rs[k].ystep = rs[k].vs >> 1; rs[k].w_lores = (img_w + rs[k].hs - 1) / rs[k].hs; rs[k].ypos = 0;
        rs[k].line0 = rs[k].line1 = comp[k].data.data();
        rs[k].buf.assign((size_t)img_w + 8, 0);
    }
    const int cr_r = float2fixed(1.40200f), cr_g = -float2fixed(0.71414f), cb_g = -float2fixed(0.34414f), cb_b = float2fixed(1.77200f);
    for (int j = 0; j < img_h; j++) {
        const uint8_t* row[3] = {nullptr, nullptr, nullptr};
        for (int k = 0; k < nc; k++) {
            Resample& r = rs[k];
            const bool y_bot = r.ystep >= (r.vs >> 1);
            row[k] = up_row(r.buf, y_bot ? r.line1 : r.line0, y_bot ? r.line0 : r.line1, r.w_lores, r.hs, r.vs);
            if (++r.ystep >= r.vs) {
                r.ystep = 0;
                r.line0 = r.line1;
                if (++r.ypos < comp[k].y) r.line1 += comp[k].w2;
            }
        }
        uint8_t* o = &out[(size_t)j * img_w * n_out];
        if (nc == 1) { std::memcpy(o, row[0], (size_t)img_w); continue; }
        for (int i = 0; i < img_w; i++, o += 3) {
            if (is_rgb) { o[0] = row[0][i]; o[1] = row[1][i]; o[2] = row[2][i]; continue; }
            const int y_fixed = (row[0][i] << 20) + (1 << 19);
            const int cr = row[2][i] - 128, cb = row[1][i] - 128;
            int r = y_fixed + cr * cr_r;
            int g = y_fixed + cr * cr_g + (int)((uint32_t)(cb * cb_g) & 0xffff0000u);
            int b = y_fixed + cb * cb_b;
            o[0] = clamp255(r >> 20); o[1] = clamp255(g >> 20); o[2] = clamp255(b >> 20);
        }
    }
    width = img_w; height = img_h; channels = n_out;
    return true;
}

}  // namespace orion_amd
