#include "rtc_parser.hpp"

#include <cstdio>
#include <fstream>
#include <sstream>

namespace orion_amd {

namespace {

// The reference treats every char <= 32 as blank, with plain (signed) char, so
// bytes >= 0x80 at either end of a line are stripped too (orion/rtc_parser.cpp:13-15).
bool is_blank(char c) { return static_cast<signed char>(c) <= 32; }

std::string strip(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && is_blank(s[a])) a++;
    while (b > a && is_blank(s[b - 1])) b--;
    return s.substr(a, b - a);
}

void read3(std::istringstream& in, float out[3]) { in >> out[0] >> out[1] >> out[2]; }

} // namespace

RtcData parse_rtc(const std::string& path) {
    RtcData rtc;
    std::ifstream file(path);
    if (!file) {
        std::printf("Error opening file %s\n", path.c_str());
        std::fflush(stdout);
        return rtc;
    }
    // Payload lines are positional: 1 obj, 2 texture, 3 depth, 4 resolution,
    // 5 eye, 6 look-at, 7 up, 8 y-view, 9.. lights.  Blank lines and lines that
    // start with '#' do not count (orion/rtc_parser.cpp:82-131).
    int field = 0, skipped = 0;
    std::string raw;
    while (std::getline(file, raw)) {
        const std::string line = strip(raw);
        if (line.empty() || line[0] == '#') { skipped++; continue; }
        field++;
        std::istringstream in(line);
        switch (field) {
        case 1: rtc.obj_file = line; break;
        case 2: rtc.texture_file = line; break;
        case 3: in >> rtc.recursion_level; break;
        case 4:
            in >> rtc.xres >> rtc.yres;
            rtc.aspect_ratio = float(rtc.xres) / float(rtc.yres);
            break;
        case 5: read3(in, rtc.view_point); break;
        case 6: read3(in, rtc.look_at); break;
        case 7: read3(in, rtc.vector_up); break;
        case 8: in >> rtc.y_view; break;
        default: {
            char tag = 0;
            in >> tag;
            if (tag != 'L' && tag != 'l') {
                std::printf("Invalid character '%c' in line %d `%s`\n", tag, field + skipped, line.c_str());
                std::fflush(stdout);
                break;
            }
            Light l{};
            read3(in, l.position);
            read3(in, l.color);
            for (float& c : l.color) c /= 255;  // file stores 0-255 RGB
            in >> l.intensity;
            rtc.lights.push_back(l);
        }
        }
    }
    return rtc;
}

void write_rtc(const std::string& path, const RtcData& rtc) {
    std::ofstream out(path);
    if (!out) {
        std::printf("Error opening file %s\n", path.c_str());
        return;
    }
    out << "# RTC scene written by orion_amd\n";
    out << rtc.obj_file << '\n' << rtc.texture_file << '\n' << rtc.recursion_level << '\n';
    out << rtc.xres << ' ' << rtc.yres << '\n';
    auto v3 = [&](const float v[3]) { out << v[0] << ' ' << v[1] << ' ' << v[2]; };
    v3(rtc.view_point); out << '\n';
    v3(rtc.look_at); out << '\n';
    v3(rtc.vector_up); out << '\n';
    out << rtc.y_view << '\n';
    for (const Light& l : rtc.lights) {
        out << "L ";
        v3(l.position);
        out << ' ' << int(l.color[0] * 255.0f) << ' ' << int(l.color[1] * 255.0f) << ' ' << int(l.color[2] * 255.0f);
        out << ' ' << l.intensity << '\n';
    }
}

} // namespace orion_amd
