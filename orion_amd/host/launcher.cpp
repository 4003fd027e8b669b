// Command-line front end with the reference's flags (orion/launcher.cpp:17-32):
//   orion_rt <rtc_file> [-o|--output FILE] [-p SPP] [-l LIGHT_SAMPLES] [-t|--threads N]
// plus --seed, --device and --bvh (split strategy of the host BVH build).  -t is accepted and ignored.
// Exit status: 0, or non-zero when the arguments are wrong or the device could not render (no image is written then).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "raytracer.hpp"

static void usage(const char* argv0) {
    std::printf("Usage: %s rtc_file [--output,-o FILE=raytracer.png] [-p PIXEL_SAMPLES=1] [-l SHADOW_RAY_SAMPLES=1]\n"
                "       [--threads,-t N (ignored)] [--seed S=1] [--device D=-1] [--bvh median|middle|sah]\n", argv0);
}

int main(int argc, char** argv) {
    std::string rtc_file, output = "raytracer.png";
    unsigned samples = 1, light_samples = 1, threads = 0;
    orion_amd::RayTracer rt;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto value = [&](const char* name) -> const char* {
            if (i + 1 >= argc) { std::printf("%s requires a value\n", name); usage(argv[0]); std::exit(105); }
            return argv[++i];
        };
        if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
        else if (a == "-o" || a == "--output") output = value("--output");
        else if (a == "-p") samples = (unsigned)std::strtoul(value("-p"), nullptr, 10);
        else if (a == "-l") light_samples = (unsigned)std::strtoul(value("-l"), nullptr, 10);
        else if (a == "-t" || a == "--threads") threads = (unsigned)std::strtoul(value("--threads"), nullptr, 10);
        else if (a == "--seed") rt.seed = std::strtoull(value("--seed"), nullptr, 10);
        else if (a == "--device") rt.device = std::atoi(value("--device"));
        else if (a == "--bvh") {
            const std::string v = value("--bvh");
            if (v == "median") rt.bvh_strategy = 0;
            else if (v == "middle") rt.bvh_strategy = 1;
            else if (v == "sah") rt.bvh_strategy = 2;
            else { std::printf("--bvh must be median, middle or sah\n"); return 105; }
        }
        else if (!a.empty() && a[0] == '-') { std::printf("Unknown option %s\n", a.c_str()); usage(argv[0]); return 109; }
        else if (rtc_file.empty()) rtc_file = a;
        else { std::printf("Unexpected argument %s\n", a.c_str()); usage(argv[0]); return 109; }
    }
    if (rtc_file.empty()) { std::printf("rtc_file is required\n"); usage(argv[0]); return 106; }
    return rt.traceRTC(rtc_file, output, samples, light_samples, threads);
}
