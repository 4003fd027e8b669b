// Render entry point with the shape of the reference's RayTracer (orion/raytracer.hpp:19-51):
// traceRTC(rtc file, output image, samples, light samples, threads).  Scene loading and image
// output stay on the host; the pixel/sample loop and trace() run in liborion_hip.so.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "scene_types.hpp"

namespace orion_amd {

class RayTracer {
public:
    // As the reference's: load errors are printed and rendering carries on with whatever loaded (an
    // unreadable model gives a black image).  Unlike the reference's void function it returns a status:
    // non-zero when the device could not create the scene (2) or render it (3); no image is written then.
    // `threads` is accepted for command-line compatibility and ignored: the GPU needs no thread count.
    int traceRTC(const std::string& rtc_file_name, const std::string& path_to_image,
                  unsigned samples, unsigned light_samples, unsigned threads);

    uint64_t seed = 1;   // key of the counter-based generator (the reference seeds from `this`)
    int device = -1;     // HIP device, -1 = current
    int bvh_strategy = 0; // 0 MEDIAN (the reference's default), 1 MIDDLE, 2 SAH (orion/avx/sbvh.hpp:83-90)

    // Radiance image of the last traceRTC call, row-major rgb, already scaled by 1/samples.
    const std::vector<float>& image() const { return image_; }
    int width() const { return width_; }
    int height() const { return height_; }

protected:
    void savePPM(const std::string& path) const;                 // orion/raytracer.cpp:240-253
    void savePNG(const std::string& path, bool normalize = true) const;  // orion/raytracer.cpp:255-286
    void saveHDR(const std::string& path) const;                 // orion/raytracer.cpp:288-303

private:
    std::vector<float> image_;
    int width_ = 0, height_ = 0;
};

// Reads a binary PPM (P6, maxval 255) into 8-bit RGB texels; returns false if unreadable.
bool load_ppm(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgb);

} // namespace orion_amd
