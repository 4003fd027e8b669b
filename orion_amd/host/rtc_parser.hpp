// .rtc scene files: same grammar and the same error behaviour as the
// reference's parse_rtc / write_rtc (orion/rtc_parser.cpp:69-163).
#pragma once

#include <string>

#include "scene_types.hpp"

namespace orion_amd {

// Reads `path`.  On an unopenable file prints "Error opening file <path>" and
// returns a default-initialised RtcData (the reference keeps going with an
// empty scene, orion/rtc_parser.cpp:73-77).
RtcData parse_rtc(const std::string& path);

// Writes the format parse_rtc reads (orion/rtc_parser.cpp:136-163), one light
// per line so that files with several lights read back.
void write_rtc(const std::string& path, const RtcData& rtc);

} // namespace orion_amd
