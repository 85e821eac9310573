// Reader for the reference's HRTF data-set format (".mhr", "MinPHR03").  See mhr.cpp.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace oalgpu {

// In-memory image of HrtfStore (core/hrtf.h:22-44); coefficients keep the reference's
// HrirArray stride (128 taps x {L,R}) so getCoeffs can blend whole arrays like hrtf.cpp:251-259.
struct HrtfData {
    uint32_t sampleRate{}, irSize{};
    std::vector<float> fieldDistance;     // farthest field first (hrtf.h:31-34)
    std::vector<uint8_t> fieldEvCount;
    std::vector<uint16_t> elevAzCount, elevIrOffset;
    std::vector<float> coeffs;            // numIrs x 128 x 2
    std::vector<uint8_t> delays;          // numIrs x 2, quarter samples
    uint32_t numIrs() const { return static_cast<uint32_t>(delays.size() / 2); }
};

// Returns an empty string on success, else a description of what is wrong with the data.
std::string ParseMhr(const void *data, size_t size, HrtfData &out);

} // namespace oalgpu
