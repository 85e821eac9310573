// One-time HRTF set-up work the reference does on the CPU when a device opens: bringing a data set to the device's
// rate (GetLoadedHrtf, core/hrtf.cpp:539-606) and building the ambisonic-to-binaural decoder of the HRTF
// post-process (DirectHrtfState::build, core/hrtf.cpp:266-366).  Host code, double precision where the reference's is.
#pragma once
#include <cstdint>
#include <vector>

#include "mhr.hpp"

namespace oalgpu {

// core/hrtf.cpp:539-606: every HRIR through the polyphase resampler, delays and IrSize rescaled; h.sampleRate = devRate after.
void ResampleHrtfData(HrtfData &h, uint32_t devRate);

struct DirectHrtfBuild {
    std::vector<float> coeffs;      // numChans x 128 x 2: HrtfChannelState::mCoeffs
    std::vector<float> hfScale;     // numChans: mHfScale
    uint32_t irSize{0};             // DirectHrtfState::mIrSize
    float xoverNorm{0.0f};          // the band splitters' XOverFreq / rate
};
// points: numPoints x {elevation, azimuth} (radians); matrix: numPoints x 16 (ChannelCoeffs rows);
// orderHfGain: 5 (per ambisonic order).  numChans <= 16 ambisonic channels (ACN order).
DirectHrtfBuild BuildDirectHrtf(const HrtfData &h, uint32_t irSize, bool perHrirMin, const float *points, const float *matrix,
    uint32_t numPoints, uint32_t numChans, float xoverFreq, const float *orderHfGain);

} // namespace oalgpu
