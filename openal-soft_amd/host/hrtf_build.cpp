// See hrtf_build.hpp.  Operation for operation what core/hrtf.cpp does (cited per block), so that the results equal
// the compiled reference's bit for bit (tests/test_hrtf_build.py) -- up to the polyphase resampler, whose double
// sums are pinned to 1e-12 and come out as the same floats.
#include "hrtf_build.hpp"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

#include "polyphase.hpp"

namespace oalgpu {
namespace {

constexpr uint32_t kHrirLength = 128, kHistory = 64, kMaxHrirDelay = kHistory - 1;
constexpr uint32_t kDelayFracBits = 2, kDelayFracOne = 1u << kDelayFracBits, kDelayFracHalf = kDelayFracOne >> 1;   // core/hrtf.h:65-69

// float2uint / float2int (common/alnumeric.h): truncation
uint32_t Float2Uint(float f) { return f > 0.0f ? static_cast<uint32_t>(f) : 0u; }

struct IdxBlend { uint32_t idx; float blend; };
IdxBlend CalcEvIndex(uint32_t evcount, float ev)
{   // core/hrtf.cpp:167-173
    ev = (0.318309886183790671538f * ev + 0.5f) * static_cast<float>(evcount - 1);
    const uint32_t idx = Float2Uint(ev);
    return IdxBlend{std::min(idx, evcount - 1u), ev - static_cast<float>(idx)};
}
IdxBlend CalcAzIndex(uint32_t azcount, float az)
{   // core/hrtf.cpp:178-184
    az = (0.318309886183790671538f * 0.5f * az + 1.0f) * static_cast<float>(azcount);
    const uint32_t idx = Float2Uint(az);
    return IdxBlend{idx % azcount, az - static_cast<float>(idx)};
}

} // namespace

void ResampleHrtfData(HrtfData &h, uint32_t devRate)
{
    if(h.sampleRate == devRate || devRate == 0) return;
    Polyphase rs;
    rs.init(h.sampleRate, devRate);
    std::array<double, kHrirLength> in{}, out{};
    const uint32_t n = h.numIrs();
    for(uint32_t ir = 0; ir < n; ++ir)
        for(int e = 0; e < 2; ++e)
        {   // :545-560: one ear's 128 taps in, 128 out
            float *c = h.coeffs.data() + size_t{ir} * kHrirLength * 2 + e;
            for(uint32_t k = 0; k < kHrirLength; ++k) in[k] = c[2 * k];
            rs.process(in.data(), kHrirLength, out.data(), kHrirLength);
            for(uint32_t k = 0; k < kHrirLength; ++k) c[2 * k] = static_cast<float>(out[k]);
        }
    // :563-590: the delays for the new rate; shrunk together should the largest one no longer fit the history
    const float rateScale = static_cast<float>(devRate) / static_cast<float>(h.sampleRate);
    std::vector<float> nd(h.delays.size());
    float maxDelay = 0.0f;
    for(size_t i = 0; i < nd.size(); ++i)
    {
        nd[i] = std::round(static_cast<float>(h.delays[i]) * rateScale) / static_cast<float>(kDelayFracOne);
        maxDelay = std::max(maxDelay, nd[i]);
    }
    float delayScale = static_cast<float>(kDelayFracOne);
    if(maxDelay > static_cast<float>(kMaxHrirDelay)) delayScale *= static_cast<float>(kMaxHrirDelay) / maxDelay;
    for(size_t i = 0; i < nd.size(); ++i)
    {
        const int v = static_cast<int>(nd[i] * delayScale + 0.5f);          // float2int, then al::saturate_cast<u8>
        h.delays[i] = static_cast<uint8_t>(std::clamp(v, 0, 255));
    }
    // :595-599
    const float newIrSize = std::round(static_cast<float>(h.irSize) * rateScale);
    h.irSize = static_cast<uint32_t>(std::min(static_cast<float>(kHrirLength), newIrSize));
    h.sampleRate = devRate;
}

DirectHrtfBuild BuildDirectHrtf(const HrtfData &h, uint32_t irSize, bool perHrirMin, const float *points, const float *matrix,
    uint32_t numPoints, uint32_t numChans, float xoverFreq, const float *orderHfGain)
{
    DirectHrtfBuild out;
    out.xoverNorm = static_cast<float>(static_cast<double>(xoverFreq) / h.sampleRate);      // :280-281
    out.hfScale.resize(numChans);
    for(uint32_t c = 0; c < numChans; ++c)
    {   // :285-291: AmbiIndex::OrderFromChannel = floor(sqrt(acn))
        uint32_t order = 0;
        while((order + 1) * (order + 1) <= c) ++order;
        out.hfScale[c] = orderHfGain[order];
    }
    struct Impulse { uint32_t ir, ldelay, rdelay; };
    std::vector<Impulse> impulses;
    uint32_t minDelay = kHistory * kDelayFracOne, maxDelay = 0;
    const uint32_t evcount = h.fieldEvCount[0];
    for(uint32_t p = 0; p < numPoints; ++p)
    {   // :296-323: the closest measured HRIR of field 0 to every virtual speaker
        const IdxBlend e0 = CalcEvIndex(evcount, points[2 * p]);
        const uint32_t e1 = std::min(e0.idx + 1u, evcount - 1u);
        const uint32_t ir0 = h.elevIrOffset[e0.idx], ir1 = h.elevIrOffset[e1];
        const IdxBlend a0 = CalcAzIndex(h.elevAzCount[e0.idx], points[2 * p + 1]);
        const IdxBlend a1 = CalcAzIndex(h.elevAzCount[e1], points[2 * p + 1]);
        const uint32_t idx[4] = {ir0 + a0.idx, ir0 + ((a0.idx + 1) % h.elevAzCount[e0.idx]), ir1 + a1.idx,
            ir1 + ((a1.idx + 1) % h.elevAzCount[e1])};
        const uint32_t ir = idx[(e0.blend >= 0.5f ? 2u : 0u) + (a1.blend >= 0.5f ? 1u : 0u)];
        const Impulse im{ir, h.delays[size_t{ir} * 2], h.delays[size_t{ir} * 2 + 1]};
        minDelay = std::min(minDelay, std::min(im.ldelay, im.rdelay));
        maxDelay = std::max(maxDelay, std::max(im.ldelay, im.rdelay));
        impulses.push_back(im);
    }
    auto delayRound = [](uint32_t d) { return (d + kDelayFracHalf) >> kDelayFracBits; };
    std::vector<double> tmp(size_t{numChans} * kHrirLength * 2, 0.0);
    maxDelay = 0;
    for(uint32_t p = 0; p < numPoints; ++p)
    {   // :334-355: every channel's response = sum over the speakers of (decoder gain x delayed HRIR), in double
        const Impulse &im = impulses[p];
        const uint32_t base = perHrirMin ? std::min(im.ldelay, im.rdelay) : minDelay;
        const uint32_t ld = delayRound(im.ldelay - base), rd = delayRound(im.rdelay - base);
        maxDelay = std::max(maxDelay, std::max(im.ldelay, im.rdelay) - base);
        const float *hrir = h.coeffs.data() + size_t{im.ir} * kHrirLength * 2;
        for(uint32_t c = 0; c < numChans; ++c)
        {
            const double mult = matrix[size_t{p} * 16 + c];
            double *res = tmp.data() + size_t{c} * kHrirLength * 2;
            for(uint32_t k = 0; k + ld < kHrirLength; ++k) res[(k + ld) * 2] += double(hrir[k * 2]) * mult;
            for(uint32_t k = 0; k + rd < kHrirLength; ++k) res[(k + rd) * 2 + 1] += double(hrir[k * 2 + 1]) * mult;
        }
    }
    out.coeffs.resize(tmp.size());
    for(size_t i = 0; i < tmp.size(); ++i) out.coeffs[i] = static_cast<float>(tmp[i]);
    out.irSize = std::min(delayRound(maxDelay) + irSize, kHrirLength);       // :363-365
    return out;
}

} // namespace oalgpu
