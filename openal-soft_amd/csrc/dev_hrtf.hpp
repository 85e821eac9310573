// HRTF device functions: the dual-ear FIR of MixHrtf/MixHrtfBlend/MixDirectHrtf in gather
// form, and HrtfStore::getCoeffs.
//
// The reference scatters (core/mixer/hrtfbase.h:17-89):
//     for i in 0..N: Accum[i+j][ear] += Coeffs[j][ear] * (In[64 - Delay[ear] + i] * g(i)),  j < IrSize
// so output frame n receives its terms in order of ascending input index i.  The gather form
// below walks i upward for a fixed n, i.e. applies the very same terms in the very same order;
// with EXACT arithmetic (mul, then add) a frame is bit-identical to the scatter loop.
#pragma once
#include "dev_math.hpp"

#pragma clang fp contract(off)

namespace oalgpu {

enum HrtfGainKind { kGainRamp, kGainOldFade, kGainNewFade };

// g(i) of the three loops: MixHrtfBase (gain + step*i), MixHrtfBlendBase old filter
// (oldGain/N * (N-i)) and new filter (newStep*i, starting at i=1).
template<int KIND>
__device__ __forceinline__ float HrtfGain(float gain, float step, uint32_t i, uint32_t total)
{
    if constexpr(KIND == kGainRamp) return gain + step * float(i);
    else if constexpr(KIND == kGainOldFade) return step * float(total - i);
    else return step * float(i);
}

// Adds to (accL, accR) every term of output frame n.
//   in      : [History(64) | samples(N)]  (HrtfSamples in DoHrtfMix, core/voice.cpp:835-842)
//   coeffs  : HrirArray, interleaved {L,R}; irsize already rounded up to even in EXACT mode
//             (ApplyCoeffs SSE processes (IrSize+1)>>1 pairs, mixer_sse.cpp:46-51)
template<bool EXACT, int KIND, typename InPtr, typename CoefPtr>
__device__ __forceinline__ void HrtfGatherFrame(float &accL, float &accR, InPtr in, CoefPtr coeffs,
    uint32_t irsize, uint32_t delayL, uint32_t delayR, float gain, float step, uint32_t total, uint32_t n)
{
    const uint32_t first = (KIND == kGainNewFade) ? 1u : 0u;
    uint32_t lo = (n + 1 > irsize) ? n + 1 - irsize : 0u;
    if(lo < first) lo = first;
    if(total == 0) return;
    const uint32_t hi = n < total - 1 ? n : total - 1;
    for(uint32_t i = lo; i <= hi && hi != 0xffffffffu; ++i)
    {
        const float g = HrtfGain<KIND>(gain, step, i, total);
        const float left = in[kHist - delayL + i] * g;
        const float right = in[kHist - delayR + i] * g;
        const uint32_t j = n - i;
        accL = madd<EXACT>(accL, coeffs[j * 2 + 0], left);
        accR = madd<EXACT>(accR, coeffs[j * 2 + 1], right);
    }
}

// ---- HrtfStore::getCoeffs, core/hrtf.cpp:167-260 -------------------------------------------
struct HrtfStoreDev {
    uint32_t irSize, numFields, numElevs, numIrs;
    const float *fieldDistance;
    const uint8_t *fieldEvCount;
    const uint16_t *elevAzCount, *elevIrOffset;
    const float *coeffs;     // numIrs x 128 x 2
    const uint8_t *delays;   // numIrs x 2
};

struct HrirBlend { uint32_t idx[4]; float w[4]; float passthru; uint32_t delay[2]; };

__host__ __device__ __forceinline__ HrirBlend HrtfBlendFor(const HrtfStoreDev &st, float elevation, float azimuth,
    float distance, float spread)
{
    constexpr float invPi = 0.318309886183790671538f;
    HrirBlend b;
    const float dirfact = 1.0f - ((invPi / 2.0f) * spread);

    uint32_t ebase = 0, field = 0;
    for(; field + 1 < st.numFields; ++field)
    {
        if(distance >= st.fieldDistance[field]) break;
        ebase += st.fieldEvCount[field];
    }
    const uint32_t evcount = st.fieldEvCount[field];

    // CalcEvIndex, hrtf.cpp:167-173
    const float evf = (invPi * elevation + 0.5f) * float(evcount - 1);
    const uint32_t evraw = float2uint(evf);
    const uint32_t ev0 = evraw < evcount - 1u ? evraw : evcount - 1u;
    const float evBlend = evf - float(evraw);
    const uint32_t ev1 = (ev0 + 1u < evcount - 1u) ? ev0 + 1u : evcount - 1u;

    const uint32_t ir0 = st.elevIrOffset[ebase + ev0], ir1 = st.elevIrOffset[ebase + ev1];
    const uint32_t azc0 = st.elevAzCount[ebase + ev0], azc1 = st.elevAzCount[ebase + ev1];
    // CalcAzIndex, hrtf.cpp:178-184
    const float az0f = (invPi * 0.5f * azimuth + 1.0f) * float(azc0);
    const uint32_t az0raw = float2uint(az0f);
    const uint32_t az0 = az0raw % azc0;
    const float az0Blend = az0f - float(az0raw);
    const float az1f = (invPi * 0.5f * azimuth + 1.0f) * float(azc1);
    const uint32_t az1raw = float2uint(az1f);
    const uint32_t az1 = az1raw % azc1;
    const float az1Blend = az1f - float(az1raw);

    b.idx[0] = ir0 + az0;
    b.idx[1] = ir0 + ((az0 + 1u) % azc0);
    b.idx[2] = ir1 + az1;
    b.idx[3] = ir1 + ((az1 + 1u) % azc1);
    b.w[0] = (1.0f - evBlend) * (1.0f - az0Blend) * dirfact;
    b.w[1] = (1.0f - evBlend) * (az0Blend) * dirfact;
    b.w[2] = (evBlend) * (1.0f - az1Blend) * dirfact;
    b.w[3] = (evBlend) * (az1Blend) * dirfact;

    for(int e = 0; e < 2; ++e)
    {
        const float d = float(st.delays[b.idx[0] * 2 + e]) * b.w[0] + float(st.delays[b.idx[1] * 2 + e]) * b.w[1]
            + float(st.delays[b.idx[2] * 2 + e]) * b.w[2] + float(st.delays[b.idx[3] * 2 + e]) * b.w[3];
        b.delay[e] = fastf2u(d * 0.25f);
    }
    b.passthru = float(1.0 / 1.41421356237309504880) * (1.0f - dirfact);   // PassthruCoeff, hrtf.cpp:81
    return b;
}

// element e (0..255) of the blended HrirArray, hrtf.cpp:247-259: starts from the pass-through
// tap (elements 0,1) or 0, then adds the four weighted HRIRs in order (mul, then add).
__device__ __forceinline__ float HrtfBlendElement(const HrtfStoreDev &st, const HrirBlend &b, uint32_t e)
{
    float v = (e < 2) ? b.passthru : 0.0f;
#pragma unroll
    for(int c = 0; c < 4; ++c)
        v = st.coeffs[size_t{b.idx[c]} * (kHrirLen * 2) + e] * b.w[c] + v;
    return v;
}

} // namespace oalgpu
