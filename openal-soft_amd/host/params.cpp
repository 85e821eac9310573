#include "params.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "tables.hpp"

namespace oalgpu {
namespace {

constexpr uint32_t kFracOne = 1u << OALGPU_MIXER_FRAC_BITS;

// float2uint (common/alnumeric.h:223-240): truncating, clamping float -> unsigned.
uint32_t TruncToUint(float f)
{
    int32_t bits;
    std::memcpy(&bits, &f, sizeof(bits));
    const uint32_t keep = static_cast<uint32_t>(bits >> 31) ^ 0xffffffffu;
    const int shift = ((bits >> 23) & 0xff) - (127 + 23);
    if(shift < -23) return 0;
    if(shift > 8) return 0xffffffffu & keep;
    const uint32_t mant = (static_cast<uint32_t>(bits) & 0x7fffffu) | 0x800000u;
    return ((shift < 0) ? (mant >> -shift) : (mant << shift)) & keep;
}

int BsincFamily(int resampler)
{
    switch(resampler)
    {
    case OALGPU_RESAMPLER_FAST_BSINC12: case OALGPU_RESAMPLER_BSINC12: return 12;
    case OALGPU_RESAMPLER_FAST_BSINC24: case OALGPU_RESAMPLER_BSINC24: return 24;
    default: return 48;
    }
}

} // namespace

void PrepareResampler(int resampler, uint32_t increment, oalgpu_interp_state *out)
{
    *out = oalgpu_interp_state{};
    switch(resampler)
    {
    case OALGPU_RESAMPLER_POINT: out->kind = 0; return;
    case OALGPU_RESAMPLER_LINEAR: out->kind = 1; return;
    case OALGPU_RESAMPLER_SPLINE: out->kind = 2; out->table = 0; return;
    case OALGPU_RESAMPLER_GAUSSIAN: out->kind = 2; out->table = 1; return;
    default: break;
    }
    const bool fastOnly = resampler == OALGPU_RESAMPLER_FAST_BSINC12
        || resampler == OALGPU_RESAMPLER_FAST_BSINC24 || resampler == OALGPU_RESAMPLER_FAST_BSINC48;
    const int family = BsincFamily(resampler);
    const BsincTable *table = GetBsincTable(family);

    // BsincPrepare: scale index + curve-fitted interpolation factor when down-sampling.
    size_t si = 15;
    float sf = 0.0f;
    if(increment > kFracOne)
    {
        sf = float(kFracOne) / float(increment) - table->scaleBase;
        sf = std::max(0.0f, 16.0f * sf * table->scaleRange - 1.0f);
        si = TruncToUint(sf);
        sf -= float(si);
        sf = 1.0f - std::sqrt(1.0f - sf * sf);
    }
    out->kind = (!fastOnly && increment > kFracOne) ? 4 : 3;
    out->table = family;
    out->sf = sf;
    out->m = table->m[si];
    out->l = out->m / 2u - 1u;
    out->filter_offset = table->filterOffset[si];
}

void DesignBiquadFromSlope(int type, float f0norm, float gain, float slope, float c[5])
{
    gain = std::max(gain, 0.001f);                       // setParamsFromSlope: -60 dB floor
    DesignBiquad(type, f0norm, gain, std::sqrt((gain + 1.0f / gain) * (1.0f / slope - 1.0f) + 2.0f), c);
}

void DesignBiquadFromBandwidth(int type, float f0norm, float gain, float bandwidth, float c[5])
{   // rcpQFromBandwidth, biquad.h:71-75
    const float w0 = 3.14159265358979323846f * 2.0f * f0norm;
    DesignBiquad(type, f0norm, gain, 2.0f * std::sinh(std::log(2.0f) / 2.0f * bandwidth * w0 / std::sin(w0)), c);
}

void DesignBiquad(int type, float f0norm, float gain, float rcpQ, float c[5])
{
    gain = std::max(gain, 0.00001f);                     // SetParams: -100 dB floor
    const float w0 = 3.14159265358979323846f * 2.0f * std::min(f0norm, 0.49f);
    const float sw = std::sin(w0), cw = std::cos(w0);
    const float alpha = sw / 2.0f * rcpQ;
    float a0 = 1.0f, a1 = 0.0f, a2 = 0.0f, b0 = 1.0f, b1 = 0.0f, b2 = 0.0f;
    switch(type)
    {
    case OALGPU_BIQUAD_HIGHSHELF:
        {
            const float sa = 2.0f * std::sqrt(gain) * alpha;
            b0 = gain * ((gain + 1.0f) + (gain - 1.0f) * cw + sa);
            b1 = -2.0f * gain * ((gain - 1.0f) + (gain + 1.0f) * cw);
            b2 = gain * ((gain + 1.0f) + (gain - 1.0f) * cw - sa);
            a0 = (gain + 1.0f) - (gain - 1.0f) * cw + sa;
            a1 = 2.0f * ((gain - 1.0f) - (gain + 1.0f) * cw);
            a2 = (gain + 1.0f) - (gain - 1.0f) * cw - sa;
        }
        break;
    case OALGPU_BIQUAD_LOWSHELF:
        {
            const float sa = 2.0f * std::sqrt(gain) * alpha;
            b0 = gain * ((gain + 1.0f) - (gain - 1.0f) * cw + sa);
            b1 = 2.0f * gain * ((gain - 1.0f) - (gain + 1.0f) * cw);
            b2 = gain * ((gain + 1.0f) - (gain - 1.0f) * cw - sa);
            a0 = (gain + 1.0f) + (gain - 1.0f) * cw + sa;
            a1 = -2.0f * ((gain - 1.0f) + (gain + 1.0f) * cw);
            a2 = (gain + 1.0f) + (gain - 1.0f) * cw - sa;
        }
        break;
    case OALGPU_BIQUAD_PEAKING:
        b0 = 1.0f + alpha * gain; b1 = -2.0f * cw; b2 = 1.0f - alpha * gain;
        a0 = 1.0f + alpha / gain; a1 = -2.0f * cw; a2 = 1.0f - alpha / gain;
        break;
    case OALGPU_BIQUAD_LOWPASS:
        b0 = (1.0f - cw) / 2.0f; b1 = 1.0f - cw; b2 = (1.0f - cw) / 2.0f;
        a0 = 1.0f + alpha; a1 = -2.0f * cw; a2 = 1.0f - alpha;
        break;
    case OALGPU_BIQUAD_HIGHPASS:
        b0 = (1.0f + cw) / 2.0f; b1 = -(1.0f + cw); b2 = (1.0f + cw) / 2.0f;
        a0 = 1.0f + alpha; a1 = -2.0f * cw; a2 = 1.0f - alpha;
        break;
    case OALGPU_BIQUAD_BANDPASS:
        b0 = alpha; b1 = 0.0f; b2 = -alpha;
        a0 = 1.0f + alpha; a1 = -2.0f * cw; a2 = 1.0f - alpha;
        break;
    }
    c[0] = b0 / a0; c[1] = b1 / a0; c[2] = b2 / a0; c[3] = a1 / a0; c[4] = a2 / a0;
}

void ApplyBiquadTarget(oalgpu_biquad *f, const float c[5])
{
    float *tgt[5] = {&f->tb0, &f->tb1, &f->tb2, &f->ta1, &f->ta2};
    bool changed = false;
    for(int i = 0; i < 5; ++i)
    {
        changed |= !(std::fabs(c[i] - *tgt[i]) <= 0.015625f);
        *tgt[i] = c[i];
    }
    auto snap = [f] {
        f->counter = 0;
        f->b0 = f->tb0; f->b1 = f->tb1; f->b2 = f->tb2; f->a1 = f->ta1; f->a2 = f->ta2;
    };
    if(!changed) { if(f->counter <= 0) snap(); }
    else if(f->counter >= 0) f->counter = 8 * 32;   // InterpSteps * SamplesPerStep
    else snap();
}

float SplitterCoeff(float f0norm)
{
    const float w = 3.14159265358979323846f * 2.0f * std::min(f0norm, 0.49f);
    const float cw = std::cos(w);
    if(cw > 1.1920928955078125e-07f) return (std::sin(w) - 1.0f) / cw;
    return cw * -0.5f;
}

namespace {

// the gain product of a section at angular frequency w and, in out[1..o], the derived
// coefficients (NfcFilterCreateN / NfcFilterAdjustN, nfc.cpp:56-203; B1..B4 :49-52)
float NfcSection(int o, float w, float *out)
{
    const float r = 0.5f * w;
    if(o == 1)
    {
        const float b00 = 1.0f * r;
        const float g0 = 1.0f + b00;
        out[1] = 2.0f * b00 / g0;
        return g0;
    }
    if(o == 2)
    {
        const float b10 = 3.0f * r, b11 = 3.0f * (r * r);
        const float g1 = 1.0f + b10 + b11;
        out[1] = (2.0f * b10 + 4.0f * b11) / g1;
        out[2] = 4.0f * b11 / g1;
        return g1;
    }
    if(o == 3)
    {
        const float b10 = 3.6778f * r, b11 = 6.4595f * (r * r), b00 = 2.3222f * r;
        const float g1 = 1.0f + b10 + b11, g0 = 1.0f + b00;
        out[1] = (2.0f * b10 + 4.0f * b11) / g1;
        out[2] = 4.0f * b11 / g1;
        out[3] = 2.0f * b00 / g0;
        return g1 * g0;
    }
    const float b10 = 4.2076f * r, b11 = 11.4877f * (r * r), b00 = 5.7924f * r, b01 = 9.1401f * (r * r);
    const float g1 = 1.0f + b10 + b11, g0 = 1.0f + b00 + b01;
    out[1] = (2.0f * b10 + 4.0f * b11) / g1;
    out[2] = 4.0f * b11 / g1;
    out[3] = (2.0f * b00 + 4.0f * b01) / g0;
    out[4] = 4.0f * b01 / g0;
    return g1 * g0;
}

} // namespace

void NfcInit(float w1, NfcDesign &d)
{
    std::memset(&d, 0, sizeof(d));
    for(int o = 1; o <= 4; ++o)
    {
        const float g = NfcSection(o, w1, d.a[o]);
        d.baseGain[o] = 1.0f / g;
        d.a[o][0] = 1.0f;
        for(int k = 1; k <= o; ++k) d.b[o][k] = d.a[o][k];
    }
}

void NfcAdjust(float w0, NfcDesign &d)
{
    for(int o = 1; o <= 4; ++o)
    {
        const float g = NfcSection(o, w0, d.b[o]);
        d.a[o][0] = d.baseGain[o] * g;
    }
}

} // namespace oalgpu
