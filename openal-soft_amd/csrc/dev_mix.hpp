// Gain-ramped mixing and the IIR filters (device functions).
#pragma once
#include "dev_math.hpp"
#include "../../include/oalgpu.h"

#pragma clang fp contract(off)

namespace oalgpu {

// One line of Mix_* (MixLine, core/mixer/mixer_c.cpp:150-186; the SSE variant
// mixer_sse.cpp:94-195 produces the same values) resolved per sample index p:
//   step = (target-current)*delta;  |step| > eps  =>  p < fade_len uses current + step*p
//   afterwards (or from p=0 when there is no ramp) the constant target gain, unless it is
//   below GainSilenceThreshold;  a fade longer than this mix leaves the line untouched after it.
struct MixLineGain {
    float cur, step, tgt;
    uint32_t fadeLen;      // samples using the ramp (0 when |step| <= eps)
    bool steady;           // constant part is mixed
    float newCur;          // value written back to Gains.Current
};

__device__ __forceinline__ MixLineGain PrepareMixLine(float cur, float tgt, uint32_t counter, uint32_t n)
{
    MixLineGain g;
    const float delta = (counter > 0) ? 1.0f / float(counter) : 0.0f;
    const uint32_t fadeLen = counter < n ? counter : n;
    g.cur = cur;
    g.tgt = tgt;
    g.step = (tgt - cur) * delta;
    g.fadeLen = 0;
    g.steady = true;
    g.newCur = tgt;
    if(fabsf(g.step) > kFltEpsilon)
    {
        g.fadeLen = fadeLen;
        if(fadeLen < counter)
        {
            g.newCur = cur + g.step * float(fadeLen);
            g.steady = false;
            return g;
        }
    }
    if(!(fabsf(tgt) > kGainSilence)) g.steady = false;
    return g;
}

// contribution of input sample `in` at index p to this line (added to the bus by the caller)
__device__ __forceinline__ bool MixLineActive(const MixLineGain &g, uint32_t p)
{ return p < g.fadeLen || g.steady; }
__device__ __forceinline__ float MixLineValue(const MixLineGain &g, float in, uint32_t p)
{
    if(p < g.fadeLen) return in * (g.cur + g.step * float(p));
    return in * g.tgt;
}

// ---- BiquadInterpFilter (core/filters/biquad.cpp) -----------------------------------------
struct BiquadState {                 // same field order as oalgpu_biquad
    float z1, z2, b0, b1, b2, a1, a2, tb0, tb1, tb2, ta1, ta2;
    int32_t counter;
};
static_assert(sizeof(BiquadState) == sizeof(oalgpu_biquad), "layout");

__device__ __forceinline__ void BiquadSnap(BiquadState &f)
{ f.counter = 0; f.b0 = f.tb0; f.b1 = f.tb1; f.b2 = f.tb2; f.a1 = f.ta1; f.a2 = f.ta2; }
__device__ __forceinline__ void BiquadClear(BiquadState &f)       // biquad.h:152-157
{ f.z1 = f.z2 = 0.0f; BiquadSnap(f); }
__device__ __forceinline__ void BiquadLerp(BiquadState &f, float a)
{
    f.b0 = lerpf(f.b0, f.tb0, a); f.b1 = lerpf(f.b1, f.tb1, a); f.b2 = lerpf(f.b2, f.tb2, a);
    f.a1 = lerpf(f.a1, f.ta1, a); f.a2 = lerpf(f.a2, f.ta2, a);
}

// check_set + BiquadInterpFilter::setParams state machine, biquad.cpp:38-43,131-149
__device__ __forceinline__ void BiquadSetTarget(BiquadState &f, const float c[5])
{
    bool changed = !(fabsf(c[0] - f.tb0) <= 0.015625f);
    changed |= !(fabsf(c[1] - f.tb1) <= 0.015625f);
    changed |= !(fabsf(c[2] - f.tb2) <= 0.015625f);
    changed |= !(fabsf(c[3] - f.ta1) <= 0.015625f);
    changed |= !(fabsf(c[4] - f.ta2) <= 0.015625f);
    f.tb0 = c[0]; f.tb1 = c[1]; f.tb2 = c[2]; f.ta1 = c[3]; f.ta2 = c[4];
    if(!changed) { if(f.counter <= 0) BiquadSnap(f); }
    else if(f.counter >= 0) f.counter = 256;
    else BiquadSnap(f);
}

// BiquadFilter::dualProcess, biquad.cpp:254-282 (transposed direct form II, two in cascade).
template<typename SrcPtr, typename DstPtr>
__device__ __forceinline__ void BiquadDualRaw(BiquadState &f0, BiquadState &f1, SrcPtr src, DstPtr dst, uint32_t n)
{
    float z01 = f0.z1, z02 = f0.z2, z11 = f1.z1, z12 = f1.z2;
    const float b00 = f0.b0, b01 = f0.b1, b02 = f0.b2, a01 = f0.a1, a02 = f0.a2;
    const float b10 = f1.b0, b11 = f1.b1, b12 = f1.b2, a11 = f1.a1, a12 = f1.a2;
    for(uint32_t i = 0; i < n; ++i)
    {
        const float x0 = src[i];
        const float y0 = x0 * b00 + z01;
        z01 = x0 * b01 - y0 * a01 + z02;
        z02 = x0 * b02 - y0 * a02;
        const float y1 = y0 * b10 + z11;
        z11 = y0 * b11 - y1 * a11 + z12;
        z12 = y0 * b12 - y1 * a12;
        dst[i] = y1;
    }
    f0.z1 = z01; f0.z2 = z02; f1.z1 = z11; f1.z2 = z12;
}

// BiquadInterpFilter::dualProcess, biquad.cpp:284-343: coefficients step toward the target
// every 32 samples over 8 steps while the counter runs.
template<typename SrcPtr, typename DstPtr>
__device__ __forceinline__ void BiquadDualInterp(BiquadState &f0, BiquadState &f1, SrcPtr src, DstPtr dst, uint32_t n)
{
    const int maxcounter = f0.counter > f1.counter ? f0.counter : f1.counter;
    uint32_t off = 0;
    if(maxcounter > 0)
    {
        int counter = maxcounter / 32;
        uint32_t steprem = uint32_t(32 - (maxcounter & 31));
        while(counter > 0)
        {
            const uint32_t td = steprem < (n - off) ? steprem : (n - off);
            BiquadDualRaw(f0, f1, src + off, dst + off, td);
            steprem -= td;
            if(steprem)
            {
                steprem = 32 - steprem;
                f0.counter = (counter * 32) | int(steprem);
                f1.counter = f0.counter;
                return;
            }
            off += td;
            steprem = 32;
            --counter;
            if(!counter)
            {
                BiquadSnap(f0); BiquadSnap(f1);
                break;
            }
            const float a = 1.0f / float(counter + 1);
            BiquadLerp(f0, a);
            BiquadLerp(f1, a);
            if(off == n)
            {
                f0.counter = counter * 32;
                f1.counter = f0.counter;
                return;
            }
        }
    }
    BiquadDualRaw(f0, f1, src + off, dst + off, n - off);
}

// ---- BandSplitter::processHfScale(input, output, hfscale), splitter.cpp:65-97 -------------
struct SplitterState { float coeff, lpZ1, lpZ2, apZ1; };
static_assert(sizeof(SplitterState) == sizeof(oalgpu_splitter), "layout");

template<typename SrcPtr, typename DstPtr>
__device__ __forceinline__ void SplitterHfScale(SplitterState &s, SrcPtr in, DstPtr out, uint32_t n, float hfscale)
{
    const float apCoeff = s.coeff;
    const float lpCoeff = s.coeff * 0.5f + 0.5f;
    float lpZ1 = s.lpZ1, lpZ2 = s.lpZ2, apZ1 = s.apZ1;
    for(uint32_t i = 0; i < n; ++i)
    {
        const float x = in[i];
        const float d0 = (x - lpZ1) * lpCoeff;
        const float lpY0 = lpZ1 + d0;
        lpZ1 = lpY0 + d0 * lpCoeff;          // sic: splitter.cpp:79 (this variant only)
        const float d1 = (lpY0 - lpZ2) * lpCoeff;
        const float lpY1 = lpZ2 + d1;
        lpZ2 = lpY1 + d1;
        const float apY = x * apCoeff + apZ1;
        apZ1 = x - apY * apCoeff;
        out[i] = (apY - lpY1) * hfscale + lpY1;
    }
    s.lpZ1 = lpZ1; s.lpZ2 = lpZ2; s.apZ1 = apZ1;
}

// ---- BandSplitter::processScale(samples, hfscale, lfscale), splitter.cpp:133-161, in place ---
template<typename Ptr>
__device__ __forceinline__ void SplitterScale(SplitterState &s, Ptr buf, uint32_t n, float hfscale, float lfscale)
{
    const float apCoeff = s.coeff;
    const float lpCoeff = s.coeff * 0.5f + 0.5f;
    float lpZ1 = s.lpZ1, lpZ2 = s.lpZ2, apZ1 = s.apZ1;
    for(uint32_t i = 0; i < n; ++i)
    {
        const float x = buf[i];
        const float d0 = (x - lpZ1) * lpCoeff;
        const float lpY0 = lpZ1 + d0;
        lpZ1 = lpY0 + d0;
        const float d1 = (lpY0 - lpZ2) * lpCoeff;
        const float lpY1 = lpZ2 + d1;
        lpZ2 = lpY1 + d1;
        const float apY = x * apCoeff + apZ1;
        apZ1 = x - apY * apCoeff;
        buf[i] = (apY - lpY1) * hfscale + lpY1 * lfscale;
    }
    s.lpZ1 = lpZ1; s.lpZ2 = lpZ2; s.apZ1 = apZ1;
}

} // namespace oalgpu
