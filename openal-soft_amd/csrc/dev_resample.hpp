// Resampler inner loops (device functions shared by the per-call kernel and the fused voice
// kernel).  One call produces ONE output sample; lanes of a wavefront produce consecutive
// outputs, so the 16.16 stepping of the reference
//     frac += increment; pos += frac>>16; frac &= 0xffff      (core/mixer/mixer_c.cpp:116-118)
// is evaluated in closed form: t = frac0 + i*increment, pos = t>>16, frac = t&0xffff
// (no overflow: i < 1024, increment <= MaxPitch<<16, core/voice.cpp:54-58).
//
// Summation orders follow the x86 variants the reference auto-selects:
//   linear  : a + (b-a)*mu                      mixer_c.cpp:44-46 == mixer_sse41.cpp:84-87
//   cubic   : products then (r0+r1)+(r2+r3) in the 4-wide body, (r0+r3)+(r1+r2) in the
//             dst.size()&3 tail                 mixer_sse41.cpp:163-212
//   bsinc   : 4 lane-partial sums over j mod 4, reduced (l0+l3)+(l1+l2)
//                                               mixer_sse.cpp:232-329
#pragma once
#include "dev_math.hpp"

#pragma clang fp contract(off)

namespace oalgpu {

struct ResampleSpec {
    int32_t kind;          // 0 point, 1 linear, 2 cubic, 3 fast bsinc, 4 bsinc
    uint32_t m, l;         // bsinc taps / left offset
    float sf;              // bsinc scale interpolation factor
    const float *filter;   // cubic: float[32][8]; bsinc: BsincState::filter
};

// src: the reference's mResampleData (index 0 = MaxResamplerEdge samples before the position).
// i: output index inside this call; n: the call's dst size (cubic tail rule).
template<bool EXACT, typename SrcPtr, typename TabPtr>
__device__ __forceinline__ float ResampleAt(int kind, uint32_t m, uint32_t l, float sf, TabPtr filter,
    SrcPtr src, uint32_t frac0, uint32_t increment, uint32_t i, uint32_t n)
{
    const uint32_t t = frac0 + i * increment;
    const uint32_t pos = t >> kFracBits;
    const uint32_t frac = t & kFracMask;
    switch(kind)
    {
    case 0:
        return src[kMaxEdge + pos];
    case 1:
        return lerpf(src[kMaxEdge + pos], src[kMaxEdge + pos + 1], float(frac) * (1.0f / kFracOne));
    case 2:
        {
            const uint32_t pi = frac >> 11;
            const float pf = float(frac & 2047u) * (1.0f / 2048.0f);
            const uint32_t base = kMaxEdge - 1 + pos;
            float r[4];
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                const float f = madd<EXACT>(filter[pi * 8 + k], pf, filter[pi * 8 + 4 + k]);
                r[k] = f * src[base + k];
            }
            if(i < (n & ~3u)) return (r[0] + r[1]) + (r[2] + r[3]);
            return (r[0] + r[3]) + (r[1] + r[2]);
        }
    default:
        {
            const uint32_t pi = frac >> 11;
            const float pf = float(frac & 2047u) * (1.0f / 2048.0f);
            const uint32_t base = kMaxEdge - l + pos;
            const uint32_t fil = 2u * pi * m, phd = fil + m;
            const uint32_t scd = fil + 64u * m, spd = scd + m;   // BSincPhaseCount*2*m further on
            float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
            if(kind == 3)
            {
                for(uint32_t j = 0; j < m; j += 4)
                {
                    r0 = madd<EXACT>(r0, madd<EXACT>(filter[fil + j + 0], pf, filter[phd + j + 0]), src[base + j + 0]);
                    r1 = madd<EXACT>(r1, madd<EXACT>(filter[fil + j + 1], pf, filter[phd + j + 1]), src[base + j + 1]);
                    r2 = madd<EXACT>(r2, madd<EXACT>(filter[fil + j + 2], pf, filter[phd + j + 2]), src[base + j + 2]);
                    r3 = madd<EXACT>(r3, madd<EXACT>(filter[fil + j + 3], pf, filter[phd + j + 3]), src[base + j + 3]);
                }
            }
            else
            {
                for(uint32_t j = 0; j < m; j += 4)
                {
#define OALGPU_BSINC_TAP(R, K) { \
    const float fs = madd<EXACT>(filter[fil + j + K], sf, filter[scd + j + K]); \
    const float ps = madd<EXACT>(filter[phd + j + K], sf, filter[spd + j + K]); \
    R = madd<EXACT>(R, madd<EXACT>(fs, pf, ps), src[base + j + K]); }
                    OALGPU_BSINC_TAP(r0, 0) OALGPU_BSINC_TAP(r1, 1) OALGPU_BSINC_TAP(r2, 2) OALGPU_BSINC_TAP(r3, 3)
#undef OALGPU_BSINC_TAP
                }
            }
            return (r0 + r3) + (r1 + r2);
        }
    }
}

} // namespace oalgpu
