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

// Layout of the coefficient rows the kernels read.  In the reference table (HBM) phase row pi
// of a bsinc filter is [fil(m) | phd(m)] at 2*m*pi and the scale deltas [scd(m) | spd(m)] sit
// 64*m floats further on (core/bsinc_tables.cpp:255-339); the voice kernel re-stages the rows
// in LDS with a padded stride (16-byte aligned, bank-spread) and reads them as float4.
struct TabLayout {
    uint32_t bsincStride;   // floats between phase rows
    uint32_t scdBase;       // offset of the scale-delta block
    uint32_t cubicStride;   // floats between cubic phase rows ({coeffs[4], deltas[4]})
};
__host__ __device__ inline TabLayout ReferenceTabLayout(uint32_t m) { return TabLayout{2u * m, 64u * m, 8u}; }

// src: the reference's mResampleData (index 0 = MaxResamplerEdge samples before the position).
// i: output index inside this call; n: the call's dst size (cubic tail rule).
// WIDE: coefficient rows are 16-byte aligned (LDS staging) and are read four taps at a time.
template<bool EXACT, bool WIDE = false, typename SrcPtr, typename TabPtr>
__device__ __forceinline__ float ResampleAt(int kind, uint32_t m, uint32_t l, float sf, TabPtr filter,
    const TabLayout lay, SrcPtr src, uint32_t frac0, uint32_t increment, uint32_t i, uint32_t n)
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
            const uint32_t row = pi * lay.cubicStride;
            float r[4];
#pragma unroll
            for(int k = 0; k < 4; ++k)
            {
                const float f = madd<EXACT>(filter[row + k], pf, filter[row + 4 + k]);
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
            const uint32_t fil = pi * lay.bsincStride, phd = fil + m;
            const uint32_t scd = fil + lay.scdBase, spd = scd + m;
            float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f, r3 = 0.0f;
            if constexpr(WIDE)
            {
                if(kind == 3)
                {
                    for(uint32_t j = 0; j < m; j += 4)
                    {
                        const float4 f = *reinterpret_cast<const float4*>(&filter[fil + j]);
                        const float4 p = *reinterpret_cast<const float4*>(&filter[phd + j]);
                        r0 = madd<EXACT>(r0, madd<EXACT>(f.x, pf, p.x), src[base + j + 0]);
                        r1 = madd<EXACT>(r1, madd<EXACT>(f.y, pf, p.y), src[base + j + 1]);
                        r2 = madd<EXACT>(r2, madd<EXACT>(f.z, pf, p.z), src[base + j + 2]);
                        r3 = madd<EXACT>(r3, madd<EXACT>(f.w, pf, p.w), src[base + j + 3]);
                    }
                }
                else
                {
                    for(uint32_t j = 0; j < m; j += 4)
                    {
                        const float4 f = *reinterpret_cast<const float4*>(&filter[fil + j]);
                        const float4 p = *reinterpret_cast<const float4*>(&filter[phd + j]);
                        const float4 sc = *reinterpret_cast<const float4*>(&filter[scd + j]);
                        const float4 sp = *reinterpret_cast<const float4*>(&filter[spd + j]);
                        r0 = madd<EXACT>(r0, madd<EXACT>(madd<EXACT>(f.x, sf, sc.x), pf, madd<EXACT>(p.x, sf, sp.x)), src[base + j + 0]);
                        r1 = madd<EXACT>(r1, madd<EXACT>(madd<EXACT>(f.y, sf, sc.y), pf, madd<EXACT>(p.y, sf, sp.y)), src[base + j + 1]);
                        r2 = madd<EXACT>(r2, madd<EXACT>(madd<EXACT>(f.z, sf, sc.z), pf, madd<EXACT>(p.z, sf, sp.z)), src[base + j + 2]);
                        r3 = madd<EXACT>(r3, madd<EXACT>(madd<EXACT>(f.w, sf, sc.w), pf, madd<EXACT>(p.w, sf, sp.w)), src[base + j + 3]);
                    }
                }
            }
            else if(kind == 3)
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
