// Device-side arithmetic helpers shared by every kernel.
//
// The reference mixer runs IEEE binary32 with FTZ+DAZ and *no* FMA contraction
// (core/fpu_ctrl.cpp:32-50; CMake builds without -mfma; vmadd = add(mul) in
// core/mixer/mixer_sse.cpp:37-38).  This library is compiled with -ffp-contract=off and
// -fgpu-flush-denormals-to-zero, so plain `a*b + c` below is two rounded operations exactly
// like the reference; a fused multiply-add only happens where fmaf() is written out, which the
// kernels do in their FAST instantiations only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

namespace oalgpu {

constexpr int kLine = 1024;            // BufferLineSize
constexpr int kMaxPad = 48;            // MaxResamplerPadding
constexpr int kMaxEdge = 24;           // MaxResamplerEdge
constexpr int kHist = 64;              // HrtfHistoryLength
constexpr int kHrirLen = 128;          // HrirLength
constexpr int kFracBits = 16;
constexpr uint32_t kFracOne = 1u << kFracBits;
constexpr uint32_t kFracMask = kFracOne - 1u;
constexpr float kGainSilence = 0.00001f;            // GainSilenceThreshold, core/mixer/defs.h:28
constexpr float kFltEpsilon = 1.1920928955078125e-07f;
constexpr int kResampleDataSize = kLine + 256 + kMaxPad;   // DeviceBase::mResampleData, device.h:282

// a + b*c the way the selected math mode wants it.
template<bool EXACT>
__device__ __forceinline__ float madd(float a, float b, float c)
{
    if constexpr(EXACT) return a + b * c;
    else return __builtin_fmaf(b, c, a);
}

__device__ __forceinline__ float lerpf(float a, float b, float mu) { return a + (b - a) * mu; }

// float2uint, common/alnumeric.h:223-240: truncation with clamping.  (Host-callable: the host evaluates
// HrtfStore::getCoeffs' index arithmetic for parameter blocks with the very same operations.)
__host__ __device__ __forceinline__ uint32_t float2uint(float f)
{
    const int32_t bits = __builtin_bit_cast(int32_t, f);
    const uint32_t keep = static_cast<uint32_t>(bits >> 31) ^ 0xffffffffu;
    const int shift = ((bits >> 23) & 0xff) - (127 + 23);
    if(shift < -23) return 0u;
    if(shift > 8) return 0xffffffffu & keep;
    const uint32_t mant = (static_cast<uint32_t>(bits) & 0x7fffffu) | 0x800000u;
    return ((shift < 0) ? (mant >> -shift) : (mant << shift)) & keep;
}

// fastf2u, common/alnumeric.h:163-189: cvtss2si = round to nearest even.
__host__ __device__ __forceinline__ uint32_t fastf2u(float f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return static_cast<uint32_t>(__float2int_rn(f));
#else
    return static_cast<uint32_t>(static_cast<int32_t>(__builtin_lrintf(f)));    // the default rounding mode: to nearest even
#endif
}

} // namespace oalgpu
