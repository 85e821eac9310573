// Device helpers the EffectState kernels share (effects_kernels.hip, effects2_kernels.hip).
#pragma once
#include "wave_common.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

// BiquadFilter::process, core/filters/biquad.cpp:176-201
template<typename SrcPtr, typename DstPtr>
__device__ __forceinline__ void BiquadRaw(BiquadState &f, SrcPtr src, DstPtr dst, uint32_t n)
{
    float z1 = f.z1, z2 = f.z2;
    const float b0 = f.b0, b1 = f.b1, b2 = f.b2, a1 = f.a1, a2 = f.a2;
    for(uint32_t i = 0; i < n; ++i)
    {
        const float x = src[i];
        const float y = x * b0 + z1;
        z1 = x * b1 - y * a1 + z2;
        z2 = x * b2 - y * a2;
        dst[i] = y;
    }
    f.z1 = z1; f.z2 = z2;
}

// MixSamples(src, out[c], Current, Target, counter) for c < nlines, thread = frames t, t + 256, ..
__device__ __forceinline__ void MixOntoLines(const float *src, float *out, uint32_t nlines, float *cur, const float *tgt,
    uint32_t counter, uint32_t n, uint32_t t)
{
    for(uint32_t c = 0; c < nlines; ++c)
    {
        const MixLineGain g = PrepareMixLine(cur[c], tgt[c], counter, n);
        for(uint32_t p = t; p < n; p += 256u)
            if(MixLineActive(g, p)) { float *o = out + size_t{c} * kLine + p; *o = *o + MixLineValue(g, src[p], p); }
    }
    __syncthreads();
    for(uint32_t c = t; c < nlines; c += 256u) cur[c] = PrepareMixLine(cur[c], tgt[c], counter, n).newCur;
    __syncthreads();
}

// y[i] = step(x[i]) for i = 0 .. n-1 IN ORDER, on one wavefront: a recurrence whose state lives in `step`.  Each round
// every lane fetches the NV inputs of one sample (load(i, v): anything that does not depend on the recurrence -- LDS
// or global reads, per-sample coefficient math -- happens here, 64 samples at once), the samples then go through
// step() one after the other, their inputs broadcast with v_readlane; step() runs uniformly on all lanes, so no
// memory latency sits on the recurrence's dependent chain.  dst (LDS) may alias what load() reads.
template<int NV, typename Load, typename Step>
__device__ __forceinline__ void WaveSerial(uint32_t n, uint32_t lane, float *dst, Load &&load, Step &&step)
{
    for(uint32_t base = 0; base < n; base += 64u)
    {
        const uint32_t i = base + lane;
        float v[NV];
        load(i < n ? i : n - 1u, v);
        const uint32_t cnt = __builtin_amdgcn_readfirstlane((n - base) < 64u ? (n - base) : 64u);
        // every lane holds the same y: all of them store it to the one address (one LDS instruction, no VALU select)
        if(cnt == 64u)
        {
#pragma unroll
            for(int k = 0; k < 64; ++k)
            {
                float x[NV];
#pragma unroll
                for(int q = 0; q < NV; ++q) x[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[q]), k));
                dst[base + uint32_t(k)] = step(x);
            }
        }
        else
        {
            for(uint32_t k = 0; k < cnt; ++k)
            {
                float x[NV];
#pragma unroll
                for(int q = 0; q < NV; ++q)
                    x[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[q]), int(__builtin_amdgcn_readfirstlane(k))));
                dst[base + k] = step(x);
            }
        }
    }
}

} // namespace
} // namespace oalgpu
