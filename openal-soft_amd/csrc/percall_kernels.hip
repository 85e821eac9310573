// Per-call kernels: one launch per reference kernel call (Resample_*, Mix_*, MixHrtf_*,
// MixHrtfBlend_*, MixDirectHrtf_*, dual biquad, getCoeffs) over device buffers.  These are the
// single-voice building blocks the parity tests pin bit-exact (EXACT) against the oracle; the
// batched path (voice_kernel.hip) reuses the same device functions.
#include "kernels.hpp"

#include "dev_hrtf.hpp"
#include "dev_mix.hpp"
#include "dev_resample.hpp"

#pragma clang fp contract(off)

namespace oalgpu {

template<bool EXACT>
__global__ void __launch_bounds__(256) ResampleKernel(ResampleSpec spec, const float *__restrict__ src,
    uint32_t frac, uint32_t increment, float *__restrict__ dst, uint32_t n)
{
    for(uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        dst[i] = ResampleAt<EXACT>(spec.kind, spec.m, spec.l, spec.sf, spec.filter, ReferenceTabLayout(spec.m), src, frac, increment, i, n);
}

__global__ void __launch_bounds__(256) MixKernel(const float *__restrict__ in, uint32_t n, float *__restrict__ out,
    uint32_t nlines, float *__restrict__ cur, const float *__restrict__ tgt, uint32_t counter, uint32_t outpos)
{
    for(uint32_t c = 0; c < nlines; ++c)
    {
        const MixLineGain g = PrepareMixLine(cur[c], tgt[c], counter, n);
        for(uint32_t p = threadIdx.x; p < n; p += blockDim.x)
        {
            if(MixLineActive(g, p))
            {
                float *o = out + size_t{c} * kLine + outpos + p;
                *o = *o + MixLineValue(g, in[p], p);
            }
        }
        __syncthreads();
        if(threadIdx.x == 0) cur[c] = g.newCur;
    }
}

// blend != 0: MixHrtfBlendBase (old filter fading out, then new filter fading in from i=1).
template<bool EXACT>
__global__ void __launch_bounds__(256) MixHrtfKernel(const float *__restrict__ in, float *__restrict__ accum,
    uint32_t irsize, const float *__restrict__ coeffs, uint32_t dL, uint32_t dR, float gain, float step,
    const float *__restrict__ oldcoeffs, uint32_t odL, uint32_t odR, float oldgain, int blend, uint32_t n)
{
    const uint32_t ir = EXACT ? ((irsize + 1u) & ~1u) : irsize;
    const uint32_t nout = n + ir - 1;
    for(uint32_t o = threadIdx.x; o < nout; o += blockDim.x)
    {
        float l = accum[o * 2 + 0], r = accum[o * 2 + 1];
        if(!blend)
            HrtfGatherFrame<EXACT, kGainRamp>(l, r, in, coeffs, ir, dL, dR, gain, step, n, o);
        else
        {
            if(oldgain > kGainSilence)
                HrtfGatherFrame<EXACT, kGainOldFade>(l, r, in, oldcoeffs, ir, odL, odR, 0.0f, oldgain / float(n), n, o);
            if(step * float(n) > kGainSilence)
                HrtfGatherFrame<EXACT, kGainNewFade>(l, r, in, coeffs, ir, dL, dR, 0.0f, step, n, o);
        }
        accum[o * 2 + 0] = l;
        accum[o * 2 + 1] = r;
    }
}

// MixDirectHrtfBase, core/mixer/hrtfbase.h:91-133.  One workgroup; `temp` = nch x 1024 scratch.
template<bool EXACT>
__global__ void __launch_bounds__(256) MixDirectHrtfKernel(float *__restrict__ left, float *__restrict__ right,
    const float *__restrict__ in, uint32_t nch, float *__restrict__ accum, SplitterState *__restrict__ splitters,
    const float *__restrict__ hfscales, const float *__restrict__ chanCoeffs, uint32_t irsize, uint32_t n,
    float *__restrict__ temp)
{
    const uint32_t ir = EXACT ? ((irsize + 1u) & ~1u) : irsize;
    if(threadIdx.x < nch)
    {
        SplitterState s = splitters[threadIdx.x];
        SplitterHfScale(s, in + size_t{threadIdx.x} * kLine, temp + size_t{threadIdx.x} * kLine, n, hfscales[threadIdx.x]);
        splitters[threadIdx.x] = s;
    }
    __syncthreads();
    const uint32_t nout = n + ir - 1;
    for(uint32_t o = threadIdx.x; o < kLine + kHrirLen; o += blockDim.x)
    {
        float l = accum[o * 2 + 0], r = accum[o * 2 + 1];
        if(o < nout)
        {
            for(uint32_t c = 0; c < nch; ++c)
            {
                const float *t = temp + size_t{c} * kLine;
                const float *co = chanCoeffs + size_t{c} * kHrirLen * 2;
                const uint32_t lo = (o + 1 > ir) ? o + 1 - ir : 0u;
                const uint32_t hi = o < n - 1 ? o : n - 1;
                for(uint32_t i = lo; i <= hi; ++i)
                {
                    const float x = t[i];
                    l = madd<EXACT>(l, co[(o - i) * 2 + 0], x);
                    r = madd<EXACT>(r, co[(o - i) * 2 + 1], x);
                }
            }
        }
        if(o < n)
        {
            left[o] = left[o] + l;
            right[o] = right[o] + r;
        }
        // keep (l, r) for the shift below
        temp[size_t{nch} * kLine + o * 2 + 0] = l;
        temp[size_t{nch} * kLine + o * 2 + 1] = r;
    }
    __syncthreads();
    // hrtfbase.h:127-132: frames [n, n+128) move to the front, the following n frames are cleared
    for(uint32_t o = threadIdx.x; o < kLine + kHrirLen; o += blockDim.x)
    {
        float l = 0.0f, r = 0.0f;
        if(o < kHrirLen)
        {
            l = temp[size_t{nch} * kLine + (o + n) * 2 + 0];
            r = temp[size_t{nch} * kLine + (o + n) * 2 + 1];
        }
        else if(o >= kHrirLen + n)
        {
            l = temp[size_t{nch} * kLine + o * 2 + 0];
            r = temp[size_t{nch} * kLine + o * 2 + 1];
        }
        accum[o * 2 + 0] = l;
        accum[o * 2 + 1] = r;
    }
}

__global__ void BiquadDualKernel(BiquadState *__restrict__ f0, BiquadState *__restrict__ f1,
    const float *__restrict__ src, float *__restrict__ dst, uint32_t n)
{
    if(threadIdx.x != 0 || blockIdx.x != 0) return;
    BiquadState a = *f0, b = *f1;
    BiquadDualInterp(a, b, src, dst, n);
    *f0 = a; *f1 = b;
}

// one workgroup of 256 threads per direction: element e of the blended HrirArray per thread
__global__ void __launch_bounds__(256) GetCoeffsKernel(HrtfStoreDev st, const float *__restrict__ dirs,
    float *__restrict__ coeffs, uint32_t *__restrict__ delays)
{
    const uint32_t d = blockIdx.x;
    const HrirBlend b = HrtfBlendFor(st, dirs[d * 4 + 0], dirs[d * 4 + 1], dirs[d * 4 + 2], dirs[d * 4 + 3]);
    coeffs[size_t{d} * (kHrirLen * 2) + threadIdx.x] = HrtfBlendElement(st, b, threadIdx.x);
    if(threadIdx.x < 2) delays[d * 2 + threadIdx.x] = b.delay[threadIdx.x];
}

// ---- launchers ----------------------------------------------------------------------------
void LaunchResample(hipStream_t s, bool exact, const ResampleSpec &spec, const float *src, uint32_t frac,
    uint32_t increment, float *dst, uint32_t n)
{
    const uint32_t blocks = (n + 255u) / 256u;
    if(exact) hipLaunchKernelGGL(ResampleKernel<true>, dim3(blocks), dim3(256), 0, s, spec, src, frac, increment, dst, n);
    else hipLaunchKernelGGL(ResampleKernel<false>, dim3(blocks), dim3(256), 0, s, spec, src, frac, increment, dst, n);
}

void LaunchMix(hipStream_t s, const float *in, uint32_t n, float *out, uint32_t nlines, float *cur,
    const float *tgt, uint32_t counter, uint32_t outpos)
{ hipLaunchKernelGGL(MixKernel, dim3(1), dim3(256), 0, s, in, n, out, nlines, cur, tgt, counter, outpos); }

void LaunchMixHrtf(hipStream_t s, bool exact, const float *in, float *accum, uint32_t irsize, const float *coeffs,
    uint32_t dL, uint32_t dR, float gain, float step, const float *oldcoeffs, uint32_t odL, uint32_t odR,
    float oldgain, int blend, uint32_t n)
{
    if(exact) hipLaunchKernelGGL(MixHrtfKernel<true>, dim3(1), dim3(256), 0, s, in, accum, irsize, coeffs, dL, dR, gain,
        step, oldcoeffs, odL, odR, oldgain, blend, n);
    else hipLaunchKernelGGL(MixHrtfKernel<false>, dim3(1), dim3(256), 0, s, in, accum, irsize, coeffs, dL, dR, gain,
        step, oldcoeffs, odL, odR, oldgain, blend, n);
}

void LaunchMixDirectHrtf(hipStream_t s, bool exact, float *left, float *right, const float *in, uint32_t nch,
    float *accum, SplitterState *splitters, const float *hfscales, const float *chanCoeffs, uint32_t irsize,
    uint32_t n, float *temp)
{
    if(exact) hipLaunchKernelGGL(MixDirectHrtfKernel<true>, dim3(1), dim3(256), 0, s, left, right, in, nch, accum,
        splitters, hfscales, chanCoeffs, irsize, n, temp);
    else hipLaunchKernelGGL(MixDirectHrtfKernel<false>, dim3(1), dim3(256), 0, s, left, right, in, nch, accum,
        splitters, hfscales, chanCoeffs, irsize, n, temp);
}

void LaunchBiquadDual(hipStream_t s, BiquadState *f0, BiquadState *f1, const float *src, float *dst, uint32_t n)
{ hipLaunchKernelGGL(BiquadDualKernel, dim3(1), dim3(64), 0, s, f0, f1, src, dst, n); }

void LaunchGetCoeffs(hipStream_t s, const HrtfStoreDev &st, const float *dirs, uint32_t count, float *coeffs,
    uint32_t *delays)
{ hipLaunchKernelGGL(GetCoeffsKernel, dim3(count), dim3(256), 0, s, st, dirs, coeffs, delays); }

} // namespace oalgpu
