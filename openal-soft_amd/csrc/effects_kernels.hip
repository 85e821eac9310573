// The small EffectStates behind the effect-slot boundary, one workgroup per instance:
//
//   EqualizerState::process   alc/effects/equalizer.cpp:167-187   per wet channel: two DualBiquads, MixSamples 1 -> 1
//   ModulatorState::process   alc/effects/modulator.cpp:165-209   per wet channel: high-pass, x carrier, MixSamples 1 -> 1
//   EchoState::process        alc/effects/echo.cpp:119-163        two-tap feedback delay with a damping shelf, MixSamples 1 -> N
//   DedicatedState::process   alc/effects/dedicated.cpp:102-108   MixSamples 1 -> N
//
// (update() -- coefficient design, delays, the carrier's period -- is host work, effects_api.hip; what it derives
// from the ambisonic layer, the target channel / gain per wet channel and the panned gains, comes from the caller.)
// Like the EAX reverb, these keep the reference's operation order in every mode: the recurrences run one wavefront per
// channel (four channels side by side, WaveSerial of effects_dev.hpp), everything around them over the workgroup -- the output is bit-identical
// to the reference's, except the sinusoid carrier (the GPU's sinf against libm).  A block-scan form of the
// biquads was built first and dropped: the equalizer's high-Q peaking sections and the modulator's 200 Hz
// high-pass amplify float32 rounding so much (the reference's own serial loop is 7e-6 of the block maximum away
// from a double-precision run) that a reordered evaluation lands at 1e-4 .. 1e-3, which is no parity statement.
#include "effects_dev.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

__device__ __forceinline__ float Carrier(int wave, uint32_t index, float scale)
{   // SinFunc / SawFunc / SquareFunc / OneFunc, modulator.cpp:49-69
    switch(wave)
    {
    case 1: return sinf(float(index) * scale);
    case 2: return float(index) * scale - 1.0f;
    case 3: return float(float(index) * scale < 0.5f) * 2.0f - 1.0f;
    default: return 1.0f;
    }
}

__global__ void __launch_bounds__(256) EffectKernel(FxLaunch F)
{
    __shared__ float buf[4][kLine];
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t n = F.n;
    FxState &S = *F.st;

    if(F.kind == OALGPU_EFFECT_EQUALIZER || F.kind == OALGPU_EFFECT_MODULATOR)
    {
        const bool mod = F.kind == OALGPU_EFFECT_MODULATOR;
        for(uint32_t c0 = 0; c0 < F.numIn; c0 += 4u)
        {
            const uint32_t c = c0 + wave;
            const bool mine = c < F.numIn && F.target[c] < F.nlines;
            if(mine)
            {
                const float *in = F.wetIn + size_t{c} * kLine;
                for(uint32_t i = lane; i < n; i += 64u) buf[wave][i] = in[i];
                WaveSync();
                BiquadState f0 = S.bq[c][0], f1 = S.bq[c][1], f2 = S.bq[c][2], f3 = S.bq[c][3];
                // the recurrences: in the reference's operation order, one sample after the other (WaveSerial keeps the
                // LDS reads off the dependent chain); the equalizer's two DualBiquad passes run sample-interleaved,
                // which gives every filter the same input sequence
                auto one = [](BiquadState &f, float x) {
                    const float y = x * f.b0 + f.z1;
                    f.z1 = x * f.b1 - y * f.a1 + f.z2;
                    f.z2 = x * f.b2 - y * f.a2;
                    return y;
                };
                float *row = buf[wave];
                if(mod) WaveSerial<1>(n, lane, row, [&](uint32_t i, float *v) { v[0] = row[i]; }, [&](const float *x) { return one(f0, x[0]); });
                else
                    WaveSerial<1>(n, lane, row, [&](uint32_t i, float *v) { v[0] = row[i]; },
                        [&](const float *x) { return one(f3, one(f2, one(f1, one(f0, x[0])))); });
                if(lane == 0)
                {
                    S.bq[c][0].z1 = f0.z1; S.bq[c][0].z2 = f0.z2;
                    if(!mod)
                    {
                        S.bq[c][1].z1 = f1.z1; S.bq[c][1].z2 = f1.z2; S.bq[c][2].z1 = f2.z1; S.bq[c][2].z2 = f2.z2;
                        S.bq[c][3].z1 = f3.z1; S.bq[c][3].z2 = f3.z2;
                    }
                }
                WaveSync();
                if(mod)
                    for(uint32_t i = lane; i < n; i += 64u)
                        buf[wave][i] = buf[wave][i] * Carrier(F.modWave, (F.modIndex + i) % F.modRange, F.modScale);
            }
            __syncthreads();
            // MixSamples in channel order (two channels may share a target line)
            for(uint32_t w = 0; w < 4u; ++w)
            {
                const uint32_t cc = c0 + w;
                if(cc >= F.numIn || F.target[cc] >= F.nlines) continue;
                const uint32_t counter = mod ? (n < 64u ? n : 64u) : n;
                const MixLineGain g = PrepareMixLine(S.cur[cc], F.tgtGain[cc], counter, n);
                float *out = F.outLines + size_t{F.target[cc]} * kLine;
                for(uint32_t p = t; p < n; p += 256u)
                    if(MixLineActive(g, p)) out[p] = out[p] + MixLineValue(g, buf[w][p], p);
                __syncthreads();
                if(t == 0) S.cur[cc] = g.newCur;
                __syncthreads();
            }
        }
        return;
    }
    if(F.kind == OALGPU_EFFECT_COMPRESSOR)
    {   // CompressorState::process, compressor.cpp:115-181: gains[i] = 1 / envelope(i) from wet channel 0, then
        // out[target c][i] += in[c][i] * gains[i] * gain (no ramp; silent gains skipped)
        for(uint32_t i = t; i < n; i += 256u) buf[1][i] = F.wetIn[i];
        __syncthreads();
        if(wave == 0)
        {
            float env = S.env;
            const float *src = buf[1];
            WaveSerial<1>(n, lane, buf[0],
                [&](uint32_t i, float *v) { v[0] = F.compOn ? fminf(fmaxf(fabsf(src[i]), 0.5f), 2.0f) : 1.0f; },
                [&](const float *x) {
                    const float amplitude = x[0];
                    if(amplitude > env) env = fminf(env * F.attackMult, amplitude);
                    else if(amplitude < env) env = fmaxf(env * F.releaseMult, amplitude);
                    return 1.0f / env;
                });
            if(lane == 0) S.env = env;
        }
        __syncthreads();
        for(uint32_t c = 0; c < F.numIn; ++c)
        {
            if(F.target[c] >= F.nlines || !(fabsf(F.tgtGain[c]) > 0.00001f)) continue;
            const float *in = F.wetIn + size_t{c} * kLine;
            float *out = F.outLines + size_t{F.target[c]} * kLine;
            for(uint32_t i = t; i < n; i += 256u) out[i] = out[i] + in[i] * buf[0][i] * F.tgtGain[c];
            __syncthreads();
        }
        return;
    }
    if(F.kind == OALGPU_EFFECT_DEDICATED)
    {
        for(uint32_t i = t; i < n; i += 256u) buf[0][i] = F.wetIn[i];
        __syncthreads();
        MixOntoLines(buf[0], F.outLines, F.nlines, S.cur, F.tgtGains, n, n, t);
        return;
    }
    // ---- echo: buf[0] = first tap, buf[1] = second tap, buf[2] = what goes into the delay line, buf[3] = input
    const uint32_t mask = F.delayMask, d1 = F.tap[0], d2 = F.tap[1], off = F.offset;
    for(uint32_t i = t; i < n; i += 256u)
    {
        buf[3][i] = F.wetIn[i];
        // the taps as far as they lie before this block; the rest is written by the block itself
        buf[0][i] = F.delay[(off + i - d1) & mask];
        buf[1][i] = F.delay[(off + i - d2) & mask];
    }
    __syncthreads();
    if(t == 0)
    {   // delaybuf[offset] = in; out1 = delaybuf[tap1]; out2 = delaybuf[tap2]; delaybuf[offset] += filter(out2) * feed
        BiquadState f = S.bq[0][0];
        float z1 = f.z1, z2 = f.z2;
        for(uint32_t i = 0; i < n; ++i)
        {
            // a tap shorter than the block reads what an earlier iteration stored; a tap of exactly 0 distance
            // cannot occur (mDelayTap[0] >= 1)
            if(i >= d1) buf[0][i] = buf[2][i - d1];
            if(i >= d2) buf[1][i] = buf[2][i - d2];
            const float x = buf[1][i];
            const float y = x * f.b0 + z1;                       // BiquadFilter::processOne
            z1 = x * f.b1 - y * f.a1 + z2;
            z2 = x * f.b2 - y * f.a2;
            buf[2][i] = buf[3][i] + y * F.feedGain;
        }
        S.bq[0][0].z1 = z1; S.bq[0][0].z2 = z2;
    }
    __syncthreads();
    for(uint32_t i = t; i < n; i += 256u) F.delay[(off + i) & mask] = buf[2][i];
    MixOntoLines(buf[0], F.outLines, F.nlines, S.cur, F.tgtGains, n, n, t);
    MixOntoLines(buf[1], F.outLines, F.nlines, S.cur + 32, F.tgtGains + F.nlines, n, n, t);
}

} // namespace

void LaunchEffect(hipStream_t s, const FxLaunch &F) { hipLaunchKernelGGL(EffectKernel, dim3(1), dim3(256), 0, s, F); }

} // namespace oalgpu
