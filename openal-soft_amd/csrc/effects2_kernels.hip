// The remaining EffectStates of SURVEY.md 8f rank 4, one workgroup per instance (the frequency shifter: one per
// A-Format line for its Hilbert transforms):
//
//   ChorusState::process      alc/effects/chorus.cpp:320-427      B- to A-Format, four LFO-modulated feedback delay lines read
//                                                                 through the 256-step gaussian, A- to B-Format (also the flanger)
//   DistortionState::process  alc/effects/distortion.cpp:197-306  B- to A-Format, 4x zero-stuffed low-pass, wave shaper, band-pass, decimate
//   AutowahState::process     alc/effects/autowah.cpp:124-196     envelope follower on channel 0, a peaking filter that changes every sample
//   VmorpherState::process    alc/effects/vmorpher.cpp:279-335    two 4-band formant filters (state variable) blended by an LFO
//   FshifterState::process    alc/effects/fshifter.cpp:216-365    STFT (1024 / hop 256, double precision) -> analytic signal -> phase rotation
//   PshifterState::process    alc/effects/pshifter.cpp:213-460    STFT (1024 / hop 128) phase vocoder, up to second order (9 channels)
//
// As for the other effects the reference's operation order is kept in every mode.  Recurrences run through WaveSerial
// (effects_dev.hpp): their inputs are fetched 64 samples at a time, the dependent chain itself sees no memory.  Where
// the reference's loop is serial only in appearance it is done in parallel WITHOUT reordering any sum:
//   * the chorus' feedback `delaybuf[offset] += delaybuf[offset - avgdelay]*feedback` is a comb of avgdelay (0 .. 12)
//     independent chains, and its taps only read samples at least 23 behind the write position, i.e. values that are
//     final: the block's delay line is completed first, all taps are then read at once;
//   * the frequency shifter's FFT butterflies of one stage are independent; with the reference's twiddle factors
//     (the host builds them by the same repeated multiplication, effects_api.hip) the transform is bit-identical.
// Results are bit-identical to the reference's except where libm's sinf / cosf meet the GPU's (chorus sinusoid LFO,
// autowah, the morpher's sinusoid LFO): those go through double precision and agree to the last bit almost always.
#include "effects_dev.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr uint32_t kInvalid = 0xffffffffu;
constexpr float kB2ACoeff = static_cast<float>(0.25 / 1.7320508075688772935);   // DecodeCoeff, chorus.cpp:59
constexpr float kA2BCoeff = static_cast<float>(0.5 * 1.7320508075688772935);    // EncodeCoeff, :69
constexpr float kPiF = 3.14159265358979323846f;

__device__ __forceinline__ float SinViaDouble(float x) { return float(sin(double(x))); }
__device__ __forceinline__ float CosViaDouble(float x) { return float(cos(double(x))); }

// sign of B2A[c][i], which is also the sign of A2B[i][c] (chorus.cpp:60-75 / fshifter.cpp:61-76; distortion.cpp:50-65
// numbers its tetrahedron differently): bit i of the entry = negative
__device__ __forceinline__ bool AfNegative(bool distortion, uint32_t c, uint32_t i)
{
    const uint32_t chorus = (0x0u) | (0xcu << 4) | (0x6u << 8) | (0xau << 12);
    const uint32_t dist = (0x0u) | (0x6u << 4) | (0xcu << 8) | (0xau << 12);
    return (((distortion ? dist : chorus) >> (c * 4u + i)) & 1u) != 0u;
}
__device__ __forceinline__ float B2A(bool distortion, uint32_t c, uint32_t i)
{ return i == 0u ? 0.25f : (AfNegative(distortion, c, i) ? -kB2ACoeff : kB2ACoeff); }
__device__ __forceinline__ float A2B(bool distortion, uint32_t i, uint32_t c)
{ return i == 0u ? 1.0f : (AfNegative(distortion, c, i) ? -kA2BCoeff : kA2BCoeff); }

// mABuffer[c][p]: `sample + in*gain` over the inputs, from 0
__device__ __forceinline__ float AFormat(const Fx2Launch &F, bool distortion, uint32_t c, uint32_t p)
{
    const uint32_t numInput = F.numIn < 4u ? F.numIn : 4u;
    float a = 0.0f;
    for(uint32_t i = 0; i < numInput; ++i) a = a + F.wetIn[size_t{i} * kLine + p] * B2A(distortion, c, i);
    return a;
}

// The A-Format effects' tail: rows[c] = mBBuffer[c] -> the output lines, through the up-sampler of a device above
// first order (BandSplitter::processHfScale, the two-span form core/filters/splitter.cpp:65-97, then MixSamples
// onto every line) or onto the channel's own target line (chorus.cpp:393-427 and alike)
__device__ void FxOutput(const Fx2Launch &F, Fx2State &S, float (*rows)[kLine], uint32_t nchan, uint32_t t)
{
    const uint32_t lane = t & 63u, wave = __builtin_amdgcn_readfirstlane(t >> 6), n = F.n;
    if(F.upsample)
    {
        for(uint32_t c0 = 0; c0 < nchan; c0 += 4u)
        {
            const uint32_t c = c0 + wave;
            if(c >= nchan || F.target[c] == kInvalid) continue;
            const float apc = F.splitCoeff, lpc = F.splitCoeff * 0.5f + 0.5f, hf = F.hfScale[c ? 1 : 0];
            float z1 = S.split[c][0], z2 = S.split[c][1], az = S.split[c][2];
            float *row = rows[c];
            WaveSerial<1>(n, lane, row, [&](uint32_t i, float *v) { v[0] = row[i]; },
                [&](const float *x) {
                    const float in = x[0];
                    const float d0 = (in - z1) * lpc;
                    const float y0 = z1 + d0;
                    z1 = y0 + d0 * lpc;
                    const float d1 = (y0 - z2) * lpc;
                    const float y1 = z2 + d1;
                    z2 = y1 + d1;
                    const float ay = in * apc + az;
                    az = in - ay * apc;
                    return (ay - y1) * hf + y1;
                });
            if(lane == 0) { S.split[c][0] = z1; S.split[c][1] = z2; S.split[c][2] = az; }
        }
        __syncthreads();
        for(uint32_t c = 0; c < nchan; ++c)
            if(F.target[c] != kInvalid) MixOntoLines(rows[c], F.outLines, F.nlines, S.upCur[c], F.upTgt + c * 32u, n, n, t);
        return;
    }
    for(uint32_t c = 0; c < nchan; ++c)
    {
        if(F.target[c] >= F.nlines) continue;
        const MixLineGain g = PrepareMixLine(S.cur[c], F.tgtGain[c], n, n);
        float *out = F.outLines + size_t{F.target[c]} * kLine;
        for(uint32_t p = t; p < n; p += 256u)
            if(MixLineActive(g, p)) out[p] = out[p] + MixLineValue(g, rows[c][p], p);
        __syncthreads();
        if(t == 0) S.cur[c] = g.newCur;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// chorus / flanger.  Dynamic LDS: mBBuffer [4][1024] | mModDelays [2][1024] | the line's timeline [chHist + 1024]
// (chHist samples of the ring in front of the write position, then this block)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ChorusKernel(Fx2Launch F)
{
    extern __shared__ float lds[];
    float (*Bb)[kLine] = reinterpret_cast<float(*)[kLine]>(lds);
    uint32_t *mod = reinterpret_cast<uint32_t*>(lds + 4 * kLine);
    float *T = lds + 6 * kLine;
    const uint32_t t = threadIdx.x, n = F.n, H = F.chHist, mask = F.delayMask;
    Fx2State &S = *F.st;

    // calcTriangleDelays / calcSinusoidDelays, chorus.cpp:254-318
    for(uint32_t i = t; i < n; i += 256u)
    {
        for(uint32_t side = 0; side < 2u; ++side)
        {
            const uint32_t offset = (F.lfoStart[side] + i) % F.lfoRange;
            const float norm = float(offset) * F.lfoScale;
            const float lfo = F.chWave ? (1.0f - fabsf(2.0f - norm)) : SinViaDouble(norm);
            mod[side * kLine + i] = uint32_t(__float2int_rn(lfo * F.chDepth) + F.chDelay);
        }
        for(uint32_t k = 0; k < 4u; ++k) Bb[k][i] = 0.0f;
    }
    for(uint32_t c = 0; c < 4u; ++c)
    {
        float *ring = F.delay + size_t{c} * (mask + 1u);
        __syncthreads();
        for(uint32_t j = t; j < H; j += 256u) T[j] = ring[(F.offset - H + j) & mask];
        for(uint32_t i = t; i < n; i += 256u) T[H + i] = AFormat(F, false, c, i);
        __syncthreads();
        // delaybuf[offset] = input; ...; delaybuf[offset] += delaybuf[(offset - avgdelay)]*feedback: avgdelay chains (avgdelay
        // is mDelay, a count of 1/256 samples, shifted by MixerFracBits = 16 -- 0 .. 12 samples), each carried in a register
        const uint32_t D = F.chAvgDelay;
        if(D == 0u)
            for(uint32_t i = t; i < n; i += 256u) T[H + i] = T[H + i] + T[H + i] * F.chFeedback;
        else if(t < D && t < n)
        {
            float prev = T[H + t - D];
            uint32_t i = t;
            for(; i + 7u * D < n; i += 8u * D)
            {
                float a[8];
#pragma unroll
                for(uint32_t k = 0; k < 8u; ++k) a[k] = T[H + i + k * D];
#pragma unroll
                for(uint32_t k = 0; k < 8u; ++k) { prev = a[k] + prev * F.chFeedback; T[H + i + k * D] = prev; }
            }
            for(; i < n; i += D) { prev = T[H + i] + prev * F.chFeedback; T[H + i] = prev; }
        }
        __syncthreads();
        const uint32_t *md = mod + (c < 2u ? 0u : kLine);
        for(uint32_t i = t; i < n; i += 256u)
        {
            const uint32_t m = md[i], d = H + i - (m >> 8), ph = m & 255u;
            const float s = T[d + 1u] * F.cubic[256u + ph] + T[d] * F.cubic[ph] + T[d - 1u] * F.cubic[256u - ph]
                + T[d - 2u] * F.cubic[512u - ph];
            for(uint32_t k = 0; k < 4u; ++k) Bb[k][i] = Bb[k][i] + s * A2B(false, k, c);
            ring[(F.offset + i) & mask] = T[H + i];
        }
    }
    __syncthreads();
    FxOutput(F, S, Bb, 4u, t);
}

// ---------------------------------------------------------------------------------------------------------------
// distortion: wavefront c = A-Format line c, 256 input samples (1024 oversampled) per round
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float BiquadStep(const BiquadState &f, float &z1, float &z2, float x)
{   // BiquadFilter::process, core/filters/biquad.cpp:176-201
    const float y = x * f.b0 + z1;
    z1 = x * f.b1 - y * f.a1 + z2;
    z2 = x * f.b2 - y * f.a2;
    return y;
}

__global__ void __launch_bounds__(256) DistortionKernel(Fx2Launch F)
{
    __shared__ float tmp[4][kLine], Bb[4][kLine];
    const uint32_t t = threadIdx.x, lane = t & 63u, n = F.n;
    const uint32_t c = __builtin_amdgcn_readfirstlane(t >> 6);
    Fx2State &S = *F.st;
    for(uint32_t i = t; i < n; i += 256u)
        for(uint32_t k = 0; k < 4u; ++k) Bb[k][i] = 0.0f;
    BiquadState lp = S.lp[c], bp = S.bp[c];
    const float fc = F.edgeCoeff;
    for(uint32_t base = 0; base < n; base += 256u)
    {
        const uint32_t todo = ((n - base) < 256u ? (n - base) : 256u) * 4u;
        float *row = tmp[c];
        // zero stuffing (x4 to keep the power) + low-pass
        WaveSerial<1>(todo, lane, row,
            [&](uint32_t i, float *v) { v[0] = !(i & 3u) ? AFormat(F, true, c, (i >> 2) + base) * 4.0f : 0.0f; },
            [&](const float *x) { return BiquadStep(lp, lp.z1, lp.z2, x[0]); });
        WaveSync();
        for(uint32_t i = lane; i < todo; i += 64u)
        {
            float smp = row[i];
            smp = (1.0f + fc) * smp / (1.0f + fc * fabsf(smp));
            smp = (-1.0f - fc) * smp / (1.0f + fc * fabsf(smp));
            smp = (1.0f + fc) * smp / (1.0f + fc * fabsf(smp));
            row[i] = smp;
        }
        WaveSync();
        WaveSerial<1>(todo, lane, row, [&](uint32_t i, float *v) { v[0] = row[i]; },
            [&](const float *x) { return BiquadStep(bp, bp.z1, bp.z2, x[0]); });
        __syncthreads();
        // A- to B-Format, one sample out of four, the lines in order
        for(uint32_t j = t; j < (todo >> 2); j += 256u)
            for(uint32_t i = 0; i < 4u; ++i)
            {
                float b = Bb[i][base + j];
                for(uint32_t a = 0; a < 4u; ++a) b = b + tmp[a][j * 4u] * A2B(true, i, a);
                Bb[i][base + j] = b;
            }
        __syncthreads();
    }
    if(lane == 0) { S.lp[c].z1 = lp.z1; S.lp[c].z2 = lp.z2; S.bp[c].z1 = bp.z1; S.bp[c].z2 = bp.z2; }
    __syncthreads();
    FxOutput(F, S, Bb, 4u, t);
}

// ---------------------------------------------------------------------------------------------------------------
// autowah
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) AutowahKernel(Fx2Launch F)
{
    __shared__ float buf[4][kLine], cosw[kLine], alph[kLine];
    const uint32_t t = threadIdx.x, lane = t & 63u, n = F.n;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    Fx2State &S = *F.st;
    if(wave == 0)
    {   // the envelope follower, autowah.cpp:128-146
        float env = S.envDelay;
        WaveSerial<1>(n, lane, cosw, [&](uint32_t i, float *v) { v[0] = F.peakGain * fabsf(F.wetIn[i]); },
            [&](const float *x) {
                const float a = (x[0] > env) ? F.attackRate : F.releaseRate;
                env = x[0] + (env - x[0]) * a;
                return env;
            });
        if(lane == 0) S.envDelay = env;
    }
    __syncthreads();
    for(uint32_t i = t; i < n; i += 256u)
    {
        const float w0 = fminf(F.bandwidthNorm * cosw[i] + F.freqMinNorm, 0.46f) * (kPiF * 2.0f);
        cosw[i] = CosViaDouble(w0);
        alph[i] = SinViaDouble(w0) * (0.5f / 5.0f);
    }
    __syncthreads();
    const float res = F.resonanceGain;
    for(uint32_t c0 = 0; c0 < F.numIn; c0 += 4u)
    {
        const uint32_t c = c0 + wave;
        if(c < F.numIn && F.target[c] < F.nlines)
        {
            const float *in = F.wetIn + size_t{c} * kLine;
            float z1 = S.awZ[c][0], z2 = S.awZ[c][1];
            // the peaking filter of this sample (autowah.cpp:166-182): coefficients in load(), the history in step()
            WaveSerial<6>(n, lane, buf[wave],
                [&](uint32_t i, float *v) {
                    const float alpha = alph[i], cw = cosw[i];
                    const float b0 = 1.0f + alpha * res, b1 = -2.0f * cw, b2 = 1.0f - alpha * res;
                    const float a0 = 1.0f / (1.0f + alpha / res), a1 = -2.0f * cw, a2 = 1.0f - alpha / res;
                    v[0] = in[i]; v[1] = b0 * a0; v[2] = b1 * a0; v[3] = a1 * a0; v[4] = b2 * a0; v[5] = a2 * a0;
                },
                [&](const float *x) {
                    const float out = x[0] * x[1] + z1;
                    z1 = x[0] * x[2] - out * x[3] + z2;
                    z2 = x[0] * x[4] - out * x[5];
                    return out;
                });
            if(lane == 0) { S.awZ[c][0] = z1; S.awZ[c][1] = z2; }
        }
        __syncthreads();
        for(uint32_t w = 0; w < 4u; ++w)
        {
            const uint32_t cc = c0 + w;
            if(cc >= F.numIn || F.target[cc] >= F.nlines) continue;
            const MixLineGain g = PrepareMixLine(S.cur[cc], F.tgtGain[cc], n, n);
            float *out = F.outLines + size_t{F.target[cc]} * kLine;
            for(uint32_t p = t; p < n; p += 256u)
                if(MixLineActive(g, p)) out[p] = out[p] + MixLineValue(g, buf[w][p], p);
            __syncthreads();
            if(t == 0) S.cur[cc] = g.newCur;
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// vocal morpher: lanes 0..7 of a channel's wavefront each run one formant filter
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) VmorpherKernel(Fx2Launch F)
{
    __shared__ float fo[4][8][256], lfo[256];
    const uint32_t t = threadIdx.x, lane = t & 63u, n = F.n;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    Fx2State &S = *F.st;
    // FormantFilter::process set-up, vmorpher.cpp:114-118
    const float g = F.vmG[lane & 7u], gain = F.vmGain[lane & 7u];
    const float h = 1.0f / (1.0f + (g * 0.2f) + (g * g)), coeff = 0.2f + g;
    for(uint32_t base = 0; base < n; base += 256u)
    {
        const uint32_t td = (n - base) < 256u ? (n - base) : 256u;
        if(t < td)
        {   // Oscillate<Sin/Saw/Triangle/Half>, vmorpher.cpp:79-106
            const uint32_t index = (F.vmIndex + F.vmStep * (base + t + 1u)) & 0xffffffu;
            float v = 0.5f;
            if(F.vmWave == 1) v = SinViaDouble(float(index) * (kPiF * 2.0f / 16777216.0f)) * 0.5f + 0.5f;
            else if(F.vmWave == 2) v = fabsf(float(index) * (2.0f / 16777216.0f) - 1.0f);
            else if(F.vmWave == 3) v = float(index) / 16777216.0f;
            lfo[t] = v;
        }
        for(uint32_t c0 = 0; c0 < F.numIn; c0 += 4u)
        {
            const uint32_t c = c0 + wave;
            __syncthreads();
            if(c < F.numIn && F.target[c] < F.nlines)
            {
                const float *in = F.wetIn + size_t{c} * kLine + base;
                float s1 = S.vmS[c][lane & 7u][0], s2 = S.vmS[c][lane & 7u][1];
                for(uint32_t r = 0; r < td; r += 64u)
                {
                    const float xv = (r + lane) < td ? in[r + lane] : 0.0f;
                    const uint32_t cnt = __builtin_amdgcn_readfirstlane((td - r) < 64u ? (td - r) : 64u);
                    for(uint32_t k = 0; k < cnt; ++k)
                    {
                        const float x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xv), int(__builtin_amdgcn_readfirstlane(k))));
                        const float Hh = (x - coeff * s1 - s2) * h;
                        const float B = g * Hh + s1;
                        const float Lo = g * B + s2;
                        s1 = g * Hh + B;
                        s2 = g * B + Lo;
                        if(lane < 8u) fo[wave][lane][r + k] = B * gain;
                    }
                }
                if(lane < 8u) { S.vmS[c][lane][0] = s1; S.vmS[c][lane][1] = s2; }
            }
            __syncthreads();
            for(uint32_t w = 0; w < 4u; ++w)
            {
                const uint32_t cc = c0 + w;
                if(cc >= F.numIn || F.target[cc] >= F.nlines) continue;
                const MixLineGain gl = PrepareMixLine(S.cur[cc], F.tgtGain[cc], n - base, td);
                if(t < td)
                {
                    const float a = (((0.0f + fo[w][0][t]) + fo[w][1][t]) + fo[w][2][t]) + fo[w][3][t];
                    const float b = (((0.0f + fo[w][4][t]) + fo[w][5][t]) + fo[w][6][t]) + fo[w][7][t];
                    const float blended = a + (b - a) * lfo[t];
                    float *out = F.outLines + size_t{F.target[cc]} * kLine + base;
                    if(MixLineActive(gl, t)) out[t] = out[t] + MixLineValue(gl, blended, t);
                }
                __syncthreads();
                if(t == 0) S.cur[cc] = gl.newCur;
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// frequency shifter.  FshifterLineKernel: workgroup c = A-Format line c -- FIFO, window, complex_hilbert
// (common/alcomplex.cpp:199-215: inverse FFT, keep the positive frequencies, forward FFT), overlap-add.
// FshifterMixKernel: the phase rotation, A- to B-Format, the output stage.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t Rev10(uint32_t v) { return __brev(v) >> 22; }

// complex_fft, common/alcomplex.cpp:105-148, on data already in bit-reversed order: tw[step2 + j] = w^j of the
// stage, built by the host the way the reference builds it (u *= w); conj = the forward transform (sign -1)
__device__ void Fft1024(FsPair *buf, const FsPair *tw, bool conj, uint32_t t)
{
    for(uint32_t i = 0; i < 10u; ++i)
    {
        const uint32_t step2 = 1u << i;
        for(uint32_t p = t; p < 512u; p += 256u)
        {
            const uint32_t j = p & (step2 - 1u), k = ((p >> i) << (i + 1u)) | j;
            const FsPair a = buf[k], b = buf[k + step2];
            FsPair tmp = b;
            if(j)
            {
                FsPair u = tw[step2 + j];
                if(conj) u.y = -u.y;
                tmp.x = b.x * u.x - b.y * u.y;
                tmp.y = b.x * u.y + b.y * u.x;
            }
            buf[k + step2] = FsPair{a.x - tmp.x, a.y - tmp.y};
            buf[k] = FsPair{a.x + tmp.x, a.y + tmp.y};
        }
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) FshifterLineKernel(Fx2Launch F)
{
    __shared__ FsPair buf[1024];
    const uint32_t t = threadIdx.x, n = F.n, c = blockIdx.x;
    double *inFifo = F.fsIn + size_t{c} * 1024u;
    FsPair *outFifo = F.fsOutFifo + size_t{c} * 256u, *accum = F.fsAccum + size_t{c} * 1024u;
    FsPair *outdata = F.fsOutdata + size_t{c} * kLine;
    const uint32_t numInput = F.numIn < 4u ? F.numIn : 4u;
    uint32_t count = F.fsCount, pos = F.fsPos;
    for(uint32_t base = 0; base < n;)
    {
        const uint32_t todo = (256u - count) < (n - base) ? (256u - count) : (n - base);
        for(uint32_t j = t; j < todo; j += 256u)
        {
            double a = 0.0;
            for(uint32_t i = 0; i < numInput; ++i)
                a = a + double(F.wetIn[size_t{i} * kLine + base + j]) * double(B2A(false, c, i));
            inFifo[pos + count + j] = a;
            outdata[base + j] = outFifo[count + j];
        }
        count += todo; base += todo;
        if(count < 256u) break;
        count = 0; pos = (pos + 256u) & 1023u;
        __syncthreads();
        for(uint32_t k = t; k < 1024u; k += 256u)
            buf[Rev10(k)] = FsPair{inFifo[(pos + k) & 1023u] * double(F.fsWindow[k]), 0.0};
        __syncthreads();
        Fft1024(buf, F.fsTw, false, t);
        FsPair v[4];
        for(uint32_t q = 0; q < 4u; ++q)
        {
            const uint32_t k = t + q * 256u;
            const double inv = 1.0 / 1024.0;
            const FsPair d = buf[k];
            if(k == 0u || k == 512u) v[q] = FsPair{d.x * inv, d.y * inv};
            else if(k < 512u) v[q] = FsPair{d.x * (inv * 2.0), d.y * (inv * 2.0)};
            else v[q] = FsPair{0.0, 0.0};
        }
        __syncthreads();
        for(uint32_t q = 0; q < 4u; ++q) buf[Rev10(t + q * 256u)] = v[q];
        __syncthreads();
        Fft1024(buf, F.fsTw, true, t);
        for(uint32_t k = t; k < 1024u; k += 256u)
        {
            const double w = 2.0 / 4.0 * double(F.fsWindow[k]);
            const FsPair a = buf[k], o = accum[(pos + k) & 1023u];
            accum[(pos + k) & 1023u] = FsPair{o.x + w * a.x, o.y + w * a.y};
        }
        __syncthreads();
        outFifo[t] = accum[pos + t];
        accum[pos + t] = FsPair{0.0, 0.0};
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256) FshifterMixKernel(Fx2Launch F)
{
    __shared__ float Bb[4][kLine];
    const uint32_t t = threadIdx.x, n = F.n;
    Fx2State &S = *F.st;
    for(uint32_t p = t; p < n; p += 256u)
    {
        float b[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for(uint32_t c = 0; c < 4u; ++c)
        {
            const FsPair in = F.fsOutdata[size_t{c} * kLine + p];
            const FsPair cs = F.fsPhase[(F.fsPhaseIdx[c] + p * F.fsPhaseStep[c]) & 65535u];
            const float out = float(in.x * cs.x + in.y * cs.y * F.fsSign[c]);
            for(uint32_t i = 0; i < 4u; ++i) b[i] = b[i] + out * A2B(false, i, c);
        }
        for(uint32_t i = 0; i < 4u; ++i) Bb[i][p] = b[i];
    }
    __syncthreads();
    FxOutput(F, S, Bb, 4u, t);
}


// ---------------------------------------------------------------------------------------------------------------
// pitch shifter (alc/effects/pshifter.cpp:213-460): a phase vocoder -- 1024-point STFT, hop 128; the W channel's
// bins are analysed for their true frequency, moved to round(k * pitch) and re-synthesised with accumulated phases,
// the other channels keep their phase offset from W.  PshifterLineKernel: workgroup c = channel c.  A hop of
// channel c > 0 needs what W's analysis of the SAME hop left in mLastPhase / mSumPhase, so every workgroup redoes
// W's analysis for itself (one more transform per hop) instead of waiting for workgroup 0; the state a launch
// reads (input history, phases) and the state it leaves are separate buffers, so no workgroup reads what another
// one writes.  The transforms are plain radix-2 complex FFTs in LDS, not pffft's butterfly order, and atan2f /
// hypotf / sinf / cosf are the GPU's: parity is a tolerance here (tests/test_effects2.py), not bit equality.
// ---------------------------------------------------------------------------------------------------------------
struct Cf { float x, y; };

// data in bit-reversed order -> natural order; tw[m] = e^(-2 pi i m / 1024); inverse: the conjugate, unnormalised
__device__ void FftF1024(Cf *buf, const Cf *tw, bool inverse, uint32_t t)
{
    for(uint32_t i = 0; i < 10u; ++i)
    {
        const uint32_t step2 = 1u << i;
        for(uint32_t p = t; p < 512u; p += 256u)
        {
            const uint32_t j = p & (step2 - 1u), k = ((p >> i) << (i + 1u)) | j;
            const Cf a = buf[k], b = buf[k + step2];
            Cf u = tw[j << (9u - i)];
            if(inverse) u.y = -u.y;
            const Cf tmp{b.x * u.x - b.y * u.y, b.x * u.y + b.y * u.x};
            buf[k + step2] = Cf{a.x - tmp.x, a.y - tmp.y};
            buf[k] = Cf{a.x + tmp.x, a.y + tmp.y};
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float WrapPhase(float tmp)
{   // qpd = float2int(tmp); tmp -= qpd + qpd%2, pshifter.cpp:296-297
    const int32_t qpd = int32_t(tmp);
    return tmp - float(qpd + (qpd % 2));
}

__global__ void __launch_bounds__(256) PshifterLineKernel(Fx2Launch F)
{
    __shared__ Cf buf[1024];
    __shared__ float mag[513], ph[513], sm[513], sf[513], last[513], sum[513];
    const uint32_t t = threadIdx.x, n = F.n, c = blockIdx.x;
    const float kExpected = kPiF * 2.0f / 8.0f, kInvPi = 0.318309886183790671538f;
    const float *ring0 = F.psRingIn, *ringC = F.psRingIn + size_t{c} * 1024u;
    const float *in0 = F.wetIn, *inC = F.wetIn + size_t{c} * kLine;
    float *accum = F.psAccum + size_t{c} * 1024u, *outFifo = F.psOutFifo + size_t{c} * 128u, *rows = F.psRows + size_t{c} * kLine;
    // the timeline of channel ch: the 1024 samples before this launch, then this block
    auto timeline = [&](const float *ring, const float *in, uint32_t x) { return x < 1024u ? ring[x] : in[x - 1024u]; };
    for(uint32_t k = t; k < 513u; k += 256u) { last[k] = F.psPhaseIn[k]; sum[k] = F.psPhaseIn[513u + k]; }
    uint32_t count = F.psCount, pos = F.psPos;
    const uint32_t pitchI = F.psPitchI;
    const uint32_t binCount = min(513u, ((513u << 16) - 32768u - 1u) / pitchI + 1u);
    __syncthreads();
    for(uint32_t base = 0; base < n;)
    {
        const uint32_t todo = (128u - count) < (n - base) ? (128u - count) : (n - base);
        for(uint32_t j = t; j < todo; j += 256u) rows[base + j] = outFifo[count + j];
        count += todo; base += todo;
        if(count < 128u) break;
        count = 0; pos = (pos + 128u) & 1023u;
        const uint32_t start = base;                    // the window: timeline[base .. base + 1024)
        __syncthreads();
        for(uint32_t pass = 0; pass < 2u; ++pass)
        {   // pass 0: W's analysis (every workgroup); pass 1: this channel (workgroup 0: W's own synthesis)
            const bool own = pass == 1u;
            if(own && c == 0u) { /* W: the spectrum analysed in pass 0 is still what is needed; go on to synthesis */ }
            else
            {
                const float *ring = own ? ringC : ring0, *in = own ? inC : in0;
                for(uint32_t k = t; k < 1024u; k += 256u)
                    buf[Rev10(k)] = Cf{timeline(ring, in, start + k) * F.psWindow[k], 0.0f};
                __syncthreads();
                FftF1024(buf, reinterpret_cast<const Cf*>(F.psTw), false, t);
                for(uint32_t k = t; k < 513u; k += 256u)
                {
                    const Cf v = (k == 0u || k == 512u) ? Cf{buf[k].x, 0.0f} : buf[k];
                    mag[k] = hypotf(v.x, v.y);
                    ph[k] = atan2f(v.y, v.x);
                }
                __syncthreads();
            }
            if(!own)
            {   // W: true frequency per bin, moved bins, accumulated synthesis phases (pshifter.cpp:268-356)
                for(uint32_t k = t; k < 513u; k += 256u)
                {
                    float tmp = (ph[k] - last[k]) - float(k & 7u) * kExpected;
                    last[k] = ph[k];
                    tmp = WrapPhase(tmp * kInvPi) * 4.0f;
                    ph[k] = (float(k) + tmp) * F.psPitch;           // freqbin * mPitchShift
                }
                __syncthreads();
            }
            else if(c != 0u)
            {
                for(uint32_t k = t; k < 513u; k += 256u) ph[k] = ph[k] - last[k];     // phase offset from W
                __syncthreads();
            }
            if(!(own && c == 0u))
            {   // mSynthesisBuffer[j]: the source bins k with round(k * pitch) == j, in ascending order
                const uint32_t kcount = own ? binCount : 513u;
                for(uint32_t j = t; j < 513u; j += 256u)
                {
                    const uint64_t lo = (uint64_t{j} << 16) < 32768u ? 0u : ((uint64_t{j} << 16) - 32768u + pitchI - 1u) / pitchI;
                    const uint64_t hiEx = (((uint64_t{j} + 1u) << 16) - 32768u + pitchI - 1u) / pitchI;     // first k beyond j
                    float m = 0.0f, f = 0.0f;
                    for(uint64_t k = lo; k < hiEx && k < kcount; ++k)
                    {
                        if(m < mag[k]) f = ph[k];
                        m = m + mag[k];
                    }
                    sm[j] = m; sf[j] = f;
                }
                __syncthreads();
            }
            if(!own)
            {
                for(uint32_t k = t; k < 513u; k += 256u)
                {
                    float tmp = (sf[k] - float(k & ~7u)) * kExpected;
                    tmp = WrapPhase((tmp + sum[k]) * kInvPi);
                    sum[k] = tmp * kPiF;
                }
                __syncthreads();
                if(c != 0u) continue;                  // the other channels only needed W's phases
            }
            // polar(magnitude, phase) -> the Hermitian spectrum -> inverse transform
            for(uint32_t k = t; k < 513u; k += 256u)
            {
                const float phase = (c == 0u) ? sum[k] : WrapPhase((sum[k] + sf[k]) * kInvPi) * kPiF;
                Cf v{sm[k] * CosViaDouble(phase), sm[k] * SinViaDouble(phase)};
                if(k == 0u || k == 512u) v.y = 0.0f;
                buf[Rev10(k)] = v;
                if(k != 0u && k != 512u) buf[Rev10(1024u - k)] = Cf{v.x, -v.y};
            }
            __syncthreads();
            FftF1024(buf, reinterpret_cast<const Cf*>(F.psTw), true, t);
            for(uint32_t k = t; k < 1024u; k += 256u)
            {
                const float y = F.psWindow[k] * buf[k].x * (3.0f / 8.0f / 1024.0f);
                accum[(pos + k) & 1023u] = accum[(pos + k) & 1023u] + y;
            }
            __syncthreads();
            if(t < 128u) { outFifo[t] = accum[pos + t]; accum[pos + t] = 0.0f; }
            __syncthreads();
            if(c == 0u) break;                          // W is done after its single pass pair
        }
    }
    // what the next launch starts from
    for(uint32_t i = t; i < 1024u; i += 256u) F.psRingOut[size_t{c} * 1024u + i] = timeline(ringC, inC, n + i);
    if(c == 0u)
        for(uint32_t k = t; k < 513u; k += 256u) { F.psPhaseOut[k] = last[k]; F.psPhaseOut[513u + k] = sum[k]; }
}

__global__ void __launch_bounds__(256) PshifterMixKernel(Fx2Launch F)
{
    __shared__ float Bb[9][kLine];
    const uint32_t t = threadIdx.x, n = F.n;
    const uint32_t nchan = F.numIn < 9u ? F.numIn : 9u;
    for(uint32_t c = 0; c < nchan; ++c)
        for(uint32_t p = t; p < n; p += 256u) Bb[c][p] = F.psRows[size_t{c} * kLine + p];
    __syncthreads();
    FxOutput(F, *F.st, Bb, nchan, t);
}

} // namespace

void LaunchEffect2(hipStream_t s, const Fx2Launch &F, uint32_t ldsBytes)
{
    switch(F.kind)
    {
    case OALGPU_EFFECT_CHORUS: hipLaunchKernelGGL(ChorusKernel, dim3(1), dim3(256), ldsBytes, s, F); break;
    case OALGPU_EFFECT_DISTORTION: hipLaunchKernelGGL(DistortionKernel, dim3(1), dim3(256), 0, s, F); break;
    case OALGPU_EFFECT_AUTOWAH: hipLaunchKernelGGL(AutowahKernel, dim3(1), dim3(256), 0, s, F); break;
    case OALGPU_EFFECT_VMORPHER: hipLaunchKernelGGL(VmorpherKernel, dim3(1), dim3(256), 0, s, F); break;
    case OALGPU_EFFECT_FSHIFTER:
        hipLaunchKernelGGL(FshifterLineKernel, dim3(4), dim3(256), 0, s, F);
        hipLaunchKernelGGL(FshifterMixKernel, dim3(1), dim3(256), 0, s, F);
        break;
    case OALGPU_EFFECT_PSHIFTER:
        hipLaunchKernelGGL(PshifterLineKernel, dim3(F.numIn < 9u ? F.numIn : 9u), dim3(256), 0, s, F);
        hipLaunchKernelGGL(PshifterMixKernel, dim3(1), dim3(256), 0, s, F);
        break;
    default: break;
    }
}

} // namespace oalgpu
