// The stage behind the buses: ambisonic dry lines -> speaker feeds, dither, and
// the interleaved / converted PCM a host takes away -- so that config 2 ends in its 7.1 feeds and the
// D2H copy of an update is the final PCM, not raw buses.
//
//   BFormatDec::process        core/bformatdec.cpp:60-95    (AmbiDecPostProcess, alc/alu.cpp:282-287)
//     BandSplitter::process    core/filters/splitter.cpp:28-63   (dual-band decoders)
//     MixSamples, Counter = 0  core/mixer/mixer_c.cpp:150-186    (constant gains, silence threshold)
//   ApplyDither                alc/alu.cpp:2309-2332
//   SampleConv<T>, Write<T>    alc/alu.cpp:2335-2390
//
// Integer results (the PCM, the dither sequence) are bit-exact in every mode; the decoder's floats are
// bit-exact in EXACT mode (the reference's operation order: per input channel HF then LF, multiply and
// add separately) and within the FAST tolerance otherwise.
#include "dev_wave.hpp"
#include "dev_resample.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr float kSilence = 1.0e-5f;          // GainSilenceThreshold, core/mixer/defs.h:28

// ---- BandSplitter::process: hp / lp of one dry line, state (lpZ1, lpZ2, apZ1) -----------------------------
struct Band2 { float hp, lp; };
__device__ __forceinline__ Band2 BandStep(Sp3 &s, float x, float apCoeff, float lpCoeff)
{
    const float d0 = (x - s.a) * lpCoeff;
    const float lpY0 = s.a + d0;
    s.a = lpY0 + d0;
    const float d1 = (lpY0 - s.b) * lpCoeff;
    const float lpY1 = s.b + d1;
    s.b = lpY1 + d1;
    const float apY = x * apCoeff + s.c;
    s.c = x - apY * apCoeff;
    return Band2{apY - lpY1, lpY1};
}

// one wavefront per dry line.  EXACT: the reference's serial loop on one lane; FAST: the 3-state block
// scan of dev_wave.hpp (SplitterScan) with both bands kept.
template<bool EXACT>
__global__ void __launch_bounds__(64) BandSplitKernel(const float *lines, SplitterState *states, float *bands /* [line][2][1024] */,
    uint32_t n)
{
    const uint32_t lane = threadIdx.x, ch = blockIdx.x;
    const float *in = lines + size_t{ch} * kLine;
    float *hp = bands + size_t{ch} * 2 * kLine, *lp = hp + kLine;
    SplitterState st = states[ch];
    const float apCoeff = st.coeff, lpCoeff = st.coeff * 0.5f + 0.5f;
    if constexpr (EXACT)
    {
        if(lane == 0)
        {
            Sp3 s{st.lpZ1, st.lpZ2, st.apZ1};
            for(uint32_t i = 0; i < n; ++i) { const Band2 b = BandStep(s, in[i], apCoeff, lpCoeff); hp[i] = b.hp; lp[i] = b.lp; }
            states[ch].lpZ1 = s.a; states[ch].lpZ2 = s.b; states[ch].apZ1 = s.c;
        }
        return;
    }
    else
    {
        // lane l owns samples [17 l, 17 l + 17) (64 runs of 17 cover 1024); its samples are requested up front
        // (independent loads), the runs have equal length (zeros past n): one transition matrix for all
        constexpr uint32_t seg = 17;
        const uint32_t begin = lane * seg;
        float x[seg];
#pragma unroll
        for(uint32_t i = 0; i < seg; ++i) x[i] = (begin + i < n) ? in[begin + i] : 0.0f;
        Mat3 M{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        Sp3 q{0, 0, 0};
#pragma unroll
        for(uint32_t i = 0; i < seg; ++i)
        {
            BandStep(M.c0, 0.0f, apCoeff, lpCoeff); BandStep(M.c1, 0.0f, apCoeff, lpCoeff); BandStep(M.c2, 0.0f, apCoeff, lpCoeff);
            BandStep(q, x[i], apCoeff, lpCoeff);
        }
        Sp3 e = q;
        Sp3 s0{st.lpZ1, st.lpZ2, st.apZ1};
        Mat3 P = M;
#pragma unroll
        for(int step = 0; step < 6; ++step)
        {
            const int d = 1 << step;
            Sp3 o;
            o.a = __shfl_up(e.a, d); o.b = __shfl_up(e.b, d); o.c = __shfl_up(e.c, d);
            const Sp3 mo = MatVec3(P, o);
            if(int(lane) >= d) { e.a += mo.a; e.b += mo.b; e.c += mo.c; }
            const Sp3 ms = MatVec3(P, s0);
            if(lane & uint32_t(d)) s0 = ms;
            if(step < 5) P = MatMul3(P, P);
        }
        Sp3 prevE;
        prevE.a = __shfl_up(e.a, 1); prevE.b = __shfl_up(e.b, 1); prevE.c = __shfl_up(e.c, 1);
        Sp3 start = s0;
        if(lane > 0) { start.a += prevE.a; start.b += prevE.b; start.c += prevE.c; }
        // the state behind sample n-1: the run that holds it stops there
        const uint32_t lastLane = (n - 1u) / seg, lastLen = n - lastLane * seg;
        Sp3 fin = start;
#pragma unroll
        for(uint32_t i = 0; i < seg; ++i)
        {
            const Band2 b = BandStep(start, x[i], apCoeff, lpCoeff);
            if(begin + i < n) { hp[begin + i] = b.hp; lp[begin + i] = b.lp; }
            if(i + 1u == lastLen) fin = start;
        }
        const float z0 = __shfl(fin.a, int(lastLane)), z1 = __shfl(fin.b, int(lastLane)), z2 = __shfl(fin.c, int(lastLane));
        if(lane == 0) { states[ch].lpZ1 = z0; states[ch].lpZ2 = z1; states[ch].apZ1 = z2; }
    }
}

// out[c][i] += sum over the dry lines j, in the reference's order (j ascending; per j HF then LF), of
// in_j[i] * gain[j][c]; a gain whose magnitude does not exceed GainSilenceThreshold is skipped as MixLine skips it
template<bool EXACT, bool DUAL>
__global__ void __launch_bounds__(64) BFormatMixKernel(float *out, const float *lines, const float *bands, const float *gainsHf /* [in][32] */,
    const float *gainsLf, uint32_t nin, uint32_t nout, uint32_t n)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if(i >= n) return;
    // eight output lines at a time: their sums advance together over the dry lines (every sum still in the
    // reference's order), so that a dry line's samples are read once per eight outputs and the loads of one j
    // do not wait for the adds of the previous one
    for(uint32_t c0 = 0; c0 < nout; c0 += 8u)
    {
        float acc[8];
#pragma unroll
        for(uint32_t c = 0; c < 8u; ++c) acc[c] = (c0 + c < nout) ? out[size_t{c0 + c} * kLine + i] : 0.0f;
        for(uint32_t j = 0; j < nin; ++j)
        {
            float hp, lp = 0.0f;
            if constexpr (DUAL) { hp = bands[(size_t{j} * 2) * kLine + i]; lp = bands[(size_t{j} * 2 + 1) * kLine + i]; }
            else hp = lines[size_t{j} * kLine + i];
#pragma unroll
            for(uint32_t c = 0; c < 8u; ++c)
            {
                const uint32_t cc = (c0 + c < nout) ? c0 + c : c0;      // gains are padded to 32 per dry line
                const float gh = gainsHf[j * 32u + cc];
                if(fabsf(gh) > kSilence) acc[c] = EXACT ? acc[c] + hp * gh : __builtin_fmaf(hp, gh, acc[c]);
                if constexpr (DUAL)
                {
                    const float gl = gainsLf[j * 32u + cc];
                    if(fabsf(gl) > kSilence) acc[c] = EXACT ? acc[c] + lp * gl : __builtin_fmaf(lp, gl, acc[c]);
                }
            }
        }
#pragma unroll
        for(uint32_t c = 0; c < 8u; ++c) if(c0 + c < nout) out[size_t{c0 + c} * kLine + i] = acc[c];
    }
}

// ---- ApplyDither: the reference draws two values of its LCG (alu.cpp:444-448) per sample, line after line;
// draw k of the sequence is an affine map of the seed, built by square-and-multiply
__device__ __forceinline__ uint32_t DitherJump(uint32_t seed, uint32_t k)
{
    uint32_t accA = 1u, accC = 0u;                   // composition so far: x -> accA x + accC
    uint32_t a = 96314165u, c = 907633515u;          // one step
    while(k)
    {
        if(k & 1u) { accC = a * accC + c; accA = a * accA; }
        c = a * c + c; a = a * a;
        k >>= 1;
    }
    return accA * seed + accC;
}

__global__ void __launch_bounds__(256) DitherKernel(float *lines, uint32_t nlines, uint32_t n, float quantScale, uint32_t seed)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x, line = blockIdx.y;
    if(i >= n || line >= nlines) return;
    const uint32_t draw = (line * n + i) * 2u;
    const uint32_t rng0 = DitherJump(seed, draw + 1u);
    const uint32_t rng1 = rng0 * 96314165u + 907633515u;
    constexpr double invRange = 1.0 / 4294967295.0;
    float val = lines[size_t{line} * kLine + i] * quantScale;
    val += float(double(rng0) * invRange - double(rng1) * invRange);
    const float invScale = 1.0f / quantScale;
    lines[size_t{line} * kLine + i] = rintf(val) * invScale;      // fast_roundf: round to nearest even
}

// ---- SampleConv<T> + Write<T>: frame i, channel c of the interleaved output
template<typename T> __device__ __forceinline__ T ConvSample(float v);
template<> __device__ __forceinline__ float ConvSample<float>(float v) { return v; }
template<> __device__ __forceinline__ int32_t ConvSample<int32_t>(float v)
{ return __float2int_rn(fminf(fmaxf(v * 2147483648.0f, -2147483648.0f), 2147483520.0f)); }
template<> __device__ __forceinline__ int16_t ConvSample<int16_t>(float v)
{ return int16_t(__float2int_rn(fminf(fmaxf(v * 32768.0f, -32768.0f), 32767.0f))); }
template<> __device__ __forceinline__ int8_t ConvSample<int8_t>(float v)
{ return int8_t(__float2int_rn(fminf(fmaxf(v * 128.0f, -128.0f), 127.0f))); }
template<> __device__ __forceinline__ uint32_t ConvSample<uint32_t>(float v) { return uint32_t(ConvSample<int32_t>(v)) + 2147483648u; }
template<> __device__ __forceinline__ uint16_t ConvSample<uint16_t>(float v) { return uint16_t(uint16_t(ConvSample<int16_t>(v)) + 32768u); }
template<> __device__ __forceinline__ uint8_t ConvSample<uint8_t>(float v) { return uint8_t(uint8_t(ConvSample<int8_t>(v)) + 128u); }

template<typename T>
__global__ void __launch_bounds__(256) WriteKernel(const float *lines, uint32_t nlines, uint32_t n, uint32_t frameStep, T *out)
{
    const uint32_t k = blockIdx.x * 256u + threadIdx.x;          // element of the interleaved buffer
    if(k >= n * frameStep) return;
    const uint32_t i = k / frameStep, c = k % frameStep;
    out[k] = ConvSample<T>(c < nlines ? lines[size_t{c} * kLine + i] : 0.0f);
}

// ---- SampleConverter::convert, core/converter.cpp:236-330 ------------------------------------------------------
// The reference walks the input in chunks of <= 1024 frames: per channel [PrevSamples | new input] -> float
// (LoadSample<T>, :31-49), mResample into <= 1024 outputs, StoreSample<T> (:100-118).  Every chunk's source line
// is a window of ONE timeline -- the converter's prep samples followed by the call's input frames -- so the
// host only does the integer bookkeeping (the chunk table) and one launch computes every output of the call:
// block.y = chunk, block.z = channel, thread = output i of the chunk (the cubic resampler's summation order
// depends on i and on the chunk's size, hence the table).  Another block row files the next call's prep samples.
template<typename T> __device__ __forceinline__ float LoadDevSample(const void *src, size_t k);
template<> __device__ __forceinline__ float LoadDevSample<int8_t>(const void *s, size_t k) { return float(static_cast<const int8_t*>(s)[k]) * (1.0f / 128.0f); }
template<> __device__ __forceinline__ float LoadDevSample<uint8_t>(const void *s, size_t k) { return float(int8_t(static_cast<const uint8_t*>(s)[k] - 128u)) * (1.0f / 128.0f); }
template<> __device__ __forceinline__ float LoadDevSample<int16_t>(const void *s, size_t k) { return float(static_cast<const int16_t*>(s)[k]) * (1.0f / 32768.0f); }
template<> __device__ __forceinline__ float LoadDevSample<uint16_t>(const void *s, size_t k) { return float(int16_t(static_cast<const uint16_t*>(s)[k] - 32768u)) * (1.0f / 32768.0f); }
template<> __device__ __forceinline__ float LoadDevSample<int32_t>(const void *s, size_t k) { return float(static_cast<const int32_t*>(s)[k]) * (1.0f / 2147483648.0f); }
template<> __device__ __forceinline__ float LoadDevSample<uint32_t>(const void *s, size_t k) { return float(int32_t(static_cast<const uint32_t*>(s)[k] - 2147483648u)) * (1.0f / 2147483648.0f); }
template<> __device__ __forceinline__ float LoadDevSample<float>(const void *s, size_t k) { return static_cast<const float*>(s)[k]; }

__device__ __forceinline__ float LoadDevSampleAny(int type, const void *src, size_t k)
{
    switch(type)
    {
    case OALGPU_OUT_I8: return LoadDevSample<int8_t>(src, k);
    case OALGPU_OUT_U8: return LoadDevSample<uint8_t>(src, k);
    case OALGPU_OUT_I16: return LoadDevSample<int16_t>(src, k);
    case OALGPU_OUT_U16: return LoadDevSample<uint16_t>(src, k);
    case OALGPU_OUT_I32: return LoadDevSample<int32_t>(src, k);
    case OALGPU_OUT_U32: return LoadDevSample<uint32_t>(src, k);
    default: return LoadDevSample<float>(src, k);
    }
}
__device__ __forceinline__ void StoreDevSampleAny(int type, void *dst, size_t k, float v)
{
    switch(type)
    {
    case OALGPU_OUT_I8: static_cast<int8_t*>(dst)[k] = ConvSample<int8_t>(v); break;
    case OALGPU_OUT_U8: static_cast<uint8_t*>(dst)[k] = ConvSample<uint8_t>(v); break;
    case OALGPU_OUT_I16: static_cast<int16_t*>(dst)[k] = ConvSample<int16_t>(v); break;
    case OALGPU_OUT_U16: static_cast<uint16_t*>(dst)[k] = ConvSample<uint16_t>(v); break;
    case OALGPU_OUT_I32: static_cast<int32_t*>(dst)[k] = ConvSample<int32_t>(v); break;
    case OALGPU_OUT_U32: static_cast<uint32_t*>(dst)[k] = ConvSample<uint32_t>(v); break;
    default: static_cast<float*>(dst)[k] = v; break;
    }
}

// element g of channel c's timeline: the prep samples the call started with, then the call's input
struct ConvTimeline {
    const float *prev; const void *src; uint32_t prep0, channels, chan, srcFrames; int type; uint32_t base;
    __device__ __forceinline__ float operator[](uint32_t idx) const
    {
        const uint32_t g = base + idx;
        if(g < prep0) return prev[g];
        const uint32_t f = g - prep0;
        return f < srcFrames ? LoadDevSampleAny(type, src, size_t{f} * channels + chan) : 0.0f;
    }
};

__global__ void __launch_bounds__(256) SampleConvertKernel(ConvertJob J, const ConvertChunk *chunks)
{
    const uint32_t c = blockIdx.z, t = threadIdx.x;
    if(blockIdx.y == J.numChunks)
    {   // the prep samples the next call starts from: timeline[endBase .. endBase + nextPrep), zero-filled
        if(blockIdx.x == 0 && t < kMaxPad)
        {
            const ConvTimeline tl{J.prev + size_t{c} * kMaxPad, J.src, J.prep0, J.channels, c, J.srcFrames, J.srcType, J.endBase};
            J.newPrev[size_t{c} * kMaxPad + t] = t < J.nextPrep ? tl[t] : 0.0f;
        }
        return;
    }
    const ConvertChunk k = chunks[blockIdx.y];
    const uint32_t i = blockIdx.x * 256u + t;
    if(i >= k.dstSize) return;
    const ConvTimeline tl{J.prev + size_t{c} * kMaxPad, J.src, J.prep0, J.channels, c, J.srcFrames, J.srcType, k.srcBase};
    float v;
    if(J.increment == kFracOne) v = tl[kMaxEdge + i];                  // the copy "resampler" of :208-213
    else v = ResampleAt<true>(J.spec.kind, J.spec.m, J.spec.l, J.spec.sf, J.spec.filter, ReferenceTabLayout(J.spec.m), tl, k.frac0,
        J.increment, i, k.dstSize);
    StoreDevSampleAny(J.dstType, J.dst, (size_t{k.dstBase} + i) * J.channels + c, v);
}

} // namespace

void LaunchSampleConvert(hipStream_t s, const ConvertJob &J, const ConvertChunk *chunks)
{
    hipLaunchKernelGGL(SampleConvertKernel, dim3(kLine / 256, J.numChunks + 1u, J.channels), dim3(256), 0, s, J, chunks);
}

void LaunchBFormatDecode(hipStream_t s, bool exact, float *out, const float *lines, SplitterState *states, float *bands,
    const float *gainsHf, const float *gainsLf, uint32_t nin, uint32_t nout, uint32_t n)
{
    const dim3 mgrid((n + 63u) / 64u);
    if(gainsLf)
    {
        if(exact)
        {
            hipLaunchKernelGGL(BandSplitKernel<true>, dim3(nin), dim3(64), 0, s, lines, states, bands, n);
            hipLaunchKernelGGL((BFormatMixKernel<true, true>), mgrid, dim3(64), 0, s, out, lines, bands, gainsHf, gainsLf, nin, nout, n);
        }
        else
        {
            hipLaunchKernelGGL(BandSplitKernel<false>, dim3(nin), dim3(64), 0, s, lines, states, bands, n);
            hipLaunchKernelGGL((BFormatMixKernel<false, true>), mgrid, dim3(64), 0, s, out, lines, bands, gainsHf, gainsLf, nin, nout, n);
        }
    }
    else if(exact) hipLaunchKernelGGL((BFormatMixKernel<true, false>), mgrid, dim3(64), 0, s, out, lines, bands, gainsHf, gainsLf, nin, nout, n);
    else hipLaunchKernelGGL((BFormatMixKernel<false, false>), mgrid, dim3(64), 0, s, out, lines, bands, gainsHf, gainsLf, nin, nout, n);
}

uint32_t DitherAdvanceSeed(uint32_t seed, uint32_t draws)
{   // host copy of DitherJump
    uint32_t accA = 1u, accC = 0u, a = 96314165u, c = 907633515u, k = draws;
    while(k)
    {
        if(k & 1u) { accC = a * accC + c; accA = a * accA; }
        c = a * c + c; a = a * a;
        k >>= 1;
    }
    return accA * seed + accC;
}

void LaunchDither(hipStream_t s, float *lines, uint32_t nlines, uint32_t n, float quantScale, uint32_t seed)
{ hipLaunchKernelGGL(DitherKernel, dim3((n + 255u) / 256u, nlines), dim3(256), 0, s, lines, nlines, n, quantScale, seed); }

void LaunchWriteSamples(hipStream_t s, int sampleType, const float *lines, uint32_t nlines, uint32_t n, uint32_t frameStep, void *out)
{
    const dim3 grid((n * frameStep + 255u) / 256u), block(256);
    switch(sampleType)
    {
    case 0: hipLaunchKernelGGL(WriteKernel<int8_t>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<int8_t*>(out)); break;
    case 1: hipLaunchKernelGGL(WriteKernel<uint8_t>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<uint8_t*>(out)); break;
    case 2: hipLaunchKernelGGL(WriteKernel<int16_t>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<int16_t*>(out)); break;
    case 3: hipLaunchKernelGGL(WriteKernel<uint16_t>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<uint16_t*>(out)); break;
    case 4: hipLaunchKernelGGL(WriteKernel<int32_t>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<int32_t*>(out)); break;
    case 5: hipLaunchKernelGGL(WriteKernel<uint32_t>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<uint32_t*>(out)); break;
    default: hipLaunchKernelGGL(WriteKernel<float>, grid, block, 0, s, lines, nlines, n, frameStep, static_cast<float*>(out)); break;
    }
}

} // namespace oalgpu
