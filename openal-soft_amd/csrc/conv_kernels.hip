// Convolution reverb: ConvolutionState::process, alc/effects/convolution.cpp:623-716, for a mono
// impulse response (mChans.size() == 1), replacing the pffft calls (common/pffft.cpp:1476-1552
// transform, :2115-2166 zconvolve_accumulate) with a wavefront FFT in LDS.
//
// The reference walks the update in 128-sample blocks: a 128-tap time-domain FIR over the newest
// input (apply_fir :205-250) plus, one block late, the inverse FFT of
//     Y_t = sum_s  X_{t-s} * H_{s+1}          (X_t = FFT of input block t zero-padded to 256,
//                                               H_s  = spectrum of impulse-response taps 128s..128s+127)
// overlap-added across blocks (:644-652, :695-707).  Nothing in that recurrence depends on the
// OUTPUT, so one update is three launches:
//   ConvSpectraKernel  one wavefront per input block completed in this update: real FFT-256 into
//                      the spectrum ring (mComplexData, indexed by curseg counting down, :710);
//   ConvMacKernel      the sum over segments -- the only part with real traffic (2 x 511 KiB per
//                      update for a 65 536-tap response): segment chunks x 128 bins, every filter
//                      spectrum read once and reused for all (<= 8) blocks of the update through a
//                      sliding register window of input spectra; chunk partials in a fixed order;
//   ConvOutputKernel   one workgroup: chunk sum, inverse FFT per block, overlap-add chain, the
//                      128-tap FIR, then MixSamples into the target lines with the
//                      Current -> Target gain ramp over the update (NormalMix :298-304).
// Spectra use pffft's packing idea in plain order: float2[128], bin 0 = (DC, Nyquist), bins
// 1..127 complex.  The filter spectra are computed in double on the host and pre-scaled by 1/256
// (:444-457); the forward/inverse pair here is unnormalised like pffft (inverse(forward(x)) = 256 x).
#include "dev_wave.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kSeg = 128;                 // ConvolveUpdateSamples
constexpr int kMaxBlocks = 8;             // blocks completed by one update of <= 1024 samples

__device__ __forceinline__ f2 cmul(f2 a, f2 b) { return f2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ f2 conj2(f2 a) { return f2{a.x, -a.y}; }

// In-place complex FFT of 128 points in LDS by one wavefront (radix-2 decimation in time, one
// butterfly per lane and stage).  tw[k] = exp(-2 pi i k / 128), k < 64.  INV conjugates them.
template<bool INV>
__device__ __forceinline__ void WaveFft128(f2 *buf, const f2 *tw, uint32_t lane)
{
    // bit-reversal permutation (7 bits): each lane owns elements lane and lane + 64
    {
        const uint32_t i0 = lane, i1 = lane + 64u;
        const uint32_t r0 = __brev(i0) >> 25, r1 = __brev(i1) >> 25;
        const f2 a = buf[i0], b = buf[i1];
        WaveSync();
        buf[r0] = a; buf[r1] = b;
        WaveSync();
    }
#pragma unroll
    for(int lg = 1; lg <= 7; ++lg)
    {
        const uint32_t half = 1u << (lg - 1);
        const uint32_t j = lane & (half - 1u);
        const uint32_t i0 = ((lane >> (lg - 1)) << lg) + j;
        const uint32_t i1 = i0 + half;
        f2 w = tw[j << (7 - lg)];
        if(INV) w = conj2(w);
        const f2 a = buf[i0], t = cmul(w, buf[i1]);
        buf[i0] = f2{a.x + t.x, a.y + t.y};
        buf[i1] = f2{a.x - t.x, a.y - t.y};
        WaveSync();
    }
}

// Real FFT of 256 samples (x[2n], x[2n+1] packed as z[n]) -> packed spectrum in `spec`.
// tw256[k] = exp(-2 pi i k / 256), k < 128.  z is clobbered.
__device__ __forceinline__ void RealFft256Forward(f2 *z, f2 *spec, const f2 *tw128, const f2 *tw256, uint32_t lane)
{
    WaveFft128<false>(z, tw128, lane);
    for(uint32_t k = lane; k < 128u; k += 64)
    {
        const f2 a = z[k], b = conj2(z[(128u - k) & 127u]);
        const f2 xe = f2{0.5f * (a.x + b.x), 0.5f * (a.y + b.y)};
        const f2 d = f2{0.5f * (a.x - b.x), 0.5f * (a.y - b.y)};
        const f2 xo = f2{d.y, -d.x};                      // d / i
        if(k == 0) spec[0] = f2{xe.x + xo.x, xe.x - xo.x}; // DC, Nyquist (both real)
        else
        {
            const f2 t = cmul(tw256[k], xo);
            spec[k] = f2{xe.x + t.x, xe.y + t.y};
        }
    }
}

// Inverse of the above, unnormalised (256 x): packed spectrum -> z[n] = (x[2n], x[2n+1]).
__device__ __forceinline__ void RealFft256Inverse(const f2 *spec, f2 *z, const f2 *tw128, const f2 *tw256, uint32_t lane)
{
    for(uint32_t k = lane; k < 128u; k += 64)
    {
        f2 xk, xm;                                        // X[k], conj(X[128-k])
        if(k == 0) { xk = f2{spec[0].x, 0.0f}; xm = f2{spec[0].y, 0.0f}; }
        else { xk = spec[k]; xm = conj2(spec[128u - k]); }
        const f2 s = f2{xk.x + xm.x, xk.y + xm.y};
        const f2 d = f2{xk.x - xm.x, xk.y - xm.y};
        const f2 t = cmul(conj2(tw256[k]), d);            // e^{+2 pi i k/256} (X[k] - conj X[128-k])
        z[k] = f2{s.x - t.y, s.y + t.x};                  // s + i t
    }
    WaveSync();
    WaveFft128<true>(z, tw128, lane);
}

struct ConvLayout {
    uint32_t numSegs, ringSlots, nlines, n; // frequency-domain segments, ring slots (numSegs + 8), target lines, samples
    uint32_t fifoPos, curSeg, numBlocks;  // mFifoPos / mCurrentSegment at entry; blocks completed by this update
    uint32_t numChunks, segsPerChunk;
    const float *wetIn;                   // channel 0 of the slot's wet bus (n samples)
    float *xhist;                         // the 256 input samples before this update
    f2 *ring;                             // [numSegs][128] input spectra (mComplexData head)
    const f2 *filt;                       // [numSegs][128] filter spectra (mComplexData tail), pre-scaled
    const float *fir;                     // taps 0..127 of the response
    float *outFifo;                       // mOutput[0]: [0,128) pending, [128,256) saved second half
    f2 *partial;                          // [numChunks][kMaxBlocks][128] chunk sums
    float *cur;                           // Current gains [nlines]
    const float *tgt;                     // Target gains [nlines]
    float *outLines;                      // nlines x 1024, accumulated into
    const f2 *tw128, *tw256;
};

// timeline sample i: i < 256 -> history, else this update's input
__device__ __forceinline__ float Timeline(const ConvLayout &C, uint32_t i)
{ return i < 256u ? C.xhist[i] : C.wetIn[i - 256u]; }

__global__ void __launch_bounds__(64) ConvSpectraKernel(ConvLayout C)
{
    __shared__ f2 z[128];
    __shared__ f2 spec[128];
    const uint32_t lane = threadIdx.x, t = blockIdx.x;
    // block t of this update starts at timeline index 256 - fifoPos + 128 t and is zero-padded to 256
    const uint32_t start = 256u - C.fifoPos + kSeg * t;
    z[lane] = f2{Timeline(C, start + 2u * lane), Timeline(C, start + 2u * lane + 1u)};
    z[lane + 64] = f2{0.0f, 0.0f};
    WaveSync();
    RealFft256Forward(z, spec, C.tw128, C.tw256, lane);
    WaveSync();
    // curseg counts down; the ring has 8 slots more than there are segments, so the spectra this
    // update writes never replace one that an earlier block of the same update still reads
    const uint32_t seg = (C.curSeg + C.ringSlots - t) % C.ringSlots;
    C.ring[size_t{seg} * 128 + lane] = spec[lane];
    C.ring[size_t{seg} * 128 + lane + 64] = spec[lane + 64];
}

// Y_t = sum_{i<S} ring[(c_t + i) mod R] * filt[i],  c_t = curSeg - t (mod R); chunk = segments
// [i0, i0 + segsPerChunk); thread = one packed bin.
__global__ void __launch_bounds__(128) ConvMacKernel(ConvLayout C)
{
    const uint32_t f = threadIdx.x, chunk = blockIdx.x;
    const uint32_t S = C.numSegs, R = C.ringSlots, K = C.numBlocks;
    const uint32_t i0 = chunk * C.segsPerChunk;
    const uint32_t i1 = (i0 + C.segsPerChunk < S) ? i0 + C.segsPerChunk : S;
    f2 acc[kMaxBlocks];
    f2 xs[kMaxBlocks];                          // xs[t] = ring[(curSeg - t + i) mod S] for the current i
#pragma unroll
    for(int t = 0; t < kMaxBlocks; ++t) { acc[t] = f2{0.0f, 0.0f}; xs[t] = f2{0.0f, 0.0f}; }
    if(i0 < i1)
    {
        // state "before i0": xs[u] = ring[curSeg - u + i0 - 1]; the loop shifts it into place
#pragma unroll
        for(int u = 0; u + 1 < kMaxBlocks; ++u)
            if(uint32_t(u) + 1u < K) xs[u] = C.ring[size_t{(C.curSeg + i0 + R - uint32_t(u) - 1u) % R} * 128 + f];
    }
    for(uint32_t i = i0; i < i1; ++i)
    {
        // xs[t] = ring[(curSeg - t + i) mod R]: one new spectrum per segment, the others slide
#pragma unroll
        for(int t = kMaxBlocks - 1; t >= 1; --t) xs[t] = xs[t - 1];
        xs[0] = C.ring[size_t{(C.curSeg + i) % R} * 128 + f];
        const f2 h = C.filt[size_t{i} * 128 + f];
#pragma unroll
        for(int t = 0; t < kMaxBlocks; ++t)
        {
            if(uint32_t(t) >= K) continue;
            const f2 x = xs[t];
            if(f == 0) { acc[t].x = __builtin_fmaf(x.x, h.x, acc[t].x); acc[t].y = __builtin_fmaf(x.y, h.y, acc[t].y); }
            else
            {
                acc[t].x = __builtin_fmaf(x.x, h.x, __builtin_fmaf(-x.y, h.y, acc[t].x));
                acc[t].y = __builtin_fmaf(x.x, h.y, __builtin_fmaf(x.y, h.x, acc[t].y));
            }
        }
    }
#pragma unroll
    for(int t = 0; t < kMaxBlocks; ++t)
        if(uint32_t(t) < K) C.partial[(size_t{chunk} * kMaxBlocks + t) * 128 + f] = acc[t];
}

__global__ void __launch_bounds__(1024) ConvOutputKernel(ConvLayout C)
{
    __shared__ f2 zbuf[kMaxBlocks][128];       // per block: spectrum, then z = (x[2n], x[2n+1]) * 256/256
    __shared__ f2 spec[kMaxBlocks][128];
    __shared__ float tl[256 + kLine];          // timeline
    __shared__ float chan[kLine];
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t K = C.numBlocks, n = C.n, p0 = C.fifoPos;

    for(uint32_t i = t; i < 256u + n; i += blockDim.x) tl[i] = Timeline(C, i);
    if(wave < K)
    {
        for(uint32_t f = lane; f < 128u; f += 64)
        {
            f2 s = {0.0f, 0.0f};
            for(uint32_t c = 0; c < C.numChunks; ++c)
            {
                const f2 p = C.partial[(size_t{c} * kMaxBlocks + wave) * 128 + f];
                s.x += p.x; s.y += p.y;
            }
            spec[wave][f] = s;
        }
        WaveSync();
        RealFft256Inverse(spec[wave], zbuf[wave], C.tw128, C.tw256, lane);
    }
    __syncthreads();
    // ifft_b[q] = ((float*)zbuf[b])[q], q < 256.  Sample i of the update sits at fifo offset
    // q = (p0 + i) mod 128 of block b = (p0 + i) / 128 counted from the first block touched.
    if(t < n)
    {
        const uint32_t a = p0 + t, b = a >> 7, q = a & 127u;
        float v;
        if(b == 0) v = C.outFifo[q];                                   // pending output of earlier updates
        else
        {
            const float *cur = reinterpret_cast<const float*>(zbuf[b - 1]);
            const float prevHalf = (b >= 2) ? reinterpret_cast<const float*>(zbuf[b - 2])[128 + q] : C.outFifo[128 + q];
            v = cur[q] + prevHalf;
        }
        // apply_fir: the first 128 taps over the newest input
        float acc0 = 0.0f, acc1 = 0.0f;
        const float *x = tl + 256 + t;
#pragma unroll 8
        for(int k = 0; k < kSeg; k += 2)
        {
            acc0 = __builtin_fmaf(C.fir[k], x[-k], acc0);
            acc1 = __builtin_fmaf(C.fir[k + 1], x[-k - 1], acc1);
        }
        chan[t] = (acc0 + acc1) + v;
    }
    __syncthreads();
    // state for the next update
    float nv = 0.0f;
    if(K > 0 && t < 256u)
    {
        const float *last = reinterpret_cast<const float*>(zbuf[K - 1]);
        if(t < 128u)
            nv = last[t] + ((K >= 2) ? reinterpret_cast<const float*>(zbuf[K - 2])[128 + t] : C.outFifo[128 + t]);
        else nv = last[t];
    }
    __syncthreads();                           // every read of the old fifo is done
    if(K > 0 && t < 256u) C.outFifo[t] = nv;
    if(t < 256u) C.xhist[t] = tl[n + t];
    // MixSamples(chan, out, Current, Target, Counter = n, OutPos = 0)
    for(uint32_t c = 0; c < C.nlines; ++c)
    {
        const MixLineGain g = PrepareMixLine(C.cur[c], C.tgt[c], n, n);
        if(t < n && MixLineActive(g, t))
        {
            float *o = C.outLines + size_t{c} * kLine + t;
            *o = *o + MixLineValue(g, chan[t], t);
        }
        __syncthreads();
        if(t == 0) C.cur[c] = g.newCur;
    }
}

} // namespace

void LaunchConvolution(hipStream_t s, const ConvLayoutHost &h)
{
    ConvLayout C{};
    C.numSegs = h.numSegs; C.ringSlots = h.ringSlots; C.nlines = h.nlines; C.n = h.n;
    C.fifoPos = h.fifoPos; C.curSeg = h.curSeg; C.numBlocks = h.numBlocks;
    C.numChunks = h.numChunks; C.segsPerChunk = h.segsPerChunk;
    C.wetIn = h.wetIn; C.xhist = h.xhist;
    C.ring = reinterpret_cast<f2*>(h.ring); C.filt = reinterpret_cast<const f2*>(h.filt);
    C.fir = h.fir; C.outFifo = h.outFifo; C.partial = reinterpret_cast<f2*>(h.partial);
    C.cur = h.cur; C.tgt = h.tgt; C.outLines = h.outLines;
    C.tw128 = reinterpret_cast<const f2*>(h.tw128); C.tw256 = reinterpret_cast<const f2*>(h.tw256);
    if(h.numBlocks)
    {
        hipLaunchKernelGGL(ConvSpectraKernel, dim3(h.numBlocks), dim3(64), 0, s, C);
        hipLaunchKernelGGL(ConvMacKernel, dim3(h.numChunks), dim3(128), 0, s, C);
    }
    hipLaunchKernelGGL(ConvOutputKernel, dim3(1), dim3(1024), 0, s, C);
}

} // namespace oalgpu
