// Convolution reverb: ConvolutionState::process, alc/effects/convolution.cpp:623-716, for impulse responses
// of 1 .. 8 channels (mono, stereo .. 7.1, first-order B-Format: one ChannelData each, all fed from the slot's
// one input channel), replacing the pffft calls (common/pffft.cpp:1476-1552
// transform, :2115-2166 zconvolve_accumulate) with a wavefront FFT in LDS.
//
// The reference walks the update in 128-sample blocks: a 128-tap time-domain FIR over the newest
// input (apply_fir :205-250) plus, one block late, the inverse FFT of
//     Y_t = sum_s  X_{t-s} * H_{s+1}          (X_t = FFT of input block t zero-padded to 256,
//                                               H_s  = spectrum of impulse-response taps 128s..128s+127)
// overlap-added across blocks (:644-652, :695-707).  Nothing in that recurrence depends on the
// OUTPUT, so one update is ONE launch of numChunks workgroups of eight wavefronts (ConvFusedKernel):
//   spectra   every workgroup transforms the update's <= 8 completed input blocks itself, a wavefront per
//             block (real FFT-256 in LDS; a few microseconds of redundant work instead of a launch and an
//             HBM round trip); workgroup 0 also files them in the spectrum ring (mComplexData, indexed by
//             curseg counting down, :710) for later updates;
//   MAC       the sum over the workgroup's chunk of segments -- the only part with real traffic (2 x 511 KiB
//             per update for a 65 536-tap response): 128 bins x 4 pairs of blocks per workgroup, a pair's two
//             input spectra sliding through registers so that every segment costs one filter and one input
//             load; the chunk's partial sums go to HBM;
//   output    the LAST workgroup to finish (a ticket counter behind a device-scope fence) adds the chunk
//             partials in chunk order, runs the inverse FFT per block, the overlap-add chain and the 128-tap
//             FIR, then MixSamples into the target lines with the Current -> Target gain ramp over the
//             update (NormalMix :298-304).  Which workgroup that is does not change a bit of the result.
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
    uint32_t channels;                    // mChans.size(); blockIdx.y
    uint32_t upsample;                    // mMix == UpsampleMix: band-split + HF/LF scale per channel before the mix
    const float *wetIn;                   // channel 0 of the slot's wet bus (n samples)
    float *xhist;                         // the 256 input samples before this update
    f2 *ring;                             // [numSegs][128] input spectra (mComplexData head)
    const f2 *filt;                       // [channel][numSegs][128] filter spectra (mComplexData tail), pre-scaled
    const float *fir;                     // [channel][128] taps 0..127 of the response
    float *outFifo;                       // [channel][256] mOutput[c]: [0,128) pending, [128,256) saved second half
    f2 *partial;                          // [channel][numChunks][kMaxBlocks][128] chunk sums
    float *cur;                           // [channel][nlines] Current gains
    const float *tgt;                     // [channel][nlines] Target gains
    float *outLines;                      // nlines x 1024, accumulated into
    const f2 *tw128, *tw256;
    uint32_t *ticket;                     // [channel] workgroups that have delivered their chunk, [8]: channels whose
                                          // output is ready (all 0 between launches)
    float *firOut;                        // [channel][1024] apply_fir of the update, computed in slices by all workgroups
    float *chanOut;                       // [channel][1024] mChans[c].mBuffer of the update (channels > 1)
    SplitterState *split;                 // [channel] mChans[c].mFilter
    const float *hfScale, *lfScale;       // [channel] mHfScale / mLfScale
};

// timeline sample i: i < 256 -> history, else this update's input
__device__ __forceinline__ float Timeline(const ConvLayout &C, uint32_t i)
{ return i < 256u ? C.xhist[i] : C.wetIn[i - 256u]; }

constexpr uint32_t kConvThreads = 512;

// The chunk partials and the FIR slices travel from the workgroups that make them to the last one through
// device-scope (agent) relaxed atomics: write-through stores and L2-coherent loads of exactly those words.
// A release/acquire fence pair would do too, but at device scope it writes back and invalidates the WHOLE L2
// of the XCD -- while the next update's voice kernel is filling it with stream rows on the other stream.
__device__ __forceinline__ void StoreCoherent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float LoadCoherent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void StoreCoherent(f2 *p, f2 v)
{ StoreCoherent(reinterpret_cast<float*>(p), v.x); StoreCoherent(reinterpret_cast<float*>(p) + 1, v.y); }
__device__ __forceinline__ f2 LoadCoherent(const f2 *p)
{ return f2{LoadCoherent(reinterpret_cast<const float*>(p)), LoadCoherent(reinterpret_cast<const float*>(p) + 1)}; }

__global__ void __launch_bounds__(kConvThreads) ConvFusedKernel(ConvLayout C)
{
    __shared__ f2 zbuf[kMaxBlocks][128];       // FFT work space; output stage: z = (x[2n], x[2n+1]) per block
    __shared__ f2 spec[kMaxBlocks][128];       // this update's input spectra; output stage: the summed Y_t
    __shared__ float tl[256 + kLine];          // timeline
    __shared__ float chan[kLine];
    __shared__ float firLds[kSeg];
    __shared__ f2 tw128[64], tw256[128];       // the FFTs read a twiddle per butterfly stage: LDS, not HBM latency
    __shared__ uint32_t isLast;
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t K = C.numBlocks, n = C.n, p0 = C.fifoPos, R = C.ringSlots;
    const uint32_t ch = blockIdx.y;            // the impulse response's channel this workgroup convolves with
    const f2 *filt = C.filt + size_t{ch} * C.numSegs * 128;
    const float *fir = C.fir + size_t{ch} * kSeg;
    float *outFifo = C.outFifo + size_t{ch} * 256;
    f2 *partial = C.partial + size_t{ch} * C.numChunks * kMaxBlocks * 128;
    float *firOut = C.firOut + size_t{ch} * kLine;

    for(uint32_t i = t; i < 256u + n; i += kConvThreads) tl[i] = Timeline(C, i);
    if(t < uint32_t(kSeg)) { firLds[t] = fir[t]; tw256[t] = C.tw256[t]; }
    if(t < 64u) tw128[t] = C.tw128[t];
    __syncthreads();
    if(K)
    {
        // ---- input spectra: block `wave` of this update starts at timeline index 256 - fifoPos + 128 wave
        // and is zero-padded to 256
        if(wave < K)
        {
            const uint32_t start = 256u - p0 + kSeg * wave;
            zbuf[wave][lane] = f2{tl[start + 2u * lane], tl[start + 2u * lane + 1u]};
            zbuf[wave][lane + 64] = f2{0.0f, 0.0f};
            WaveSync();
            RealFft256Forward(zbuf[wave], spec[wave], tw128, tw256, lane);
            WaveSync();
            if(blockIdx.x == 0 && ch == 0)
            {   // curseg counts down; the ring has 8 slots more than there are segments, so the spectra this
                // update files never replace one that a block of the same update still reads
                const uint32_t seg = (C.curSeg + R - wave) % R;
                C.ring[size_t{seg} * 128 + lane] = spec[wave][lane];
                C.ring[size_t{seg} * 128 + lane + 64] = spec[wave][lane + 64];
            }
        }
        __syncthreads();
        // ---- Y_t = sum_{i<S} X(t, i) * filt[i],  X(t, i) = ring[(curSeg - t + i) mod R] -- block t - i of THIS
        // update while i <= t (still in LDS), an older one from the ring otherwise.  Thread = one packed bin
        // of the blocks 2q and 2q + 1: X(2q + 1, i) = X(2q, i - 1) slides through a register.
        const uint32_t f = t & 127u, q = t >> 7;
        const uint32_t tb = 2u * q;
        const uint32_t i0 = blockIdx.x * C.segsPerChunk;
        const uint32_t i1 = (i0 + C.segsPerChunk < C.numSegs) ? i0 + C.segsPerChunk : C.numSegs;
        auto X = [&](uint32_t tt, uint32_t i) -> f2
        {
            if(i <= tt) return spec[tt - i][f];
            return C.ring[size_t{(C.curSeg + i + R - tt) % R} * 128 + f];
        };
        f2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};
        if(tb < K && i0 < i1)
        {
            auto mac = [&](f2 x0, f2 x1, f2 h)
            {
                if(f == 0)
                {   // bin 0 = (DC, Nyquist): two real products
                    acc0.x = __builtin_fmaf(x0.x, h.x, acc0.x); acc0.y = __builtin_fmaf(x0.y, h.y, acc0.y);
                    acc1.x = __builtin_fmaf(x1.x, h.x, acc1.x); acc1.y = __builtin_fmaf(x1.y, h.y, acc1.y);
                }
                else
                {
                    acc0.x = __builtin_fmaf(x0.x, h.x, __builtin_fmaf(-x0.y, h.y, acc0.x));
                    acc0.y = __builtin_fmaf(x0.x, h.y, __builtin_fmaf(x0.y, h.x, acc0.y));
                    acc1.x = __builtin_fmaf(x1.x, h.x, __builtin_fmaf(-x1.y, h.y, acc1.x));
                    acc1.y = __builtin_fmaf(x1.x, h.y, __builtin_fmaf(x1.y, h.x, acc1.y));
                }
            };
            f2 x1 = X(tb + 1u < K ? tb + 1u : tb, i0);
            uint32_t i = i0;
            // the segments that still meet this update's own blocks (first chunk only)
            for(; i < i1 && i <= tb + 1u; ++i) { const f2 x0 = X(tb, i); mac(x0, x1, filt[size_t{i} * 128 + f]); x1 = x0; }
            // the rest straight from the ring: eight segments' loads in flight
            uint32_t slot = (C.curSeg + i + R - tb) % R;
            for(; i + 8u <= i1; i += 8u)
            {
                f2 xs[8], hs[8];
#pragma unroll
                for(uint32_t k = 0; k < 8u; ++k)
                {
                    xs[k] = C.ring[size_t{slot} * 128 + f];
                    hs[k] = filt[size_t{i + k} * 128 + f];
                    slot = (slot + 1u == R) ? 0u : slot + 1u;
                }
#pragma unroll
                for(uint32_t k = 0; k < 8u; ++k) { mac(xs[k], x1, hs[k]); x1 = xs[k]; }
            }
            for(; i < i1; ++i) { const f2 x0 = X(tb, i); mac(x0, x1, filt[size_t{i} * 128 + f]); x1 = x0; }
        }
        if(tb < K) StoreCoherent(&partial[(size_t{blockIdx.x} * kMaxBlocks + tb) * 128 + f], acc0);
        if(tb + 1u < K) StoreCoherent(&partial[(size_t{blockIdx.x} * kMaxBlocks + tb + 1u) * 128 + f], acc1);
        // apply_fir (the first 128 taps over the newest input) of this workgroup's slice of the update
        {
            const uint32_t per = (n + gridDim.x - 1u) / gridDim.x;
            for(uint32_t j = t; j < per && blockIdx.x * per + j < n; j += kConvThreads)
            {
                const uint32_t i = blockIdx.x * per + j;
                float acc0 = 0.0f, acc1 = 0.0f;
                const float *x = tl + 256 + i;
#pragma unroll 8
                for(int k = 0; k < kSeg; k += 2)
                {
                    acc0 = __builtin_fmaf(firLds[k], x[-k], acc0);
                    acc1 = __builtin_fmaf(firLds[k + 1], x[-k - 1], acc1);
                }
                StoreCoherent(&firOut[i], acc0 + acc1);
            }
        }
        // ---- the last workgroup to deliver does the rest (the barrier waits for every thread's stores)
        __syncthreads();
        if(t == 0) isLast = (__hip_atomic_fetch_add(C.ticket + ch, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1u : 0u;
        __syncthreads();
        if(!isLast) return;
        if(t == 0) C.ticket[ch] = 0u;
        // the chunk partials in chunk order: thread = one bin of two blocks, 32 chunks' loads in flight
        for(uint32_t w = tb; w < tb + 2u && w < K; ++w)
        {
            f2 s = {0.0f, 0.0f};
            uint32_t c = 0;
            for(; c + 32u <= C.numChunks; c += 32u)
            {
                f2 p[32];
#pragma unroll
                for(uint32_t k = 0; k < 32u; ++k) p[k] = LoadCoherent(&partial[(size_t{c + k} * kMaxBlocks + w) * 128 + f]);
#pragma unroll
                for(uint32_t k = 0; k < 32u; ++k) { s.x += p[k].x; s.y += p[k].y; }
            }
            for(; c < C.numChunks; ++c)
            {
                const f2 p = LoadCoherent(&partial[(size_t{c} * kMaxBlocks + w) * 128 + f]);
                s.x += p.x; s.y += p.y;
            }
            spec[w][f] = s;
        }
        __syncthreads();
        if(wave < K) RealFft256Inverse(spec[wave], zbuf[wave], tw128, tw256, lane);
        __syncthreads();
    }
    // ifft_b[q] = ((float*)zbuf[b])[q], q < 256.  Sample i of the update sits at fifo offset
    // q = (p0 + i) mod 128 of block b = (p0 + i) / 128 counted from the first block touched.
    for(uint32_t i = t; i < n; i += kConvThreads)
    {
        const uint32_t a = p0 + i, b = a >> 7, q = a & 127u;
        float v;
        if(b == 0) v = outFifo[q];                                     // pending output of earlier updates
        else
        {
            const float *cur = reinterpret_cast<const float*>(zbuf[b - 1]);
            const float prevHalf = (b >= 2) ? reinterpret_cast<const float*>(zbuf[b - 2])[128 + q] : outFifo[128 + q];
            v = cur[q] + prevHalf;
        }
        float fir;
        if(K) fir = LoadCoherent(&firOut[i]);
        else
        {   // no block completes in this update (a single workgroup): apply_fir here
            float acc0 = 0.0f, acc1 = 0.0f;
            const float *x = tl + 256 + i;
#pragma unroll 8
            for(int k = 0; k < kSeg; k += 2)
            {
                acc0 = __builtin_fmaf(firLds[k], x[-k], acc0);
                acc1 = __builtin_fmaf(firLds[k + 1], x[-k - 1], acc1);
            }
            fir = acc0 + acc1;
        }
        chan[i] = fir + v;
    }
    __syncthreads();
    // state for the next update
    float nv = 0.0f;
    if(K > 0 && t < 256u)
    {
        const float *last = reinterpret_cast<const float*>(zbuf[K - 1]);
        if(t < 128u)
            nv = last[t] + ((K >= 2) ? reinterpret_cast<const float*>(zbuf[K - 2])[128 + t] : outFifo[128 + t]);
        else nv = last[t];
    }
    __syncthreads();                           // every read of the old fifo is done
    if(K > 0 && t < 256u) outFifo[t] = nv;
    if(C.upsample)
    {   // UpsampleMix (:306-316): mChans[c].mFilter.processScale(src, mHfScale, mLfScale) in front of the mix
        if(wave == 0)
        {
            SplitterState sp = C.split[ch];
            SplitterScan<false>(sp, chan, n, C.hfScale[ch], C.lfScale[ch], lane);
            if(lane == 0) C.split[ch] = sp;
        }
        __syncthreads();
    }
    const uint32_t nch = C.channels;
    if(nch > 1u)
    {   // the channels are mixed in channel order by ONE workgroup: the last one whose channel is ready
        for(uint32_t i = t; i < n; i += kConvThreads) StoreCoherent(&C.chanOut[size_t{ch} * kLine + i], chan[i]);
        __syncthreads();
        if(t == 0) isLast = (__hip_atomic_fetch_add(C.ticket + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nch - 1u) ? 1u : 0u;
        __syncthreads();
        if(!isLast) return;
        if(t == 0) C.ticket[8] = 0u;
    }
    if(t < 256u) C.xhist[t] = tl[n + t];
    // MixSamples(chan_c, out, Current_c, Target_c, Counter = n, OutPos = 0) for c = 0, 1, ..: a line's sum takes the
    // channels in that order; four lines at a time, their gains and their output samples requested together
    for(uint32_t c0 = 0; c0 < C.nlines; c0 += 4u)
    {
        float o[4][2];
#pragma unroll
        for(uint32_t k = 0; k < 4u; ++k)
        {
            const uint32_t c = (c0 + k < C.nlines) ? c0 + k : C.nlines - 1u;
#pragma unroll
            for(uint32_t j = 0; j < 2u; ++j)
            {
                const uint32_t i = t + kConvThreads * j;
                o[k][j] = (i < n) ? C.outLines[size_t{c} * kLine + i] : 0.0f;
            }
        }
        for(uint32_t cc = 0; cc < nch; ++cc)
        {
            float cu[4], tg[4], x[2];
#pragma unroll
            for(uint32_t k = 0; k < 4u; ++k)
            {
                const uint32_t c = (c0 + k < C.nlines) ? c0 + k : C.nlines - 1u;
                cu[k] = C.cur[size_t{cc} * C.nlines + c]; tg[k] = C.tgt[size_t{cc} * C.nlines + c];
            }
#pragma unroll
            for(uint32_t j = 0; j < 2u; ++j)
            {
                const uint32_t i = t + kConvThreads * j;
                x[j] = (i >= n) ? 0.0f : (nch > 1u ? LoadCoherent(&C.chanOut[size_t{cc} * kLine + i]) : chan[i]);
            }
#pragma unroll
            for(uint32_t k = 0; k < 4u; ++k)
            {
                if(c0 + k >= C.nlines) break;
                const MixLineGain g = PrepareMixLine(cu[k], tg[k], n, n);
#pragma unroll
                for(uint32_t j = 0; j < 2u; ++j)
                {
                    const uint32_t i = t + kConvThreads * j;
                    if(i < n && MixLineActive(g, i)) o[k][j] = o[k][j] + MixLineValue(g, x[j], i);
                }
            }
        }
#pragma unroll
        for(uint32_t k = 0; k < 4u; ++k)
        {
            if(c0 + k >= C.nlines) break;
#pragma unroll
            for(uint32_t j = 0; j < 2u; ++j)
            {
                const uint32_t i = t + kConvThreads * j;
                if(i < n) C.outLines[size_t{c0 + k} * kLine + i] = o[k][j];
            }
        }
    }
    __syncthreads();                           // every thread has read the Current gains
    for(uint32_t k = t; k < nch * C.nlines; k += kConvThreads) C.cur[k] = PrepareMixLine(C.cur[k], C.tgt[k], n, n).newCur;
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
    C.ticket = h.ticket; C.firOut = h.firOut;
    C.channels = h.channels; C.upsample = h.upsample; C.chanOut = h.chanOut;
    C.split = h.split; C.hfScale = h.hfScale; C.lfScale = h.lfScale;
    // no block completes in this update: only the output stage (one workgroup per channel)
    hipLaunchKernelGGL(ConvFusedKernel, dim3(h.numBlocks ? h.numChunks : 1u, h.channels), dim3(kConvThreads), 0, s, C);
}

} // namespace oalgpu
