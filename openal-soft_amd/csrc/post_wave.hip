// HRTF post-process of one update, FAST mode: DeviceBase::Process(HrtfPostProcess),
// alc/alu.cpp:289-298 -> MixDirectHrtf_* (MixDirectHrtfBase, core/mixer/hrtfbase.h:91-133).
//
// One workgroup, one wavefront per dry (ambisonic) channel -- channels wrap around when there are
// more than 16.  Per channel:
//   1. BandSplitter::processHfScale (core/filters/splitter.cpp:65-97) is a 3-state linear
//      recurrence; the wavefront runs it as a block scan (each lane owns a run of samples, block
//      start states by a 6-step Kogge-Stone scan over the lanes) instead of 1024 serial steps;
//   2. the channel's decoder HRIR (same input for both ears) is applied with the ear-packed
//      v_pk_fma_f32 FIR of voice_wave.hip: lane l owns output frames [18l, 18l+18) of the
//      1152-frame accumulator, coefficients come through the scalar cache.
// Then the per-wave accumulators are summed in channel order together with the carried
// HrtfAccumData, the first n frames are added to RealOut L/R, and the accumulator is shifted
// (hrtfbase.h:119-132).  EXACT mode keeps the term-by-term kernel in percall_kernels.hip.
#include "dev_wave.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kPostMaxWaves = 16;
constexpr int kPostR = 18;                                  // 64 * 18 = 1152 = kLine + kHrirLen
constexpr int kPostFrames = kLine + kHrirLen;

struct PostWaveLds {
    union {
        float x[128 + kPostFrames];             // x[k] = filtered channel sample k - 128, zero padded
        f2 dump[kPostFrames];                   // end of kernel: this wave's accumulator
    };
};

template<int TAPS>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(kPostMaxWaves * 64) PostDirectHrtfKernel(float *__restrict__ left,
    float *__restrict__ right, const float *__restrict__ in, uint32_t nch, float *__restrict__ accum,
    SplitterState *__restrict__ splitters, const float *__restrict__ hfscales, const float *__restrict__ chanCoeffs,
    uint32_t n)
{
    __shared__ PostWaveLds sm[kPostMaxWaves];
    const uint32_t t = threadIdx.x, lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t nwaves = blockDim.x >> 6;
    PostWaveLds &w = sm[wave];

    f2 acc[kPostR];
#pragma unroll
    for(int r = 0; r < kPostR; ++r) acc[r] = f2{0.0f, 0.0f};

    for(uint32_t c = wave; c < nch; c += nwaves)
    {
        WaveSync();
        for(uint32_t k = lane; k < 128u; k += 64) w.x[k] = 0.0f;
        for(uint32_t k = lane; k < uint32_t(kPostFrames); k += 64) w.x[128 + k] = (k < n) ? in[size_t{c} * kLine + k] : 0.0f;
        WaveSync();
        SplitterState st = splitters[c];
        SplitterScan<true>(st, w.x + 128, n, hfscales[c], 1.0f, lane);
        if(lane == 0) splitters[c] = st;
        WaveSync();
        cf16 *co = (cf16*)(uintptr_t)(chanCoeffs + size_t{c} * kHrirLen * 2);
        FirMainPk<kPostR, TAPS>(acc, &w.x[128 + kPostR * lane], co);
    }

    WaveSync();
#pragma unroll
    for(int r = 0; r < kPostR; ++r) w.dump[kPostR * lane + r] = acc[r];
    __syncthreads();
    // total[o] = carried accumulator + channel contributions (wave order); outputs and the shift
    const f2 *acc2 = reinterpret_cast<const f2*>(accum);
    f2 *accOut = reinterpret_cast<f2*>(accum);
    f2 tot[2];
    uint32_t oidx[2];
    int cnt = 0;
    for(uint32_t o = t; o < uint32_t(kPostFrames); o += blockDim.x)
    {
        f2 s = acc2[o];
        for(uint32_t ww = 0; ww < nwaves && ww < nch; ++ww) { const f2 v = sm[ww].dump[o]; s.x += v.x; s.y += v.y; }
        if(o < n) { left[o] = left[o] + s.x; right[o] = right[o] + s.y; }
        tot[cnt] = s; oidx[cnt] = o; ++cnt;
    }
    __syncthreads();                           // every thread has read the old accumulator
    // hrtfbase.h:127-132: frames [n, n+128) move to the front, the following n frames are cleared,
    // anything beyond stays
    for(int k = 0; k < cnt; ++k)
    {
        const uint32_t o = oidx[k];
        if(o >= n && o < n + kHrirLen) accOut[o - n] = tot[k];
    }
    __syncthreads();
    for(int k = 0; k < cnt; ++k)
    {
        const uint32_t o = oidx[k];
        if(o >= uint32_t(kHrirLen) && o < kHrirLen + n) accOut[o] = f2{0.0f, 0.0f};
    }
}

} // namespace

void LaunchPostDirectHrtfFast(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, float *accum,
    SplitterState *splitters, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n)
{
    const uint32_t waves = nch < uint32_t(kPostMaxWaves) ? nch : uint32_t(kPostMaxWaves);
    // at least as many threads as needed to cover the 1152 frames in two passes
    const uint32_t threads = (waves < 9u ? 9u : waves) * 64u;
    if(irsize <= 64)
        hipLaunchKernelGGL(PostDirectHrtfKernel<64>, dim3(1), dim3(threads), 0, s, left, right, in, nch, accum, splitters,
            hfscales, chanCoeffs, n);
    else
        hipLaunchKernelGGL(PostDirectHrtfKernel<128>, dim3(1), dim3(threads), 0, s, left, right, in, nch, accum, splitters,
            hfscales, chanCoeffs, n);
}

} // namespace oalgpu
