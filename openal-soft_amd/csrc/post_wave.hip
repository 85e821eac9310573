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

struct Sp3 { float a, b, c; };                              // (lpZ1, lpZ2, apZ1)

// one sample of processHfScale on state s; returns the output
__device__ __forceinline__ float SplitStep(Sp3 &s, float x, float apCoeff, float lpCoeff, float hf)
{
    const float d0 = (x - s.a) * lpCoeff;
    const float lpY0 = s.a + d0;
    s.a = __builtin_fmaf(d0, lpCoeff, lpY0);               // sic: splitter.cpp:79
    const float d1 = (lpY0 - s.b) * lpCoeff;
    const float lpY1 = s.b + d1;
    s.b = lpY1 + d1;
    const float apY = __builtin_fmaf(x, apCoeff, s.c);
    s.c = __builtin_fmaf(-apY, apCoeff, x);
    return __builtin_fmaf(apY - lpY1, hf, lpY1);
}

struct Mat3 { Sp3 c0, c1, c2; };
__device__ __forceinline__ Sp3 MatVec3(const Mat3 &m, const Sp3 &v)
{
    Sp3 r;
    r.a = __builtin_fmaf(m.c2.a, v.c, __builtin_fmaf(m.c1.a, v.b, m.c0.a * v.a));
    r.b = __builtin_fmaf(m.c2.b, v.c, __builtin_fmaf(m.c1.b, v.b, m.c0.b * v.a));
    r.c = __builtin_fmaf(m.c2.c, v.c, __builtin_fmaf(m.c1.c, v.b, m.c0.c * v.a));
    return r;
}
__device__ __forceinline__ Mat3 MatMul3(const Mat3 &a, const Mat3 &b)
{ return Mat3{MatVec3(a, b.c0), MatVec3(a, b.c1), MatVec3(a, b.c2)}; }

// in place over buf[0..n); the lane stride (seg) is odd so the per-lane runs hit distinct banks
__device__ __forceinline__ void SplitterScan(SplitterState &st, float *buf, uint32_t n, float hf, uint32_t lane)
{
    const float apCoeff = st.coeff, lpCoeff = st.coeff * 0.5f + 0.5f;
    const uint32_t seg = ((n + 63u) / 64u) | 1u;
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t end = (begin + seg < n) ? begin + seg : n;
    Mat3 M{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for(uint32_t i = 0; i < seg; ++i)
    {
        SplitStep(M.c0, 0.0f, apCoeff, lpCoeff, hf); SplitStep(M.c1, 0.0f, apCoeff, lpCoeff, hf);
        SplitStep(M.c2, 0.0f, apCoeff, lpCoeff, hf);
    }
    Sp3 q{0, 0, 0};
    for(uint32_t i = begin; i < end; ++i) SplitStep(q, buf[i], apCoeff, lpCoeff, hf);
    Sp3 e = q;                                   // -> sum_{k<=l} M^(l-k) q_k
    Sp3 s0{st.lpZ1, st.lpZ2, st.apZ1};           // -> M^lane S_0
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
    for(uint32_t i = begin; i < end; ++i) buf[i] = SplitStep(start, buf[i], apCoeff, lpCoeff, hf);
    const int lastLane = int((n - 1u) / seg);
    st.lpZ1 = __shfl(start.a, lastLane); st.lpZ2 = __shfl(start.b, lastLane); st.apZ1 = __shfl(start.c, lastLane);
}

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
        SplitterScan(st, w.x + 128, n, hfscales[c], lane);
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
