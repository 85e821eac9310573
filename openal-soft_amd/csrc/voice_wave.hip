// The FAST HRTF hot path: one WAVEFRONT per voice (BASELINE configs[2], the headline metric).
//
//   "for every Playing|Stopping voice: voice->mix(...)"   alc/alu.cpp:2201-2206
//      -> Voice::mix                                       core/voice.cpp:988-1233
//      -> LoadResampledSamples / Resample_*                voice.cpp:642-824, mixer_sse.cpp:199-329
//      -> DoFilters (dual biquad)                          voice.cpp:255-267
//      -> DoHrtfMix -> MixHrtf / MixHrtfBlend              voice.cpp:827-902, mixer/hrtfbase.h:17-89
//
// Why a wavefront per voice.  fp32 FMA throughput on gfx950 is only reached by v_pk_fma_f32
// (tools/ubench_fma.hip on MI355X: 124-130 TFLOP/s packed vs 63-100 scalar), and the dual-ear FIR
// is 128 of the 176 FMAs per output sample, so the kernel is built around packed FMAs whose two
// halves are the two EARS: the accumulator pair (L,R) of a frame, the coefficient pair
// (Coeffs[j][0], Coeffs[j][1]) exactly as HrirArray stores it -- fetched through the scalar
// cache into an SGPR pair -- and the input pair x'[i] = (In[64-delayL+i], In[64-delayR+i])*g(i)
// staged once per voice in LDS as float2.  Every operand is a naturally aligned register pair;
// nothing is broadcast, shuffled or re-packed.  Lane l owns R consecutive output frames
// [R*l, R*l+R) of the 64*R-frame accumulator for the whole launch (R = 17: 1088 frames =
// 1024 + IrSize 64; R = 18: 1152 = 1024 + HrirLength for IrSize <= 128), keeps a sliding window
// of R+3 inputs in registers and needs 4 new ds_read_b64 per 4 taps x R frames x 2 ears
// (= 136 FMAs); the lane stride R is odd/coprime to the LDS banking, so those reads are
// conflict-free.  Voices of a wavefront follow each other through the same registers, so bus
// traffic is one partial per WORKGROUP per launch.
//
// The resampler reads its coefficient rows from an LDS copy transposed to [tap pair][phase] so
// that the 32 phases map onto the 64 LDS banks: any per-lane phase pattern is conflict-free
// (ds_read_b64), and (fil + pf*phd) and the tap MACs are packed over tap pairs.
//
// Arithmetic is FAST mode (FMA, own summation order): results agree with the reference within
// the tolerance stated in DESIGN.md; all integer state (positions, loop wrap, play state,
// delays, fade counters) is bit-exact.  EXACT mode and every other configuration (sends, non-HRTF
// buses) run voice_kernel.hip.
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef const f2 __attribute__((address_space(4))) cf2;        // scalar-cache (constant) loads
typedef const uint32_t __attribute__((address_space(4))) cu32;

constexpr int kWWaves = 4;                    // wavefronts (= concurrent voices) per workgroup
constexpr int kWThreads = kWWaves * 64;
constexpr int kTabPairs = 24;                 // staged resampler rows: up to 48 taps

__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }

// Lanes of a wavefront exchange data through LDS in program order (the LDS executes one wave's
// operations in order); this only stops the COMPILER from moving memory operations across.
__device__ __forceinline__ void WaveSync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef const u4 __attribute__((address_space(4))) cu4;

__device__ __forceinline__ VoiceCtl LoadCtlScalar(const VoiceCtl *p)
{
    union { VoiceCtl c; u4 q[sizeof(VoiceCtl) / 16]; } u;
    cu4 *src = (cu4*)(uintptr_t)p;
#pragma unroll
    for(size_t k = 0; k < sizeof(VoiceCtl) / 16; ++k) u.q[k] = src[k];
    return u.c;
}

template<int R, int TAPS>
struct WaveLds {
    static constexpr int kFrames = 64 * R;
    static constexpr int kX = TAPS + kFrames;           // x2[k] = x'[k - TAPS]
    static constexpr int kQ = TAPS / 64 + 1;            // old-filter fade: frames l + 64q, q < kQ
    union {
        f2 x2[kX];                                      // FIR inputs (both ears), zero padded
        float rd[kResampleDataSize + 8];                // DeviceBase::mResampleData (dead before x2 is built)
    };
    float in[kHist + kLine];                            // [Hrtf.History | resampled, filtered samples]
    f2 cold[TAPS + 128];                                // cold[k] = Hrtf.Old.Coeffs[k - 64], zero padded
    f2 xo[64];                                          // old-filter fade-out inputs (i < 64), both ears
    int32_t best;
    uint32_t pad[3];
};

template<int R, int TAPS>
struct WgLds {
    WaveLds<R, TAPS> w[kWWaves];
    f2 tabF[kTabPairs * 32];                            // [tap pair][phase] = fil[2p], fil[2p+1]
    f2 tabP[kTabPairs * 32];                            //                    = phd[2p], phd[2p+1]
    uint32_t tabKey, tabM, tabL;
    uint32_t pad;
};

// ---- resampler, staged rows -----------------------------------------------------------------
// One output sample of Resample_FastBSinc / Resample_Cubic (core/mixer/mixer_c.cpp:52-83,
// the SSE variants mixer_sse.cpp:199-329 compute the same terms): sum_j (fil[j] + pf*phd[j]) *
// src[pos + j - l], taps processed two at a time.
template<int M>
__device__ __forceinline__ float ResampleStaged(const f2 *tabF, const f2 *tabP, const float *rd, uint32_t l,
    uint32_t frac0, uint32_t increment, uint32_t i)
{
    const uint32_t t = frac0 + i * increment;
    const uint32_t pos = t >> kFracBits;
    const uint32_t frac = t & kFracMask;
    const uint32_t pi = frac >> 11;
    const f2 pf = splat(float(frac & 2047u) * (1.0f / 2048.0f));
    const float *s = rd + (kMaxEdge - l + pos);
    f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
#pragma unroll
    for(int p = 0; p < M / 2; p += 2)
    {
        const f2 c0 = pkfma(pf, tabP[p * 32 + pi], tabF[p * 32 + pi]);
        const f2 s0 = {s[2 * p], s[2 * p + 1]};
        r0 = pkfma(c0, s0, r0);
        const f2 c1 = pkfma(pf, tabP[(p + 1) * 32 + pi], tabF[(p + 1) * 32 + pi]);
        const f2 s1 = {s[2 * p + 2], s[2 * p + 3]};
        r1 = pkfma(c1, s1, r1);
    }
    return (r0.x + r0.y) + (r1.x + r1.y);
}

template<int R, int TAPS>
__device__ __forceinline__ float ResampleStagedM(const WgLds<R, TAPS> &sm, const float *rd, uint32_t m, uint32_t l,
    uint32_t frac0, uint32_t increment, uint32_t i)
{
    switch(m)
    {
    case 4: return ResampleStaged<4>(sm.tabF, sm.tabP, rd, l, frac0, increment, i);
    case 12: return ResampleStaged<12>(sm.tabF, sm.tabP, rd, l, frac0, increment, i);
    case 24: return ResampleStaged<24>(sm.tabF, sm.tabP, rd, l, frac0, increment, i);
    default: return ResampleStaged<48>(sm.tabF, sm.tabP, rd, l, frac0, increment, i);
    }
}

// LoadResampledSamples, core/voice.cpp:642-824, for one real channel of a static voice, by one
// wavefront.  Produces samplesToLoad resampled samples at w.in[kHist..]; updates prev[v] when
// Playing.  Integer logic identical to voice_kernel.hip's LoadResampled.
template<int R, int TAPS>
__device__ __forceinline__ void LoadResampledWave(WgLds<R, TAPS> &sm, WaveLds<R, TAPS> &w, const DeviceLayout &L,
    uint32_t v, uint32_t lane, int kind, uint32_t rsM, uint32_t rsL, float rsSf, uint32_t filterOffset, bool playing,
    int32_t intPos, uint32_t fracPos, uint32_t increment, uint32_t samplesToLoad, uint32_t samplesToMix,
    int32_t bufferItem, bool looping)
{
    float *rdata = w.rd;
    float *srcBuffer = rdata + kMaxEdge;
    float *mixing = w.in + kHist;
    if(lane < kMaxPad) rdata[lane] = L.prev[size_t{v} * kMaxPad + lane];
    const float *filter = L.tables + filterOffset;
    const uint32_t tableKey = filterOffset * 8u + uint32_t(kind);
    const bool staged = (kind == 2 || kind == 3) && sm.tabKey == tableKey;
    const uint32_t sM = kind == 2 ? 4u : rsM, sL = kind == 2 ? 1u : rsL;
    WaveSync();

    for(uint32_t loaded = 0; loaded < samplesToLoad;)
    {
        uint32_t bdst, bsrc;
        CalcBufferSize(fracPos, increment, samplesToLoad - loaded, bdst, bsrc);
        uint32_t srcDelay = 0;
        bool silent = false;
        if(intPos < 0)
        {
            srcDelay = uint32_t(-intPos);
            if(srcDelay >= bsrc)
            {   // voice.cpp:679-697: everything needed is before the buffer start
                for(uint32_t k = lane; k < bdst; k += 64) mixing[loaded + k] = 0.0f;
                for(uint32_t k = lane; k < bsrc; k += 64) srcBuffer[k] = 0.0f;
                silent = true;
            }
            else
                for(uint32_t k = lane; k < srcDelay; k += 64) srcBuffer[k] = 0.0f;
        }
        if(silent)
        {
            WaveSync();
            loaded += bdst;
            if(loaded < samplesToLoad)
            {
                fracPos += bdst * increment;
                const uint32_t srcOffset = fracPos >> kFracBits;
                fracPos &= kFracMask;
                intPos = AddSat(intPos, int32_t(srcOffset));
            }
            continue;
        }

        if(bufferItem < 0)
        {   // voice.cpp:704-719: hold the available sample nearest zero
            const uint32_t avail = bsrc < uint32_t(kMaxEdge) ? bsrc : uint32_t(kMaxEdge);
            const uint32_t tofill = bsrc > uint32_t(kMaxEdge) ? bsrc : uint32_t(kMaxEdge);
            if(lane == 0)
            {
                uint32_t best = 0;
                for(uint32_t i = 1; i < avail; ++i)
                    if(fabsf(srcBuffer[i]) < fabsf(srcBuffer[best])) best = i;
                w.best = int32_t(best);
            }
            WaveSync();
            const uint32_t best = uint32_t(w.best);
            const float hold = srcBuffer[best];
            WaveSync();
            for(uint32_t k = best + 1 + lane; k < tofill; k += 64) srcBuffer[k] = hold;
        }
        else
        {
            const uint32_t upos = intPos < 0 ? 0u : uint32_t(intPos);
            FillFromBuffer<64>(srcBuffer + srcDelay, bsrc - srcDelay, L.buffers[bufferItem], looping, upos, lane);
        }
        WaveSync();

        // voice.cpp:764-769
        if(increment == kFracOne && fracPos == 0)
        {
            for(uint32_t k = lane; k < bdst; k += 64) mixing[loaded + k] = srcBuffer[k];
        }
        else if(staged)
        {
            for(uint32_t k = lane; k < bdst; k += 64)
                mixing[loaded + k] = ResampleStagedM(sm, rdata, sM, sL, fracPos, increment, k);
        }
        else
        {
            const TabLayout lay = ReferenceTabLayout(rsM);
            for(uint32_t k = lane; k < bdst; k += 64)
                mixing[loaded + k] = ResampleAt<false, false>(kind, rsM, rsL, rsSf, filter, lay, rdata, fracPos, increment, k, bdst);
        }

        // voice.cpp:772-785: history for the next update, taken at the end-of-mix position
        if(playing)
        {
            const uint32_t loadEnd = loaded + bdst;
            if(samplesToMix > loaded && samplesToMix <= loadEnd)
            {
                const uint32_t dstOffset = samplesToMix - loaded;
                const uint32_t srcOffset = uint32_t((uint64_t{dstOffset} * increment + fracPos) >> kFracBits);
                if(lane < kMaxPad) L.prev[size_t{v} * kMaxPad + lane] = rdata[srcOffset + lane];
            }
        }
        loaded += bdst;
        if(loaded < samplesToLoad)
        {
            fracPos += bdst * increment;
            const uint32_t srcOffset = fracPos >> kFracBits;
            fracPos &= kFracMask;
            if(intPos < 0) intPos += int32_t(srcOffset);
            else intPos = AddSat(intPos, int32_t(srcOffset));
            // voice.cpp:807-810: slide the last 48 source samples to the front
            WaveSync();
            float carry = 0.0f;
            if(lane < kMaxPad) carry = rdata[srcOffset + lane];
            WaveSync();
            if(lane < kMaxPad) rdata[lane] = carry;
        }
        WaveSync();
    }
}

// ---- wave-parallel dual biquad (time-invariant coefficients) ---------------------------------
// The cascade is linear in its state s = (z01, z02, z11, z12): s' = A s + B x.  Lane l owns
// samples [l*seg, (l+1)*seg).  (1) M = A^seg (the same for every lane) from seg zero-input steps
// of the four unit states; (2) forced response q_l of the lane's block from a zero state;
// (3) block-start states by a 6-step Kogge-Stone scan of S_l = M^(l) S_0 + sum M^(l-1-k) q_k
// using M, M^2, ... M^32; (4) each lane re-runs the true recurrence from its start state.
// seg is odd so that lanes reading sample l*seg + r hit distinct LDS banks.
struct Dual4 { float a, b, c, d; };

__device__ __forceinline__ float DualStepF(Dual4 &s, float x, const BiquadState &f0, const BiquadState &f1)
{
    const float y0 = __builtin_fmaf(x, f0.b0, s.a);
    s.a = __builtin_fmaf(x, f0.b1, __builtin_fmaf(-y0, f0.a1, s.b));
    s.b = __builtin_fmaf(x, f0.b2, -y0 * f0.a2);
    const float y1 = __builtin_fmaf(y0, f1.b0, s.c);
    s.c = __builtin_fmaf(y0, f1.b1, __builtin_fmaf(-y1, f1.a1, s.d));
    s.d = __builtin_fmaf(y0, f1.b2, -y1 * f1.a2);
    return y1;
}

struct Mat4 { Dual4 c0, c1, c2, c3; };    // columns

__device__ __forceinline__ Dual4 MatVec(const Mat4 &m, const Dual4 &v)
{
    Dual4 r;
    r.a = __builtin_fmaf(m.c3.a, v.d, __builtin_fmaf(m.c2.a, v.c, __builtin_fmaf(m.c1.a, v.b, m.c0.a * v.a)));
    r.b = __builtin_fmaf(m.c3.b, v.d, __builtin_fmaf(m.c2.b, v.c, __builtin_fmaf(m.c1.b, v.b, m.c0.b * v.a)));
    r.c = __builtin_fmaf(m.c3.c, v.d, __builtin_fmaf(m.c2.c, v.c, __builtin_fmaf(m.c1.c, v.b, m.c0.c * v.a)));
    r.d = __builtin_fmaf(m.c3.d, v.d, __builtin_fmaf(m.c2.d, v.c, __builtin_fmaf(m.c1.d, v.b, m.c0.d * v.a)));
    return r;
}
__device__ __forceinline__ Mat4 MatMul(const Mat4 &a, const Mat4 &b)
{ return Mat4{MatVec(a, b.c0), MatVec(a, b.c1), MatVec(a, b.c2), MatVec(a, b.c3)}; }

__device__ __forceinline__ void BiquadDualWaveScan(BiquadState &f0, BiquadState &f1, float *buf /* in place */,
    uint32_t n, uint32_t lane)
{
    const uint32_t seg = ((n + 63u) / 64u) | 1u;
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t end = (begin + seg < n) ? begin + seg : n;
    Mat4 M{{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for(uint32_t i = 0; i < seg; ++i)
    {
        DualStepF(M.c0, 0.0f, f0, f1); DualStepF(M.c1, 0.0f, f0, f1);
        DualStepF(M.c2, 0.0f, f0, f1); DualStepF(M.c3, 0.0f, f0, f1);
    }
    Dual4 q{0, 0, 0, 0};
    for(uint32_t i = begin; i < end; ++i) DualStepF(q, buf[i], f0, f1);
    // inclusive scan: after it, lane l holds E_l = sum_{k<=l} M^(l-k) q_k  (state at the END of
    // block l for a zero initial state); P_l = M^(l+1) applied to S_0 is added below.
    Dual4 e = q;
    Dual4 s0{f0.z1, f0.z2, f1.z1, f1.z2};        // becomes M^lane S_0 (binary powers of M)
    Mat4 P = M;                                  // M^(2^step)
#pragma unroll
    for(int step = 0; step < 6; ++step)
    {
        const int d = 1 << step;
        Dual4 o;
        o.a = __shfl_up(e.a, d); o.b = __shfl_up(e.b, d); o.c = __shfl_up(e.c, d); o.d = __shfl_up(e.d, d);
        const Dual4 mo = MatVec(P, o);
        if(int(lane) >= d) { e.a += mo.a; e.b += mo.b; e.c += mo.c; e.d += mo.d; }
        const Dual4 ms = MatVec(P, s0);
        if(lane & uint32_t(d)) s0 = ms;
        if(step < 5) P = MatMul(P, P);
    }
    Dual4 prevE;
    prevE.a = __shfl_up(e.a, 1); prevE.b = __shfl_up(e.b, 1); prevE.c = __shfl_up(e.c, 1); prevE.d = __shfl_up(e.d, 1);
    Dual4 start = s0;
    if(lane > 0) { start.a += prevE.a; start.b += prevE.b; start.c += prevE.c; start.d += prevE.d; }
    for(uint32_t i = begin; i < end; ++i) buf[i] = DualStepF(start, buf[i], f0, f1);
    const int lastLane = int((n - 1u) / seg);
    f0.z1 = __shfl(start.a, lastLane); f0.z2 = __shfl(start.b, lastLane);
    f1.z1 = __shfl(start.c, lastLane); f1.z2 = __shfl(start.d, lastLane);
}

// ---- dual-ear FIR, packed over the ears -------------------------------------------------------
// acc[r] = (L,R) of output frame R*lane + r.  xw = &x2[TAPS + R*lane] (x' of the lane's first
// frame).  co[j] = (Coeffs[j][0], Coeffs[j][1]) through the scalar cache.
template<int R, int TAPS>
__device__ __forceinline__ void FirMainPk(f2 (&acc)[R], const f2 *xw, cf2 *co)
{
    f2 w[R + 3];
#pragma unroll
    for(int k = 0; k < R + 3; ++k) w[k] = xw[k - 3];
#pragma unroll
    for(int b = 0; b < TAPS / 4; ++b)
    {
#pragma unroll
        for(int jj = 0; jj < 4; ++jj)
        {
            const f2 c = co[4 * b + jj];
#pragma unroll
            for(int r = 0; r < R; ++r) acc[r] = pkfma(c, w[r + 3 - jj], acc[r]);
        }
        if(b + 1 < TAPS / 4)
        {
#pragma unroll
            for(int k = R + 2; k >= 4; --k) w[k] = w[k - 4];
#pragma unroll
            for(int k = 0; k < 4; ++k) w[k] = xw[-4 * (b + 1) - 3 + k];
        }
    }
}

template<int R, int TAPS>
__global__ void __launch_bounds__(kWThreads, 2) VoiceWaveKernel(DeviceLayout L, uint32_t samplesToDo)
{
    using WL = WaveLds<R, TAPS>;
    __shared__ WgLds<R, TAPS> sm;
    const uint32_t t = threadIdx.x;
    const uint32_t lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t group = blockIdx.x;
    const uint32_t vpw = L.waveVoices;
    const uint32_t irStride = L.irStride;
    WL &w = sm.w[wave];
    const uint32_t N = samplesToDo;

    // ---- workgroup prologue: pick and stage the resampler rows most voices will use
    {
        const uint32_t gBegin = group * kWWaves * vpw;
        if(wave == 0)
        {
            const uint32_t cand = gBegin + lane;
            bool eligible = false;
            uint32_t off = 0, m = 0, l = 0;
            int kind = 0;
            if(lane < kWWaves * vpw && cand < L.numVoices)
            {
                const VoiceCtl &c = L.ctl[cand];
                kind = c.rsKind; off = c.rsFilterOffset; m = c.rsM; l = c.rsL;
                if(kind == 2) { m = 4; l = 1; }
                eligible = (kind == 2 || (kind == 3 && (m == 12 || m == 24 || m == 48)))
                    && (c.playState == OALGPU_VOICE_PLAYING || c.playState == OALGPU_VOICE_STOPPING);
            }
            const unsigned long long mask = __ballot(eligible);
            if(mask)
            {
                const int first = __ffsll((long long)mask) - 1;
                const uint32_t key = uint32_t(__shfl(int(off * 8u + uint32_t(kind)), first));
                const uint32_t fm = uint32_t(__shfl(int(m), first)), fl = uint32_t(__shfl(int(l), first));
                if(lane == 0) { sm.tabKey = key; sm.tabM = fm; sm.tabL = fl; }
            }
            else if(lane == 0) { sm.tabKey = 0xffffffffu; sm.tabM = 0; sm.tabL = 0; }
        }
        __syncthreads();
        const uint32_t key = sm.tabKey, m = sm.tabM;
        if(key != 0xffffffffu)
        {
            const float *filter = L.tables + (key >> 3);
            for(uint32_t idx = t; idx < (m / 2u) * 32u; idx += kWThreads)
            {
                const uint32_t p = idx >> 5, pi = idx & 31u;
                const float *row = filter + pi * 2u * m;
                sm.tabF[idx] = f2{row[2u * p], row[2u * p + 1u]};
                sm.tabP[idx] = f2{row[m + 2u * p], row[m + 2u * p + 1u]};
            }
        }
        // zero padding of the old-filter coefficient array (never overwritten)
        for(uint32_t k = lane; k < uint32_t(TAPS + 128); k += 64) w.cold[k] = f2{0.0f, 0.0f};
        __syncthreads();
    }

    f2 acc[R];
#pragma unroll
    for(int r = 0; r < R; ++r) acc[r] = f2{0.0f, 0.0f};
    f2 accO[WL::kQ];
#pragma unroll
    for(int q = 0; q < WL::kQ; ++q) accO[q] = f2{0.0f, 0.0f};

    const uint32_t vBegin = (group * kWWaves + wave) * vpw;
    const uint32_t vEnd = (vBegin + vpw < L.numVoices) ? vBegin + vpw : L.numVoices;
    for(uint32_t v = vBegin; v < vEnd; ++v)
    {
        // per-voice control block through the scalar cache (read once, before this wave's own
        // write-back at the end of the voice; nobody else touches voice v during the launch)
        const VoiceCtl ctl = LoadCtlScalar(L.ctl + v);
        const int vstate = ctl.playState;
        if(vstate != OALGPU_VOICE_PLAYING && vstate != OALGPU_VOICE_STOPPING) continue;
        const bool playing = vstate == OALGPU_VOICE_PLAYING;
        const uint32_t increment = ctl.step;
        if(increment < 1)
        {   // voice.cpp:1002-1010
            if(!playing && lane == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
            continue;
        }
        int32_t bufPosInt = ctl.position;
        uint32_t bufPosFrac = ctl.positionFrac;
        int32_t bufferItem = ctl.curBuffer;
        int32_t loopItem = ctl.loopBuffer;
        if(loopItem >= 0 && bufferItem >= 0)
        {   // voice.cpp:1015-1019
            if(bufPosInt >= 0 && uint32_t(bufPosInt) >= L.buffers[bufferItem].loopEnd) loopItem = -1;
        }
        const bool dirty = (ctl.flags & kFlagHrtfDirty) != 0;

        LoadResampledWave(sm, w, L, v, lane, ctl.rsKind, ctl.rsM, ctl.rsL, ctl.rsSf, ctl.rsFilterOffset, playing,
            bufPosInt, bufPosFrac, increment, N, N, bufferItem, loopItem >= 0);

        const uint32_t counter = (ctl.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093

        // ---- DoFilters, direct path (voice.cpp:255-267): in place on w.in[kHist..]
        {
            BiquadSlot *slots = &L.dfilt[size_t{v} * 2];
            BiquadState f0 = slots[0].f, f1 = slots[1].f;
            if(ctl.flags & kFlagDirectFilter)
            {
                if(f0.counter <= 0 && f1.counter <= 0)
                    BiquadDualWaveScan(f0, f1, w.in + kHist, N, lane);
                else if(lane == 0)
                    BiquadDualInterp(f0, f1, w.in + kHist, w.in + kHist, N);
                if(lane == 0) { slots[0].f = f0; slots[1].f = f1; }
            }
            else if(lane == 0)
            {
                BiquadClear(f0); BiquadClear(f1);
                slots[0].f = f0; slots[1].f = f1;
            }
            WaveSync();
        }

        // ---- DoHrtfMix, voice.cpp:827-902
        w.in[lane] = L.hist[size_t{v} * kHist + lane];
        WaveSync();
        if(playing) L.hist[size_t{v} * kHist + lane] = w.in[N + lane];

        const float targetGain = ctl.hrtfTgtGain * (playing ? 1.0f : 0.0f);
        const float oldGain = counter ? ctl.hrtfOldGain : ctl.hrtfTgtGain;   // voice.cpp:1100
        uint32_t fademix = 0;
        float blendGain = targetGain;
        if(counter)
        {
            fademix = N < counter ? N : counter;
            if(counter > fademix)
                blendGain = lerpf(oldGain, targetGain, float(fademix) / float(counter));
        }
        const float newStep = fademix ? blendGain / float(fademix) : 0.0f;
        const float gainAfterBlend = fademix ? blendGain : oldGain;
        const uint32_t todo = N - fademix;
        float endGain = targetGain;
        if(todo && counter > N)
            endGain = lerpf(gainAfterBlend, targetGain, float(todo) / float(counter - fademix));
        const float mainStep = todo ? (endGain - gainAfterBlend) / float(todo) : 0.0f;
        const bool oldOn = fademix && oldGain > kGainSilence;
        const bool newOn = fademix && newStep * float(fademix) > kGainSilence;
        const uint32_t odL = ctl.hrtfOldDelay[0], odR = ctl.hrtfOldDelay[1];
        const uint32_t dL = ctl.hrtfTgtDelay[0], dR = ctl.hrtfTgtDelay[1];
        const float oldStep = fademix ? oldGain / float(fademix) : 0.0f;
        // A voice whose target filter was not replaced since its last mix has Old == Target
        // (coefficients and delays; voice.cpp:869 / :1100), so the old-filter fade-out and the
        // new-filter fade-in of MixHrtfBlend act on the same taps and their gains are summed.
        const bool merged = !dirty;

        // x'[i] = (In[64 - dL + i], In[64 - dR + i]) * g(i); zero pads on both sides
        w.x2[lane] = f2{0.0f, 0.0f};
        if(TAPS > 64) w.x2[64 + lane] = f2{0.0f, 0.0f};
        for(uint32_t k = TAPS + N + lane; k < uint32_t(WL::kX); k += 64) w.x2[k] = f2{0.0f, 0.0f};
        for(uint32_t i = lane; i < N; i += 64)
        {
            float g;
            if(i < fademix)
            {
                g = newOn ? newStep * float(i) : 0.0f;
                if(merged && oldOn) g += oldStep * float(fademix - i);
            }
            else g = gainAfterBlend + mainStep * float(i - fademix);
            w.x2[TAPS + i] = f2{w.in[kHist - dL + i] * g, w.in[kHist - dR + i] * g};
        }
        // old-filter fade-out inputs (one per lane) and coefficients, replaced filters only
        const bool oldPass = !merged && oldOn;
        if(oldPass)
        {
            f2 xo = {0.0f, 0.0f};
            if(lane < fademix)
            {
                const float g = oldStep * float(fademix - lane);
                xo = f2{w.in[kHist - odL + lane] * g, w.in[kHist - odR + lane] * g};
            }
            w.xo[lane] = xo;
            const f2 *oc = reinterpret_cast<const f2*>(L.hrtfOld + size_t{v} * irStride * 2);
            for(uint32_t k = lane; k < uint32_t(TAPS); k += 64) w.cold[64 + k] = (k < irStride) ? oc[k] : f2{0.0f, 0.0f};
        }
        WaveSync();

        cf2 *co = (cf2*)(uintptr_t)(L.hrtfTgt + size_t{v} * irStride * 2);
        if(irStride == uint32_t(TAPS))
            FirMainPk<R, TAPS>(acc, &w.x2[TAPS + R * lane], co);
        else
        {   // other HRIR lengths: 16-tap segments (irStride is a multiple of 16, zero padded)
            const f2 *xw = &w.x2[TAPS + R * lane];
            for(uint32_t seg = 0; seg * 16u < irStride; ++seg)
                FirMainPk<R, 16>(acc, xw - 16 * seg, co + 16 * seg);
        }
        if(oldPass)
        {   // frames lane + 64q receive cOld[lane + 64q - i] * xo[i], i < 64
#pragma unroll
            for(int i = 0; i < 64; ++i)
            {
                const f2 xi = w.xo[i];               // same address in every lane: LDS broadcast
#pragma unroll
                for(int q = 0; q < WL::kQ; ++q) accO[q] = pkfma(w.cold[64 + lane + 64 * q - i], xi, accO[q]);
            }
        }

        // voice.cpp:1094-1101 / :869-873,900: Old <- Target, Old.Gain <- reached gain
        if(dirty && (counter == 0 || fademix))
        {
            const float *tg = L.hrtfTgt + size_t{v} * irStride * 2;
            float *od = L.hrtfOld + size_t{v} * irStride * 2;
            for(uint32_t k = lane; k < irStride * 2; k += 64) od[k] = tg[k];
        }
        // ---- voice.cpp:1116-1232: flags, position, loop wrap / end of buffer ----
        if(lane == 0)
        {
            VoiceCtl &c = L.ctl[v];
            if(counter == 0 || fademix) { c.hrtfOldDelay[0] = dL; c.hrtfOldDelay[1] = dR; }
            c.hrtfOldGain = todo ? endGain : gainAfterBlend;
            uint32_t flags = ctl.flags | kFlagFading;
            if(counter == 0 || fademix) flags &= ~kFlagHrtfDirty;
            c.flags = flags;
            if(!playing) c.playState = OALGPU_VOICE_STOPPED;
            else
            {
                bufPosFrac += increment * N;
                const uint32_t samplesDone = bufPosFrac >> kFracBits;
                bufPosInt = AddSat(bufPosInt, int32_t(samplesDone));
                bufPosFrac &= kFracMask;
                if(bufferItem >= 0 && bufPosInt > 0)
                {
                    const BufferItem &b = L.buffers[bufferItem];
                    if(loopItem >= 0)
                    {
                        uint32_t pos = uint32_t(bufPosInt);
                        if(pos >= b.loopEnd)
                        {
                            pos = ((pos - b.loopStart) % (b.loopEnd - b.loopStart)) + b.loopStart;
                            bufPosInt = int32_t(pos);
                        }
                    }
                    else if(uint32_t(bufPosInt) >= b.sampleLen)
                        bufferItem = -1;
                }
                c.position = bufPosInt;
                c.positionFrac = bufPosFrac;
                c.curBuffer = bufferItem;
                if(bufferItem < 0)
                {
                    c.loopBuffer = -1;
                    c.playState = OALGPU_VOICE_STOPPING;
                }
            }
        }
        WaveSync();
    }

    // ---- one partial per workgroup: waves dump their accumulators, then a fixed-order sum
    {
        f2 *dump = w.x2;                     // [frame] = (L, R), frames < 64R
        WaveSync();
#pragma unroll
        for(int r = 0; r < R; ++r) dump[R * lane + r] = acc[r];
        WaveSync();
#pragma unroll
        for(int q = 0; q < WL::kQ; ++q)
        {
            const f2 cur = dump[lane + 64 * q];
            dump[lane + 64 * q] = f2{cur.x + accO[q].x, cur.y + accO[q].y};
        }
        __syncthreads();
        f2 *ph = reinterpret_cast<f2*>(L.partHrtf) + size_t{group} * (kLine + kHrirLen);
        for(uint32_t k = t; k < uint32_t(kLine + kHrirLen); k += kWThreads)
        {
            f2 s = {0.0f, 0.0f};
            if(k < uint32_t(WL::kFrames))
            {
                s = sm.w[0].x2[k];
#pragma unroll
                for(int ww = 1; ww < kWWaves; ++ww) { const f2 o = sm.w[ww].x2[k]; s.x += o.x; s.y += o.y; }
            }
            ph[k] = s;
        }
    }
}

} // namespace

bool WaveKernelApplies(bool exact, const DeviceLayout &L)
{ return !exact && L.hrtf && L.numSends == 0 && L.irStride >= 8 && L.irStride <= 128; }

uint32_t WaveKernelGroups(const DeviceLayout &L)
{ return (L.numVoices + kWWaves * L.waveVoices - 1u) / (kWWaves * L.waveVoices); }

hipError_t LaunchVoiceWave(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo)
{
    const uint32_t groups = WaveKernelGroups(L);
    if(L.irStride <= 64)
        hipLaunchKernelGGL((VoiceWaveKernel<17, 64>), dim3(groups), dim3(kWThreads), 0, s, L, samplesToDo);
    else
        hipLaunchKernelGGL((VoiceWaveKernel<18, 128>), dim3(groups), dim3(kWThreads), 0, s, L, samplesToDo);
    return hipGetLastError();
}

} // namespace oalgpu
