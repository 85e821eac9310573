// Building blocks of the voice kernel that stages a voice in LDS (voice_wave.hip: one wavefront per voice):
// scalar-cache views of VoiceCtl, the
// LDS layouts, the register gather of the source window, the staged-row resampler, the wave-level
// LoadResampledSamples / DoFilters (core/voice.cpp:642-824, :255-267) and the filter scans.
#pragma once
#include "dev_wave.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kWWaves = 4;                    // wavefronts (= concurrent voices) per workgroup
constexpr int kWThreads = kWWaves * 64;
constexpr int kTabPairs = 24;                 // staged resampler rows: up to 48 taps (matrix-pipe kernels: 12, see WgLds)

constexpr int kPre = 17;                      // prefetched source samples per lane (17*64 = 1088)

// ---- scalar-cache views of the per-voice control block and the buffer table -------------------
// VoiceCtl (kernels.hpp) in two pieces: the head (bytes 0..47: everything needed to locate and
// resample the source) is fetched one voice AHEAD so that voice's source window can be
// requested from HBM while the current voice is still in its FIR; the tail (bytes 64..95: HRTF
// delays and gains) is fetched at the start of the voice and first needed after resampling.
struct VoiceHead {
    int32_t playState, position;
    uint32_t positionFrac;
    int32_t curBuffer, loopBuffer;
    uint32_t step;
    int32_t rsKind;
    uint32_t rsM, rsL;
    float rsSf;
    uint32_t rsFilterOffset, flags;
};
static_assert(sizeof(VoiceHead) == 48 && offsetof(VoiceCtl, flags) == 44, "VoiceHead mirrors the first 48 bytes of VoiceCtl");
static_assert(offsetof(VoiceCtl, hrtfOldDelay) == 72 && offsetof(VoiceCtl, hrtfTgtGain) == 92, "VoiceCtl tail layout");

__device__ __forceinline__ VoiceHead LoadHeadScalar(const VoiceCtl *p)
{
    union { VoiceHead h; u4 q[3]; } u;
    cu4 *src = (cu4*)(uintptr_t)p;
    u.q[0] = src[0]; u.q[1] = src[1]; u.q[2] = src[2];
    return u.h;
}

struct VoiceTail { uint32_t oldDelay[2]; float oldGain; uint32_t tgtDelay[2]; float tgtGain; };
__device__ __forceinline__ VoiceTail LoadTailScalar(const VoiceCtl *p)
{
    cu4 *src = (cu4*)(uintptr_t)p;
    const u4 a = src[4], b = src[5];          // bytes 64..95
    VoiceTail t;
    t.oldDelay[0] = a.z; t.oldDelay[1] = a.w;
    t.oldGain = __builtin_bit_cast(float, uint32_t(b.x));
    t.tgtDelay[0] = b.y; t.tgtDelay[1] = b.z;
    t.tgtGain = __builtin_bit_cast(float, uint32_t(b.w));
    return t;
}

// VoiceCtl::buf, bytes 96..127 of the voice's line
__device__ __forceinline__ BufferItem LoadCtlBufferScalar(const VoiceCtl *p)
{
    static_assert(offsetof(VoiceCtl, buf) == 96 && sizeof(BufferItem) == 32, "VoiceCtl::buf layout");
    union { BufferItem b; u4 q[2]; } u;
    cu4 *src = (cu4*)(uintptr_t)p;
    u.q[0] = src[6]; u.q[1] = src[7];
    return u.b;
}

// MF: the dual-ear FIR runs on the matrix pipe (FirMfmaH, dev_wave.hpp); its inputs are staged per ear as
// packed f16 pairs, leading halves and remainders (xh, in the place of x2), and the next voice's reversed
// response is parked in hr while this voice's registers are still busy with its write-back.
// The resampler reads its source samples in pairs; a pair at an even index of rd is one aligned ds_read_b64, a pair at
// an odd index would be a ds_read2_b32 (half the LDS rate) -- so a prefetched window is parked twice, rd and rd2 = rd
// shifted by one sample, and an odd pair is read from rd2 at the even index below it.  rd2 starts 32 banks away from rd
// (1376 = 21 x 64 + 32 dwords): the lanes of a wavefront read pairs within ~60 consecutive samples, the even ones out
// of rd, the odd ones out of rd2, on disjoint halves of the 64 banks.
constexpr int kRdFloats = 1376;
constexpr int kRd2Floats = kMaxEdge + kPre * 64 + 8;
static_assert(kRdFloats >= kResampleDataSize + 8 && kRdFloats % 64 == 32, "rd2 sits half the banks away from rd");

template<int R, int TAPS, bool MF = false>
struct alignas(16) WaveLds {
    static constexpr int kFrames = 64 * R;
    static constexpr int kX = TAPS + kFrames;           // x2[k] = x'[k - TAPS]
    static constexpr int kQ = TAPS / 64 + 1;            // old-filter fade: frames l + 64q, q < kQ
    union {
        f2 x2[kX];                                      // FIR inputs (both ears), zero padded
        uint32_t xh[2][2][MF ? kXhDw : 1];              // MF: FIR inputs [ear][hi | lo], two frames per dword
        struct {
            float rd[kRdFloats];                        // DeviceBase::mResampleData (dead before x2 is built)
            float rd2[kRd2Floats];                      // rd2[i] = rd[i + 1]: the same window one sample on, so that a source
                                                        // PAIR at an odd index is an aligned 8-byte read too (see ResampleRunStaged)
        };
    };
    float in[kHist + kLine];                            // [Hrtf.History | resampled, filtered samples]
    f2 cold[MF ? 1 : TAPS + 128];                       // cold[k] = Hrtf.Old.Coeffs[k - 64], zero padded
    f2 xo[MF ? 1 : 64];                                 // old-filter fade-out inputs (i < 64), both ears
    uint32_t hro[2][2][MF ? kHrDw : 1];                 // MF: a replaced filter's OLD response, reversed like hr
    float fst[32];                                      // the voice's two BiquadSlots (2 x 16 dwords)
    int32_t best;
    uint32_t pad[3];
    uint32_t hr[2][2][MF ? kHrDw : 1];                  // MF: the target HRIR reversed [ear][hi | lo], see FirMfmaH
};

template<int R, int TAPS, bool MF = false>
struct WgLds {
    // The matrix-pipe kernels stage rows of up to 24 taps only (bsinc48 voices take the unstaged path there):
    // with 48-tap rows their workgroup would not leave room on a CU for the post-stream reduction beside two
    // of them (LDS is allocated in granules of 1280 bytes: 2 x 62 + 4 granules = 160 KB).
    static constexpr int kPairs = MF ? 12 : kTabPairs;
    WaveLds<R, TAPS, MF> w[kWWaves];
    alignas(16) f2 tabF[kPairs * 32];                   // [tap pair][phase] = fil[2p], fil[2p+1]
    f2 tabP[kPairs * 32];                               //                    = phd[2p], phd[2p+1]
    uint32_t tabKey, tabM, tabL;
    uint32_t pad;
    uint32_t res[12];                                   // RES (voice_wave.hip): what thread 0 learned about the update -- go, samples, records, map, rows
};

// ---- source window ----------------------------------------------------------------------------
// The first chunk of LoadBufferStatic (core/voice.cpp:500-544) as a register gather: lane l
// requests elements l, l+64, ... of the `count` source samples starting at buffer position
// dataPos (loop wrap; past-the-end holds the last sample).  The loads are issued here and only
// waited for when the values are stored to LDS, one voice later.
// The loads are GLOBAL loads (the buffer's data pointer comes out of a BufferItem in memory, so
// the compiler would otherwise issue FLAT loads, which also count on lgkmcnt: every LDS wait of
// the FIR that runs while the window is in flight would then wait for the window), and they
// deliver the raw element: the int16 -> float conversion happens when the window is stored to
// LDS (GatherDecode), so that no load has to be waited for here.
#define OALGPU_WINDOW_LOAD(p) (*(p))
template<int FMT>
__device__ __forceinline__ float LoadRawGlobal(const void *data, size_t idx)
{
    if constexpr (FMT == OALGPU_FMT_FLOAT)
        return OALGPU_WINDOW_LOAD(reinterpret_cast<const __attribute__((address_space(1))) float*>(
            (const __attribute__((address_space(1))) void*)data) + idx);
    else
        return __builtin_bit_cast(float, int32_t(OALGPU_WINDOW_LOAD(reinterpret_cast<const __attribute__((address_space(1))) int16_t*>(
            (const __attribute__((address_space(1))) void*)data) + idx)));
}

__device__ __forceinline__ float GatherDecode(float raw, bool isShort)
{ return isShort ? float(__builtin_bit_cast(int32_t, raw)) * (1.0f / 32768.0f) : raw; }

template<int FMT, int NPRE = kPre, int NT = 64>
__device__ __forceinline__ void GatherStaticT(float (&pre)[NPRE], uint32_t count, const BufferItem &b, bool looping,
    uint32_t dataPos, uint32_t lane)
{
    const uint32_t fs = b.frameStep;
    if(!looping)
    {   // past the end: the last sample (index clamp, so that every element is one unconditional load)
        const bool any = b.sampleLen > dataPos;
        const uint32_t lastIdx = b.sampleLen - 1u;
#pragma unroll
        for(int i = 0; i < NPRE; ++i)
        {
            const uint32_t k = lane + uint32_t(NT) * uint32_t(i);
            const uint32_t idx = (dataPos + k < lastIdx) ? dataPos + k : lastIdx;
            pre[i] = (k < count && any) ? LoadRawGlobal<FMT>(b.data, size_t{idx} * fs) : 0.0f;
        }
    }
    else
    {   // GatherWraps() said: at most one wrap inside the window, no division per element
        const uint32_t ls = b.loopStart, le = b.loopEnd;
        const uint32_t first = le - dataPos;
#pragma unroll
        for(int i = 0; i < NPRE; ++i)
        {
            const uint32_t k = lane + uint32_t(NT) * uint32_t(i);
            const uint32_t idx = (k < first) ? dataPos + k : ls + (k - first);
            pre[i] = (k < count) ? LoadRawGlobal<FMT>(b.data, size_t{idx} * fs) : 0.0f;
        }
    }
}

// The common shape -- a mono buffer, the window and the kPre*64 elements the gather touches all
// inside the buffer and before the loop end: one uniform base, lane offset, immediate offsets.
// (Elements past `count` are loaded and never used.)  ~20 instructions instead of ~20 per element;
// the kernel is issue-bound, so this is worth ~3500 cycles per requested window.
template<int FMT, int NPRE = kPre, int NT = 64>
__device__ __forceinline__ void GatherLinearT(float (&pre)[NPRE], const BufferItem &b, uint32_t dataPos, uint32_t lane)
{
    if constexpr (FMT == OALGPU_FMT_FLOAT)
    {
        const __attribute__((address_space(1))) float *p =
            reinterpret_cast<const __attribute__((address_space(1))) float*>((const __attribute__((address_space(1))) void*)b.data) + dataPos;
        const __attribute__((address_space(1))) float *pl = p + lane;
#pragma unroll
        for(int i = 0; i < NPRE; ++i) pre[i] = OALGPU_WINDOW_LOAD(pl + NT * i);
    }
    else
    {
        const __attribute__((address_space(1))) int16_t *p =
            reinterpret_cast<const __attribute__((address_space(1))) int16_t*>((const __attribute__((address_space(1))) void*)b.data) + dataPos;
        const __attribute__((address_space(1))) int16_t *pl = p + lane;
#pragma unroll
        for(int i = 0; i < NPRE; ++i) pre[i] = __builtin_bit_cast(float, int32_t(pl[NT * i]));
    }
}

template<int NPRE = kPre, int NT = 64>
__device__ __forceinline__ bool GatherIsLinear(uint32_t count, const BufferItem &b, bool looping, uint32_t dataPos)
{
    return b.frameStep == 1u && uint64_t{dataPos} + uint32_t(NPRE * NT) <= b.sampleLen
        && (!looping || uint64_t{dataPos} + count <= b.loopEnd);
}

// The register gather covers the formats and loop shapes that matter for throughput; anything
// else is filled by the generic LoadBufferStatic loop (FillFromBuffer) when the voice starts.
__device__ __forceinline__ bool GatherCovers(uint32_t count, const BufferItem &b, bool looping, uint32_t dataPos)
{
    if(b.fmt != OALGPU_FMT_FLOAT && b.fmt != OALGPU_FMT_SHORT) return false;
    if(!looping) return true;
    return dataPos < b.loopEnd && count <= (b.loopEnd - dataPos) + (b.loopEnd - b.loopStart);
}

template<int NPRE = kPre, int NT = 64>
__device__ __forceinline__ void GatherStatic(float (&pre)[NPRE], uint32_t count, const BufferItem &b, bool looping,
    uint32_t dataPos, uint32_t lane)
{
    if(GatherIsLinear<NPRE, NT>(count, b, looping, dataPos))
    {
        if(b.fmt == OALGPU_FMT_FLOAT) GatherLinearT<OALGPU_FMT_FLOAT, NPRE, NT>(pre, b, dataPos, lane);
        else GatherLinearT<OALGPU_FMT_SHORT, NPRE, NT>(pre, b, dataPos, lane);
    }
    else if(b.fmt == OALGPU_FMT_FLOAT) GatherStaticT<OALGPU_FMT_FLOAT, NPRE, NT>(pre, count, b, looping, dataPos, lane);
    else GatherStaticT<OALGPU_FMT_SHORT, NPRE, NT>(pre, count, b, looping, dataPos, lane);
}

// What the first pass of LoadResampledSamples' loop will ask for (core/voice.cpp:600-640,
// :662-753), decided from the voice head alone.  `prefetch`: the plain case -- the voice mixes,
// has a buffer, starts at a non-negative position and its first chunk fits kPre*64 samples -- so
// the chunk can be gathered into registers ahead of time.
struct SrcPlan { bool prefetch; uint32_t bdst, bsrc; };

template<int WINDOW = kPre * 64>
__device__ __forceinline__ SrcPlan PlanSource(const VoiceHead &h, uint32_t samplesToLoad)
{
    SrcPlan p{false, 0u, 0u};
    const bool mixes = h.playState == OALGPU_VOICE_PLAYING || h.playState == OALGPU_VOICE_STOPPING;
    if(!mixes || h.step < 1u) return p;
    CalcBufferSize(h.positionFrac, h.step, samplesToLoad, p.bdst, p.bsrc);
    p.prefetch = h.curBuffer >= 0 && h.position >= 0 && p.bsrc <= uint32_t(WINDOW);
    return p;
}

// ---- resampler, staged rows -----------------------------------------------------------------
// One wavefront's share (outputs lane, lane+64, ...) of a Resample_FastBSinc / Resample_Cubic
// call (core/mixer/mixer_c.cpp:52-83; the SSE variants mixer_sse.cpp:199-329 compute the same
// terms): out[k] = sum_j (fil[j] + pf*phd[j]) * src[pos + j - l], taps two at a time in packed
// FMAs.  The taps of an output are cut into G groups of NP pairs and software-pipelined: the LDS
// reads of the next group (or of the next output's first group) are in flight while the current
// group is multiplied.  Every lane runs every pass of the loop (the store is predicated), so the
// look-ahead reads need no branch; past the last output they read unused words of this wave's
// own LDS block.  rdb = rd + MaxResamplerEdge - l.
// DUAL: the window is parked twice (WaveLds::rd2): rdb points into rd, rd2b to the same index of rd2.
template<int M, bool DUAL = false>
__device__ __forceinline__ void ResampleRunStaged(const f2 *tabF, const f2 *tabP, const float *rdb, uint32_t frac0,
    uint32_t increment, uint32_t bdst, float *out, float *sink, uint32_t lane, const float *rd2b = nullptr, uint32_t rdbIndex = 0)
{
    // Two complete outputs are in flight: while the 3*NP LDS reads of one are outstanding
    // (rows F, P and the source pairs S) the other is multiplied, which covers the LDS latency
    // at two waves per SIMD.  bsinc48 (24 pairs) takes its outputs in two halves.
    constexpr int NP = M <= 24 ? M / 2 : 12;
    constexpr int G = (M / 2) / NP;               // 1, or 2 for bsinc48
    f2 FA[NP], PA[NP], SA[NP], FB[NP], PB[NP], SB[NP];
    const uint32_t tstep = 64u * increment;
    uint32_t t = frac0 + lane * increment;
    auto load = [&](f2 (&F)[NP], f2 (&P)[NP], f2 (&S)[NP], uint32_t tt, int g)
    {
        const uint32_t pi = (tt >> 11) & 31u;
        const f2 *tf = tabF + pi, *tp = tabP + pi;
        const float *s = rdb + (tt >> kFracBits);
#pragma unroll
        for(int q = 0; q < NP; ++q)
        {
            F[q] = tf[(g * NP + q) * 32];
            P[q] = tp[(g * NP + q) * 32];
        }
        if constexpr (DUAL)
        {   // the pair (s[2j], s[2j+1]) as ONE aligned 8-byte read: out of rd when it starts at an even index, else out of rd2
            const uint32_t pos = tt >> kFracBits;
            const bool odd = ((rdbIndex + pos) & 1u) != 0u;
            const f2 *sp = reinterpret_cast<const f2*>(odd ? rd2b + pos - 1u : s);
#pragma unroll
            for(int q = 0; q < NP; ++q) S[q] = sp[g * NP + q];
        }
        else
        {
#pragma unroll
        for(int q = 0; q < NP; ++q) S[q] = f2{s[2 * (g * NP + q)], s[2 * (g * NP + q) + 1]};
        }
    };
    auto compute = [&](const f2 (&F)[NP], const f2 (&P)[NP], const f2 (&S)[NP], uint32_t tt, f2 &r0, f2 &r1)
    {
        const f2 pf = splat(float(tt & 2047u) * (1.0f / 2048.0f));
#pragma unroll
        for(int q = 0; q < NP; ++q)
        {
            const f2 c = pkfma(pf, P[q], F[q]);
            if(q & 1) r1 = pkfma(c, S[q], r1);
            else r0 = pkfma(c, S[q], r0);
        }
    };
    auto store = [&](uint32_t k, f2 r0, f2 r1)
    {   // lanes past the end write a scratch word instead of branching
        float *dst = (k < bdst) ? out + k : sink;
        *dst = (r0.x + r0.y) + (r1.x + r1.y);
    };
    if constexpr(G == 1)
    {
        load(FA, PA, SA, t, 0);
#pragma unroll 1
        for(uint32_t kb = 0; kb < bdst; kb += 128)
        {
            const uint32_t t0 = t, t1 = t + tstep;
            t = t1 + tstep;
            load(FB, PB, SB, t1, 0);
            f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
            compute(FA, PA, SA, t0, r0, r1);
            store(kb + lane, r0, r1);
            load(FA, PA, SA, t, 0);
            f2 u0 = {0.0f, 0.0f}, u1 = {0.0f, 0.0f};
            compute(FB, PB, SB, t1, u0, u1);
            store(kb + 64u + lane, u0, u1);
        }
    }
    else
    {
        load(FA, PA, SA, t, 0);
#pragma unroll 1
        for(uint32_t kb = 0; kb < bdst; kb += 64)
        {
            const uint32_t t0 = t;
            t += tstep;
            load(FB, PB, SB, t0, 1);
            f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
            compute(FA, PA, SA, t0, r0, r1);
            load(FA, PA, SA, t, 0);
            compute(FB, PB, SB, t0, r0, r1);
            store(kb + lane, r0, r1);
        }
    }
}

// ---- resampler, staged rows, NT lanes ------------------------------------------------------------
// Outputs tid, tid + NT, ... of a Resample_FastBSinc / Resample_Cubic call (mixer_c.cpp:52-83), rows
// from the [tap pair][phase] LDS copy.  Register-lean form of ResampleRunStaged: the taps of an output
// go in groups of NP <= 6 pairs through two register sets, the reads of the next group (or of the next
// output's first group) are in flight while one is multiplied.  rdb = rd + MaxResamplerEdge - l.
template<int M, int NT>
__device__ __forceinline__ void ResampleRunBlock(const f2 *tabF, const f2 *tabP, const float *rdb, uint32_t frac0,
    uint32_t increment, uint32_t bdst, float *out, uint32_t tid)
{
    constexpr int NP = (M / 2 >= 6) ? 6 : M / 2;
    constexpr int G = (M / 2) / NP;               // 1 (cubic, bsinc12), 2 (bsinc24), 4 (bsinc48)
    static_assert(G == 1 || G % 2 == 0, "groups alternate between two register sets");
    f2 FA[NP], PA[NP], SA[NP], FB[NP], PB[NP], SB[NP];
    const uint32_t tstep = uint32_t(NT) * increment;
    const uint32_t tlast = frac0 + (bdst - 1u) * increment;       // look-ahead reads stay inside the window
    uint32_t t = frac0 + tid * increment;
    auto load = [&](f2 (&F)[NP], f2 (&P)[NP], f2 (&S)[NP], uint32_t tt, int g)
    {
        const uint32_t pi = (tt >> 11) & 31u;
        const f2 *tf = tabF + pi, *tp = tabP + pi;
        const float *s = rdb + (tt >> kFracBits);
#pragma unroll
        for(int q = 0; q < NP; ++q)
        {
            F[q] = tf[(g * NP + q) * 32];
            P[q] = tp[(g * NP + q) * 32];
        }
#pragma unroll
        for(int q = 0; q < NP; ++q) S[q] = f2{s[2 * (g * NP + q)], s[2 * (g * NP + q) + 1]};
    };
    auto compute = [&](const f2 (&F)[NP], const f2 (&P)[NP], const f2 (&S)[NP], uint32_t tt, f2 &r0, f2 &r1)
    {
        const f2 pf = splat(float(tt & 2047u) * (1.0f / 2048.0f));
#pragma unroll
        for(int q = 0; q < NP; ++q)
        {
            const f2 c = pkfma(pf, P[q], F[q]);
            if(q & 1) r1 = pkfma(c, S[q], r1);
            else r0 = pkfma(c, S[q], r0);
        }
    };
    auto clampT = [&](uint32_t tt, uint32_t k) { return k < bdst ? tt : tlast; };
    if constexpr(G == 1)
    {
        load(FA, PA, SA, clampT(t, tid), 0);
#pragma unroll 1
        for(uint32_t kb = 0; kb < bdst; kb += 2u * NT)
        {
            const uint32_t k0 = kb + tid, k1 = k0 + NT;
            const uint32_t t0 = clampT(t, k0), t1 = clampT(t + tstep, k1);
            t += 2u * tstep;
            load(FB, PB, SB, t1, 0);
            f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
            compute(FA, PA, SA, t0, r0, r1);
            if(k0 < bdst) out[k0] = (r0.x + r0.y) + (r1.x + r1.y);
            load(FA, PA, SA, clampT(t, k1 + NT), 0);
            f2 u0 = {0.0f, 0.0f}, u1 = {0.0f, 0.0f};
            compute(FB, PB, SB, t1, u0, u1);
            if(k1 < bdst) out[k1] = (u0.x + u0.y) + (u1.x + u1.y);
        }
    }
    else
    {
        load(FA, PA, SA, clampT(t, tid), 0);
#pragma unroll 1
        for(uint32_t kb = 0; kb < bdst; kb += NT)
        {
            const uint32_t k0 = kb + tid;
            const uint32_t t0 = clampT(t, k0);
            t += tstep;
            const uint32_t tn = clampT(t, k0 + NT);
            f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
#pragma unroll
            for(int g = 0; g < G; g += 2)
            {
                load(FB, PB, SB, t0, g + 1);
                compute(FA, PA, SA, t0, r0, r1);
                if(g + 2 < G) load(FA, PA, SA, t0, g + 2);
                else load(FA, PA, SA, tn, 0);
                compute(FB, PB, SB, t0, r0, r1);
            }
            if(k0 < bdst) out[k0] = (r0.x + r0.y) + (r1.x + r1.y);
        }
    }
}

// ---- resampler, staged rows, a ring of register sets ----------------------------------------------------------------
// The same outputs as ResampleRunStaged (lane, lane + 64, ...) for the kernels that also hold line accumulators in
// registers: the taps of an output go in groups of NP <= 6 pairs through a RING of SETS register sets -- while one group
// is multiplied the reads of the next SETS - 1 groups are in flight (SETS = 3: 36 reads, as many as ResampleRunStaged
// keeps outstanding with its two complete outputs, in 108 registers instead of 144).  The loop is unrolled over
// lcm(G, SETS) groups so that every set and tap group has a static index.  Source pairs come as aligned 8-byte reads
// out of rd / rd2 when the window was parked twice (DUAL).  Outputs past bdst go to `sink`; their look-ahead reads are
// clamped to the last output's position.
template<int M, int SETS, bool DUAL>
__device__ __forceinline__ void ResampleRunRing(const f2 *tabF, const f2 *tabP, const float *rdb, uint32_t frac0,
    uint32_t increment, uint32_t bdst, float *out, float *sink, uint32_t lane, const float *rd2b, uint32_t rdbIndex)
{
    constexpr int NP = (M / 2 >= 6) ? 6 : M / 2;
    constexpr int G = (M / 2) / NP;               // 1 (cubic, bsinc12), 2 (bsinc24), 4 (bsinc48)
    constexpr int U = (G % SETS == 0) ? G : ((SETS % G == 0) ? SETS : G * SETS);      // lcm for G in {1, 2, 4}, SETS in {2, 3}
    static_assert(U % G == 0 && U % SETS == 0, "the unrolled body covers whole outputs and whole turns of the ring");
    f2 F[SETS][NP], P[SETS][NP], S[SETS][NP];
    const uint32_t tstep = 64u * increment;
    const uint32_t tlast = frac0 + (bdst - 1u) * increment;
    const uint32_t nOut = (bdst + 63u) / 64u;
    auto load = [&](int set, uint32_t tt, int g)
    {
        tt = tt < tlast ? tt : tlast;
        const uint32_t pi = (tt >> 11) & 31u;
        const f2 *tf = tabF + pi, *tp = tabP + pi;
        const uint32_t pos = tt >> kFracBits;
#pragma unroll
        for(int q = 0; q < NP; ++q)
        {
            F[set][q] = tf[(g * NP + q) * 32];
            P[set][q] = tp[(g * NP + q) * 32];
        }
        if constexpr (DUAL)
        {
            const bool odd = ((rdbIndex + pos) & 1u) != 0u;
            const f2 *sp = reinterpret_cast<const f2*>(odd ? rd2b + pos - 1u : rdb + pos);
#pragma unroll
            for(int q = 0; q < NP; ++q) S[set][q] = sp[g * NP + q];
        }
        else
        {
            const float *s = rdb + pos;
#pragma unroll
            for(int q = 0; q < NP; ++q) S[set][q] = f2{s[2 * (g * NP + q)], s[2 * (g * NP + q) + 1]};
        }
    };
    uint32_t tb = frac0 + lane * increment;       // time of the output the unrolled body starts with
    // prologue: groups 0 .. SETS-2
#pragma unroll
    for(int j = 0; j < SETS - 1; ++j) load(j % SETS, tb + uint32_t(j / G) * tstep, j % G);
#pragma unroll 1
    for(uint32_t ob = 0; ob < nOut; ob += uint32_t(U / G))
    {
        f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
#pragma unroll
        for(int u = 0; u < U; ++u)
        {
            constexpr int dummy = 0; (void)dummy;
            const int ahead = u + SETS - 1;
            load(ahead % SETS, tb + uint32_t(ahead / G) * tstep, ahead % G);
            const uint32_t tt = tb + uint32_t(u / G) * tstep;
            const f2 pf = splat(float(tt & 2047u) * (1.0f / 2048.0f));
            if(u % G == 0) { r0 = f2{0.0f, 0.0f}; r1 = f2{0.0f, 0.0f}; }
#pragma unroll
            for(int q = 0; q < NP; ++q)
            {
                const f2 c = pkfma(pf, P[u % SETS][q], F[u % SETS][q]);
                if(q & 1) r1 = pkfma(c, S[u % SETS][q], r1);
                else r0 = pkfma(c, S[u % SETS][q], r0);
            }
            if(u % G == G - 1)
            {
                const uint32_t k = (ob + uint32_t(u / G)) * 64u + lane;
                float *dst = (k < bdst) ? out + k : sink;
                *dst = (r0.x + r0.y) + (r1.x + r1.y);
            }
        }
        tb += uint32_t(U / G) * tstep;
    }
}

#ifndef OALGPU_RING_SETS
#define OALGPU_RING_SETS 2
#endif
template<class SM>
__device__ __forceinline__ void ResampleRunRingM(const SM &sm, const float *rdb, uint32_t m, uint32_t frac0, uint32_t increment,
    uint32_t bdst, float *out, float *sink, uint32_t lane, const float *rd2b, uint32_t rdbIndex)
{
    constexpr int SETS = OALGPU_RING_SETS;
    if(rd2b)
    {
        if(m == 24) { ResampleRunRing<24, SETS, true>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, rd2b, rdbIndex); return; }
        if(m == 12) { ResampleRunRing<12, SETS, true>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, rd2b, rdbIndex); return; }
    }
    switch(m)
    {
    case 4: ResampleRunRing<4, SETS, false>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, nullptr, 0u); break;
    case 12: ResampleRunRing<12, SETS, false>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, nullptr, 0u); break;
    case 24: ResampleRunRing<24, SETS, false>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, nullptr, 0u); break;
    default: ResampleRunRing<48, SETS, false>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, nullptr, 0u); break;
    }
}

template<int NT, class SM>
__device__ __forceinline__ void ResampleRunBlockM(const SM &sm, const float *rdb, uint32_t m, uint32_t frac0,
    uint32_t increment, uint32_t bdst, float *out, uint32_t tid)
{
    switch(m)
    {
    case 4: ResampleRunBlock<4, NT>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, tid); break;
    case 12: ResampleRunBlock<12, NT>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, tid); break;
    case 24: ResampleRunBlock<24, NT>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, tid); break;
    default: ResampleRunBlock<48, NT>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, tid); break;
    }
}

template<class SM>
__device__ __forceinline__ void ResampleRunStagedM(const SM &sm, const float *rdb, uint32_t m,
    uint32_t frac0, uint32_t increment, uint32_t bdst, float *out, float *sink, uint32_t lane, const float *rd2b = nullptr,
    uint32_t rdbIndex = 0)
{
    if(rd2b)
    {   // the window was parked twice: source pairs as aligned 8-byte reads (the 24- and 12-tap kernels)
        if(m == 24) { ResampleRunStaged<24, true>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, rd2b, rdbIndex); return; }
        if(m == 12) { ResampleRunStaged<12, true>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane, rd2b, rdbIndex); return; }
    }
    switch(m)
    {
    case 4: ResampleRunStaged<4>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane); break;
    case 12: ResampleRunStaged<12>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane); break;
    case 24: ResampleRunStaged<24>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane); break;
    default: ResampleRunStaged<48>(sm.tabF, sm.tabP, rdb, frac0, increment, bdst, out, sink, lane); break;
    }
}

// LoadResampledSamples, core/voice.cpp:642-824, for one real channel of a static voice, by one
// wavefront.  Produces samplesToLoad resampled samples at w.in[kHist..]; updates prev[v] when
// Playing.  Integer logic identical to voice_kernel.hip's LoadResampled.  When plan.prefetch is
// set the first chunk's source samples are already on their way in `pre` (GatherStatic) and
// `prevv` holds mPrevSamples[lane].
// LEAN: the register-lean staged resampler (kernels that run at four wavefronts per SIMD).
template<bool LEAN = false, bool PROF = false, class SM, class WV, class LT>
__device__ __forceinline__ void LoadResampledWave(SM &sm, WV &w, const LT &L,
    uint32_t v, uint32_t lane, const VoiceHead &h, bool playing, uint32_t samplesToLoad, uint32_t samplesToMix,
    int32_t bufferItem, bool looping, const SrcPlan &plan, uint32_t mixOffset = 0, const WaveProf &prof = WaveProf{nullptr, 0u})
{
    float *rdata = w.rd;
    float *srcBuffer = rdata + kMaxEdge;
    float *mixing = w.in + kHist + mixOffset;     // mixOffset: a delayed start's output position (voice.cpp:1023-1046)
    const int kind = h.rsKind;
    const uint32_t rsM = h.rsM, rsL = h.rsL, increment = h.step;
    int32_t intPos = h.position;
    uint32_t fracPos = h.positionFrac;
    // plan.prefetch: mPrevSamples and the first chunk's window (plan.bsrc samples) were parked in
    // rdata / srcBuffer by the previous pass (ParkNextVoice)
    if(!plan.prefetch && lane < kMaxPad) rdata[lane] = L.prev[size_t{v} * kMaxPad + lane];
    const float *filter = L.tables + h.rsFilterOffset;
    const uint32_t tableKey = h.rsFilterOffset * 8u + uint32_t(kind);
    const bool staged = (kind == 2 || kind == 3) && sm.tabKey == tableKey;
    const uint32_t sM = kind == 2 ? 4u : rsM, sL = kind == 2 ? 1u : rsL;
    WaveSync();

    bool firstPass = true;
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
            firstPass = false;
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
        else if(firstPass && plan.prefetch) {}      // already in srcBuffer (bsrc == plan.bsrc)
        else
        {
            const uint32_t upos = intPos < 0 ? 0u : uint32_t(intPos);
            if(h.flags & kFlagQueue)
                FillFromQueue<64>(srcBuffer + srcDelay, bsrc - srcDelay, L.buffers, bufferItem, h.loopBuffer, upos, lane,
                    [] { WaveSync(); });
            else
                FillFromBuffer<64>(srcBuffer + srcDelay, bsrc - srcDelay, L.buffers[bufferItem], looping, upos, lane);
        }
        firstPass = false;
        WaveSync();
        if constexpr (PROF) { if(prof.times && lane == 0 && loaded == 0) prof.times[size_t{v} * 8 + 7] = __builtin_readcyclecounter(); }

        // voice.cpp:764-769
        if((increment == kFracOne && fracPos == 0) || (PROF && (prof.ablate & 2u)))
        {
            for(uint32_t k = lane; k < bdst; k += 64) mixing[loaded + k] = srcBuffer[k];
        }
        else if(staged)
        {
            // (the first chunk of a prefetched window is in rd AND rd2)
            const bool dual = plan.prefetch && loaded == 0;
            if constexpr (LEAN)
                ResampleRunRingM(sm, rdata + (kMaxEdge - sL), sM, fracPos, increment, bdst, mixing + loaded,
                    reinterpret_cast<float*>(&w.pad[0]), lane, dual ? w.rd2 + (kMaxEdge - sL) : nullptr, uint32_t(kMaxEdge) - sL);
            else
            {
                ResampleRunStagedM(sm, rdata + (kMaxEdge - sL), sM, fracPos, increment, bdst, mixing + loaded,
                    reinterpret_cast<float*>(&w.pad[0]), lane, dual ? w.rd2 + (kMaxEdge - sL) : nullptr, uint32_t(kMaxEdge) - sL);
            }
        }
        else
        {
            const TabLayout lay = ReferenceTabLayout(rsM);
            for(uint32_t k = lane; k < bdst; k += 64)
                mixing[loaded + k] = ResampleAt<false, false>(kind, rsM, rsL, h.rsSf, filter, lay, rdata, fracPos, increment, k, bdst);
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
// BiquadFilter::dualProcess (core/filters/biquad.cpp:254-282) is two transposed-direct-form-II
// sections in cascade; each is linear in its state s = (z1, z2): s' = A s + Bv x with
// A = [[-a1, 1], [-a2, 0]].  Lane l owns the run of samples [17 l, 17 l + 17) -- an odd length, so
// the per-lane runs hit distinct LDS banks -- and keeps it in registers for both sections:
//   (1) M = A^17;  (2) forced response q_l of the run from a zero state;  (3) run-start states
//   S_l = M^l S_0 + sum_{k<l} M^(l-1-k) q_k by a scan over the lanes (ScanLinear2);
//   (4) the true recurrence from S_l.  Lanes past the last sample only produce values nobody
// reads.  The serial loop of the reference differs from this by rounding only.
// one section over the lane's run x[0..17) (zeros past the end of the data); z1/z2: state in, state after the last
// sample out.  cnt = this lane's samples of the run, lastLane = the lane that holds sample n - 1.
//   * the forced response is NOT the recurrence run once more: the run's end state from a zero state is
//     sum_i x[i] A^(16-i) Bv, Bv = (b1 - a1 b0, b2 - a2 b0), and c_j = A^j Bv comes out of the loop that raises A to
//     M = A^17 anyway -- 17 independent FMA pairs behind a chain that is one operation per step, where the recurrence
//     is a chain of two per step and sample;
//   * the recurrence itself keeps x b1 + z2 off the chain: out -> z1 is ONE dependent FMA.
__device__ __forceinline__ void BiquadWaveScan(float (&x)[kBqSeg], uint32_t cnt, const BiquadState &f,
    float &z1, float &z2, uint32_t lane, int lastLane)
{
    const float b0 = f.b0, b1 = f.b1, b2 = f.b2, a1 = f.a1, a2 = f.a2;
    S2 m0{1.0f, 0.0f}, m1{0.0f, 1.0f};                // columns of A^j
    S2 c{__builtin_fmaf(-a1, b0, b1), __builtin_fmaf(-a2, b0, b2)};      // A^j Bv
    S2 ea{0.0f, 0.0f}, eb{0.0f, 0.0f};                // two partial sums: the additions are a chain too
#pragma unroll
    for(int j = 0; j < kBqSeg; ++j)
    {
        const float xv = x[kBqSeg - 1 - j];
        if(j & 1) { eb.a = __builtin_fmaf(xv, c.a, eb.a); eb.b = __builtin_fmaf(xv, c.b, eb.b); }
        else { ea.a = __builtin_fmaf(xv, c.a, ea.a); ea.b = __builtin_fmaf(xv, c.b, ea.b); }
        c = S2{__builtin_fmaf(-a1, c.a, c.b), -a2 * c.a};
        m0 = S2{__builtin_fmaf(-a1, m0.a, m0.b), -a2 * m0.a};
        m1 = S2{__builtin_fmaf(-a1, m1.a, m1.b), -a2 * m1.a};
    }
    S2 e{ea.a + eb.a, ea.b + eb.b};
    {   // lane 0 starts from the filter's state: the scan carries it along
        const S2 mz = Mv2(m0, m1, S2{z1, z2});
        if(lane == 0u) { e.a += mz.a; e.b += mz.b; }
    }
    e = ScanLinear2(e, m0, m1, lane);
    // the run's start state: the end state of the lane before (wave_shr:1; lane 0 keeps the filter's state)
    S2 st;
    st.a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z1), __builtin_bit_cast(int, e.a), 0x138, 0xF, 0xF, false));
    st.b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z2), __builtin_bit_cast(int, e.b), 0x138, 0xF, 0xF, false));
    S2 zc = st;                                       // the state after this lane's last sample
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i)
    {
        const float xv = x[i];
        const float t1 = __builtin_fmaf(xv, b1, st.b);
        const float y = __builtin_fmaf(xv, b0, st.a);
        st.a = __builtin_fmaf(-y, a1, t1);
        st.b = __builtin_fmaf(xv, b2, -y * a2);
        x[i] = y;
        if(uint32_t(i) + 1u == cnt) zc = st;
    }
    z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc.a), lastLane));
    z2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc.b), lastLane));
}

__device__ __forceinline__ void BiquadDualWaveScan(BiquadState &f0, BiquadState &f1, float *buf /* in place */,
    uint32_t n, uint32_t lane)
{
    // runs of 17 whatever n is: nothing in the scan then depends on a run length, and a short block is short anyway
    const uint32_t begin = lane * uint32_t(kBqSeg) < n ? lane * uint32_t(kBqSeg) : n;
    const uint32_t cnt = (begin + uint32_t(kBqSeg) < n) ? uint32_t(kBqSeg) : n - begin;
    const int lastLane = int((n - 1u) / uint32_t(kBqSeg));
    float x[kBqSeg];
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) x[i] = (uint32_t(i) < cnt) ? buf[begin + i] : 0.0f;
    BiquadWaveScan(x, cnt, f0, f0.z1, f0.z2, lane, lastLane);
    BiquadWaveScan(x, cnt, f1, f1.z1, f1.z2, lane, lastLane);
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) if(uint32_t(i) < cnt) buf[begin + i] = x[i];
}

// DoFilters (voice.cpp:255-267) for one filter pair of a voice, in place over buf[0..n):
// settled coefficients run the wave-parallel block scan, an interpolating filter the serial
// reference loop on lane 0; an inactive pair is cleared (voice.cpp:264-265).  `fst` = this
// wave's 32-dword LDS scratch holding the two BiquadSlots' first 16 dwords each.
__device__ __forceinline__ void WaveDoFilters(float *fst, BiquadSlot *slots, bool filterActive, float *buf, uint32_t n,
    uint32_t lane)
{
    BiquadState f0, f1;
    {
        const float *a = fst, *b = fst + 16;
        f0 = BiquadState{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], __builtin_bit_cast(int32_t, a[12])};
        f1 = BiquadState{b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], __builtin_bit_cast(int32_t, b[12])};
    }
    if(filterActive)
    {
        if(f0.counter <= 0 && f1.counter <= 0)
        {
            BiquadDualWaveScan(f0, f1, buf, n, lane);
            if(lane == 0) { slots[0].f.z1 = f0.z1; slots[0].f.z2 = f0.z2; slots[1].f.z1 = f1.z1; slots[1].f.z2 = f1.z2; }
        }
        else
        {
            if(lane == 0) BiquadDualInterp(f0, f1, buf, buf, n);
            if(lane == 0) { slots[0].f = f0; slots[1].f = f1; }
        }
    }
    else
    {   // skip the store when the pair already is clear
        const bool clean0 = f0.z1 == 0.0f && f0.z2 == 0.0f && f0.counter == 0 && f0.b0 == f0.tb0 && f0.b1 == f0.tb1
            && f0.b2 == f0.tb2 && f0.a1 == f0.ta1 && f0.a2 == f0.ta2;
        const bool clean1 = f1.z1 == 0.0f && f1.z2 == 0.0f && f1.counter == 0 && f1.b0 == f1.tb0 && f1.b1 == f1.tb1
            && f1.b2 == f1.tb2 && f1.a1 == f1.ta1 && f1.a2 == f1.ta2;
        if(!(clean0 && clean1) && lane == 0)
        {
            BiquadClear(f0); BiquadClear(f1);
            slots[0].f = f0; slots[1].f = f1;
        }
    }
}

// NfcFilterN::process, core/filters/nfc.cpp:222-288: the filter of order o is a cascade of sections, each a
// short linear recurrence in its integrator states --
//   first order   y = x g - a1 z0;            out = y + b1 z0;            z0 += y
//   second order  y = x g - a1 z0 - a2 z1;    out = y + b1 z0 + b2 z1;    z1 += z0; z0 += y
// (order 1: one first-order section; 2: one second-order; 3: second then first; 4: two second-order) -- so a
// wavefront runs each section as the same block scan as the biquads: lane l owns the run [17 l, 17 l + 17),
// M = A^17, the run's forced response, run-start states by a 6-step Kogge-Stone scan with M, M^2 .. M^32, then
// the recurrence itself from the true start state.  The reference's serial loop differs by rounding only.
__device__ __forceinline__ float Nfc1Step(float &z, float x, float g, float a1, float b1)
{
    const float y = __builtin_fmaf(x, g, -a1 * z);
    const float out = __builtin_fmaf(b1, z, y);
    z += y;
    return out;
}
__device__ __forceinline__ float Nfc2Step(S2 &s, float x, float g, float a1, float a2, float b1, float b2)
{
    const float y = __builtin_fmaf(x, g, __builtin_fmaf(-a1, s.a, -a2 * s.b));
    const float out = __builtin_fmaf(b1, s.a, __builtin_fmaf(b2, s.b, y));
    s.b += s.a;
    s.a += y;
    return out;
}
__device__ __forceinline__ void Nfc1WaveScan(float (&x)[kBqSeg], uint32_t cnt, uint32_t seg, float g, float a1, float b1, float &z,
    uint32_t lane, int lastLane)
{
    float m = 1.0f;                                   // (1 - a1)^seg
    for(uint32_t i = 0; i < seg; ++i) m = __builtin_fmaf(-a1, m, m);
    float e = 0.0f;
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) if(uint32_t(i) < seg) Nfc1Step(e, x[i], g, a1, b1);
    float s0 = z, p = m;
#pragma unroll
    for(int step = 0; step < 6; ++step)
    {
        const int d = 1 << step;
        const float o = __shfl_up(e, d);
        if(int(lane) >= d) e = __builtin_fmaf(p, o, e);
        if(lane & uint32_t(d)) s0 = p * s0;
        p = p * p;
    }
    const float prevE = __shfl_up(e, 1);
    float st = s0;
    if(lane > 0) st += prevE;
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i)
        if(uint32_t(i) < cnt) x[i] = Nfc1Step(st, x[i], g, a1, b1);
    z = __shfl(st, lastLane);
}
__device__ __forceinline__ void Nfc2WaveScan(float (&x)[kBqSeg], uint32_t cnt, uint32_t seg, float g, float a1, float a2, float b1,
    float b2, float &z0, float &z1, uint32_t lane, int lastLane)
{
    S2 m0{1.0f, 0.0f}, m1{0.0f, 1.0f};                 // columns of M = A^seg, A = [[1 - a1, -a2], [1, 1]]
    for(uint32_t i = 0; i < seg; ++i)
    {
        Nfc2Step(m0, 0.0f, g, a1, a2, b1, b2);
        Nfc2Step(m1, 0.0f, g, a1, a2, b1, b2);
    }
    S2 e{0.0f, 0.0f};
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) if(uint32_t(i) < seg) Nfc2Step(e, x[i], g, a1, a2, b1, b2);
    S2 s0{z0, z1};
    S2 p0 = m0, p1 = m1;
#pragma unroll
    for(int step = 0; step < 6; ++step)
    {
        const int d = 1 << step;
        const S2 o{__shfl_up(e.a, d), __shfl_up(e.b, d)};
        const S2 mo = Mv2(p0, p1, o);
        if(int(lane) >= d) { e.a += mo.a; e.b += mo.b; }
        const S2 ms = Mv2(p0, p1, s0);
        if(lane & uint32_t(d)) s0 = ms;
        if(step < 5) { const S2 n0 = Mv2(p0, p1, p0), n1 = Mv2(p0, p1, p1); p0 = n0; p1 = n1; }
    }
    const S2 prevE{__shfl_up(e.a, 1), __shfl_up(e.b, 1)};
    S2 st = s0;
    if(lane > 0) { st.a += prevE.a; st.b += prevE.b; }
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i)
        if(uint32_t(i) < cnt) x[i] = Nfc2Step(st, x[i], g, a1, a2, b1, b2);
    z0 = __shfl(st.a, lastLane);
    z1 = __shfl(st.b, lastLane);
}

// order o over src[0..n) -> dst (both LDS), by one wavefront; the filter's states move on
__device__ __forceinline__ void NfcWaveScan(NfcState &st, uint32_t o, const float *src, float *dst, uint32_t n, uint32_t lane)
{
    const float a0 = st.a[o][0], a1 = st.a[o][1], a2 = st.a[o][2], a3 = st.a[o][3], a4 = st.a[o][4];
    const float b1 = st.b[o][1], b2 = st.b[o][2], b3 = st.b[o][3], b4 = st.b[o][4];
    float z0 = st.z[o][0], z1 = st.z[o][1], z2 = st.z[o][2], z3 = st.z[o][3];
    const uint32_t seg = ((n + 63u) / 64u) | 1u;      // <= kBqSeg for n <= 1024
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t cnt = (begin + seg < n) ? seg : n - begin;
    const int lastLane = int((n - 1u) / seg);
    float x[kBqSeg];
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) x[i] = (uint32_t(i) < cnt) ? src[begin + i] : 0.0f;
    if(o == 1) Nfc1WaveScan(x, cnt, seg, a0, a1, b1, z0, lane, lastLane);
    else
    {
        Nfc2WaveScan(x, cnt, seg, a0, a1, a2, b1, b2, z0, z1, lane, lastLane);
        if(o == 3) Nfc1WaveScan(x, cnt, seg, 1.0f, a3, b3, z2, lane, lastLane);
        else if(o >= 4) Nfc2WaveScan(x, cnt, seg, 1.0f, a3, a4, b3, b4, z2, z3, lane, lastLane);
    }
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) if(uint32_t(i) < cnt) dst[begin + i] = x[i];
    if(lane == 0) { st.z[o][0] = z0; st.z[o][1] = z1; st.z[o][2] = z2; st.z[o][3] = z3; }
}


} // namespace
} // namespace oalgpu
