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
#include "dev_wave.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kWWaves = 4;                    // wavefronts (= concurrent voices) per workgroup
constexpr int kWThreads = kWWaves * 64;
constexpr int kTabPairs = 24;                 // staged resampler rows: up to 48 taps

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
    float fst[32];                                      // the voice's two BiquadSlots (2 x 16 dwords)
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
template<int FMT>
__device__ __forceinline__ float LoadRawGlobal(const void *data, size_t idx)
{
    if constexpr (FMT == OALGPU_FMT_FLOAT)
        return reinterpret_cast<const __attribute__((address_space(1))) float*>(
            (const __attribute__((address_space(1))) void*)data)[idx];
    else
        return __builtin_bit_cast(float, int32_t(reinterpret_cast<const __attribute__((address_space(1))) int16_t*>(
            (const __attribute__((address_space(1))) void*)data)[idx]));
}

__device__ __forceinline__ float GatherDecode(float raw, bool isShort)
{ return isShort ? float(__builtin_bit_cast(int32_t, raw)) * (1.0f / 32768.0f) : raw; }

template<int FMT>
__device__ __forceinline__ void GatherStaticT(float (&pre)[kPre], uint32_t count, const BufferItem &b, bool looping,
    uint32_t dataPos, uint32_t lane)
{
    const uint32_t fs = b.frameStep;
    if(!looping)
    {   // past the end: the last sample (index clamp, so that every element is one unconditional load)
        const bool any = b.sampleLen > dataPos;
        const uint32_t lastIdx = b.sampleLen - 1u;
#pragma unroll
        for(int i = 0; i < kPre; ++i)
        {
            const uint32_t k = lane + 64u * uint32_t(i);
            const uint32_t idx = (dataPos + k < lastIdx) ? dataPos + k : lastIdx;
            pre[i] = (k < count && any) ? LoadRawGlobal<FMT>(b.data, size_t{idx} * fs) : 0.0f;
        }
    }
    else
    {   // GatherWraps() said: at most one wrap inside the window, no division per element
        const uint32_t ls = b.loopStart, le = b.loopEnd;
        const uint32_t first = le - dataPos;
#pragma unroll
        for(int i = 0; i < kPre; ++i)
        {
            const uint32_t k = lane + 64u * uint32_t(i);
            const uint32_t idx = (k < first) ? dataPos + k : ls + (k - first);
            pre[i] = (k < count) ? LoadRawGlobal<FMT>(b.data, size_t{idx} * fs) : 0.0f;
        }
    }
}

// The common shape -- a mono buffer, the window and the kPre*64 elements the gather touches all
// inside the buffer and before the loop end: one uniform base, lane offset, immediate offsets.
// (Elements past `count` are loaded and never used.)  ~20 instructions instead of ~20 per element;
// the kernel is issue-bound, so this is worth ~3500 cycles per requested window.
template<int FMT>
__device__ __forceinline__ void GatherLinearT(float (&pre)[kPre], const BufferItem &b, uint32_t dataPos, uint32_t lane)
{
    if constexpr (FMT == OALGPU_FMT_FLOAT)
    {
        const __attribute__((address_space(1))) float *p =
            reinterpret_cast<const __attribute__((address_space(1))) float*>((const __attribute__((address_space(1))) void*)b.data) + dataPos;
        const __attribute__((address_space(1))) float *pl = p + lane;
#pragma unroll
        for(int i = 0; i < kPre; ++i) pre[i] = pl[64 * i];
    }
    else
    {
        const __attribute__((address_space(1))) int16_t *p =
            reinterpret_cast<const __attribute__((address_space(1))) int16_t*>((const __attribute__((address_space(1))) void*)b.data) + dataPos;
        const __attribute__((address_space(1))) int16_t *pl = p + lane;
#pragma unroll
        for(int i = 0; i < kPre; ++i) pre[i] = __builtin_bit_cast(float, int32_t(pl[64 * i]));
    }
}

__device__ __forceinline__ bool GatherIsLinear(uint32_t count, const BufferItem &b, bool looping, uint32_t dataPos)
{
    return b.frameStep == 1u && uint64_t{dataPos} + uint32_t(kPre * 64) <= b.sampleLen
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

__device__ __forceinline__ void GatherStatic(float (&pre)[kPre], uint32_t count, const BufferItem &b, bool looping,
    uint32_t dataPos, uint32_t lane)
{
    if(GatherIsLinear(count, b, looping, dataPos))
    {
        if(b.fmt == OALGPU_FMT_FLOAT) GatherLinearT<OALGPU_FMT_FLOAT>(pre, b, dataPos, lane);
        else GatherLinearT<OALGPU_FMT_SHORT>(pre, b, dataPos, lane);
    }
    else if(b.fmt == OALGPU_FMT_FLOAT) GatherStaticT<OALGPU_FMT_FLOAT>(pre, count, b, looping, dataPos, lane);
    else GatherStaticT<OALGPU_FMT_SHORT>(pre, count, b, looping, dataPos, lane);
}

// What the first pass of LoadResampledSamples' loop will ask for (core/voice.cpp:600-640,
// :662-753), decided from the voice head alone.  `prefetch`: the plain case -- the voice mixes,
// has a buffer, starts at a non-negative position and its first chunk fits kPre*64 samples -- so
// the chunk can be gathered into registers ahead of time.
struct SrcPlan { bool prefetch; uint32_t bdst, bsrc; };

__device__ __forceinline__ SrcPlan PlanSource(const VoiceHead &h, uint32_t samplesToLoad)
{
    SrcPlan p{false, 0u, 0u};
    const bool mixes = h.playState == OALGPU_VOICE_PLAYING || h.playState == OALGPU_VOICE_STOPPING;
    if(!mixes || h.step < 1u) return p;
    CalcBufferSize(h.positionFrac, h.step, samplesToLoad, p.bdst, p.bsrc);
    p.prefetch = h.curBuffer >= 0 && h.position >= 0 && p.bsrc <= uint32_t(kPre * 64);
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
template<int M>
__device__ __forceinline__ void ResampleRunStaged(const f2 *tabF, const f2 *tabP, const float *rdb, uint32_t frac0,
    uint32_t increment, uint32_t bdst, float *out, float *sink, uint32_t lane)
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

template<int R, int TAPS>
__device__ __forceinline__ void ResampleRunStagedM(const WgLds<R, TAPS> &sm, const float *rdb, uint32_t m,
    uint32_t frac0, uint32_t increment, uint32_t bdst, float *out, float *sink, uint32_t lane)
{
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
template<int R, int TAPS>
__device__ __forceinline__ void LoadResampledWave(WgLds<R, TAPS> &sm, WaveLds<R, TAPS> &w, const DeviceLayout &L,
    uint32_t v, uint32_t lane, const VoiceHead &h, bool playing, uint32_t samplesToLoad, uint32_t samplesToMix,
    int32_t bufferItem, bool looping, const SrcPlan &plan)
{
    float *rdata = w.rd;
    float *srcBuffer = rdata + kMaxEdge;
    float *mixing = w.in + kHist;
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
            FillFromBuffer<64>(srcBuffer + srcDelay, bsrc - srcDelay, L.buffers[bufferItem], looping, upos, lane);
        }
        firstPass = false;
        WaveSync();
        if(L.phaseTimes && lane == 0 && loaded == 0) L.phaseTimes[size_t{v} * 8 + 7] = __builtin_readcyclecounter();

        // voice.cpp:764-769
        if((increment == kFracOne && fracPos == 0) || (L.ablate & 2u))
        {
            for(uint32_t k = lane; k < bdst; k += 64) mixing[loaded + k] = srcBuffer[k];
        }
        else if(staged)
        {
            ResampleRunStagedM(sm, rdata + (kMaxEdge - sL), sM, fracPos, increment, bdst, mixing + loaded,
                reinterpret_cast<float*>(&w.pad[0]), lane);
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
// sections in cascade; each is linear in its state s = (z1, z2): s' = A s + B x with
// A = [[-a1, 1], [-a2, 0]].  Lane l owns the run of samples [l*seg, (l+1)*seg) -- seg odd, so
// the per-lane runs hit distinct LDS banks -- and keeps it in registers for both sections:
//   (1) M = A^seg;  (2) forced response q_l of the run from a zero state;  (3) run-start states
//   S_l = M^l S_0 + sum_{k<l} M^(l-1-k) q_k by a 6-step Kogge-Stone scan with M, M^2, .. M^32;
//   (4) the true recurrence from S_l.  Lanes past the last sample only produce values nobody
// reads.  The serial loop of the reference differs from this by rounding only.
constexpr int kBqSeg = 17;                    // ceil(1024 / 64) | 1
struct S2 { float a, b; };

__device__ __forceinline__ float BqStep(S2 &s, float x, float b0, float b1, float b2, float a1, float a2)
{
    const float y = __builtin_fmaf(x, b0, s.a);
    s.a = __builtin_fmaf(x, b1, __builtin_fmaf(-y, a1, s.b));
    s.b = __builtin_fmaf(x, b2, -y * a2);
    return y;
}
__device__ __forceinline__ S2 Mv2(S2 c0, S2 c1, S2 v)          // [c0 c1] * v
{ return S2{__builtin_fmaf(c1.a, v.b, c0.a * v.a), __builtin_fmaf(c1.b, v.b, c0.b * v.a)}; }

// one section over the lane's run x[0..cnt); z1/z2: state in, state after the last sample out
__device__ __forceinline__ void BiquadWaveScan(float (&x)[kBqSeg], uint32_t cnt, uint32_t seg, const BiquadState &f,
    float &z1, float &z2, uint32_t lane, int lastLane)
{
    const float b0 = f.b0, b1 = f.b1, b2 = f.b2, a1 = f.a1, a2 = f.a2;
    S2 m0{1.0f, 0.0f}, m1{0.0f, 1.0f};                 // columns of M = A^seg
    for(uint32_t i = 0; i < seg; ++i)
    {
        m0 = S2{__builtin_fmaf(-a1, m0.a, m0.b), -a2 * m0.a};
        m1 = S2{__builtin_fmaf(-a1, m1.a, m1.b), -a2 * m1.a};
    }
    S2 e{0.0f, 0.0f};
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) if(uint32_t(i) < seg) BqStep(e, x[i], b0, b1, b2, a1, a2);
    S2 s0{z1, z2};                                    // -> M^lane S_0
    S2 p0 = m0, p1 = m1;                              // M^(2^step)
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
        if(uint32_t(i) < cnt) x[i] = BqStep(st, x[i], b0, b1, b2, a1, a2);
    z1 = __shfl(st.a, lastLane);
    z2 = __shfl(st.b, lastLane);
}

__device__ __forceinline__ void BiquadDualWaveScan(BiquadState &f0, BiquadState &f1, float *buf /* in place */,
    uint32_t n, uint32_t lane)
{
    const uint32_t seg = ((n + 63u) / 64u) | 1u;      // <= kBqSeg for n <= 1024
    const uint32_t begin = lane * seg < n ? lane * seg : n;
    const uint32_t cnt = (begin + seg < n) ? seg : n - begin;
    const int lastLane = int((n - 1u) / seg);
    float x[kBqSeg];
#pragma unroll
    for(int i = 0; i < kBqSeg; ++i) x[i] = (uint32_t(i) < cnt) ? buf[begin + i] : 0.0f;
    BiquadWaveScan(x, cnt, seg, f0, f0.z1, f0.z2, lane, lastLane);
    BiquadWaveScan(x, cnt, seg, f1, f1.z1, f1.z2, lane, lastLane);
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

// NfcFilterN::process, core/filters/nfc.cpp:222-288, section of order o over src[0..n) -> dst, on
// one lane in the reference's operation order (the sections are short recurrences; a block
// scan like the biquads' is the obvious next step)
__device__ __forceinline__ void NfcSerial(NfcState &st, uint32_t o, const float *src, float *dst, uint32_t n)
{
    const float a0 = st.a[o][0], a1 = st.a[o][1], a2 = st.a[o][2], a3 = st.a[o][3], a4 = st.a[o][4];
    const float b1 = st.b[o][1], b2 = st.b[o][2], b3 = st.b[o][3], b4 = st.b[o][4];
    float z0 = st.z[o][0], z1 = st.z[o][1], z2 = st.z[o][2], z3 = st.z[o][3];
    if(o == 1)
    {
        for(uint32_t i = 0; i < n; ++i)
        {
            const float y = src[i] * a0 - a1 * z0;
            dst[i] = y + b1 * z0;
            z0 += y;
        }
    }
    else
    {
        for(uint32_t i = 0; i < n; ++i)
        {
            const float y0 = src[i] * a0 - a1 * z0 - a2 * z1;
            const float out0 = y0 + b1 * z0 + b2 * z1;
            z1 += z0;
            z0 += y0;
            if(o == 2) { dst[i] = out0; continue; }
            if(o == 3)
            {
                const float y1 = out0 - a3 * z2;
                dst[i] = y1 + b3 * z2;
                z2 += y1;
                continue;
            }
            const float y1 = out0 - a3 * z2 - a4 * z3;
            dst[i] = y1 + b3 * z2 + b4 * z3;
            z3 += z2;
            z2 += y1;
        }
    }
    st.z[o][0] = z0; st.z[o][1] = z1; st.z[o][2] = z2; st.z[o][3] = z3;
}

// One line's share of a stream row's gain block (kernels.hpp LineBlockDwords): contributions of
// several MixSamples calls onto the same row and line add up -- the constant gains, and for
// the ramped frames the per-frame values (a contribution without a ramp adds its constant there)
struct RowLineGain {
    float gain{0.0f}, cur{0.0f}, step{0.0f};
    uint32_t fadeLen{0};
    __device__ __forceinline__ void add(const MixLineGain &g)
    {
        const float constant = g.steady ? g.tgt : 0.0f;
        gain += constant;
        if(g.fadeLen) { cur += g.cur; step += g.step; fadeLen = g.fadeLen; }
        else cur += constant;
    }
};
__device__ __forceinline__ void StoreRowBlock(uint32_t *blk, uint32_t ls, uint32_t lane, const RowLineGain &r, bool live)
{
    uint32_t maxFade = r.fadeLen;
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = uint32_t(__shfl_xor(int(maxFade), d)); maxFade = o > maxFade ? o : maxFade; }
    if(lane < ls)
    {
        blk[lane] = __builtin_bit_cast(uint32_t, r.gain);
        blk[ls + lane] = __builtin_bit_cast(uint32_t, r.fadeLen ? r.cur - r.gain : 0.0f);
        blk[2u * ls + lane] = __builtin_bit_cast(uint32_t, r.fadeLen ? r.step : 0.0f);
    }
    if(lane < 8u) blk[3u * ls + lane] = lane == 0u ? (live ? 1u : 0u) : (lane == 1u ? maxFade : 0u);
}

// NL == 0: HRTF voices (DoHrtfMix into the wave's register accumulator).
// NL == 1: voices panned onto dry lines (MixSamples, voice.cpp:962-963).  Holding N lines x 1024
// frames of accumulator per wavefront would take 16 N VGPRs per lane on top of the resampler's
// ~230 (measured: 209 spilled registers at 2 waves/SIMD, or 1 wave/SIMD and a latency-bound
// resampler), so this variant stops after DoFilters: it leaves the voice's 1024 filtered
// samples in HBM (L.streams, 4 KB per voice) together with the resolved MixLine gain of every
// line (L.lineGains), and LinesMixKernel below turns those into partial buses.
// SENDS: the context has auxiliary sends (voice.cpp:966-983): every send with a slot leaves a
// stream row too -- the unfiltered resampled samples shared by all sends (and the direct path)
// whose filter is inactive, or its own filtered copy -- with a gain block over the wet lines.
// Stream rows of a voice: [0] unfiltered, [1] direct-filtered (dry-line contexts), [2+i] send i filtered.
template<int R, int TAPS, int NL, bool SENDS>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(kWThreads, 2) VoiceWaveKernel(DeviceLayout L, uint32_t samplesToDo)
{
    using WL = WaveLds<R, TAPS>;
    __shared__ WgLds<R, TAPS> sm;
    const uint32_t t = threadIdx.x;
    const uint32_t lane0 = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t group = blockIdx.x;
    const uint32_t vpw = L.waveVoices;
    const uint32_t irStride = L.irStride;
    WL &w = sm.w[wave];
    const uint32_t N = samplesToDo;

    // The workgroup owns kWWaves*vpw consecutive voices; a wavefront takes every SECOND one of its
    // half of them (wave 0: v, v+2, ..; wave 1: v+1, v+3, ..).  Voices that cost more -- an active
    // filter, a replaced HRIR -- tend to come in regular patterns (every n-th source of a scene);
    // with consecutive voices per wavefront a period-4 pattern puts both expensive voices of a
    // group of four on the same wavefront, and the launch lasts as long as its slowest wavefront.
    const uint32_t vBegin = group * kWWaves * vpw + (wave & 1u) + 2u * (wave >> 1) * vpw;
    const uint32_t vEnd = (vBegin + 2u * vpw < L.numVoices) ? vBegin + 2u * vpw : L.numVoices;   // v = vBegin + 2k < vEnd
    const uint32_t vCount = vBegin < vEnd ? (vEnd - vBegin + 1u) / 2u : 0u;
    // profiling aid (OALGPU_PHASE_TIMES): per-wavefront stamps behind the per-voice ones
    auto waveStamp = [&](int slot)
    {
        if(L.phaseTimes && lane0 == 0)
            L.phaseTimes[size_t{L.numVoices} * 8 + size_t{group * kWWaves + wave} * 4 + slot] = __builtin_readcyclecounter();
    };
    waveStamp(0);
    // Two workgroups share a CU, and the launch fills the machine exactly once: workgroup g and
    // g + gridDim/2 land on the same CU (the dispatcher deals the first half one per CU, then the
    // second half).  Run in the same order, their wavefronts sit in the same phase at the same
    // time -- all eight in the LDS-bound resampler, then all eight in the VALU-bound FIR.  The
    // second half therefore takes its voices in reverse order: voices differ in cost, so the two
    // workgroups of a CU drift out of phase within the first voice and stay complementary.
    const bool rev = group >= (gridDim.x + 1u) / 2u;
    auto voiceAt = [&](uint32_t j) { return vBegin + 2u * (rev ? vCount - 1u - j : j); };

    // Voices are processed in passes; pass 0 only requests the first voice's source window and
    // stages the workgroup's resampler rows.  The request for the NEXT voice's window sits at one
    // point of the pass -- after this voice's FIR inputs are built, before its FIR runs -- so the
    // HBM latency of every window but the first is covered by ~1100 packed FMAs.
    VoiceHead headN{};
    VoiceTail tailN{};
    SrcPlan planN{false, 0u, 0u};
    BufferItem bufN{};
    bool loopingN = false;
    float fstC = 0.0f;                      // SENDS: the direct filter's state words, carried to the next pass
    if(vCount)
    {
        headN = LoadHeadScalar(L.ctl + voiceAt(0)); bufN = LoadCtlBufferScalar(L.ctl + voiceAt(0));
        if constexpr (NL == 0) tailN = LoadTailScalar(L.ctl + voiceAt(0));
    }
    // the voice whose resampler rows the workgroup stages (see the prologue below): every wavefront
    // reads its head itself, so that the choice needs no barrier
    const uint32_t keyVoice = group * kWWaves * vpw;
    VoiceHead headK{};
    if(keyVoice < L.numVoices) headK = LoadHeadScalar(L.ctl + keyVoice);

    f2 acc[R];
#pragma unroll
    for(int r = 0; r < R; ++r) acc[r] = f2{0.0f, 0.0f};
    f2 accO[WL::kQ];
#pragma unroll
    for(int q = 0; q < WL::kQ; ++q) accO[q] = f2{0.0f, 0.0f};

    for(uint32_t pass = 0; pass <= vCount; ++pass)
    {
        // The lane index is re-derived per pass behind an opaque move: addresses built from it are
        // then a few VALU per voice instead of loop invariants that the allocator, already at the
        // 256-register limit of 2 waves/SIMD, would keep in scratch.
        uint32_t lane = lane0;
        asm volatile("" : "+v"(lane));
        const bool first = pass == 0;
        // Instruction arbitration between the two wavefronts of a SIMD favours the older one: the
        // workgroup that arrived first runs at full speed, its partner on what is left (measured:
        // 62 K against 78 K cycles for the same work, and the launch lasts until the slower one
        // ends, running alone for the last 16 K).  The younger workgroup therefore takes priority
        // for the first half of its voices and hands it back for the second.
        if(rev)
        {
            if(pass == 0) __builtin_amdgcn_s_setprio(1);
            else if(pass == vCount / 2u + 1u) __builtin_amdgcn_s_setprio(0);
        }
        const uint32_t v = first ? 0u : voiceAt(pass - 1u);  // meaningless in pass 0
        const bool haveNext = pass < vCount;
        const uint32_t vn = haveNext ? voiceAt(pass) : 0u;  // the voice to request (the first one in pass 0)
        // The next voice's request: these registers live from the request to the end of the pass,
        // where they are parked in LDS.
        float preN[kPre];
        float prevN = 0.0f, histN = 0.0f, fstN = 0.0f;
        f2 oldN[TAPS / 64];
        bool dirtyN = false;
#pragma unroll
        for(int q = 0; q < TAPS / 64; ++q) oldN[q] = f2{0.0f, 0.0f};
#pragma unroll
        for(int i = 0; i < kPre; ++i) preN[i] = 0.0f;
        auto requestNext = [&]()
        {
            if(haveNext)
            {   // its head is in headN
                planN = PlanSource(headN, N);
                loopingN = false;
                if(headN.curBuffer >= 0)
                {
                    // voice.cpp:1015-1019: a position at or past the loop end plays on without looping
                    loopingN = headN.loopBuffer >= 0 && !(headN.position >= 0 && uint32_t(headN.position) >= bufN.loopEnd);
                    planN.prefetch = planN.prefetch && GatherCovers(planN.bsrc, bufN, loopingN, uint32_t(headN.position));
                }
                // (the window first: each gather variant starts by waiting for older loads into its
                // registers -- the variants share them -- and must not find a fresh one in front of it)
                if(planN.prefetch)
                {
                    GatherStatic(preN, planN.bsrc, bufN, loopingN, uint32_t(headN.position), lane);
                    prevN = (lane < kMaxPad) ? L.prev[size_t{vn} * kMaxPad + lane] : 0.0f;
                }
                fstN = (lane < 32u) ? reinterpret_cast<const float*>(L.dfilt + size_t{vn} * 2)[lane] : 0.0f;
                if constexpr (NL == 0)
                {
                    histN = L.hist[size_t{vn} * kHist + lane];
                    dirtyN = (headN.flags & kFlagHrtfDirty) != 0;
                    const f2 *oc = reinterpret_cast<const f2*>(L.hrtfOld + size_t{vn} * irStride * 2);
    #pragma unroll
                    for(int q = 0; q < TAPS / 64; ++q)
                        oldN[q] = (dirtyN && lane + 64u * q < irStride) ? oc[lane + 64u * q] : f2{0.0f, 0.0f};
                }
            }
        };

        // ---------------- part 1: this voice up to its FIR inputs ----------------
        bool active = false;
        VoiceHead head{};
        BufferItem buf{};
        bool looping = false, playing = false, dirty = false, oldPass = false;
        VoiceTail tail{};
        uint32_t counter = 0, fademix = 0, todo = 0;
        float endGain = 0.0f, gainAfterBlend = 0.0f;
        int32_t bufferItem = -1;
        if(!first)
        {
            // head, plan, buffer and the gathered window were requested one pass ago
            head = headN;
            tail = tailN;
            buf = bufN;
            looping = loopingN;
            // next voice's head (and the HRTF delays/gains behind it): in flight while this voice resamples
            if(haveNext)
            {
                headN = LoadHeadScalar(L.ctl + vn); bufN = LoadCtlBufferScalar(L.ctl + vn);
                if constexpr (NL == 0) tailN = LoadTailScalar(L.ctl + vn);
            }

            const int vstate = head.playState;
            const bool mixes = vstate == OALGPU_VOICE_PLAYING || vstate == OALGPU_VOICE_STOPPING;
            playing = vstate == OALGPU_VOICE_PLAYING;
            active = mixes && head.step >= 1u;
            // voice.cpp:1002-1010
            if(mixes && !active && !playing && lane == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
            if constexpr (SENDS || NL > 0)
            {   // nothing to mix for this voice in this update
                if(!active && lane < L.streamsPerVoice)
                    L.lineGains[(size_t{v} * L.streamsPerVoice + lane) * LineBlockDwords(L.lineStride) + 3u * L.lineStride] = 0u;
            }
        }
        auto stamp = [&](int slot)
        {
            if(L.phaseTimes && lane == 0) L.phaseTimes[size_t{v} * 8 + slot] = __builtin_readcyclecounter();
        };
        if(active)
        {
            stamp(0);
            bufferItem = head.curBuffer;
            dirty = (head.flags & kFlagHrtfDirty) != 0;

            // Hrtf.History (w.in[0..63]), the direct filter state (w.fst; SENDS: fstC), a replaced
            // filter's old coefficients (w.cold) and the source window were parked by the last pass
            const float fstv = fstC;
            const SrcPlan plan = planN;
            LoadResampledWave(sm, w, L, v, lane, head, playing, N, N, bufferItem, looping, plan);
            asm volatile("" : "+v"(lane));      // addresses used from here on are rebuilt, not carried across the resampler
            if constexpr (NL > 0) requestNext();
            if(head.flags & kFlagAmbiScale)
            {   // VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091
                const AmbiScaleState a = L.ambi[v];
                SplitterState sp{a.coeff, a.lpZ1, a.lpZ2, a.apZ1};
                WaveSync();
                SplitterScan<false>(sp, w.in + kHist, N, a.hfScale, a.lfScale, lane);
                WaveSync();
                if(lane == 0) { L.ambi[v].lpZ1 = sp.lpZ1; L.ambi[v].lpZ2 = sp.lpZ2; L.ambi[v].apZ1 = sp.apZ1; }
            }

            stamp(1);
            counter = (head.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093

            const bool directFilter = (head.flags & kFlagDirectFilter) && !(L.ablate & 8u);
            if constexpr (SENDS || NL > 0)
            {   // ---- stream rows that must leave before the direct filter overwrites w.in
                const uint32_t ls = L.lineStride, spv = L.streamsPerVoice, numSends = L.numSends, wetCh = L.wetChannels;
                const uint32_t wetBase = L.hrtf ? 0u : L.numDry;
                float *rowsV = L.streams + size_t{v} * spv * kLine;
                uint32_t *blkV = L.lineGains + size_t{v} * spv * LineBlockDwords(ls);
                RowLineGain row0;                       // the unfiltered row's merged gains, line = lane
                bool row0Live = false;
                if constexpr (NL > 0)
                {
                    if(!directFilter)
                    {   // MixSamples onto the dry lines rides on the unfiltered row (DoNfcMix: only
                        // the W line is mixed from the voice's own samples, voice.cpp:908-909)
                        const uint32_t nd = (L.nfc && (head.flags & kFlagNfc)) ? 1u : L.numDry, ndAll = L.numDry;
                        float tg = 0.0f, cu = 0.0f;
                        if(lane < nd)
                        {
                            tg = playing ? L.gainTgt[size_t{v} * ndAll + lane] : 0.0f;    // SilentCoeffs when Stopping
                            cu = counter ? L.gainCur[size_t{v} * ndAll + lane] : tg;      // voice.cpp:1094-1112
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        if(lane < nd) { L.gainCur[size_t{v} * ndAll + lane] = g.newCur; row0.add(g); }
                        row0Live = true;
                    }
                }
                if constexpr (SENDS)
                {
                    for(uint32_t si = 0; si < numSends; ++si)
                    {
                        const int32_t slot = L.ctl[v].sendSlot[si];
                        uint32_t *blkS = blkV + size_t{2u + si} * LineBlockDwords(ls);
                        if(slot < 0) { if(lane == 0) blkS[3u * ls] = 0u; continue; }
                        const bool sendFilter = (head.flags >> (kFlagSendFilterShift + si)) & 1u;
                        // the send's gains onto its slot's wet lines (voice.cpp:978-979)
                        const uint32_t base = wetBase + uint32_t(slot) * wetCh;
                        const bool mine = lane >= base && lane < base + wetCh;
                        float tg = 0.0f, cu = 0.0f;
                        float *curp = L.sendCur + (size_t{v} * numSends + si) * wetCh + (lane - base);
                        if(mine)
                        {
                            tg = playing ? L.sendTgt[(size_t{v} * numSends + si) * wetCh + (lane - base)] : 0.0f;
                            cu = counter ? *curp : tg;
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        if(mine) *curp = g.newCur;
                        // filter state of this send
                        BiquadSlot *slots = &L.sfilt[(size_t{v} * numSends + si) * 2];
                        WaveSync();
                        if(lane < 32u) w.fst[lane] = reinterpret_cast<const float*>(slots)[lane];
                        WaveSync();
                        if(sendFilter)
                        {   // its own filtered copy, built in the (currently unused) resampler scratch
                            float *tmp = w.rd;
                            for(uint32_t k = lane; k < N; k += 64) tmp[k] = w.in[kHist + k];
                            WaveSync();
                            WaveDoFilters(w.fst, slots, true, tmp, N, lane);
                            WaveSync();
                            float *dst = rowsV + size_t{2u + si} * kLine;
                            for(uint32_t k = lane; k < uint32_t(kLine); k += 64) dst[k] = (k < N) ? tmp[k] : 0.0f;
                            RowLineGain r;
                            if(mine) r.add(g);
                            StoreRowBlock(blkS, ls, lane, r, true);
                        }
                        else
                        {
                            WaveDoFilters(w.fst, slots, false, w.in + kHist, N, lane);
                            if(mine) row0.add(g);
                            row0Live = true;
                            if(lane == 0) blkS[3u * ls] = 0u;
                        }
                    }
                }
                if(row0Live)
                    for(uint32_t k = lane; k < uint32_t(kLine); k += 64) rowsV[k] = (k < N) ? w.in[kHist + k] : 0.0f;
                StoreRowBlock(blkV, ls, lane, row0, row0Live);
                WaveSync();
            }

            // ---- DoFilters, direct path (voice.cpp:255-267): in place on w.in[kHist..]
            {
                if constexpr (SENDS) { if(lane < 32u) w.fst[lane] = fstv; }
                WaveSync();
                WaveDoFilters(w.fst, &L.dfilt[size_t{v} * 2], directFilter, w.in + kHist, N, lane);
                WaveSync();
            }
            if constexpr (NL > 0)
            {   // the direct-filtered row (voice.cpp:962-963 after an active DoFilters)
                const bool nfcV = L.nfc && (head.flags & kFlagNfc);
                const uint32_t ls = L.lineStride, spv = L.streamsPerVoice, nd = nfcV ? 1u : L.numDry;
                uint32_t *blk1 = L.lineGains + (size_t{v} * spv + 1u) * LineBlockDwords(ls);
                if(directFilter)
                {
                    float *dst = L.streams + (size_t{v} * spv + 1u) * kLine;
                    for(uint32_t k = lane; k < uint32_t(kLine); k += 64) dst[k] = (k < N) ? w.in[kHist + k] : 0.0f;
                    float tg = 0.0f, cu = 0.0f;
                    if(lane < nd)
                    {
                        tg = playing ? L.gainTgt[size_t{v} * L.numDry + lane] : 0.0f;
                        cu = counter ? L.gainCur[size_t{v} * L.numDry + lane] : tg;
                    }
                    const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                    RowLineGain r;
                    if(lane < nd) { L.gainCur[size_t{v} * L.numDry + lane] = g.newCur; r.add(g); }
                    StoreRowBlock(blk1, ls, lane, r, true);
                }
                else if(lane == 0) blk1[3u * ls] = 0u;
                if(L.nfc)
                {   // DoNfcMix, voice.cpp:911-931: one row per ambisonic order above 0, the voice's
                    // (direct-filtered) samples through that order's NFC section
                    const uint32_t rowBase = 2u + L.numSends, ndAll = L.numDry;
                    uint32_t line = 1;
                    for(uint32_t o = 1; o <= L.nfcOrders; ++o)
                    {
                        uint32_t *blkN = L.lineGains + (size_t{v} * spv + rowBase + o - 1u) * LineBlockDwords(ls);
                        if(!nfcV) { if(lane == 0) blkN[3u * ls] = 0u; continue; }
                        const uint32_t cnt = L.chansPerOrder[o];
                        WaveSync();
                        if(lane == 0) NfcSerial(L.nfc[v], o, w.in + kHist, w.rd, N);
                        WaveSync();
                        float *dst = L.streams + (size_t{v} * spv + rowBase + o - 1u) * kLine;
                        for(uint32_t k = lane; k < uint32_t(kLine); k += 64) dst[k] = (k < N) ? w.rd[k] : 0.0f;
                        const bool mine = lane >= line && lane < line + cnt;
                        float tg = 0.0f, cu = 0.0f;
                        if(mine)
                        {
                            tg = playing ? L.gainTgt[size_t{v} * ndAll + lane] : 0.0f;
                            cu = counter ? L.gainCur[size_t{v} * ndAll + lane] : tg;
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        RowLineGain r;
                        if(mine) { L.gainCur[size_t{v} * ndAll + lane] = g.newCur; r.add(g); }
                        StoreRowBlock(blkN, ls, lane, r, true);
                        line += cnt;
                    }
                }
            }

            stamp(2);
            if constexpr (NL == 0)
            {
            // ---- DoHrtfMix, voice.cpp:827-902
            WaveSync();
            if(playing) L.hist[size_t{v} * kHist + lane] = w.in[N + lane];

            const float targetGain = tail.tgtGain * (playing ? 1.0f : 0.0f);
            const float oldGain = counter ? tail.oldGain : tail.tgtGain;   // voice.cpp:1100
            float blendGain = targetGain;
            if(counter)
            {
                fademix = N < counter ? N : counter;
                if(counter > fademix)
                    blendGain = lerpf(oldGain, targetGain, float(fademix) / float(counter));
            }
            const float newStep = fademix ? blendGain / float(fademix) : 0.0f;
            gainAfterBlend = fademix ? blendGain : oldGain;
            todo = N - fademix;
            endGain = targetGain;
            if(todo && counter > N)
                endGain = lerpf(gainAfterBlend, targetGain, float(todo) / float(counter - fademix));
            const float mainStep = todo ? (endGain - gainAfterBlend) / float(todo) : 0.0f;
            const bool oldOn = fademix && oldGain > kGainSilence;
            const bool newOn = fademix && newStep * float(fademix) > kGainSilence;
            const uint32_t odL = tail.oldDelay[0], odR = tail.oldDelay[1];
            const uint32_t dL = tail.tgtDelay[0], dR = tail.tgtDelay[1];
            const float oldStep = fademix ? oldGain / float(fademix) : 0.0f;
            // A voice whose target filter was not replaced since its last mix has Old == Target
            // (coefficients and delays; voice.cpp:869 / :1100), so the old-filter fade-out and
            // the new-filter fade-in of MixHrtfBlend act on the same taps and their gains are
            // summed.
            const bool merged = !dirty;

            // x'[i] = (In[64 - dL + i], In[64 - dR + i]) * g(i); zero pads on both sides
            w.x2[lane] = f2{0.0f, 0.0f};
            if(TAPS > 64) w.x2[64 + lane] = f2{0.0f, 0.0f};
            for(uint32_t k = TAPS + N + lane; k < uint32_t(WL::kX); k += 64) w.x2[k] = f2{0.0f, 0.0f};
            if(!(L.ablate & 16u))
            {
                const float *inL = w.in + (kHist - dL), *inR = w.in + (kHist - dR);
                if(lane < N)
                {   // i = lane: the only pass that can touch the fade (fademix <= 64)
                    const uint32_t i = lane;
                    float g;
                    if(i < fademix)
                    {
                        g = newOn ? newStep * float(i) : 0.0f;
                        if(merged && oldOn) g += oldStep * float(fademix - i);
                    }
                    else g = gainAfterBlend + mainStep * float(i - fademix);
                    w.x2[TAPS + i] = f2{inL[i] * g, inR[i] * g};
                }
                const float gbase = gainAfterBlend - mainStep * float(fademix);
                // all reads first, then all writes: in and x2 are members of one LDS object, so
                // the compiler keeps a read behind every earlier write (one LDS round trip per
                // frame row otherwise).  lane + 64 j <= 1023: inside w.in for any N.
                float xl[kLine / 64 - 1], xr[kLine / 64 - 1];
#pragma unroll
                for(int j = 0; j < kLine / 64 - 1; ++j)
                {
                    const uint32_t i = lane + 64u * uint32_t(j + 1);
                    xl[j] = inL[i]; xr[j] = inR[i];
                }
#pragma unroll
                for(int j = 0; j < kLine / 64 - 1; ++j)
                {
                    const uint32_t i = lane + 64u * uint32_t(j + 1);
                    const float g = __builtin_fmaf(mainStep, float(i), gbase);
                    if(i < N) w.x2[TAPS + i] = f2{xl[j] * g, xr[j] * g};
                }
            }
            // old-filter fade-out inputs (one per lane) and coefficients, replaced filters only
            oldPass = !merged && oldOn;
            if(oldPass)
            {
                f2 xo = {0.0f, 0.0f};
                if(lane < fademix)
                {
                    const float g = oldStep * float(fademix - lane);
                    xo = f2{w.in[kHist - odL + lane] * g, w.in[kHist - odR + lane] * g};
                }
                w.xo[lane] = xo;
            }
            WaveSync();
            }
            stamp(3);
        }

        // ---------------- the next voice's source window leaves HBM now ----------------
        // NL == 0: before the FIR, whose ~1100 packed FMAs cover the latency.  NL > 0 (no FIR in
        // this kernel): right after the resampler, ahead of the filters and the stream-row stores.
        if constexpr (NL > 0) { if(!active) requestNext(); }
        else requestNext();

        if(first)
        {   // ---- workgroup prologue: pick and stage the resampler rows most voices will use
            // Normally those of the workgroup's first voice, whose head every wavefront has read
            // itself (no barrier, no second round trip at kernel start); if that voice does not
            // qualify, wavefront 0 looks for one that does.
            const uint32_t gBegin = group * kWWaves * vpw;
            const int kK = headK.rsKind;
            const uint32_t mK = kK == 2 ? 4u : headK.rsM, lK = kK == 2 ? 1u : headK.rsL;
            const bool eligK = keyVoice < L.numVoices && (kK == 2 || (kK == 3 && (mK == 12 || mK == 24 || mK == 48)))
                && (headK.playState == OALGPU_VOICE_PLAYING || headK.playState == OALGPU_VOICE_STOPPING);
            uint32_t key = headK.rsFilterOffset * 8u + uint32_t(kK), m = mK;
            if(eligK) { if(t == 0) { sm.tabKey = key; sm.tabM = mK; sm.tabL = lK; } }
            else
            {
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
                    const int firstLane = __ffsll((long long)mask) - 1;
                    const uint32_t key = uint32_t(__shfl(int(off * 8u + uint32_t(kind)), firstLane));
                    const uint32_t fm = uint32_t(__shfl(int(m), firstLane)), fl = uint32_t(__shfl(int(l), firstLane));
                    if(lane == 0) { sm.tabKey = key; sm.tabM = fm; sm.tabL = fl; }
                }
                else if(lane == 0) { sm.tabKey = 0xffffffffu; sm.tabM = 0; sm.tabL = 0; }
            }
            __syncthreads();
            key = sm.tabKey; m = sm.tabM;
            }
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

        // ---------------- part 2: FIR and state write-back ----------------
        if(active)
        {
            stamp(4);
            cf16 *co = (cf16*)(uintptr_t)(L.hrtfTgt + size_t{v} * irStride * 2);
            if(NL > 0 || (L.ablate & 1u)) {}
            else if(irStride == uint32_t(TAPS))
                FirMainPk<R, TAPS>(acc, &w.x2[TAPS + R * lane], co);
            else
            {   // other HRIR lengths: 16-tap segments (irStride is a multiple of 16, zero padded)
                const f2 *xw = &w.x2[TAPS + R * lane];
                for(uint32_t seg = 0; seg * 16u < irStride; ++seg)
                    FirMainPk<R, 16>(acc, xw - 16 * seg, co + 2 * seg);
            }
            if(NL == 0 && oldPass && !(L.ablate & 1u))
            {   // frames lane + 64q receive cOld[lane + 64q - i] * xo[i], i < 64
#pragma unroll 1
                for(int i0 = 0; i0 < 64; i0 += 8)
                {
#pragma unroll
                    for(int ii = 0; ii < 8; ++ii)
                    {
                        const int i = i0 + ii;
                        const f2 xi = w.xo[i];           // same address in every lane: LDS broadcast
#pragma unroll
                        for(int q = 0; q < WL::kQ; ++q) accO[q] = pkfma(w.cold[64 + lane + 64 * q - i], xi, accO[q]);
                    }
                }
            }
        }

        // ---------------- the next voice's state is parked in LDS ----------------
        // After the FIR every LDS word the next pass starts from is free (rd shares x2; in[0..63],
        // fst and cold were last read above), and what was requested before the FIR has landed.
        // Parked here, ahead of this voice's write-back stores, no wait of the next pass has to
        // sit behind those stores, and the ~21 registers of the request are dead outside the FIR.
        if(haveNext)
        {
            WaveSync();
            if(planN.prefetch)
            {
                const bool isShort = bufN.fmt == OALGPU_FMT_SHORT;
                if(lane < kMaxPad) w.rd[lane] = prevN;
#pragma unroll
                for(int i = 0; i < kPre; ++i)
                {
                    // all kPre*64 words (rd has room; the resampler never reads past bsrc + padding)
                    w.rd[kMaxEdge + lane + 64u * uint32_t(i)] = GatherDecode(preN[i], isShort);
                }
            }
            if constexpr (NL == 0)
            {
                w.in[lane] = histN;
                if(dirtyN)
                {
#pragma unroll
                    for(int q = 0; q < TAPS / 64; ++q) w.cold[64 + lane + 64 * q] = oldN[q];
                }
            }
            if constexpr (SENDS) fstC = fstN;
            else if(lane < 32u) w.fst[lane] = fstN;
            WaveSync();
        }

        if(first) waveStamp(1);
        if(active)
        {
            stamp(5);
            // voice.cpp:1094-1101 / :869-873,900: Old <- Target, Old.Gain <- reached gain
            if(NL == 0 && dirty && (counter == 0 || fademix))
            {
                const float *tg = L.hrtfTgt + size_t{v} * irStride * 2;
                float *od = L.hrtfOld + size_t{v} * irStride * 2;
                for(uint32_t k = lane; k < irStride * 2; k += 64) od[k] = tg[k];
            }
            // ---- voice.cpp:1116-1232: flags, position, loop wrap / end of buffer ----
            if(lane == 0)
            {
                VoiceCtl &c = L.ctl[v];
                if(NL == 0)
                {
                    if(counter == 0 || fademix) { c.hrtfOldDelay[0] = tail.tgtDelay[0]; c.hrtfOldDelay[1] = tail.tgtDelay[1]; }
                    c.hrtfOldGain = todo ? endGain : gainAfterBlend;
                }
                uint32_t flags = head.flags | kFlagFading;
                if(NL > 0 || counter == 0 || fademix) flags &= ~kFlagHrtfDirty;
                c.flags = flags;
                if(!playing) c.playState = OALGPU_VOICE_STOPPED;
                else
                {
                    int32_t bufPosInt = head.position;
                    uint32_t bufPosFrac = head.positionFrac + head.step * N;
                    const uint32_t samplesDone = bufPosFrac >> kFracBits;
                    bufPosInt = AddSat(bufPosInt, int32_t(samplesDone));
                    bufPosFrac &= kFracMask;
                    if(bufferItem >= 0 && bufPosInt > 0)
                    {
                        if(looping)
                        {
                            uint32_t pos = uint32_t(bufPosInt);
                            if(pos >= buf.loopEnd)
                            {
                                pos = ((pos - buf.loopStart) % (buf.loopEnd - buf.loopStart)) + buf.loopStart;
                                bufPosInt = int32_t(pos);
                            }
                        }
                        else if(uint32_t(bufPosInt) >= buf.sampleLen)
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
            stamp(6);
        }
    }

    waveStamp(2);
    if constexpr (NL > 0) return;
    // ---- one partial per workgroup: waves dump their accumulators, then a fixed-order sum
    {
        f2 *dump = w.x2;                     // [frame] = (L, R), frames < 64R
        WaveSync();
#pragma unroll
        for(int r = 0; r < R; ++r) dump[R * lane0 + r] = acc[r];
        WaveSync();
#pragma unroll
        for(int q = 0; q < WL::kQ; ++q)
        {
            const f2 cur = dump[lane0 + 64 * q];
            dump[lane0 + 64 * q] = f2{cur.x + accO[q].x, cur.y + accO[q].y};
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
    waveStamp(3);
}

// ---- MixSamples of every stream row onto the mix lines -------------------------------------------
// out[c][f] += row_r[f] * gain_{r,c}(f) over the live rows of one voice group, in row order:
// thread t owns frame 256*blockIdx.x + t of every line, blockIdx.y is the voice group.  The
// group's live rows are compacted into an LDS list first (dead rows -- idle voices, rows a voice
// does not use -- cost one flag read, not 4 KB); the rows are then read once, coalesced, a batch
// in flight, and multiplied by their constant gain vectors (LDS, same address in all lanes).
// One partial bus per group, summed by BusReduceKernel in group order: deterministic.
// A gain ramp (MixLine with Counter <= 64, voice.cpp:1093) only ever covers the first 64 frames:
// LinesRampKernel adds s * (ramp(f) - constant) to those frames of the partial bus afterwards.
template<int S>                                   // gain vector width: mix lines padded to 8 / 16 / 32
__global__ void __launch_bounds__(256) LinesMixKernel(DeviceLayout L, uint32_t samplesToDo)
{
    constexpr uint32_t kBlk = 3u * S + 8u;         // LineBlockDwords(S)
    constexpr uint32_t kMixBatch = 16u;            // stream rows in flight per thread
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn[];
    // dyn: gains[maxRows][S] | rowIdx[maxRows]
    __shared__ uint32_t waveBase[4];
    const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
    const uint32_t f = blockIdx.x * 256u + t;
    const uint32_t g = blockIdx.y;
    const uint32_t spv = L.streamsPerVoice;
    const uint32_t per = (L.numVoices + L.numLineGroups - 1u) / L.numLineGroups;
    const uint32_t v0 = g * per;
    const uint32_t nv = (v0 < L.numVoices) ? ((v0 + per < L.numVoices) ? per : L.numVoices - v0) : 0u;
    const uint32_t maxRows = per * spv;
    float *gains = reinterpret_cast<float*>(dyn);
    uint32_t *rowIdx = dyn + size_t{maxRows} * S;
    const uint32_t *blk0 = L.lineGains + size_t{v0} * spv * kBlk;

    // ---- compact the live rows in row order: ballot + prefix per pass of 256 potential rows
    uint32_t nLive = 0;
    for(uint32_t r0 = 0; r0 < nv * spv; r0 += 256)
    {
        const uint32_t r = r0 + t;
        const bool live = r < nv * spv && blk0[size_t{r} * kBlk + 3u * S] != 0u;
        const unsigned long long ml = __ballot(live);
        if(lane == 0) waveBase[wave] = uint32_t(__popcll(ml));
        __syncthreads();
        uint32_t bl = nLive, tl = 0;
        for(uint32_t w = 0; w < 4; ++w) { if(w < wave) bl += waveBase[w]; tl += waveBase[w]; }
        if(live) rowIdx[bl + uint32_t(__popcll(ml & ((1ull << lane) - 1ull)))] = r;
        nLive += tl;
        __syncthreads();
    }
    // ---- their gain vectors
    for(uint32_t k = t; k < nLive * S; k += 256)
        gains[k] = __builtin_bit_cast(float, blk0[size_t{rowIdx[k / S]} * kBlk + (k % S)]);
    __syncthreads();

    float acc[S];
#pragma unroll
    for(int c = 0; c < S; ++c) acc[c] = 0.0f;
    const float *rows = L.streams + size_t{v0} * spv * kLine + f;
    for(uint32_t b = 0; b < nLive; b += kMixBatch)
    {
        float s[kMixBatch];
#pragma unroll
        for(uint32_t k = 0; k < kMixBatch; ++k)
            s[k] = rows[size_t{rowIdx[(b + k < nLive) ? b + k : nLive - 1u]} * kLine];
#pragma unroll
        for(uint32_t k = 0; k < kMixBatch; ++k)
        {
            const bool in = b + k < nLive;
            const float sk = in ? s[k] : 0.0f;
            const float4 *g4 = reinterpret_cast<const float4*>(gains + size_t{in ? b + k : nLive - 1u} * S);
#pragma unroll
            for(int q = 0; q < S / 4; ++q)
            {
                const float4 x = g4[q];
                acc[4 * q] = __builtin_fmaf(sk, x.x, acc[4 * q]);
                acc[4 * q + 1] = __builtin_fmaf(sk, x.y, acc[4 * q + 1]);
                acc[4 * q + 2] = __builtin_fmaf(sk, x.z, acc[4 * q + 2]);
                acc[4 * q + 3] = __builtin_fmaf(sk, x.w, acc[4 * q + 3]);
            }
            // keep the gain reads of later rows from being hoisted above (register pressure)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float *pl = L.partLines + size_t{g} * L.mixLines * kLine;
#pragma unroll
    for(int c = 0; c < S; ++c)
        if(uint32_t(c) < L.mixLines) pl[size_t(c) * kLine + f] = (f < samplesToDo) ? acc[c] : 0.0f;
}

// One wavefront per voice group: lane = frame (< 64).  The group's rows with a ramp are listed
// first (64 potential rows per ballot); then 16 of them at a time have their sample and their
// ramp vectors (A | B, 2S <= 64 floats: one coalesced load per row) in flight together, the
// vectors go through LDS so that every lane can read every line's pair.
constexpr uint32_t kRampListMax = 2048;           // >= 60000 / (4 * (1 + 8)) potential rows per group
template<int S>
__global__ void __launch_bounds__(64) LinesRampKernel(DeviceLayout L, uint32_t samplesToDo)
{
    constexpr uint32_t kBlk = 3u * S + 8u;
    constexpr uint32_t kChunk = 16;
    __shared__ __attribute__((aligned(16))) float ab[kChunk][2 * S];
    __shared__ uint32_t rampIdx[kRampListMax];
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x;
    const uint32_t spv = L.streamsPerVoice;
    const uint32_t per = (L.numVoices + L.numLineGroups - 1u) / L.numLineGroups;
    const uint32_t v0 = g * per;
    const uint32_t nv = (v0 < L.numVoices) ? ((v0 + per < L.numVoices) ? per : L.numVoices - v0) : 0u;
    const uint32_t *blk0 = L.lineGains + size_t{v0} * spv * kBlk;
    const float *rows = L.streams + size_t{v0} * spv * kLine + lane;
    const float ff = float(lane);
    uint32_t n = 0;
    for(uint32_t r0 = 0; r0 < nv * spv; r0 += 64)
    {
        const uint32_t r = r0 + lane;
        uint2 fl = make_uint2(0u, 0u);
        if(r < nv * spv) fl = *reinterpret_cast<const uint2*>(blk0 + size_t{r} * kBlk + 3u * S);
        const bool ramp = fl.x != 0u && fl.y != 0u;
        const unsigned long long mr = __ballot(ramp);
        if(ramp) rampIdx[n + uint32_t(__popcll(mr & ((1ull << lane) - 1ull)))] = r;
        n += uint32_t(__popcll(mr));
    }
    if(n == 0u) return;
    WaveSync();
    float acc[S];
#pragma unroll
    for(int c = 0; c < S; ++c) acc[c] = 0.0f;
    for(uint32_t b = 0; b < n; b += kChunk)
    {
        const uint32_t m = (n - b < kChunk) ? n - b : kChunk;
        float sv[kChunk], abv[kChunk];
        uint32_t len[kChunk];
#pragma unroll
        for(uint32_t j = 0; j < kChunk; ++j)
        {
            const uint32_t rr = rampIdx[b + ((j < m) ? j : m - 1u)];
            sv[j] = rows[size_t{rr} * kLine];
            abv[j] = (lane < 2u * S) ? __builtin_bit_cast(float, blk0[size_t{rr} * kBlk + S + lane]) : 0.0f;
            len[j] = blk0[size_t{rr} * kBlk + 3u * S + 1u];
        }
        WaveSync();
#pragma unroll
        for(uint32_t j = 0; j < kChunk; ++j) if(lane < 2u * S) ab[j][lane] = abv[j];
        WaveSync();
#pragma unroll
        for(uint32_t j = 0; j < kChunk; ++j)
        {
            if(j >= m) continue;
            const float sk = (lane < len[j]) ? sv[j] : 0.0f;
            const float4 *a4 = reinterpret_cast<const float4*>(&ab[j][0]);
            const float4 *b4 = reinterpret_cast<const float4*>(&ab[j][S]);
#pragma unroll
            for(int q = 0; q < S / 4; ++q)
            {
                const float4 a = a4[q], bb = b4[q];
                acc[4 * q] = __builtin_fmaf(sk, a.x + bb.x * ff, acc[4 * q]);
                acc[4 * q + 1] = __builtin_fmaf(sk, a.y + bb.y * ff, acc[4 * q + 1]);
                acc[4 * q + 2] = __builtin_fmaf(sk, a.z + bb.z * ff, acc[4 * q + 2]);
                acc[4 * q + 3] = __builtin_fmaf(sk, a.w + bb.w * ff, acc[4 * q + 3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float *pl = L.partLines + size_t{g} * L.mixLines * kLine;
#pragma unroll
    for(int c = 0; c < S; ++c)
        if(uint32_t(c) < L.mixLines && lane < samplesToDo) pl[size_t(c) * kLine + lane] += acc[c];
}

} // namespace

bool WaveKernelApplies(bool exact, const DeviceLayout &L)
{
    if(exact || L.mixLines > 32 || L.numSends > 6) return false;
    if(L.hrtf) return L.irStride >= 8 && L.irStride <= 128;
    return L.numDry >= 1;
}

const char *WaveKernelName(const DeviceLayout &L)
{
    const bool sends = L.numSends != 0;
    if(!L.hrtf) return sends ? "VoiceWaveKernel<17, 64, 1, true>" : "VoiceWaveKernel<17, 64, 1, false>";
    if(L.irStride <= 64) return sends ? "VoiceWaveKernel<17, 64, 0, true>" : "VoiceWaveKernel<17, 64, 0, false>";
    return sends ? "VoiceWaveKernel<18, 128, 0, true>" : "VoiceWaveKernel<18, 128, 0, false>";
}

uint32_t WaveKernelGroups(const DeviceLayout &L)
{ return (L.numVoices + kWWaves * L.waveVoices - 1u) / (kWWaves * L.waveVoices); }

hipError_t LaunchVoiceWave(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo)
{
    const uint32_t groups = WaveKernelGroups(L);
    const bool sends = L.numSends != 0;
    const dim3 grid(groups), block(kWThreads);
    if(!L.hrtf)
    {
        if(sends) hipLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, true>), grid, block, 0, s, L, samplesToDo);
        else hipLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, false>), grid, block, 0, s, L, samplesToDo);
    }
    else if(L.irStride <= 64)
    {
        if(sends) hipLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, true>), grid, block, 0, s, L, samplesToDo);
        else hipLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, false>), grid, block, 0, s, L, samplesToDo);
    }
    else
    {
        if(sends) hipLaunchKernelGGL((VoiceWaveKernel<18, 128, 0, true>), grid, block, 0, s, L, samplesToDo);
        else hipLaunchKernelGGL((VoiceWaveKernel<18, 128, 0, false>), grid, block, 0, s, L, samplesToDo);
    }
    if(L.streams)
    {   // stream rows -> partial mix-line buses
        const uint32_t per = (L.numVoices + L.numLineGroups - 1u) / L.numLineGroups;
        const uint32_t maxRows = per * L.streamsPerVoice;
        const dim3 mgrid(kLine / 256, L.numLineGroups);
        const size_t lds = (size_t{maxRows} + size_t{maxRows} * L.lineStride) * sizeof(uint32_t);
        const dim3 rgrid(L.numLineGroups);
        if(L.lineStride <= 8)
        {
            hipLaunchKernelGGL(LinesMixKernel<8>, mgrid, dim3(256), lds, s, L, samplesToDo);
            hipLaunchKernelGGL(LinesRampKernel<8>, rgrid, dim3(64), 0, s, L, samplesToDo);
        }
        else if(L.lineStride <= 16)
        {
            hipLaunchKernelGGL(LinesMixKernel<16>, mgrid, dim3(256), lds, s, L, samplesToDo);
            hipLaunchKernelGGL(LinesRampKernel<16>, rgrid, dim3(64), 0, s, L, samplesToDo);
        }
        else
        {
            hipLaunchKernelGGL(LinesMixKernel<32>, mgrid, dim3(256), lds, s, L, samplesToDo);
            hipLaunchKernelGGL(LinesRampKernel<32>, rgrid, dim3(64), 0, s, L, samplesToDo);
        }
    }
    return hipGetLastError();
}

} // namespace oalgpu
