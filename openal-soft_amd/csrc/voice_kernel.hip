// The batched voice path: "for every Playing|Stopping voice: voice->mix(...)" for one update
// (ProcessContexts voice loop, alc/alu.cpp:2201-2206 -> Voice::mix, core/voice.cpp:988-1233),
// as three launches:
//
//   ApplyParamsKernel  parameter side: scatter the CalcVoiceParams results into the voice
//                      arrays, HrtfStore::getCoeffs HRIR blend (core/hrtf.cpp:192-260) and the
//                      BiquadInterpFilter::setParams state machine (biquad.cpp:131-149).
//   VoiceMixKernel     one 256-thread workgroup per group of voices.  Per voice: coalesced
//                      source-window load with format decode into LDS (LoadResampledSamples,
//                      voice.cpp:642-824), resample (4 outputs per thread), serial dual-biquad
//                      IIRs on spare lanes (DoFilters, voice.cpp:255-267), then either the
//                      register-tiled dual-ear FIR (DoHrtfMix, voice.cpp:827-902) or the
//                      gain-ramped bus mix (MixSamples).  Bus contributions stay in registers
//                      across the group's voices and leave as ONE partial per group.
//   BusReduceKernel    sums the per-group partials in group order into the bus block.
//
// Deterministic by construction: fixed voice->group mapping, fixed summation order, no atomics.
#include <hip/hip_ext.h>
#include "kernels.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kInLen = kHist + kLine;          // [History | samples], DoHrtfMix HrtfSamples
constexpr int kXPad = 128;                     // zero padding either side of the x' arrays
constexpr int kXLen = kXPad + kLine + kXPad;
constexpr int kXOldLen = kXPad + 256;         // old-filter fade: <= 64 inputs, frames < 256
constexpr int kTabFloats = 3328;              // fast bsinc up to m=48, full bsinc up to m=24, cubic

struct alignas(16) SharedMem {
    float rdata[kResampleDataSize + 8];        // DeviceBase::mResampleData
    float in[kInLen];                          // [hrtf history | resampled samples]
    float filt[kWaves][kInLen];                // filter outputs, one per concurrently running IIR
    float xl[kXLen], xr[kXLen];                // x'_ear[i] = In[64-delay+i]*g(i), zero padded
    float xol[kXOldLen], xor_[kXOldLen];       // old-filter fade-out inputs
    float tab[kTabFloats];                     // resampler coefficient rows (padded stride)
    float coT[kHrirLen * 2], coO[kHrirLen * 2];// this voice's Hrtf.Target / Hrtf.Old coefficients
    float gCur[7][32], gTgt[7][32];            // Gains.Current/Target snapshot: [direct | send i][line]
    int32_t sendSlot[8];
    int32_t best;
};

} // namespace

// ---------------------------------------------------------------------------------------------
// Parameter side
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) ApplyParamsKernel(DeviceLayout L, const ParamRecord *__restrict__ recs)
{
    ApplyRecordWave(L, recs[blockIdx.x], threadIdx.x);
}

// VoiceFlag::IsAmbisonic + the channel's splitter and scales (Voice::prepare, voice.cpp:1353-1380)
__global__ void SetAmbiScaleKernel(DeviceLayout L, uint32_t v, AmbiScaleState st)
{
    L.ambi[v] = st;
    L.ctl[v].flags |= kFlagAmbiScale;
}

// Voice::mStartTime ahead of the device clock by `samples` output samples (voice.cpp:1023-1046)
__global__ void SetStartDelayKernel(DeviceLayout L, uint32_t v, uint32_t samples)
{
    L.startDelay[v] = samples;
    if(samples) L.ctl[v].flags |= kFlagDelayed;
    else L.ctl[v].flags &= ~kFlagDelayed;
}

// A callback source's storage as the voice sees it this update (voice.cpp:726-752): mNumCallbackBlocks valid blocks,
// read from mCallbackBlockOffset -- to every voice kernel a static, non-looping buffer of that length and position
__global__ void SetVoiceWindowKernel(DeviceLayout L, uint32_t v, int32_t buffer, uint32_t sampleLen, int32_t position)
{
    const_cast<BufferItem*>(L.buffers)[buffer].sampleLen = sampleLen;      // (a callback buffer's descriptor is the one that is not immutable)
    L.ctl[v].buf.sampleLen = sampleLen;
    L.ctl[v].position = position;
}

void LaunchSetVoiceWindow(hipStream_t s, const DeviceLayout &L, uint32_t voice, int32_t buffer, uint32_t sampleLen, int32_t position)
{ hipLaunchKernelGGL(SetVoiceWindowKernel, dim3(1), dim3(1), 0, s, L, voice, buffer, sampleLen, position); }

// oalgpu_voice_events_async: what changed about the voices since the last report -- the play state (a source that ran out of
// buffer sets itself Stopping, core/voice.cpp:1201-1232), the buffer a streaming source has got to and the buffers it left
// behind (voice.cpp:1182-1218) -- appended to a slot of pinned host memory by the threads that find a difference; the
// workgroup that finishes last writes the count in front of them.  snapshot: [voice][3] what the last report saw.
__global__ void __launch_bounds__(256) VoiceEventsKernel(DeviceLayout L, uint32_t *snapshot, uint32_t *hostSlot, uint32_t capacity, uint32_t *counters)
{
    const uint32_t v = blockIdx.x * 256u + threadIdx.x;
    if(v < L.numVoices)
    {
        const VoiceCtl &c = L.ctl[v];
        const uint32_t st = uint32_t(c.playState), cb = uint32_t(c.curBuffer), qd = L.queueDone[v];
        uint32_t *snap = snapshot + size_t{v} * 3;
        if(snap[0] != st || snap[1] != cb || snap[2] != qd)
        {
            // (a voice the host has just started -- Playing, on a buffer, not Playing in the report before -- is no news to the host)
            const bool started = st == uint32_t(OALGPU_VOICE_PLAYING) && int32_t(cb) >= 0 && snap[0] != uint32_t(OALGPU_VOICE_PLAYING);
            // (nor is a voice that was Stopped in the last report and still is: the first report's snapshot differs from every slot
            // nobody ever initialised -- a context with more than the report's capacity of unused slots overflowed it)
            const bool idle = st == uint32_t(OALGPU_VOICE_STOPPED) && snap[0] == uint32_t(OALGPU_VOICE_STOPPED);
            snap[0] = st; snap[1] = cb; snap[2] = qd;
            const uint32_t idx = (started || idle) ? 0xffffffffu : atomicAdd(&counters[0], 1u);
            if(idx < capacity)
            {
                uint32_t *e = hostSlot + 4u + size_t{idx} * 8u;
                e[0] = v; e[1] = st; e[2] = cb; e[3] = qd; e[4] = uint32_t(c.position); e[5] = c.positionFrac;
                e[6] = (c.flags & kFlagFading) ? 1u : 0u; e[7] = 0u;
            }
        }
    }
    __syncthreads();
    if(threadIdx.x == 0)
    {
        __threadfence_system();
        const uint32_t done = atomicAdd(&counters[1], 1u);
        if(done + 1u == gridDim.x)
        {
            const uint32_t n = atomicExch(&counters[0], 0u);
            counters[1] = 0u;
            hostSlot[0] = n;
            __threadfence_system();
        }
    }
}

void LaunchVoiceEvents(hipStream_t s, const DeviceLayout &L, uint32_t *snapshot, uint32_t *hostSlot, uint32_t capacity, uint32_t *counters)
{
    hipLaunchKernelGGL(VoiceEventsKernel, dim3((L.numVoices + 255u) / 256u), dim3(256), 0, s, L, snapshot, hostSlot, capacity, counters);
}

void LaunchSetStartDelay(hipStream_t s, const DeviceLayout &L, uint32_t voice, uint32_t samples)
{ hipLaunchKernelGGL(SetStartDelayKernel, dim3(1), dim3(1), 0, s, L, voice, samples); }

void LaunchSetAmbiScale(hipStream_t s, const DeviceLayout &L, uint32_t voice, const AmbiScaleState &st)
{ hipLaunchKernelGGL(SetAmbiScaleKernel, dim3(1), dim3(1), 0, s, L, voice, st); }

// NFCtrlFilter.adjust(w0) + VoiceFlag::HasNfc: the coefficients (designed on the host); the
// delay elements are left as they are
__global__ void SetNfcKernel(DeviceLayout L, uint32_t v, NfcState st)
{
    const uint32_t t = threadIdx.x;
    float *dst = reinterpret_cast<float*>(&L.nfc[v]);
    const float *src = reinterpret_cast<const float*>(&st);
    if(t < 50) dst[t] = src[t];                          // a[5][5] | b[5][5]
    if(t == 0) L.ctl[v].flags |= kFlagNfc;
}

void LaunchSetNfc(hipStream_t s, const DeviceLayout &L, uint32_t voice, const NfcState &coeffs)
{ hipLaunchKernelGGL(SetNfcKernel, dim3(1), dim3(64), 0, s, L, voice, coeffs); }

// Voice::prepare (core/voice.cpp:1235-1397) + InitVoice's source attach (al/source.cpp:639-670)
// for `count` static mono voices: mixing state cleared, filters default-constructed
// (BiquadInterpFilter{}: identity coefficients, mCounter = -1), mStep = 0, Playing, not fading.
__global__ void __launch_bounds__(64) InitVoicesKernel(DeviceLayout L, const VoiceInitRecord *__restrict__ recs)
{
    const VoiceInitRecord r = recs[blockIdx.x];
    const uint32_t v = r.voice, t = threadIdx.x;
    BiquadState def{};
    def.b0 = 1.0f; def.tb0 = 1.0f; def.counter = -1;
    if(t == 0)
    {
        VoiceCtl c{};
        c.playState = OALGPU_VOICE_PLAYING;
        c.position = r.position;
        c.positionFrac = r.positionFrac;
        c.curBuffer = r.buffer;
        c.loopBuffer = r.looping ? r.buffer : -1;
        c.flags = (L.hrtf ? kFlagHasHrtf : 0u) | (r.queue ? kFlagQueue : 0u);
        L.queueDone[v] = 0u;
        for(int i = 0; i < 6; ++i) c.sendSlot[i] = -1;
        if(r.buffer >= 0) c.buf = L.buffers[r.buffer];
        L.ctl[v] = c;
        L.dfilt[size_t{v} * 2 + 0].f = def;
        L.dfilt[size_t{v} * 2 + 1].f = def;
    }
    if(t < kMaxPad) L.prev[size_t{v} * kMaxPad + t] = 0.0f;
    if(L.nfc && t < 20) reinterpret_cast<float*>(L.nfc[v].z)[t] = 0.0f;
    if(L.hrtf)
    {
        L.hist[size_t{v} * kHist + t] = 0.0f;
        for(uint32_t k = t; k < L.irStride * 2; k += 64)
        {
            L.hrtfOld[size_t{v} * L.irStride * 2 + k] = 0.0f;
            L.hrtfTgt[size_t{v} * L.irStride * 2 + k] = 0.0f;
        }
    }
    else if(t < L.numDry)
    {
        L.gainCur[size_t{v} * L.numDry + t] = 0.0f;
        L.gainTgt[size_t{v} * L.numDry + t] = 0.0f;
    }
    if(t < L.numSends * 2) L.sfilt[size_t{v} * L.numSends * 2 + t].f = def;
    for(uint32_t k = t; k < L.numSends * L.wetChannels; k += 64)
    {
        L.sendCur[size_t{v} * L.numSends * L.wetChannels + k] = 0.0f;
        L.sendTgt[size_t{v} * L.numSends * L.wetChannels + k] = 0.0f;
    }
}

void LaunchInitVoices(hipStream_t s, const DeviceLayout &L, const VoiceInitRecord *recs, uint32_t count)
{
    if(count) hipLaunchKernelGGL(InitVoicesKernel, dim3(count), dim3(64), 0, s, L, recs);
}

// the weighted sum of ApplyHrtfTargetWave (kernels.hpp) into a row of the parameter block instead of the voice's filter
__global__ void __launch_bounds__(64) BlendRowsKernel(DeviceLayout L, const ParamRecord *__restrict__ recs, float *__restrict__ rows)
{
    const ParamRecord &r = recs[blockIdx.x];
    if(r.keepHrtf) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t i0 = r.hrtfIdx[0], i1 = r.hrtfIdx[1], i2 = r.hrtfIdx[2], i3 = r.hrtfIdx[3];
    const float w0 = r.hrtfW[0], w1 = r.hrtfW[1], w2 = r.hrtfW[2], w3 = r.hrtfW[3];
    const uint32_t live = ((L.irSize + 1u) & ~1u) * 2u;
    for(uint32_t e = lane; e < L.irStride * 2; e += 64)
    {
        float x = (e < 2) ? r.hrtfPass : 0.0f;
        x = L.hrirs[size_t{i0} * (kHrirLen * 2) + e] * w0 + x;
        x = L.hrirs[size_t{i1} * (kHrirLen * 2) + e] * w1 + x;
        x = L.hrirs[size_t{i2} * (kHrirLen * 2) + e] * w2 + x;
        x = L.hrirs[size_t{i3} * (kHrirLen * 2) + e] * w3 + x;
        rows[size_t{blockIdx.x} * L.irStride * 2 + e] = (e < live) ? x : 0.0f;
    }
}
void LaunchBlendRows(hipStream_t s, const DeviceLayout &L, const ParamRecord *recs, uint32_t count, float *rows)
{
    if(count) hipLaunchKernelGGL(BlendRowsKernel, dim3(count), dim3(64), 0, s, L, recs, rows);
}

void LaunchApplyParams(hipStream_t s, const DeviceLayout &L, const ParamRecord *recs, uint32_t count)
{
    if(count) hipLaunchKernelGGL(ApplyParamsKernel, dim3(count), dim3(64), 0, s, L, recs);
}

// A moved voice of an HRTF context: new target filter, delays and gain; the filter is marked as replaced.  The record is the
// host's 24 bytes as they are (oalgpu_voice_move, read straight out of a pinned ring slot): BOTH halves of HrtfStore::getCoeffs
// (core/hrtf.cpp:192-260) run here -- the index half (HrtfBlendFor: the same float operations as on the host, every lane the
// same values) and the weighted sum of the four HRIRs.  The calling thread, which is what bounds the pipelined boundary,
// used to evaluate the index half for every moved voice (11 us per 1024 records).
struct RawMove { uint32_t voice; float ev, az, dist, spread, gain; };
static_assert(sizeof(RawMove) == sizeof(oalgpu_voice_move), "RawMove mirrors oalgpu_voice_move");
__global__ void __launch_bounds__(64) ApplyMovesKernel(DeviceLayout L, HrtfStoreDev st, const RawMove *__restrict__ moves)
{
    const RawMove m = moves[blockIdx.x];
    const uint32_t v = m.voice, lane = threadIdx.x;
    const HrirBlend b = HrtfBlendFor(st, m.ev, m.az, m.dist, m.spread);
    if(lane == 0)
    {
        VoiceCtl &ctl = L.ctl[v];
        ctl.flags |= kFlagHasHrtf | kFlagHrtfDirty;
        ctl.hrtfTgtDelay[0] = b.delay[0]; ctl.hrtfTgtDelay[1] = b.delay[1];
        ctl.hrtfTgtGain = m.gain;
    }
    ApplyHrtfTargetWave(L, v, b.idx, b.w, b.passthru, lane);
}

// evDone: an event bound to the dispatch's completion (null: none); moves may be pinned host memory
void LaunchApplyMoves(hipStream_t s, const DeviceLayout &L, const HrtfStoreDev &st, const void *moves, uint32_t count, hipEvent_t evDone)
{
    if(count) hipExtLaunchKernelGGL(ApplyMovesKernel, dim3(count), dim3(64), 0, s, nullptr, evDone, 0u, L, st, static_cast<const RawMove*>(moves));
}

// A voice's HRTF target as the reference's own parameter stage computed it (Hrtf.Target after CalcHrtfPanning,
// alc/alu.cpp:1214-1216 / :1256-1258 / :1296-1298): coefficients copied, not blended here; the filter is marked as replaced.
__global__ void __launch_bounds__(64) ApplyTargetsKernel(DeviceLayout L, const TargetRecord *__restrict__ recs,
    const float *__restrict__ coeffs)
{
    const TargetRecord &r = recs[blockIdx.x];
    const uint32_t v = r.voice, lane = threadIdx.x;
    if(lane == 0)
    {
        VoiceCtl &ctl = L.ctl[v];
        ctl.flags |= kFlagHasHrtf | kFlagHrtfDirty;
        ctl.hrtfTgtDelay[0] = r.delay[0]; ctl.hrtfTgtDelay[1] = r.delay[1];
        ctl.hrtfTgtGain = r.gain;
    }
    // (the mixers apply IrSize taps rounded up to even, mixer_sse.cpp:46-51: what lies beyond must not reach the
    // fixed-length FIRs -- see ApplyHrtfTargetWave)
    const uint32_t live = ((L.irSize + 1u) & ~1u) * 2u;
    const float *src = coeffs + size_t{blockIdx.x} * (kHrirLen * 2);
    for(uint32_t e = lane; e < L.irStride * 2; e += 64)
        L.hrtfTgt[size_t{v} * L.irStride * 2 + e] = (e < live) ? src[e] : 0.0f;
}

void LaunchApplyTargets(hipStream_t s, const DeviceLayout &L, const TargetRecord *recs, const float *coeffs, uint32_t count)
{
    if(count) hipLaunchKernelGGL(ApplyTargetsKernel, dim3(count), dim3(64), 0, s, L, recs, coeffs);
}

// ---------------------------------------------------------------------------------------------
// Voice kernel
// ---------------------------------------------------------------------------------------------
namespace {

// ---- resampler coefficient rows staged in LDS ------------------------------------------------
// bsinc: phase row pi = [fil(m) | phd(m)] at pi*(2m+4); for the scale-interpolating kernel the
// [scd(m) | spd(m)] rows follow 32 rows later.  The +4 keeps rows 16-byte aligned for float4
// reads and spreads consecutive rows over different LDS bank groups.  cubic: 32 rows of 12.
__device__ __forceinline__ bool TableFitsLds(int kind, uint32_t m)
{
    if(kind == 2) return true;
    if(kind == 3) return 32u * (2u * m + 4u) <= uint32_t(kTabFloats);
    if(kind == 4) return 64u * (2u * m + 4u) <= uint32_t(kTabFloats);
    return false;
}

__device__ __forceinline__ TabLayout LdsTabLayout(uint32_t m) { return TabLayout{2u * m + 4u, 32u * (2u * m + 4u), 12u}; }

__device__ __forceinline__ void StageTable(SharedMem &sm, const float *__restrict__ filter, int kind, uint32_t m)
{
    const uint32_t t = threadIdx.x;
    if(kind == 2)
    {
        for(uint32_t k = t; k < 256u; k += kThreads) sm.tab[(k >> 3) * 12u + (k & 7u)] = filter[k];
    }
    else
    {
        const uint32_t row = 2u * m, stride = row + 4u;
        const uint32_t blocks = kind == 4 ? 2u : 1u;
        for(uint32_t b = 0; b < blocks; ++b)
            for(uint32_t k = t; k < 32u * row; k += kThreads)
                sm.tab[b * 32u * stride + (k / row) * stride + (k % row)] = filter[b * 32u * row + k];
    }
}

// LoadResampledSamples, core/voice.cpp:642-824, for one real channel of a static voice.
// Produces samplesToLoad resampled samples at sm.in[kHist..]; updates prev[v] when Playing.
template<bool EXACT>
__device__ __forceinline__ void LoadResampled(SharedMem &sm, const DeviceLayout &L, uint32_t v, const VoiceCtl &ctl,
    bool playing, int32_t intPos, uint32_t fracPos, uint32_t increment, uint32_t samplesToLoad,
    uint32_t samplesToMix, int32_t bufferItem, bool looping, uint32_t &stagedTable, uint32_t mixOffset = 0)
{
    const uint32_t t = threadIdx.x;
    float *rdata = sm.rdata;
    float *srcBuffer = rdata + kMaxEdge;
    float *mixing = sm.in + kHist + mixOffset;       // mixOffset: a delayed start's output position (voice.cpp:1023-1046)
    if(t < kMaxPad) rdata[t] = L.prev[size_t{v} * kMaxPad + t];
    const float *filter = L.tables + ctl.rsFilterOffset;
    const int kind = ctl.rsKind;
    const uint32_t rsM = ctl.rsM, rsL = ctl.rsL;
    const float rsSf = ctl.rsSf;
    const bool resamples = !(increment == kFracOne && fracPos == 0) || samplesToLoad > 1200u;
    const bool useLds = kind >= 2 && TableFitsLds(kind, rsM);
    // the staged slice is identified by (offset, kind); most voices of a group share it
    const uint32_t tableKey = ctl.rsFilterOffset * 8u + uint32_t(kind);
    if(useLds && resamples && stagedTable != tableKey)
    {
        __syncthreads();                 // previous voice may still be reading sm.tab
        StageTable(sm, filter, kind, rsM);
        stagedTable = tableKey;
    }
    __syncthreads();

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
                for(uint32_t k = t; k < bdst; k += kThreads) mixing[loaded + k] = 0.0f;
                for(uint32_t k = t; k < bsrc; k += kThreads) srcBuffer[k] = 0.0f;
                silent = true;
            }
            else
                for(uint32_t k = t; k < srcDelay; k += kThreads) srcBuffer[k] = 0.0f;
        }
        if(silent)
        {
            __syncthreads();
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
            if(t == 0)
            {
                uint32_t best = 0;
                for(uint32_t i = 1; i < avail; ++i)
                    if(fabsf(srcBuffer[i]) < fabsf(srcBuffer[best])) best = i;
                sm.best = int32_t(best);
            }
            __syncthreads();
            const uint32_t best = uint32_t(sm.best);
            const float hold = srcBuffer[best];
            __syncthreads();
            for(uint32_t k = best + 1 + t; k < tofill; k += kThreads) srcBuffer[k] = hold;
        }
        else
        {
            const uint32_t upos = intPos < 0 ? 0u : uint32_t(intPos);
            if(ctl.flags & kFlagQueue)
                FillFromQueue<kThreads>(srcBuffer + srcDelay, bsrc - srcDelay, L.buffers, bufferItem, ctl.loopBuffer, upos, t,
                    [] { __syncthreads(); });
            else
                FillFromBuffer<kThreads>(srcBuffer + srcDelay, bsrc - srcDelay, L.buffers[bufferItem], looping, upos, t);
        }
        __syncthreads();

        // voice.cpp:764-769
        if(increment == kFracOne && fracPos == 0)
        {
            for(uint32_t k = t; k < bdst; k += kThreads) mixing[loaded + k] = srcBuffer[k];
        }
        else if(useLds)
        {
            const TabLayout lay = LdsTabLayout(rsM);
            for(uint32_t k = t; k < bdst; k += kThreads)
                mixing[loaded + k] = ResampleAt<EXACT, true>(kind, rsM, rsL, rsSf, sm.tab, lay, rdata, fracPos, increment, k, bdst);
        }
        else
        {
            const TabLayout lay = ReferenceTabLayout(rsM);
            for(uint32_t k = t; k < bdst; k += kThreads)
                mixing[loaded + k] = ResampleAt<EXACT, false>(kind, rsM, rsL, rsSf, filter, lay, rdata, fracPos, increment, k, bdst);
        }

        // voice.cpp:772-785: history for the next update, taken at the end-of-mix position
        if(playing)
        {
            const uint32_t loadEnd = loaded + bdst;
            if(samplesToMix > loaded && samplesToMix <= loadEnd)
            {
                const uint32_t dstOffset = samplesToMix - loaded;
                const uint32_t srcOffset = uint32_t((uint64_t{dstOffset} * increment + fracPos) >> kFracBits);
                if(t < kMaxPad) L.prev[size_t{v} * kMaxPad + t] = rdata[srcOffset + t];
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
            __syncthreads();
            float carry = 0.0f;
            if(t < kMaxPad) carry = rdata[srcOffset + t];
            __syncthreads();
            if(t < kMaxPad) rdata[t] = carry;
        }
        __syncthreads();
    }
}

// ---- wave-parallel dual biquad (FAST, time-invariant coefficients) ---------------------------
// The cascade is a linear system with state s = (z01, z02, z11, z12):  s' = A s + B x.
// Lane L owns samples [L*seg, (L+1)*seg).  (1) every lane measures M = A^seg by running seg
// zero-input steps from the four unit states; (2) runs its samples from a zero state to get the
// forced response q_L; (3) the block-start states follow from the 64-step recurrence
// S_{L+1} = M S_L + q_L; (4) each lane re-runs the true recurrence from S_L over its samples.
// ~6*seg filter steps per lane instead of 1024 serial ones.  Rounding differs from the serial
// loop only through the propagated start states (tolerance class of FAST mode).
struct Dual4 { float a, b, c, d; };

__device__ __forceinline__ float DualStep(Dual4 &s, float x, const BiquadState &f0, const BiquadState &f1)
{
    const float y0 = __builtin_fmaf(x, f0.b0, s.a);
    s.a = __builtin_fmaf(x, f0.b1, __builtin_fmaf(-y0, f0.a1, s.b));
    s.b = __builtin_fmaf(x, f0.b2, -y0 * f0.a2);
    const float y1 = __builtin_fmaf(y0, f1.b0, s.c);
    s.c = __builtin_fmaf(y0, f1.b1, __builtin_fmaf(-y1, f1.a1, s.d));
    s.d = __builtin_fmaf(y0, f1.b2, -y1 * f1.a2);
    return y1;
}

__device__ __forceinline__ void BiquadDualWave(BiquadState &f0, BiquadState &f1, const float *src, float *dst,
    uint32_t n, uint32_t lane)
{
    const uint32_t seg = (n + 63u) / 64u;
    const uint32_t begin = lane * seg;
    const uint32_t end = (begin + seg < n) ? begin + seg : n;
    // (1) M = A^seg, column by column
    Dual4 m0{1, 0, 0, 0}, m1{0, 1, 0, 0}, m2{0, 0, 1, 0}, m3{0, 0, 0, 1};
    for(uint32_t i = 0; i < seg; ++i)
    {
        DualStep(m0, 0.0f, f0, f1); DualStep(m1, 0.0f, f0, f1);
        DualStep(m2, 0.0f, f0, f1); DualStep(m3, 0.0f, f0, f1);
    }
    // (2) forced response of this lane's block
    Dual4 q{0, 0, 0, 0};
    for(uint32_t i = begin; i < end; ++i) DualStep(q, src[i], f0, f1);
    // (3) block-start states
    Dual4 cur{f0.z1, f0.z2, f1.z1, f1.z2}, start = cur;
    for(uint32_t L = 0; L < 64u; ++L)
    {
        if(lane == L) start = cur;
        const float qa = __shfl(q.a, int(L)), qb = __shfl(q.b, int(L)), qc = __shfl(q.c, int(L)), qd = __shfl(q.d, int(L));
        Dual4 nx;
        nx.a = m0.a * cur.a + m1.a * cur.b + m2.a * cur.c + m3.a * cur.d + qa;
        nx.b = m0.b * cur.a + m1.b * cur.b + m2.b * cur.c + m3.b * cur.d + qb;
        nx.c = m0.c * cur.a + m1.c * cur.b + m2.c * cur.c + m3.c * cur.d + qc;
        nx.d = m0.d * cur.a + m1.d * cur.b + m2.d * cur.c + m3.d * cur.d + qd;
        cur = nx;
    }
    // (4) the real recurrence from the block-start state
    for(uint32_t i = begin; i < end; ++i) dst[i] = DualStep(start, src[i], f0, f1);
    // final state = state after the last sample, held by the lane that owns it
    const int lastLane = int((n - 1u) / seg);
    f0.z1 = __shfl(start.a, lastLane); f0.z2 = __shfl(start.b, lastLane);
    f1.z1 = __shfl(start.c, lastLane); f1.z2 = __shfl(start.d, lastLane);
}

// Register-tiled dual-ear FIR (FAST): thread t owns output frames 4t..4t+3.  Per 8 taps it
// reads x'[n0-j0-8 .. n0-j0+3] per ear as three aligned ds_read_b128 and the 8x2 coefficients
// as four broadcast ds_read_b128, against 64 FMAs.
__device__ __forceinline__ void FirMain(float (&accL)[4], float (&accR)[4], const float *xl, const float *xr,
    const float *coeffs, uint32_t irStride, uint32_t n0)
{
    for(uint32_t j0 = 0; j0 < irStride; j0 += 8)
    {
        const float4 *pl = reinterpret_cast<const float4*>(xl + kXPad + n0 - j0 - 8);
        const float4 *pr = reinterpret_cast<const float4*>(xr + kXPad + n0 - j0 - 8);
        const float4 *pc = reinterpret_cast<const float4*>(coeffs + j0 * 2);
        const float4 l0 = pl[0], l1 = pl[1], l2 = pl[2];
        const float4 r0 = pr[0], r1 = pr[1], r2 = pr[2];
        const float4 c0 = pc[0], c1 = pc[1], c2 = pc[2], c3 = pc[3];
        const float wl[12] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w, l2.x, l2.y, l2.z, l2.w};
        const float wr[12] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        const float cf[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w};
#pragma unroll
        for(int jj = 0; jj < 8; ++jj)
        {
#pragma unroll
            for(int r = 0; r < 4; ++r)
            {
                accL[r] = __builtin_fmaf(cf[jj * 2 + 0], wl[8 + r - jj], accL[r]);
                accR[r] = __builtin_fmaf(cf[jj * 2 + 1], wr[8 + r - jj], accR[r]);
            }
        }
    }
}

// one output frame, one ear, all taps (the 128 tail frames and the old-filter fade)
__device__ __forceinline__ float FirOne(float acc, const float *x /* x'[frame] */, const float *coeffs,
    uint32_t irStride, uint32_t ear)
{
    for(uint32_t j = 0; j < irStride; ++j)
        acc = __builtin_fmaf(coeffs[j * 2 + ear], x[-int32_t(j)], acc);
    return acc;
}

// NfcFilterN::process, core/filters/nfc.cpp:222-288: the filter of order o over src[0..n) -> dst, on one lane
// in the reference's operation order (the wavefront kernel runs the sections as block scans, wave_common.hpp)
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

template<bool EXACT, int LINES>
__global__ void __launch_bounds__(kThreads) VoiceMixKernel(DeviceLayout L, uint32_t samplesToDo, uint32_t carryAccum)
{
    __shared__ SharedMem sm;
    const uint32_t t = threadIdx.x;
    const uint32_t wave = t >> 6, lane = t & 63;
    const uint32_t group = blockIdx.x;
    const uint32_t numSends = L.numSends, wetCh = L.wetChannels, numDry = L.numDry;
    const uint32_t irStride = L.irStride;
    const uint32_t irEff = EXACT ? ((L.irSize + 1u) & ~1u) : L.irSize;

    // bus accumulators kept in registers across this group's voices
    float lineAcc[LINES][4];
#pragma unroll
    for(int c = 0; c < LINES; ++c)
#pragma unroll
        for(int k = 0; k < 4; ++k) lineAcc[c][k] = 0.0f;
    // FAST: frames 4t..4t+3 (main), 1024+(t&127) ear t>>7 (tail), (t&127)[+128] ear t>>7 (old fade)
    float hL[4] = {0, 0, 0, 0}, hR[4] = {0, 0, 0, 0};
    float hTail = 0.0f, hOld0 = 0.0f, hOld1 = 0.0f;
    // EXACT: frames t + 256k, k < 5, both ears, one ordered chain per frame
    float eL[5] = {0, 0, 0, 0, 0}, eR[5] = {0, 0, 0, 0, 0};

    if(L.hrtf && carryAccum && group == 0)
    {   // group 0 continues the running accumulator (HrtfAccumData tail carried by MixDirectHrtf)
        const float *acc = L.bus + BusAccumOffset(L);
        if constexpr(EXACT)
        {
#pragma unroll
            for(int k = 0; k < 5; ++k)
            {
                const uint32_t o = t + 256u * k;
                if(o < kLine + kHrirLen) { eL[k] = acc[o * 2]; eR[k] = acc[o * 2 + 1]; }
            }
        }
        else
        {
#pragma unroll
            for(int r = 0; r < 4; ++r) { hL[r] = acc[(4 * t + r) * 2]; hR[r] = acc[(4 * t + r) * 2 + 1]; }
            hTail = acc[(kLine + (t & 127)) * 2 + (t >> 7)];
        }
    }

    // zero padding of the x' arrays (never written below)
    for(uint32_t k = t; k < kXLen; k += kThreads) { sm.xl[k] = 0.0f; sm.xr[k] = 0.0f; }
    for(uint32_t k = t; k < kXOldLen; k += kThreads) { sm.xol[k] = 0.0f; sm.xor_[k] = 0.0f; }
    __syncthreads();

    uint32_t stagedTable = 0xffffffffu;
    const uint32_t vBegin = group * L.voicesPerGroup;
    const uint32_t vEnd = (vBegin + L.voicesPerGroup < L.numVoices) ? vBegin + L.voicesPerGroup : L.numVoices;
    for(uint32_t v = vBegin; v < vEnd; ++v)
    {
        const VoiceCtl ctl = L.ctl[v];
        const int vstate = ctl.playState;
        if(vstate != OALGPU_VOICE_PLAYING && vstate != OALGPU_VOICE_STOPPING) continue;
        const bool playing = vstate == OALGPU_VOICE_PLAYING;
        const uint32_t increment = ctl.step;
        if(increment < 1)
        {   // voice.cpp:1002-1010
            if(!playing && t == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
            continue;
        }
        int32_t bufPosInt = ctl.position;
        uint32_t bufPosFrac = ctl.positionFrac;
        int32_t bufferItem = ctl.curBuffer;
        int32_t loopItem = ctl.loopBuffer;
        if(loopItem >= 0 && bufferItem >= 0 && !(ctl.flags & kFlagQueue))
        {   // voice.cpp:1015-1019 (static voices)
            if(bufPosInt >= 0 && uint32_t(bufPosInt) >= L.buffers[bufferItem].loopEnd) loopItem = -1;
        }
        // ---- delayed start, voice.cpp:1023-1046: the voice starts `outPos` samples into this update (or not
        // yet).  It is on its first mix then (no fade, empty history, cleared filters), so the samplesToMix
        // samples are produced at offset outPos of the voice's sample line behind zeros, and everything
        // downstream mixes a full line: what lands in front of outPos is exactly zero.
        uint32_t outPos = 0;
        if(ctl.flags & kFlagDelayed)
        {
            if(t == 0) sm.best = int32_t(L.startDelay[v]);
            __syncthreads();
            const uint32_t d = uint32_t(sm.best);
            __syncthreads();
            if(!playing)
            {   // "supposed to be stopping but hasn't actually started yet"
                if(t == 0) { L.ctl[v].playState = OALGPU_VOICE_STOPPED; L.ctl[v].flags = ctl.flags & ~kFlagDelayed; L.startDelay[v] = 0u; }
                continue;
            }
            if(d >= samplesToDo) { if(t == 0) L.startDelay[v] = d - samplesToDo; continue; }
            outPos = d;
            if(t == 0) L.startDelay[v] = 0u;
        }
        const uint32_t samplesToMix = samplesToDo - outPos;
        const uint32_t N = samplesToDo;                     // the line downstream stages see
        for(uint32_t k = t; k < outPos; k += kThreads) sm.in[kHist + k] = 0.0f;
        const bool hasHrtf = (ctl.flags & kFlagHasHrtf) != 0;

        // snapshot of this voice's gain pairs, send slots and HRIRs (read by every thread below
        // while single threads write the updated state back to HBM)
        if(t < 32)
        {
            if(!L.hrtf && t < numDry)
            {
                sm.gCur[0][t] = L.gainCur[size_t{v} * numDry + t];
                sm.gTgt[0][t] = L.gainTgt[size_t{v} * numDry + t];
            }
            if(t < 6) sm.sendSlot[t] = L.ctl[v].sendSlot[t];
        }
        for(uint32_t k = t; k < numSends * wetCh; k += kThreads)
        {
            sm.gCur[1 + k / wetCh][k % wetCh] = L.sendCur[size_t{v} * numSends * wetCh + k];
            sm.gTgt[1 + k / wetCh][k % wetCh] = L.sendTgt[size_t{v} * numSends * wetCh + k];
        }
        if(hasHrtf && t < irStride * 2)
        {
            sm.coT[t] = L.hrtfTgt[size_t{v} * irStride * 2 + t];
            sm.coO[t] = L.hrtfOld[size_t{v} * irStride * 2 + t];
        }

        LoadResampled<EXACT>(sm, L, v, ctl, playing, bufPosInt, bufPosFrac, increment, samplesToMix, samplesToMix, bufferItem,
            loopItem >= 0, stagedTable, outPos);

        if(ctl.flags & kFlagAmbiScale)
        {   // ---- VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091
            if(t == 0)
            {
                AmbiScaleState &a = L.ambi[v];
                SplitterState sp{a.coeff, a.lpZ1, a.lpZ2, a.apZ1};
                SplitterScale(sp, sm.in + kHist + outPos, samplesToMix, a.hfScale, a.lfScale);
                a.lpZ1 = sp.lpZ1; a.lpZ2 = sp.lpZ2; a.apZ1 = sp.apZ1;
            }
            __syncthreads();
        }

        const uint32_t counter = (ctl.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093

        // ---- DoFilters for the direct path and every send (voice.cpp:255-267,945-946,971-973).
        // Up to four filter targets at a time, one per wave, each into its own LDS line.
        // FAST + settled coefficients: the wave-parallel block method; otherwise the serial
        // recurrence on lane 0.  Inactive filters are cleared.  target 0 = direct, 1+i = send i.
        const uint32_t numTargets = 1 + numSends;
        for(uint32_t base = 0; base < numTargets; base += kWaves)
        {
            const uint32_t tg = base + wave;
            if(tg < numTargets)
            {
                const bool isDirect = tg == 0;
                const uint32_t si = tg - 1;
                const bool used = isDirect || sm.sendSlot[isDirect ? 0 : si] >= 0;
                const bool active = isDirect ? (ctl.flags & kFlagDirectFilter) != 0
                    : (ctl.flags >> (kFlagSendFilterShift + si)) & 1u;
                BiquadSlot *slots = isDirect ? &L.dfilt[size_t{v} * 2] : &L.sfilt[(size_t{v} * numSends + si) * 2];
                if(used)
                {
                    BiquadState f0 = slots[0].f, f1 = slots[1].f;
                    const bool settled = f0.counter <= 0 && f1.counter <= 0;
                    if(active) for(uint32_t k = lane; k < outPos; k += 64) sm.filt[wave][kHist + k] = 0.0f;
                    if(active && !EXACT && settled)
                    {
                        BiquadDualWave(f0, f1, sm.in + kHist + outPos, sm.filt[wave] + kHist + outPos, samplesToMix, lane);
                        if(lane == 0) { slots[0].f = f0; slots[1].f = f1; }
                    }
                    else if(lane == 0)
                    {
                        if(active) BiquadDualInterp(f0, f1, sm.in + kHist + outPos, sm.filt[wave] + kHist + outPos, samplesToMix);
                        else { BiquadClear(f0); BiquadClear(f1); }
                        slots[0].f = f0; slots[1].f = f1;
                    }
                }
            }
            __syncthreads();

            for(uint32_t w = 0; w < kWaves; ++w)
            {
                const uint32_t tg2 = base + w;
                if(tg2 >= numTargets) break;
                const bool isDirect = tg2 == 0;
                const uint32_t si = tg2 - 1;
                const bool active = isDirect ? (ctl.flags & kFlagDirectFilter) != 0
                    : (ctl.flags >> (kFlagSendFilterShift + si)) & 1u;
                float *inbuf = active ? sm.filt[w] : sm.in;         // [64 history slots | samples]
                const float *samples = inbuf + kHist;

                if(isDirect && hasHrtf)
                {
                    // ---------------- DoHrtfMix, voice.cpp:827-902 ----------------
                    if(t < kHist) inbuf[t] = L.hist[size_t{v} * kHist + t];
                    __syncthreads();
                    if(playing && t < kHist) L.hist[size_t{v} * kHist + t] = inbuf[N + t];

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
                    const float *oldCo = sm.coO;
                    const float *tgtCo = sm.coT;

                    if constexpr(EXACT)
                    {
#pragma unroll
                        for(int k = 0; k < 5; ++k)
                        {
                            const uint32_t o = t + 256u * k;
                            if(o >= kLine + kHrirLen) continue;
                            if(oldOn)
                                HrtfGatherFrame<true, kGainOldFade>(eL[k], eR[k], inbuf, oldCo, irEff, odL, odR, 0.0f,
                                    oldGain / float(fademix), fademix, o);
                            if(newOn)
                                HrtfGatherFrame<true, kGainNewFade>(eL[k], eR[k], inbuf, tgtCo, irEff, dL, dR, 0.0f, newStep,
                                    fademix, o);
                            if(todo && o >= fademix)
                                HrtfGatherFrame<true, kGainRamp>(eL[k], eR[k], inbuf + fademix, tgtCo, irEff, dL, dR,
                                    gainAfterBlend, mainStep, todo, o - fademix);
                        }
                    }
                    else
                    {
                        // x'[i] for the target filter: fade-in ramp for i < fademix, then the gain ramp
                        for(uint32_t i = t; i < N; i += kThreads)
                        {
                            float g;
                            if(i < fademix) g = newOn ? newStep * float(i) : 0.0f;
                            else g = gainAfterBlend + mainStep * float(i - fademix);
                            sm.xl[kXPad + i] = inbuf[kHist - dL + i] * g;
                            sm.xr[kXPad + i] = inbuf[kHist - dR + i] * g;
                        }
                        if(t < 64)
                        {
                            float lv = 0.0f, rv = 0.0f;
                            if(oldOn && t < fademix)
                            {
                                const float g = (oldGain / float(fademix)) * float(fademix - t);
                                lv = inbuf[kHist - odL + t] * g;
                                rv = inbuf[kHist - odR + t] * g;
                            }
                            sm.xol[kXPad + t] = lv;
                            sm.xor_[kXPad + t] = rv;
                        }
                        __syncthreads();
                        FirMain(hL, hR, sm.xl, sm.xr, tgtCo, irStride, 4 * t);
                        {
                            const uint32_t ear = t >> 7, fr = t & 127;
                            const float *xe = ear ? sm.xr : sm.xl;
                            hTail = FirOne(hTail, xe + kXPad + kLine + fr, tgtCo, irStride, ear);
                            if(oldOn)
                            {
                                const float *xo = ear ? sm.xor_ : sm.xol;
                                hOld0 = FirOne(hOld0, xo + kXPad + fr, oldCo, irStride, ear);
                                if(irStride > 64) hOld1 = FirOne(hOld1, xo + kXPad + 128 + fr, oldCo, irStride, ear);
                            }
                        }
                    }
                    __syncthreads();
                    // voice.cpp:1094-1101 / :869-873,900: Old <- Target, Old.Gain <- reached gain
                    if((counter == 0 || fademix) && t < irStride * 2)
                        L.hrtfOld[size_t{v} * irStride * 2 + t] = tgtCo[t];
                    if(t == 0)
                    {
                        VoiceCtl &c = L.ctl[v];
                        if(counter == 0 || fademix) { c.hrtfOldDelay[0] = dL; c.hrtfOldDelay[1] = dR; }
                        c.hrtfOldGain = todo ? endGain : gainAfterBlend;
                    }
                    continue;
                }

                // ---------------- MixSamples (voice.cpp:962-963, 978-979) ----------------
                // `src` onto the bus lines [lineBase, lineBase + nlines) with the gains [gainOff + rel] of the
                // target's Current/Target snapshot
                const float *curSnap = sm.gCur[tg2], *tgtSnap = sm.gTgt[tg2];
                auto mixOnto = [&](const float *src, uint32_t lineBase, uint32_t nlines, uint32_t gainOff, float *cur)
                {
                    float s4[4];
#pragma unroll
                    for(int k = 0; k < 4; ++k) s4[k] = (t + 256u * k < N) ? src[t + 256u * k] : 0.0f;
#pragma unroll
                    for(int c = 0; c < LINES; ++c)
                    {
                        const uint32_t rel = uint32_t(c) - lineBase;
                        if(uint32_t(c) < lineBase || rel >= nlines) continue;
                        const float tg = playing ? tgtSnap[gainOff + rel] : 0.0f;      // SilentCoeffs when Stopping
                        const float cu = counter ? curSnap[gainOff + rel] : tg;        // voice.cpp:1094-1112
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
#pragma unroll
                        for(int k = 0; k < 4; ++k)
                        {
                            const uint32_t p = t + 256u * k;
                            if(p < N && MixLineActive(g, p)) lineAcc[c][k] = lineAcc[c][k] + MixLineValue(g, s4[k], p);
                        }
                        if(t == 0) cur[gainOff + rel] = g.newCur;
                    }
                };
                if(isDirect)
                {
                    float *cur = L.gainCur + size_t{v} * numDry;
                    const bool nfcV = L.nfc && (ctl.flags & kFlagNfc);
                    // DoNfcMix (voice.cpp:904-932): only the W line is mixed from the voice's own samples ...
                    mixOnto(samples, 0u, nfcV ? 1u : numDry, 0u, cur);
                    if(nfcV)
                    {   // ... every ambisonic order above 0 from the samples through that order's NFC section
                        // (NfcFilterN::process, one lane, in the reference's operation order)
                        uint32_t line = 1;
                        for(uint32_t o = 1; o <= L.nfcOrders; ++o)
                        {
                            const uint32_t cnt = L.chansPerOrder[o];
                            __syncthreads();
                            if(t == 0) NfcSerial(L.nfc[v], o, samples, sm.xl, N);
                            __syncthreads();
                            mixOnto(sm.xl, line, cnt, line, cur);
                            line += cnt;
                        }
                    }
                }
                else
                {
                    const int32_t slot = sm.sendSlot[si];
                    if(slot < 0) continue;
                    mixOnto(samples, (L.hrtf ? 0u : numDry) + uint32_t(slot) * wetCh, wetCh, 0u,
                        L.sendCur + (size_t{v} * numSends + si) * wetCh);
                }
            }
            __syncthreads();
        }

        // ---- voice.cpp:1116-1232: flags, position, loop wrap / end of buffer ----
        if(t == 0)
        {
            VoiceCtl &c = L.ctl[v];
            c.flags = (ctl.flags | kFlagFading) & ~(((counter == 0 || N) ? kFlagHrtfDirty : 0u) | kFlagDelayed);
            if(!playing) c.playState = OALGPU_VOICE_STOPPED;
            else
            {
                bufPosFrac += increment * samplesToMix;
                const uint32_t samplesDone = bufPosFrac >> kFracBits;
                bufPosInt = AddSat(bufPosInt, int32_t(samplesDone));
                bufPosFrac &= kFracMask;
                if(bufferItem >= 0 && bufPosInt > 0 && (ctl.flags & kFlagQueue))
                {   // a streaming source: buffers the position ran past are done (voice.cpp:1182-1194)
                    uint32_t buffersDone = 0;
                    const int32_t before = bufferItem;
                    AdvanceQueue(L.buffers, bufferItem, ctl.loopBuffer, bufPosInt, buffersDone);
                    if(buffersDone) L.queueDone[v] += buffersDone;
                    if(bufferItem >= 0 && bufferItem != before) c.buf = L.buffers[bufferItem];
                }
                else if(bufferItem >= 0 && bufPosInt > 0)
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
        __syncthreads();
    }

    // ---- one partial per group ----
    float *pl = L.partLines + size_t{group} * L.mixLines * kLine;
#pragma unroll
    for(int c = 0; c < LINES; ++c)
    {
        if(uint32_t(c) >= L.mixLines) continue;
#pragma unroll
        for(int k = 0; k < 4; ++k) pl[size_t{uint32_t(c)} * kLine + t + 256u * k] = lineAcc[c][k];
    }
    if(L.hrtf)
    {
        float *ph = L.partHrtf + size_t{group} * (kLine + kHrirLen) * 2;
        if constexpr(EXACT)
        {
#pragma unroll
            for(int k = 0; k < 5; ++k)
            {
                const uint32_t o = t + 256u * k;
                if(o < kLine + kHrirLen) { ph[o * 2] = eL[k]; ph[o * 2 + 1] = eR[k]; }
            }
        }
        else
        {
            float *stage = &sm.filt[0][0];          // the filter lines are dead by now
            __syncthreads();
#pragma unroll
            for(int r = 0; r < 4; ++r) { stage[(4 * t + r) * 2] = hL[r]; stage[(4 * t + r) * 2 + 1] = hR[r]; }
            stage[(kLine + (t & 127)) * 2 + (t >> 7)] = hTail;
            __syncthreads();
            stage[(t & 127) * 2 + (t >> 7)] += hOld0;
            stage[(128 + (t & 127)) * 2 + (t >> 7)] += hOld1;
            __syncthreads();
            for(uint32_t k = t; k < (kLine + kHrirLen) * 2; k += kThreads) ph[k] = stage[k];
        }
    }
}
template<bool EXACT, int LINES>
hipError_t LaunchVoiceMixT(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, bool carry)
{
    hipLaunchKernelGGL((VoiceMixKernel<EXACT, LINES>), dim3(L.numGroups), dim3(kThreads), 0, s, L, samplesToDo,
        carry ? 1u : 0u);
    return hipGetLastError();
}

// Sums the per-group partials into the bus block.  One 1024-thread workgroup per 64 consecutive
// bus floats: wave w adds its contiguous slice of groups in group order (coalesced 256-byte
// rows, 8 loads in flight), then the 16 slice sums are added in slice order -- a fixed
// summation tree, so the result is deterministic.  Lines no voice mixes into (HRTF context:
// the dry/real lines) are zero-filled, which is the caller-side clear of alc/alu.cpp:2417 and
// :2196-2198.
// Shape: the partials are cut into kReduceSegs runs of groups whose sums are then added in run
// order; a workgroup is FOUR wavefronts that take four runs each.  Four waves of <= 32 VGPRs and
// 4 KB of LDS fit on a CU BESIDE the two resident workgroups of the wavefront voice kernel (which
// leave 32 registers per lane -- 512 - 2 x 240 -- and 9 KB of LDS), so on the post stream the reduction of update k
// runs in the shadow of the voice kernel of update k+1 instead of taking workgroup slots from it
// (a 16-wave workgroup does: it holds back one voice workgroup of every CU it lands on, and that
// launch, which fills the machine exactly once, ends that much later).
// (kReduceWaves = 16, one run per wavefront, when nothing else is running: oalgpu_mix_voices.)
// (the partial buses are read once: streamed past L2, like the stores that wrote them -- profiles/r4/nt_partials_ab.txt)
#define OALGPU_PARTIAL_LOAD(p) __builtin_nontemporal_load(p)
constexpr int kReduceSegs = 16;
template<int kReduceWaves>
__global__ void __launch_bounds__(kReduceWaves * 64) __attribute__((amdgpu_num_vgpr(48))) BusReduceKernel(DeviceLayout L, const float *__restrict__ carry)
{
    // (the post-stream shape hands its runs over four at a time: 1 KB, ONE allocation granule of LDS -- beside two voice
    // workgroups of the dry-line kernels a CU has four granules to spare, and a launch of more reduction workgroups than
    // CUs (config 4: 336) must not keep a voice workgroup of the next update waiting for LDS)
    __shared__ float slice[kReduceWaves == 4 ? 4 : kReduceSegs][64];
    if constexpr (kReduceWaves == 4) __builtin_amdgcn_s_setprio(3);      // (the post-stream shape: a short chain beside a voice kernel that fills the SIMDs)
    const uint32_t wave0 = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t idx = blockIdx.x * 64u + lane;
    const uint32_t dryLines = L.numDry + L.numReal;
    const uint32_t wetLines = L.numSlots * L.wetChannels;
    const uint32_t lineFloats = (dryLines + wetLines) * kLine;
    const uint32_t total = lineFloats + (kLine + kHrirLen) * 2;

    // 32-bit element offsets off one uniform base per source (the partial buses are far below
    // 2^32 floats): one address register per load keeps the wave inside its 32 VGPRs
    const bool fromLines = blockIdx.x * 64u < lineFloats;          // lineFloats is a multiple of 1024: uniform per workgroup
    bool have = false;
    uint32_t off = 0, stride = 0;
    if(fromLines)
    {
        const uint32_t line = idx / kLine, p = idx % kLine;
        int32_t from = -1;
        if(line < dryLines) { if(!L.hrtf && line < L.numDry) from = int32_t(line); }
        else from = int32_t((L.hrtf ? 0u : L.numDry) + (line - dryLines));
        if(from >= 0) { have = true; off = uint32_t(from) * kLine + p; stride = L.mixLines * kLine; }
    }
    else if(L.hrtf && idx < total)
    {
        have = true;
        off = idx - lineFloats;
        stride = uint32_t(kLine + kHrirLen) * 2u;
    }
    // uniform per wavefront (64 | lineFloats) and kept in SGPRs as a GLOBAL pointer, so that a
    // load's address is one 32-bit VGPR offset
    typedef const __attribute__((address_space(1))) float *gfloatp;
    gfloatp base;
    {
        const uint64_t b = reinterpret_cast<uint64_t>(fromLines ? L.partLines : L.partHrtf);
        const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(b)), hi = __builtin_amdgcn_readfirstlane(uint32_t(b >> 32));
        base = reinterpret_cast<gfloatp>((uint64_t{hi} << 32) | lo);
    }

    const uint32_t ngroups = fromLines ? L.numLineGroups : L.numGroups;
    const uint32_t per = (ngroups + kReduceSegs - 1) / kReduceSegs;
    if constexpr (kReduceWaves == 4)
    {   // The post-stream shape: a wavefront's four runs advance TOGETHER, eight groups of each per step -- 32 loads in flight
        // instead of 8, a quarter of the dependent round trips (the kernel is nothing but their latency: 16 of them per
        // wavefront, 8.4 us, before).  Every run is still summed in group order, so the result does not change by a bit.
        typedef const __attribute__((address_space(1))) char *gcharp;
        const uint32_t sb = stride * 4u;
        uint32_t gA[4], gE[4], oo[4];
        float sum[4];
        uint32_t common = per;
#pragma unroll
        for(int q = 0; q < 4; ++q)
        {
            const uint32_t seg = wave0 + 4u * uint32_t(q);
            gA[q] = seg * per < ngroups ? seg * per : ngroups;
            gE[q] = (gA[q] + per < ngroups) ? gA[q] + per : ngroups;
            oo[q] = (off + gA[q] * stride) * 4u;
            sum[q] = 0.0f;
            common = (gE[q] - gA[q]) < common ? gE[q] - gA[q] : common;
        }
        if(have)
        {
#pragma unroll 1
            for(uint32_t j = 0; j + 8 <= common; j += 8)
            {
                float v[4][8];
#pragma unroll
                for(int q = 0; q < 4; ++q)
#pragma unroll
                    for(int k = 0; k < 8; ++k) v[q][k] = OALGPU_PARTIAL_LOAD(reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + (oo[q] + uint32_t(k) * sb)));
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
#pragma unroll
                    for(int k = 0; k < 8; ++k) sum[q] = sum[q] + v[q][k];
                    oo[q] += 8u * sb; gA[q] += 8u;
                }
            }
#pragma unroll
            for(int q = 0; q < 4; ++q)
                for(; gA[q] < gE[q]; ++gA[q]) { sum[q] = sum[q] + *reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + oo[q]); oo[q] += sb; }
        }
        // segments in order 0..15 (segment = wavefront + 4 q), exactly as the 16-wavefront shape sums them
        float t = (carry && idx >= lineFloats && idx < total) ? carry[idx - lineFloats] : 0.0f;
#pragma unroll
        for(int q = 0; q < 4; ++q)
        {
            slice[wave0][lane] = sum[q];
            __syncthreads();
            if(wave0 == 0) { t = t + slice[0][lane]; t = t + slice[1][lane]; t = t + slice[2][lane]; t = t + slice[3][lane]; }
            __syncthreads();
        }
        if(wave0 == 0 && idx < total && (idx < lineFloats || L.hrtf)) L.bus[idx] = t;
        return;
    }
    else
#pragma unroll 1
    for(uint32_t seg = wave0; seg < uint32_t(kReduceSegs); seg += kReduceWaves)
    {
        const uint32_t g0 = seg * per < ngroups ? seg * per : ngroups;
        const uint32_t g1 = (g0 + per < ngroups) ? g0 + per : ngroups;
        float sum = 0.0f;
        if(have)
        {
            // byte offsets (the partial buses are far below 4 GB)
            typedef const __attribute__((address_space(1))) char *gcharp;
            const uint32_t sb = stride * 4u;
            uint32_t g = g0, o = (off + g0 * stride) * 4u;
#pragma unroll 1
            for(; g + 8 <= g1; g += 8)
            {
                float v[8];
#pragma unroll
                for(int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + (o + uint32_t(k) * sb));
#pragma unroll
                for(int k = 0; k < 8; ++k) sum = sum + v[k];
                o += 8u * sb;
            }
            for(; g < g1; ++g) { sum = sum + *reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + o); o += sb; }
        }
        slice[seg][lane] = sum;
    }
    __syncthreads();
    if(wave0 == 0 && idx < total)
    {
        float t = (carry && idx >= lineFloats) ? carry[idx - lineFloats] : 0.0f;
        t = t + slice[0][lane];
#pragma unroll 5
        for(int w = 1; w < kReduceSegs; ++w) t = t + slice[w][lane];
        if(idx < lineFloats || L.hrtf) L.bus[idx] = t;
    }
}

} // namespace

hipError_t LaunchVoiceMix(hipStream_t s, bool exact, const DeviceLayout &L, uint32_t samplesToDo, bool carryAccum)
{
    const uint32_t lines = L.mixLines;
    if(exact)
    {
        if(lines <= 8) return LaunchVoiceMixT<true, 8>(s, L, samplesToDo, carryAccum);
        if(lines <= 16) return LaunchVoiceMixT<true, 16>(s, L, samplesToDo, carryAccum);
        return LaunchVoiceMixT<true, 32>(s, L, samplesToDo, carryAccum);
    }
    if(lines <= 8) return LaunchVoiceMixT<false, 8>(s, L, samplesToDo, carryAccum);
    if(lines <= 16) return LaunchVoiceMixT<false, 16>(s, L, samplesToDo, carryAccum);
    return LaunchVoiceMixT<false, 32>(s, L, samplesToDo, carryAccum);
}

// evDone: an event bound to the dispatch's completion (null: none)
void LaunchBusReduce(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const float *carry, bool besideVoiceKernel, hipEvent_t evDone)
{
    const uint32_t total = uint32_t(BusFloats(L));
    (void)samplesToDo;
    if(besideVoiceKernel)
        hipExtLaunchKernelGGL(BusReduceKernel<4>, dim3((total + 63u) / 64u), dim3(4 * 64), 0, s, nullptr, evDone, 0u, L, carry);
    else
        hipExtLaunchKernelGGL(BusReduceKernel<16>, dim3((total + 63u) / 64u), dim3(16 * 64), 0, s, nullptr, evDone, 0u, L, carry);
}

} // namespace oalgpu
