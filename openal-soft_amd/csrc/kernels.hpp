// Host-visible declarations of the kernel launchers and the device-side data layout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_hrtf.hpp"
#include "dev_mix.hpp"
#include "dev_resample.hpp"
#include "dev_voice.hpp"

namespace oalgpu {

// ---------------------------------------------------------------------------------------------
// HBM layout of the batched path.  Everything Voice::mix reads or mutates lives in
// structure-of-arrays form indexed by voice, so a workgroup streaming voice v touches a few
// contiguous segments:
//   ctl[v]        VoiceCtl (128 B): integer state + resampler state + flags (core/voice.h:176-270)
//   prev[v][48]   mPrevSamples[0]
//   dfilt[v][2]   DirectParams::{LowPass,HighPass}
//   hrtfOld/Tgt   [v][irStride][2]  HrtfFilter::Coeffs of Hrtf.Old / Hrtf.Target
//   hist[v][64]   DirectParams::Hrtf.History
//   gainCur/Tgt   [v][numDry]       DirectParams::Gains
//   sfilt[v][sends][2], sendCur/Tgt [v][sends][wet]   SendParams
// ---------------------------------------------------------------------------------------------
enum VoiceFlagBits : uint32_t {
    kFlagFading = 1u << 0,          // VoiceFlag::IsFading
    kFlagHasHrtf = 1u << 1,         // VoiceFlag::HasHrtf
    kFlagDirectFilter = 1u << 2,    // mDirect.FilterActive
    kFlagHrtfDirty = 1u << 3,       // Hrtf.Target replaced since the last mix (Old != Target)
    kFlagAmbiScale = 1u << 4,       // VoiceFlag::IsAmbisonic: ambi[v] holds the channel's splitter and scales
    kFlagNfc = 1u << 5,             // VoiceFlag::HasNfc: nfc[v] holds DirectParams::NFCtrlFilter
    kFlagQueue = 1u << 7,           // a streaming (buffer-queue) source, not VoiceFlag::IsStatic (voice.cpp:563-594, :1182-1194)
    kFlagDelayed = 1u << 6,         // mStartTime lies ahead: startDelay[v] samples until the voice starts (voice.cpp:1023-1046)
    kFlagSendFilterShift = 8        // bits 8..13: mSend[i].FilterActive
};

struct alignas(16) VoiceCtl {
    int32_t playState;
    int32_t position;               // mPosition
    uint32_t positionFrac;          // mPositionFrac
    int32_t curBuffer;              // mCurrentBuffer (-1 = null)
    int32_t loopBuffer;             // mLoopBuffer (-1 = null)
    uint32_t step;                  // mStep
    int32_t rsKind;                 // which Resample_* (oalgpu_interp_state.kind)
    uint32_t rsM, rsL;
    float rsSf;
    uint32_t rsFilterOffset;        // float offset of the filter inside the table blob
    uint32_t flags;
    int32_t sendSlot[6];
    uint32_t hrtfOldDelay[2];
    float hrtfOldGain;
    uint32_t hrtfTgtDelay[2];
    float hrtfTgtGain;
    BufferItem buf;                 // copy of buffers[curBuffer] (descriptors are immutable once registered):
                                    // the voice kernel gets head, HRTF delays and buffer in ONE round trip
};
static_assert(sizeof(VoiceCtl) == 128, "VoiceCtl is one 128-byte line");

// ChannelData::mAmbiSplitter (BandSplitter: coefficient + three delay elements), mAmbiHFScale,
// mAmbiLFScale of a B-Format channel voice
struct alignas(16) AmbiScaleState { float coeff, lpZ1, lpZ2, apZ1, hfScale, lfScale; uint32_t pad[2]; };

// DirectParams::NFCtrlFilter (NfcFilter, core/filters/nfc.h): sections of order 1..4;
// a[o] = {a0, a1..ao}, b[o] = {-, b1..bo}, z[o] = the section's delay elements
struct alignas(16) NfcState { float a[5][5], b[5][5], z[5][4]; uint32_t pad[10]; };
static_assert(sizeof(NfcState) == 320, "NfcState");

// a BiquadState padded to 64 bytes so each filter is one aligned segment
struct alignas(16) BiquadSlot { BiquadState f; uint32_t pad[3]; };
static_assert(sizeof(BiquadSlot) == 64, "BiquadSlot");

// The resolved MixLine gains (dev_mix.hpp MixLineGain) of one stream row over the mix lines, as
// the voice kernel's tail (WgMixRows, voice_wave.hip) consumes them: a block of 3*S + 8 dwords per row, S = lineStride,
//   [gain[S] | rampA[S] | rampB[S] | live, rampLen, 0...]
// every frame uses gain[c] (0 when the line's constant part is not mixed); frames below rampLen
// add rampA[c] + rampB[c]*frame, the ramp's distance from the constant (0 for lines without a
// ramp).  live == 0: nothing to mix from this row in this update.
__host__ __device__ inline uint32_t LineBlockDwords(uint32_t lineStride) { return 3u * lineStride + 8u; }

// Parameter block applied on the GPU by ApplyParamsKernel (one per changed voice).
struct ParamRecord {
    uint32_t voice;
    uint32_t step;
    int32_t rsKind;
    uint32_t rsM, rsL;
    float rsSf;
    uint32_t rsFilterOffset;
    uint32_t flags;                 // kFlagDirectFilter | send filter bits
    int32_t sendSlot[6];
    float dirLp[5], dirHp[5];       // designed biquad coefficients (host libm)
    float sendLp[6][5], sendHp[6][5];
    float hrtfDir[4];               // elevation, azimuth, distance, spread
    float hrtfGain;
    float dryGains[32];
    float sendGains[6][25];
    // HrtfStore::getCoeffs' index half (core/hrtf.cpp:192-245), evaluated by the host when the record is built
    // (HrtfBlendFor, the same operations as on the device): the four HRIRs, their weights, the pass-through
    // tap and the blended delays -- what is left for the GPU is one round of loads and the weighted sum
    uint32_t hrtfIdx[4];
    float hrtfW[4];
    float hrtfPass;
    uint32_t hrtfDelay[2];
    uint32_t keepHrtf;              // the HRTF target stays as it is (handed over by oalgpu_voice_set_hrtf_targets)
};

struct DeviceLayout {
    // configuration
    uint32_t numVoices, numDry, numReal, numSends, numSlots, wetChannels;
    uint32_t hrtf, irSize, irStride;       // irStride: taps stored per voice filter (irSize rounded up to 8)
    uint32_t voicesPerGroup, numGroups;
    uint32_t waveVoices;                    // voice_wave.hip: voices per wavefront (0 = not used)
    uint32_t firMfma;                       // voice_wave.hip: the HRTF FIR (IrSize <= 64) on the matrix pipe in split half
                                            // precision (default) instead of packed VALU FMAs (OALGPU_CTX_FIR_VALU)
    uint32_t mixLines;                      // lines accumulated by the voice kernel
    uint32_t sliceLines;                    // voice_slice.hip: the context's voices are mixed a wavefront per 256-frame slice (0 = no)
    uint32_t accLines;                      // voice_wave.hip: the mix lines accumulate in the wavefronts' registers (<= 8 lines;
                                            // the kernel's ACCL: 4, 6 or 8) instead of leaving stream rows; 0 = stream rows
    uint32_t rows8, rowsVpg;                // voice_rows.hip (OALGPU_CTX_ROW_SLICES): dry lines and sends with the rows kept in LDS -- a wavefront per
                                            // voice produces, a wavefront per 128-frame slice of every line consumes; rowsVpg voices per workgroup
    uint32_t wave16;                        // voice_wave16.hip: one voice per wavefront, this many (4, 8 or 16) wavefronts per workgroup (HRTF
                                            // contexts without sends, IrSize <= 64; 0 = voice_wave.hip's two voices per wavefront,
                                            // OALGPU_CTX_WAVE_PAIRS); numGroups = voices / wave16 then
    // tables + buffers
    const float *tables;                    // [bsinc12 | bsinc24 | bsinc48 | spline | gaussian]
    const BufferItem *buffers;
    // voice state
    VoiceCtl *ctl;
    float *prev;
    BiquadSlot *dfilt;
    float *hrtfOld, *hrtfTgt, *hist;
    float *gainCur, *gainTgt;
    BiquadSlot *sfilt;
    float *sendCur, *sendTgt;
    AmbiScaleState *ambi;                   // [voice]
    uint32_t *startDelay;                   // [voice] samples until a delayed voice starts (kFlagDelayed)
    uint32_t *queueDone;                    // [voice] buffers a streaming voice has played through (AsyncBufferCompleteEvent counts)
    NfcState *nfc;                          // [voice], null unless the context has NFC
    uint32_t chansPerOrder[5];              // DeviceBase::NumChannelsPerOrder (NFC contexts)
    uint32_t nfcOrders;                     // orders 1.. with lines (0 = no NFC)
    // partial buses written by the voice kernel: [group][mixLines][1024], [group][1152][2]
    float *partLines, *partHrtf;
    // wavefront kernel, dry-line and send mixing: stream rows [voice][streamsPerVoice][1024] and
    // their gain blocks, consumed by the kernel's own tail (WgMixRows), which writes one partial bus per workgroup
    float *streams;
    uint32_t *lineGains;
    uint32_t numLineGroups;                 // groups of partLines (== numGroups)
    uint32_t lineStride;                    // gain vector width of a stream row: mixLines rounded up to 8 / 16 / 32
    uint32_t streamsPerVoice;               // stream rows per voice: 2 + numSends (see voice_wave.hip)
    // final bus block: [(numDry+numReal) x 1024 | numSlots*wetChannels x 1024 | 1152 x 2]
    float *bus;
    const float *hrirs;                     // HrtfStoreDev::coeffs: the store's HRIRs (ApplyRecordWave)
};

__host__ __device__ inline size_t BusWetOffset(const DeviceLayout &L) { return size_t{L.numDry + L.numReal} * kLine; }
__host__ __device__ inline size_t BusAccumOffset(const DeviceLayout &L)
{ return BusWetOffset(L) + size_t{L.numSlots} * L.wetChannels * kLine; }
__host__ __device__ inline size_t BusFloats(const DeviceLayout &L)
{ return BusAccumOffset(L) + size_t{kLine + kHrirLen} * 2; }

// The weighted sum of HrtfStore::getCoeffs (core/hrtf.cpp:247-259) into voice v's target filter, by one wavefront.
__device__ __forceinline__ void ApplyHrtfTargetWave(const DeviceLayout &L, uint32_t v, const uint32_t (&idx)[4], const float (&w)[4],
    float pass, uint32_t lane)
{
    const uint32_t i0 = idx[0], i1 = idx[1], i2 = idx[2], i3 = idx[3];
    const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
    // The mixers apply IrSize taps (rounded up to even: ApplyCoeffs works on pairs, mixer_sse.cpp:46-51); a data
    // set resampled to the device's rate carries non-zero taps beyond that (GetLoadedHrtf resamples whole
    // HrirArrays), which the voice kernels' fixed-length FIRs must not see
    const uint32_t live = ((L.irSize + 1u) & ~1u) * 2u;
    for(uint32_t e = lane; e < L.irStride * 2; e += 64)
    {   // hrtf.cpp:247-259: the pass-through tap (elements 0, 1) or 0, then the four weighted HRIRs in order
        float x = (e < 2) ? pass : 0.0f;
        x = L.hrirs[size_t{i0} * (kHrirLen * 2) + e] * w0 + x;
        x = L.hrirs[size_t{i1} * (kHrirLen * 2) + e] * w1 + x;
        x = L.hrirs[size_t{i2} * (kHrirLen * 2) + e] * w2 + x;
        x = L.hrirs[size_t{i3} * (kHrirLen * 2) + e] * w3 + x;
        __builtin_nontemporal_store((e < live) ? x : 0.0f, &L.hrtfTgt[size_t{v} * L.irStride * 2 + e]);    // (read by the next launch)
    }
}

// One parameter record applied by one wavefront (ApplyParamsKernel): the CalcVoiceParams results scattered into
// the voice arrays, the weighted sum of HrtfStore::getCoeffs (core/hrtf.cpp:247-259) and the
// BiquadInterpFilter::setParams state machine (biquad.cpp:131-149).
__device__ __forceinline__ void ApplyRecordWave(const DeviceLayout &L, const ParamRecord &r, uint32_t lane)
{
    const uint32_t v = r.voice;
    VoiceCtl &ctl = L.ctl[v];
    if(lane == 0)
    {
        ctl.step = r.step;
        ctl.rsKind = r.rsKind; ctl.rsM = r.rsM; ctl.rsL = r.rsL; ctl.rsSf = r.rsSf;
        ctl.rsFilterOffset = r.rsFilterOffset;
        const uint32_t keep = ctl.flags & (kFlagFading | kFlagHasHrtf | kFlagAmbiScale | kFlagNfc | kFlagDelayed | kFlagQueue
            | (r.keepHrtf ? uint32_t(kFlagHrtfDirty) : 0u));
        ctl.flags = keep | (r.flags & ~(kFlagFading | kFlagHasHrtf | kFlagHrtfDirty | kFlagAmbiScale | kFlagNfc | kFlagDelayed | kFlagQueue))
            | (L.hrtf ? (kFlagHasHrtf | (r.keepHrtf ? 0u : uint32_t(kFlagHrtfDirty))) : 0u);
        for(int i = 0; i < 6; ++i) ctl.sendSlot[i] = (uint32_t(i) < L.numSends) ? r.sendSlot[i] : -1;
        if(L.hrtf && !r.keepHrtf)
        {
            ctl.hrtfTgtDelay[0] = r.hrtfDelay[0]; ctl.hrtfTgtDelay[1] = r.hrtfDelay[1];
            ctl.hrtfTgtGain = r.hrtfGain;
        }
    }
    if(lane == 1) BiquadSetTarget(L.dfilt[size_t{v} * 2 + 0].f, r.dirLp);
    if(lane == 2) BiquadSetTarget(L.dfilt[size_t{v} * 2 + 1].f, r.dirHp);
    if(lane >= 8 && lane < 8 + 2 * L.numSends)
    {
        const uint32_t i = (lane - 8) >> 1, hp = (lane - 8) & 1u;
        BiquadSetTarget(L.sfilt[(size_t{v} * L.numSends + i) * 2 + hp].f, hp ? r.sendHp[i] : r.sendLp[i]);
    }
    if(L.hrtf) { if(!r.keepHrtf) ApplyHrtfTargetWave(L, v, r.hrtfIdx, r.hrtfW, r.hrtfPass, lane); }
    else if(lane < L.numDry)
        L.gainTgt[size_t{v} * L.numDry + lane] = r.dryGains[lane];
    for(uint32_t k = lane; k < L.numSends * L.wetChannels; k += 64)
        L.sendTgt[size_t{v} * L.numSends * L.wetChannels + k] = r.sendGains[k / L.wetChannels][k % L.wetChannels];
}

struct VoiceInitRecord { uint32_t voice; int32_t buffer, looping, position; uint32_t positionFrac; int32_t queue; };

// ---------------------------------------------------------------------------------------------
// The resident voice kernel (OALGPU_CTX_RESIDENT, voice_wave.hip): ONE launch of the HRTF voice kernel stays on its
// voices over many updates.  Every update is still submitted by its own oalgpu_mix_update: the host writes the update's
// slot (its parameter block and its length) and then rings the doorbell -- plain stores into memory the kernel polls
// with system-scope loads (fine-grained device memory behind the BAR, or pinned host memory).  A workgroup starts update
// u as soon as IT is through with u - 1: it installs the block's records of its own voices (ApplyRecordLean), mixes,
// stores its partial bus written-through into set u % kResidentSets and counts itself in on that set's arrival counter,
// which the update's reduction (BusReduceResidentKernel, its own launch on the context's reduce stream) polls.
// Counters only ever grow; every wait is a signed difference against a target the host passes.
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kResidentSets = 4;           // partial-bus sets: a workgroup may be this many updates ahead of the reduction
constexpr uint32_t kResidentSlots = 16;         // doorbell slots (the host stays <= kResidentDepth updates ahead of the post-process)
constexpr uint32_t kResidentDepth = 8;
constexpr unsigned long long kResidentWatchdogTicks = 200000000ull;    // s_memrealtime ticks (100 MHz): 2 s without progress = give up

// recs / map: the update's parameter block (0: none); rows: its records' blended target HRIRs, [record][irStride][2]
struct ResidentSlot { unsigned long long recs, map, rows; uint32_t samples, pad; };      // 32 bytes
struct alignas(64) ResidentDoor {
    uint32_t seq;                               // updates rung so far: update u may run once int32(seq - u) > 0
    uint32_t exitSeq[4];                        // [launch id % 4]: that launch leaves once its next update is >= this (host: park).
                                                // Per launch: the next launch is queued -- and the door is ringing for it -- while
                                                // the one that was parked is still finishing what had been rung for IT
    uint32_t pad[11];
    ResidentSlot slot[kResidentSlots];          // slot[u % kResidentSlots]: written before seq
};
// device counters of a resident context, one uint32 each, 64 bytes apart (index x 16)
enum ResidentCounter : uint32_t {
    kRcArrive0 = 0,                             // .. kRcArrive0 + kResidentSets - 1: voice workgroups that stored their partial into the set
    kRcRedRead = 4,                             // reduction workgroups that are through READING their partial set
    kRcRedDone = 5,                             // reduction workgroups whose sums are in the bus block
    kRcPostDone = 6,                            // post-process FIR workgroups that are through (stores included)
    kRcStarted = 7,                             // voice workgroups of all resident launches that have started
    // where the time between updates goes (s_memrealtime ticks, 10 ns, summed; oalgpu_resident_stats):
    kRcWaitDoor = 8,                            // voice workgroups (all of them): waiting for the doorbell
    kRcWaitRed = 9,                             // voice workgroups: waiting for the reduction that still reads their partial set
    kRcWaitArrive = 10,                         // the reduction's first workgroup: waiting for the voice workgroups' arrival
    kRcWaitPost = 11,                           // the reduction's first workgroup: waiting for the post-process of the update before
    kRcWaitRedDone = 12,                        // the post-process's first split workgroup: waiting for the reduction
    kRcWaitSplit = 13,                          // the post-process's first FIR workgroup: waiting for the splits
    kRcInstall = 14,                            // voice workgroups: from the doorbell seen to the parameter block installed (both barriers)
    kRcBusy = 15,                               // voice workgroups: from there to the arrival on the partial set
    kRcTop = 16,                                // voice workgroups: from the arrival to the next update seen (the door's loads and both waits above)
    kRcCount = 17
};
// pinned host words the kernels write (system scope), 64 bytes apart (index x 16)
enum ResidentHostFlag : uint32_t { kRhResident = 0, kRhError = 1, kRhProgress = 2,
    kRhFault = 3,           // what a voice workgroup that gave up was looking at: [+0] update, [+1] doorbell, [+2] exit word, [+3] reduction counter,
                            // [+4] the count it wanted, [+5] workgroup, [+6] launch id, [+7] ticks waited
    kRhCount = 4 };
struct ResidentArgs {
    const ResidentDoor *door;
    uint32_t *counters;                         // [kRcCount * 16]
    uint32_t *hostFlags;                        // [kRhCount * 16], pinned host memory
    float *partBase;                            // kResidentSets sets of [group][1152][2]
    uint32_t setStride;                         // floats between two sets
    uint32_t base, endSeq;                      // this launch's first update, and the update in front of which it leaves by itself
    uint32_t redPerUpdate;                      // reduction workgroups per update
    uint32_t startedTarget, launchId;           // kRcStarted reaches startedTarget when every workgroup of THIS launch has started
};

// ---- launchers (percall_kernels.hip) ----
void LaunchResample(hipStream_t s, bool exact, const ResampleSpec &spec, const float *src, uint32_t frac,
    uint32_t increment, float *dst, uint32_t n);
void LaunchMix(hipStream_t s, const float *in, uint32_t n, float *out, uint32_t nlines, float *cur,
    const float *tgt, uint32_t counter, uint32_t outpos);
void LaunchMixHrtf(hipStream_t s, bool exact, const float *in, float *accum, uint32_t irsize, const float *coeffs,
    uint32_t dL, uint32_t dR, float gain, float step, const float *oldcoeffs, uint32_t odL, uint32_t odR,
    float oldgain, int blend, uint32_t n);
void LaunchMixDirectHrtf(hipStream_t s, bool exact, float *left, float *right, const float *in, uint32_t nch,
    float *accum, SplitterState *splitters, const float *hfscales, const float *chanCoeffs, uint32_t irsize,
    uint32_t n, float *temp);
void LaunchBiquadDual(hipStream_t s, BiquadState *f0, BiquadState *f1, const float *src, float *dst, uint32_t n);
void LaunchGetCoeffs(hipStream_t s, const HrtfStoreDev &st, const float *dirs, uint32_t count, float *coeffs,
    uint32_t *delays);

// ---- launcher (post_wave.hip): FAST MixDirectHrtf, one wavefront per dry channel ----
void LaunchPostDirectHrtfFast(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, float *accum,
    SplitterState *splitters, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n, float *temp,
    hipEvent_t evDone = nullptr);

// the same as ONE launch (pipelined FAST HRTF contexts): see post_wave.hip
// workgroups of the reduction inside LaunchReducePostFused: 64 bus columns each (BusFloats(L) / 64, rounded up)
inline uint32_t ReducePostReduceGroups(const DeviceLayout &L) { return uint32_t((BusFloats(L) + 63u) / 64u); }
void LaunchReducePostFused(hipStream_t s, const DeviceLayout &L, const float *carry, float *left, float *right, const float *in, uint32_t nch,
    const float *accIn, float *carryOut, const SplitterState *spIn, SplitterState *spOut, const float *hfscales, const float *chanCoeffs,
    uint32_t irsize, uint32_t n, float *xf, uint32_t *arrived, uint32_t epoch, const float runPower[4], hipEvent_t evDone,
    float *hostOut, uint32_t *hostFlag, uint32_t hostSeq, uint32_t *outArrived, uint32_t outTarget, uint32_t *reduced, uint32_t reducedEpoch);
void LaunchPostDirectHrtfFused(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, const float *accIn, float *carryOut,
    const SplitterState *spIn, SplitterState *spOut, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n,
    float *xf, uint32_t *arrived, uint32_t epoch, const float runPower[4], hipEvent_t evDone = nullptr,
    float *hostOut = nullptr, uint32_t *hostFlag = nullptr, uint32_t hostSeq = 0, uint32_t *outArrived = nullptr, uint32_t outTarget = 0);
// resident contexts (OALGPU_CTX_RESIDENT): the reduction and the post-process of one update as launches that wait for device
// counters instead of for the voice kernel's end (post_wave.hip)
void LaunchBusReduceResident(hipStream_t s, const DeviceLayout &L, const float *carry, uint32_t *counters, uint32_t *hostFlags, uint32_t set,
    uint32_t arriveTarget, uint32_t postDoneTarget);
uint32_t PostResidentFirGroups();
void LaunchPostResident(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, const float *accIn, float *carryOut,
    const SplitterState *spIn, SplitterState *spOut, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n,
    float *xf, uint32_t *arrived, uint32_t epoch, const float runPower[4], hipEvent_t evDone, float *hostOut, uint32_t *hostFlag, uint32_t hostSeq,
    uint32_t *counters, uint32_t *hostFlags, uint32_t redDoneTarget, uint32_t postDoneTarget, uint32_t progressValue);

// ---- launchers (output_kernels.hip): BFormatDec, ApplyDither, Write<T> behind the buses ----
// gainsHf / gainsLf: [dry line][32] (column = output line); gainsLf null = single-band decoder; bands =
// scratch for [dry line][hp | lp][1024]
void LaunchBFormatDecode(hipStream_t s, bool exact, float *out, const float *lines, SplitterState *states, float *bands,
    const float *gainsHf, const float *gainsLf, uint32_t nin, uint32_t nout, uint32_t n);
void LaunchDither(hipStream_t s, float *lines, uint32_t nlines, uint32_t n, float quantScale, uint32_t seed);
uint32_t DitherAdvanceSeed(uint32_t seed, uint32_t draws);
void LaunchWriteSamples(hipStream_t s, int sampleType, const float *lines, uint32_t nlines, uint32_t n, uint32_t frameStep, void *out);
// ---- launcher (effects_kernels.hip): equalizer / modulator / echo / dedicated, one workgroup per instance ----
constexpr uint32_t kFxMaxIn = 16;
struct FxState {                       // device-resident per instance
    BiquadState bq[kFxMaxIn][4];       // EQ: four per wet channel; MOD: [c][0] high-pass; ECHO: [0][0] damping shelf
    float cur[64];                     // Current gains -- EQ / MOD: [wet channel]; DEDICATED: [line]; ECHO: [tap][32]
    float env;                         // CompressorState::mEnvFollower
};
struct FxLaunch {
    int kind, exact;
    uint32_t numIn, nlines, n;
    FxState *st;
    const float *wetIn; float *outLines;
    uint32_t target[kFxMaxIn]; float tgtGain[kFxMaxIn];   // EQ / MOD: mChans[c].mTargetChannel / mTargetGain
    const float *tgtGains;                                // DEDICATED: [nlines]; ECHO: [2][nlines] (device memory)
    uint32_t modIndex, modRange; float modScale; int modWave;        // 0 one, 1 sin, 2 saw, 3 square
    float *delay; uint32_t delayMask, offset, tap[2]; float feedGain;
    int compOn; float attackMult, releaseMult;                       // compressor
};
void LaunchEffect(hipStream_t s, const FxLaunch &F);

// ---- launcher (effects2_kernels.hip): chorus / flanger, distortion, autowah, vocal morpher, frequency shifter ----
struct Fx2State {                      // device-resident per instance
    float cur[kFxMaxIn];               // mChans[c].mCurrentGain
    float upCur[9][32];                // UpsampleParams::mCurrentGains (A-Format effects: 4 rows; pitch shifter: 9)
    float split[9][3];                 // UpsampleParams::mSplitter: lp_z1, lp_z2, ap_z1
    BiquadState lp[4], bp[4];          // distortion: mChans[c].mLowpass / mBandpass
    float envDelay, awZ[kFxMaxIn][2];  // autowah: mEnvDelay, mChans[c].mFilter
    float vmS[kFxMaxIn][8][2];         // vocal morpher: [channel][vowel A 0..3 | vowel B 4..7] {mS1, mS2}
};
struct FsPair { double x, y; };        // std::complex<double>
struct Fx2Launch {
    int kind;
    uint32_t numIn, nlines, n;
    Fx2State *st;
    const float *wetIn; float *outLines;
    uint32_t target[kFxMaxIn]; float tgtGain[kFxMaxIn];
    int upsample; float hfScale[2], splitCoeff; const float *upTgt;       // mUpsampler: [4 or 9][32] target gains (device memory)
    // chorus (alc/effects/chorus.cpp)
    int chWave; uint32_t lfoStart[2], lfoRange; float lfoScale, chDepth; int32_t chDelay; float chFeedback;
    uint32_t chAvgDelay, chHist;
    float *delay; uint32_t delayMask, offset; const float *cubic;
    // distortion
    float edgeCoeff;
    // autowah
    float attackRate, releaseRate, resonanceGain, peakGain, freqMinNorm, bandwidthNorm;
    // vocal morpher: the eight formants (the same for every channel)
    int vmWave; uint32_t vmIndex, vmStep; float vmG[8], vmGain[8];
    // frequency shifter
    double *fsIn; FsPair *fsOutFifo, *fsAccum, *fsOutdata; const FsPair *fsTw, *fsPhase; const float *fsWindow;
    uint32_t fsCount, fsPos, fsPhaseStep[4], fsPhaseIdx[4]; double fsSign[4];
    // pitch shifter: [parity] = what this launch reads, [parity ^ 1] = what it leaves
    const float *psRingIn; float *psRingOut;            // [9][1024] the last 1024 input samples per channel
    const float *psPhaseIn; float *psPhaseOut;          // mLastPhase[513] | mSumPhase[513]
    float *psAccum, *psOutFifo, *psRows;                // mOutputAccum [9][1024]; the hop's output [9][128]; mBBuffer [9][1024]
    const float *psTw, *psWindow;                       // e^(-2 pi i m / 1024), m < 512 (re, im); the Hann window
    uint32_t psCount, psPos, psPitchI; float psPitch;
};
void LaunchEffect2(hipStream_t s, const Fx2Launch &F, uint32_t ldsBytes);

// SampleConverter::convert (core/converter.cpp:236-330): one launch per call, see output_kernels.hip
struct ConvertChunk { uint32_t srcBase, frac0, dstBase, dstSize; };      // timeline index of SrcData[0], DataPosFrac, first output, DstSize
struct ConvertJob {
    ResampleSpec spec; uint32_t increment;
    const float *prev; float *newPrev;          // [channel][48] PrevSamples before / after the call
    const void *src; void *dst;                 // interleaved frames, device memory
    uint32_t prep0, srcFrames, channels, numChunks, endBase, nextPrep;
    int srcType, dstType;
};
void LaunchSampleConvert(hipStream_t s, const ConvertJob &J, const ConvertChunk *chunks);

// ---- launcher (conv_kernels.hip): ConvolutionState::process for a mono response ----
struct ConvLayoutHost {
    uint32_t numSegs, ringSlots, nlines, n, fifoPos, curSeg, numBlocks, numChunks, segsPerChunk;
    const float *wetIn;
    float *xhist, *ring;
    const float *filt, *fir;
    float *outFifo, *partial, *cur;
    const float *tgt;
    float *outLines;
    const float *tw128, *tw256;
    uint32_t *ticket;
    float *firOut;
    uint32_t channels, upsample;
    float *chanOut;
    SplitterState *split;
    const float *hfScale, *lfScale;
};
void LaunchConvolution(hipStream_t s, const ConvLayoutHost &h);

// ---- launchers (voice_kernel.hip) ----
void LaunchInitVoices(hipStream_t s, const DeviceLayout &L, const VoiceInitRecord *recs, uint32_t count);
void LaunchApplyParams(hipStream_t s, const DeviceLayout &L, const ParamRecord *recs, uint32_t count);
// rows[record][irStride][2] = the records' target HRIRs as ApplyHrtfTargetWave leaves them in hrtfTgt (HRTF contexts; records with
// keepHrtf are skipped): evaluated once per parameter block, for the resident voice kernel's install
void LaunchBlendRows(hipStream_t s, const DeviceLayout &L, const ParamRecord *recs, uint32_t count, float *rows);
// moves: `count` oalgpu_voice_move records (24 bytes each: voice, the getCoeffs arguments, the gain), device or pinned host memory
void LaunchApplyMoves(hipStream_t s, const DeviceLayout &L, const HrtfStoreDev &st, const void *moves, uint32_t count, hipEvent_t evDone = nullptr);
// Hrtf.Target handed over as the reference's parameter stage left it: coeffs = [count][128][2] (HrirArray)
struct TargetRecord { uint32_t voice; uint32_t delay[2]; float gain; };
void LaunchApplyTargets(hipStream_t s, const DeviceLayout &L, const TargetRecord *recs, const float *coeffs, uint32_t count);
// returns hipSuccess or the launch error
hipError_t LaunchVoiceMix(hipStream_t s, bool exact, const DeviceLayout &L, uint32_t samplesToDo, bool carryAccum);
// besideVoiceKernel: the post-stream shape (4-wave workgroups of <= 32 VGPRs that fit on a CU next to
// the wavefront voice kernel's two workgroups)
// carry: the carried HrtfAccumData (1152 x 2) added to the voices' partial sums -- the bus's own accumulator region (in place) or
// the buffer the fused post-process left it in; null: none
void LaunchBusReduce(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const float *carry, bool besideVoiceKernel = false,
    hipEvent_t evDone = nullptr);

// ---- launchers (voice_wave.hip): the FAST HRTF hot path, one wavefront per voice ----
void LaunchSetAmbiScale(hipStream_t s, const DeviceLayout &L, uint32_t voice, const AmbiScaleState &st);
void LaunchSetNfc(hipStream_t s, const DeviceLayout &L, uint32_t voice, const NfcState &coeffs);

// ---- launcher (pan_kernels.hip): CalcDirectionCoeffs + ComputePanGains for `count` voices ----
struct AmbiMapEntry { uint32_t index; float scale; };          // BFChannelConfig, core/device.h
struct PanRecord { uint32_t voice; float dir[3]; float spread; float dryGain; float sendGain[6]; };
void LaunchPanGains(hipStream_t s, const DeviceLayout &L, const PanRecord *recs, uint32_t count, const AmbiMapEntry *dryMap,
    const AmbiMapEntry *wetMaps);
void LaunchSetStartDelay(hipStream_t s, const DeviceLayout &L, uint32_t voice, uint32_t samples);
// oalgpu_voice_events_async: hostSlot = [count, 0, 0, 0 | capacity x 8 dwords] of pinned host memory; counters: two device words, zero
void LaunchVoiceEvents(hipStream_t s, const DeviceLayout &L, uint32_t *snapshot, uint32_t *hostSlot, uint32_t capacity, uint32_t *counters);
void LaunchSetVoiceWindow(hipStream_t s, const DeviceLayout &L, uint32_t voice, int32_t buffer, uint32_t sampleLen, int32_t position);

// ---- launcher (adpcm_kernels.hip): IMA4 / MS ADPCM blocks -> interleaved 16-bit PCM, one thread per block and channel ----
void LaunchDecodeAdpcm(hipStream_t s, bool msadpcm, const uint8_t *src, int16_t *dst, uint32_t numBlocks, uint32_t samplesPerBlock,
    uint32_t channels, uint32_t sampleLen);
bool WaveKernelApplies(bool exact, const DeviceLayout &L);
uint32_t WaveKernelAccLines(const DeviceLayout &L);     // the ACCL the wavefront kernel would run this layout with (0: stream rows)
const char *WaveKernelName(const DeviceLayout &L);
uint32_t WaveKernelGroups(const DeviceLayout &L);
// the measurement variant's extras (OALGPU_CTX_PROFILE, tools/phase_times.py): s_memtime stamps
// [voice][8] | [wavefront][4], and the stages to skip; production launches pass null
struct WaveProf { unsigned long long *times; uint32_t ablate; };
// nextRecs / nextMap: a parameter block every wavefront installs for the voices it mixed, in its epilogue (null: none; only
// the kernels WaveKernelAppliesRecords names)
hipError_t LaunchVoiceWave(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const WaveProf *prof = nullptr,
    hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr, const ParamRecord *nextRecs = nullptr, const int32_t *nextMap = nullptr,
    const float *nextRows = nullptr);
bool WaveKernelAppliesRecords(const DeviceLayout &L);
// ---- launcher (voice_slice.hip): dry-line / send contexts with up to 24 mix lines, a wavefront per 256-frame slice ----
uint32_t SliceKernelLines(const DeviceLayout &L);        // the accumulator lines the slice kernel would run this layout with (0: not its)
const char *SliceKernelName();
hipError_t LaunchVoiceSlice(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, hipEvent_t evStart = nullptr, hipEvent_t evStop = nullptr);
// the resident launch of the HRTF hot path (OALGPU_CTX_RESIDENT): see ResidentDoor above
bool WaveKernelHasResident(const DeviceLayout &L);
// ---- launcher (voice_rows.hip): dry lines and sends, the rows kept on the compute unit ----
bool RowsKernelApplies(const DeviceLayout &L);
const char *RowsKernelName();
uint32_t RowsWavesPerGroup();
hipError_t LaunchVoiceRows(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const WaveProf *prof, hipEvent_t evStart, hipEvent_t evStop,
    const ParamRecord *nextRecs, const int32_t *nextMap);
// ---- launcher (voice_wave16.hip): the HRTF hot path at four wavefronts per SIMD, one voice per wavefront ----
bool Wave16Applies(const DeviceLayout &L);
// the rows of workgroup-sized runs of voices onto partial buses, as a launch of its own (voice_wave.hip)
hipError_t LaunchStreamRowsMix(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, uint32_t vpg, hipEvent_t evStop);
uint32_t Wave16WavesFor(uint32_t voices, uint32_t cus);
uint32_t Wave16Groups(const DeviceLayout &L);
const char *Wave16KernelName(const DeviceLayout &L);
hipError_t LaunchVoiceWave16(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const WaveProf *prof, hipEvent_t evStart, hipEvent_t evStop,
    const ParamRecord *nextRecs, const int32_t *nextMap, const float *nextRows);
hipError_t LaunchVoiceWaveResident(hipStream_t s, const DeviceLayout &L, const ResidentArgs &args, hipEvent_t evStart, hipEvent_t evStop);
int WaveResidentGroupsPerCu(const DeviceLayout &L);
hipError_t LaunchVoiceWave16Resident(hipStream_t s, const DeviceLayout &L, const ResidentArgs &args, hipEvent_t evStart, hipEvent_t evStop);
int Wave16ResidentGroupsPerCu();

} // namespace oalgpu
