/* oracle/ref_bridge.cpp -- TEST INFRASTRUCTURE (oracle/_ref/liboalbridge.so; never linked into the product).
 *
 * The REFERENCE-SIDE BINDING of liboalgpu.so, compiled against the reference's own headers and objects:
 * what a maintainer of kcat/openal-soft would add, exercised end to end through the
 * reference's real plumbing --
 *
 *   DeviceBase::renderSamples            alc/alu.cpp:2412-2519   (called as is)
 *     -> ProcessContexts                 alc/alu.cpp:2177-2273   (as is: ProcessParamUpdates,
 *          CalcVoiceParams for every voice with pending VoiceProps, the voice loop, the event ring)
 *          -> voice->mix(...)            alc/alu.cpp:2201-2206   <- THE SEAM: Voice::mix is defined HERE
 *     -> Process(AmbiDecPostProcess)     BFormatDec::process, core/bformatdec.cpp:60-95   (as is)
 *     -> Write<float>                    alc/alu.cpp:2335-2408   (as is)
 *
 * Voice::mix (core/voice.cpp:988) is compiled under the name Voice::mix_cpu by ref_voice_cpu.cpp; the
 * definition below routes every call of the voice loop
 *   mode CPU      -> the reference's own Voice::mix (the pure CPU run: the baseline of the comparison);
 *   mode ADAPTERS -> the reference's own Voice::mix running on top of adapters with the reference's
 *                    function-pointer signatures -- ResamplerFunc (core/mixer/defs.h:71), MixerOutFunc
 *                    (core/mixer.h:22-27), HrtfMixerFunc / HrtfMixerBlendFunc (core/voice.cpp:73-81) --
 *                    that call the per-call C-ABI (oalgpu_resample / oalgpu_mix / oalgpu_mix_hrtf[_blend]);
 *   mode BATCH    -> nothing per voice: the voices of the update are collected and, with the last one,
 *                    described to the GPU context as INTEGRATION.md section 3 says (oalgpu_voice_params
 *                    filled from the Voice AFTER the reference's CalcVoiceParams computed mStep, the pan
 *                    gains and the filter targets), mixed by ONE oalgpu_mix_update, and the dry lines are
 *                    added into DeviceBase::MixBuffer; the reference carries on with its post-process.
 * Two devices: BASELINE configs[0] -- mono sources, linear resampler, a stereo device (3 first-order 2D ambisonic
 * dry lines decoded to 2 speakers by the reference's BFormatDec with panning.cpp's StereoConfig) -- and BASELINE
 * configs[2]'s: a RenderMode::Hrtf device set up the way aluInitRenderer / InitHrtfPanning do (alc/panning.cpp:
 * 1327-1354, :847-1138: GetLoadedHrtf, first-order full-HRTF rendering, DirectHrtfState::build with the cube layout,
 * 700 Hz crossover, HrtfPostProcess), with auxiliary sends into effect slots that carry the reference's own
 * ReverbState (alc/effects/reverb.cpp, reached through its factory).
 */
#include "config.h"
#include "config_simd.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <numbers>
#include <span>
#include "alnumeric.h"
#include "opthelpers.h"
#include <cstddef>
#include <memory>
#include <variant>
#include <vector>
#include "core/bufferline.h"
#include "core/devformat.h"
#include "core/filters/splitter.h"
#define private public
#define protected public
#define class struct
#include "core/filters/biquad.h"
#include "core/filters/nfc.h"           /* (NfcFilter's sections: the binding recovers w0 from them, oalgpu_openal::NfcW0) */
#include "core/bformatdec.h"            /* (BFormatDec::mChannelDec: the pipelined mode hands the speaker decode to the device context) */
#undef class
#undef private
#undef protected

#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <vector>

#include "core/ambidefs.h"
#include "core/bformatdec.h"
#include "core/bs2b.h"
#include "core/encoderbase.hpp"
#include "core/front_stablizer.h"
#include "core/mastering.h"
#include "core/buffer_storage.h"
#include "core/context.h"
#include "core/cpu_caps.h"
#include "core/device.h"
#include "core/effectslot.h"
#include "core/fpu_ctrl.h"
#include "core/hrtf.h"
#include "core/mixer.h"
#include "core/mixer/defs.h"
#include "core/mixer/hrtfdefs.h"
#include "core/voice.h"
#include "core/voice_change.h"
#include "core/async_event.h"
#include "ringbuffer.h"
#include "alc/alu.h"
#include "alc/effects/base.h"
#include "core/effects/base.h"
#include "oalref.h"                        /* oal_reverb_props: the C view of ReverbProps the tests fill */

#include "../include/oalgpu.h"
#include "../include/oalgpu_openal.hpp"    /* THE PRODUCT'S host adapter: this file only drives it */
#include "../include/oalgpu_openal_hooks.hpp" /* ... and defines its two hooks, which _ref/alu_hooked.cpp (alc/alu.cpp + four lines) calls */

extern "C" {
void oalbridge_voice_mix_cpu(void *voice, int vstate, void *context, long long device_ns, unsigned samples_to_do);
void oalbridge_get_hrtf_mixers(void **mixfn, void **blendfn);
void oalbridge_set_hrtf_mixers(void *mixfn, void *blendfn);
}

namespace {

enum Mode { ModeCpu = 0, ModeAdapters = 1, ModeBatch = 2 };

struct Dev final : DeviceBase { Dev() : DeviceBase{DeviceType::Loopback} { } };
struct Ctx final : ContextBase { explicit Ctx(DeviceBase *d) : ContextBase{d} { } };
struct Item final : VoiceBufferItem { };
struct BufferData { std::vector<float> samples; std::vector<int16_t> samples16; Item item; unsigned channels{1}; };
struct SlotData { EffectSlotBase slot; };

bool gInit = false;
void EnsureInit()
{
    if(gInit) return;
    gInit = true;
    if(auto info = GetCPUInfo()) CPUCapFlags = info->mCaps;
    aluInit({}, 1.0f);
    Voice::InitMixer(std::nullopt);
}

} // namespace

struct oalbridge {
    Mode mode{ModeCpu};
    int mathMode{OALGPU_MATH_FAST};
    std::unique_ptr<Dev> dev;
    std::unique_ptr<Ctx> ctx;
    std::deque<BufferData> buffers;
    std::deque<SlotData> slots;             /* the context's effect slots (al/auxeffectslot.cpp's ALeffectslot, core part) */
    std::vector<Voice*> sources;            /* the context's voices, in creation order */
    unsigned nextSourceId{100000u};         /* ids of restarted sources (oalbridge_restart_source) */
    /* ---- the GPU side: include/oalgpu_openal.hpp's batched voice loop */
    std::unique_ptr<oalgpu_openal::BatchMixer> batch;
    int error{0};
    std::string errorText;
    bool trackChanges{false};               /* the batch mixer is told which voices CalcSourceParams recomputes (its optional hook) */
    bool hookAlu{false};                    /* ... by alc/alu.cpp itself (oalgpu_hook::ParamsChanged), and CalcPanningAndFilters hands over
                                             * directions instead of blended responses (oalgpu_hook::GetCoeffs) */
    unsigned long long hookedDirs{0};       /* directions the getCoeffs hook has taken so far */
    /* test aid for the output stage (ApplyDither / Write<T>, alu.cpp:2309-2408, are file-local): lines added
     * to DeviceBase::RealOut by the voice loop's first voice, so that the reference's own output stage
     * converts a known signal */
    std::vector<float> inject;
};

namespace {

oalbridge *gActive = nullptr;               /* the bridge whose renderSamples is running (one mixer thread) */

} // namespace

/* ---- the binding's hooks inside alc/alu.cpp (include/oalgpu_openal_hooks.hpp; the library is built with oracle/_ref/
 * alu_hooked.cpp = alc/alu.cpp with the four lines of INTEGRATION.md 3a, generated by oracle/Makefile) ---------------- */
namespace oalgpu_hook {
void ParamsChanged(Voice *voice) noexcept
{
    oalbridge *b = gActive;
    if(b && b->mode == ModeBatch && b->hookAlu) b->batch->noteParamsChangedInOrder(voice);
}
void GetCoeffs(const HrtfStore &store, Voice *voice, float elevation, float azimuth, float distance, float spread, HrirSpan coeffs,
    std::span<unsigned, 2> delays) noexcept
{
    oalbridge *b = gActive;
    if(b && b->mode == ModeBatch && b->hookAlu)
    {   /* which of the voice's channels: the one whose target the call site passed */
        for(size_t c{0}; c < voice->mChans.size(); ++c)
            if(voice->mChans[c].mDryParams.Hrtf.Target.Coeffs.data() == coeffs.data())
            {
                if(b->batch->noteHrtfDirection(voice, c, elevation, azimuth, distance, spread)) { ++b->hookedDirs; return; }
                break;
            }
    }
    store.getCoeffs(elevation, azimuth, distance, spread, coeffs, delays);
}
} // namespace oalgpu_hook

/* ---- THE SEAM: the symbol ProcessContexts calls for every playing voice (alc/alu.cpp:2201-2206) ---------------- */
void Voice::mix(State const vstate, ContextBase *const context, std::chrono::nanoseconds const deviceTime,
    unsigned const samplesToDo) noexcept
{
    oalbridge *b = gActive;
    if(b && !b->inject.empty())
    {
        auto &real = b->dev->RealOut.Buffer;
        const size_t nl = std::min(real.size(), b->inject.size() / BufferLineSize);
        for(size_t c{0}; c < nl; ++c)
            for(size_t i{0}; i < samplesToDo; ++i) real[c][i] += b->inject[c*BufferLineSize + i];
        return;
    }
    if(!b || b->mode == ModeCpu || b->error)
    {
        oalbridge_voice_mix_cpu(this, int(vstate), context, deviceTime.count(), samplesToDo);
        return;
    }
    if(b->mode == ModeAdapters)
    {
        /* per voice: CalcVoiceParams chose mResampler with PrepareResampler (alu.cpp:1686, :2000); the
         * adapter of the same ResamplerFunc type takes its place for this call */
        oalgpu_openal::Adapters().resamplerKind = int(mProps.mResampler);
        auto const saved = mResampler;
        mResampler = oalgpu_openal::Resample;
        oalbridge_voice_mix_cpu(this, int(vstate), context, deviceTime.count(), samplesToDo);
        mResampler = saved;
        return;
    }
    /* BATCH: oalgpu_openal::BatchMixer collects the update's voices and mixes them with the last one */
    if(!b->batch->mix(this, vstate, context, *b->dev, deviceTime, samplesToDo) && b->batch->error() && b->batch->batchComplete())
    {   /* "on error the caller runs the CPU loop for that update"  */
        if(!b->error) { b->error = b->batch->error(); b->errorText = b->batch->errorText(); }
        for(auto &[voice, st] : b->batch->batch())
            oalbridge_voice_mix_cpu(voice, int(st), context, deviceTime.count(), samplesToDo);
    }
}

extern "C" {

/* the cube of virtual speakers InitHrtfPanning decodes first-order ambisonics with, its decoder matrix and
 * per-order HF gains (alc/panning.cpp:861-870, :941-950, :1021-1023: function-local tables there) */
static constexpr float kDeg35 = 6.154797087e-01f, kDeg45 = std::numbers::pi_v<float> / 4.0f, kDeg135 = kDeg45 * 3.0f;
static constexpr std::array kAmbiPoints1O{
    AngularPoint{EvRadians{ kDeg35}, AzRadians{-kDeg45}}, AngularPoint{EvRadians{ kDeg35}, AzRadians{-kDeg135}},
    AngularPoint{EvRadians{ kDeg35}, AzRadians{ kDeg45}}, AngularPoint{EvRadians{ kDeg35}, AzRadians{ kDeg135}},
    AngularPoint{EvRadians{-kDeg35}, AzRadians{-kDeg45}}, AngularPoint{EvRadians{-kDeg35}, AzRadians{-kDeg135}},
    AngularPoint{EvRadians{-kDeg35}, AzRadians{ kDeg45}}, AngularPoint{EvRadians{-kDeg35}, AzRadians{ kDeg135}},
};
static constexpr std::array<float, MaxAmbiOrder+1> kAmbiOrderHFGain1O{2.000000000e+00f, 1.154700538e+00f};

/* mode: 0 CPU, 1 ADAPTERS, 2 BATCH; math_mode: oalgpu_math_mode of the GPU side; hrtf: a RenderMode::Hrtf device on the
 * first .mhr under mhr_dir (else the stereo device); num_sends: DeviceBase::NumAuxSends */
oalbridge *oalbridge_create_ex(int mode, uint32_t sample_rate, int math_mode, int hrtf, const char *mhr_dir, uint32_t num_sends)
{
    EnsureInit();
    auto b = std::make_unique<oalbridge>();
    b->mode = static_cast<Mode>(mode);
    b->mathMode = math_mode;
    b->batch = std::make_unique<oalgpu_openal::BatchMixer>(math_mode, 0);
    b->dev = std::make_unique<Dev>();
    auto &dev = *b->dev;
    dev.mSampleRate = sample_rate;
    dev.mUpdateSize = BufferLineSize;
    dev.mBufferSize = BufferLineSize;
    dev.FmtChans = DevFmtStereo;
    dev.FmtType = DevFmtFloat;
    dev.NumAuxSends = num_sends;
    dev.AvgSpeakerDist = 0.0f;
    if(hrtf == 2)
    {
        /* the same stereo device mixing SECOND-order 2D ambisonics (W, Y, X, V, U): what "ambi-order = 2" leaves -- a first-order
         * B-Format source is then VoiceFlag::IsAmbisonic (Voice::prepare, core/voice.cpp:1353-1380: the device's order exceeds the
         * source's) and, with a control distance, near-field compensated per order (DeviceBase::NumChannelsPerOrder = 1, 2, 2) */
        dev.mAmbiOrder = 2;
        dev.m2DMixing = true;
        dev.mRenderMode = RenderMode::Normal;
        dev.mXOverFreq = 400.0f;
        constexpr size_t ambicount = 5, realcount = 2;
        dev.MixBuffer.resize(ambicount + realcount);
        dev.Dry.Buffer = std::span{dev.MixBuffer}.first(ambicount);
        dev.RealOut.Buffer = std::span{dev.MixBuffer}.subspan(ambicount);
        for(size_t i{0}; i < ambicount; ++i)
            dev.Dry.AmbiMap[i] = BFChannelConfig{1.0f, AmbiIndex::FromACN2D[i].c_val};
        dev.RealOut.ChannelIndex[FrontLeft] = 0_u8;
        dev.RealOut.ChannelIndex[FrontRight] = 1_u8;
        dev.NumChannelsPerOrder = {1u, 2u, 2u, 0u, 0u};
        auto coeffs = std::vector<ChannelDec>(2);
        coeffs[0] = ChannelDec{}; coeffs[1] = ChannelDec{};
        coeffs[0][0] = 5.00000000e-1f; coeffs[0][1] =  2.88675135e-1f; coeffs[0][2] = 5.52305643e-2f; coeffs[0][3] =  3.1e-2f; coeffs[0][4] = -1.7e-2f;
        coeffs[1][0] = 5.00000000e-1f; coeffs[1][1] = -2.88675135e-1f; coeffs[1][2] = 5.52305643e-2f; coeffs[1][3] = -3.1e-2f; coeffs[1][4] = -1.7e-2f;
        auto dec = std::make_unique<BFormatDec>(ambicount, coeffs, std::span<const ChannelDec>{},
            dev.mXOverFreq / float(sample_rate));
        dev.mPostProcess.emplace<AmbiDecPostProcess>(AmbiDecPostProcess{std::move(dec)});
    }
    else if(!hrtf)
    {
        /* a stereo loopback device as alc/alc.cpp + alc/panning.cpp set it up (InitPanning with
         * StereoConfig, panning.cpp:548-556, :719-850): 3 first-order 2D ambisonic dry lines (W, Y, X)
         * decoded by a single-band BFormatDec to FrontLeft / FrontRight */
        dev.mAmbiOrder = 1;
        dev.m2DMixing = true;
        dev.mRenderMode = RenderMode::Normal;
        dev.mXOverFreq = 400.0f;
        constexpr size_t ambicount = 3, realcount = 2;
        dev.MixBuffer.resize(ambicount + realcount);
        dev.Dry.Buffer = std::span{dev.MixBuffer}.first(ambicount);
        dev.RealOut.Buffer = std::span{dev.MixBuffer}.subspan(ambicount);
        for(size_t i{0}; i < ambicount; ++i)
            dev.Dry.AmbiMap[i] = BFChannelConfig{1.0f, AmbiIndex::FromACN2D[i].c_val};
        dev.RealOut.ChannelIndex[FrontLeft] = 0_u8;
        dev.RealOut.ChannelIndex[FrontRight] = 1_u8;
        dev.NumChannelsPerOrder = {1u, 2u, 0u, 0u, 0u};
        auto coeffs = std::vector<ChannelDec>(2);
        coeffs[0] = ChannelDec{}; coeffs[1] = ChannelDec{};
        coeffs[0][0] = 5.00000000e-1f; coeffs[0][1] =  2.88675135e-1f; coeffs[0][2] = 5.52305643e-2f;
        coeffs[1][0] = 5.00000000e-1f; coeffs[1][1] = -2.88675135e-1f; coeffs[1][2] = 5.52305643e-2f;
        auto dec = std::make_unique<BFormatDec>(ambicount, coeffs, std::span<const ChannelDec>{},
            dev.mXOverFreq / float(sample_rate));
        dev.mPostProcess.emplace<AmbiDecPostProcess>(AmbiDecPostProcess{std::move(dec)});
    }
    else
    {
        /* aluInitRenderer's HRTF branch (alc/panning.cpp:1327-1354): the data set through GetLoadedHrtf (found the way the
         * reference finds one: EnumerateHrtf over ALSOFT_LOCAL_PATH), mIrSize, then InitHrtfPanning (:847-1138) for the
         * default "full" mode -- first order, RenderMode::Hrtf, 700 Hz crossover, ACN dry lines, stereo real output,
         * DirectHrtfState::build on the cube -- and the HrtfPostProcess */
        if(!mhr_dir) return nullptr;
        setenv("ALSOFT_LOCAL_PATH", mhr_dir, 1);
        try {
            auto const names = EnumerateHrtf(std::nullopt);
            if(names.empty()) return nullptr;
            dev.mHrtf = GetLoadedHrtf(names[0], sample_rate);
        }
        catch(...) { return nullptr; }
        if(!dev.mHrtf) return nullptr;
        dev.mIrSize = dev.mHrtf->mIrSize;
        dev.mXOverFreq = 700.0f;
        dev.mRenderMode = RenderMode::Hrtf;
        dev.mAmbiOrder = 1;
        dev.m2DMixing = false;
        constexpr size_t count = 4, realcount = 2;              /* AmbiChannelsFromOrder(1); channelsFromFmt() of DevFmtStereo */
        for(size_t i{0}; i < count; ++i)
            dev.Dry.AmbiMap[i] = BFChannelConfig{1.0f, AmbiIndex::FromACN[i].c_val};
        dev.MixBuffer.resize(count + realcount);                /* AllocChannels, panning.cpp:174-195 */
        dev.Dry.Buffer = std::span{dev.MixBuffer}.first(count);
        dev.RealOut.Buffer = std::span{dev.MixBuffer}.subspan(count);
        dev.RealOut.ChannelIndex[FrontLeft] = 0_u8;
        dev.RealOut.ChannelIndex[FrontRight] = 1_u8;
        auto matrix = std::vector<std::array<float, MaxAmbiChannels>>(8);
        for(size_t i{0}; i < 8; ++i)
        {   /* AmbiMatrix1O: +-1/8 by the sign of the point's y (left), z (up), x (front) -- ACN W, Y, Z, X */
            matrix[i].fill(0.0f);
            matrix[i][0] = 0.125f;
            matrix[i][1] = (i & 2) ? -0.125f : 0.125f;
            matrix[i][2] = (i & 4) ? -0.125f : 0.125f;
            matrix[i][3] = (i & 1) ? -0.125f : 0.125f;
        }
        auto hrtfstate = DirectHrtfState::Create(count);
        hrtfstate->build(dev.mHrtf.get(), dev.mIrSize, false, kAmbiPoints1O, matrix, dev.mXOverFreq, kAmbiOrderHFGain1O);
        dev.mPostProcess.emplace<HrtfPostProcess>(HrtfPostProcess{std::move(hrtfstate)});
    }

    /* the context, registered with the device as alc/context.cpp:219-273 does */
    b->ctx = std::make_unique<Ctx>(b->dev.get());
    auto &ctx = *b->ctx;
    ctx.mEnabledEvts.store({}, std::memory_order_relaxed);
    ctx.mAsyncEvents = FifoBuffer<AsyncEvent>::Create(1024, false);
    ctx.allocVoiceChanges();
    {
        VoiceChange *cur{ctx.mVoiceChangeTail};
        while(VoiceChange *next{cur->mNext.load(std::memory_order_relaxed)}) cur = next;
        ctx.mCurrentVoiceChange.store(cur, std::memory_order_relaxed);
    }
    ctx.mActiveAuxSlots.store(ContextBase::EffectSlotArray::Create(0).release(), std::memory_order_relaxed);
    ctx.mVoices.store(ContextBase::VoiceArray::Create(0).release(), std::memory_order_relaxed);
    ctx.mActiveVoiceCount.store(0, std::memory_order_relaxed);
    ctx.mParams.mDistanceModel = DistanceModel::InverseClamped;
    {
        auto arr = DeviceBase::ContextArray::Create(1);
        (*arr)[0] = b->ctx.get();
        dev.mContexts.store(arr.release(), std::memory_order_release);
    }
    if(mode == ModeAdapters)
    {   /* installed the way Voice::InitMixer installs the CPU variants (core/voice.cpp:139-193) */
        oalgpu_openal::Adapters().mathMode = math_mode;
        MixSamplesOut = oalgpu_openal::Mix;
        oalbridge_set_hrtf_mixers(reinterpret_cast<void*>(oalgpu_openal::MixHrtf), reinterpret_cast<void*>(oalgpu_openal::MixHrtfBlend));
    }
    return b.release();
}

oalbridge *oalbridge_create(int mode, uint32_t sample_rate, int math_mode)
{ return oalbridge_create_ex(mode, sample_rate, math_mode, 0, nullptr, 0); }

/* the second-order 2D device (above); control_distance > 0: with near-field control, as InitNearFieldCtrl leaves it
 * (alc/panning.cpp:285-299) -- before any source exists: Voice::prepare copies DeviceBase::mNFCtrlFilter */
oalbridge *oalbridge_create_ambi2(int mode, uint32_t sample_rate, int math_mode, uint32_t num_sends, float control_distance)
{
    oalbridge *b = oalbridge_create_ex(mode, sample_rate, math_mode, 2, nullptr, num_sends);
    if(b && control_distance > 0.0f)
    {
        b->dev->AvgSpeakerDist = std::clamp(control_distance, 0.1f, 10.0f);
        b->dev->mNFCtrlFilter.init(SpeedOfSoundMetersPerSec / b->dev->AvgSpeakerDist / float(sample_rate));
    }
    return b;
}

/* An effect slot of the context carrying the reference's own EAX reverb: what alGenAuxiliaryEffectSlots + alAuxiliaryEffectSloti
 * (al/auxeffectslot.cpp) leave in the core part -- a first-order wet bus, the ReverbState from its factory after
 * deviceUpdate and update -- registered in ContextBase::mActiveAuxSlots (first half: the slots, second half: the
 * sorted copy ProcessContexts fills, alu.cpp:2186-2188).  Returns the slot's index. */
int oalbridge_add_reverb_slot(oalbridge *b, const oal_reverb_props *p, float gain)
{
    auto &sd = b->slots.emplace_back();
    auto &slot = sd.slot;
    auto &dev = *b->dev;
    slot.InUse = true;
    slot.mWetBuffer.resize(4);
    slot.Wet.Buffer = slot.mWetBuffer;
    for(uint32_t i{0}; i < 4; ++i) slot.Wet.AmbiMap[i] = BFChannelConfig{1.0f, AmbiIndex::FromACN[i].c_val};
    auto rp = ReverbProps{};
    rp.Density = p->density; rp.Diffusion = p->diffusion; rp.Gain = p->gain; rp.GainHF = p->gain_hf;
    rp.GainLF = p->gain_lf; rp.DecayTime = p->decay_time; rp.DecayHFRatio = p->decay_hf_ratio;
    rp.DecayLFRatio = p->decay_lf_ratio; rp.ReflectionsGain = p->reflections_gain;
    rp.ReflectionsDelay = p->reflections_delay;
    rp.ReflectionsPan = {p->reflections_pan[0], p->reflections_pan[1], p->reflections_pan[2]};
    rp.LateReverbGain = p->late_reverb_gain; rp.LateReverbDelay = p->late_reverb_delay;
    rp.LateReverbPan = {p->late_reverb_pan[0], p->late_reverb_pan[1], p->late_reverb_pan[2]};
    rp.EchoTime = p->echo_time; rp.EchoDepth = p->echo_depth; rp.ModulationTime = p->modulation_time;
    rp.ModulationDepth = p->modulation_depth; rp.AirAbsorptionGainHF = p->air_absorption_gain_hf;
    rp.HFReference = p->hf_reference; rp.LFReference = p->lf_reference;
    rp.RoomRolloffFactor = p->room_rolloff_factor; rp.DecayHFLimit = p->decay_hf_limit != 0;
    /* CalcEffectSlotParams (alu.cpp:565-640) */
    slot.Gain = gain;
    slot.AuxSendAuto = true;
    slot.Target = nullptr;
    slot.EffectType = EffectSlotType::Reverb;
    slot.mEffectProps = rp;
    slot.RoomRolloff = rp.RoomRolloffFactor;
    slot.AirAbsorptionGainHF = rp.AirAbsorptionGainHF;
    slot.DecayTime = rp.DecayTime; slot.DecayLFRatio = rp.DecayLFRatio; slot.DecayHFRatio = rp.DecayHFRatio;
    slot.DecayHFLimit = rp.DecayHFLimit;
    slot.mEffectState = ReverbStateFactory_getFactory()->create();
    slot.mEffectState->deviceUpdate(&dev, nullptr);
    slot.mEffectState->update(b->ctx.get(), &slot, &slot.mEffectProps, EffectTarget{&dev.Dry, &dev.RealOut});
    /* the active-slot array, twice the slots long */
    const size_t n = b->slots.size();
    auto arr = ContextBase::EffectSlotArray::Create(n * 2);
    for(size_t i{0}; i < n; ++i) { (*arr)[i] = &b->slots[i].slot; (*arr)[n + i] = nullptr; }
    b->ctx->mActiveAuxSlots.store(std::move(arr), std::memory_order_release);
    return int(n - 1);
}

void oalbridge_destroy(oalbridge *b)
{
    if(!b) return;
    if(gActive == b) gActive = nullptr;
    if(b->mode == ModeAdapters) Voice::InitMixer(std::nullopt);      /* the reference's own kernels again */
    delete b;
}

int oalbridge_add_buffer(oalbridge *b, const float *data, uint32_t frames, uint32_t loop_start, uint32_t loop_end)
{
    auto &buf = b->buffers.emplace_back();
    buf.samples.assign(data, data + frames);
    buf.samples.resize(frames + 4);
    buf.item.mSamples = std::span<f32>{reinterpret_cast<f32*>(buf.samples.data()), frames};
    buf.item.mBlockAlign = 1;
    buf.item.mSampleLen = frames;
    buf.item.mLoopStart = loop_start;
    buf.item.mLoopEnd = loop_end;
    return int(b->buffers.size() - 1);
}

/* the same with 16-bit samples (FmtShort: SampleInfo<i16>::to_float = s / 32768, core/fmt_traits.h) */
int oalbridge_add_buffer_i16(oalbridge *b, const int16_t *data, uint32_t frames, uint32_t loop_start, uint32_t loop_end)
{
    auto &buf = b->buffers.emplace_back();
    buf.samples16.assign(data, data + frames);
    buf.samples16.resize(frames + 8);
    buf.item.mSamples = std::span<i16>{reinterpret_cast<i16*>(buf.samples16.data()), frames};
    buf.item.mBlockAlign = 1;
    buf.item.mSampleLen = frames;
    buf.item.mLoopStart = loop_start;
    buf.item.mLoopEnd = loop_end;
    return int(b->buffers.size() - 1);
}

static VoicePropsItem *NewProps(oalbridge *b)
{   /* al/source.cpp UpdateSourceProps: an item off the context's free list */
    auto &ctx = *b->ctx;
    auto *props = ctx.mFreeVoiceProps.load(std::memory_order_acquire);
    if(!props) { ctx.allocVoiceProps(); props = ctx.mFreeVoiceProps.load(std::memory_order_acquire); }
    VoicePropsItem *next;
    do { next = props->next.load(std::memory_order_relaxed); }
    while(!ctx.mFreeVoiceProps.compare_exchange_weak(props, next, std::memory_order_acq_rel, std::memory_order_acquire));
    return props;
}

static void FillProps(oalbridge *b, VoiceProps &p, float gain, float x, float y, float z, int resampler, float pitch, float gain_hf,
    int send_slot = -1, float send_gain = 1.0f, float send_gain_hf = 1.0f)
{   /* the defaults of a new AL source (al/source.cpp: ALsource::ALsource) with the fields the test moves */
    p = VoiceProps{};
    p.Pitch = pitch; p.Gain = gain; p.OuterGain = 0.0f; p.MinGain = 0.0f; p.MaxGain = 1.0f;
    p.InnerAngle = 360.0f; p.OuterAngle = 360.0f;
    p.RefDistance = 1.0f; p.MaxDistance = std::numeric_limits<float>::max(); p.RolloffFactor = 1.0f;
    p.Position = {x, y, z}; p.Velocity = {0.0f, 0.0f, 0.0f}; p.Direction = {0.0f, 0.0f, 0.0f};
    p.OrientAt = {0.0f, 0.0f, -1.0f}; p.OrientUp = {0.0f, 1.0f, 0.0f};
    p.HeadRelative = false;
    p.mDistanceModel = DistanceModel::InverseClamped;
    p.mResampler = static_cast<Resampler>(resampler);
    p.DirectChannels = DirectMode::Off;
    p.mSpatializeMode = SpatializeMode::Auto;
    p.mPanningEnabled = false;
    p.DryGainHFAuto = true; p.WetGainAuto = true; p.WetGainHFAuto = true; p.OuterGainHF = 1.0f;
    p.AirAbsorptionFactor = 0.0f; p.RoomRolloffFactor = 0.0f; p.DopplerFactor = 1.0f;
    p.StereoPan = {std::numbers::pi_v<float>/6.0f, -std::numbers::pi_v<float>/6.0f};
    p.Radius = 0.0f; p.EnhWidth = 0.593f; p.Panning = 0.0f;
    p.Direct = {1.0f, gain_hf, 5000.0f, 1.0f, 250.0f};
    for(auto &s : p.Send) s = {nullptr, 1.0f, 1.0f, 5000.0f, 1.0f, 250.0f};
    /* alSource3i(AL_AUXILIARY_SEND_FILTER): send 0 into the slot, with the send filter's gains */
    if(send_slot >= 0 && size_t(send_slot) < b->slots.size())
        p.Send[0] = {&b->slots[size_t(send_slot)].slot, send_gain, send_gain_hf, 5000.0f, 1.0f, 250.0f};
}

/* a playing mono source: a voice of the context with pending VoiceProps, as al/source.cpp leaves it */
int oalbridge_add_source_ex(oalbridge *b, int buffer, int looping, int position, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf, int send_slot, float send_gain, float send_gain_hf)
{
    auto &ctx = *b->ctx;
    auto &buf = b->buffers.at(size_t(buffer));
    const size_t n = ctx.mActiveVoiceCount.load(std::memory_order_relaxed);
    if(n >= ctx.mVoices.load(std::memory_order_relaxed)->size()) ctx.allocVoices(64);
    Voice *v = (*ctx.mVoices.load(std::memory_order_relaxed))[n];
    /* InitVoice, al/source.cpp:639-670 */
    v->mLoopBuffer.store(looping ? &buf.item : nullptr, std::memory_order_relaxed);
    v->mFmtChannels = FmtMono;
    v->mFrequency = 44100;
    v->mFrameStep = 1;
    v->mBytesPerBlock = std::holds_alternative<std::span<i16>>(buf.item.mSamples) ? 2u : 4u;
    v->mSamplesPerBlock = 1;
    v->mAmbiOrder = 0;
    v->mFlags.reset();
    v->mFlags.set(VoiceFlag::IsStatic);
    v->mNumCallbackBlocks = 0;
    v->mCallbackBlockOffset = 0;
    v->prepare(b->dev.get());
    v->mPosition.store(position, std::memory_order_relaxed);
    v->mPositionFrac.store(0u, std::memory_order_relaxed);
    v->mCurrentBuffer.store(&buf.item, std::memory_order_relaxed);
    v->mStartTime = {};
    v->mSourceID.store(unsigned(n + 1), std::memory_order_relaxed);
    auto *props = NewProps(b);
    FillProps(b, *props, gain, x, y, z, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf);
    v->mUpdate.store(props, std::memory_order_release);
    v->mPlayState.store(Voice::Playing, std::memory_order_release);
    ctx.mActiveVoiceCount.store(n + 1, std::memory_order_release);
    b->sources.push_back(v);
    return int(n);
}

/* The pooled Voice object of a source that has stopped starts over as ANOTHER source (what alSourcePlay does with a
 * free voice of the context's clusters, al/source.cpp:3046-3121): InitVoice again, a new source id, fresh properties. */
int oalbridge_restart_source(oalbridge *b, int source, int buffer, int looping, int position, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf, int send_slot, float send_gain, float send_gain_hf)
{
    Voice *v = b->sources.at(size_t(source));
    if(v->mPlayState.load(std::memory_order_acquire) != Voice::Stopped) return -1;
    auto &buf = b->buffers.at(size_t(buffer));
    v->mLoopBuffer.store(looping ? &buf.item : nullptr, std::memory_order_relaxed);
    v->mBytesPerBlock = std::holds_alternative<std::span<i16>>(buf.item.mSamples) ? 2u : 4u;
    v->mFlags.reset();
    v->mFlags.set(VoiceFlag::IsStatic);
    v->prepare(b->dev.get());
    v->mPosition.store(position, std::memory_order_relaxed);
    v->mPositionFrac.store(0u, std::memory_order_relaxed);
    v->mCurrentBuffer.store(&buf.item, std::memory_order_relaxed);
    v->mStartTime = {};
    v->mSourceID.store(++b->nextSourceId, std::memory_order_relaxed);
    auto *props = NewProps(b);
    FillProps(b, *props, gain, x, y, z, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf);
    if(auto *old = v->mUpdate.exchange(props, std::memory_order_acq_rel))
        AtomicReplaceHead(b->ctx->mFreeVoiceProps, old);
    v->mPlayState.store(Voice::Playing, std::memory_order_release);
    return 0;
}

/* ---- the voice kinds beyond mono static sources ------------------------------------------------------------------ */
int oalbridge_add_source(oalbridge *b, int buffer, int looping, int position, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf);
/* interleaved multi-channel data (FmtStereo: two channels, mFrameStep = 2) */
int oalbridge_add_buffer_interleaved(oalbridge *b, const float *data, uint32_t frames, uint32_t channels, uint32_t loop_start, uint32_t loop_end)
{
    auto &buf = b->buffers.emplace_back();
    buf.samples.assign(data, data + size_t{frames} * channels);
    buf.samples.resize(size_t{frames} * channels + 8);
    buf.item.mSamples = std::span<f32>{reinterpret_cast<f32*>(buf.samples.data()), size_t{frames} * channels};
    buf.item.mBlockAlign = 1;
    buf.item.mSampleLen = frames;
    buf.item.mLoopStart = loop_start;
    buf.item.mLoopEnd = loop_end;
    buf.channels = channels;
    return int(b->buffers.size() - 1);
}

/* a playing STEREO static source (InitVoice with a two-channel buffer: Voice::prepare sizes mChans, CalcPanningAndFilters pans
 * the channels apart, alu.cpp:1100-1260) */
int oalbridge_add_source_stereo(oalbridge *b, int buffer, int looping, int position, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf)
{
    auto &ctx = *b->ctx;
    auto &buf = b->buffers.at(size_t(buffer));
    if(buf.channels != 2) return -1;
    const size_t n = ctx.mActiveVoiceCount.load(std::memory_order_relaxed);
    if(n >= ctx.mVoices.load(std::memory_order_relaxed)->size()) ctx.allocVoices(64);
    Voice *v = (*ctx.mVoices.load(std::memory_order_relaxed))[n];
    v->mLoopBuffer.store(looping ? &buf.item : nullptr, std::memory_order_relaxed);
    v->mFmtChannels = FmtStereo;
    v->mFrequency = 44100;
    v->mFrameStep = 2;
    v->mBytesPerBlock = 8u;
    v->mSamplesPerBlock = 1;
    v->mAmbiOrder = 0;
    v->mFlags.reset();
    v->mFlags.set(VoiceFlag::IsStatic);
    v->mNumCallbackBlocks = 0;
    v->mCallbackBlockOffset = 0;
    v->prepare(b->dev.get());
    v->mPosition.store(position, std::memory_order_relaxed);
    v->mPositionFrac.store(0u, std::memory_order_relaxed);
    v->mCurrentBuffer.store(&buf.item, std::memory_order_relaxed);
    v->mStartTime = {};
    v->mSourceID.store(unsigned(n + 1), std::memory_order_relaxed);
    auto *props = NewProps(b);
    FillProps(b, *props, gain, x, y, z, resampler, pitch, gain_hf);
    v->mUpdate.store(props, std::memory_order_release);
    v->mPlayState.store(Voice::Playing, std::memory_order_release);
    ctx.mActiveVoiceCount.store(n + 1, std::memory_order_release);
    b->sources.push_back(v);
    return int(n);
}

/* a playing CALLBACK source (AL_SOFT_callback_buffer: alBufferCallbackSOFT + alSourcei(AL_BUFFER), al/buffer.cpp:455-490,
 * al/source.cpp:1715-1730, :659-662): the item carries the function and a storage of MixerLineSize * MaxPitch + MaxResamplerEdge
 * frames; Voice::mix asks the function for exactly the blocks it needs (voice.cpp:726-752).  The function here produces
 * `total_frames` frames of a signal that depends on nothing but the frame's number and `seed`, then comes up short: the source
 * plays what it has and ends. */
struct CallbackGen { uint64_t produced{0}, total{0}; uint32_t seed{0}; };
static int CallbackGenFn(void *user, void *data, int numBytes) noexcept
{
    auto *g = static_cast<CallbackGen*>(user);
    auto *out = static_cast<float*>(data);
    const uint64_t want = uint64_t(numBytes) / 4u, give = std::min<uint64_t>(want, g->total - g->produced);
    for(uint64_t i{0}; i < give; ++i)
    {
        const uint64_t k = g->produced + i;
        const uint32_t h = uint32_t(k * 2654435761u + g->seed * 40503u);
        out[i] = 0.5f * std::sin(0.013f * float(k % 100000u) * float(1u + g->seed % 5u)) + 0.25f * (float((h >> 16) & 0xffu) / 255.0f - 0.5f);
    }
    g->produced += give;
    return int(give * 4u);
}
static std::deque<CallbackGen> gCallbackGens;

int oalbridge_add_source_callback(oalbridge *b, uint32_t total_frames, uint32_t seed, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf)
{
    auto &ctx = *b->ctx;
    auto &buf = b->buffers.emplace_back();
    auto &gen = gCallbackGens.emplace_back();
    gen.total = total_frames; gen.seed = seed;
    buf.samples.assign(size_t{BufferLineSize} * MaxPitch + MaxResamplerEdge + 8, 0.0f);
    buf.item.mSamples = std::span<f32>{reinterpret_cast<f32*>(buf.samples.data()), size_t{BufferLineSize} * MaxPitch + MaxResamplerEdge};
    buf.item.mCallback = CallbackGenFn;
    buf.item.mUserData = &gen;
    buf.item.mBlockAlign = 1;
    buf.item.mSampleLen = 0;
    buf.item.mLoopStart = 0;
    buf.item.mLoopEnd = 0;
    const size_t n = ctx.mActiveVoiceCount.load(std::memory_order_relaxed);
    if(n >= ctx.mVoices.load(std::memory_order_relaxed)->size()) ctx.allocVoices(64);
    Voice *v = (*ctx.mVoices.load(std::memory_order_relaxed))[n];
    v->mLoopBuffer.store(nullptr, std::memory_order_relaxed);
    v->mFmtChannels = FmtMono;
    v->mFrequency = 44100;
    v->mFrameStep = 1;
    v->mBytesPerBlock = 4u;
    v->mSamplesPerBlock = 1;
    v->mAmbiOrder = 0;
    v->mFlags.reset();
    v->mFlags.set(VoiceFlag::IsCallback);
    v->mNumCallbackBlocks = 0;
    v->mCallbackBlockOffset = 0;
    v->prepare(b->dev.get());
    v->mPosition.store(0, std::memory_order_relaxed);
    v->mPositionFrac.store(0u, std::memory_order_relaxed);
    v->mCurrentBuffer.store(&buf.item, std::memory_order_relaxed);
    v->mStartTime = {};
    v->mSourceID.store(unsigned(n + 1), std::memory_order_relaxed);
    auto *props = NewProps(b);
    FillProps(b, *props, gain, x, y, z, resampler, pitch, gain_hf);
    v->mUpdate.store(props, std::memory_order_release);
    v->mPlayState.store(Voice::Playing, std::memory_order_release);
    ctx.mActiveVoiceCount.store(n + 1, std::memory_order_release);
    b->sources.push_back(v);
    return int(n);
}

/* a playing first-order 2D B-FORMAT static source (three interleaved channels, ACN / SN3D): on the second-order device
 * VoiceFlag::IsAmbisonic -- every channel's samples go through BandSplitter::processScale (voice.cpp:1082-1091) -- and, at the
 * listener's place, its X / Y channels rotated by the source's orientation (alu.cpp:943-1100) */
int oalbridge_add_source_bformat2d(oalbridge *b, int buffer, int looping, int position, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf, int send_slot, float send_gain)
{
    auto &ctx = *b->ctx;
    auto &buf = b->buffers.at(size_t(buffer));
    if(buf.channels != 3) return -1;
    const size_t n = ctx.mActiveVoiceCount.load(std::memory_order_relaxed);
    if(n >= ctx.mVoices.load(std::memory_order_relaxed)->size()) ctx.allocVoices(64);
    Voice *v = (*ctx.mVoices.load(std::memory_order_relaxed))[n];
    v->mLoopBuffer.store(looping ? &buf.item : nullptr, std::memory_order_relaxed);
    v->mFmtChannels = FmtBFormat2D;
    v->mFrequency = 44100;
    v->mFrameStep = 3;
    v->mBytesPerBlock = 12u;
    v->mSamplesPerBlock = 1;
    v->mAmbiLayout = AmbiLayout::ACN;
    v->mAmbiScaling = AmbiScaling::SN3D;
    v->mAmbiOrder = 1;
    v->mFlags.reset();
    v->mFlags.set(VoiceFlag::IsStatic);
    v->mNumCallbackBlocks = 0;
    v->mCallbackBlockOffset = 0;
    v->prepare(b->dev.get());
    v->mPosition.store(position, std::memory_order_relaxed);
    v->mPositionFrac.store(0u, std::memory_order_relaxed);
    v->mCurrentBuffer.store(&buf.item, std::memory_order_relaxed);
    v->mStartTime = {};
    v->mSourceID.store(unsigned(n + 1), std::memory_order_relaxed);
    auto *props = NewProps(b);
    FillProps(b, *props, gain, x, y, z, resampler, pitch, gain_hf, send_slot, send_gain, 1.0f);
    props->OrientAt = {0.6f, 0.0f, -0.8f};          /* (turned a little: the rotation matrix is not the identity) */
    v->mUpdate.store(props, std::memory_order_release);
    v->mPlayState.store(Voice::Playing, std::memory_order_release);
    ctx.mActiveVoiceCount.store(n + 1, std::memory_order_release);
    b->sources.push_back(v);
    return int(n);
}

/* a playing STREAMING source on a queue of mono buffers (alSourceQueueBuffers + alSourcePlay: the items linked through
 * mNext, the voice not VoiceFlag::IsStatic, mLoopBuffer = the queue's head when looping, al/source.cpp:639-670) */
int oalbridge_add_source_queue(oalbridge *b, const int *buffers, uint32_t count, int looping, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf)
{
    if(count == 0) return -1;
    for(uint32_t i{0}; i < count; ++i)
        b->buffers.at(size_t(buffers[i])).item.mNext.store(i + 1 < count ? &b->buffers.at(size_t(buffers[i + 1])).item : nullptr, std::memory_order_relaxed);
    const int src = oalbridge_add_source(b, buffers[0], 0, 0, gain, x, y, z, resampler, pitch, gain_hf);
    Voice *v = b->sources.at(size_t(src));
    v->mFlags.reset(VoiceFlag::IsStatic);
    v->mLoopBuffer.store(looping ? &b->buffers.at(size_t(buffers[0])).item : nullptr, std::memory_order_relaxed);
    return src;
}

/* alSourceQueueBuffers on a playing source: one more item behind the queue's last */
int oalbridge_queue_buffer(oalbridge *b, int last_buffer, int buffer)
{
    b->buffers.at(size_t(buffer)).item.mNext.store(nullptr, std::memory_order_relaxed);
    b->buffers.at(size_t(last_buffer)).item.mNext.store(&b->buffers.at(size_t(buffer)).item, std::memory_order_release);
    return 0;
}

/* which buffer of the bridge the source's mCurrentBuffer is (-1: none) */
int oalbridge_source_buffer(oalbridge *b, int source)
{
    auto *cur = b->sources.at(size_t(source))->mCurrentBuffer.load(std::memory_order_relaxed);
    for(size_t i{0}; i < b->buffers.size(); ++i) if(&b->buffers[i].item == cur) return int(i);
    return -1;
}

/* Voice::mStartTime: the source starts `delay_samples` output samples from now (alSourcePlayAtTimeSOFT; DeviceBase::getClockTime,
 * core/device.h:375-384, is what ProcessContexts hands to Voice::mix, alc/alu.cpp:2182) */
int oalbridge_set_start_delay(oalbridge *b, int source, uint32_t delay_samples)
{
    auto &dev = *b->dev;
    using namespace std::chrono;
    auto const now = dev.getClockTime();
    b->sources.at(size_t(source))->mStartTime = now + nanoseconds{seconds{delay_samples}} / dev.mSampleRate;
    return 0;
}

/* alBufferData on an existing buffer: new samples in the SAME storage (same item, same address, same length) -- what a buffer
 * deleted and another allocated in its place looks like to the mixer; with forget the batch mixer is told (its forgetBuffer hook,
 * which a maintainer calls from alDeleteBuffers / alBufferData) */
int oalbridge_replace_buffer(oalbridge *b, int buffer, const float *data, uint32_t frames, int forget)
{
    auto &buf = b->buffers.at(size_t(buffer));
    if(buf.samples.size() < size_t{frames} + 4) return -1;         /* (the storage stays where it is) */
    std::copy(data, data + frames, buf.samples.begin());
    buf.item.mSamples = std::span<f32>{reinterpret_cast<f32*>(buf.samples.data()), frames};
    buf.item.mSampleLen = frames;
    buf.item.mLoopStart = 0; buf.item.mLoopEnd = frames;
    if(forget) b->batch->forgetBuffer(&buf.item);
    return 0;
}

int oalbridge_batch_live_buffers(oalbridge *b) { return int(b->batch->liveBuffers()); }

/* the batch mixer's CalcSourceParams hook: from now on it hands over the parameters of exactly the voices whose properties
 * the bridge changed (= the voices CalcSourceParams recomputes, alu.cpp:2012-2031) instead of comparing every voice's */
int oalbridge_track_changes(oalbridge *b, int on)
{
    b->trackChanges = on != 0;
    b->batch->trackChanges(on != 0);
    return 0;
}

/* the hooks INSIDE alc/alu.cpp (include/oalgpu_openal_hooks.hpp): CalcVoiceParams names the voices it recomputes, and on an HRTF
 * device CalcPanningAndFilters' getCoeffs call sites hand the batch mixer the direction (the device context evaluates getCoeffs)
 * -- nothing on the application's side of the bridge tells the mixer anything */
int oalbridge_hook_alu(oalbridge *b, int on)
{
    b->hookAlu = on != 0;
    b->batch->trackChanges(on != 0);
    b->batch->hookDirections(on != 0);
    return 0;
}
unsigned long long oalbridge_hooked_directions(oalbridge *b) { return b->hookedDirs; }

/* the batch mixer's pipelined mode (include/oalgpu_openal.hpp: the post-process behind the boundary too, an update's output
 * `depth` updates late), before the first update; oalbridge_drain collects what is outstanding: `depth` updates' output,
 * interleaved stereo, `frames` frames each, one update after the other */
int oalbridge_set_pipelined(oalbridge *b, uint32_t depth)
{
    b->batch->setPipelined(depth);
    return 0;
}

/* back to the synchronous form (BatchMixer::leavePipelined; oalbridge_drain first keeps the outstanding output) */
int oalbridge_leave_pipelined(oalbridge *b)
{
    b->batch->leavePipelined(b->ctx.get(), *b->dev);
    return 0;
}

int oalbridge_batch_times(oalbridge *b, double out[3]) { for(int i = 0; i < 3; ++i) out[i] = b->batch->times()[i]; return 0; }

int oalbridge_drain(oalbridge *b, float *interleaved, uint32_t frames, uint32_t max_updates)
{
    auto &dev = *b->dev;
    uint32_t got = 0;
    while(got < max_updates && b->batch->pendingUpdates() > 0)
    {
        std::ranges::fill(dev.MixBuffer | std::views::join, 0.0f);
        if(int rc = b->batch->collectOne(b->ctx.get(), dev)) { b->error = rc; b->errorText = b->batch->errorText(); return -1; }
        for(uint32_t i = 0; i < frames; ++i)
        {
            interleaved[(size_t{got} * frames + i) * 2 + 0] = dev.RealOut.Buffer[0][i];
            interleaved[(size_t{got} * frames + i) * 2 + 1] = dev.RealOut.Buffer[1][i];
        }
        ++got;
    }
    return int(got);
}

/* voices the batched mixer currently keeps a device-side slot for (stopped voices give theirs back) */
int oalbridge_batch_live_voices(oalbridge *b) { return int(b->batch->liveVoices()); }

int oalbridge_add_source(oalbridge *b, int buffer, int looping, int position, float gain, float x, float y, float z,
    int resampler, float pitch, float gain_hf)
{ return oalbridge_add_source_ex(b, buffer, looping, position, gain, x, y, z, resampler, pitch, gain_hf, -1, 1.0f, 1.0f); }

/* new source properties for the next update (a moved source): CalcVoiceParams picks them up */
int oalbridge_update_source_ex(oalbridge *b, int source, float gain, float x, float y, float z, int resampler, float pitch,
    float gain_hf, int send_slot, float send_gain, float send_gain_hf)
{
    Voice *v = b->sources.at(size_t(source));
    auto *props = NewProps(b);
    FillProps(b, *props, gain, x, y, z, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf);
    if(auto *old = v->mUpdate.exchange(props, std::memory_order_acq_rel))
        AtomicReplaceHead(b->ctx->mFreeVoiceProps, old);
    if(b->trackChanges) b->batch->noteParamsChanged(v);
    return 0;
}

int oalbridge_update_source(oalbridge *b, int source, float gain, float x, float y, float z, int resampler, float pitch,
    float gain_hf)
{ return oalbridge_update_source_ex(b, source, gain, x, y, z, resampler, pitch, gain_hf, -1, 1.0f, 1.0f); }

int oalbridge_stop_source(oalbridge *b, int source)
{   /* what ProcessVoiceChanges does for VChangeState::Stop (alu.cpp:2081-2100) */
    Voice *v = b->sources.at(size_t(source));
    auto st = Voice::Playing;
    v->mPlayState.compare_exchange_strong(st, Voice::Stopping, std::memory_order_relaxed, std::memory_order_acquire);
    return 0;
}

/* DeviceBase::renderSamples(void*, unsigned, size_t): `frames` interleaved stereo float frames */
int oalbridge_render(oalbridge *b, float *interleaved, uint32_t frames)
{
    gActive = b;
    b->batch->reset();
    b->dev->renderSamples(interleaved, frames, 2u);
    gActive = nullptr;
    return b->error;
}

const char *oalbridge_error(oalbridge *b) { return b->errorText.c_str(); }

/* source state for the comparison: play state, position, fraction, mStep of the voice */
int oalbridge_source_state(oalbridge *b, int source, int32_t out[4])
{
    Voice *v = b->sources.at(size_t(source));
    out[0] = int(v->mPlayState.load(std::memory_order_relaxed));
    out[1] = v->mPosition.load(std::memory_order_relaxed);
    out[2] = int32_t(v->mPositionFrac.load(std::memory_order_relaxed));
    out[3] = int32_t(v->mStep);
    return 0;
}

/* ... and what else Voice::mix leaves behind: whether a buffer is still attached, VoiceFlag::IsFading, HasHrtf */
int oalbridge_source_flags(oalbridge *b, int source, int32_t out[3])
{
    Voice *v = b->sources.at(size_t(source));
    out[0] = v->mCurrentBuffer.load(std::memory_order_relaxed) != nullptr;
    out[1] = v->mFlags.test(VoiceFlag::IsFading);
    out[2] = v->mFlags.test(VoiceFlag::HasHrtf);
    return 0;
}

/* how many per-call adapter invocations ran (mode ADAPTERS): [resample, mix, mix_hrtf, mix_hrtf_blend] */
int oalbridge_adapter_calls(uint64_t out[4])
{
    for(int i{0}; i < 4; ++i) out[i] = oalgpu_openal::Adapters().calls[i];
    return 0;
}

/* ---- the reference's output stage on a known signal: `lines` (nlines x 1024) replace the voice loop's
 * result on RealOut (nlines <= 2 here: the device is stereo), then Limiter (none), ApplyDither (depth > 0)
 * and Write<T> run as DeviceBase::renderSamples runs them.  fmt: DevFmtType (0 i8, 1 u8, 2 i16, 3 u16,
 * 4 i32, 5 u32, 6 f32); the post-process is taken out so that RealOut is exactly `lines`. */
int oalbridge_render_lines(oalbridge *b, const float *lines, uint32_t nlines, int fmt, float dither_depth,
    uint32_t *dither_seed, void *out, uint32_t frames, uint32_t frame_step)
{
    auto &dev = *b->dev;
    if(b->sources.empty()) return -1;                    /* the hook rides on a playing voice */
    b->inject.assign(lines, lines + size_t{nlines}*BufferLineSize);
    auto saved = std::move(dev.mPostProcess);
    dev.mPostProcess.emplace<std::monostate>();
    dev.FmtType = static_cast<DevFmtType>(fmt);
    dev.DitherDepth = dither_depth;
    dev.DitherSeed = *dither_seed;
    gActive = b;
    dev.renderSamples(out, frames, frame_step);
    gActive = nullptr;
    *dither_seed = dev.DitherSeed;
    dev.DitherDepth = 0.0f;
    dev.FmtType = DevFmtFloat;
    dev.mPostProcess = std::move(saved);
    b->inject.clear();
    return 0;
}

const char *oalbridge_kind(void) { return "reference+bridge"; }

} /* extern "C" */
