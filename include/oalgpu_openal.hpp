/* oalgpu_openal.hpp -- the OpenAL Soft side of liboalgpu.so, header-only: what a maintainer of kcat/openal-soft
 * drops into the tree to put the GPU behind alc/alu.cpp's voice loop (INTEGRATION.md sections 3 and 4).
 *
 * It needs the reference's own headers (include it from a translation unit that already has core/voice.h,
 * core/device.h, core/mixer.h, core/mixer/defs.h, core/mixer/hrtfdefs.h and core/filters/biquad.h) and nothing
 * else of this repository besides oalgpu.h.  Two layers:
 *
 *  (1) adapters with the reference's function-pointer signatures on top of the per-call C-ABI --
 *        ResamplerFunc        core/mixer/defs.h:71-72     -> oalgpu_resample
 *        MixerOutFunc         core/mixer.h:22-27          -> oalgpu_mix
 *        HrtfMixerFunc        core/voice.cpp:73-75        -> oalgpu_mix_hrtf
 *        HrtfMixerBlendFunc   core/voice.cpp:76-81        -> oalgpu_mix_hrtf_blend
 *      installed exactly where Voice::InitMixer (core/voice.cpp:139-193) installs the CPU variants;
 *
 *  (2) BatchMixer: the voice loop itself.  ProcessContexts calls voice->mix() for every playing voice
 *      (alc/alu.cpp:2201-2206); with the GPU behind it the calls of one update are collected and, with the last
 *      one, described to the device context -- oalgpu_voice_params filled from the Voice AFTER the reference's own
 *      CalcVoiceParams computed mStep, the pan gains and the filter targets (alu.cpp:1512-1710) -- and mixed by
 *      ONE oalgpu_mix_update; the dry lines are added into DeviceBase::MixBuffer and the state the rest of the
 *      reference looks at (positions, play state, fade flag) is read back.
 *
 * BiquadInterpFilter keeps its target coefficients private; the shelf gains are recovered from them
 * (ShelfGainAt), so the including translation unit must see them -- upstream that is one friend declaration in
 * core/filters/biquad.h; this repository's compiled bridge (oracle/ref_bridge.cpp) opens the class instead.
 * Mono float32 static buffers, no auxiliary sends: the scope of BASELINE configs[0]. */
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <span>
#include <string>
#include <utility>
#include <vector>

#include "oalgpu.h"

namespace oalgpu_openal {

/* ---- (1) adapters ------------------------------------------------------------------------------------------ */
struct AdapterState {
    int device{0};
    int mathMode{OALGPU_MATH_EXACT};
    int resamplerKind{OALGPU_RESAMPLER_LINEAR};      /* of the voice Voice::mix is working on */
};
inline AdapterState &Adapters() { static AdapterState s; return s; }

/* `src` is DeviceBase::mResampleData as voice.cpp:768 hands it over: it begins MaxResamplerEdge samples before the
 * first source sample, which is the convention of oalgpu_resample too */
inline void Resample(InterpState const*, std::span<float const> src, unsigned frac, unsigned increment,
    std::span<float> dst) noexcept
{
    const AdapterState &a = Adapters();
    oalgpu_resample(a.device, a.mathMode, a.resamplerKind, increment, src.data(), src.size(), frac, dst.data(), dst.size());
}

inline void Mix(std::span<float const> in, std::span<FloatBufferLine> out, std::span<float> cur,
    std::span<float const> tgt, std::size_t counter, std::size_t outpos) noexcept
{
    oalgpu_mix(Adapters().device, in.data(), in.size(), out[0].data(), out.size(), cur.data(), tgt.data(), counter, outpos);
}

inline void MixHrtf(std::span<float const> in, std::span<f32x2> accum, unsigned irSize, MixHrtfFilter const *f,
    std::size_t n) noexcept
{
    const uint32_t delay[2]{f->Delay[0], f->Delay[1]};
    oalgpu_mix_hrtf(Adapters().device, Adapters().mathMode, in.data(), &accum[0][0], irSize, &f->Coeffs[0][0], delay,
        f->Gain, f->GainStep, n);
}

inline void MixHrtfBlend(std::span<float const> in, std::span<f32x2> accum, unsigned irSize, HrtfFilter const *oldp,
    MixHrtfFilter const *newp, std::size_t n) noexcept
{
    const uint32_t od[2]{oldp->Delay[0], oldp->Delay[1]}, nd[2]{newp->Delay[0], newp->Delay[1]};
    oalgpu_mix_hrtf_blend(Adapters().device, Adapters().mathMode, in.data(), &accum[0][0], irSize, &oldp->Coeffs[0][0], od,
        oldp->Gain, &newp->Coeffs[0][0], nd, newp->GainStep, n);
}

/* ---- (2) the batched voice loop ------------------------------------------------------------------------------ */
/* the shelf gains CalcPanningAndFilters designed the voice's direct filters with (alu.cpp:1619-1637):
 * BiquadFilter::SetParams (biquad.cpp:48-129) builds the shelves with A = gain, so a high shelf answers
 * gain^2 at Nyquist and a low shelf gain^2 at DC */
inline float ShelfGainAt(const BiquadInterpFilter &f, float z /* +1: DC, -1: Nyquist */)
{
    const auto &c = f.mTargetCoeffs;
    return std::sqrt(std::max((c.mB0 + c.mB1*z + c.mB2) / (1.0f + c.mA1*z + c.mA2), 0.0f));
}

class BatchMixer {
public:
    explicit BatchMixer(int mathMode = OALGPU_MATH_FAST, int device = 0) : mMathMode{mathMode}, mDevice{device} { }
    BatchMixer(const BatchMixer&) = delete;
    BatchMixer &operator=(const BatchMixer&) = delete;
    ~BatchMixer() { if(mGpu) oalgpu_context_destroy(mGpu); }

    int error() const { return mError; }
    const std::string &errorText() const { return mErrorText; }

    /* Voice::mix's place in the voice loop: returns true when this call completed the update's batch and the
     * batch was mixed (false: more voices to come, or an error -- then error() != 0 and the caller runs the CPU
     * loop over batch() for that update). */
    bool mix(Voice *voice, Voice::State vstate, ContextBase *context, DeviceBase &dev, unsigned samplesToDo)
    {
        if(mSeen == 0)
        {
            mExpected = 0;
            for(Voice *v : context->getVoicesSpanAcquired())
            {
                auto const st = v->mPlayState.load(std::memory_order_acquire);
                if(st != Voice::Stopped && st != Voice::Pending) ++mExpected;
            }
            mBatch.clear();
        }
        mBatch.emplace_back(voice, vstate);
        if(++mSeen != mExpected) return false;
        mSeen = 0;
        return flush(dev, samplesToDo) == 0;
    }
    bool batchComplete() const { return mSeen == 0; }
    void reset() { mSeen = 0; }                     /* a new update begins (DeviceBase::renderSamples) */
    const std::vector<std::pair<Voice*, Voice::State>> &batch() const { return mBatch; }

private:
    int fail(int rc, const char *what)
    {
        if(!mError) { mError = rc; mErrorText = std::string(what) + ": " + oalgpu_last_error(); }
        return rc;
    }

    int flush(DeviceBase &dev, unsigned samplesToDo)
    {
        if(!mGpu)
        {
            oalgpu_context_desc d{};
            d.device = mDevice; d.math_mode = mMathMode; d.sample_rate = dev.mSampleRate;
            d.num_dry_channels = uint32_t(dev.Dry.Buffer.size());
            d.num_real_channels = uint32_t(dev.RealOut.Buffer.size());
            d.num_aux_sends = 0; d.num_slots = 0; d.wet_channels = 4; d.hrtf = 0;
            d.max_voices = 1024; d.max_buffers = 256; d.voices_per_group = 0; d.flags = 0;
            if(int rc = oalgpu_context_create(&d, &mGpu)) return fail(rc, "oalgpu_context_create");
        }
        std::vector<uint32_t> ids;
        std::vector<oalgpu_voice_params> params;
        for(auto &[voice, vstate] : mBatch)
        {
            auto it = mVoiceIndex.find(voice);
            if(it == mVoiceIndex.end())
            {   /* a voice that starts playing: register its buffer once, InitVoice (al/source.cpp:639-670) */
                auto *item = voice->mCurrentBuffer.load(std::memory_order_relaxed);
                if(!item) continue;
                auto hb = mBufferHandle.find(item);
                if(hb == mBufferHandle.end())
                {
                    auto const *span = std::get_if<std::span<f32>>(&item->mSamples);
                    if(!span) return fail(OALGPU_ERR_INVALID, "oalgpu_openal: float buffers only");
                    const int h = oalgpu_buffer_register(mGpu, span->data(), OALGPU_FMT_FLOAT, voice->mFrameStep,
                        item->mSampleLen, item->mLoopStart, item->mLoopEnd);
                    if(h < 0) return fail(h, "oalgpu_buffer_register");
                    hb = mBufferHandle.emplace(item, h).first;
                }
                const uint32_t idx = uint32_t(mVoiceIndex.size());
                oalgpu_voice_desc vd{hb->second, voice->mLoopBuffer.load(std::memory_order_relaxed) != nullptr,
                    voice->mPosition.load(std::memory_order_relaxed), voice->mPositionFrac.load(std::memory_order_relaxed),
                    voice->mFrequency};
                if(int rc = oalgpu_voice_init(mGpu, idx, &vd)) return fail(rc, "oalgpu_voice_init");
                it = mVoiceIndex.emplace(voice, idx).first;
                mLastState[voice] = Voice::Playing;
            }
            if(mLastState[voice] != int(vstate))
            {   /* ProcessVoiceChanges' play-state changes (alu.cpp:2057-2151) */
                if(int rc = oalgpu_voice_set_state(mGpu, it->second, int(vstate))) return fail(rc, "oalgpu_voice_set_state");
                mLastState[voice] = int(vstate);
            }
            /* what CalcVoiceParams left in the Voice (alu.cpp:1512-1710, :2012-2031) */
            oalgpu_voice_params p{};
            p.step = voice->mStep;
            p.resampler = int(voice->mProps.mResampler);
            auto &chan = voice->mChans[0];
            const float inv_rate = 1.0f / float(dev.mSampleRate);
            p.direct_filter.active = voice->mDirect.FilterActive ? 1 : 0;
            p.direct_filter.hf_norm = voice->mProps.Direct.HFReference * inv_rate;
            p.direct_filter.lf_norm = voice->mProps.Direct.LFReference * inv_rate;
            p.direct_filter.gain_hf = voice->mDirect.FilterActive ? ShelfGainAt(chan.mDryParams.LowPass, -1.0f) : 1.0f;
            p.direct_filter.gain_lf = voice->mDirect.FilterActive ? ShelfGainAt(chan.mDryParams.HighPass, 1.0f) : 1.0f;
            for(size_t c{0}; c < dev.Dry.Buffer.size(); ++c) p.dry_gains[c] = chan.mDryParams.Gains.Target[c];
            for(int s{0}; s < OALGPU_MAX_SENDS; ++s)
            {
                p.send_slot[s] = -1;
                p.send_filter[s] = oalgpu_filter_params{0, 1.0f, 5000.0f*inv_rate, 1.0f, 250.0f*inv_rate};
            }
            ids.push_back(it->second);
            params.push_back(p);
        }
        if(!ids.empty())
            if(int rc = oalgpu_voice_set_params(mGpu, ids.data(), params.data(), ids.size())) return fail(rc, "oalgpu_voice_set_params");
        /* the voice loop: one batched update, then the dry lines join the device's mixing buffer */
        if(int rc = oalgpu_mix_update(mGpu, samplesToDo, 0)) return fail(rc, "oalgpu_mix_update");
        mLines.resize((dev.Dry.Buffer.size() + dev.RealOut.Buffer.size()) * BufferLineSize);
        if(int rc = oalgpu_read_dry(mGpu, mLines.data())) return fail(rc, "oalgpu_read_dry");
        for(size_t c{0}; c < dev.Dry.Buffer.size(); ++c)
            for(size_t i{0}; i < samplesToDo; ++i)
                dev.Dry.Buffer[c][i] += mLines[c*BufferLineSize + i];
        /* the state the reference mutates in place stays authoritative on the device; what the rest of the
         * reference looks at (GetSourceOffset, the play state) is read back */
        for(auto &[voice, vstate] : mBatch)
        {
            auto it = mVoiceIndex.find(voice);
            if(it == mVoiceIndex.end()) continue;
            oalgpu_voice_state st{};
            if(int rc = oalgpu_voice_readback(mGpu, it->second, &st)) return fail(rc, "oalgpu_voice_readback");
            voice->mPosition.store(st.position, std::memory_order_relaxed);
            voice->mPositionFrac.store(st.position_frac, std::memory_order_relaxed);
            if(st.fading) voice->mFlags.set(VoiceFlag::IsFading);
            if(st.play_state != int(vstate))
            {
                voice->mPlayState.store(static_cast<Voice::State>(st.play_state), std::memory_order_release);
                mLastState[voice] = st.play_state;
            }
        }
        return 0;
    }

    int mMathMode, mDevice;
    oalgpu_context *mGpu{nullptr};
    std::map<const VoiceBufferItem*, int> mBufferHandle;
    std::map<const Voice*, uint32_t> mVoiceIndex;
    std::map<const Voice*, int> mLastState;
    unsigned mExpected{0}, mSeen{0};
    std::vector<std::pair<Voice*, Voice::State>> mBatch;
    std::vector<float> mLines;
    int mError{0};
    std::string mErrorText;
};

} // namespace oalgpu_openal
