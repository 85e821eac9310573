/* oalgpu_openal.hpp -- the OpenAL Soft side of liboalgpu.so, header-only: what a maintainer of kcat/openal-soft
 * drops into the tree to put the GPU behind alc/alu.cpp's voice loop (INTEGRATION.md sections 3 and 4).
 *
 * It needs the reference's own headers (include it from a translation unit that already has core/voice.h,
 * core/device.h, core/context.h, core/effectslot.h, core/hrtf.h, core/async_event.h, ringbuffer.h, core/mixer.h,
 * core/mixer/defs.h, core/mixer/hrtfdefs.h and core/filters/biquad.h) and nothing else of this repository besides
 * oalgpu.h.  Two layers:
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
 *      one, described to the device context -- from the Voice AFTER the reference's own CalcVoiceParams computed
 *      mStep, the pan gains, the filter targets and, on an HRTF device, Hrtf.Target (alu.cpp:1512-1710) -- and
 *      mixed by ONE oalgpu_mix_update.  What the voices produced joins the reference's own mixing buffers exactly
 *      where Voice::mix would have put it: the dry lines into DeviceBase::Dry (MixSamples, voice.cpp:962-963), the
 *      dual-ear FIR's output into DeviceBase::HrtfAccumData (DoHrtfMix, voice.cpp:827-902), every send into its
 *      slot's wet buffer (voice.cpp:966-979); the reference carries on with its effect slots (alu.cpp:2209-2257) and
 *      its post-process (DeviceBase::Process(HrtfPostProcess) / (AmbiDecPostProcess), alu.cpp:282-298), which also
 *      owns the HRTF accumulator's tail.  The state the rest of the reference looks at (positions, play state, the
 *      end of a source) is read back.
 *
 * Scope: static sources (VoiceFlag::IsStatic) of any channel count -- a channel of interleaved frames is one device voice,
 * Voice::mix's per-ChannelData loop (voice.cpp:1058-1091) -- and mono streaming sources (buffer queues that grow while
 * they play, voice.cpp:563-594, :1182-1218), of every PCM sample type and both ADPCM types, with delayed starts
 * (voice.cpp:1023-1046), on a device in RenderMode::Normal (ambisonic dry lines) or RenderMode::Hrtf, with up to MaxSendCount
 * auxiliary sends into the context's active effect slots.  Buffers are registered when a voice first starts on them and given
 * up when their storage changes (oalgpu_buffer_release).  On a RenderMode::Normal device also B-Format sources the device's order
 * exceeds (VoiceFlag::IsAmbisonic: a device voice per channel with its HF / LF scales, oalgpu_voice_set_ambi_scale; voice.cpp:1082-1091)
 * and near-field-compensated voices (VoiceFlag::HasNfc on a device with a control distance: oalgpu_context_set_nfc once,
 * oalgpu_voice_set_nfc with the w0 CalcPanningAndFilters adjusted the voice's NFCtrlFilter to; voice.cpp:904-932).  Mono callback sources
 * (VoiceFlag::IsCallback: the library calls the buffer's function from inside oalgpu_mix_update, voice.cpp:726-752, :1155-1180) on either
 * device.  What is left -- UHJ sources, direct channels, B-Format sources on an HRTF device (they mix into the ambisonic dry lines the
 * device context of an HRTF device does not have) -- makes mix() report an error BEFORE anything on the device changed, and the
 * caller runs the reference's own loop for that update; the device context then starts every voice over from its Voice.
 *
 * BiquadInterpFilter keeps its target coefficients private, NfcFilter its sections and BFormatDec its matrix; the shelf gains,
 * w0 and the decoder are read from them (ShelfGainAt, NfcW0, createContext), so the including translation unit must see them --
 * upstream that is one friend declaration each in core/filters/biquad.h, core/filters/nfc.h and core/bformatdec.h; this repository's compiled bridge (oracle/ref_bridge.cpp) opens the
 * classes instead. */
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <chrono>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <span>
#include <string>
#include <utility>
#include <variant>
#include <vector>

#include "oalgpu.h"

namespace oalgpu_openal {

/* ---- (1) adapters ------------------------------------------------------------------------------------------ */
struct AdapterState {
    int device{0};
    int mathMode{OALGPU_MATH_EXACT};
    int resamplerKind{OALGPU_RESAMPLER_LINEAR};      /* of the voice Voice::mix is working on */
    uint64_t calls[4]{};                             /* resample, mix, mix_hrtf, mix_hrtf_blend: invocations so far */
};
inline AdapterState &Adapters() { static AdapterState s; return s; }

/* `src` is DeviceBase::mResampleData as voice.cpp:768 hands it over: it begins MaxResamplerEdge samples before the
 * first source sample, which is the convention of oalgpu_resample too */
inline void Resample(InterpState const*, std::span<float const> src, unsigned frac, unsigned increment,
    std::span<float> dst) noexcept
{
    AdapterState &a = Adapters();
    ++a.calls[0];
    oalgpu_resample(a.device, a.mathMode, a.resamplerKind, increment, src.data(), src.size(), frac, dst.data(), dst.size());
}

inline void Mix(std::span<float const> in, std::span<FloatBufferLine> out, std::span<float> cur,
    std::span<float const> tgt, std::size_t counter, std::size_t outpos) noexcept
{
    ++Adapters().calls[1];
    oalgpu_mix(Adapters().device, in.data(), in.size(), out[0].data(), out.size(), cur.data(), tgt.data(), counter, outpos);
}

inline void MixHrtf(std::span<float const> in, std::span<f32x2> accum, unsigned irSize, MixHrtfFilter const *f,
    std::size_t n) noexcept
{
    const uint32_t delay[2]{f->Delay[0], f->Delay[1]};
    ++Adapters().calls[2];
    oalgpu_mix_hrtf(Adapters().device, Adapters().mathMode, in.data(), &accum[0][0], irSize, &f->Coeffs[0][0], delay,
        f->Gain, f->GainStep, n);
}

inline void MixHrtfBlend(std::span<float const> in, std::span<f32x2> accum, unsigned irSize, HrtfFilter const *oldp,
    MixHrtfFilter const *newp, std::size_t n) noexcept
{
    const uint32_t od[2]{oldp->Delay[0], oldp->Delay[1]}, nd[2]{newp->Delay[0], newp->Delay[1]};
    ++Adapters().calls[3];
    oalgpu_mix_hrtf_blend(Adapters().device, Adapters().mathMode, in.data(), &accum[0][0], irSize, &oldp->Coeffs[0][0], od,
        oldp->Gain, &newp->Coeffs[0][0], nd, newp->GainStep, n);
}

/* ---- (2) the batched voice loop ------------------------------------------------------------------------------ */
/* the shelf gains CalcPanningAndFilters designed the voice's filters with (alu.cpp:1619-1657):
 * BiquadFilter::SetParams (biquad.cpp:48-129) builds the shelves with A = gain, so a high shelf answers
 * gain^2 at Nyquist and a low shelf gain^2 at DC */
inline float ShelfGainAt(const BiquadInterpFilter &f, float z /* +1: DC, -1: Nyquist */)
{
    const auto &c = f.mTargetCoeffs;
    return std::sqrt(std::max((c.mB0 + c.mB1*z + c.mB2) / (1.0f + c.mA1*z + c.mA2), 0.0f));
}

/* the w0 NfcFilter::adjust was last called with (alu.cpp:919-941, :1328-1341, :1415-1427): the first-order section keeps
 * b1 = 2 (w0 / 2) / (1 + w0 / 2)  (NfcFilterAdjust1, core/filters/nfc.cpp:75-83) */
inline float NfcW0(const NfcFilter &f)
{
    const float b1 = f.first.mCoeffs.b1;
    return 2.0f * b1 / (2.0f - b1);
}

class BatchMixer {
public:
    explicit BatchMixer(int mathMode = OALGPU_MATH_FAST, int device = 0, unsigned maxVoices = 4096)
        : mMathMode{mathMode}, mDevice{device}, mMaxVoices{maxVoices} { }
    BatchMixer(const BatchMixer&) = delete;
    BatchMixer &operator=(const BatchMixer&) = delete;
    ~BatchMixer()
    {
        if(mGpu) oalgpu_context_destroy(mGpu);
        for(oalgpu_reverb *r : mReverbs) oalgpu_reverb_destroy(r);
    }

    int error() const { return mError; }
    const std::string &errorText() const { return mErrorText; }

    /* Voice::mix's place in the voice loop: returns true when this call completed the update's batch and the
     * batch was mixed (false: more voices to come, or an error -- then error() != 0 and the caller runs the CPU
     * loop over batch() for that update).  deviceTime: Voice::mix's own argument (a delayed start,
     * voice.cpp:1023-1046, is measured against it). */
    bool mix(Voice *voice, Voice::State vstate, ContextBase *context, DeviceBase &dev, std::chrono::nanoseconds deviceTime,
        unsigned samplesToDo)
    {
        auto const voices = context->getVoicesSpanAcquired();
        if(mSeen == 0)
        {
            mExpected = 0;
            for(Voice *v : voices)
            {
                auto const st = v->mPlayState.load(std::memory_order_acquire);
                if(st != Voice::Stopped && st != Voice::Pending) ++mExpected;
            }
            mBatch.clear(); mBatchIndex.clear(); mBatchNamed.clear(); mBatchDirs.clear();
            mOrderedAt = 0; mDirAt = 0;
            mCursor = 0;
        }
        /* the voice loop walks the context's voices in order (alu.cpp:2201-2206): so does the cursor, and a voice's place in
         * that array is where its entry lives (voices are pooled; the array only grows, core/context.h:159-160) */
        /* (the voices CalcVoiceParams named through its hook come in that order too, alu.cpp:2177-2206: the list is consumed in step
         * with the cursor -- one pointer comparison per voice of the array instead of a hash lookup per mixed voice) */
        bool named = false;
        size_t dirFrom = mDirAt;
        while(mCursor < voices.size())
        {
            const bool hit = mOrderedAt < mOrdered.size() && mOrdered[mOrderedAt] == voices[mCursor];
            if(hit) ++mOrderedAt;
            dirFrom = mDirAt;
            while(mDirAt < mDirList.size() && mDirList[mDirAt].voice == voices[mCursor]) ++mDirAt;
            if(voices[mCursor] == voice) { named = hit; break; }
            ++mCursor;
        }
        const uint32_t dirCount = uint32_t(mDirAt - dirFrom);
        if(mCursor == voices.size()) { mSeen = 0; failText(OALGPU_ERR_INVALID, "oalgpu_openal: a voice outside the context's voice array"); return false; }
        /* pipelined, with the parameter hook: a voice that just plays on -- known to the device context, no change of state, of
         * parameters or of its queue -- needs no word; only the others are walked when the batch is complete */
        const size_t place = mCursor++;
        bool quiet = false;
        if(mDepth && mTrack && !mResync && place < mEntries.size())
        {
            const Entry &e = mEntries[place];
            quiet = e.live && !e.queue && vstate == Voice::Playing && e.lastState == int(Voice::Playing)
                && e.sourceId == voice->mSourceID.load(std::memory_order_relaxed) && !(voice->mStartTime > deviceTime)
                && voice->mStep >= 1u && !named && dirCount == 0u && (mChanged.empty() || !mChanged.count(voice));
        }
        if(!quiet)
        {
            mBatch.emplace_back(voice, vstate);
            mBatchIndex.push_back(uint32_t(place));
            mBatchNamed.push_back(named ? 1 : 0);
            mBatchDirs.emplace_back(uint32_t(dirFrom), dirCount);
        }
        if(++mSeen != mExpected) return false;
        mSeen = 0;
        /* (names that were not met on the way -- the order of the two loops is the reference's, not this file's, to keep: then every
         * voice is compared this once) */
        while(mOrderedAt < mOrdered.size() && mCursor < voices.size())
        {
            if(mOrdered[mOrderedAt] == voices[mCursor]) ++mOrderedAt;
            ++mCursor;
        }
        mCompareAll = mOrderedAt != mOrdered.size();
        const int rc = flush(context, dev, deviceTime, samplesToDo);
        mChanged.clear();
        mOrdered.clear(); mOrderedAt = 0; mCompareAll = false;
        mDirList.clear(); mDirAt = 0;
        if(rc != 0)
        {
            mResync = true;                 /* the caller mixes this update on the CPU: what the device context holds is stale then */
            if(mHookDirs && dev.mHrtf) restoreTargets(*dev.mHrtf);     /* ... and its Voice::mix reads the targets the hook skipped */
            if(mDepth) leavePipelined(context, dev);   /* ... and the CPU loop needs the device's own post-process back */
        }
        return rc == 0;
    }
    bool batchComplete() const { return mSeen == 0; }
    void reset() { mSeen = 0; }                     /* a new update begins (DeviceBase::renderSamples) */
    const std::vector<std::pair<Voice*, Voice::State>> &batch() const { return mBatch; }
    size_t liveVoices() const { size_t n{0}; for(auto const &e : mEntries) n += e.live ? 1u : 0u; return n; }
    size_t liveBuffers() const { return mBuffers.size(); }

    /* ---- the pipelined mode: nothing of an update is waited for in the update itself ----
     * The synchronous form above reads the update's buses and every voice's state back before ProcessContexts goes on -- five
     * blocking calls per update.  setPipelined(depth), before the first update, moves what is left of the update behind the
     * boundary as well: the device's HRTF post-process (MixDirectHrtf over the dry lines and the accumulator, alu.cpp:289-298)
     * runs on the GPU behind the voices, with the decoder the device was set up with (HrtfPostProcess::mHrtfState, handed over
     * as it is; the device keeps std::monostate in mPostProcess while the mode lasts), and an update's two output lines are
     * added to DeviceBase::RealOut `depth` updates later (oalgpu_read_output_async / oalgpu_output_wait: by then they have
     * long landed in pinned memory).  Voice state comes back the same way, as a report of what CHANGED
     * (oalgpu_voice_events_async): a source that ran out of buffer is heard of `depth` updates late -- Voice::mStartTime-style
     * latency the application sees as `depth` x update size more output latency -- and Voice::mPosition is refreshed with such
     * a report only (GetSourceOffset: oalgpu_voices_readback).  On a RenderMode::Normal device the post-process that moves is the
     * speaker decode (AmbiDecPostProcess: BFormatDec's matrix and crossover go to oalgpu_set_bformat_decoder) and the lines that
     * come back are the device's real output lines.  Auxiliary sends on an HRTF device: the effect slots move too -- EAX reverb
     * slots are bound once (oalgpu_reverb_create / _update from the slot's ReverbProps, oalgpu_slot_set_reverb), see createContext.
     * Scope: a constant update size; sends on RenderMode::Normal devices and other effect types are not pipelined.  drain() collects what is
     * outstanding (before the device stops, or before leaving the mode). */
    void setPipelined(unsigned depth) { mDepth = std::min(depth, 2u); }
    bool pipelined() const { return mDepth != 0u; }
    int drain(ContextBase *context, DeviceBase &dev)
    {
        while(!mPending.empty()) { if(int rc = collect(context, dev)) return rc; }
        return 0;
    }
    /* back to the synchronous form (also what a failed update does by itself): the voice state of what is outstanding is
     * applied, its output lines are dropped (drain() before, to have them), the device gets its HrtfPostProcess back and the
     * accumulator's tail is the reference's again.  One update's tail of the HRTF accumulator stays behind on the GPU: a seam
     * of IrSize samples, once. */
    void leavePipelined(ContextBase *context, DeviceBase &dev)
    {
        if(!mDepth) return;
        const int savedError = mError;
        const std::string savedText = mErrorText;
        while(!mPending.empty()) { if(collect(context, dev, true)) { mPending.clear(); break; } }     /* (drain() first to keep their output) */
        mError = savedError; mErrorText = savedText;
        if(mSavedPost.mHrtfState) dev.mPostProcess.emplace<HrtfPostProcess>(std::move(mSavedPost));
        if(mSavedDec.mAmbiDecoder)
        {
            dev.mPostProcess.emplace<AmbiDecPostProcess>(std::move(mSavedDec));
            if(mGpu) (void)oalgpu_set_bformat_decoder(mGpu, 0u, nullptr, nullptr, 0.0f);
        }
        if(mGpu && mHrtf) (void)oalgpu_set_carry_accum(mGpu, 0);
        mDepth = 0;
    }
    size_t pendingUpdates() const { return mPending.size(); }
    /* where flush() spent its time so far, seconds: [0] the walk over the update's voices (what to tell the device context),
     * [1] the submission (oalgpu_mix_update and the two requests behind it), [2] collecting the update `depth` back */
    const double *times() const { return mTimes; }
    int collectOne(ContextBase *context, DeviceBase &dev) { return mPending.empty() ? 0 : collect(context, dev); }

    /* ---- two optional hooks for the maintainer's side of the seam ----
     * CalcSourceParams (alc/alu.cpp:2012-2031) knows which voices it recomputed: with one call at its end the update hands
     * over exactly those voices' parameters instead of comparing every voice's (a voice's Hrtf.Target alone is 1 KB).
     * Without the hook (trackChanges(false), the default) every voice is compared. */
    void trackChanges(bool on) { mTrack = on; }
    void noteParamsChanged(const Voice *voice) { mChanged.insert(voice); }
    /* the same from inside CalcVoiceParams (oalgpu_hook::ParamsChanged): the calls come in the order of the context's voice array,
     * as the voice loop's do, so a list does */
    void noteParamsChangedInOrder(const Voice *voice) { mOrdered.push_back(voice); }
    /* The third: CalcPanningAndFilters' three HrtfStore::getCoeffs call sites (alc/alu.cpp:1214-1216, :1256-1258, :1296-1298),
     * include/oalgpu_openal_hooks.hpp.  With hookDirections(true) on an HRTF device the call hands over the DIRECTION -- elevation,
     * azimuth, distance, spread: 16 bytes -- and returns true: the reference neither indexes nor blends the four responses, the
     * voice's Hrtf.Target.Coeffs / .Delay stay as they are, and flush() passes the directions on as oalgpu_voice_move records
     * (the device context evaluates both halves of getCoeffs itself, bit for bit: SURVEY.md 8 f1) instead of 1 KB of blended
     * HrirArray per moved voice.  Returns false -- the caller runs the reference's getCoeffs -- when the mode is off or the voice
     * is none of this mixer's business.  An update that goes to the CPU loop first brings the skipped targets up to date
     * (restoreTargets). */
    void hookDirections(bool on) { mHookDirs = on; }
    bool noteHrtfDirection(const Voice *voice, size_t chan, float ev, float az, float dist, float spread)
    {
        if(!mHookDirs || !mHrtfKnown || !mHrtf || chan > 255u) return false;
        /* (in the order of the context's voice array, like noteParamsChangedInOrder's names: the seam picks them up in step) */
        mDirList.push_back(DirNote{voice, {ev, az, dist, spread}, uint8_t(chan)});
        return true;
    }
    /* the reference's own getCoeffs for every target the hook has skipped: before its Voice::mix reads them (an update on the
     * CPU loop) -- store: DeviceBase::mHrtf */
    void restoreTargets(const HrtfStore &store)
    {
        auto restore = [&store](const Voice *v, size_t c, const std::array<float, 4> &d)
        {
            auto *voice = const_cast<Voice*>(v);
            if(c >= voice->mChans.size()) return;
            store.getCoeffs(d[0], d[1], d[2], d[3], voice->mChans[c].mDryParams.Hrtf.Target.Coeffs, voice->mChans[c].mDryParams.Hrtf.Target.Delay);
        };
        for(Entry &e : mEntries)
        {
            if(!e.live || !e.voice) continue;
            for(size_t c{0}; c < e.chans.size(); ++c)
                if(e.chans[c].dirStale) { restore(e.voice, c, e.chans[c].dir); e.chans[c].dirStale = false; }
        }
        for(auto const &n : mDirList) restore(n.voice, n.chan, n.dir);     /* (this update's, not yet filed) */
        mDirList.clear();
    }
    /* alDeleteBuffers / alBufferData on a buffer the mixer has seen (the storage is freed and the item reused,
     * core/buffer_storage.h:47-77, core/voice.h:84-98): its HBM copy is given up; without the hook a changed item is noticed
     * the next time a voice starts on it (the key below) */
    void forgetBuffer(const VoiceBufferItem *item)
    {
        auto it = mBuffers.find(item);
        if(it == mBuffers.end() || !mGpu) return;
        releaseEntry(it->second);
        mBuffers.erase(it);
    }

private:
    /* what a registered copy was made of: an item whose storage was replaced no longer matches */
    struct BufferKey {
        const void *data{nullptr}; size_t kind{0}; unsigned frameStep{0}, sampleLen{0}, loopStart{0}, loopEnd{0}, blockAlign{0};
        bool operator==(const BufferKey &o) const
        { return data == o.data && kind == o.kind && frameStep == o.frameStep && sampleLen == o.sampleLen && loopStart == o.loopStart
            && loopEnd == o.loopEnd && blockAlign == o.blockAlign; }
    };
    struct BufferEntry { BufferKey key; int handle{-1}; std::vector<int> views; };
    struct Chan {
        uint32_t index{0};
        bool haveParams{false};
        oalgpu_voice_params params{};
        bool haveTarget{false};
        HrtfFilter target{};                        /* the Hrtf.Target the device context was last given */
        std::array<float, 4> dir{};                 /* hookDirections: the direction the device context was last given INSTEAD ... */
        bool dirStale{false};                       /* ... and the Voice's own Hrtf.Target does not hold its response */
        bool haveNfc{false};                        /* near-field control: the w0 the device context was last given (-1: none) */
        float w0{-1.0f};
    };
    struct Entry {
        bool live{false};
        const Voice *voice{nullptr};                /* whose entry (the Voice object at this place of the context's array) */
        int lastState{int(Voice::Playing)};
        unsigned sourceId{0};
        std::vector<Chan> chans;                    /* one device voice per mixed channel (Voice::mix's ChannelData loop) */
        bool queue{false};                          /* a streaming source: the items it has been linked through, in order */
        std::vector<std::pair<const VoiceBufferItem*, int>> chain;
        uint32_t doneSeen{0};
        bool callback{false};                       /* a callback source (AL_SOFT_callback_buffer): the library calls the buffer's function */
    };
    /* oalgpu_callback_fn -> the item's CallbackType (core/voice.h:85-86): called by oalgpu_mix_update on the mixer thread, for the
     * byte counts Voice::mix would ask for (voice.cpp:726-752) */
    static int32_t CallbackTrampoline(void *item_, void *data, int32_t numBytes)
    {
        auto *item = static_cast<VoiceBufferItem*>(item_);
        return int32_t(item->mCallback(item->mUserData, data, numBytes));
    }

    int fail(int rc, const char *what)
    {
        if(!mError) { mError = rc; mErrorText = std::string(what) + ": " + oalgpu_last_error(); }
        return rc;
    }
    int failText(int rc, const char *what)
    {
        if(!mError) { mError = rc; mErrorText = what; }
        return rc;
    }

    /* DeviceBase -> oalgpu_context_desc; an HRTF device hands over the store it renders with (DeviceBase::mHrtf) */
    int createContext(ContextBase *context, DeviceBase &dev)
    {
        auto const auxspan = std::span{*context->mActiveAuxSlots.load(std::memory_order_acquire)};
        auto const auxslots = auxspan.first(auxspan.size() >> 1);
        mHrtf = dev.mRenderMode == RenderMode::Hrtf;
        mHrtfKnown = true;
        oalgpu_context_desc d{};
        d.device = mDevice; d.math_mode = mMathMode; d.sample_rate = dev.mSampleRate;
        d.num_dry_channels = uint32_t(dev.Dry.Buffer.size());
        d.num_real_channels = dev.RealOut.Buffer.data() != dev.Dry.Buffer.data() ? uint32_t(dev.RealOut.Buffer.size()) : 0u;
        d.num_aux_sends = dev.NumAuxSends;
        d.num_slots = dev.NumAuxSends ? uint32_t(std::max<size_t>(auxslots.size(), 1)) : 0u;
        d.wet_channels = 4;
        if(dev.NumAuxSends && !auxslots.empty())
        {
            d.wet_channels = uint32_t(auxslots[0]->Wet.Buffer.size());
            for(EffectSlotBase *slot : auxslots)
                if(slot->Wet.Buffer.size() != d.wet_channels)
                    return failText(OALGPU_ERR_INVALID, "oalgpu_openal: the effect slots' wet buses differ in size");
        }
        d.hrtf = mHrtf ? 1 : 0;
        d.max_voices = mMaxVoices; d.max_buffers = 1024; d.voices_per_group = 0; d.flags = 0;
        if(int rc = oalgpu_context_create(&d, &mGpu)) return fail(rc, "oalgpu_context_create");
        mNumSlots = d.num_slots; mWetChannels = d.wet_channels;
        mNfc = false;
        if(!mHrtf && dev.AvgSpeakerDist > 0.0f)
        {   /* InitNearFieldCtrl (alc/panning.cpp:285-299): the control filter's w1 and the lines per order */
            uint32_t perOrder[5]{};
            for(size_t o{0}; o < 5 && o < dev.NumChannelsPerOrder.size(); ++o) perOrder[o] = dev.NumChannelsPerOrder[o];
            const float w1 = SpeedOfSoundMetersPerSec / dev.AvgSpeakerDist / float(dev.mSampleRate);
            if(int rc = oalgpu_context_set_nfc(mGpu, w1, perOrder)) return fail(rc, "oalgpu_context_set_nfc");
            mNfc = true;
        }
        if(mHrtf)
        {
            HrtfStore const *store = dev.mHrtf.get();
            if(!store) return failText(OALGPU_ERR_NO_HRTF, "oalgpu_openal: RenderMode::Hrtf without DeviceBase::mHrtf");
            std::vector<float> dist; std::vector<uint8_t> evc; std::vector<uint16_t> azc, iro; std::vector<uint8_t> delays;
            for(auto const &f : store->mFields) { dist.push_back(f.distance); evc.push_back(f.evCount.c_val); }
            for(auto const &e : store->mElev) { azc.push_back(e.azCount.c_val); iro.push_back(e.irOffset.c_val); }
            for(auto const &dl : store->mDelays) { delays.push_back(dl[0].c_val); delays.push_back(dl[1].c_val); }
            /* (DeviceBase::mIrSize: the taps the mixers apply -- the store's IrSize unless "hrtf-size" shortened it,
             * alc/panning.cpp:1345-1350) */
            if(int rc = oalgpu_hrtf_load_store(mGpu, store->mSampleRate, dev.mIrSize, dist.data(), evc.data(), uint32_t(dist.size()),
                azc.data(), iro.data(), uint32_t(azc.size()), &store->mCoeffs[0][0][0], delays.data(), uint32_t(store->mCoeffs.size())))
                return fail(rc, "oalgpu_hrtf_load_store");
            if(!mDepth)
            {   /* DeviceBase::HrtfAccumData and its tail stay the reference's: MixDirectHrtf runs there (alu.cpp:289-298) */
                if(int rc = oalgpu_set_carry_accum(mGpu, 0)) return fail(rc, "oalgpu_set_carry_accum");
            }
        }
        if(mDepth && !mHrtf)
        {   /* a RenderMode::Normal device: the post-process that moves behind the boundary is the speaker decode, BFormatDec::process
             * (DeviceBase::Process(AmbiDecPostProcess), alu.cpp:282-287) -- the decoder's matrix, per input channel in BFormatDec, goes
             * over per output line; a dual-band decoder takes its crossover with it (core/bformatdec.cpp:27-95) */
            auto *pp = std::get_if<AmbiDecPostProcess>(&dev.mPostProcess);
            if(dev.NumAuxSends || !pp || !pp->mAmbiDecoder || dev.RealOut.Buffer.data() == dev.Dry.Buffer.data())
                return failText(OALGPU_ERR_INVALID, "oalgpu_openal: the pipelined mode covers devices without auxiliary sends whose post-process "
                    "is HrtfPostProcess or AmbiDecPostProcess");
            BFormatDec &dec = *pp->mAmbiDecoder;
            const size_t nreal = dev.RealOut.Buffer.size(), ndry = dev.Dry.Buffer.size();
            std::vector<float> hf(nreal * OALGPU_MAX_AMBI_CHANNELS, 0.0f), lf;
            if(auto *single = std::get_if<BFormatDec::SBandDecoderVector>(&dec.mChannelDec))
            {
                for(size_t in{0}; in < ndry && in < single->size(); ++in)
                    for(size_t out{0}; out < nreal; ++out) hf[out * OALGPU_MAX_AMBI_CHANNELS + in] = (*single)[in].mGains[out];
            }
            else
            {
                auto &dual = std::get<BFormatDec::DBandDecoderVector>(dec.mChannelDec);
                lf.assign(hf.size(), 0.0f);
                for(size_t in{0}; in < ndry && in < dual.size(); ++in)
                    for(size_t out{0}; out < nreal; ++out)
                    {
                        hf[out * OALGPU_MAX_AMBI_CHANNELS + in] = dual[in].mGains[BFormatDec::sHFBand][out];
                        lf[out * OALGPU_MAX_AMBI_CHANNELS + in] = dual[in].mGains[BFormatDec::sLFBand][out];
                    }
            }
            if(int rc = oalgpu_set_bformat_decoder(mGpu, uint32_t(nreal), hf.data(), lf.empty() ? nullptr : lf.data(), dev.mXOverFreq / float(dev.mSampleRate)))
                return fail(rc, "oalgpu_set_bformat_decoder");
            mSavedDec = std::move(*pp);
            dev.mPostProcess.emplace<std::monostate>();
            mEntryOfIndex.assign(mMaxVoices, -1);
            mIndexBorn.assign(mMaxVoices, 0u);
        }
        else if(mDepth)
        {   /* the post-process moves behind the boundary with the decoder InitHrtfPanning built (alc/panning.cpp:1100-1134) */
            auto *pp = std::get_if<HrtfPostProcess>(&dev.mPostProcess);
            if(!pp || !pp->mHrtfState)
                return failText(OALGPU_ERR_INVALID, "oalgpu_openal: the pipelined mode covers devices whose post-process is HrtfPostProcess or "
                    "AmbiDecPostProcess");
            /* Auxiliary sends: the effect slots move behind the boundary as well -- nothing of a wet bus comes back in this mode, so what
             * the slots' effects add to the dry lines has to be added THERE, in front of the device's post-process.  Covered: EAX reverb
             * slots (EffectSlotBase::mEffectProps as ReverbProps, the slot's gain) on a device whose dry lines are ACN first order in
             * order -- oalgpu_reverb_update pans for that map (ComputePanGains with AmbiMap[i] = {1, i}); the properties as they are
             * when the mode is entered (a changed slot goes through leavePipelined / setPipelined).  The reference's own effect leg
             * (alu.cpp:2209-2257) keeps running on the wet buffers nobody mixes into any more: silence in, silence out. */
            if(dev.NumAuxSends)
            {
                if(dev.Dry.Buffer.size() != 4u) return failText(OALGPU_ERR_INVALID, "oalgpu_openal: pipelined sends need first-order ACN dry lines");
                for(size_t i{0}; i < 4u; ++i)
                    if(dev.Dry.AmbiMap[i].Index != i || dev.Dry.AmbiMap[i].Scale != 1.0f)
                        return failText(OALGPU_ERR_INVALID, "oalgpu_openal: pipelined sends need first-order ACN dry lines");
                for(size_t sidx{0}; sidx < auxslots.size(); ++sidx)
                {
                    EffectSlotBase *slot = auxslots[sidx];
                    auto const *rp = std::get_if<ReverbProps>(&slot->mEffectProps);
                    if(slot->EffectType != EffectSlotType::Reverb || !rp)
                        return failText(OALGPU_ERR_INVALID, "oalgpu_openal: the pipelined mode binds EAX reverb slots only");
                    oalgpu_reverb *rev{nullptr};
                    if(int rc = oalgpu_reverb_create(mDevice, dev.mSampleRate, uint32_t(dev.Dry.Buffer.size()), &rev)) return fail(rc, "oalgpu_reverb_create");
                    mReverbs.push_back(rev);
                    oalgpu_reverb_props q{};
                    q.density = rp->Density; q.diffusion = rp->Diffusion; q.gain = rp->Gain; q.gain_hf = rp->GainHF; q.gain_lf = rp->GainLF;
                    q.decay_time = rp->DecayTime; q.decay_hf_ratio = rp->DecayHFRatio; q.decay_lf_ratio = rp->DecayLFRatio;
                    q.reflections_gain = rp->ReflectionsGain; q.reflections_delay = rp->ReflectionsDelay;
                    q.late_reverb_gain = rp->LateReverbGain; q.late_reverb_delay = rp->LateReverbDelay;
                    for(size_t k{0}; k < 3u; ++k) { q.reflections_pan[k] = rp->ReflectionsPan[k]; q.late_reverb_pan[k] = rp->LateReverbPan[k]; }
                    q.echo_time = rp->EchoTime; q.echo_depth = rp->EchoDepth; q.modulation_time = rp->ModulationTime;
                    q.modulation_depth = rp->ModulationDepth; q.air_absorption_gain_hf = rp->AirAbsorptionGainHF;
                    q.hf_reference = rp->HFReference; q.lf_reference = rp->LFReference; q.room_rolloff_factor = rp->RoomRolloffFactor;
                    q.decay_hf_limit = rp->DecayHFLimit ? 1 : 0;
                    if(int rc = oalgpu_slot_set_reverb(mGpu, uint32_t(sidx), rev)) return fail(rc, "oalgpu_slot_set_reverb");
                    if(int rc = oalgpu_reverb_update(rev, &q, slot->Gain)) return fail(rc, "oalgpu_reverb_update");
                }
            }
            DirectHrtfState &st = *pp->mHrtfState;
            std::vector<float> coeffs(st.mChannels.size() * HrirLength * 2), hf(st.mChannels.size());
            for(size_t c{0}; c < st.mChannels.size(); ++c)
            {
                std::memcpy(&coeffs[c * HrirLength * 2], &st.mChannels[c].mCoeffs[0][0], sizeof(float) * HrirLength * 2);
                hf[c] = st.mChannels[c].mHfScale;
            }
            if(int rc = oalgpu_set_direct_hrtf(mGpu, coeffs.data(), hf.data(), dev.mXOverFreq / float(dev.mSampleRate), st.mIrSize))
                return fail(rc, "oalgpu_set_direct_hrtf");
            if(int rc = oalgpu_set_carry_accum(mGpu, 1)) return fail(rc, "oalgpu_set_carry_accum");
            mSavedPost = std::move(*pp);
            dev.mPostProcess.emplace<std::monostate>();
            mEntryOfIndex.assign(mMaxVoices, -1);
            mIndexBorn.assign(mMaxVoices, 0u);
        }
        for(uint32_t i = 0; i < mMaxVoices; ++i) mFreeIndex.push_back(mMaxVoices - 1u - i);
        return 0;
    }

    void releaseEntry(BufferEntry &be)
    {
        for(int v : be.views) if(v >= 0) (void)oalgpu_buffer_release(mGpu, v);
        if(be.handle >= 0) (void)oalgpu_buffer_release(mGpu, be.handle);
        be.views.clear(); be.handle = -1;
    }

    /* al::Buffer storage behind a VoiceBufferItem (InitVoice, al/source.cpp:639-670): registered when a voice first starts on
     * it; an item whose storage, length, format or loop points changed since -- a buffer deleted and another allocated at the
     * same address, alBufferData on it -- gives its old copy up (the library frees it when the last voice playing it lets go)
     * and is registered anew */
    BufferEntry *bufferEntry(const Voice *voice, const VoiceBufferItem *item)
    {
        BufferKey key;
        key.kind = item->mSamples.index();
        key.data = std::visit([](auto const &s) -> const void* { return s.data(); }, item->mSamples);
        key.frameStep = voice->mFrameStep; key.sampleLen = item->mSampleLen; key.loopStart = item->mLoopStart; key.loopEnd = item->mLoopEnd;
        key.blockAlign = item->mBlockAlign;
        auto hb = mBuffers.find(item);
        if(hb != mBuffers.end())
        {
            if(hb->second.key == key) return &hb->second;
            releaseEntry(hb->second);
            mBuffers.erase(hb);
        }
        /* SampleVariant (core/buffer_storage.h:35-43) and oalgpu_fmt_type list the PCM types in the same order */
        int h;
        if(key.kind <= size_t(OALGPU_FMT_ALAW))
            h = oalgpu_buffer_register(mGpu, key.data, int(key.kind), key.frameStep, key.sampleLen, key.loopStart, key.loopEnd);
        else
            h = oalgpu_buffer_register_adpcm(mGpu, key.data, key.kind == 7 ? OALGPU_ADPCM_IMA4 : OALGPU_ADPCM_MS, key.frameStep,
                key.blockAlign, key.sampleLen, key.loopStart, key.loopEnd);
        if(h < 0) { fail(h, "oalgpu_buffer_register"); return nullptr; }
        BufferEntry be; be.key = key; be.handle = h;
        return &mBuffers.emplace(item, std::move(be)).first->second;
    }
    /* the buffer a channel's device voice reads: the item's copy itself (mono data, or a mono source whose samples feed both of
     * its channels: mDuplicateMono, voice.cpp:1058-1059), or one channel of interleaved frames */
    int channelBuffer(BufferEntry &be, unsigned channel, bool wholeBuffer)
    {
        if(wholeBuffer) return be.handle;
        if(be.views.size() <= channel) be.views.resize(channel + 1u, -1);
        if(be.views[channel] < 0)
        {
            const int v = oalgpu_buffer_channel_view(mGpu, be.handle, channel);
            if(v < 0) return fail(v, "oalgpu_buffer_channel_view");
            be.views[channel] = v;
        }
        return be.views[channel];
    }

    int slotOf(std::span<EffectSlotBase*const> auxslots, std::span<FloatBufferLine> target) const
    {
        for(size_t s{0}; s < auxslots.size(); ++s)
            if(auxslots[s]->Wet.Buffer.data() == target.data()) return int(s);
        return -1;
    }

    /* what Voice::mix does when a voice reaches the end of its buffer (core/voice.cpp:1201-1232) */
    static void endOfSource(Voice *voice, ContextBase *context)
    {
        auto const sourceID = voice->mSourceID.load(std::memory_order_relaxed);
        voice->mCurrentBuffer.store(nullptr, std::memory_order_release);
        voice->mLoopBuffer.store(nullptr, std::memory_order_relaxed);
        voice->mSourceID.store(0u, std::memory_order_release);
        voice->mPlayState.store(Voice::Stopping, std::memory_order_release);
        if(context->mEnabledEvts.load(std::memory_order_acquire).test(AsyncEnableBits::SourceState))
        {   /* SendSourceStoppedEvent, core/voice.cpp:195-208 */
            auto *ring = context->mAsyncEvents.get();
            if(auto const evt_vec = ring->getWriteVector(); !evt_vec[0].empty())
            {
                auto &evt = InitAsyncEvent<AsyncSourceStateEvent>(evt_vec[0].front());
                evt.mId = sourceID;
                evt.mState = AsyncSrcState::Stop;
                ring->writeAdvance(1);
            }
        }
    }

    void dropEntry(Entry &e)
    {
        for(Chan &ch : e.chans)
        {   /* the device voice lets go of its buffer (a released one is freed then) and the slot goes back */
            (void)oalgpu_voice_set_state(mGpu, ch.index, OALGPU_VOICE_STOPPED);
            if(!mEntryOfIndex.empty()) mEntryOfIndex[ch.index] = -1;
            mFreeIndex.push_back(ch.index);
        }
        e = Entry{};
    }

    /* Buffers a streaming source has played through leave its queue on the device context as well (oalgpu_voice_queue_unqueue:
     * the voice's hold moves on, a buffer the application has released among them is freed) and the front of the chain: without
     * this a long-running stream kept every buffer it had ever queued alive, until the buffer table was full. */
    void leaveBehind(Entry &e, uint32_t done)
    {
        if(!e.queue || done == 0 || e.chain.size() < 2) return;
        const uint32_t n = done < e.chain.size() - 1 ? done : uint32_t(e.chain.size() - 1);
        for(const Chan &ch : e.chans)
            if(oalgpu_voice_queue_unqueue(mGpu, ch.index, n) != 0) return;      /* (what the device context does not know as played stays) */
        e.chain.erase(e.chain.begin(), e.chain.begin() + n);
    }

    /* a streaming source's queue (VoiceBufferItem::mNext): items the voice has not been linked through yet are registered
     * and linked behind the last one (alSourceQueueBuffers appends while the source plays) */
    int extendChain(const Voice *voice, Entry &e)
    {
        const VoiceBufferItem *last = e.chain.back().first;
        while(const VoiceBufferItem *next = last->mNext.load(std::memory_order_acquire))
        {
            BufferEntry *be = bufferEntry(voice, next);
            if(!be) return mError;
            if(int rc = oalgpu_buffer_queue_link(mGpu, e.chain.back().second, be->handle)) return fail(rc, "oalgpu_buffer_queue_link");
            e.chain.emplace_back(next, be->handle);
            last = next;
        }
        return 0;
    }

    int flush(ContextBase *context, DeviceBase &dev, std::chrono::nanoseconds deviceTime, unsigned samplesToDo)
    {
        if(!mGpu) { if(int rc = createContext(context, dev)) return rc; }
        auto const auxspan = std::span{*context->mActiveAuxSlots.load(std::memory_order_acquire)};
        auto const auxslots = auxspan.first(auxspan.size() >> 1);
        if(dev.NumAuxSends && auxslots.size() > mNumSlots)
            return failText(OALGPU_ERR_CAPACITY, "oalgpu_openal: more active effect slots than the device context was created for");
        if(mResync)
        {   /* an update went to the CPU loop: the voices moved on there; every entry starts over from its Voice */
            for(Entry &e : mEntries) if(e.live) dropEntry(e);
            mResync = false;
        }
        if(mEntries.size() < context->getVoicesSpanAcquired().size()) mEntries.resize(context->getVoicesSpanAcquired().size());

        /* ---- pass 1: what the library cannot mix sends the whole update to the CPU loop BEFORE anything on the device changed */
        for(auto &[voice, vstate] : mBatch)
        {
            if(voice->mDecoder)
                return failText(OALGPU_ERR_INVALID, "oalgpu_openal: UHJ sources are not batched");
            if(voice->mFlags.test(VoiceFlag::IsCallback) && (mDepth || voice->mFmtChannels != FmtMono || voice->mSamplesPerBlock != 1u))
                return failText(OALGPU_ERR_INVALID, "oalgpu_openal: callback sources are batched as mono PCM sources, and not in the pipelined mode "
                    "(their state is mirrored back after every update)");
            if(voice->mFlags.test(VoiceFlag::HasNfc) && !mNfc)
                return failText(OALGPU_ERR_INVALID, "oalgpu_openal: a near-field-compensated voice on a device context without near-field control");
            const bool hrtfVoice = voice->mFlags.test(VoiceFlag::HasHrtf);
            if(hrtfVoice != mHrtf || (!mHrtf && voice->mDirect.Buffer.data() != dev.Dry.Buffer.data()))
                return failText(OALGPU_ERR_INVALID, "oalgpu_openal: direct-channel voices, and B-Format sources on an HRTF device, are not batched");
            if(!voice->mFlags.test(VoiceFlag::IsStatic))
            {
                if(voice->mFmtChannels != FmtMono || voice->mDuplicateMono)
                    return failText(OALGPU_ERR_INVALID, "oalgpu_openal: streaming sources are batched as mono sources only");
            }
        }

        ++mUpdateNo;
        auto const tWalk = std::chrono::steady_clock::now();
        std::vector<uint32_t> ids, tgtIds, tgtDelays;
        std::vector<oalgpu_voice_move> moves;         /* hookDirections: directions instead of blended responses */
        std::vector<oalgpu_voice_params> params;
        std::vector<float> tgtCoeffs, tgtGains;
        mMixed.clear();
        for(size_t bi{0}; bi < mBatch.size(); ++bi)
        {
            Voice *voice = mBatch[bi].first;
            const Voice::State vstate = mBatch[bi].second;
            Entry &e = mEntries[mBatchIndex[bi]];
            const unsigned sourceId = voice->mSourceID.load(std::memory_order_relaxed);
            if(e.live && e.sourceId != sourceId && sourceId != 0u)
                dropEntry(e);                       /* the Voice object now plays another source (voices are pooled) */
            /* ---- a delayed start (voice.cpp:1023-1046): untouched until the update it starts in */
            unsigned outPos = 0u;
            if(voice->mStartTime > deviceTime)
            {
                if(vstate == Voice::Stopping)
                {
                    voice->mPlayState.store(Voice::Stopped, std::memory_order_release);
                    if(e.live) dropEntry(e);
                    continue;
                }
                auto const diff = voice->mStartTime - deviceTime;
                if(diff >= std::chrono::seconds{1}) continue;
                outPos = static_cast<unsigned>(std::chrono::round<std::chrono::seconds>(diff * dev.mSampleRate).count());
                if(outPos >= samplesToDo) continue;
            }
            if(voice->mStep < 1u)
            {   /* voice.cpp:1002-1010 */
                if(vstate == Voice::Stopping) { voice->mPlayState.store(Voice::Stopped, std::memory_order_release); if(e.live) dropEntry(e); }
                continue;
            }
            const size_t nch = (voice->mFmtChannels == FmtMono && !voice->mDuplicateMono) ? 1u : voice->mChans.size();
            bool started = false;
            if(!e.live)
            {   /* a voice that starts playing: InitVoice (al/source.cpp:639-670) */
                auto *item = voice->mCurrentBuffer.load(std::memory_order_relaxed);
                if(!item)
                {   /* nothing the device context ever saw: a voice told to stop before its first mix (voice.cpp:1002-1010) */
                    if(vstate == Voice::Stopping) voice->mPlayState.store(Voice::Stopped, std::memory_order_release);
                    continue;
                }
                if(mFreeIndex.size() < nch) return failText(OALGPU_ERR_CAPACITY, "oalgpu_openal: more playing voices than max_voices");
                if(voice->mFlags.test(VoiceFlag::IsCallback))
                {   /* the samples come from the buffer's function (voice.cpp:726-752): the library calls it, through the trampoline, from
                     * inside oalgpu_mix_update, and mirrors mNumCallbackBlocks / mCallbackBlockOffset / CallbackStopped */
                    e.live = true; e.voice = voice; e.sourceId = sourceId; e.lastState = int(Voice::Playing);
                    e.queue = false; e.callback = true; e.doneSeen = 0;
                    Chan ch;
                    ch.index = mFreeIndex.back(); mFreeIndex.pop_back();
                    if(int rc = oalgpu_voice_init_callback(mGpu, ch.index, int(item->mSamples.index()), voice->mPositionFrac.load(std::memory_order_relaxed),
                        &CallbackTrampoline, const_cast<VoiceBufferItem*>(item)))
                        return fail(rc, "oalgpu_voice_init_callback");
                    e.chans.push_back(ch);
                    started = true;
                }
                else
                {
                BufferEntry *be = bufferEntry(voice, item);
                if(!be) return mError;
                e.live = true; e.voice = voice; e.sourceId = sourceId; e.lastState = int(Voice::Playing);
                e.queue = !voice->mFlags.test(VoiceFlag::IsStatic);
                auto *loop = voice->mLoopBuffer.load(std::memory_order_relaxed);
                if(e.queue)
                {
                    if(loop && loop != item)
                        return failText(OALGPU_ERR_INVALID, "oalgpu_openal: a looping queue entered behind its first buffer is not batched");
                    e.chain.assign(1, {item, be->handle});
                    if(int rc = extendChain(voice, e)) return rc;
                }
                for(size_t c{0}; c < nch; ++c)
                {
                    Chan ch;
                    ch.index = mFreeIndex.back(); mFreeIndex.pop_back();
                    if(!mEntryOfIndex.empty() && c == 0) { mEntryOfIndex[ch.index] = int32_t(mBatchIndex[bi]); mIndexBorn[ch.index] = mUpdateNo; }
                    const int h = channelBuffer(*be, unsigned(c), voice->mFmtChannels == FmtMono);
                    if(h < 0) return h;
                    int rc;
                    if(e.queue)
                        rc = oalgpu_voice_init_queue(mGpu, ch.index, h, loop != nullptr, voice->mPosition.load(std::memory_order_relaxed),
                            voice->mPositionFrac.load(std::memory_order_relaxed));
                    else
                    {
                        oalgpu_voice_desc vd{h, loop != nullptr, voice->mPosition.load(std::memory_order_relaxed),
                            voice->mPositionFrac.load(std::memory_order_relaxed), voice->mFrequency};
                        rc = oalgpu_voice_init(mGpu, ch.index, &vd);
                    }
                    if(rc) return fail(rc, "oalgpu_voice_init");
                    if(voice->mFlags.test(VoiceFlag::IsAmbisonic))
                    {   /* Voice::prepare's splitter and scales for this channel (voice.cpp:1353-1380) */
                        auto const &cd = voice->mChans[c];
                        if(int rc2 = oalgpu_voice_set_ambi_scale(mGpu, ch.index, dev.mXOverFreq / float(dev.mSampleRate), cd.mAmbiHFScale, cd.mAmbiLFScale))
                            return fail(rc2, "oalgpu_voice_set_ambi_scale");
                    }
                    e.chans.push_back(ch);
                }
                e.doneSeen = 0;
                started = true;
                }
            }
            else if(e.queue) { if(int rc = extendChain(voice, e)) return rc; }
            mMixed.emplace_back(voice, &e);
            if(e.lastState != int(vstate))
            {   /* ProcessVoiceChanges' play-state changes (alu.cpp:2057-2151) */
                for(Chan &ch : e.chans)
                    if(int rc = oalgpu_voice_set_state(mGpu, ch.index, int(vstate))) return fail(rc, "oalgpu_voice_set_state");
                e.lastState = int(vstate);
            }
            if(outPos)
                for(Chan &ch : e.chans)
                    if(int rc = oalgpu_voice_set_start_delay(mGpu, ch.index, outPos)) return fail(rc, "oalgpu_voice_set_start_delay");
            /* what CalcVoiceParams left in the Voice (alu.cpp:1512-1710, :2012-2031); with the maintainer's hook only for the
             * voices it ran for */
            if(mTrack && !started && !mCompareAll && !mBatchNamed[bi] && !mChanged.count(voice)) continue;
            const float inv_rate = 1.0f / float(dev.mSampleRate);
            for(size_t c{0}; c < nch; ++c)
            {
                Chan &ch = e.chans[c];
                auto &chan = voice->mChans[c];
                if(mNfc)
                {   /* NFCtrlFilter.adjust(w0) and VoiceFlag::HasNfc (alu.cpp:919-941, :1328-1341): a voice without the flag mixes its
                     * samples onto every line unfiltered (voice.cpp:959-963) */
                    const bool has = voice->mFlags.test(VoiceFlag::HasNfc);
                    const float w0 = has ? NfcW0(chan.mDryParams.NFCtrlFilter) : -1.0f;
                    if(!ch.haveNfc || ch.w0 != w0)
                    {
                        if(!has && ch.haveNfc && ch.w0 >= 0.0f)
                            return failText(OALGPU_ERR_INVALID, "oalgpu_openal: a voice that loses VoiceFlag::HasNfc while it plays is not batched");
                        if(has) { if(int rc = oalgpu_voice_set_nfc(mGpu, ch.index, w0)) return fail(rc, "oalgpu_voice_set_nfc"); }
                        ch.haveNfc = true; ch.w0 = w0;
                    }
                }
                oalgpu_voice_params p{};
                p.step = voice->mStep;
                p.resampler = int(voice->mProps.mResampler);
                p.direct_filter.active = voice->mDirect.FilterActive ? 1 : 0;
                p.direct_filter.hf_norm = voice->mProps.Direct.HFReference * inv_rate;
                p.direct_filter.lf_norm = voice->mProps.Direct.LFReference * inv_rate;
                p.direct_filter.gain_hf = voice->mDirect.FilterActive ? ShelfGainAt(chan.mDryParams.LowPass, -1.0f) : 1.0f;
                p.direct_filter.gain_lf = voice->mDirect.FilterActive ? ShelfGainAt(chan.mDryParams.HighPass, 1.0f) : 1.0f;
                /* an HRTF device without sends: that is all of it -- seven words against the ones sent last time, not the whole
                 * record (a moved source's direction is the move record's below) */
                const bool lean = mHrtf && dev.NumAuxSends == 0u;
                if(lean && ch.haveParams && ch.params.step == p.step && ch.params.resampler == p.resampler
                    && std::memcmp(&ch.params.direct_filter, &p.direct_filter, sizeof(p.direct_filter)) == 0)
                    goto params_unchanged;
                if(mHrtf) p.hrtf_dist = OALGPU_HRTF_KEEP_TARGET;       /* Hrtf.Target itself is handed over below */
                else for(size_t l{0}; l < dev.Dry.Buffer.size(); ++l) p.dry_gains[l] = chan.mDryParams.Gains.Target[l];
                for(unsigned s{0}; s < unsigned(OALGPU_MAX_SENDS); ++s)
                {
                    p.send_slot[s] = -1;
                    p.send_filter[s] = oalgpu_filter_params{0, 1.0f, 5000.0f*inv_rate, 1.0f, 250.0f*inv_rate};
                    if(s >= dev.NumAuxSends || voice->mSend[s].Buffer.empty()) continue;
                    /* mSend[s].Buffer = the slot's Wet.Buffer (alu.cpp:1676-1683, :1729-1736) */
                    p.send_slot[s] = slotOf(auxslots, voice->mSend[s].Buffer);
                    if(p.send_slot[s] < 0) return failText(OALGPU_ERR_INVALID, "oalgpu_openal: a send targets a buffer that is no active slot's");
                    auto &wet = chan.mWetParams[s];
                    const bool on = voice->mSend[s].FilterActive;
                    p.send_filter[s].active = on ? 1 : 0;
                    p.send_filter[s].hf_norm = voice->mProps.Send[s].HFReference * inv_rate;
                    p.send_filter[s].lf_norm = voice->mProps.Send[s].LFReference * inv_rate;
                    p.send_filter[s].gain_hf = on ? ShelfGainAt(wet.LowPass, -1.0f) : 1.0f;
                    p.send_filter[s].gain_lf = on ? ShelfGainAt(wet.HighPass, 1.0f) : 1.0f;
                    for(uint32_t l{0}; l < mWetChannels; ++l) p.send_gains[s][l] = wet.Gains.Target[l];
                }
                /* only what changed goes over: BiquadInterpFilter::setParams with unchanged targets is a no-op (biquad.cpp:131-149) */
                if(!ch.haveParams || std::memcmp(&ch.params, &p, sizeof(p)) != 0)
                {
                    ch.params = p; ch.haveParams = true;
                    ids.push_back(ch.index);
                    params.push_back(p);
                }
                params_unchanged:
                const DirNote *note = nullptr;
                for(uint32_t k{0}; k < mBatchDirs[bi].second; ++k)
                    if(mDirList[mBatchDirs[bi].first + k].chan == c) note = &mDirList[mBatchDirs[bi].first + k];
                if(note)
                {   /* the getCoeffs hook took this channel's direction: 24 bytes go over, the device context blends */
                    ch.dir = note->dir; ch.dirStale = true;
                    moves.push_back(oalgpu_voice_move{ch.index, ch.dir[0], ch.dir[1], ch.dir[2], ch.dir[3], chan.mDryParams.Hrtf.Target.Gain});
                    ch.haveTarget = false;              /* (what the Voice holds is not what the device context has) */
                }
                else if(mHrtf && ch.dirStale)
                {   /* (parameters recomputed without a new direction do not exist -- CalcPanningAndFilters always asks for the
                     * response -- but a comparison of every voice does come here: the device context has the right target) */
                }
                else if(mHrtf)
                {
                    auto const &tg = chan.mDryParams.Hrtf.Target;
                    if(!ch.haveTarget || tg.Gain != ch.target.Gain || tg.Delay[0] != ch.target.Delay[0] || tg.Delay[1] != ch.target.Delay[1]
                        || std::memcmp(&tg.Coeffs, &ch.target.Coeffs, sizeof(tg.Coeffs)) != 0)
                    {
                        ch.target = tg; ch.haveTarget = true;
                        tgtIds.push_back(ch.index);
                        tgtDelays.push_back(tg.Delay[0]); tgtDelays.push_back(tg.Delay[1]);
                        tgtGains.push_back(tg.Gain);
                        tgtCoeffs.insert(tgtCoeffs.end(), &tg.Coeffs[0][0], &tg.Coeffs[0][0] + HrirLength*2);
                    }
                }
            }
        }
        mTimes[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - tWalk).count();
        if(!ids.empty())
            if(int rc = oalgpu_voice_set_params(mGpu, ids.data(), params.data(), ids.size())) return fail(rc, "oalgpu_voice_set_params");
        if(!tgtIds.empty())
            if(int rc = oalgpu_voice_set_hrtf_targets(mGpu, tgtIds.data(), tgtCoeffs.data(), tgtDelays.data(), tgtGains.data(), tgtIds.size()))
                return fail(rc, "oalgpu_voice_set_hrtf_targets");
        if(!moves.empty())
            if(int rc = oalgpu_voice_move_async(mGpu, moves.data(), moves.size())) return fail(rc, "oalgpu_voice_move_async");

        /* (pipelined mode: the update is submitted with its post-process and nothing is read back here, see submitPipelined) */
        if(mDepth) return submitPipelined(context, dev, samplesToDo);

        /* the voice loop: one batched update; its buses join the reference's mixing buffers */
        if(int rc = oalgpu_mix_update(mGpu, samplesToDo, 0)) return fail(rc, "oalgpu_mix_update");
        if(!mHrtf)
        {
            mLines.resize((dev.Dry.Buffer.size() + dev.RealOut.Buffer.size()) * BufferLineSize);
            if(int rc = oalgpu_read_dry(mGpu, mLines.data())) return fail(rc, "oalgpu_read_dry");
            for(size_t c{0}; c < dev.Dry.Buffer.size(); ++c)
                for(size_t i{0}; i < samplesToDo; ++i)
                    dev.Dry.Buffer[c][i] += mLines[c*BufferLineSize + i];
        }
        else
        {   /* DoHrtfMix's target (voice.cpp:832): frames [0, samplesToDo + IrSize) of this update's contributions */
            mLines.resize((BufferLineSize + HrirLength) * 2);
            if(int rc = oalgpu_read_hrtf_accum(mGpu, mLines.data())) return fail(rc, "oalgpu_read_hrtf_accum");
            const size_t frames = std::min<size_t>(samplesToDo + dev.mIrSize, dev.HrtfAccumData.size());
            for(size_t i{0}; i < frames; ++i)
            {
                dev.HrtfAccumData[i][0] += mLines[i*2 + 0];
                dev.HrtfAccumData[i][1] += mLines[i*2 + 1];
            }
        }
        if(dev.NumAuxSends)
        {
            mLines.resize(size_t{mWetChannels} * BufferLineSize);
            for(size_t s{0}; s < auxslots.size(); ++s)
            {
                if(int rc = oalgpu_read_wet(mGpu, uint32_t(s), mLines.data())) return fail(rc, "oalgpu_read_wet");
                auto wet = auxslots[s]->Wet.Buffer;
                for(size_t c{0}; c < wet.size(); ++c)
                    for(size_t i{0}; i < samplesToDo; ++i)
                        wet[c][i] += mLines[c*BufferLineSize + i];
            }
        }

        /* the state the reference mutates in place stays authoritative on the device; what the rest of the
         * reference looks at (GetSourceOffset, the play state, the end of a source, a queue's progress) is read back from the
         * voice's first channel */
        std::vector<uint32_t> rb;
        for(auto &ve : mMixed) rb.push_back(ve.second->chans[0].index);
        mBrief.resize(rb.size());
        if(!rb.empty())
            if(int rc = oalgpu_voices_readback(mGpu, rb.data(), rb.size(), mBrief.data())) return fail(rc, "oalgpu_voices_readback");
        for(size_t k{0}; k < mMixed.size(); ++k)
        {
            Voice *voice = mMixed[k].first;
            Entry &e = *mMixed[k].second;
            const oalgpu_voice_brief &st = mBrief[k];
            if(st.fading) voice->mFlags.set(VoiceFlag::IsFading);
            if(e.lastState == int(Voice::Stopping))
            {   /* voice.cpp:1119-1123: faded out; the Voice object returns to the pool */
                voice->mPlayState.store(Voice::Stopped, std::memory_order_release);
                dropEntry(e);
                continue;
            }
            if(e.callback)
            {   /* voice.cpp:1155-1180: what the update left of the callback storage (the device-side position is relative to it) */
                oalgpu_callback_state cs{};
                if(int rc = oalgpu_voice_callback_state(mGpu, e.chans[0].index, &cs)) return fail(rc, "oalgpu_voice_callback_state");
                voice->mPosition.store(cs.position, std::memory_order_relaxed);
                voice->mPositionFrac.store(cs.position_frac, std::memory_order_relaxed);
                voice->mNumCallbackBlocks = cs.num_blocks;
                voice->mCallbackBlockOffset = cs.block_offset;
                voice->mFlags.set(VoiceFlag::CallbackStopped, cs.stopped != 0);
                if(!cs.has_buffer)
                {
                    endOfSource(voice, context);
                    e.lastState = int(Voice::Stopping);
                }
                continue;
            }
            voice->mPosition.store(st.position, std::memory_order_relaxed);
            voice->mPositionFrac.store(st.position_frac, std::memory_order_relaxed);
            if(e.queue)
            {   /* voice.cpp:1182-1218: where the queue has got to, and the buffers it left behind */
                const unsigned sourceID = voice->mSourceID.load(std::memory_order_relaxed);
                if(st.has_buffer)
                    for(auto const &link : e.chain)
                        if(link.second == st.current_buffer)
                        { voice->mCurrentBuffer.store(const_cast<VoiceBufferItem*>(link.first), std::memory_order_release); break; }
                const uint32_t done = st.buffers_done - e.doneSeen;
                e.doneSeen = st.buffers_done;
                leaveBehind(e, done);
                if(done > 0 && context->mEnabledEvts.load(std::memory_order_acquire).test(AsyncEnableBits::BufferCompleted))
                {
                    auto *ring = context->mAsyncEvents.get();
                    if(auto const evt_vec = ring->getWriteVector(); !evt_vec[0].empty())
                    {
                        auto &evt = InitAsyncEvent<AsyncBufferCompleteEvent>(evt_vec[0].front());
                        evt.mId = sourceID;
                        evt.mCount = done;
                        ring->writeAdvance(1);
                    }
                }
            }
            if(!st.has_buffer)
            {
                endOfSource(voice, context);
                e.lastState = int(Voice::Stopping);     /* (the device context set it itself) */
            }
        }
        return 0;
    }

    /* ---- the pipelined mode: the update is submitted; what an earlier one produced is collected ---- */
    struct Pending { uint32_t outTicket, evTicket; unsigned samples; uint64_t update; };
    int submitPipelined(ContextBase *context, DeviceBase &dev, unsigned samplesToDo)
    {
        auto const t0 = std::chrono::steady_clock::now();
        if(int rc = oalgpu_mix_update(mGpu, samplesToDo, 1)) return fail(rc, "oalgpu_mix_update");
        /* a voice the reference told to stop fades out in this very update (voice.cpp:1119-1123): its Voice returns to the pool
         * now, its device slot behind the update (the calls below are queued behind it) */
        for(auto &ve : mMixed)
            if(ve.second->live && ve.second->lastState == int(Voice::Stopping))
            {
                ve.first->mPlayState.store(Voice::Stopped, std::memory_order_release);
                dropEntry(*ve.second);
            }
        Pending p{0u, 0u, samplesToDo, mUpdateNo};
        if(int rc = oalgpu_read_output_async(mGpu, &p.outTicket)) return fail(rc, "oalgpu_read_output_async");
        if(int rc = oalgpu_voice_events_async(mGpu, &p.evTicket)) return fail(rc, "oalgpu_voice_events_async");
        mPending.push_back(p);
        auto const t1 = std::chrono::steady_clock::now();
        /* the update `depth` back: its output lines and what it changed about the voices have long landed in pinned memory */
        while(mPending.size() > mDepth) { if(int rc = collect(context, dev)) return rc; }
        mTimes[1] += std::chrono::duration<double>(t1 - t0).count();
        mTimes[2] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count();
        return 0;
    }
    int collect(ContextBase *context, DeviceBase &dev, bool dropOutput = false)
    {
        const Pending p = mPending.front();
        mPending.erase(mPending.begin());
        /* the update's output lines: HrtfPostProcess's result (alu.cpp:289-298), `depth` updates late */
        const size_t nreal = dev.RealOut.Buffer.size();
        mLines.resize(nreal * BufferLineSize);
        if(int rc = oalgpu_output_wait(mGpu, p.outTicket, mLines.data(), mLines.size())) return fail(rc, "oalgpu_output_wait");
        for(size_t c{0}; c < nreal && !dropOutput; ++c)
            for(size_t i{0}; i < p.samples; ++i)
                dev.RealOut.Buffer[c][i] += mLines[c*BufferLineSize + i];
        /* what changed about the voices in that update */
        mEvents.resize(1024);
        size_t count{0};
        const int rc = oalgpu_voice_events_wait(mGpu, p.evTicket, mEvents.data(), mEvents.size(), &count);
        if(rc == OALGPU_ERR_CAPACITY) return resyncStates(context);
        if(rc) return fail(rc, "oalgpu_voice_events_wait");
        auto const voices = context->getVoicesSpanAcquired();
        for(size_t k{0}; k < count; ++k)
        {
            const oalgpu_voice_event &ev = mEvents[k];
            if(ev.voice >= mEntryOfIndex.size() || mEntryOfIndex[ev.voice] < 0 || mIndexBorn[ev.voice] > p.update) continue;
            const size_t place = size_t(mEntryOfIndex[ev.voice]);
            if(place >= mEntries.size() || place >= voices.size()) continue;
            Entry &e = mEntries[place];
            if(!e.live || e.chans.empty() || e.chans[0].index != ev.voice) continue;
            applyState(voices[place], e, context, ev.fading != 0, ev.position, ev.position_frac, ev.has_buffer != 0, ev.current_buffer, ev.buffers_done);
        }
        return 0;
    }
    /* more changes than a report holds (hundreds of sources ending in one update): every live voice is read back */
    int resyncStates(ContextBase *context)
    {
        auto const voices = context->getVoicesSpanAcquired();
        std::vector<uint32_t> rb; std::vector<size_t> places;
        for(size_t i{0}; i < mEntries.size() && i < voices.size(); ++i)
            if(mEntries[i].live && !mEntries[i].chans.empty()) { rb.push_back(mEntries[i].chans[0].index); places.push_back(i); }
        mBrief.resize(rb.size());
        if(!rb.empty())
            if(int rc = oalgpu_voices_readback(mGpu, rb.data(), rb.size(), mBrief.data())) return fail(rc, "oalgpu_voices_readback");
        for(size_t k{0}; k < rb.size(); ++k)
        {
            const oalgpu_voice_brief &st = mBrief[k];
            applyState(voices[places[k]], mEntries[places[k]], context, st.fading != 0, st.position, st.position_frac, st.has_buffer != 0,
                st.current_buffer, st.buffers_done);
        }
        return 0;
    }
    /* the state Voice::mix leaves in the Voice after an update (voice.cpp:1116-1232), from what the device context reports */
    void applyState(Voice *voice, Entry &e, ContextBase *context, bool fading, int32_t position, uint32_t positionFrac, bool hasBuffer,
        int32_t currentBuffer, uint32_t buffersDone)
    {
        if(fading) voice->mFlags.set(VoiceFlag::IsFading);
        if(e.lastState == int(Voice::Stopping)) return;          /* (on its way out: the Voice is not touched any more) */
        voice->mPosition.store(position, std::memory_order_relaxed);
        voice->mPositionFrac.store(positionFrac, std::memory_order_relaxed);
        if(e.queue)
        {   /* voice.cpp:1182-1218: where the queue has got to, and the buffers it left behind */
            const unsigned sourceID = voice->mSourceID.load(std::memory_order_relaxed);
            if(hasBuffer)
                for(auto const &link : e.chain)
                    if(link.second == currentBuffer)
                    { voice->mCurrentBuffer.store(const_cast<VoiceBufferItem*>(link.first), std::memory_order_release); break; }
            /* (a report that is older than what a resynchronisation has read in between: nothing new in it) */
            const int32_t ahead = int32_t(buffersDone - e.doneSeen);
            const uint32_t done = ahead > 0 ? uint32_t(ahead) : 0u;
            if(ahead > 0) e.doneSeen = buffersDone;
            leaveBehind(e, done);
            if(done > 0 && context->mEnabledEvts.load(std::memory_order_acquire).test(AsyncEnableBits::BufferCompleted))
            {
                auto *ring = context->mAsyncEvents.get();
                if(auto const evt_vec = ring->getWriteVector(); !evt_vec[0].empty())
                {
                    auto &evt = InitAsyncEvent<AsyncBufferCompleteEvent>(evt_vec[0].front());
                    evt.mId = sourceID;
                    evt.mCount = done;
                    ring->writeAdvance(1);
                }
            }
        }
        if(!hasBuffer)
        {
            endOfSource(voice, context);
            e.lastState = int(Voice::Stopping);     /* (the device context set it itself) */
        }
    }

    int mMathMode, mDevice;
    unsigned mMaxVoices;
    unsigned mDepth{0};
    double mTimes[3]{0.0, 0.0, 0.0};
    uint64_t mUpdateNo{0};
    std::vector<Pending> mPending;
    std::vector<int32_t> mEntryOfIndex;             /* [device voice] -> place of its entry (pipelined mode) */
    std::vector<uint64_t> mIndexBorn;               /* [device voice] the update it was initialised in */
    std::vector<oalgpu_voice_event> mEvents;
    HrtfPostProcess mSavedPost;
    std::vector<oalgpu_reverb*> mReverbs;           /* the pipelined mode's EAX reverb instances, one per bound effect slot */
    AmbiDecPostProcess mSavedDec;                   /* (a RenderMode::Normal device's, while its decode runs behind the boundary) */
    oalgpu_context *mGpu{nullptr};
    bool mHrtf{false};
    bool mNfc{false};                               /* the device context has near-field control (DeviceBase::AvgSpeakerDist > 0) */
    uint32_t mNumSlots{0}, mWetChannels{4};
    std::map<const VoiceBufferItem*, BufferEntry> mBuffers;
    std::vector<Entry> mEntries;                    /* [place in the context's voice array] */
    std::vector<uint32_t> mFreeIndex;
    unsigned mExpected{0}, mSeen{0};
    size_t mCursor{0};
    std::vector<std::pair<Voice*, Voice::State>> mBatch;
    std::vector<uint32_t> mBatchIndex;
    std::vector<std::pair<Voice*, Entry*>> mMixed;
    bool mTrack{false}, mResync{false};
    /* hookDirections: per voice, the directions the getCoeffs hook handed over (pending: not yet passed on to the device context;
     * stale: the Voice's own Hrtf.Target does not hold them) */
    struct DirNote { const Voice *voice; std::array<float, 4> dir; uint8_t chan; };
    std::vector<DirNote> mDirList;                  /* this update's directions, in the order of the voice array */
    size_t mDirAt{0};
    std::vector<std::pair<uint32_t, uint32_t>> mBatchDirs;   /* [batch entry] its notes in mDirList: first, count */
    bool mHookDirs{false}, mHrtfKnown{false};
    std::unordered_set<const Voice*> mChanged;
    std::vector<const Voice*> mOrdered;             /* noteParamsChangedInOrder: this update's names, in voice-array order */
    size_t mOrderedAt{0};
    bool mCompareAll{false};
    std::vector<uint8_t> mBatchNamed;               /* [batch entry] the voice was named through mOrdered */
    std::vector<float> mLines;
    std::vector<oalgpu_voice_brief> mBrief;
    int mError{0};
    std::string mErrorText;
};

} // namespace oalgpu_openal
