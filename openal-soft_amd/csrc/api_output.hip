// What comes back: the output ring, voice-event reports, the caller's stream, bus and voice read-backs.
#include "api_context.hpp"

static size_t OutputLineFloats(const oalgpu_context *c)
{
    return size_t{c->L.numReal ? c->L.numReal : c->L.numDry} * kLine;
}

int oalgpu_read_output_async(oalgpu_context *c, uint32_t *ticket)
{
    if(!c || !ticket) return Fail(OALGPU_ERR_INVALID, "null argument");
    const size_t floats = OutputLineFloats(c);
    // (a resident voice kernel stays where it is unless the ring's buffers have yet to be allocated)
    if(int rc = (c->outFloats != floats) ? UseCtx(c) : UseCtxResident(c)) return rc;
    if(c->outFloats != floats)
    {
        for(uint32_t k = 0; k < oalgpu_context::kIoSlots; ++k)
        {
            if(c->outHost[k]) { HIP_TRY(hipHostFree(c->outHost[k])); c->outHost[k] = nullptr; }
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->outHost[k]), floats * sizeof(float), hipHostMallocDefault));
            if(!c->outDone[k]) HIP_TRY(hipEventCreateWithFlags(&c->outDone[k], hipEventDisableTiming));
        }
        c->outFloats = floats;
        if(!c->outFlags)
        {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->outFlags), oalgpu_context::kIoSlots * 64, hipHostMallocDefault));
            std::memset(c->outFlags, 0, oalgpu_context::kIoSlots * 64);
            HIP_TRY(c->outArrived.alloc(1)); HIP_TRY(c->outArrived.zero());
        }
        // from the next update on the post-process kernel fills the slots itself (the fused FAST post-process of an HRTF
        // context with its two output lines; everything else keeps the copy below)
        c->outRing = c->L.hrtf && c->L.numReal == 2 && c->useWave && floats == size_t{2} * kLine;
    }
    const uint32_t slot = c->outNext % oalgpu_context::kIoSlots;
    if(c->outRingWritten)
    {   // the update submitted last is already writing this slot
        c->outRingWritten = false;
        c->outViaRing[slot] = true;
        c->outUpdate[slot] = c->updatesSubmitted;
        *ticket = c->outNext++;
        return OALGPU_OK;
    }
    c->outViaRing[slot] = false;
    c->outUpdate[slot] = c->updatesSubmitted;
    // behind whatever produced the lines: the post stream of a pipelined context, else the main one
    hipStream_t s = (c->useWave && c->ownStream && !c->serialOnly && c->postStream) ? c->postStream : c->stream;
    const float *src = c->L.numReal ? c->L.bus + size_t{c->L.numDry} * kLine : c->L.bus;
    HIP_TRY(hipMemcpyAsync(c->outHost[slot], src, floats * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(c->outDone[slot], s));
    // (a resident context's next reduction runs on a stream of its own and rewrites these lines: it has to wait for the copy)
    if(c->res.running) c->res.copyPending = c->outDone[slot];
    *ticket = c->outNext++;
    return OALGPU_OK;
}

int oalgpu_output_wait(oalgpu_context *c, uint32_t ticket, float *out, size_t out_floats)
{
    if(!c || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    // Four ring slots.  Where the post-process kernel fills the slots itself (outRing), the update submitted AFTER the newest
    // ticket is already writing the slot of the ticket four back -- at submit time, not when its own ticket is drawn -- so only
    // three tickets may be outstanding there; a slot filled by a copy is overwritten only by oalgpu_read_output_async itself.
    const uint32_t live = c->outRing ? oalgpu_context::kIoSlots - 1u : oalgpu_context::kIoSlots;
    if(ticket >= c->outNext || c->outNext - ticket > live)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_output_wait: the ticket's slot was reused (three tickets may be outstanding)");
    if(out_floats < c->outFloats) return Fail(OALGPU_ERR_INVALID, "oalgpu_output_wait: the buffer is smaller than the output lines");
    if(int rc = UseCtxResident(c)) return rc;
    const uint32_t slot = ticket % oalgpu_context::kIoSlots;
    if(c->outViaRing[slot])
    {   // the kernel raises the slot's sequence number behind its lines
        const uint32_t *flag = c->outFlags + size_t{slot} * 16;
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::seconds(10);
        uint32_t spins = 0;
        const uint32_t want = c->outSlotSeq[slot];
        while(__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want)
        {
            __builtin_ia32_pause();
            if((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() > deadline)
            {
                HIP_TRY(hipStreamSynchronize(c->postStream ? c->postStream : c->stream));
                if(__atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) return Fail(OALGPU_ERR_HIP, "oalgpu_output_wait: the output slot was never written");
            }
        }
    }
    else HIP_TRY(hipEventSynchronize(c->outDone[slot]));
    if(c->outUpdate[slot] > c->updatesKnownDone) c->updatesKnownDone = c->outUpdate[slot];
    std::memcpy(out, c->outHost[slot], c->outFloats * sizeof(float));
    return OALGPU_OK;
}

/* What changed about the voices since the last report (the first one reports every voice that is not Stopped): see oalgpu.h */
int oalgpu_voice_events_async(oalgpu_context *c, uint32_t *ticket)
{
    if(!c || !ticket) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    constexpr uint32_t slots = oalgpu_context::kIoSlots;
    if(!c->evHost[0])
    {
        for(uint32_t k = 0; k < slots; ++k)
        {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->evHost[k]), (4u + size_t{oalgpu_context::kEvCap} * 8u) * sizeof(uint32_t), hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&c->evDone[k], hipEventDisableTiming));
        }
        HIP_TRY(c->evSnapshot.alloc(size_t{c->L.numVoices} * 3)); HIP_TRY(c->evSnapshot.zero());       // (0 = Stopped, buffer 0: what a slot never used looks like but for the buffer)
        HIP_TRY(c->evCounters.alloc(2)); HIP_TRY(c->evCounters.zero());
        std::vector<uint32_t> init(size_t{c->L.numVoices} * 3, 0u);
        for(uint32_t v = 0; v < c->L.numVoices; ++v) { init[size_t{v} * 3 + 0] = uint32_t(OALGPU_VOICE_STOPPED); init[size_t{v} * 3 + 1] = 0xffffffffu; }
        HIP_TRY(hipMemcpyAsync(c->evSnapshot.p, init.data(), init.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    const uint32_t slot = c->evNext % slots;
    // (four slots: the ticket four back must have been collected -- its event is long done -- or is given up)
    LaunchVoiceEvents(c->stream, c->L, c->evSnapshot.p, c->evHost[slot], oalgpu_context::kEvCap, c->evCounters.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->evDone[slot], c->stream));
    *ticket = c->evNext++;
    return OALGPU_OK;
}

int oalgpu_voice_events_wait(oalgpu_context *c, uint32_t ticket, oalgpu_voice_event *out, size_t capacity, size_t *count)
{
    if(!c || !count || (!out && capacity)) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(ticket >= c->evNext || c->evNext - ticket > oalgpu_context::kIoSlots)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_events_wait: the ticket's slot was reused (four tickets may be outstanding)");
    if(int rc = UseCtxResident(c)) return rc;
    const uint32_t slot = ticket % oalgpu_context::kIoSlots;
    HIP_TRY(hipEventSynchronize(c->evDone[slot]));
    const uint32_t *h = c->evHost[slot];
    const uint32_t n = h[0];
    *count = n;
    if(n > oalgpu_context::kEvCap || n > capacity)
        return Fail(OALGPU_ERR_CAPACITY, "oalgpu_voice_events_wait: " + std::to_string(n) + " changes, more than the report holds (read the voices back: oalgpu_voices_readback)");
    for(uint32_t i = 0; i < n; ++i)
    {
        const uint32_t *e = h + 4u + size_t{i} * 8u;
        out[i] = oalgpu_voice_event{e[0], int32_t(e[1]), int32_t(e[2]) >= 0 ? 1 : 0, int32_t(e[2]), e[3], int32_t(e[4]), e[5], int32_t(e[6])};
        if(e[0] < c->L.numVoices) c->queueDoneKnown[e[0]] = e[3];      // (what oalgpu_voice_queue_unqueue checks against)
    }
    return OALGPU_OK;
}

int oalgpu_set_stream(oalgpu_context *c, void *hip_stream)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    if(c->postStream) HIP_TRY(hipStreamSynchronize(c->postStream));
    c->postPending = false;
    if(c->ownStream && c->stream) { (void)hipStreamDestroy(c->stream); c->stream = nullptr; }
    if(hip_stream) { c->stream = static_cast<hipStream_t>(hip_stream); c->ownStream = false; }
    else
    {   // back to a private stream: in the highest priority class again, apart from the post stream's
        // hardware queues (see oalgpu_context_create)
        int prioLeast = 0, prioGreatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLeast, &prioGreatest));
        HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prioGreatest));
        c->ownStream = true;
    }
    return OALGPU_OK;
}


int oalgpu_read_dry(oalgpu_context *c, float *out)
{
    if(!c || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = oalgpu_sync(c)) return rc;
    HIP_TRY(hipMemcpy(out, c->L.bus, BusWetOffset(c->L) * sizeof(float), hipMemcpyDeviceToHost));
    return OALGPU_OK;
}

/* BFormatDec(inchans = num_dry_channels, coeffs, coeffslf, xover_f0norm), core/bformatdec.cpp:27-58 */
int oalgpu_set_bformat_decoder(oalgpu_context *c, uint32_t num_out, const float *coeffs_hf, const float *coeffs_lf,
    float xover_norm)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(c->L.hrtf) return Fail(OALGPU_ERR_INVALID, "oalgpu_set_bformat_decoder: an HRTF context post-processes with MixDirectHrtf");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    if(num_out == 0 || !coeffs_hf) { c->decOn = false; return OALGPU_OK; }
    if(num_out > c->L.numReal || num_out > 32u)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_set_bformat_decoder: more output channels than real output lines");
    if(coeffs_lf && !(xover_norm > 0.0f && xover_norm < 0.5f))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_set_bformat_decoder: a dual-band decoder needs 0 < xover_norm < 0.5");
    const uint32_t nin = c->L.numDry;
    // decoder[j].mGains[out] = coeffs[out][j] (bformatdec.cpp:33-38): stored [dry line][32]
    std::vector<float> hf(size_t{nin} * 32, 0.0f), lf(size_t{nin} * 32, 0.0f);
    for(uint32_t j = 0; j < nin && j < OALGPU_MAX_AMBI_CHANNELS; ++j)
        for(uint32_t o = 0; o < num_out; ++o)
        {
            hf[j * 32 + o] = coeffs_hf[size_t{o} * OALGPU_MAX_AMBI_CHANNELS + j];
            if(coeffs_lf) lf[j * 32 + o] = coeffs_lf[size_t{o} * OALGPU_MAX_AMBI_CHANNELS + j];
        }
    HIP_TRY(c->decGainsHf.alloc(hf.size())); HIP_TRY(c->decGainsHf.upload(hf.data(), hf.size()));
    HIP_TRY(c->decGainsLf.alloc(lf.size())); HIP_TRY(c->decGainsLf.upload(lf.data(), lf.size()));
    HIP_TRY(c->decBands.alloc(size_t{nin} * 2 * kLine)); HIP_TRY(c->decBands.zero());
    std::vector<SplitterState> sp(nin);
    for(auto &s : sp) s = SplitterState{coeffs_lf ? SplitterCoeff(xover_norm) : 0.0f, 0.0f, 0.0f, 0.0f};
    HIP_TRY(c->decSplit.alloc(nin)); HIP_TRY(c->decSplit.upload(sp.data(), nin));
    c->decOut = num_out; c->decDual = coeffs_lf != nullptr; c->decOn = true;
    return OALGPU_OK;
}

/* the device's output format: DevFmtType (core/devformat.h:56-64), DitherDepth / DitherSeed (alc/alc.cpp) */
int oalgpu_set_output(oalgpu_context *c, int sample_type, float dither_depth, uint32_t dither_seed)
{
    if(c) { if(int rc = FlushPendingMix(c)) return rc; }
    // (a parameter block that waits for a resident update was applied BEFORE this call: it goes in first, as on the launched path)
    if(c && c->res.pendingBlock) { if(int rc = UseCtx(c)) return rc; }
    if(!c || sample_type < OALGPU_OUT_I8 || sample_type > OALGPU_OUT_F32 || dither_depth < 0.0f)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_set_output: bad arguments");
    c->outType = sample_type; c->ditherDepth = dither_depth; c->ditherSeed = dither_seed;
    return OALGPU_OK;
}

/* ApplyDither + Write<T> (alc/alu.cpp:2309-2408) over the output lines of the last update, then ONE D2H
 * copy of the interleaved PCM: frames [0, samples_to_do) x frame_step samples of the configured type */
int oalgpu_read_output(oalgpu_context *c, void *out, uint32_t samples_to_do, uint32_t frame_step)
{
    static const size_t bytesPer[7] = {1, 1, 2, 2, 4, 4, 4};
    if(!c || !out || samples_to_do == 0 || samples_to_do > kLine || frame_step == 0 || frame_step > 64)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_read_output: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = JoinPost(c)) return rc;
    const DeviceLayout &L = c->L;
    // RealOut: the real output lines, or the dry lines themselves where the device has none (core/device.h:300)
    float *lines = L.numReal ? L.bus + size_t{L.numDry} * kLine : L.bus;
    const uint32_t nlines = std::min(L.numReal ? L.numReal : L.numDry, frame_step);
    const uint32_t all = L.numReal ? L.numReal : L.numDry;
    if(c->ditherDepth > 0.0f)
    {
        LaunchDither(c->stream, lines, all, samples_to_do, c->ditherDepth, c->ditherSeed);
        c->ditherSeed = DitherAdvanceSeed(c->ditherSeed, all * samples_to_do * 2u);
    }
    const size_t nbytes = size_t{samples_to_do} * frame_step * bytesPer[c->outType];
    if(c->pcm.n < nbytes) HIP_TRY(c->pcm.alloc(size_t{kLine} * 64 * 4));
    LaunchWriteSamples(c->stream, c->outType, lines, nlines, samples_to_do, frame_step, c->pcm.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(out, c->pcm.p, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return OALGPU_OK;
}

int oalgpu_read_wet(oalgpu_context *c, uint32_t slot, float *out)
{
    if(!c || !out || slot >= c->L.numSlots) return Fail(OALGPU_ERR_INVALID, "bad slot");
    if(int rc = oalgpu_sync(c)) return rc;
    const size_t n = size_t{c->L.wetChannels} * kLine;
    HIP_TRY(hipMemcpy(out, c->L.bus + BusWetOffset(c->L) + slot * n, n * sizeof(float), hipMemcpyDeviceToHost));
    return OALGPU_OK;
}

int oalgpu_read_hrtf_accum(oalgpu_context *c, float *out)
{
    if(!c || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = oalgpu_sync(c)) return rc;
    // HrtfAccumData as the last update left it: shifted by the post-process (the fused one files it in carryBuf)
    HIP_TRY(hipMemcpy(out, c->carryInBuf ? c->carryBuf.p : c->L.bus + BusAccumOffset(c->L), size_t{kLine + kHrirLen} * 2 * sizeof(float), hipMemcpyDeviceToHost));
    return OALGPU_OK;
}

int oalgpu_bus_device_ptr(oalgpu_context *c, void **ptr, size_t *nfloats, void **hip_stream)
{
    if(c) { if(int rc = FlushPendingMix(c)) return rc; }
    // (a parameter block that waits for a resident update was applied BEFORE this call: it goes in first, as on the launched path)
    if(c && c->res.pendingBlock) { if(int rc = UseCtx(c)) return rc; }
    if(!c || !ptr || !nfloats) return Fail(OALGPU_ERR_INVALID, "null argument");
    *ptr = c->L.bus;
    *nfloats = BusFloats(c->L);
    if(hip_stream)      // the pipelined path produces the bus on the post stream, the serial entry points on the main one
        *hip_stream = (c->useWave && c->ownStream && !c->serialOnly && c->postStream) ? c->postStream : c->stream;
    return OALGPU_OK;
}

int oalgpu_voice_readback(oalgpu_context *c, uint32_t v, oalgpu_voice_state *out)
{
    if(!c || !out || v >= c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_readback: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    const DeviceLayout &L = c->L;
    std::memset(out, 0, sizeof(*out));
    VoiceCtl ctl;
    HIP_TRY(hipMemcpy(&ctl, L.ctl + v, sizeof(ctl), hipMemcpyDeviceToHost));
    out->play_state = ctl.playState; out->position = ctl.position; out->position_frac = ctl.positionFrac;
    out->has_buffer = ctl.curBuffer >= 0; out->fading = (ctl.flags & kFlagFading) != 0;
    out->hrtf_old_gain = ctl.hrtfOldGain;
    out->hrtf_old_delay[0] = ctl.hrtfOldDelay[0]; out->hrtf_old_delay[1] = ctl.hrtfOldDelay[1];
    HIP_TRY(hipMemcpy(out->prev_samples, L.prev + size_t{v} * kMaxPad, sizeof(out->prev_samples), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(out->hrtf_history, L.hist + size_t{v} * kHist, sizeof(out->hrtf_history), hipMemcpyDeviceToHost));
    if(!L.hrtf)
        HIP_TRY(hipMemcpy(out->dry_current, L.gainCur + size_t{v} * L.numDry, L.numDry * sizeof(float), hipMemcpyDeviceToHost));
    BiquadSlot slots[2 * OALGPU_MAX_SENDS];
    HIP_TRY(hipMemcpy(slots, L.dfilt + size_t{v} * 2, 2 * sizeof(BiquadSlot), hipMemcpyDeviceToHost));
    std::memcpy(&out->direct_lp, &slots[0].f, sizeof(oalgpu_biquad));
    std::memcpy(&out->direct_hp, &slots[1].f, sizeof(oalgpu_biquad));
    for(uint32_t s = 0; s < OALGPU_MAX_SENDS; ++s) { oalgpu_biquad_reset(&out->send_lp[s]); oalgpu_biquad_reset(&out->send_hp[s]); }
    if(L.numSends)
    {
        HIP_TRY(hipMemcpy(slots, L.sfilt + size_t{v} * L.numSends * 2, L.numSends * 2 * sizeof(BiquadSlot), hipMemcpyDeviceToHost));
        std::vector<float> cur(size_t{L.numSends} * L.wetChannels);
        HIP_TRY(hipMemcpy(cur.data(), L.sendCur + size_t{v} * L.numSends * L.wetChannels, cur.size() * sizeof(float), hipMemcpyDeviceToHost));
        for(uint32_t s = 0; s < L.numSends; ++s)
        {
            std::memcpy(&out->send_lp[s], &slots[s * 2].f, sizeof(oalgpu_biquad));
            std::memcpy(&out->send_hp[s], &slots[s * 2 + 1].f, sizeof(oalgpu_biquad));
            std::memcpy(out->send_current[s], cur.data() + size_t{s} * L.wetChannels, L.wetChannels * sizeof(float));
        }
    }
    return OALGPU_OK;
}

int oalgpu_voices_readback(oalgpu_context *c, const uint32_t *voices, size_t count, oalgpu_voice_brief *out)
{
    if(!c || !voices || !out) return Fail(OALGPU_ERR_INVALID, "oalgpu_voices_readback: null argument");
    if(count == 0) return OALGPU_OK;
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    // one copy of the control lines the voices span (128 bytes each), not one round trip per voice
    uint32_t lo = 0xffffffffu, hi = 0u;
    for(size_t i = 0; i < count; ++i)
    {
        if(voices[i] >= c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voices_readback: bad voice index");
        lo = std::min(lo, voices[i]); hi = std::max(hi, voices[i]);
    }
    c->ctlHost.resize(size_t{hi - lo} + 1u);
    HIP_TRY(hipMemcpy(c->ctlHost.data(), c->L.ctl + lo, c->ctlHost.size() * sizeof(VoiceCtl), hipMemcpyDeviceToHost));
    for(size_t i = 0; i < count; ++i)
    {
        const VoiceCtl &ctl = c->ctlHost[voices[i] - lo];
        out[i] = oalgpu_voice_brief{ctl.playState, ctl.position, ctl.positionFrac, ctl.curBuffer >= 0 ? 1 : 0,
            (ctl.flags & kFlagFading) ? 1 : 0, ctl.curBuffer, 0u};
    }
    // streaming sources: the buffers they have played through (AsyncBufferCompleteEvent counts, voice.cpp:1207-1218)
    bool anyQueue = false;
    for(size_t i = 0; i < count; ++i) anyQueue = anyQueue || (c->ctlHost[voices[i] - lo].flags & kFlagQueue);
    if(anyQueue)
    {
        c->doneHost.resize(size_t{hi - lo} + 1u);
        HIP_TRY(hipMemcpy(c->doneHost.data(), c->queueDone.p + lo, c->doneHost.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for(size_t i = 0; i < count; ++i)
        {
            out[i].buffers_done = c->doneHost[voices[i] - lo];
            c->queueDoneKnown[voices[i]] = out[i].buffers_done;
        }
    }
    return OALGPU_OK;
}


