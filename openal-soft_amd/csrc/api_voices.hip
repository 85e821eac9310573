// Buffers, voices, parameters: handles and their lifetime, voice starts and queues, parameter records and blocks, moves, pan.
#include "api_context.hpp"

// ---- buffer handles: allocation, references, release ----
static bool BufferLive(const oalgpu_context *c, int h) { return h >= 0 && uint32_t(h) < c->numBuffers && c->bufHost[size_t(h)].live; }
int AllocBufferHandle(oalgpu_context *c, uint32_t *out)
{
    if(!c->freeBuffers.empty()) { *out = c->freeBuffers.back(); c->freeBuffers.pop_back(); }
    else if(c->numBuffers < c->desc.max_buffers) *out = c->numBuffers++;
    else return Fail(OALGPU_ERR_CAPACITY, "buffer table full");
    c->bufHost[*out] = oalgpu_context::BufHost{};
    c->bufHost[*out].live = true;
    return OALGPU_OK;
}
static void BufferUnref(oalgpu_context *c, int h);
// (the caller has made sure nothing on the device still reads the buffer: hipFree waits for the device besides)
static void BufferFreeNow(oalgpu_context *c, int h)
{
    auto &b = c->bufHost[size_t(h)];
    if(c->bufferData[size_t(h)]) { (void)hipFree(c->bufferData[size_t(h)]); c->bufferData[size_t(h)] = nullptr; }
    const int32_t parent = b.parent, next = b.next;
    b = oalgpu_context::BufHost{};
    c->bufferLoopLen[size_t(h)] = 0;
    c->freeBuffers.push_back(uint32_t(h));
    if(parent >= 0) BufferUnref(c, parent);
    if(next >= 0) BufferUnref(c, next);
}
static void BufferUnref(oalgpu_context *c, int h)
{
    if(h < 0) return;
    auto &b = c->bufHost[size_t(h)];
    if(b.refs) --b.refs;
    if(b.refs == 0 && b.released && b.live) BufferFreeNow(c, h);
}
static void SetVoiceHead(oalgpu_context *c, uint32_t voice, int h)
{
    const int old = c->voiceHead[voice];
    c->voiceHead[voice] = h;
    if(h >= 0) ++c->bufHost[size_t(h)].refs;
    if(old >= 0) BufferUnref(c, old);
}

int oalgpu_buffer_release(oalgpu_context *c, int buffer)
{
    if(!c || !BufferLive(c, buffer)) return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_release: not a registered buffer");
    if(c->bufHost[size_t(buffer)].released) return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_release: released before");
    if(int rc = UseCtx(c)) return rc;
    for(const auto &cb : c->cbVoices)
        if(cb.buffer == buffer && !cb.retired) return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_release: a callback source's storage is the library's own");
    // initialisations that wait for the next update name their buffers: they are on the device before anything is freed
    if(int rc = FlushInits(c)) return rc;
    auto &b = c->bufHost[size_t(buffer)];
    b.released = true;
    if(b.refs == 0) BufferFreeNow(c, buffer);
    return OALGPU_OK;
}

int oalgpu_buffer_info(oalgpu_context *c, int buffer, int32_t *live, int32_t *release_pending, uint32_t *references)
{
    if(!c || buffer < 0 || uint32_t(buffer) >= c->bufHost.size()) return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_info: bad handle");
    const auto &b = c->bufHost[size_t(buffer)];
    if(live) *live = b.live ? 1 : 0;
    if(release_pending) *release_pending = (b.live && b.released) ? 1 : 0;
    if(references) *references = b.refs;
    return OALGPU_OK;
}

int oalgpu_buffer_register(oalgpu_context *c, const void *data, int fmt_type, uint32_t frame_step,
    uint32_t sample_len, uint32_t loop_start, uint32_t loop_end)
{
    static const size_t bytesPer[7] = {1, 2, 4, 4, 8, 1, 1};
    if(!c || !data || fmt_type < 0 || fmt_type > OALGPU_FMT_ALAW || frame_step == 0 || sample_len == 0
        || loop_end > sample_len || loop_start >= (loop_end ? loop_end : 1u))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_register: bad arguments");
    if(c->freeBuffers.empty() && c->numBuffers >= c->desc.max_buffers) return Fail(OALGPU_ERR_CAPACITY, "buffer table full");
    if(int rc = UseCtx(c)) return rc;
    const size_t nbytes = size_t{sample_len} * frame_step * bytesPer[fmt_type];
    void *dev = nullptr;
    HIP_TRY(hipMalloc(&dev, nbytes + 16));
    const hipError_t e = hipMemcpy(dev, data, nbytes, hipMemcpyHostToDevice);
    if(e != hipSuccess) { (void)hipFree(dev); return Fail(OALGPU_ERR_HIP, hipGetErrorString(e)); }
    uint32_t h = 0;
    if(int rc = AllocBufferHandle(c, &h)) { (void)hipFree(dev); return rc; }
    c->bufferData[h] = dev;
    c->bufferLoopLen[h] = loop_end > loop_start ? loop_end - loop_start : 0u;
    BufferItem item{dev, fmt_type, frame_step, sample_len, loop_start, loop_end, 0};
    HIP_TRY(hipMemcpy(c->buffers.p + h, &item, sizeof(item), hipMemcpyHostToDevice));
    return int(h);
}

int oalgpu_voice_init(oalgpu_context *c, uint32_t voice, const oalgpu_voice_desc *d)
{
    if(c) { if(int rc = FlushPendingMix(c)) return rc; }
    // (a parameter block that waits for a resident update was applied BEFORE this call: it goes in first, as on the launched path)
    if(c && c->res.pendingBlock) { if(int rc = UseCtx(c)) return rc; }
    if(!c || !d || voice >= c->L.numVoices || !BufferLive(c, d->buffer) || c->bufHost[size_t(d->buffer)].released
        || d->position_frac >= kFracOne)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_init: bad arguments");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(d->looping && c->bufferLoopLen[size_t(d->buffer)] == 0)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_init: a looping voice needs a buffer registered with loop_end > loop_start");
    RetireCallbackVoice(c, voice);
    c->initPending.push_back(VoiceInitRecord{voice, d->buffer, d->looping ? 1 : 0, d->position, d->position_frac, 0});
    if(c->voiceHead[voice] >= 0 && c->bufHost[size_t(c->voiceHead[voice])].released)
    {   // the slot's old buffer may be freed by this: the device must be through with it (and the initialisation on it first)
        if(int rc = UseCtx(c)) return rc;
        if(int rc = FlushInits(c)) return rc;
    }
    SetVoiceHead(c, voice, d->buffer);
    c->queueDoneKnown[voice] = c->queueUnqueued[voice] = 0;
    return OALGPU_OK;
}

/* ---- streaming sources: a queue of buffers (VoiceBufferItem::mNext, core/voice.h:85) --------------------
 * oalgpu_buffer_queue_link(buffer, next) is alSourceQueueBuffers' linking (next < 0 ends the queue);
 * oalgpu_voice_init_queue starts a voice that is NOT VoiceFlag::IsStatic on the queue's first buffer:
 * LoadBufferQueue (voice.cpp:563-594) crawls the queue, a looping voice returns to `first_buffer` when it
 * ends, and Voice::mix leaves finished buffers behind (voice.cpp:1182-1194). */
int oalgpu_buffer_queue_link(oalgpu_context *c, int buffer, int next_buffer)
{
    if(!c || !BufferLive(c, buffer) || (next_buffer >= 0 && (!BufferLive(c, next_buffer) || c->bufHost[size_t(next_buffer)].released)))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_queue_link: bad buffer");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    {   // the link holds its target
        const int32_t old = c->bufHost[size_t(buffer)].next;
        c->bufHost[size_t(buffer)].next = next_buffer < 0 ? -1 : next_buffer;
        if(next_buffer >= 0) ++c->bufHost[size_t(next_buffer)].refs;
        if(old >= 0) BufferUnref(c, old);
    }
    const int32_t next = next_buffer < 0 ? 0 : next_buffer + 1;
    HIP_TRY(hipMemcpy(reinterpret_cast<char*>(c->buffers.p + buffer) + offsetof(BufferItem, next), &next, sizeof(next),
        hipMemcpyHostToDevice));
    return OALGPU_OK;
}

int oalgpu_voice_init_queue(oalgpu_context *c, uint32_t voice, int first_buffer, int looping, int32_t position,
    uint32_t position_frac)
{
    if(c) { if(int rc = FlushPendingMix(c)) return rc; }
    // (a parameter block that waits for a resident update was applied BEFORE this call: it goes in first, as on the launched path)
    if(c && c->res.pendingBlock) { if(int rc = UseCtx(c)) return rc; }
    if(!c || voice >= c->L.numVoices || !BufferLive(c, first_buffer) || c->bufHost[size_t(first_buffer)].released || position_frac >= kFracOne)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_init_queue: bad arguments");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    RetireCallbackVoice(c, voice);
    c->initPending.push_back(VoiceInitRecord{voice, first_buffer, looping ? 1 : 0, position, position_frac, 1});
    if(c->voiceHead[voice] >= 0 && c->bufHost[size_t(c->voiceHead[voice])].released)
    {
        if(int rc = UseCtx(c)) return rc;
        if(int rc = FlushInits(c)) return rc;
    }
    SetVoiceHead(c, voice, first_buffer);
    c->queueDoneKnown[voice] = c->queueUnqueued[voice] = 0;
    return OALGPU_OK;
}

/* where a streaming voice is: its current buffer (-1: the queue ended) and the number of buffers it has
 * played through since it was initialised (what AsyncBufferCompleteEvent counts, voice.cpp:1207-1218) */
int oalgpu_voice_queue_state(oalgpu_context *c, uint32_t voice, int32_t *current_buffer, uint32_t *buffers_done)
{
    if(!c || voice >= c->L.numVoices || !current_buffer || !buffers_done)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_queue_state: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    HIP_TRY(hipMemcpy(current_buffer, &c->ctl.p[voice].curBuffer, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(buffers_done, c->queueDone.p + voice, sizeof(uint32_t), hipMemcpyDeviceToHost));
    c->queueDoneKnown[voice] = *buffers_done;
    return OALGPU_OK;
}

/* alSourceUnqueueBuffers: the first `count` buffers of the voice's queue -- processed ones: the library checks the count against
 * what the host has READ BACK of the voice (oalgpu_voice_queue_state, oalgpu_voices_readback) -- leave the queue: the voice's hold
 * moves on to the buffer behind them, so that a released buffer among them can be freed while the source plays on. */
int oalgpu_voice_queue_unqueue(oalgpu_context *c, uint32_t voice, uint32_t count)
{
    if(!c || voice >= c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_queue_unqueue: bad arguments");
    if(count == 0) return OALGPU_OK;
    if(c->queueUnqueued[voice] + count > c->queueDoneKnown[voice])
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_queue_unqueue: more buffers than the voice is known to have played through (read its state back first)");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    int head = c->voiceHead[voice];
    for(uint32_t i = 0; i < count && head >= 0; ++i) head = c->bufHost[size_t(head)].next;
    c->queueUnqueued[voice] += count;
    SetVoiceHead(c, voice, head);
    return OALGPU_OK;
}

/* IMA4 / MS ADPCM data (FmtIMA4 / FmtMSADPCM, core/buffer_storage.h; LoadSamples, core/voice.cpp:288-484):
 * decoded once, on the GPU, into interleaved 16-bit PCM; the handle then behaves like an OALGPU_FMT_SHORT
 * buffer with frame_step = channels (oalgpu_buffer_channel_view splits a stereo one). */
int oalgpu_buffer_register_adpcm(oalgpu_context *c, const void *data, int adpcm_type, uint32_t channels,
    uint32_t samples_per_block, uint32_t sample_len, uint32_t loop_start, uint32_t loop_end)
{
    if(!c || !data || (adpcm_type != OALGPU_ADPCM_IMA4 && adpcm_type != OALGPU_ADPCM_MS) || channels < 1 || channels > 2
        || sample_len == 0 || loop_end > sample_len || loop_start >= (loop_end ? loop_end : 1u)
        || samples_per_block < (adpcm_type == OALGPU_ADPCM_MS ? 3u : 2u) || samples_per_block > 65536u)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_register_adpcm: bad arguments");
    if(c->freeBuffers.empty() && c->numBuffers >= c->desc.max_buffers) return Fail(OALGPU_ERR_CAPACITY, "buffer table full");
    if(int rc = UseCtx(c)) return rc;
    const uint32_t numBlocks = (sample_len + samples_per_block - 1u) / samples_per_block;
    const size_t blockBytes = adpcm_type == OALGPU_ADPCM_MS ? size_t{(samples_per_block - 2u) / 2u + 7u} * channels
        : size_t{(samples_per_block - 1u) / 2u + 4u} * channels;
    const size_t nbytes = size_t{numBlocks} * blockBytes;
    void *comp = nullptr, *pcm = nullptr;
    HIP_TRY(hipMalloc(&comp, nbytes + 16));
    hipError_t e = hipMemcpy(comp, data, nbytes, hipMemcpyHostToDevice);
    if(e == hipSuccess) e = hipMalloc(&pcm, size_t{sample_len} * channels * sizeof(int16_t) + 16);
    if(e != hipSuccess) { (void)hipFree(comp); return Fail(OALGPU_ERR_HIP, hipGetErrorString(e)); }
    LaunchDecodeAdpcm(c->stream, adpcm_type == OALGPU_ADPCM_MS, static_cast<const uint8_t*>(comp), static_cast<int16_t*>(pcm),
        numBlocks, samples_per_block, channels, sample_len);
    e = hipGetLastError();
    if(e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(comp);
    if(e != hipSuccess) { (void)hipFree(pcm); return Fail(OALGPU_ERR_HIP, hipGetErrorString(e)); }
    uint32_t h = 0;
    if(int rc = AllocBufferHandle(c, &h)) { (void)hipFree(pcm); return rc; }
    c->bufferData[h] = pcm;
    c->bufferLoopLen[h] = loop_end > loop_start ? loop_end - loop_start : 0u;
    BufferItem item{pcm, OALGPU_FMT_SHORT, channels, sample_len, loop_start, loop_end, 0};
    HIP_TRY(hipMemcpy(c->buffers.p + h, &item, sizeof(item), hipMemcpyHostToDevice));
    return int(h);
}

/* Voice::mStartTime (core/voice.h:209): the voice starts `samples` output samples from the beginning of
 * the next update (Voice::mix's delayed start, voice.cpp:1023-1046: outPos = round((mStartTime -
 * deviceTime) * rate)); updates that end before that leave the voice untouched. */
int oalgpu_voice_set_start_delay(oalgpu_context *c, uint32_t voice, uint32_t samples)
{
    if(!c || voice >= c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_start_delay: bad voice");
    if(c->cbOfVoice[voice] >= 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_start_delay: not for callback voices");
    if(samples >= c->desc.sample_rate)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_start_delay: a start a second or more ahead is not scheduled yet (voice.cpp:1036-1038)");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    LaunchSetStartDelay(c->stream, c->L, voice, samples);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

int oalgpu_voice_set_ambi_scale(oalgpu_context *c, uint32_t voice, float xover_norm, float hf_scale, float lf_scale)
{
    if(!c || voice >= c->L.numVoices || !(xover_norm > 0.0f) || !(xover_norm < 0.5f))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_ambi_scale: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    AmbiScaleState st{};
    st.coeff = SplitterCoeff(xover_norm);
    st.hfScale = hf_scale; st.lfScale = lf_scale;
    LaunchSetAmbiScale(c->stream, c->L, voice, st);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

int oalgpu_context_set_nfc(oalgpu_context *c, float w1, const uint32_t channels_per_order[5])
{
    if(!c || !channels_per_order || !(w1 > 0.0f)) return Fail(OALGPU_ERR_INVALID, "oalgpu_context_set_nfc: bad arguments");
    if(c->L.hrtf)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_context_set_nfc: near-field control is for dry-line contexts (HRTF voices mix through DoHrtfMix)");
    if(c->L.nfc) return Fail(OALGPU_ERR_INVALID, "oalgpu_context_set_nfc: already set");
    uint32_t lines = channels_per_order[0], orders = 0;
    if(channels_per_order[0] != 1) return Fail(OALGPU_ERR_INVALID, "oalgpu_context_set_nfc: channels_per_order[0] must be 1 (W)");
    for(uint32_t o = 1; o < 5 && channels_per_order[o]; ++o) { lines += channels_per_order[o]; ++orders; }
    if(orders == 0 || lines > c->L.numDry) return Fail(OALGPU_ERR_INVALID, "oalgpu_context_set_nfc: orders do not fit the dry bus");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    DeviceLayout &L = c->L;
    const size_t nv = L.numVoices;
    HIP_TRY(c->nfc.alloc(nv)); HIP_TRY(c->nfc.zero());
    if(c->useWave)
    {   // the wavefront kernel: every order adds one stream row per voice (near-field contexts mix through stream rows)
        L.accLines = 0;
        L.sliceLines = 0;
        if(L.rows8)
        {   // (back to the wavefront-per-voice grid: the rows kernel has no near-field rows)
            L.rows8 = 0; L.rowsVpg = 0;
            L.numGroups = std::max<uint32_t>(1u, WaveKernelGroups(L)); L.numLineGroups = L.numGroups;
        }
        const uint32_t spv = 2u + L.numSends + orders;
        HIP_TRY(c->streams.alloc(nv * spv * kLine)); HIP_TRY(c->streams.zero());
        HIP_TRY(c->lineGains.alloc(nv * spv * LineBlockDwords(L.lineStride))); HIP_TRY(c->lineGains.zero());
        L.streams = c->streams.p; L.lineGains = c->lineGains.p; L.streamsPerVoice = spv;
    }
    L.nfc = c->nfc.p;
    L.nfcOrders = orders;
    for(int o = 0; o < 5; ++o) L.chansPerOrder[o] = (uint32_t(o) <= orders) ? channels_per_order[o] : 0u;
    NfcInit(w1, c->nfcDevice);
    return OALGPU_OK;
}

int oalgpu_voice_set_nfc(oalgpu_context *c, uint32_t voice, float w0)
{
    if(!c || voice >= c->L.numVoices || !(w0 >= 0.0f)) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_nfc: bad arguments");
    if(!c->L.nfc) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_nfc: oalgpu_context_set_nfc first");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    NfcDesign d = c->nfcDevice;                  // chandata.NFCtrlFilter = device->mNFCtrlFilter, then adjust(w0)
    NfcAdjust(w0, d);
    NfcState st{};
    std::memcpy(st.a, d.a, sizeof(st.a));
    std::memcpy(st.b, d.b, sizeof(st.b));
    LaunchSetNfc(c->stream, c->L, voice, st);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

int oalgpu_buffer_channel_view(oalgpu_context *c, int buffer, uint32_t channel)
{
    if(!c || !BufferLive(c, buffer) || c->bufHost[size_t(buffer)].released) return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_channel_view: bad buffer");
    if(c->freeBuffers.empty() && c->numBuffers >= c->desc.max_buffers) return Fail(OALGPU_ERR_CAPACITY, "buffer table full");
    if(int rc = UseCtx(c)) return rc;
    static const size_t bytesPer[7] = {1, 2, 4, 4, 8, 1, 1};
    BufferItem item{};
    HIP_TRY(hipMemcpy(&item, c->buffers.p + buffer, sizeof(item), hipMemcpyDeviceToHost));
    if(channel >= item.frameStep) return Fail(OALGPU_ERR_INVALID, "oalgpu_buffer_channel_view: channel >= frame_step");
    item.data = static_cast<const char*>(item.data) + size_t{channel} * bytesPer[item.fmt];
    uint32_t h = 0;
    if(int rc = AllocBufferHandle(c, &h)) return rc;
    item.next = 0;
    HIP_TRY(hipMemcpy(c->buffers.p + h, &item, sizeof(item), hipMemcpyHostToDevice));
    c->bufferData[h] = nullptr;                            // the storage belongs to `buffer`: the view holds it
    c->bufferLoopLen[h] = c->bufferLoopLen[size_t(buffer)];
    c->bufHost[h].parent = buffer;
    ++c->bufHost[size_t(buffer)].refs;
    return int(h);
}

static int BuildParamRecords(oalgpu_context *c, const uint32_t *voices, const oalgpu_voice_params *params,
    size_t count, std::vector<ParamRecord> &recs)
{
    const TableBlob &blob = Blob();
    recs.resize(count);
    HrtfStoreDev hostStore{};
    if(c->L.hrtf && c->hrtfLoaded) hostStore = HostStoreView(c->hrtfHost);
    for(size_t i = 0; i < count; ++i)
    {
        const oalgpu_voice_params &p = params[i];
        if(voices[i] >= c->L.numVoices || p.resampler < 0 || p.resampler > OALGPU_RESAMPLER_BSINC48)
            return Fail(OALGPU_ERR_INVALID, "voice parameters: bad voice index or resampler");
        ParamRecord &r = recs[i];
        std::memset(&r, 0, sizeof(r));
        r.voice = voices[i];
        r.step = p.step;
        oalgpu_interp_state st;
        PrepareResampler(p.resampler, p.step ? p.step : 1u, &st);
        r.rsKind = st.kind; r.rsM = st.m; r.rsL = st.l; r.rsSf = st.sf;
        r.rsFilterOffset = blob.filterBase(st);
        r.flags = p.direct_filter.active ? kFlagDirectFilter : 0u;
        DesignBiquadFromSlope(OALGPU_BIQUAD_HIGHSHELF, p.direct_filter.hf_norm, p.direct_filter.gain_hf, 1.0f, r.dirLp);
        DesignBiquadFromSlope(OALGPU_BIQUAD_LOWSHELF, p.direct_filter.lf_norm, p.direct_filter.gain_lf, 1.0f, r.dirHp);
        for(uint32_t s = 0; s < OALGPU_MAX_SENDS; ++s)
        {
            r.sendSlot[s] = -1;
            if(s >= c->L.numSends) continue;
            if(p.send_slot[s] >= int32_t(c->L.numSlots)) return Fail(OALGPU_ERR_INVALID, "send slot out of range");
            r.sendSlot[s] = p.send_slot[s] < 0 ? -1 : p.send_slot[s];
            if(p.send_filter[s].active) r.flags |= 1u << (kFlagSendFilterShift + s);
            DesignBiquadFromSlope(OALGPU_BIQUAD_HIGHSHELF, p.send_filter[s].hf_norm, p.send_filter[s].gain_hf, 1.0f, r.sendLp[s]);
            DesignBiquadFromSlope(OALGPU_BIQUAD_LOWSHELF, p.send_filter[s].lf_norm, p.send_filter[s].gain_lf, 1.0f, r.sendHp[s]);
            std::memcpy(r.sendGains[s], p.send_gains[s], sizeof(r.sendGains[s]));
        }
        r.hrtfDir[0] = p.hrtf_ev; r.hrtfDir[1] = p.hrtf_az; r.hrtfDir[2] = p.hrtf_dist; r.hrtfDir[3] = p.hrtf_spread;
        r.hrtfGain = p.hrtf_gain;
        // negative distances are reserved (the reference never passes one: a vector norm, alu.cpp:1761): -1 keeps the target
        if(c->L.hrtf && p.hrtf_dist < 0.0f && p.hrtf_dist != OALGPU_HRTF_KEEP_TARGET)
            return Fail(OALGPU_ERR_INVALID, "voice parameters: hrtf_dist < 0 is reserved (OALGPU_HRTF_KEEP_TARGET = -1)");
        r.keepHrtf = (c->L.hrtf && p.hrtf_dist == OALGPU_HRTF_KEEP_TARGET) ? 1u : 0u;
        if(c->L.hrtf && c->hrtfLoaded && !r.keepHrtf)
        {   // the index half of HrtfStore::getCoeffs (core/hrtf.cpp:192-245) on the host's copy of the store
            const HrirBlend b = HrtfBlendFor(hostStore, p.hrtf_ev, p.hrtf_az, p.hrtf_dist, p.hrtf_spread);
            for(int k = 0; k < 4; ++k) { r.hrtfIdx[k] = b.idx[k]; r.hrtfW[k] = b.w[k]; }
            r.hrtfPass = b.passthru;
            r.hrtfDelay[0] = b.delay[0]; r.hrtfDelay[1] = b.delay[1];
        }
        std::memcpy(r.dryGains, p.dry_gains, sizeof(r.dryGains));
    }
    return OALGPU_OK;
}

int oalgpu_voice_set_params(oalgpu_context *c, const uint32_t *voices, const oalgpu_voice_params *params, size_t count)
{
    if(!c || !voices || !params) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(count == 0) return OALGPU_OK;
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(int rc = BuildParamRecords(c, voices, params, count, c->paramHost)) return rc;
    NoteCallbackSteps(c, voices, params, count);
    if(c->paramDev.n < count) HIP_TRY(c->paramDev.alloc(count));
    HIP_TRY(hipMemcpyAsync(c->paramDev.p, c->paramHost.data(), count * sizeof(ParamRecord), hipMemcpyHostToDevice, c->stream));
    LaunchApplyParams(c->stream, c->L, c->paramDev.p, uint32_t(count));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));   // paramHost is reused by the next call
    return OALGPU_OK;
}

/* Hrtf.Target of `count` voices as the reference's parameter stage left it in the Voice (HrtfFilter: Coeffs, Delay, Gain,
 * core/mixer/hrtfdefs.h:36-40, written by CalcHrtfPanning, alc/alu.cpp:1214-1216 / :1256-1258 / :1296-1298). */
int oalgpu_voice_set_hrtf_targets(oalgpu_context *c, const uint32_t *voices, const float *coeffs, const uint32_t *delays,
    const float *gains, size_t count)
{
    if(!c || !voices || !coeffs || !delays || !gains) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(count == 0) return OALGPU_OK;
    if(!c->L.hrtf) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_hrtf_targets: HRTF contexts only");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    std::vector<TargetRecord> recs(count);
    for(size_t i = 0; i < count; ++i)
    {
        if(voices[i] >= c->L.numVoices || delays[2 * i] > 63u || delays[2 * i + 1] > 63u)
            return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_hrtf_targets: bad voice index or delay (MaxHrirDelay = 63)");
        recs[i] = TargetRecord{voices[i], {delays[2 * i], delays[2 * i + 1]}, gains[i]};
    }
    HIP_TRY(hipStreamSynchronize(c->stream));           // (the staging buffers of the previous call may still be read)
    if(c->tgtRecs.n < count) { HIP_TRY(c->tgtRecs.alloc(count)); HIP_TRY(c->tgtCoeffs.alloc(count * kHrirLen * 2)); }
    HIP_TRY(hipMemcpyAsync(c->tgtRecs.p, recs.data(), count * sizeof(TargetRecord), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->tgtCoeffs.p, coeffs, count * kHrirLen * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    LaunchApplyTargets(c->stream, c->L, c->tgtRecs.p, c->tgtCoeffs.p, uint32_t(count));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));           // the caller's arrays and `recs` are free again
    return OALGPU_OK;
}


int oalgpu_param_block_create(oalgpu_context *c, const uint32_t *voices, const oalgpu_voice_params *params,
    size_t count, oalgpu_param_block **out)
{
    if(!c || !voices || !params || !out || count == 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_param_block_create: bad arguments");
    *out = nullptr;
    // the records carry the index half of getCoeffs, evaluated now against the loaded store
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "oalgpu_param_block_create: HRTF context without a data set");
    if(int rc = UseCtx(c)) return rc;
    std::vector<ParamRecord> recs;
    if(int rc = BuildParamRecords(c, voices, params, count, recs)) return rc;
    auto b = std::make_unique<oalgpu_param_block>();
    b->count = uint32_t(count);
    b->device = c->desc.device;
    b->hrtfGeneration = c->hrtfGeneration;
    HIP_TRY(b->recs.alloc(count));
    HIP_TRY(b->recs.upload(recs.data(), count));
    {
        std::vector<int32_t> map(c->L.numVoices, -1);
        bool unique = true;
        for(size_t i = 0; i < count; ++i) { unique = unique && map[voices[i]] < 0; map[voices[i]] = int32_t(i); }
        if(unique)      // (a block that names a voice twice is applied by ApplyParamsKernel, record by record)
        {
            HIP_TRY(b->voiceToRec.alloc(map.size()));
            HIP_TRY(b->voiceToRec.upload(map.data(), map.size()));
            b->mapVoices = uint32_t(map.size());
            // (rows of up to 64 taps: InstallPair moves one tap pair per lane; longer responses are blended at install, ApplyRecordLean)
            if((c->res.enabled || (c->desc.flags & OALGPU_CTX_APPLY_IN_VOICE_KERNEL)) && c->L.hrtf && c->L.hrirs && c->L.irStride <= 64u)
            {   // a resident context's voice kernel installs the block itself: the HRIR blend of every record now, once
                HIP_TRY(b->rows.alloc(count * size_t{c->L.irStride} * 2));
                HIP_TRY(b->rows.zero());
                LaunchBlendRows(c->stream, c->L, b->recs.p, uint32_t(count), b->rows.p);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(c->stream));
            }
        }
    }
    for(size_t i = 0; i < count; ++i)
        if(c->cbOfVoice[voices[i]] >= 0) b->cbSteps.emplace_back(voices[i], params[i].step);
    *out = b.release();
    return OALGPU_OK;
}

int oalgpu_param_block_apply(oalgpu_context *c, oalgpu_param_block *b)
{
    if(!c || !b) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(c->L.hrtf && b->hrtfGeneration != c->hrtfGeneration)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_param_block_apply: the block was built against another HRTF data set (its HRIR indices are that store's); create it again");
    if(!c->res.cooldown && ResidentWanted(c, 1) && !c->res.pendingBlock && b->mapVoices == c->L.numVoices && b->cbSteps.empty() && b->rows.p)
    {   // a resident context: the block rides in the next update's doorbell slot and the voice kernel's wavefronts install it
        // (another entry point in between applies it the launched way: FlushResidentBlock)
        c->res.pendingBlock = b;
        b->heldBy = c;
        return OALGPU_OK;
    }
    if(c->pendingMix.active && b->mapVoices == c->L.numVoices && WaveKernelAppliesRecords(c->L) && c->initPending.empty())
    {   // the update submitted last has not been launched yet: its voice kernel installs this block (see pendingMix)
        if(int rc = UseDevice(c->desc.device)) return rc;
        if(int rc = FlushPendingMix(c, b)) return rc;
    }
    else
    {
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    LaunchApplyParams(c->stream, c->L, b->recs.p, b->count);
    HIP_TRY(hipGetLastError());
    }
    for(const auto &vs : b->cbSteps)
        if(vs.first < c->cbOfVoice.size() && c->cbOfVoice[vs.first] >= 0) c->cbVoices[size_t(c->cbOfVoice[vs.first])].step = vs.second;
    return OALGPU_OK;
}

void oalgpu_param_block_destroy(oalgpu_param_block *b)
{
    if(!b) return;
    (void)UseDevice(b->device);                 // (resident voice kernels leave: freeing device memory waits for the device)
    // a context that still holds the block for its next resident update applies it now, while it exists
    if(b->heldBy && b->heldBy->res.pendingBlock == b) (void)FlushResidentBlock(b->heldBy);
    delete b;
}

// Large-BAR boxes: fine-grained device memory is host-addressable, so a slot the host fills per update can BE device memory
// (the kernel's first read is an HBM read, not a PCIe round trip).  Checked once per context, and without trusting the
// property alone: (1) the pointer is probed through the kernel's own user-copy path (write(2) out of it, read(2) into it, on a
// pipe: an address the CPU cannot touch comes back as EFAULT, not as SIGSEGV); (2) the host stores a pattern, a KERNEL reads it
// (plain loads, as ApplyMovesKernel reads its records), the host stores a second pattern over it, a second launch reads that:
// a slot whose lines an earlier launch left in L2 must show the new stores, or the context keeps pinned host memory.
// (BAR stores do not pass through an L2; the kernels read such slots either at their start, behind the launch's own
// invalidate, or -- the resident kernel's doorbell -- with system-scope loads.)
namespace {
__global__ void ProbeReadKernel(const uint32_t *src, uint32_t *dst, uint32_t n)
{
    const uint32_t i = threadIdx.x;
    if(i < n) dst[i] = src[i] + __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
bool HostCanTouch(void *p, size_t bytes)
{
    int fd[2];
    if(pipe(fd) != 0) return false;
    bool ok = write(fd[1], p, bytes) == ssize_t(bytes);        // the kernel reads the range on our behalf
    if(ok) ok = read(fd[0], p, bytes) == ssize_t(bytes);       // ... and writes the same bytes back into it
    close(fd[0]); close(fd[1]);
    return ok;
}
}
bool HostStoresReachDevice(oalgpu_context *c)
{
    hipDeviceProp_t prop{};
    if(hipGetDeviceProperties(&prop, c->desc.device) != hipSuccess || !prop.isLargeBar) return false;
    uint32_t *probe = nullptr, *seen = nullptr;
    if(hipExtMallocWithFlags(reinterpret_cast<void**>(&probe), 256, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return false; }
    if(hipMalloc(reinterpret_cast<void**>(&seen), 256) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(probe); return false; }
    bool ok = HostCanTouch(probe, 256);
    for(uint32_t round = 0; ok && round < 2; ++round)
    {
        const uint32_t pat = round ? 0xc0de0000u : 0x5eed0000u;
        for(uint32_t i = 0; i < 64; ++i) probe[i] = pat + i;
        __builtin_ia32_sfence();
        uint32_t back[64] = {};
        hipLaunchKernelGGL(ProbeReadKernel, dim3(1), dim3(64), 0, c->stream, probe, seen, 64u);
        if(hipGetLastError() != hipSuccess || hipMemcpyAsync(back, seen, sizeof(back), hipMemcpyDeviceToHost, c->stream) != hipSuccess
            || hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); ok = false; }
        for(uint32_t i = 0; ok && i < 64; ++i) ok = back[i] == 2u * (pat + i);
    }
    (void)hipFree(probe); (void)hipFree(seen);
    return ok;
}

int oalgpu_voice_move_async(oalgpu_context *c, const oalgpu_voice_move *pans, size_t count)
{
    if(!c || !pans) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(count == 0) return OALGPU_OK;
    if(!c->L.hrtf) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_move_async: HRTF contexts only");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(count > c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_move_async: more records than voices");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(c->panCap < count)
    {   // (grows only while nothing is in flight: the first call, or a larger batch than ever before)
        if(int rc = oalgpu_sync(c)) return rc;
        for(uint32_t k = 0; k < oalgpu_context::kIoSlots; ++k)
        {
            if(c->panHost[k]) { HIP_TRY(c->panInBar ? hipFree(c->panHost[k]) : hipHostFree(c->panHost[k])); c->panHost[k] = nullptr; }
        }
        c->panInBar = HostStoresReachDevice(c);
        for(uint32_t k = 0; k < oalgpu_context::kIoSlots; ++k)
        {
            if(c->panInBar) HIP_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&c->panHost[k]), c->L.numVoices * sizeof(oalgpu_voice_move), hipDeviceMallocFinegrained));
            else HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->panHost[k]), c->L.numVoices * sizeof(oalgpu_voice_move), hipHostMallocDefault));
            // (the host only asks whether the kernel is through with the slot: no data comes back behind this event, so no system-scope fence)
            if(!c->panApplied[k]) HIP_TRY(hipEventCreateWithFlags(&c->panApplied[k], hipEventDisableTiming | hipEventDisableSystemFence));
        }
        c->panCap = c->L.numVoices;
        c->panNext = 0;
    }
    const uint32_t slot = c->panNext % oalgpu_context::kIoSlots;
    if(c->panNext >= oalgpu_context::kIoSlots && c->panUpdate[slot] > c->updatesKnownDone)
        HIP_TRY(hipEventSynchronize(c->panApplied[slot]));   // its last use, four batches ago
    c->panUpdate[slot] = c->updatesSubmitted + 1u;           // installed in front of the next update's voices
    for(size_t i = 0; i < count; ++i)
        if(pans[i].voice >= c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_move_async: bad voice index");
    // The records go into the slot as they are and the kernel reads them straight out of it (24 bytes per moved voice: device
    // memory the host stores into through the BAR, or pinned host memory read over PCIe, behind the update that is mixing): one copy and ONE runtime call on the calling thread, which is what bounds this
    // boundary -- getCoeffs' index half, which the thread used to evaluate per record, runs in the kernel.  The slot is free again
    // when the event bound to the dispatch has fired.
    std::memcpy(c->panHost[slot], pans, count * sizeof(oalgpu_voice_move));
    if(c->panInBar) __builtin_ia32_sfence();        // (write-combined stores through the BAR: out of the core before the doorbell)
    LaunchApplyMoves(c->stream, c->L, c->hrtfDev, c->panHost[slot], uint32_t(count), c->panApplied[slot]);
    HIP_TRY(hipGetLastError());
    ++c->panNext;
    return OALGPU_OK;
}


/* ---- panning on the GPU: CalcDirectionCoeffs + ComputePanGains (core/mixer.h:68-73, core/mixer.cpp:16-102) ---- */
static int UploadAmbiMap(DevBuf<AmbiMapEntry> &dst, size_t at, const uint8_t *index, const float *scale, uint32_t n)
{
    std::vector<AmbiMapEntry> m(n);
    for(uint32_t i = 0; i < n; ++i)
    {
        if(index[i] >= OALGPU_MAX_AMBI_CHANNELS) return Fail(OALGPU_ERR_INVALID, "ambisonic channel index out of range");
        m[i] = AmbiMapEntry{index[i], scale[i]};
    }
    HIP_TRY(hipMemcpy(dst.p + at, m.data(), n * sizeof(AmbiMapEntry), hipMemcpyHostToDevice));
    return OALGPU_OK;
}

int oalgpu_context_set_ambi_map(oalgpu_context *c, const uint8_t *index, const float *scale)
{
    if(!c || !index || !scale) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    return UploadAmbiMap(c->dryMap, 0, index, scale, c->L.numDry);
}

int oalgpu_slot_set_ambi_map(oalgpu_context *c, uint32_t slot, const uint8_t *index, const float *scale)
{
    if(!c || !index || !scale || slot >= c->L.numSlots) return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_ambi_map: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    return UploadAmbiMap(c->wetMaps, size_t{slot} * c->L.wetChannels, index, scale, c->L.wetChannels);
}

int oalgpu_voice_set_pan(oalgpu_context *c, const uint32_t *voices, const oalgpu_voice_pan *pans, size_t count)
{
    if(!c || !voices || !pans || count == 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_pan: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    std::vector<PanRecord> recs(count);
    for(size_t i = 0; i < count; ++i)
    {
        if(voices[i] >= c->L.numVoices) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_pan: bad voice index");
        PanRecord &r = recs[i];
        r.voice = voices[i];
        std::memcpy(r.dir, pans[i].dir, sizeof(r.dir));
        r.spread = pans[i].spread; r.dryGain = pans[i].dry_gain;
        std::memcpy(r.sendGain, pans[i].send_gain, sizeof(r.sendGain));
    }
    // (the records of the previous call may still be read by its kernel)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if(c->panRecs.n < count) HIP_TRY(c->panRecs.alloc(count));
    HIP_TRY(hipMemcpyAsync(c->panRecs.p, recs.data(), count * sizeof(PanRecord), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    LaunchPanGains(c->stream, c->L, c->panRecs.p, uint32_t(count), c->dryMap.p, c->wetMaps.p);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

int oalgpu_voice_set_state(oalgpu_context *c, uint32_t voice, int play_state)
{
    if(!c || voice >= c->L.numVoices || play_state < OALGPU_VOICE_STOPPED || play_state > OALGPU_VOICE_PENDING)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_set_state: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    const int32_t st = play_state;
    HIP_TRY(hipMemcpyAsync(&c->ctl.p[voice].playState, &st, sizeof(st), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if(c->cbOfVoice[voice] >= 0) c->cbVoices[size_t(c->cbOfVoice[voice])].state = play_state;
    if(play_state == OALGPU_VOICE_STOPPED) SetVoiceHead(c, voice, -1);      // (a released buffer the slot was the last to hold is freed: hipFree waits for the device)
    return OALGPU_OK;
}


