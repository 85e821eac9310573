// Callback sources (AL_SOFT_callback_buffer): Voice::mix's IsCallback leg mirrored on the host.
#include "api_context.hpp"

// ---- callback sources: what Voice::mix does for VoiceFlag::IsCallback, mirrored on the host ---------------------
namespace {

// CalculateBufferSize, core/voice.cpp:600-640
void CalcBufferSizeHost(uint32_t fracPos, uint32_t increment, uint32_t dstRemaining, uint32_t &dst, uint32_t &src)
{
    constexpr uint32_t srcMax = kResampleDataSize - kMaxEdge;
    const uint32_t ext = increment <= kFracOne ? 1u : 0u;
    const uint64_t srcSize = ((uint64_t{dstRemaining - ext} * increment + fracPos) >> kFracBits) + ext + kMaxEdge;
    if(srcSize <= srcMax) { dst = dstRemaining; src = uint32_t(srcSize); return; }
    const uint64_t dstSize = ((uint64_t{srcMax - kMaxEdge} << kFracBits) - fracPos) / increment;
    if(dstSize < dstRemaining) { dst = uint32_t(dstSize) & ~3u; src = srcMax; return; }
    dst = dstRemaining; src = srcMax;
}
} // namespace

// Before the update's voice kernel: the requests LoadResampledSamples would make (voice.cpp:726-752), the storage to
// the device, the voice's window; then what Voice::mix does to the voice's position and blocks afterwards (:1121-1180)
int ServiceCallbacks(oalgpu_context *c, uint32_t samplesToDo)
{
    for(auto &cb : c->cbVoices)
    {
        if(cb.state != OALGPU_VOICE_PLAYING && cb.state != OALGPU_VOICE_STOPPING) continue;
        if(!cb.hasBuffer)
        {   // an ended voice renders once more, fading out (voice.cpp:1224-1232), with no buffer to load from
            if(cb.state == OALGPU_VOICE_STOPPING) cb.state = OALGPU_VOICE_STOPPED;
            continue;
        }
        if(cb.step == 0) return Fail(OALGPU_ERR_INVALID, "a callback voice is mixed before its first oalgpu_voice_set_params (mStep = 0)");
        uint32_t frac = cb.frac, off = cb.blockOffset;
        for(uint32_t loaded = 0; loaded < samplesToDo;)
        {
            uint32_t dst, src;
            CalcBufferSizeHost(frac, cb.step, samplesToDo - loaded, dst, src);
            const uint32_t needBlocks = off + src;                      // one sample per block; the position is not negative
            if(needBlocks > cb.capacityFrames)
                return Fail(OALGPU_ERR_CAPACITY, "callback voice: the update needs more frames than the callback storage holds");
            if(!cb.stopped && needBlocks > cb.numBlocks)
            {
                const size_t byteOffset = size_t{cb.numBlocks} * cb.frameBytes;
                const uint32_t needBytes = (needBlocks - cb.numBlocks) * cb.frameBytes;
                const int32_t ret = cb.fn(cb.user, cb.data.data() + byteOffset, int32_t(needBytes));
                const uint32_t got = ret < 0 ? 0u : uint32_t(ret);      // al::saturate_cast<unsigned>
                cb.stopped = got != needBytes;
                if(got <= needBytes) cb.numBlocks += got / cb.frameBytes;
            }
            loaded += dst;
            if(loaded < samplesToDo)
            {
                frac += dst * cb.step;
                off += frac >> kFracBits;
                frac &= kFracOne - 1u;
            }
        }
        // the storage, in stream order behind the previous update's voice kernel
        const uint32_t slot = cb.slot; cb.slot ^= 1u;
        HIP_TRY(hipEventSynchronize(cb.copied[slot]));                  // the staging slot's last copy has left it
        const size_t bytes = size_t{cb.numBlocks} * cb.frameBytes;
        if(bytes)
        {
            std::memcpy(cb.pinned[slot], cb.data.data(), bytes);
            HIP_TRY(hipMemcpyAsync(c->bufferData[size_t(cb.buffer)], cb.pinned[slot], bytes, hipMemcpyHostToDevice, c->stream));
        }
        HIP_TRY(hipEventRecord(cb.copied[slot], c->stream));
        LaunchSetVoiceWindow(c->stream, c->L, cb.voice, cb.buffer, cb.numBlocks, int32_t(cb.blockOffset));
        HIP_TRY(hipGetLastError());

        if(cb.state == OALGPU_VOICE_STOPPING) { cb.state = OALGPU_VOICE_STOPPED; continue; }    // no position update when stopping
        const uint64_t total = uint64_t{cb.frac} + uint64_t{cb.step} * samplesToDo;
        const uint32_t samplesDone = uint32_t(total >> kFracBits);
        cb.frac = uint32_t(total) & (kFracOne - 1u);
        const int64_t pos = int64_t{cb.position} + samplesDone;
        cb.position = pos > 2147483647ll ? 2147483647 : int32_t(pos);
        if(cb.position > 0)
        {
            const uint32_t endOffset = cb.blockOffset + std::min(samplesDone, uint32_t(cb.position));
            const uint32_t blocksDone = endOffset;                      // / mSamplesPerBlock
            if(blocksDone == 0) cb.blockOffset = endOffset;
            else if(blocksDone < cb.numBlocks)
            {
                std::memmove(cb.data.data(), cb.data.data() + size_t{blocksDone} * cb.frameBytes,
                    size_t{cb.numBlocks - blocksDone} * cb.frameBytes);
                cb.numBlocks -= blocksDone;
                cb.blockOffset = endOffset - blocksDone;
            }
            else
            {   // the voice just ended: Stopping, so that the next render fades any residual to 0
                cb.hasBuffer = false; cb.numBlocks = 0; cb.blockOffset = 0;
                cb.state = OALGPU_VOICE_STOPPING;
            }
        }
    }
    return OALGPU_OK;
}



int oalgpu_voice_init_callback(oalgpu_context *c, uint32_t voice, int fmt_type, uint32_t position_frac,
    oalgpu_callback_fn fn, void *userptr)
{
    static const uint32_t bytesPer[7] = {1, 2, 4, 4, 8, 1, 1};
    if(!c || !fn || voice >= c->L.numVoices || fmt_type < 0 || fmt_type > OALGPU_FMT_ALAW || position_frac >= kFracOne)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_init_callback: bad arguments");
    if(c->comm) return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_init_callback: not on a sharded context");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(int rc = UseCtx(c)) return rc;
    if(c->cbOfVoice[voice] >= 0)
    {   // the voice is a callback source already: only one that has ended may start over
        if(c->cbVoices[size_t(c->cbOfVoice[voice])].state != OALGPU_VOICE_STOPPED)
            return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_init_callback: the voice already is a playing callback source");
        RetireCallbackVoice(c, voice);
    }
    const uint32_t capacityFrames = uint32_t(kLine + 256) * 10u + uint32_t(kMaxEdge);      // MixerLineSize*MaxPitch + MaxResamplerEdge, al/buffer.cpp:474
    const size_t nbytes = size_t{capacityFrames} * bytesPer[fmt_type];
    // a retired entry's buffer-table slot, device buffer, pinned staging and events serve the new source
    int32_t reuse = -1;
    for(size_t j = 0; j < c->cbVoices.size(); ++j)
        if(c->cbVoices[j].retired) { reuse = int32_t(j); break; }
    if(reuse < 0 && c->freeBuffers.empty() && c->numBuffers >= c->desc.max_buffers) return Fail(OALGPU_ERR_CAPACITY, "buffer table full");
    oalgpu_context::CbVoice fresh;
    oalgpu_context::CbVoice &cb = reuse >= 0 ? c->cbVoices[size_t(reuse)] : fresh;
    if(reuse >= 0)
    {
        if(int rc = oalgpu_sync(c)) return rc;                     // nothing in flight reads the old source's data any more
        if(cb.allocBytes < nbytes)
        {   // a wider sample type than the entry was made for
            (void)hipFree(c->bufferData[size_t(cb.buffer)]); c->bufferData[size_t(cb.buffer)] = nullptr;
            for(int k = 0; k < 2; ++k) { (void)hipHostFree(cb.pinned[k]); cb.pinned[k] = nullptr; }
            cb.allocBytes = 0;
        }
    }
    struct Undo {       // what a failure below must not leave behind
        oalgpu_context *c; oalgpu_context::CbVoice *cb; void *dev{nullptr}; bool armed{true};
        ~Undo()
        {
            if(!armed) return;
            if(dev) (void)hipFree(dev);
            for(int k = 0; k < 2; ++k)
            {
                if(cb->pinned[k]) { (void)hipHostFree(cb->pinned[k]); cb->pinned[k] = nullptr; }
                if(cb->copied[k]) { (void)hipEventDestroy(cb->copied[k]); cb->copied[k] = nullptr; }
            }
            cb->allocBytes = 0;
        }
    } undo{c, &cb};
    void *dev = cb.allocBytes ? c->bufferData[size_t(cb.buffer)] : nullptr;
    if(!cb.allocBytes)
    {
        HIP_TRY(hipMalloc(&dev, nbytes + 16));
        undo.dev = dev;
        for(int k = 0; k < 2; ++k)
        {
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&cb.pinned[k]), nbytes, hipHostMallocDefault));
            if(!cb.copied[k]) HIP_TRY(hipEventCreateWithFlags(&cb.copied[k], hipEventDisableTiming));
        }
    }
    HIP_TRY(hipMemset(dev, 0, nbytes + 16));
    for(int k = 0; k < 2; ++k) HIP_TRY(hipEventRecord(cb.copied[k], c->stream));
    uint32_t h = reuse >= 0 ? uint32_t(cb.buffer) : 0u;
    if(reuse < 0) { if(int rc = AllocBufferHandle(c, &h)) return rc; }       // (the storage is the library's own: never released by the host)
    // one frame long until the first update hands the voice its window (a static buffer has at least one)
    BufferItem item{dev, fmt_type, 1u, 1u, 0u, 0u, 0};
    HIP_TRY(hipMemcpy(c->buffers.p + h, &item, sizeof(item), hipMemcpyHostToDevice));
    undo.armed = false;
    c->bufferData[h] = dev;
    c->bufferLoopLen[h] = 0u;
    cb.voice = voice; cb.fn = fn; cb.user = userptr; cb.buffer = int32_t(h);
    cb.frameBytes = bytesPer[fmt_type]; cb.capacityFrames = capacityFrames;
    if(!cb.allocBytes) cb.allocBytes = nbytes;
    cb.data.assign(nbytes, 0);
    cb.numBlocks = 0; cb.blockOffset = 0; cb.stopped = false; cb.position = 0; cb.frac = position_frac; cb.step = 0;
    cb.state = OALGPU_VOICE_PLAYING; cb.hasBuffer = true; cb.slot = 0; cb.retired = false;
    c->initPending.push_back(VoiceInitRecord{voice, int32_t(h), 0, 0, position_frac, 0});
    if(reuse >= 0) c->cbOfVoice[voice] = reuse;
    else
    {
        c->cbOfVoice[voice] = int32_t(c->cbVoices.size());
        c->cbVoices.push_back(std::move(fresh));
    }
    return OALGPU_OK;
}

int oalgpu_voice_callback_state(oalgpu_context *c, uint32_t voice, oalgpu_callback_state *out)
{
    if(!c || !out || voice >= c->L.numVoices || c->cbOfVoice[voice] < 0)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_voice_callback_state: not a callback voice");
    const auto &cb = c->cbVoices[size_t(c->cbOfVoice[voice])];
    out->position = cb.position; out->position_frac = cb.frac;
    out->num_blocks = cb.numBlocks; out->block_offset = cb.blockOffset;
    out->stopped = cb.stopped ? 1 : 0; out->play_state = cb.state; out->has_buffer = cb.hasBuffer ? 1 : 0;
    return OALGPU_OK;
}


