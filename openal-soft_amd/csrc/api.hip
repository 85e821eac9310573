// C-ABI shim (include/oalgpu.h) over the HIP kernels.  Host logic only: argument checks,
// HBM allocation/upload, parameter preparation that the reference does with libm on its mixer
// thread (resampler state, biquad design), and kernel launches on the context's stream.
// No CPU fallback exists anywhere in this file: without a HIP device every entry point fails.
// (The rest of the C-ABI: api_comm.hip, api_percall.hip, api_hrtf.hip, api_voices.hip, api_output.hip, api_callback.hip; shared: api_context.hpp.)
#include "api_context.hpp"
namespace oalgpu {

thread_local std::string gLastError;

int Fail(int code, const std::string &msg)
{
    gLastError = msg;
    return code;
}

// resident voice kernels (OALGPU_CTX_RESIDENT) running on `device` are told to leave: see ResidentState below
void ParkResidentContexts(int device);

// Every entry point that may allocate, copy synchronously or wait for the device comes through here.  A resident voice kernel
// ends only when its host says so, and anything that synchronises the device (hipFree, a blocking hipMemcpy, the null stream)
// would wait for it for ever: so whoever selects the device first parks the resident kernels on it (they finish the updates
// that have been rung and leave; the next oalgpu_mix_update of such a context launches a new one).
int UseDevice(int device)
{
    int count = 0;
    if(hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return Fail(OALGPU_ERR_NO_DEVICE, "no HIP device available (the product has no CPU path)");
    if(device < 0 || device >= count) return Fail(OALGPU_ERR_INVALID, "device ordinal out of range");
    HIP_TRY(hipSetDevice(device));
    ParkResidentContexts(device);
    return OALGPU_OK;
}

} // namespace oalgpu



// every entry point that enqueues work on a context or reads its state goes through here: the device (which parks a resident
// voice kernel: whatever follows on the main stream then runs behind its end), a parameter block that was waiting for a
// resident update, and the deferred update
int UseCtx(oalgpu_context *c)
{
    if(int rc = UseDevice(c->desc.device)) return rc;
    if(int rc = FlushResidentBlock(c)) return rc;
    return FlushPendingMix(c);
}
// the entry points a resident voice kernel stays through (they touch neither the main stream nor anything that synchronises the device)
int UseCtxResident(oalgpu_context *c)
{
    if(!c->res.running && !c->res.pendingBlock) return UseCtx(c);
    HIP_TRY(hipSetDevice(c->desc.device));
    return OALGPU_OK;
}

// ---- resident voice kernels: who is running, and how they are told to leave ----
namespace {
std::mutex gResLock;                        // submit / park of every context's resident state
std::vector<oalgpu_context*> gResRunning;
std::atomic<int> gResCount{0};

// (gResLock held) the running launch finishes the updates that have been rung and ends; nothing waits here
void ResidentParkLocked(oalgpu_context *c)
{
    auto &R = c->res;
    if(!R.running) return;
    __atomic_store_n(&R.door->exitSeq[R.launches & 3u], R.next, __ATOMIC_RELEASE);     // (R.launches: the running launch's id)
    __builtin_ia32_sfence();                // (write-combined stores through the BAR leave the core)
    R.running = false;
    ++R.parks;
    R.shortRuns = (R.next - R.launchBase < R.shortRun) ? R.shortRuns + 1u : 0u;
    if(R.shortRuns >= 3u) { R.shortRuns = 0u; R.cooldown = 192u; }
    const uint32_t e = (R.launches - 1u) % oalgpu_context::ResidentState::kEv;
    R.evLast[e] = R.next;
    gResRunning.erase(std::remove(gResRunning.begin(), gResRunning.end(), c), gResRunning.end());
    gResCount.store(int(gResRunning.size()), std::memory_order_relaxed);
}
} // namespace

namespace oalgpu {
void ParkResidentContexts(int device)
{
    if(gResCount.load(std::memory_order_relaxed) == 0) return;
    std::lock_guard<std::mutex> g(gResLock);
    std::vector<oalgpu_context*> run = gResRunning;
    for(oalgpu_context *c : run) if(c->desc.device == device) ResidentParkLocked(c);
}
} // namespace oalgpu

// (shared with the other C-ABI translation units: api_context.hpp)

// the voice slot is re-initialised as another kind of source: its callback is not asked any more
void RetireCallbackVoice(oalgpu_context *c, uint32_t voice)
{
    if(voice < c->cbOfVoice.size() && c->cbOfVoice[voice] >= 0)
    {
        auto &cb = c->cbVoices[size_t(c->cbOfVoice[voice])];
        cb.state = OALGPU_VOICE_STOPPED;
        cb.retired = true;                          // buffer slot, device and pinned memory wait for the next callback source
        c->cbOfVoice[voice] = -1;
    }
}

// a callback voice's mStep, wherever parameters pass through the host
void NoteCallbackSteps(oalgpu_context *c, const uint32_t *voices, const oalgpu_voice_params *params, size_t count)
{
    if(c->cbVoices.empty()) return;
    for(size_t i = 0; i < count; ++i)
        if(voices[i] < c->cbOfVoice.size() && c->cbOfVoice[voices[i]] >= 0)
            c->cbVoices[size_t(c->cbOfVoice[voices[i]])].step = params[i].step;
}

int FlushInits(oalgpu_context *c)
{
    if(c->initPending.empty()) return OALGPU_OK;
    const size_t n = c->initPending.size();
    if(c->initDev.n < n) HIP_TRY(c->initDev.alloc(n));
    HIP_TRY(hipMemcpyAsync(c->initDev.p, c->initPending.data(), n * sizeof(VoiceInitRecord), hipMemcpyHostToDevice, c->stream));
    LaunchInitVoices(c->stream, c->L, c->initDev.p, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));   // the host vector is reused
    c->initPending.clear();
    return OALGPU_OK;
}






static void SetRowsGroups(oalgpu_context *c);      // (voice_rows.hip's grid; defined beside RebalanceWaveGroups)

int oalgpu_context_create(const oalgpu_context_desc *desc, oalgpu_context **out)
{
    if(!desc || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    *out = nullptr;
    if(desc->max_voices == 0 || desc->num_dry_channels == 0 || desc->num_dry_channels > OALGPU_MAX_OUTPUT_CHANNELS
        || desc->num_aux_sends > OALGPU_MAX_SENDS || desc->wet_channels > OALGPU_MAX_AMBI_CHANNELS
        || (desc->num_aux_sends && (desc->num_slots == 0 || desc->wet_channels == 0)))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_context_create: bad descriptor");
    const uint32_t mixLines = (desc->hrtf ? 0u : desc->num_dry_channels) + desc->num_slots * desc->wet_channels;
    if(mixLines > 32) return Fail(OALGPU_ERR_CAPACITY, "more than 32 mixing lines (dry + wet) are not supported yet");
    if(int rc = UseDevice(desc->device)) return rc;

    auto c = std::make_unique<oalgpu_context>();
    c->desc = *desc;
    c->exact = desc->math_mode == OALGPU_MATH_EXACT;
    // The context's two streams must never share a hardware queue: HIP deals streams of one priority
    // class round-robin onto GPU_MAX_HW_QUEUES (default 4) queues, and a process that created other
    // streams first (torch, RCCL) can leave both on the same one -- the post stream's work then
    // serialises with the voice kernels (measured: 86 instead of 52 us per update).  Each priority
    // class has its own queues, so the main stream takes the highest and the post stream the lowest.
    int prioLeast = 0, prioGreatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLeast, &prioGreatest));
    HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prioGreatest));
    HIP_TRY(hipEventCreate(&c->evStart)); HIP_TRY(hipEventCreate(&c->evVoice)); HIP_TRY(hipEventCreate(&c->evEnd));
    HIP_TRY(hipStreamCreateWithPriority(&c->postStream, hipStreamDefault, prioLeast));
    for(hipEvent_t *e : {&c->evVoiceDone[0], &c->evVoiceDone[1], &c->evReduceDone[0], &c->evReduceDone[1], &c->evPostDone})
        // ordering between the context's two streams only: no system-scope fence (the default one
        // costs ~3.5 us of cache write-back per record on the stream it sits in -- measured, tools/
        // step_period.py; hosts and peers see the results through oalgpu_sync / stream order as before)
        HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming | hipEventDisableSystemFence));

    DeviceLayout &L = c->L;
    L.numVoices = desc->max_voices;
    L.numDry = desc->num_dry_channels; L.numReal = desc->num_real_channels;
    L.numSends = desc->num_aux_sends; L.numSlots = desc->num_slots; L.wetChannels = desc->wet_channels;
    L.hrtf = desc->hrtf ? 1u : 0u;
    L.irSize = 0; L.irStride = 8;
    L.mixLines = mixLines;
    c->slotConv.assign(desc->num_slots, nullptr);
    c->slotReverb.assign(desc->num_slots, nullptr);
    c->slotEffect.assign(desc->num_slots, nullptr);
    HIP_TRY(c->reverbTicket.alloc(1)); HIP_TRY(c->reverbTicket.zero());
    uint32_t vpg = desc->voices_per_group;
    if(vpg == 0)
    {
        // enough groups to put several workgroups on each of the 256 CUs, but no more partial
        // buses than needed: aim for ~1024 groups, at least 2 voices per group
        vpg = std::max<uint32_t>(2u, (desc->max_voices + 1023u) / 1024u);
    }
    L.voicesPerGroup = vpg;
    L.numGroups = std::max<uint32_t>(1u, (desc->max_voices + vpg - 1u) / vpg);
    L.waveVoices = 0;
    // the HRTF FIR of the wavefront kernel (IrSize <= 64): the matrix pipe in split half precision
    // (FirMfmaH, dev_wave.hpp) unless the host asks for packed fp32 VALU FMAs
    L.firMfma = (desc->flags & OALGPU_CTX_FIR_VALU) ? 0u : 1u;
    c->serialOnly = (desc->flags & OALGPU_CTX_SERIAL) != 0;
    c->res.enabled = (desc->flags & OALGPU_CTX_RESIDENT) != 0 && !(desc->flags & (OALGPU_CTX_PROFILE | OALGPU_CTX_SERIAL));
    c->useWave = WaveKernelApplies(c->exact, L);
    if(c->useWave)
    {
        // one wavefront per voice, 4 wavefronts per workgroup, two workgroups per CU: aim for
        // ~512 workgroups (2048 wavefronts); voices_per_group is then voices per WORKGROUP
        L.waveVoices = desc->voices_per_group ? std::max<uint32_t>(1u, (desc->voices_per_group + 3u) / 4u)
            : std::max<uint32_t>(1u, (desc->max_voices + 2047u) / 2048u);
        L.numGroups = std::max<uint32_t>(1u, WaveKernelGroups(L));
    }
    const TableBlob &blob = Blob();
    HIP_TRY(c->tables.alloc(blob.data.size())); HIP_TRY(c->tables.upload(blob.data.data(), blob.data.size()));
    L.tables = c->tables.p;
    HIP_TRY(c->buffers.alloc(std::max<uint32_t>(desc->max_buffers, 1u))); HIP_TRY(c->buffers.zero());
    L.buffers = c->buffers.p;
    c->bufferData.assign(std::max<uint32_t>(desc->max_buffers, 1u), nullptr);
    c->bufferLoopLen.assign(std::max<uint32_t>(desc->max_buffers, 1u), 0u);
    c->bufHost.assign(std::max<uint32_t>(desc->max_buffers, 1u), oalgpu_context::BufHost{});
    c->voiceHead.assign(std::max<uint32_t>(desc->max_voices, 1u), -1);
    c->queueDoneKnown.assign(std::max<uint32_t>(desc->max_voices, 1u), 0u);
    c->queueUnqueued.assign(std::max<uint32_t>(desc->max_voices, 1u), 0u);
    c->cbOfVoice.assign(std::max<uint32_t>(desc->max_voices, 1u), -1);

    const size_t nv = desc->max_voices;
    HIP_TRY(c->ctl.alloc(nv)); HIP_TRY(c->ctl.zero()); L.ctl = c->ctl.p;
    HIP_TRY(c->prev.alloc(nv * kMaxPad)); HIP_TRY(c->prev.zero()); L.prev = c->prev.p;
    HIP_TRY(c->dfilt.alloc(nv * 2)); HIP_TRY(c->dfilt.zero()); L.dfilt = c->dfilt.p;
    HIP_TRY(c->hist.alloc(nv * kHist)); HIP_TRY(c->hist.zero()); L.hist = c->hist.p;
    HIP_TRY(c->gainCur.alloc(nv * L.numDry)); HIP_TRY(c->gainCur.zero()); L.gainCur = c->gainCur.p;
    HIP_TRY(c->gainTgt.alloc(nv * L.numDry)); HIP_TRY(c->gainTgt.zero()); L.gainTgt = c->gainTgt.p;
    HIP_TRY(c->sfilt.alloc(nv * L.numSends * 2)); HIP_TRY(c->sfilt.zero()); L.sfilt = c->sfilt.p;
    HIP_TRY(c->sendCur.alloc(nv * L.numSends * L.wetChannels)); HIP_TRY(c->sendCur.zero()); L.sendCur = c->sendCur.p;
    HIP_TRY(c->sendTgt.alloc(nv * L.numSends * L.wetChannels)); HIP_TRY(c->sendTgt.zero()); L.sendTgt = c->sendTgt.p;
    HIP_TRY(c->ambi.alloc(nv)); HIP_TRY(c->ambi.zero()); L.ambi = c->ambi.p;
    HIP_TRY(c->startDelay.alloc(nv)); HIP_TRY(c->startDelay.zero()); L.startDelay = c->startDelay.p;
    {   // identity ambisonic maps until the host hands over the device's (ACN i, scale 1)
        std::vector<AmbiMapEntry> ident(std::max<size_t>(L.numDry, size_t{L.numSlots} * L.wetChannels) + 1);
        for(size_t i = 0; i < ident.size(); ++i) ident[i] = AmbiMapEntry{uint32_t(i < OALGPU_MAX_AMBI_CHANNELS ? i : 0), 1.0f};
        HIP_TRY(c->dryMap.alloc(L.numDry)); HIP_TRY(c->dryMap.upload(ident.data(), L.numDry));
        HIP_TRY(c->wetMaps.alloc(size_t{L.numSlots} * L.wetChannels));
        for(uint32_t s = 0; s < L.numSlots; ++s)
            HIP_TRY(hipMemcpy(c->wetMaps.p + size_t{s} * L.wetChannels, ident.data(), L.wetChannels * sizeof(AmbiMapEntry), hipMemcpyHostToDevice));
    }
    HIP_TRY(c->queueDone.alloc(nv)); HIP_TRY(c->queueDone.zero()); L.queueDone = c->queueDone.p;
    L.numLineGroups = L.numGroups;
    c->groupsAllocated = L.numGroups;
    L.streams = nullptr; L.lineGains = nullptr; L.lineStride = 0; L.streamsPerVoice = 0;
    L.nfc = nullptr; L.nfcOrders = 0;
    L.hrirs = nullptr;
    for(uint32_t &n : L.chansPerOrder) n = 0;
    L.accLines = 0;
    L.sliceLines = 0;
    L.wave16 = 0;                           // (decided when the HRTF data set is known: InstallHrtfData)
    L.rows8 = 0; L.rowsVpg = 0;
    if(c->useWave && !(desc->flags & OALGPU_CTX_STREAM_ROWS))
    {
        L.accLines = WaveKernelAccLines(L);
        // dry lines AND sends, or more lines than the wavefront-per-voice kernel holds in registers: stream rows, or -- opt-in,
        // OALGPU_CTX_SLICE_LINES -- a wavefront per 256-frame slice (voice_slice.hip: a third of the traffic, twice the time); the
        // measurement variants (OALGPU_CTX_PROFILE) exist for the stream-row kernel only
        if(!L.accLines && (desc->flags & OALGPU_CTX_SLICE_LINES) && !(desc->flags & OALGPU_CTX_PROFILE)) L.sliceLines = SliceKernelLines(L);
        // ... by default the rows stay in LDS: a wavefront per voice produces, a wavefront per 128-frame slice of every line
        // consumes, the round's filters are jobs dealt to all eight wavefronts (voice_rows.hip); one workgroup per compute unit
        // (SetRowsGroups below).  OALGPU_CTX_STREAM_ROWS keeps the rows in HBM (above); OALGPU_CTX_ROW_SLICES asks for this form by name.
        if(!L.accLines && !L.sliceLines && RowsKernelApplies(L))
            L.rows8 = 1;
    }
    if(c->useWave && (!L.hrtf || L.numSends))
    {   // one partial bus per workgroup, from the wavefronts' line accumulators (accLines) or from stream rows mixed by the
        // voice kernel's tail
        L.lineStride = L.mixLines <= 8 ? 8u : (L.mixLines <= 16 ? 16u : 32u);
        L.streamsPerVoice = 2u + L.numSends;
    }
    if(c->useWave && (!L.hrtf || L.numSends) && !L.accLines && !L.sliceLines && !L.rows8) { if(int rc = AllocStreamRows(c.get())) return rc; }
    HIP_TRY(c->partLines.alloc(size_t{L.numLineGroups} * L.mixLines * kLine)); L.partLines = c->partLines.p;
    // the two-stream pipeline of oalgpu_mix_update alternates between two sets of partial buses
    HIP_TRY(c->partLines2.alloc(c->useWave && (L.streams || L.accLines || L.sliceLines || L.rows8) ? size_t{L.numLineGroups} * L.mixLines * kLine : 0));
    c->partLinesBuf[0] = c->partLines.p; c->partLinesBuf[1] = c->partLines2.p;
    HIP_TRY(c->partHrtf.alloc(L.hrtf ? size_t{L.numGroups} * (kLine + kHrirLen) * 2 : 0)); L.partHrtf = c->partHrtf.p;
    HIP_TRY(c->partHrtf2.alloc(c->useWave && L.hrtf ? size_t{L.numGroups} * (kLine + kHrirLen) * 2 : 0));
    c->partHrtfBuf[0] = c->partHrtf.p; c->partHrtfBuf[1] = c->partHrtf2.p;
    HIP_TRY(c->bus.alloc(BusFloats(L))); HIP_TRY(c->bus.zero()); L.bus = c->bus.p;
    if(desc->flags & OALGPU_CTX_PROFILE)
    {
        HIP_TRY(c->phaseTimes.alloc(nv * 16)); HIP_TRY(c->phaseTimes.zero());   // [voice][8] | [wavefront][8]
        c->prof.times = c->phaseTimes.p;
    }
    // HRTF voice filters are sized when the data set is loaded
    L.hrtfOld = nullptr; L.hrtfTgt = nullptr;

    HIP_TRY(c->dSplit.alloc(L.numDry)); HIP_TRY(c->dSplit.zero());
    HIP_TRY(c->dSplit2.alloc(L.numDry)); HIP_TRY(c->dSplit2.zero());
    HIP_TRY(c->carryBuf.alloc(size_t{kLine + kHrirLen} * 2)); HIP_TRY(c->carryBuf.zero());
    HIP_TRY(c->postArrived.alloc(1)); HIP_TRY(c->postArrived.zero());
    HIP_TRY(c->dHfScale.alloc(L.numDry)); HIP_TRY(c->dHfScale.zero());
    HIP_TRY(c->dCoeffs.alloc(size_t{L.numDry} * kHrirLen * 2)); HIP_TRY(c->dCoeffs.zero());
    HIP_TRY(c->dTemp.alloc(size_t{L.numDry} * kLine + (kLine + kHrirLen) * 2));
    if(L.rows8) SetRowsGroups(c.get());
    *out = c.release();
    return OALGPU_OK;
}

void oalgpu_context_destroy(oalgpu_context *ctx)
{
    if(!ctx) return;
    (void)UseDevice(ctx->desc.device);          // (a resident voice kernel is told to leave)
    if(ctx->res.pendingBlock) { ctx->res.pendingBlock->heldBy = nullptr; ctx->res.pendingBlock = nullptr; }
    (void)FlushPendingMix(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if(ctx->res.reduceStream) (void)hipStreamSynchronize(ctx->res.reduceStream);
    if(ctx->postStream) (void)hipStreamSynchronize(ctx->postStream);
    delete ctx->comm;
    for(auto &cb : ctx->cbVoices)
        for(int k = 0; k < 2; ++k)
        {
            if(cb.pinned[k]) (void)hipHostFree(cb.pinned[k]);
            if(cb.copied[k]) (void)hipEventDestroy(cb.copied[k]);
        }
    delete ctx;
}

// stream rows [voice][streamsPerVoice][1024] and their gain blocks, for the contexts whose voice kernel mixes them in its tail
int AllocStreamRows(oalgpu_context *c)
{
    DeviceLayout &L = c->L;
    const size_t nv = L.numVoices;
    HIP_TRY(c->streams.alloc(nv * L.streamsPerVoice * kLine)); HIP_TRY(c->streams.zero()); L.streams = c->streams.p;
    HIP_TRY(c->lineGains.alloc(nv * L.streamsPerVoice * LineBlockDwords(L.lineStride))); HIP_TRY(c->lineGains.zero());
    L.lineGains = c->lineGains.p;
    return OALGPU_OK;
}

// EffectState::process of every slot that has an effect attached (alc/alu.cpp:2209-2257): from
// channel 0 of the slot's wet bus into the dry lines, on stream `s`, after the buses are final.
static int RunEffects(oalgpu_context *c, hipStream_t s, uint32_t samples_to_do)
{
    const DeviceLayout &L = c->L;
    for(uint32_t slot = 0; slot < L.numSlots; ++slot)
    {
        const float *wet = L.bus + BusWetOffset(L) + size_t{slot} * L.wetChannels * kLine;
        if(oalgpu_convolution *conv = c->slotConv[slot])
        {
            if(int rc = oalgpu_convolution_process_device(conv, s, wet, L.bus, samples_to_do)) return rc;
        }
        if(oalgpu_effect *fx = c->slotEffect[slot])
        {
            if(int rc = oalgpu_effect_process_device(fx, s, wet, L.bus, samples_to_do)) return rc;
        }
    }
    // the EAX reverbs of all slots: one launch, instances side by side, mix-out in slot order
    oalgpu_reverb *revs[kRvBatchMax];
    const float *wets[kRvBatchMax];
    uint32_t count = 0;
    auto flush = [&]() -> int
    {
        if(!count) return OALGPU_OK;
        const int rc = oalgpu_reverb_process_batch_device(revs, wets, count, L.bus, samples_to_do, s, c->reverbTicket.p);
        count = 0;
        return rc;
    };
    for(uint32_t slot = 0; slot < L.numSlots; ++slot)
    {
        if(!c->slotReverb[slot]) continue;
        revs[count] = c->slotReverb[slot];
        wets[count] = L.bus + BusWetOffset(L) + size_t{slot} * L.wetChannels * kLine;
        if(++count == kRvBatchMax) { if(int rc = flush()) return rc; }
    }
    if(int rc = flush()) return rc;
    return OALGPU_OK;
}

// Where the reduction finds the carried HrtfAccumData: in the bus block's accumulator region (in place: the serial post-process
// and contexts that leave the post-process to their caller work there) or where the fused post-process filed it.
static const float *CarrySource(oalgpu_context *c, bool carry)
{
    const float *src = nullptr;
    if(carry && c->L.hrtf) src = c->carryInBuf ? c->carryBuf.p : c->L.bus + BusAccumOffset(c->L);
    c->carryInBuf = false;                  // the reduction's result -- partial sums + carry -- is in the bus block again
    return src;
}

// BandSplitter::processHfScale's state transition over a run of `seg` samples of silence (core/filters/splitter.cpp:65-97: the
// recurrence SplitStep<true> of dev_wave.hpp, same operations): lower triangular in (lp_z1, lp_z2), decoupled in ap_z1 --
// [[p, 0, 0], [q, r, 0], [0, 0, s]].  Data independent and the same for every channel: raised here, once per update size, instead
// of by every wavefront that scans a channel.
static void SplitterRunPowers(float coeff, uint32_t seg, float out[4])
{
    const float ap = coeff, lp = coeff * 0.5f + 0.5f;
    float st[3][3] = {{1.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 1.0f}};
    for(uint32_t i = 0; i < seg; ++i)
        for(auto &v : st)
        {
            const float d0 = (0.0f - v[0]) * lp;
            const float lpY0 = v[0] + d0;
            v[0] = std::fmaf(d0, lp, lpY0);
            const float d1 = (lpY0 - v[1]) * lp;
            const float lpY1 = v[1] + d1;
            v[1] = lpY1 + d1;
            const float apY = std::fmaf(0.0f, ap, v[2]);
            v[2] = std::fmaf(-apY, ap, 0.0f);
        }
    out[0] = st[0][0]; out[1] = st[0][1]; out[2] = st[1][1]; out[3] = st[2][2];
}

// MixDirectHrtf of a FAST wavefront-kernel context in one launch (post_wave.hip): reads the bus block's accumulator, leaves the
// shifted accumulator in carryBuf and the new splitter states in the other state buffer.
// resident: the update's reduction is BusReduceResidentKernel on the reduce stream (ResidentSubmit): the launch waits for ITS
// counter and counts itself in for the next one.
// The counters the kernels wait for only ever grow and the host keeps what they will read (postEpoch, reducedEpoch, outSeq, the
// resident targets): those mirrors move only once the launch has been accepted -- a failed launch must not leave every later
// one waiting for a count that never comes.
static int PostDirectHrtfFused(oalgpu_context *c, hipStream_t s, uint32_t samples_to_do, hipEvent_t evDone, bool resident = false)
{
    const DeviceLayout &L = c->L;
    float *left = L.bus + size_t{L.numDry} * kLine;
    SplitterState *spIn = c->dSplitCur ? c->dSplit2.p : c->dSplit.p, *spOut = c->dSplitCur ? c->dSplit.p : c->dSplit2.p;
    const uint32_t postEpoch = c->postEpoch + L.numDry;    // what the channel counter reads when this update's channels have all arrived
    const uint32_t seg = ((samples_to_do + 63u) / 64u) | 1u;
    if(c->runPowerSeg != seg) { SplitterRunPowers(c->dSplitCoeff, seg, c->runPower); c->runPowerSeg = seg; }
    const uint32_t slot = c->outNext % oalgpu_context::kIoSlots;
    const bool ring = c->outRing && c->outFlags;
    const uint32_t outTarget = c->outArrivedTotal + PostResidentFirGroups();
    auto &R = c->res;
    if(resident)
    {
        c->reduceHeld = false;
        LaunchPostResident(s, left, left + kLine, L.bus, L.numDry, L.bus + BusAccumOffset(L), c->carryBuf.p, spIn, spOut, c->dHfScale.p, c->dCoeffs.p,
            c->dIrSize, samples_to_do, c->dTemp.p, c->postArrived.p, postEpoch, c->runPower, evDone, ring ? c->outHost[slot] : nullptr,
            ring ? c->outFlags + size_t{slot} * 16 : nullptr, c->outSeq + 1u, R.counters.p, R.hostFlags, (R.posts + 1u) * R.redGroups,
            (R.posts + 1u) * R.firGroups, R.posts + 1u);
        HIP_TRY(hipGetLastError());
        ++R.posts;
    }
    else if(c->reduceHeld)
    {   // the update's reduction rides in the same launch (its partial buses: heldL's)
        c->reduceHeld = false;
        if(!c->reducedCount.p) { HIP_TRY(c->reducedCount.alloc(1)); HIP_TRY(c->reducedCount.zero()); }
        const uint32_t reducedEpoch = c->reducedEpoch + ReducePostReduceGroups(c->heldL);
        LaunchReducePostFused(s, c->heldL, CarrySource(c, c->carryAccum && L.hrtf), left, left + kLine, L.bus, L.numDry, L.bus + BusAccumOffset(L),
            c->carryBuf.p, spIn, spOut, c->dHfScale.p, c->dCoeffs.p, c->dIrSize, samples_to_do, c->dTemp.p, c->postArrived.p, postEpoch,
            c->runPower, evDone, ring ? c->outHost[slot] : nullptr, ring ? c->outFlags + size_t{slot} * 16 : nullptr, c->outSeq + 1u,
            c->outArrived.p, outTarget, c->reducedCount.p, reducedEpoch);
        HIP_TRY(hipGetLastError());
        c->reducedEpoch = reducedEpoch;
        if(ring) c->outArrivedTotal = outTarget;
    }
    else
    {
        LaunchPostDirectHrtfFused(s, left, left + kLine, L.bus, L.numDry, L.bus + BusAccumOffset(L), c->carryBuf.p, spIn, spOut,
            c->dHfScale.p, c->dCoeffs.p, c->dIrSize, samples_to_do, c->dTemp.p, c->postArrived.p, postEpoch, c->runPower, evDone,
            ring ? c->outHost[slot] : nullptr, ring ? c->outFlags + size_t{slot} * 16 : nullptr, c->outSeq + 1u, c->outArrived.p, outTarget);
        HIP_TRY(hipGetLastError());
        if(ring) c->outArrivedTotal = outTarget;
    }
    c->postEpoch = postEpoch;
    if(ring) c->outSlotSeq[slot] = ++c->outSeq;
    c->outRingWritten = ring;
    c->dSplitCur ^= 1u;
    c->carryInBuf = true;
    return OALGPU_OK;
}

// Orders the main stream behind whatever a pipelined oalgpu_mix_update left on the post stream.
int JoinPost(oalgpu_context *c)
{
    if(!c->postPending) return OALGPU_OK;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->lastPostEvent ? c->lastPostEvent : c->evPostDone, 0));
    c->postPending = false;
    return OALGPU_OK;
}


// ---- the resident voice kernel: host side (device side and protocol: kernels.hpp ResidentDoor, voice_wave.hip, post_wave.hip) ----
bool ResidentWanted(const oalgpu_context *c, int post_process)
{
    const auto &R = c->res;
    return R.enabled && !R.failed && WaveKernelHasResident(c->L) && post_process && c->hrtfLoaded && c->directSet && !c->timing && c->cbVoices.empty()
        && c->initPending.empty() && c->carryAccum && !c->comm && !c->pendingMix.active && c->useWave && c->ownStream && !c->serialOnly
        && c->L.numReal >= 2 && c->L.numSlots == 0;
}

// a parameter block that was waiting for a resident update is applied the launched way (the caller has parked the kernel)
int FlushResidentBlock(oalgpu_context *c)
{
    oalgpu_param_block *b = c->res.pendingBlock;
    if(!b) return OALGPU_OK;
    c->res.pendingBlock = nullptr;
    b->heldBy = nullptr;
    if(int rc = FlushInits(c)) return rc;
    LaunchApplyParams(c->stream, c->L, b->recs.p, b->count);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

static int ResidentGiveUp(oalgpu_context *c, const std::string &why)
{
    c->res.failed = true;
    return Fail(OALGPU_ERR_HIP, "resident voice kernel: " + why + " (the context launches per update from now on)");
}

// streams, the door, counters, partial sets: once per context, with nothing resident on the device
static int ResidentInit(oalgpu_context *c)
{
    auto &R = c->res;
    if(int rc = UseDevice(c->desc.device)) return rc;
    hipDeviceProp_t prop{};
    HIP_TRY(hipGetDeviceProperties(&prop, c->desc.device));
    // every workgroup of the launch has to be on the machine at once: nothing ever leaves to make room
    R.groupsPerCu = uint32_t(std::max(0, WaveResidentGroupsPerCu(c->L)));
    if(uint64_t{R.groupsPerCu} * uint32_t(prop.multiProcessorCount) < c->L.numGroups)
        return ResidentGiveUp(c, "the device does not hold all of the launch's workgroups at once");
    // the reduce stream between the main stream's priority class (highest) and the post stream's (lowest): a class has its own
    // hardware queues, and a queue the resident kernel sits in never moves
    int prioLeast = 0, prioGreatest = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prioLeast, &prioGreatest));
    if(prioLeast - prioGreatest < 2) return ResidentGiveUp(c, "fewer than three stream priority classes");
    HIP_TRY(hipStreamCreateWithPriority(&R.reduceStream, hipStreamDefault, (prioLeast + prioGreatest) / 2));
    R.doorInBar = HostStoresReachDevice(c);
    if(R.doorInBar) HIP_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&R.door), sizeof(ResidentDoor), hipDeviceMallocFinegrained));
    else HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&R.door), sizeof(ResidentDoor), hipHostMallocDefault));
    std::memset(R.door, 0, sizeof(ResidentDoor));
    for(uint32_t &e : R.door->exitSeq) e = 0x40000000u;
    __builtin_ia32_sfence();
    HIP_TRY(R.counters.alloc(size_t{kRcCount} * 16)); HIP_TRY(R.counters.zero());
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&R.hostFlags), size_t{kRhCount} * 16 * sizeof(uint32_t), hipHostMallocDefault));
    std::memset(R.hostFlags, 0, size_t{kRhCount} * 16 * sizeof(uint32_t));
    R.setFloats = size_t{c->L.numGroups} * (kLine + kHrirLen) * 2;
    HIP_TRY(R.part.alloc(R.setFloats * kResidentSets));
    for(uint32_t k = 0; k < oalgpu_context::ResidentState::kEv; ++k) { HIP_TRY(hipEventCreate(&R.evStart[k])); HIP_TRY(hipEventCreate(&R.evStop[k])); }
    R.firGroups = PostResidentFirGroups();
    R.redGroups = uint32_t((BusFloats(c->L) + 63u) / 64u);
    R.ready = true;
    return OALGPU_OK;
}

// the launches whose events have fired hand over their times
static void ResidentCollectTimes(oalgpu_context *c, bool all)
{
    auto &R = c->res;
    for(uint32_t k = 0; k < oalgpu_context::ResidentState::kEv; ++k)
    {
        if(!R.evOpen[k] || (R.running && k == (R.launches - 1u) % oalgpu_context::ResidentState::kEv)) continue;
        if(!all && hipEventQuery(R.evStop[k]) != hipSuccess) { (void)hipGetLastError(); continue; }
        float ms = 0.0f;
        if(hipEventElapsedTime(&ms, R.evStart[k], R.evStop[k]) == hipSuccess)
        {
            R.kernelMs += double(ms); R.kernelUpdates += R.evLast[k] - R.evFirst[k]; ++R.kernelLaunches;
        }
        else (void)hipGetLastError();
        R.evOpen[k] = false;
    }
}

static int ResidentCheckError(oalgpu_context *c)
{
    auto &R = c->res;
    if(!R.hostFlags) return OALGPU_OK;
    const uint32_t e = __atomic_load_n(R.hostFlags + 16u * kRhError, __ATOMIC_ACQUIRE);
    if(!e) return OALGPU_OK;
    __atomic_store_n(R.hostFlags + 16u * kRhError, 0u, __ATOMIC_RELEASE);
    static const char *what[4] = {"", "the voice kernel waited 2 s for the host or for its reduction", "a reduction waited 2 s for the voice kernel",
        "a reduction waited 2 s for the post-process"};
    // where everything stood: the counters the kernels wait for, beside what the host expects them to reach
    std::string state;
    {
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(R.reduceStream); (void)hipStreamSynchronize(c->postStream);
        uint32_t w[kRcCount * 16] = {};
        uint32_t pa = 0;
        if(hipMemcpy(w, R.counters.p, sizeof(w), hipMemcpyDeviceToHost) == hipSuccess
            && hipMemcpy(&pa, c->postArrived.p, sizeof(pa), hipMemcpyDeviceToHost) == hipSuccess)
        {
            char buf[512];
            std::snprintf(buf, sizeof(buf), " [updates %u, launches %u, groups %u; arrive %u %u %u %u (uses %u %u %u %u), redRead %u redDone %u (per update %u), "
                "postDone %u (posts %u x %u), postArrived %u (epoch %u), started %u (expected %u), progress %u]", R.next, R.launches, c->L.numGroups,
                w[0], w[16], w[32], w[48], R.setUses[0], R.setUses[1], R.setUses[2], R.setUses[3], w[16 * kRcRedRead], w[16 * kRcRedDone], R.redGroups,
                w[16 * kRcPostDone], R.posts, R.firGroups, pa, c->postEpoch, w[16 * kRcStarted], R.startedTotal,
                __atomic_load_n(R.hostFlags + 16u * kRhProgress, __ATOMIC_ACQUIRE));
            state = buf;
            const uint32_t *fi = R.hostFlags + 16u * kRhFault;
            std::snprintf(buf, sizeof(buf), " [the voice workgroup that gave up: update %u, doorbell %u, exit word %u, reduction counter %u of %u, workgroup %u, launch %u, "
                "%u ticks; host: seq %u exit %u %u %u %u]", fi[0], fi[1], fi[2], fi[3], fi[4], fi[5], fi[6], fi[7], R.door->seq, R.door->exitSeq[0], R.door->exitSeq[1],
                R.door->exitSeq[2], R.door->exitSeq[3]);
            if(e == 1) state += buf;
        }
        else (void)hipGetLastError();
    }
    return ResidentGiveUp(c, std::string(e < 4 ? what[e] : "a wait timed out") + state);
}

// One update of a resident context: the doorbell, its reduction (reduce stream) and its post-process (post stream).
// Returns 1 when the update has to go the launched way after all (the caller falls through), 0 when submitted, < 0 on errors.
static int ResidentSubmit(oalgpu_context *c, uint32_t samples_to_do)
{
    auto &R = c->res;
    if(!R.ready) { if(int rc = ResidentInit(c)) return R.failed ? 1 : rc; }
    HIP_TRY(hipSetDevice(c->desc.device));
    using clk = std::chrono::steady_clock;
    std::unique_lock<std::mutex> g(gResLock);
    const DeviceLayout &L = c->L;
    // a new launch: prepared here, started BEHIND the update's doorbell (the kernel finds its first update rung when it comes up)
    const bool launchNow = !R.running;
    ResidentArgs a{};
    uint32_t evk = 0;
    bool timed = false;
    if(launchNow)
    {
        // the reduce stream joins whatever the post stream still runs (the bus block and the carried accumulator are theirs too)
        if(c->postPending) HIP_TRY(hipStreamWaitEvent(R.reduceStream, c->lastPostEvent ? c->lastPostEvent : c->evPostDone, 0));
        const uint32_t k = evk = R.launches % oalgpu_context::ResidentState::kEv;
        if(R.evOpen[k]) { HIP_TRY(hipEventSynchronize(R.evStop[k])); ResidentCollectTimes(c, false); }
        timed = R.timeLaunches;           // (events bound to the dispatch cost the launch call ~15 us of host time)
        a.door = R.door; a.counters = R.counters.p; a.hostFlags = R.hostFlags; a.partBase = R.part.p; a.setStride = uint32_t(R.setFloats);
        a.base = R.next; a.endSeq = R.next + R.maxUpdates; a.redPerUpdate = R.redGroups;
        a.startedTarget = R.startedTotal + L.numGroups; a.launchId = R.launches + 1u;
        __atomic_store_n(&R.door->exitSeq[a.launchId & 3u], R.next + 0x40000000u, __ATOMIC_RELEASE);
        __atomic_store_n(&R.door->seq, R.next, __ATOMIC_RELEASE);
        __builtin_ia32_sfence();
    }
    {   // the host stays at most kResidentDepth updates ahead of the post-process (doorbell slots, queue depth)
        const auto deadline = clk::now() + std::chrono::seconds(5);
        uint32_t spins = 0;
        while(int32_t(R.posts - __atomic_load_n(R.hostFlags + 16u * kRhProgress, __ATOMIC_ACQUIRE)) >= int32_t(kResidentDepth))
        {
            __builtin_ia32_pause();
            if((++spins & 0x3ffu) == 0 && clk::now() > deadline)
            {
                ResidentParkLocked(c);
                g.unlock();
                (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(R.reduceStream); (void)hipStreamSynchronize(c->postStream);
                if(int rc = ResidentCheckError(c)) return rc;
                return ResidentGiveUp(c, "the post-process made no progress for 5 s");
            }
        }
    }
    if(__atomic_load_n(R.hostFlags + 16u * kRhError, __ATOMIC_RELAXED))
    {
        ResidentParkLocked(c);
        g.unlock();
        (void)hipStreamSynchronize(c->stream); (void)hipStreamSynchronize(R.reduceStream); (void)hipStreamSynchronize(c->postStream);
        return ResidentCheckError(c);
    }
    c->outRingWritten = false;
    // ---- the doorbell: the slot first, then the sequence number
    oalgpu_param_block *b = R.pendingBlock;
    R.pendingBlock = nullptr;
    if(b) b->heldBy = nullptr;
    ResidentSlot &sl = R.door->slot[R.next % kResidentSlots];
    sl.recs = b ? reinterpret_cast<unsigned long long>(b->recs.p) : 0ull;
    sl.map = b ? reinterpret_cast<unsigned long long>(b->voiceToRec.p) : 0ull;
    sl.rows = b ? reinterpret_cast<unsigned long long>(b->rows.p) : 0ull;
    sl.samples = samples_to_do;
    __builtin_ia32_sfence();
    __atomic_store_n(&R.door->seq, R.next + 1u, __ATOMIC_RELEASE);
    __builtin_ia32_sfence();
    if(launchNow)
    {
        const uint32_t k = evk;
        HIP_TRY(LaunchVoiceWaveResident(c->stream, L, a, timed ? R.evStart[k] : nullptr, timed ? R.evStop[k] : nullptr));
        R.evOpen[k] = timed; R.evFirst[k] = R.next; R.evLast[k] = R.next;
        ++R.launches; R.startedTotal = a.startedTarget; R.endSeq = a.endSeq; R.launchBase = R.next;
        // Nothing that waits for this kernel may get onto the machine in front of it: a reduction that polls for workgroups which
        // find no room beside it would wait for ever.  The last workgroup to start says so (a pinned word): the doorbell is rung at
        // once -- the workgroups that are there start on the update -- and the host looks for the word only in front of the first
        // reduction it launches (awaitStarted), which nobody needs before the voices of the update are through.
        R.awaitStarted = a.launchId;
        R.running = true;
        gResRunning.push_back(c);
        gResCount.store(int(gResRunning.size()), std::memory_order_relaxed);
    }
    // ---- the update's reduction and post-process: launches of their own that wait for device counters
    const uint32_t set = R.next % kResidentSets;
    if(R.awaitStarted)
    {
        const uint32_t id = R.awaitStarted;
        R.awaitStarted = 0u;
        const auto deadline = clk::now() + std::chrono::seconds(5);
        uint32_t spins = 0;
        while(__atomic_load_n(R.hostFlags + 16u * kRhResident, __ATOMIC_ACQUIRE) != id)
        {
            __builtin_ia32_pause();
            if((++spins & 0x3ffu) == 0 && clk::now() > deadline)
            {   // (it may still start, and the update has been rung: it is told to leave behind that update; the update is lost)
                ++R.next;
                ResidentParkLocked(c);
                g.unlock();
                (void)hipStreamSynchronize(c->stream);
                return ResidentGiveUp(c, "its workgroups did not all start within 5 s");
            }
        }
    }
    if(R.copyPending) { HIP_TRY(hipStreamWaitEvent(R.reduceStream, R.copyPending, 0)); R.copyPending = nullptr; }
    DeviceLayout Lr = L;
    Lr.partHrtf = R.part.p + size_t{set} * R.setFloats;
    LaunchBusReduceResident(R.reduceStream, Lr, CarrySource(c, true), R.counters.p, R.hostFlags, set, (R.setUses[set] + 1u) * L.numGroups,
        R.posts * R.firGroups);
    HIP_TRY(hipGetLastError());
    ++R.setUses[set];
    c->lastPostEvent = c->evPostDone;
    if(int rc = PostDirectHrtfFused(c, c->postStream, samples_to_do, c->evPostDone, true)) return rc;
    c->postPending = true;
    ++R.next;
    ++c->updatesSubmitted;
    R.evLast[(R.launches - 1u) % oalgpu_context::ResidentState::kEv] = R.next;
    if(R.next == R.endSeq)
    {   // the launch's own bound: it leaves by itself behind this update; the next one starts a new launch
        R.running = false;
        gResRunning.erase(std::remove(gResRunning.begin(), gResRunning.end(), c), gResRunning.end());
        gResCount.store(int(gResRunning.size()), std::memory_order_relaxed);
    }
    return OALGPU_OK;
}

int oalgpu_mix_voices(oalgpu_context *c, uint32_t samples_to_do)
{
    if(!c || samples_to_do == 0 || samples_to_do > kLine) return Fail(OALGPU_ERR_INVALID, "samples_to_do must be 1..1024");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(int rc = JoinPost(c)) return rc;
    if(!c->cbVoices.empty()) { if(int rc = ServiceCallbacks(c, samples_to_do)) return rc; }
    c->outRingWritten = false;
    if(c->useWave)    // (timing: the two events are bound to the dispatch itself -- the kernel's own start and end)
        HIP_TRY(LaunchVoiceWave(c->stream, c->L, samples_to_do, c->profArg(), c->timing ? c->evStart : nullptr, c->timing ? c->evVoice : nullptr));
    else
    {
        if(c->timing) HIP_TRY(hipEventRecord(c->evStart, c->stream));
        HIP_TRY(LaunchVoiceMix(c->stream, c->exact, c->L, samples_to_do, c->carryAccum));
        if(c->timing) HIP_TRY(hipEventRecord(c->evVoice, c->stream));
    }
    // the wavefront kernel leaves the carried HRTF accumulator tail to the reduction
    LaunchBusReduce(c->stream, c->L, samples_to_do, CarrySource(c, c->useWave && c->carryAccum));
    HIP_TRY(hipGetLastError());
    if(int rc = CommReduceBus(c, c->stream)) return rc;
    if(c->timing) { HIP_TRY(hipEventRecord(c->evEnd, c->stream)); c->timed = true; }
    return OALGPU_OK;
}

int oalgpu_post_process(oalgpu_context *c, uint32_t samples_to_do)
{
    if(!c || samples_to_do == 0 || samples_to_do > kLine) return Fail(OALGPU_ERR_INVALID, "samples_to_do must be 1..1024");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = JoinPost(c)) return rc;
    if(int rc = RunEffects(c, c->stream, samples_to_do)) return rc;
    if(!c->L.hrtf)
    {   // DeviceBase::Process(AmbiDecPostProcess), alc/alu.cpp:282-287: dry lines -> speaker feeds
        if(c->decOn)
        {
            const DeviceLayout &D = c->L;
            LaunchBFormatDecode(c->stream, c->exact, D.bus + size_t{D.numDry} * kLine, D.bus, c->decSplit.p, c->decBands.p,
                c->decGainsHf.p, c->decDual ? c->decGainsLf.p : nullptr, D.numDry, c->decOut, samples_to_do);
            HIP_TRY(hipGetLastError());
        }
        if(c->timing) { HIP_TRY(hipEventRecord(c->evEnd, c->stream)); c->timed = true; }
        return OALGPU_OK;
    }
    const DeviceLayout &L = c->L;
    if(L.numReal < 2) return Fail(OALGPU_ERR_INVALID, "HRTF post-process needs two real output lines");
    float *left = L.bus + size_t{L.numDry} * kLine;
    float *right = left + kLine;
    // (the wavefront-kernel contexts carry the accumulator through their reduction: the one-launch form; the others mix it in place)
    SplitterState *spCur = c->dSplitCur ? c->dSplit2.p : c->dSplit.p;
    if(c->exact)
        LaunchMixDirectHrtf(c->stream, true, left, right, L.bus, L.numDry, L.bus + BusAccumOffset(L), spCur,
            c->dHfScale.p, c->dCoeffs.p, c->dIrSize, samples_to_do, c->dTemp.p);
    else if(c->useWave) { if(int rc = PostDirectHrtfFused(c, c->stream, samples_to_do, nullptr)) return rc; }
    else
        LaunchPostDirectHrtfFast(c->stream, left, right, L.bus, L.numDry, L.bus + BusAccumOffset(L), spCur,
            c->dHfScale.p, c->dCoeffs.p, c->dIrSize, samples_to_do, c->dTemp.p);
    HIP_TRY(hipGetLastError());
    if(c->timing) { HIP_TRY(hipEventRecord(c->evEnd, c->stream)); c->timed = true; }
    return OALGPU_OK;
}

static int RunMixUpdate(oalgpu_context *c, uint32_t samples_to_do, int post_process, oalgpu_param_block *next);

int oalgpu_mix_update(oalgpu_context *c, uint32_t samples_to_do, int post_process)
{
    if(!c || samples_to_do == 0 || samples_to_do > kLine) return Fail(OALGPU_ERR_INVALID, "samples_to_do must be 1..1024");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(c->res.cooldown) --c->res.cooldown;
    else if(ResidentWanted(c, post_process))
    {   // the resident voice kernel: a doorbell, the update's reduction and its post-process
        const int rc = ResidentSubmit(c, samples_to_do);
        if(rc <= 0) return rc;                       // (1: the mode is not available after all -- on to the launched path)
    }
    if(int rc = UseCtx(c)) return rc;                // (submits the update deferred before this one)
    if(c->useWave && c->ownStream && !c->serialOnly && !c->timing && (c->desc.flags & OALGPU_CTX_APPLY_IN_VOICE_KERNEL) && WaveKernelAppliesRecords(c->L)
        && !(c->res.enabled && !c->res.failed && !c->res.cooldown))
    {   // submitted with the next library call on this context (see pendingMix); whatever goes wrong then is that call's error
        c->pendingMix.active = true; c->pendingMix.samples = samples_to_do; c->pendingMix.post = post_process;
        return OALGPU_OK;
    }
    return RunMixUpdate(c, samples_to_do, post_process, nullptr);
}

int FlushPendingMix(oalgpu_context *c, oalgpu_param_block *next)
{
    if(!c->pendingMix.active) return OALGPU_OK;
    c->pendingMix.active = false;
    return RunMixUpdate(c, c->pendingMix.samples, c->pendingMix.post, next);
}

// next: a parameter block the update's voice kernel installs behind the voices it mixed (null: none)
static int RunMixUpdate(oalgpu_context *c, uint32_t samples_to_do, int post_process, oalgpu_param_block *next)
{
    c->nextRecs = next ? next->recs.p : nullptr;
    c->nextMap = next ? next->voiceToRec.p : nullptr;
    c->nextRows = next ? next->rows.p : nullptr;
    struct Clear { oalgpu_context *c; ~Clear() { c->nextRecs = nullptr; c->nextMap = nullptr; c->nextRows = nullptr; } } clear{c};
    if(!(c->useWave && c->ownStream) || c->serialOnly)
    {   // one stream: the workgroup-per-voice-group kernel reads the carried accumulator itself,
        // and a caller-owned stream (RCCL ordering) is never forked
        if(int rc = oalgpu_mix_voices(c, samples_to_do)) return rc;
        if(post_process && c->commRank == 0) return oalgpu_post_process(c, samples_to_do);
        return OALGPU_OK;
    }
    c->fuseReduce = post_process && !c->comm && c->L.hrtf && c->L.numSlots == 0 && c->L.numReal >= 2 && !c->timing
        && (c->desc.flags & OALGPU_CTX_FUSED_REDUCE);
    const int rcv = oalgpu_mix_voices_overlapped(c, samples_to_do);
    c->fuseReduce = false;
    if(rcv) { c->reduceHeld = false; return rcv; }
    // sharded contexts: the effects and the post-process run where the reduced buses are, on rank 0
    return oalgpu_post_process_overlapped(c, samples_to_do, post_process && c->commRank == 0);
}

/* `count` consecutive updates in one call: update i applies param_blocks[i] (the array or an entry may be NULL) and mixes
 * -- the loop a C++ host would write, without a language binding's per-call cost between the submissions */
int oalgpu_mix_update_run(oalgpu_context *c, oalgpu_param_block *const *param_blocks, uint32_t count, uint32_t samples_to_do,
    int post_process)
{
    if(!c || count == 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_update_run: bad arguments");
    for(uint32_t i = 0; i < count; ++i)
    {
        if(param_blocks && param_blocks[i]) { if(int rc = oalgpu_param_block_apply(c, param_blocks[i])) return rc; }
        if(int rc = oalgpu_mix_update(c, samples_to_do, post_process)) return rc;
    }
    return OALGPU_OK;
}

int oalgpu_mix_voices_overlapped(oalgpu_context *c, uint32_t samples_to_do)
{
    if(!c || samples_to_do == 0 || samples_to_do > kLine) return Fail(OALGPU_ERR_INVALID, "samples_to_do must be 1..1024");
    if(!(c->useWave && c->ownStream))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_voices_overlapped: needs a FAST context (wavefront kernel) on its own streams");
    if(c->L.hrtf && !c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "HRTF context without a data set");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = FlushInits(c)) return rc;
    if(!c->cbVoices.empty()) { if(int rc = ServiceCallbacks(c, samples_to_do)) return rc; }
    c->outRingWritten = false;
    const uint32_t p = c->parity;
    DeviceLayout L = c->L;
    L.partHrtf = c->partHrtfBuf[p];
    if(L.streams || L.accLines || L.sliceLines || L.rows8) L.partLines = c->partLinesBuf[p];
    // main stream: this update's voices; its partial-bus buffer was last read by the reduction
    // of two updates ago
    // (almost always long done: then no barrier packet goes into the main queue in front of the voice kernel)
    if(c->reduceUpdate[p] > c->updatesKnownDone)
    {
        const hipError_t q = hipEventQuery(c->evReduceDone[p]);
        if(q == hipErrorNotReady) HIP_TRY(hipStreamWaitEvent(c->stream, c->evReduceDone[p], 0));
        else HIP_TRY(q);
    }
    c->reduceUpdate[p] = ++c->updatesSubmitted;
    // (timing: the two events are bound to the dispatch itself -- the kernel's own start and end)
    // The event the post stream waits for is bound to the voice kernel's dispatch (hipExtLaunchKernel's stop event: one
    // runtime call less per update than a record behind the launch).  Timing runs use that slot for their own event.
    HIP_TRY(LaunchVoiceWave(c->stream, L, samples_to_do, c->profArg(), c->timing ? c->evStart : nullptr, c->timing ? c->evVoice : c->evVoiceDone[p],
        c->nextRecs, c->nextMap, c->nextRows));
    if(c->timing) HIP_TRY(hipEventRecord(c->evVoiceDone[p], c->stream));
    // post stream: the reduction (adds the carried HRTF accumulator tail); whatever follows on that stream -- a collective, the effects, the post-process -- runs beside
    // the next update's parameter and voice kernels
    HIP_TRY(hipStreamWaitEvent(c->postStream, c->evVoiceDone[p], 0));
    // (4-wavefront workgroups: they find room on a CU as soon as ONE of the next update's voice workgroups
    // has left it; the 16-wavefront form waits for a whole CU -- measured 62 against 53 us per config-2 step)
    if(c->fuseReduce)
    {   // (oalgpu_mix_update: the reduction is launched together with the post-process that follows at once)
        c->reduceHeld = true; c->heldL = L; c->heldParity = p;
        c->parity = p ^ 1u;
        return OALGPU_OK;
    }
    LaunchBusReduce(c->postStream, L, samples_to_do, CarrySource(c, c->carryAccum && L.hrtf), true, c->evReduceDone[p]);
    HIP_TRY(hipGetLastError());
    if(int rc = CommReduceBus(c, c->postStream)) return rc;      // beside the next update's voice kernel
    c->parity = p ^ 1u;
    return OALGPU_OK;
}

void *oalgpu_post_stream(oalgpu_context *c) { return c ? static_cast<void*>(c->postStream) : nullptr; }

int oalgpu_post_process_overlapped(oalgpu_context *c, uint32_t samples_to_do, int post_process)
{
    if(!c || samples_to_do == 0 || samples_to_do > kLine) return Fail(OALGPU_ERR_INVALID, "samples_to_do must be 1..1024");
    if(!(c->useWave && c->ownStream))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_post_process_overlapped: needs a FAST context (wavefront kernel) on its own streams");
    if(post_process && c->L.hrtf && c->L.numReal < 2) return Fail(OALGPU_ERR_INVALID, "HRTF post-process needs two real output lines");
    if(int rc = UseCtx(c)) return rc;
    const DeviceLayout &L = c->L;
    bool postDoneBound = false;
    if(post_process) { if(int rc = RunEffects(c, c->postStream, samples_to_do)) return rc; }
    if(post_process && L.hrtf)
    {
        // (the update's last launch on this stream, unless timing asks for an event of its own behind it: evPostDone rides on it)
        // (with the reduction in the same launch the event is the one the voice kernel of two updates on waits for as well)
        hipEvent_t ev = c->reduceHeld ? c->evReduceDone[c->heldParity] : (c->timing ? nullptr : c->evPostDone);
        c->lastPostEvent = ev ? ev : c->evPostDone;
        if(int rc = PostDirectHrtfFused(c, c->postStream, samples_to_do, ev)) return rc;
        postDoneBound = ev != nullptr;
    }
    else c->lastPostEvent = c->evPostDone;
    if(post_process && !L.hrtf && c->decOn)
    {   // DeviceBase::Process(AmbiDecPostProcess), alc/alu.cpp:282-287: dry lines -> speaker feeds
        LaunchBFormatDecode(c->postStream, c->exact, L.bus + size_t{L.numDry} * kLine, L.bus, c->decSplit.p, c->decBands.p,
            c->decGainsHf.p, c->decDual ? c->decGainsLf.p : nullptr, L.numDry, c->decOut, samples_to_do);
        HIP_TRY(hipGetLastError());
    }
    if(c->timing) { HIP_TRY(hipEventRecord(c->evEnd, c->postStream)); c->timed = true; }
    if(!postDoneBound) HIP_TRY(hipEventRecord(c->evPostDone, c->postStream));
    c->postPending = true;
    return OALGPU_OK;
}

int oalgpu_sync(oalgpu_context *c)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;           // (a resident voice kernel finishes what has been rung and ends)
    HIP_TRY(hipStreamSynchronize(c->stream));
    if(c->res.reduceStream) HIP_TRY(hipStreamSynchronize(c->res.reduceStream));
    if(c->postStream) HIP_TRY(hipStreamSynchronize(c->postStream));
    c->postPending = false;
    c->updatesKnownDone = c->updatesSubmitted;
    if(c->res.ready) { ResidentCollectTimes(c, true); if(int rc = ResidentCheckError(c)) return rc; }
    return OALGPU_OK;
}

int oalgpu_set_timing(oalgpu_context *c, int enable)
{
    if(c) { if(int rc = FlushPendingMix(c)) return rc; }
    // (a parameter block that waits for a resident update was applied BEFORE this call: it goes in first, as on the launched path)
    if(c && c->res.pendingBlock) { if(int rc = UseCtx(c)) return rc; }
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    c->timing = enable != 0;
    c->timed = false;
    return OALGPU_OK;
}

int oalgpu_last_update_ms(oalgpu_context *c, float *total_ms, float *voice_kernel_ms)
{
    if(!c || !c->timed) return Fail(OALGPU_ERR_INVALID, "no timed update (call oalgpu_set_timing first)");
    if(int rc = oalgpu_sync(c)) return rc;
    if(total_ms) HIP_TRY(hipEventElapsedTime(total_ms, c->evStart, c->evEnd));
    if(voice_kernel_ms) HIP_TRY(hipEventElapsedTime(voice_kernel_ms, c->evStart, c->evVoice));
    return OALGPU_OK;
}

#ifdef OALGPU_MEASUREMENT      // (liboalgpu_measure.so, `make measure`: tools/measure/oalgpu_measure.h)
/* Measurement aid, OALGPU_CTX_PROFILE contexts: which stages the voice kernel's measurement variant skips
 * (1 FIR, 2 resampler, 8 direct filter, 16 FIR input build); 0 = none. */
int oalgpu_debug_set_ablate(oalgpu_context *c, uint32_t mask)
{
    if(!c || !c->phaseTimes.p) return Fail(OALGPU_ERR_INVALID, "not an OALGPU_CTX_PROFILE context");
    c->prof.ablate = mask;
    return OALGPU_OK;
}

/* Measurement aid: copies the [voice][8] s_memtime stamps the voice kernel's measurement variant
 * recorded (contexts created with OALGPU_CTX_PROFILE). */
int oalgpu_debug_phase_times(oalgpu_context *c, unsigned long long *out)
{
    if(!c || !out || !c->phaseTimes.p) return Fail(OALGPU_ERR_INVALID, "phase times were not enabled");
    if(int rc = oalgpu_sync(c)) return rc;
    HIP_TRY(hipMemcpy(out, c->phaseTimes.p, size_t{c->L.numVoices} * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return OALGPU_OK;
}

/* Same aid: the [wavefront][4] stamps behind them (kernel entry, first voice requested and parked,
 * last voice done, partial bus stored; then pass 0 in detail: first control line in registers, first request issued,
 * workgroup through the table-staging barrier); `out` holds numVoices*8 words ([wavefront][8]), *waves receives the count. */
int oalgpu_debug_wave_times(oalgpu_context *c, unsigned long long *out, uint32_t *waves)
{
    if(!c || !out || !waves || !c->phaseTimes.p) return Fail(OALGPU_ERR_INVALID, "phase times were not enabled");
    if(int rc = oalgpu_sync(c)) return rc;
    HIP_TRY(hipMemcpy(out, c->phaseTimes.p + size_t{c->L.numVoices} * 8, size_t{c->L.numVoices} * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    *waves = WaveKernelGroups(c->L) * 4u;
    return OALGPU_OK;
}

#endif // OALGPU_MEASUREMENT

int oalgpu_slot_set_convolution(oalgpu_context *c, uint32_t slot, oalgpu_convolution *conv)
{
    if(!c || slot >= c->L.numSlots) return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_convolution: bad slot");
    if(conv && ConvOutLines(conv) > c->L.numDry)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_convolution: the effect mixes into more lines than the context has dry lines");
    if(int rc = oalgpu_sync(c)) return rc;
    c->slotConv[slot] = conv;
    return OALGPU_OK;
}

// An EAX reverb instance is ONE workgroup that needs a compute unit's LDS nearly to itself (128 KB: both pipelines' rows,
// csrc/reverb_kernels.hip), and it runs on the post stream beside the NEXT update's voice kernel.  A voice kernel whose grid fills
// the machine exactly (two 78 KB workgroups on every CU) then finds the instances' CUs taken: the workgroups it cannot place wait
// for a second round, and the launch lasts 1.6 times as long (BASELINE configs[3]: 110 -> 174 us in every other update, the
// reverbs 85 -> 170 us in the updates between, profiles/r5/evidence/step_timeline_config4.txt).  With reverbs attached the
// automatic voices-per-workgroup choice therefore leaves the instances their CUs: a few more voices per wavefront, so that the
// grid fits on the CUs that are left -- one round, and the reverbs run beside it undisturbed.
// voice_rows.hip's grid: one workgroup of eight wavefronts per compute unit -- the compute units the attached EAX reverb instances need
// (a whole CU's LDS each, beside the NEXT update's voice kernel, see below) left out --, the voices dealt evenly
static void SetRowsGroups(oalgpu_context *c)
{
    DeviceLayout &L = c->L;
    uint32_t reverbs = 0;
    for(oalgpu_reverb *r : c->slotReverb) reverbs += r ? 1u : 0u;
    uint32_t cus = 256u;
    hipDeviceProp_t prop{};
    if(hipGetDeviceProperties(&prop, c->desc.device) == hipSuccess && prop.multiProcessorCount > 0) cus = uint32_t(prop.multiProcessorCount);
    else (void)hipGetLastError();
    const uint32_t waves = RowsWavesPerGroup();
    uint32_t groups = std::min<uint32_t>(cus > reverbs ? cus - reverbs : 1u, (L.numVoices + waves - 1u) / waves);
    if(c->desc.voices_per_group) groups = (L.numVoices + c->desc.voices_per_group - 1u) / c->desc.voices_per_group;
    groups = std::max<uint32_t>(1u, std::min<uint32_t>(groups, c->groupsAllocated));
    L.rowsVpg = (L.numVoices + groups - 1u) / groups;
    L.numGroups = (L.numVoices + L.rowsVpg - 1u) / L.rowsVpg;
    L.numLineGroups = L.numGroups;
}

static int RebalanceWaveGroups(oalgpu_context *c)
{
    if(c->useWave && c->L.rows8)
    {
        const uint32_t before = c->L.rowsVpg;
        DeviceLayout T = c->L;
        SetRowsGroups(c);
        if(c->L.rowsVpg != before)
        {   // (the streams may still run launches of the old grid)
            const DeviceLayout N = c->L;
            c->L = T;
            if(int rc = oalgpu_sync(c)) return rc;
            c->L = N;
        }
        return OALGPU_OK;
    }
    if(!c->useWave || c->desc.voices_per_group != 0u || c->L.wave16) return OALGPU_OK;
    uint32_t reverbs = 0;
    for(oalgpu_reverb *r : c->slotReverb) reverbs += r ? 1u : 0u;
    hipDeviceProp_t prop{};
    if(hipGetDeviceProperties(&prop, c->desc.device) != hipSuccess || prop.multiProcessorCount <= 0) { (void)hipGetLastError(); return OALGPU_OK; }
    const uint32_t cus = uint32_t(prop.multiProcessorCount);
    const uint32_t slots = 2u * (cus > reverbs ? cus - reverbs : 1u);          // two voice workgroups per compute unit
    DeviceLayout &L = c->L;
    uint32_t vpw = std::max<uint32_t>(1u, (c->desc.max_voices + 2047u) / 2048u);  // (oalgpu_context_create's rule)
    const uint32_t full = 2u * cus;
    auto groupsOf = [&](uint32_t w) { DeviceLayout T = L; T.waveVoices = w; return WaveKernelGroups(T); };
    // only a grid that was meant to fill the machine in one round is thinned out (smaller scenes leave room anyway)
    if(reverbs && groupsOf(vpw) <= full) { while(groupsOf(vpw) > slots) ++vpw; }
    if(vpw == L.waveVoices) return OALGPU_OK;
    if(int rc = oalgpu_sync(c)) return rc;
    L.waveVoices = vpw;
    L.numGroups = std::max<uint32_t>(1u, WaveKernelGroups(L));       // (never more than the context's buffers were sized for)
    L.numLineGroups = L.numGroups;
    return OALGPU_OK;
}

int oalgpu_slot_set_effect(oalgpu_context *c, uint32_t slot, oalgpu_effect *fx)
{
    if(!c || slot >= c->L.numSlots) return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_effect: bad slot");
    // (a dedicated effect may address the real output lines, which follow the dry lines in the bus block)
    if(fx && (EffectOutLines(fx) > c->L.numDry + c->L.numReal || EffectInChannels(fx) > c->L.wetChannels))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_effect: the effect's lines do not fit the context's buses");
    if(int rc = oalgpu_sync(c)) return rc;
    c->slotEffect[slot] = fx;
    return OALGPU_OK;
}

int oalgpu_slot_set_reverb(oalgpu_context *c, uint32_t slot, oalgpu_reverb *rev)
{
    if(!c || slot >= c->L.numSlots) return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_reverb: bad slot");
    if(rev && c->L.wetChannels < 4)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_reverb: the reverb reads a 4-line B-Format wet bus");
    if(rev && ReverbOutLines(rev) > c->L.numDry)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_slot_set_reverb: the effect mixes into more lines than the context has dry lines");
    if(int rc = oalgpu_sync(c)) return rc;
    if(rev) { if(int rc = oalgpu_reverb_set_math_mode(rev, c->exact ? OALGPU_MATH_EXACT : OALGPU_MATH_FAST)) return rc; }
    c->slotReverb[slot] = rev;
    return RebalanceWaveGroups(c);
}

const char *oalgpu_voice_kernel_name(oalgpu_context *c)
{
    if(!c) return "";
    if(c->useWave) return WaveKernelName(c->L);
    return c->exact ? "VoiceMixKernel<true, LINES>" : "VoiceMixKernel<false, LINES>";
}

/* Multi-GPU: whether this context's voice kernel continues the carried HRTF accumulator tail
 * (exactly one rank must, the one that runs the post-process on the reduced buses). */
int oalgpu_resident_stats(oalgpu_context *c, oalgpu_resident_info *out)
{
    if(!c || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    auto &R = c->res;
    uint32_t w[kRcCount * 16] = {};
    if(R.ready && !R.running)
    {   // the wait counters: a copy on the null stream, with nothing resident on the device
        if(int rc = UseDevice(c->desc.device)) return rc;
        HIP_TRY(hipMemcpy(w, R.counters.p, sizeof(w), hipMemcpyDeviceToHost));
    }
    std::lock_guard<std::mutex> g(gResLock);
    out->enabled = R.enabled ? 1 : 0; out->failed = R.failed ? 1 : 0; out->running = R.running ? 1 : 0;
    out->door_in_device_memory = R.doorInBar ? 1 : 0;
    out->launches = R.launches; out->updates = R.next; out->parks = uint32_t(R.parks);
    out->timed_launches = uint32_t(R.kernelLaunches); out->timed_updates = R.kernelUpdates; out->timed_kernel_ms = R.kernelMs;
    out->max_updates_per_launch = R.maxUpdates; out->pad = 0;
    // (the voice kernel's words: every 128th workgroup keeps them)
    const double tick = 0.01, groups = double(std::max<uint32_t>((c->L.numGroups + 127u) / 128u, 1u));
    out->wait_door_us = w[16 * kRcWaitDoor] * tick / groups; out->wait_reduction_us = w[16 * kRcWaitRed] * tick / groups;
    out->wait_arrival_us = w[16 * kRcWaitArrive] * tick; out->wait_post_us = w[16 * kRcWaitPost] * tick;
    out->wait_reduced_us = w[16 * kRcWaitRedDone] * tick; out->wait_split_us = w[16 * kRcWaitSplit] * tick;
    out->install_us = w[16 * kRcInstall] * tick / groups; out->busy_us = w[16 * kRcBusy] * tick / groups;
    out->top_us = w[16 * kRcTop] * tick / groups;
    return OALGPU_OK;
}

int oalgpu_resident_set_short_run(oalgpu_context *c, uint32_t updates)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;
    c->res.shortRun = updates;
    c->res.shortRuns = 0; c->res.cooldown = 0;
    return OALGPU_OK;
}

int oalgpu_resident_set_timing(oalgpu_context *c, int enable)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;           // (the running launch ends: the next one carries the events, or no longer does)
    c->res.timeLaunches = enable != 0;
    return OALGPU_OK;
}

int oalgpu_resident_set_max_updates(oalgpu_context *c, uint32_t max_updates)
{
    if(!c || max_updates == 0 || max_updates > 0x10000000u) return Fail(OALGPU_ERR_INVALID, "oalgpu_resident_set_max_updates: 1 .. 2^28");
    if(int rc = UseCtx(c)) return rc;
    c->res.maxUpdates = max_updates;
    return OALGPU_OK;
}

int oalgpu_set_carry_accum(oalgpu_context *c, int enable)
{
    if(c) { if(int rc = FlushPendingMix(c)) return rc; }
    // (a parameter block that waits for a resident update was applied BEFORE this call: it goes in first, as on the launched path)
    if(c && c->res.pendingBlock) { if(int rc = UseCtx(c)) return rc; }
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    c->carryAccum = enable != 0;
    return OALGPU_OK;
}
