// C-ABI shim (include/oalgpu.h) over the HIP kernels.  Host logic only: argument checks,
// HBM allocation/upload, parameter preparation that the reference does with libm on its mixer
// thread (resampler state, biquad design), and kernel launches on the context's stream.
// No CPU fallback exists anywhere in this file: without a HIP device every entry point fails.
#include "../../include/oalgpu.h"
#ifdef OALGPU_MEASUREMENT
#include "../../tools/measure/oalgpu_measure.h"
#endif

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>          // types and enums only: the library itself is resolved with dlopen/dlsym
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../host/mhr.hpp"
#include "../host/hrtf_build.hpp"
#include "../host/params.hpp"
#include "../host/tables.hpp"
#include "api_util.hpp"
#include "kernels.hpp"
#include "reverb_dev.hpp"

using namespace oalgpu;

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

namespace {

// One blob with every resampler table: [bsinc12 | bsinc24 | bsinc48 | spline | gaussian]
struct TableBlob {
    std::vector<float> data;
    uint32_t bsincBase[3]{};
    uint32_t cubicBase[2]{};
    TableBlob()
    {
        const int fam[3] = {12, 24, 48};
        for(int i = 0; i < 3; ++i)
        {
            const BsincTable *t = GetBsincTable(fam[i]);
            bsincBase[i] = uint32_t(data.size());
            data.insert(data.end(), t->tab.begin(), t->tab.end());
        }
        for(int i = 0; i < 2; ++i)
        {
            const CubicTable *t = GetCubicTable(i);
            cubicBase[i] = uint32_t(data.size());
            data.insert(data.end(), &t->phase[0][0], &t->phase[0][0] + 256);
        }
    }
    uint32_t filterBase(const oalgpu_interp_state &st) const
    {
        switch(st.kind)
        {
        case 2: return cubicBase[st.table ? 1 : 0];
        case 3: case 4: return bsincBase[st.table == 12 ? 0 : st.table == 24 ? 1 : 2] + st.filter_offset;
        default: return 0;
        }
    }
};
const TableBlob &Blob() { static const TableBlob b; return b; }

} // namespace

struct oalgpu_context {
    oalgpu_context_desc desc{};
    bool exact{true};
    hipStream_t stream{nullptr};
    bool ownStream{true};
    hipEvent_t evStart{nullptr}, evVoice{nullptr}, evEnd{nullptr};
    // oalgpu_mix_update pipelines two streams when the context owns them: the voice kernel of
    // update k+1 (main stream) overlaps the bus reduction and the post-process of update k
    // (post stream).  The per-workgroup partial buses are double-buffered for that.
    hipStream_t postStream{nullptr};
    hipEvent_t evVoiceDone[2]{nullptr, nullptr}, evReduceDone[2]{nullptr, nullptr}, evPostDone{nullptr};
    uint32_t parity{0};
    bool postPending{false};
    // A pipelined oalgpu_mix_update is SUBMITTED one library call late: if that next call is oalgpu_param_block_apply, the block's
    // records are installed by the update's own voice kernel -- every wavefront applies the records of the voices it has just
    // mixed, in its epilogue -- and neither ApplyParamsKernel nor its two dispatch gaps stand between two voice kernels.
    struct { bool active{false}; uint32_t samples{0}; int post{0}; } pendingMix;
    // the pipelined host boundary (oalgpu_voice_move_async / oalgpu_read_output_async): pinned ring slots
    static constexpr uint32_t kIoSlots = 4;
    oalgpu_voice_move *panHost[kIoSlots]{};
    size_t panCap{0};
    hipEvent_t panApplied[kIoSlots]{};
    uint32_t panNext{0};
    float *outHost[kIoSlots]{};
    hipEvent_t outDone[kIoSlots]{};
    uint32_t outNext{0};
    size_t outFloats{0};
    // Where the box lets the host store into device memory (large BAR), the move slots ARE device memory: the installing kernel
    // reads its records out of HBM instead of over PCIe (3 us less in front of the voice kernel, tools/ubench_largebar.hip).
    bool panInBar{false};
    // Once oalgpu_read_output_async has been used on an HRTF context, the post-process kernel stores the two output lines into
    // the next ring slot itself and raises the slot's sequence number (pinned, 64 bytes apart) behind them: reading the output
    // back costs the host no runtime call.  outRingWritten: the update submitted last did so, for slot outNext % kIoSlots.
    // oalgpu_voice_events_async: what changed about the voices since the last report, into pinned ring slots
    static constexpr uint32_t kEvCap = 1024;
    uint32_t *evHost[kIoSlots]{};
    hipEvent_t evDone[kIoSlots]{};
    uint32_t evNext{0};
    DevBuf<uint32_t> evSnapshot, evCounters;
    bool outRing{false}, outRingWritten{false};
    bool outViaRing[kIoSlots]{};
    uint32_t outSeq{0}, outSlotSeq[kIoSlots]{};    // every launch that writes a slot raises ITS number
    uint32_t outArrivedTotal{0};                   // what outArrived (the FIR workgroups of every slot-writing launch: it only grows) reads by now
    // What the host already knows to be finished saves it runtime calls: an output that has been waited for proves its update's
    // whole chain done (moves installed, voices mixed, reduced, post-processed), so the checks in front of a slot's or a
    // partial-bus buffer's reuse need not ask the runtime.  Updates are numbered from 1 as they are submitted.
    // oalgpu_mix_update of a pipelined HRTF context without effect slots and without a collective: reduction and post-process
    // are ONE launch (LaunchReducePostFused).  fuseReduce: this update's reduction was held back for it (oalgpu_mix_voices_overlapped
    // -> oalgpu_post_process_overlapped); reducedEpoch: what the launch's counter of reduction workgroups reads when they are through.
    bool fuseReduce{false}, reduceHeld{false};
    DeviceLayout heldL{};
    uint32_t heldParity{0};
    DevBuf<uint32_t> reducedCount;
    uint32_t reducedEpoch{0};
    hipEvent_t lastPostEvent{nullptr};      // what JoinPost waits for: evPostDone, or the fused launch's own event
    uint64_t updatesSubmitted{0}, updatesKnownDone{0};
    uint64_t reduceUpdate[2]{0, 0}, panUpdate[kIoSlots]{}, outUpdate[kIoSlots]{};
    uint32_t *outFlags{nullptr};
    DevBuf<uint32_t> outArrived;
    float *partHrtfBuf[2]{nullptr, nullptr};
    float *partLinesBuf[2]{nullptr, nullptr};
    bool timing{false}, timed{false};
    DeviceLayout L{};
    HrtfStoreDev hrtfDev{};
    HrtfData hrtfHost;
    bool hrtfLoaded{false};
    uint32_t hrtfGeneration{0};            // bumped by every oalgpu_hrtf_load_mhr: parameter blocks carry HRIR indices of ONE store
    bool carryAccum{true};
    bool useWave{false};                   // FAST contexts without sends (HRTF, or <= 8 dry lines): voice_wave.hip
    uint32_t groupsAllocated{0};           // workgroups the partial-bus buffers were sized for (oalgpu_context_create)
    std::vector<oalgpu_convolution*> slotConv;   // per effect slot: attached convolution reverb (not owned)
    std::vector<oalgpu_reverb*> slotReverb;      // per effect slot: attached EAX reverb (not owned)
    std::vector<oalgpu_effect*> slotEffect;      // per effect slot: equalizer / modulator / echo / dedicated (not owned)
    DevBuf<uint32_t> reverbTicket;               // mix-out order word of a reverb batch launch

    DevBuf<float> tables;
    DevBuf<BufferItem> buffers;
    DevBuf<uint32_t> startDelay;           // [voice] samples until a delayed voice starts
    DevBuf<uint32_t> queueDone;            // [voice] buffers a streaming voice has played through
    std::vector<void*> bufferData;
    std::vector<uint32_t> bufferLoopLen;   // loop_end - loop_start of every registered buffer (0: cannot loop)
    uint32_t numBuffers{0};                // handles handed out so far (released ones are reused: freeBuffers)
    // The lifetime of a buffer handle (oalgpu_buffer_release): a handle is freed -- its HBM copy, and the handle for reuse -- when
    // the host has released it AND nothing refers to it any more: no voice slot that was initialised on it (voiceHead: cleared when
    // the slot is initialised again or set to Stopped), no live buffer whose queue link points at it, no channel view of it.
    struct BufHost { bool live{false}, released{false}; int32_t parent{-1}, next{-1}; uint32_t refs{0}; };
    std::vector<BufHost> bufHost;
    std::vector<uint32_t> freeBuffers;
    std::vector<int32_t> voiceHead;        // [voice] the buffer the slot was initialised on (a queue: its first), -1: none
    std::vector<uint32_t> queueDoneKnown, queueUnqueued;   // [voice] AsyncBufferComplete counts the host has read back / given up (oalgpu_voice_queue_unqueue)
    DevBuf<VoiceCtl> ctl;
    DevBuf<float> prev, hrtfOld, hrtfTgt, hist, gainCur, gainTgt, sendCur, sendTgt;
    DevBuf<BiquadSlot> dfilt, sfilt;
    DevBuf<float> partLines, partLines2, partHrtf, partHrtf2, bus, streams;
    DevBuf<uint32_t> lineGains;
    DevBuf<AmbiScaleState> ambi;
    DevBuf<NfcState> nfc;
    NfcDesign nfcDevice{};                   // DeviceBase::mNFCtrlFilter (after init(w1))
    DevBuf<unsigned long long> phaseTimes;  // OALGPU_CTX_PROFILE: the measurement variant's stamps
    WaveProf prof{nullptr, 0u};
    const WaveProf *profArg() const { return prof.times ? &prof : nullptr; }
    DevBuf<AmbiMapEntry> dryMap, wetMaps;   // MixParams::AmbiMap of the dry bus / of every slot's wet bus
    DevBuf<PanRecord> panRecs;
    std::vector<VoiceCtl> ctlHost;          // oalgpu_voices_readback: staging
    std::vector<uint32_t> doneHost;
    DevBuf<TargetRecord> tgtRecs;           // oalgpu_voice_set_hrtf_targets: staging
    DevBuf<float> tgtCoeffs;
    bool serialOnly{false};                // OALGPU_CTX_SERIAL: no two-stream pipeline
    // multi-GPU (oalgpu_comm_init / oalgpu_comm_init_host): how this rank's bus block gets summed into rank 0's,
    // right behind the partial-bus reduction, on the stream that runs it
    struct BusTransport *comm{nullptr};
    int commRank{0}, commWorld{1};
    // the stage behind the buses (output_kernels.hip): AmbiDecPostProcess of non-HRTF contexts, dither, PCM
    bool decOn{false}, decDual{false};
    uint32_t decOut{0};
    DevBuf<float> decGainsHf, decGainsLf, decBands;
    DevBuf<SplitterState> decSplit;
    int outType{6};                        // DevFmtType order: 0 i8, 1 u8, 2 i16, 3 u16, 4 i32, 5 u32, 6 f32
    float ditherDepth{0.0f};
    uint32_t ditherSeed{22222};
    DevBuf<unsigned char> pcm;
    // HRTF store
    DevBuf<float> hFieldDist, hCoeffs;
    DevBuf<uint8_t> hEvCount, hDelays;
    DevBuf<uint16_t> hAzCount, hIrOffset;
    // DirectHrtfState
    DevBuf<SplitterState> dSplit, dSplit2;  // the post-process's splitter states; the fused FAST post-process reads one and files the other
    uint32_t dSplitCur{0};                  // which of the two holds the current states
    DevBuf<float> carryBuf;                 // HrtfAccumData as the fused post-process leaves it (1152 x 2): the next reduction's carry
    float dSplitCoeff{0.0f};                // the splitters' coefficient (one crossover for all channels) ...
    float runPower[4]{1.0f, 0.0f, 1.0f, 1.0f};  // ... and their transition over a run of runPowerSeg samples (SplitterRunPowers)
    uint32_t runPowerSeg{0};
    DevBuf<uint32_t> postArrived;           // the fused post-process's channel counter (only ever grows) ...
    uint32_t postEpoch{0};                  // ... and the value it has reached after the last launch
    const ParamRecord *nextRecs{nullptr};   // the block the voice kernel being launched installs in its epilogue (RunMixUpdate)
    const int32_t *nextMap{nullptr};
    const float *nextRows{nullptr};
    bool carryInBuf{false};                 // the carried accumulator is in carryBuf (else: in the bus block's accumulator region)
    DevBuf<float> dHfScale, dCoeffs, dTemp;
    uint32_t dIrSize{0};
    bool directSet{false};
    // staging
    DevBuf<ParamRecord> paramDev;
    std::vector<ParamRecord> paramHost;
    DevBuf<VoiceInitRecord> initDev;
    std::vector<VoiceInitRecord> initPending;
    // callback sources (oalgpu_voice_init_callback): the host's mirror of what Voice::mix keeps for them
    struct CbVoice {
        uint32_t voice{0}; int32_t buffer{-1}; uint32_t frameBytes{4}, capacityFrames{0};
        oalgpu_callback_fn fn{nullptr}; void *user{nullptr};
        std::vector<char> data;                    // BufferStorage::mData of the callback buffer: numBlocks blocks valid
        uint32_t numBlocks{0}, blockOffset{0};     // Voice::mNumCallbackBlocks / mCallbackBlockOffset (samples per block = 1)
        bool stopped{false};                       // VoiceFlag::CallbackStopped
        int32_t position{0}; uint32_t frac{0}, step{0};        // mPosition / mPositionFrac / mStep
        int state{OALGPU_VOICE_PLAYING}; bool hasBuffer{true}; // mPlayState / mCurrentBuffer != nullptr
        char *pinned[2]{nullptr, nullptr}; hipEvent_t copied[2]{nullptr, nullptr}; uint32_t slot{0};
        size_t allocBytes{0};                      // of the device buffer and each pinned one
        bool retired{false};                       // its voice slot became another source: the entry may be reused
    };
    std::vector<CbVoice> cbVoices;
    std::vector<int32_t> cbOfVoice;                // [voice] index into cbVoices, -1 = not a callback source

    // ---- the resident voice kernel (OALGPU_CTX_RESIDENT; protocol and device side: kernels.hpp ResidentDoor, voice_wave.hip) ----
    // One launch of the HRTF voice kernel stays on the machine while the host only calls oalgpu_param_block_apply,
    // oalgpu_mix_update, oalgpu_read_output_async and oalgpu_output_wait.  Per update the host writes a doorbell slot and
    // launches the update's reduction (reduce stream) and post-process (post stream), which wait for device counters.  Any
    // other entry point parks the kernel first (UseDevice): it finishes what has been rung and ends, and whatever the entry
    // point puts on the main stream runs behind it in stream order.
    struct ResidentState {
        bool enabled{false};                       // the context was created with OALGPU_CTX_RESIDENT and its layout has a resident kernel
        bool ready{false}, failed{false};          // buffers and streams exist; the mode gave up (the context then launches per update)
        bool running{false};                       // a launch is on the main stream that has not been told to leave
        std::mutex lock;                           // submit / park (another context's entry point parks this one's kernel)
        hipStream_t reduceStream{nullptr};
        ResidentDoor *door{nullptr};               // the host's view (the device reads the same address)
        bool doorInBar{false};
        DevBuf<uint32_t> counters;                 // [kRcCount][16]
        uint32_t *hostFlags{nullptr};              // pinned [kRhCount][16]
        DevBuf<float> part;                        // kResidentSets sets of partial buses
        size_t setFloats{0};
        uint32_t next{0};                          // the next update's index (counts this context's resident updates)
        uint32_t endSeq{0};                        // where the running launch ends by itself
        uint32_t launches{0}, startedTotal{0};
        uint32_t launchBase{0};                    // the running launch's first update
        // A launch pays for itself over a few dozen updates (its first updates run at the launched path's pace, and the block
        // ends with the pipeline's drain: 45.5 against 44.5 us per update for blocks of 20, 38.8 against 42.9 for blocks of 50,
        // tools/resident_block_cost.py).  A host that keeps it short -- a synchronisation every 20 updates, parameters set the
        // launched way before every update -- is better off with a launch per update: after three launches in a row that covered
        // fewer than 32 updates the context launches per update for a while, then tries again.
        uint32_t shortRuns{0}, cooldown{0};
        uint32_t shortRun{32};                     // launches that cover fewer updates count as short (0: never fall back)
        uint32_t awaitStarted{0};                  // the launch id whose "every workgroup has started" word the host has yet to see
        uint32_t maxUpdates{4096};
        uint32_t setUses[kResidentSets]{};         // updates that went into each partial set so far
        uint32_t posts{0};                         // post-processes launched in this mode
        uint32_t firGroups{0}, redGroups{0}, groupsPerCu{0};
        oalgpu_param_block *pendingBlock{nullptr}; // oalgpu_param_block_apply: rides in the next update's doorbell slot
        hipEvent_t copyPending{nullptr};           // a copy out of the bus block queued on the post stream: the next reduction waits for it
        // the launches' own times (events bound to the dispatch), collected when the launch is known to have ended
        static constexpr uint32_t kEv = 4;
        hipEvent_t evStart[kEv]{}, evStop[kEv]{};
        uint32_t evFirst[kEv]{}, evLast[kEv]{};    // the updates the launch of that event pair covered: [first, last)
        bool evOpen[kEv]{};
        bool timeLaunches{false};                  // oalgpu_set_timing: the launches carry their events
        double kernelMs{0.0};
        uint64_t kernelUpdates{0}, kernelLaunches{0}, parks{0};
    } res;

    ~oalgpu_context()
    {
        for(void *p : bufferData) if(p) (void)hipFree(p);
        for(uint32_t k = 0; k < kIoSlots; ++k)
        {
            if(panHost[k]) (void)(panInBar ? hipFree(panHost[k]) : hipHostFree(panHost[k]));
            if(outHost[k]) (void)hipHostFree(outHost[k]);
            if(evHost[k]) (void)hipHostFree(evHost[k]);
            if(evDone[k]) (void)hipEventDestroy(evDone[k]);
            for(hipEvent_t e : {panApplied[k], outDone[k]}) if(e) (void)hipEventDestroy(e);
        }
        if(outFlags) (void)hipHostFree(outFlags);
        if(res.door) (void)(res.doorInBar ? hipFree(res.door) : hipHostFree(res.door));
        if(res.hostFlags) (void)hipHostFree(res.hostFlags);
        for(uint32_t k = 0; k < ResidentState::kEv; ++k)
            for(hipEvent_t e : {res.evStart[k], res.evStop[k]}) if(e) (void)hipEventDestroy(e);
        if(res.reduceStream) (void)hipStreamDestroy(res.reduceStream);
        if(evStart) (void)hipEventDestroy(evStart);
        if(evVoice) (void)hipEventDestroy(evVoice);
        if(evEnd) (void)hipEventDestroy(evEnd);
        for(hipEvent_t e : {evVoiceDone[0], evVoiceDone[1], evReduceDone[0], evReduceDone[1], evPostDone})
            if(e) (void)hipEventDestroy(e);
        if(postStream) (void)hipStreamDestroy(postStream);
        if(stream && ownStream) (void)hipStreamDestroy(stream);
    }
};

struct oalgpu_param_block {
    DevBuf<ParamRecord> recs;
    uint32_t count{0};
    int device{0};
    uint32_t hrtfGeneration{0};                             // of the store the records' HRIR indices and weights were taken from
    DevBuf<int32_t> voiceToRec;                             // [voice of the context] -> index of its record in the block, or -1: how a
                                                            // voice kernel's wavefront finds the records of the voices it mixed
    uint32_t mapVoices{0};
    DevBuf<float> rows;                                     // [record][irStride][2]: the records' blended target HRIRs (resident contexts)
    std::vector<std::pair<uint32_t, uint32_t>> cbSteps;     // (voice, mStep) of the callback voices in the block
    oalgpu_context *heldBy{nullptr};                        // a resident context that keeps the block for its next update (res.pendingBlock)
};

static int FlushPendingMix(oalgpu_context *c, struct oalgpu_param_block *next = nullptr);
static bool ResidentWanted(const oalgpu_context *c, int post_process);
static int FlushResidentBlock(oalgpu_context *c);
// every entry point that enqueues work on a context or reads its state goes through here: the device (which parks a resident
// voice kernel: whatever follows on the main stream then runs behind its end), a parameter block that was waiting for a
// resident update, and the deferred update
static int UseCtx(oalgpu_context *c)
{
    if(int rc = UseDevice(c->desc.device)) return rc;
    if(int rc = FlushResidentBlock(c)) return rc;
    return FlushPendingMix(c);
}
// the entry points a resident voice kernel stays through (they touch neither the main stream nor anything that synchronises the device)
static int UseCtxResident(oalgpu_context *c)
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

namespace {

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

// ---- RCCL, resolved at run time: a single-GPU host never needs the library, and a process that already
// carries an RCCL (torch's) must use THAT instance rather than a second copy
struct RcclApi {
    ncclResult_t (*getUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*commInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*commDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*getErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*commCount)(const ncclComm_t, int*) = nullptr;
    bool ok = false;
    std::string why;
};

RcclApi &Rccl()
{
    static RcclApi api = []
    {
        RcclApi a;
        void *h = nullptr;
        if(dlsym(RTLD_DEFAULT, "ncclCommInitRank")) h = RTLD_DEFAULT;       // already in the process
        else
        {
            for(const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
                if((h = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        }
        if(!h)
        {   // dlerror() clears the pending message: read it once
            const char *e = dlerror();
            a.why = std::string("librccl.so not found: ") + (e ? e : "");
            return a;
        }
        a.getUniqueId = reinterpret_cast<decltype(a.getUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        a.commInitRank = reinterpret_cast<decltype(a.commInitRank)>(dlsym(h, "ncclCommInitRank"));
        a.commDestroy = reinterpret_cast<decltype(a.commDestroy)>(dlsym(h, "ncclCommDestroy"));
        a.reduce = reinterpret_cast<decltype(a.reduce)>(dlsym(h, "ncclReduce"));
        a.getErrorString = reinterpret_cast<decltype(a.getErrorString)>(dlsym(h, "ncclGetErrorString"));
        a.commCount = reinterpret_cast<decltype(a.commCount)>(dlsym(h, "ncclCommCount"));
        a.ok = a.getUniqueId && a.commInitRank && a.commDestroy && a.reduce;
        if(!a.ok) a.why = "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclReduce";
        return a;
    }();
    return api;
}

int FailRccl(const char *what, ncclResult_t r)
{
    const RcclApi &a = Rccl();
    return Fail(OALGPU_ERR_HIP, std::string(what) + ": " + (a.getErrorString ? a.getErrorString(r) : "RCCL error"));
}

} // namespace

// The one exchange of a sharded update: the bus block [dry + real lines | wet buses |
// HrtfAccumData] of every rank is summed into rank 0's, in place, on the stream that just produced it.  Two
// transports behind one interface: RCCL (ncclReduce over xGMI, one process per GPU) and a host-staged one
// (every rank's block through pinned memory into a shared-memory ring, summed by rank 0's stream in rank order)
// for ranks that RCCL cannot serve -- several processes on ONE GPU, which is how the N > 1 code of this library
// is exercised on a one-GPU box (tests/test_multi_rank.py).
struct BusTransport {
    virtual ~BusTransport() = default;
    virtual int reduceToRoot(oalgpu_context *c, hipStream_t s) = 0;
    virtual int ranks() const = 0;              // ranks the transport itself counts (RCCL: ncclCommCount)
    virtual const char *kind() const = 0;
};

namespace {

struct RcclTransport final : BusTransport {
    ncclComm_t comm{nullptr};
    ~RcclTransport() override { if(comm) (void)Rccl().commDestroy(comm); }
    int reduceToRoot(oalgpu_context *c, hipStream_t s) override
    {
        const ncclResult_t r = Rccl().reduce(c->L.bus, c->L.bus, BusFloats(c->L), ncclFloat32, ncclSum, 0, comm, s);
        if(r != ncclSuccess) return FailRccl("ncclReduce", r);
        return OALGPU_OK;
    }
    int ranks() const override
    {
        int n = 0;
        return (Rccl().commCount && Rccl().commCount(comm, &n) == ncclSuccess) ? n : -1;
    }
    const char *kind() const override { return "rccl"; }
};

__global__ void AddBusKernel(float *__restrict__ bus, const float *__restrict__ add, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) bus[i] = bus[i] + add[i];
}

// Shared-memory ring: kSlots updates deep, so that ranks may run that far ahead of rank 0 (the pipelined update
// never synchronises with the host).  produced[r] = updates of rank r whose block is in the ring; consumed =
// updates rank 0 has summed.  Host functions in stream order (hipLaunchHostFunc) move the data; they only touch
// host memory.
struct HostTransport final : BusTransport {
    static constexpr uint32_t kSlots = 4, kMaxWorld = 16;
    struct Header {
        std::atomic<uint32_t> magic;
        uint32_t world, floats;
        std::atomic<uint64_t> produced[kMaxWorld];
        std::atomic<uint64_t> consumed;
        std::atomic<uint32_t> failed;
        std::atomic<uint64_t> hello[kMaxWorld], ack[kMaxWorld];     // the attach handshake (oalgpu_comm_init_host)
    };
    std::string name;
    int fd{-1}, rank{0}, world{1};
    size_t bytes{0}, floats{0};
    Header *hdr{nullptr};
    float *ring{nullptr};                          // [rank][slot][floats]
    float *pinned[kSlots]{};                       // this rank's staging: D2H target (rank > 0), H2D source (rank 0)
    DevBuf<float> devSum;                          // rank 0: the other ranks' sum on the device
    // Host functions of one stream run in stream order, so each side counts the updates it has EXECUTED itself: a
    // sequence number handed over through a reusable host slot would be overwritten by a host that is kSlots or
    // more updates ahead of its stream (nothing throttles it: oalgpu_mix_update never synchronises)
    uint64_t executed{0};
    uint64_t submitted{0};                         // updates enqueued by the host (selects the staging slot)

    float *slot(int r, uint64_t q) const { return ring + (size_t(r) * kSlots + size_t(q % kSlots)) * floats; }
    int ranks() const override { return hdr ? int(hdr->world) : world; }
    const char *kind() const override { return "host"; }

    static bool WaitFor(const std::function<bool()> &ok, std::atomic<uint32_t> &failed)
    {
        const auto t0 = std::chrono::steady_clock::now();
        for(uint32_t spins = 0; !ok(); ++spins)
        {
            if(failed.load(std::memory_order_relaxed)) return false;
            if(spins > 64) std::this_thread::sleep_for(std::chrono::microseconds(20));
            if((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
            { failed.store(1u); return false; }
        }
        return true;
    }
    static void Produce(void *p)
    {   // rank > 0: this update's block (already in pinned memory) into the ring
        HostTransport *t = static_cast<HostTransport*>(p);
        const uint64_t q = t->executed++;
        if(!WaitFor([&] { return q < t->hdr->consumed.load(std::memory_order_acquire) + kSlots; }, t->hdr->failed)) return;
        std::memcpy(t->slot(t->rank, q), t->pinned[q % kSlots], t->floats * sizeof(float));
        t->hdr->produced[t->rank].store(q + 1, std::memory_order_release);
    }
    static void Gather(void *p)
    {   // rank 0: the other ranks' blocks of this update, summed in rank order
        HostTransport *t = static_cast<HostTransport*>(p);
        const uint64_t q = t->executed++;
        float *dst = t->pinned[q % kSlots];
        for(int r = 1; r < t->world; ++r)
        {
            if(!WaitFor([&] { return t->hdr->produced[r].load(std::memory_order_acquire) > q; }, t->hdr->failed))
            { std::memset(dst, 0, t->floats * sizeof(float)); return; }
            const float *src = t->slot(r, q);
            if(r == 1) std::memcpy(dst, src, t->floats * sizeof(float));
            else for(size_t i = 0; i < t->floats; ++i) dst[i] += src[i];
        }
        t->hdr->consumed.store(q + 1, std::memory_order_release);
    }
    int reduceToRoot(oalgpu_context *c, hipStream_t s) override
    {
        if(hdr->failed.load()) return Fail(OALGPU_ERR_HIP, "host transport: a rank timed out waiting for its peers");
        // (the pinned slot of update q is next written by the copy of update q + kSlots, which the stream runs behind
        // Produce / the H2D copy of update q: stream order alone keeps the staging slots apart)
        const uint64_t q = submitted++;
        if(rank != 0)
        {
            HIP_TRY(hipMemcpyAsync(pinned[q % kSlots], c->L.bus, bytes, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipLaunchHostFunc(s, Produce, this));
            return OALGPU_OK;
        }
        if(world == 1) return OALGPU_OK;
        HIP_TRY(hipLaunchHostFunc(s, Gather, this));
        HIP_TRY(hipMemcpyAsync(devSum.p, pinned[q % kSlots], bytes, hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(AddBusKernel, dim3(uint32_t((floats + 255) / 256)), dim3(256), 0, s, c->L.bus, devSum.p, uint32_t(floats));
        HIP_TRY(hipGetLastError());
        return OALGPU_OK;
    }
    ~HostTransport() override
    {
        for(float *p : pinned) if(p) (void)hipHostFree(p);
        if(hdr) munmap(hdr, sizeof(Header) + size_t(world) * kSlots * bytes);
        if(fd >= 0) close(fd);
        if(rank == 0 && !name.empty()) shm_unlink(name.c_str());
    }
};

int CommReduceBus(oalgpu_context *c, hipStream_t s)
{
    if(!c->comm) return OALGPU_OK;
    return c->comm->reduceToRoot(c, s);
}

} // namespace

extern "C" {

/* ---- multi-GPU: voices shard over the GPUs of a node, one context per GPU and process ----------------
 * Rank 0 calls oalgpu_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the
 * host has (a file, MPI, torch.distributed); every rank then calls oalgpu_comm_init on its context.
 * From then on oalgpu_mix_update / oalgpu_mix_voices sum-reduce the bus block to rank 0 (ncclReduce over
 * xGMI, issued by the library on the stream that produced the buses -- the context's post stream in the
 * pipelined path, so it runs beside the next update's voice kernel), and only rank 0 -- the one rank
 * that carries the HRTF accumulator tail -- runs the effect slots and the post-process. */
int oalgpu_comm_unique_id(void *out, size_t size)
{
    if(!out || size < sizeof(ncclUniqueId)) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_unique_id: 128 bytes needed");
    RcclApi &a = Rccl();
    if(!a.ok) return Fail(OALGPU_ERR_NO_DEVICE, a.why);
    ncclUniqueId id;
    const ncclResult_t r = a.getUniqueId(&id);
    if(r != ncclSuccess) return FailRccl("ncclGetUniqueId", r);
    std::memcpy(out, &id, sizeof(id));
    return OALGPU_OK;
}

int oalgpu_comm_init(oalgpu_context *c, const void *unique_id, size_t size, int rank, int world)
{
    if(!c || !unique_id || size < sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init: bad arguments");
    if(c->comm) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init: the context already has a communicator");
    if(!c->cbVoices.empty()) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init: not on a context with callback sources");
    RcclApi &a = Rccl();
    if(!a.ok) return Fail(OALGPU_ERR_NO_DEVICE, a.why);
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    auto t = std::make_unique<RcclTransport>();
    const ncclResult_t r = a.commInitRank(&t->comm, world, id, rank);
    if(r != ncclSuccess) return FailRccl("ncclCommInitRank", r);
    c->comm = t.release(); c->commRank = rank; c->commWorld = world;
    c->carryAccum = rank == 0;          // exactly one rank continues the carried HRTF accumulator
    return OALGPU_OK;
}

/* The same sharded update over the host-staged transport: `name` = a POSIX shared-memory object name ("/..."),
 * the same on every rank; rank 0 creates it, the others attach (they wait for it to appear).  For ranks that RCCL
 * cannot connect -- several processes on one GPU. */
int oalgpu_comm_init_host(oalgpu_context *c, const char *name, int rank, int world)
{
    if(!c || !name || name[0] != '/' || world < 1 || world > int(HostTransport::kMaxWorld) || rank < 0 || rank >= world)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: bad arguments");
    if(c->comm) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: the context already has a communicator");
    if(!c->cbVoices.empty()) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: not on a context with callback sources");
    if(int rc = UseCtx(c)) return rc;
    if(int rc = oalgpu_sync(c)) return rc;
    auto t = std::make_unique<HostTransport>();
    t->name = name; t->rank = rank; t->world = world;
    t->floats = BusFloats(c->L); t->bytes = t->floats * sizeof(float);
    const size_t total = sizeof(HostTransport::Header) + size_t(world) * HostTransport::kSlots * t->bytes;
    using clk = std::chrono::steady_clock;
    const auto deadline = clk::now() + std::chrono::seconds(60);
    // everything that can fail on this side comes first: a rank never announces itself and then falls over an allocation
    for(float *&p : t->pinned) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p), t->bytes, hipHostMallocDefault));
    if(rank == 0)
    {
        HIP_TRY(t->devSum.alloc(t->floats));
        shm_unlink(name);
        t->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if(t->fd < 0 || ftruncate(t->fd, off_t(total)) != 0) return Fail(OALGPU_ERR_HIP, std::string("shm_open/ftruncate ") + name + " failed");
        void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
        if(m == MAP_FAILED) return Fail(OALGPU_ERR_HIP, "mmap of the shared segment failed");
        t->hdr = static_cast<HostTransport::Header*>(m);
        t->ring = reinterpret_cast<float*>(static_cast<char*>(m) + sizeof(HostTransport::Header));
        // (a fresh segment is zero-filled: produced, consumed, failed, hello, ack start at 0)
        t->hdr->world = uint32_t(world); t->hdr->floats = uint32_t(t->floats);
        t->hdr->magic.store(0x0a16b05u, std::memory_order_release);
        // The attach is a handshake, so that no rank can sit on a segment a crashed earlier run left under the same
        // name (rank 0 only unlinks in its destructor): every other rank writes a token of its own into hello[r] and
        // trusts the segment only once THIS rank 0 has echoed it into ack[r]; a stale segment never answers.
        for(int r = 1; r < world; ++r)
        {
            uint64_t tok = 0;
            while((tok = t->hdr->hello[r].load(std::memory_order_acquire)) == 0)
            {
                if(clk::now() > deadline) return Fail(OALGPU_ERR_HIP, "oalgpu_comm_init_host: a rank did not attach within 60 s");
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            t->hdr->ack[r].store(tok, std::memory_order_release);
        }
    }
    else
    {
        const uint64_t token = ((uint64_t(getpid()) << 32) ^ uint64_t(clk::now().time_since_epoch().count()) ^ (uint64_t(rank) << 56)) | 1ull;
        bool attached = false;
        std::string why = "rank 0's segment did not appear";
        while(!attached && clk::now() < deadline)
        {
            t->fd = shm_open(name, O_RDWR, 0600);
            struct stat st{};
            if(t->fd >= 0 && (fstat(t->fd, &st) != 0 || size_t(st.st_size) < total)) { close(t->fd); t->fd = -1; }
            if(t->fd < 0) { std::this_thread::sleep_for(std::chrono::milliseconds(10)); continue; }
            void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, t->fd, 0);
            if(m == MAP_FAILED) return Fail(OALGPU_ERR_HIP, "mmap of the shared segment failed");
            auto *hdr = static_cast<HostTransport::Header*>(m);
            const auto patience = clk::now() + std::chrono::seconds(3);     // a live rank 0 answers within milliseconds
            bool said = false;
            while(clk::now() < patience && clk::now() < deadline)
            {
                if(hdr->magic.load(std::memory_order_acquire) == 0x0a16b05u)
                {
                    if(hdr->world != uint32_t(world) || hdr->floats != uint32_t(t->floats)) { why = "the ranks' contexts differ (world size or bus block)"; break; }
                    if(!said) { hdr->hello[rank].store(token, std::memory_order_release); said = true; }
                    if(hdr->ack[rank].load(std::memory_order_acquire) == token) { attached = true; break; }
                }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if(attached)
            {
                t->hdr = hdr;
                t->ring = reinterpret_cast<float*>(static_cast<char*>(m) + sizeof(HostTransport::Header));
            }
            else
            {   // nobody answered: a segment left behind by an earlier run (rank 0 replaces it), or a mismatch
                munmap(m, total); close(t->fd); t->fd = -1;
                if(why.find("differ") != std::string::npos) return Fail(OALGPU_ERR_INVALID, "oalgpu_comm_init_host: " + why);
                why = "no live rank 0 answered on the segment";
            }
        }
        if(!attached) return Fail(OALGPU_ERR_HIP, std::string("shm_open ") + name + ": " + why);
    }
    c->comm = t.release(); c->commRank = rank; c->commWorld = world;
    c->carryAccum = rank == 0;
    return OALGPU_OK;
}

/* what the context's exchange looks like from the inside: this rank, the world it was given, and the ranks the transport itself
 * counts (RCCL: ncclCommCount of the communicator the library created; -1: the library's RCCL has no such call) */
int oalgpu_comm_info(oalgpu_context *c, int *rank, int *world, int *transport_ranks, char *kind, size_t kind_size)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(rank) *rank = c->commRank;
    if(world) *world = c->commWorld;
    if(transport_ranks) *transport_ranks = c->comm ? c->comm->ranks() : 1;
    if(kind && kind_size) { std::snprintf(kind, kind_size, "%s", c->comm ? c->comm->kind() : "none"); }
    return OALGPU_OK;
}

int oalgpu_comm_destroy(oalgpu_context *c)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!c->comm) return OALGPU_OK;
    if(int rc = oalgpu_sync(c)) return rc;
    delete c->comm;
    c->comm = nullptr; c->commRank = 0; c->commWorld = 1; c->carryAccum = true;
    return OALGPU_OK;
}

const char *oalgpu_version(void) { return "oalgpu 0.1 (gfx950)"; }
const char *oalgpu_last_error(void) { return gLastError.c_str(); }

int oalgpu_device_count(void)
{
    int count = 0;
    if(hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count < 0 ? 0 : count;
}

// ---------------------------------------------------------------- host-side tables / params
int oalgpu_bsinc_table_get(int which, oalgpu_bsinc_table *out)
{
    const BsincTable *t = GetBsincTable(which);
    if(!t || !out) return Fail(OALGPU_ERR_INVALID, "bsinc table: which must be 12, 24 or 48");
    out->scaleBase = t->scaleBase; out->scaleRange = t->scaleRange;
    std::memcpy(out->m, t->m, sizeof(out->m));
    std::memcpy(out->filterOffset, t->filterOffset, sizeof(out->filterOffset));
    out->tab = t->tab.data(); out->tablen = t->tab.size();
    return OALGPU_OK;
}

int oalgpu_cubic_table_get(int which, float *out)
{
    const CubicTable *t = GetCubicTable(which);
    if(!t || !out) return Fail(OALGPU_ERR_INVALID, "cubic table: which must be 0 or 1");
    std::memcpy(out, t->phase, sizeof(t->phase));
    return OALGPU_OK;
}

int oalgpu_prepare_resampler(int resampler, uint32_t increment, oalgpu_interp_state *out)
{
    if(!out || resampler < 0 || resampler > OALGPU_RESAMPLER_BSINC48) return Fail(OALGPU_ERR_INVALID, "bad resampler");
    PrepareResampler(resampler, increment, out);
    return OALGPU_OK;
}

void oalgpu_biquad_reset(oalgpu_biquad *f)
{
    std::memset(f, 0, sizeof(*f));
    f->b0 = 1.0f; f->tb0 = 1.0f; f->counter = -1;
}

void oalgpu_biquad_set_params_from_slope(oalgpu_biquad *f, int type, float f0norm, float gain, float slope)
{
    float c[5];
    DesignBiquadFromSlope(type, f0norm, gain, slope, c);
    ApplyBiquadTarget(f, c);
}

void oalgpu_splitter_init(oalgpu_splitter *s, float f0norm)
{
    s->coeff = SplitterCoeff(f0norm);
    s->lp_z1 = s->lp_z2 = s->ap_z1 = 0.0f;
}

// ---------------------------------------------------------------- per-call kernels
int oalgpu_resample(int device, int mode, int resampler, uint32_t increment, const float *src, size_t srclen,
    uint32_t frac, float *dst, size_t n)
{
    if(!src || !dst || n == 0 || n > 4096 || frac >= kFracOne || increment < 1) return Fail(OALGPU_ERR_INVALID, "oalgpu_resample: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    oalgpu_interp_state st;
    if(int rc = oalgpu_prepare_resampler(resampler, increment, &st)) return rc;
    const TableBlob &blob = Blob();
    DevBuf<float> dTab, dSrc, dDst;
    HIP_TRY(dTab.alloc(blob.data.size())); HIP_TRY(dTab.upload(blob.data.data(), blob.data.size()));
    HIP_TRY(dSrc.alloc(srclen)); HIP_TRY(dSrc.upload(src, srclen));
    HIP_TRY(dDst.alloc(n));
    ResampleSpec spec{st.kind, st.m, st.l, st.sf, dTab.p + blob.filterBase(st)};
    LaunchResample(nullptr, mode == OALGPU_MATH_EXACT, spec, dSrc.p, frac, increment, dDst.p, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dDst.download(dst, n));
    return OALGPU_OK;
}

/* ---- SampleConverter (core/converter.h:15-59, core/converter.cpp:175-330): the second consumer of the resamplers ----
 * The integer bookkeeping of convert() / availableOut() runs here on the host, exactly as the reference's; the
 * float work of a call -- LoadSample<T>, the resampler, StoreSample<T> for every channel and chunk -- is one launch
 * (SampleConvertKernel, output_kernels.hip). */
struct oalgpu_converter {
    int device{0};
    int srcType{0}, dstType{0};
    uint32_t channels{1}, increment{kFracOne};
    uint32_t srcPrepCount{kMaxPad}, fracOffset{0};          // mSrcPrepCount, mFracOffset
    oalgpu_interp_state st{};
    DevBuf<float> tables, prev, prev2;
    DevBuf<unsigned char> src, dst;
    DevBuf<ConvertChunk> chunks;
    bool flip{false};                                       // which of prev / prev2 holds the current PrevSamples
};

static size_t DevFmtBytes(int type) { static const size_t b[7] = {1, 1, 2, 2, 4, 4, 4}; return b[type]; }

int oalgpu_converter_create(int device, int src_type, int dst_type, uint32_t channels, uint32_t src_rate, uint32_t dst_rate,
    int resampler, oalgpu_converter **out)
{
    if(!out || src_type < OALGPU_OUT_I8 || src_type > OALGPU_OUT_F32 || dst_type < OALGPU_OUT_I8 || dst_type > OALGPU_OUT_F32
        || channels < 1 || channels > 64 || src_rate < 1 || dst_rate < 1 || resampler < 0 || resampler > OALGPU_RESAMPLER_BSINC48)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_converter_create: bad arguments");
    *out = nullptr;
    if(int rc = UseDevice(device)) return rc;
    auto c = std::make_unique<oalgpu_converter>();
    c->device = device; c->srcType = src_type; c->dstType = dst_type; c->channels = channels;
    // converter.cpp:199-201: step = clamp(round(srcRate * MixerFracOne / dstRate), 1, MaxPitch * MixerFracOne)
    const double step = std::min(std::max(std::round(double(src_rate) * double(kFracOne) / double(dst_rate)), 1.0), 10.0 * double(kFracOne));
    c->increment = uint32_t(step);
    if(c->increment != kFracOne)
        if(int rc = oalgpu_prepare_resampler(resampler, c->increment, &c->st)) return rc;
    const TableBlob &blob = Blob();
    HIP_TRY(c->tables.alloc(blob.data.size())); HIP_TRY(c->tables.upload(blob.data.data(), blob.data.size()));
    HIP_TRY(c->prev.alloc(size_t{channels} * kMaxPad)); HIP_TRY(c->prev.zero());
    HIP_TRY(c->prev2.alloc(size_t{channels} * kMaxPad)); HIP_TRY(c->prev2.zero());
    *out = c.release();
    return OALGPU_OK;
}

void oalgpu_converter_destroy(oalgpu_converter *c)
{
    if(!c) return;
    (void)UseDevice(c->device);          // (a resident voice kernel on the device is told to leave first: it would sit out the synchronisation until its watchdog)
    (void)hipDeviceSynchronize();
    delete c;
}

/* SampleConverter::availableOut, converter.cpp:216-234 */
uint32_t oalgpu_converter_available_out(const oalgpu_converter *c, uint32_t src_frames)
{
    if(!c || src_frames < 1) return 0;
    const uint32_t prep = c->srcPrepCount;
    if(prep < kMaxPad && kMaxPad - prep >= src_frames) return 0;
    uint64_t size = uint64_t{prep} + src_frames - kMaxPad;
    size <<= kFracBits;
    size -= c->fracOffset;
    const uint64_t n = (size + c->increment - 1) / c->increment;
    return uint32_t(std::min<uint64_t>(std::max<uint64_t>(n, 1), 2147483647ull));
}

int oalgpu_converter_convert(oalgpu_converter *c, const void **src, uint32_t *src_frames, void *dst, uint32_t dst_frames)
{
    if(!c || !src || !src_frames || (!*src && *src_frames) || (!dst && dst_frames))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_converter_convert: bad arguments");
    if(int rc = UseDevice(c->device)) return rc;
    const uint32_t inc = c->increment, total = *src_frames;
    // ---- the chunk loop of convert(), integers only (converter.cpp:248-327)
    std::vector<ConvertChunk> chunks;
    uint32_t pos = 0, left = total, consumed = 0, prep = c->srcPrepCount, frac = c->fracOffset, base = 0;
    const uint32_t prep0 = prep;
    while(pos < dst_frames && left > 0)
    {
        const uint32_t readable = std::min(left, uint32_t(kLine) - prep);
        if(prep < kMaxPad && kMaxPad - prep >= readable)
        {   // not enough input for one output: keep what came (the reference reports it all as taken)
            prep += readable;
            left = 0;
            break;
        }
        uint64_t size = uint64_t{prep} + readable - kMaxPad;
        size <<= kFracBits;
        size -= frac;
        uint32_t dstSize = uint32_t(std::min<uint64_t>(std::max<uint64_t>((size + inc - 1) / inc, 1), uint64_t(kLine)));
        dstSize = std::min(dstSize, dst_frames - pos);
        const uint32_t posEnd = dstSize * inc + frac;
        const uint32_t srcEnd = posEnd >> kFracBits;
        const uint32_t nextPrep = std::min(prep + readable - srcEnd, uint32_t(kMaxPad));
        chunks.push_back(ConvertChunk{base, frac, pos, dstSize});
        const uint32_t srcRead = std::min(left, srcEnd + nextPrep - prep);
        base += srcEnd;
        prep = nextPrep;
        frac = posEnd & kFracMask;
        consumed += srcRead; left -= srcRead;
        pos += dstSize;
    }
    if(prep != c->srcPrepCount || !chunks.empty())
    {
        const size_t srcBytes = size_t{total} * c->channels * DevFmtBytes(c->srcType);
        const size_t dstBytes = size_t{pos} * c->channels * DevFmtBytes(c->dstType);
        if(c->src.n < srcBytes) HIP_TRY(c->src.alloc(srcBytes));
        if(c->dst.n < dstBytes) HIP_TRY(c->dst.alloc(std::max<size_t>(dstBytes, 16)));
        if(c->chunks.n < chunks.size()) HIP_TRY(c->chunks.alloc(std::max<size_t>(chunks.size(), 1)));
        HIP_TRY(hipMemcpy(c->src.p, *src, srcBytes, hipMemcpyHostToDevice));
        if(!chunks.empty()) HIP_TRY(c->chunks.upload(chunks.data(), chunks.size()));
        ConvertJob J{};
        const TableBlob &blob = Blob();
        J.spec = ResampleSpec{c->st.kind, c->st.m, c->st.l, c->st.sf, inc != kFracOne ? c->tables.p + blob.filterBase(c->st) : c->tables.p};
        J.increment = inc;
        J.prev = c->flip ? c->prev2.p : c->prev.p; J.newPrev = c->flip ? c->prev.p : c->prev2.p;
        J.src = c->src.p; J.dst = c->dst.p;
        J.prep0 = prep0; J.srcFrames = total; J.channels = c->channels; J.numChunks = uint32_t(chunks.size());
        J.endBase = base; J.nextPrep = prep;
        J.srcType = c->srcType; J.dstType = c->dstType;
        LaunchSampleConvert(nullptr, J, c->chunks.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if(dstBytes) HIP_TRY(hipMemcpy(dst, c->dst.p, dstBytes, hipMemcpyDeviceToHost));
        c->flip = !c->flip;
    }
    c->srcPrepCount = prep; c->fracOffset = frac;
    *src = static_cast<const unsigned char*>(*src) + size_t{consumed} * c->channels * DevFmtBytes(c->srcType);
    *src_frames = left;
    return int(pos);
}

int oalgpu_mix(int device, const float *in, size_t n, float *out, size_t nlines, float *current_gains,
    const float *target_gains, size_t counter, size_t outpos)
{
    if(!in || !out || !current_gains || !target_gains || n == 0 || n + outpos > kLine || nlines == 0 || nlines > 64)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    DevBuf<float> dIn, dOut, dCur, dTgt;
    HIP_TRY(dIn.alloc(n)); HIP_TRY(dIn.upload(in, n));
    HIP_TRY(dOut.alloc(nlines * kLine)); HIP_TRY(dOut.upload(out, nlines * kLine));
    HIP_TRY(dCur.alloc(nlines)); HIP_TRY(dCur.upload(current_gains, nlines));
    HIP_TRY(dTgt.alloc(nlines)); HIP_TRY(dTgt.upload(target_gains, nlines));
    LaunchMix(nullptr, dIn.p, uint32_t(n), dOut.p, uint32_t(nlines), dCur.p, dTgt.p, uint32_t(counter), uint32_t(outpos));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dOut.download(out, nlines * kLine));
    HIP_TRY(dCur.download(current_gains, nlines));
    return OALGPU_OK;
}

static int MixHrtfCommon(int device, int mode, const float *in, float *accum, uint32_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float step, const float *oldcoeffs, const uint32_t olddelay[2], float oldgain,
    int blend, size_t n)
{
    if(!in || !accum || !coeffs || !delay || n == 0 || n > kLine || irsize < 8 || irsize > kHrirLen
        || delay[0] > 63 || delay[1] > 63)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_hrtf: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    constexpr size_t accLen = (kLine + kHrirLen) * 2;
    DevBuf<float> dIn, dAcc, dCo, dOld;
    HIP_TRY(dIn.alloc(n + kHist)); HIP_TRY(dIn.upload(in, n + kHist));
    HIP_TRY(dAcc.alloc(accLen)); HIP_TRY(dAcc.upload(accum, accLen));
    HIP_TRY(dCo.alloc(kHrirLen * 2)); HIP_TRY(dCo.upload(coeffs, kHrirLen * 2));
    HIP_TRY(dOld.alloc(kHrirLen * 2));
    if(blend) HIP_TRY(dOld.upload(oldcoeffs, kHrirLen * 2));
    LaunchMixHrtf(nullptr, mode == OALGPU_MATH_EXACT, dIn.p, dAcc.p, irsize, dCo.p, delay[0], delay[1], gain, step,
        dOld.p, blend ? olddelay[0] : 0, blend ? olddelay[1] : 0, oldgain, blend, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dAcc.download(accum, accLen));
    return OALGPU_OK;
}

int oalgpu_mix_hrtf(int device, int mode, const float *in, float *accum, uint32_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float gainstep, size_t n)
{ return MixHrtfCommon(device, mode, in, accum, irsize, coeffs, delay, gain, gainstep, nullptr, nullptr, 0.0f, 0, n); }

int oalgpu_mix_hrtf_blend(int device, int mode, const float *in, float *accum, uint32_t irsize,
    const float *oldcoeffs, const uint32_t olddelay[2], float oldgain, const float *newcoeffs,
    const uint32_t newdelay[2], float newgainstep, size_t n)
{
    if(!oldcoeffs || !olddelay || olddelay[0] > 63 || olddelay[1] > 63) return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_hrtf_blend: bad arguments");
    return MixHrtfCommon(device, mode, in, accum, irsize, newcoeffs, newdelay, 0.0f, newgainstep, oldcoeffs, olddelay,
        oldgain, 1, n);
}

int oalgpu_mix_direct_hrtf(int device, int mode, float *left, float *right, const float *in, size_t nch,
    float *accum, oalgpu_splitter *splitters, const float *hfscales, const float *chan_coeffs, size_t irsize, size_t n)
{
    if(!left || !right || !in || !accum || !splitters || !hfscales || !chan_coeffs || nch == 0 || nch > 64 || n == 0
        || n > kLine || irsize < 8 || irsize > kHrirLen)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_direct_hrtf: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    constexpr size_t accLen = (kLine + kHrirLen) * 2;
    DevBuf<float> dL, dR, dIn, dAcc, dHf, dCo, dTemp;
    DevBuf<SplitterState> dSp;
    HIP_TRY(dL.alloc(kLine)); HIP_TRY(dL.upload(left, kLine));
    HIP_TRY(dR.alloc(kLine)); HIP_TRY(dR.upload(right, kLine));
    HIP_TRY(dIn.alloc(nch * kLine)); HIP_TRY(dIn.upload(in, nch * kLine));
    HIP_TRY(dAcc.alloc(accLen)); HIP_TRY(dAcc.upload(accum, accLen));
    HIP_TRY(dHf.alloc(nch)); HIP_TRY(dHf.upload(hfscales, nch));
    HIP_TRY(dCo.alloc(nch * kHrirLen * 2)); HIP_TRY(dCo.upload(chan_coeffs, nch * kHrirLen * 2));
    HIP_TRY(dTemp.alloc(nch * kLine + accLen));
    HIP_TRY(dSp.alloc(nch)); HIP_TRY(dSp.upload(reinterpret_cast<const SplitterState*>(splitters), nch));
    if(mode == OALGPU_MATH_EXACT)
        LaunchMixDirectHrtf(nullptr, true, dL.p, dR.p, dIn.p, uint32_t(nch), dAcc.p, dSp.p, dHf.p, dCo.p,
            uint32_t(irsize), uint32_t(n), dTemp.p);
    else
        LaunchPostDirectHrtfFast(nullptr, dL.p, dR.p, dIn.p, uint32_t(nch), dAcc.p, dSp.p, dHf.p, dCo.p,
            uint32_t(irsize), uint32_t(n), dTemp.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dL.download(left, kLine)); HIP_TRY(dR.download(right, kLine));
    HIP_TRY(dAcc.download(accum, accLen));
    HIP_TRY(dSp.download(reinterpret_cast<SplitterState*>(splitters), nch));
    return OALGPU_OK;
}

int oalgpu_biquad_dual_process(int device, oalgpu_biquad *f0, oalgpu_biquad *f1, const float *src, float *dst, size_t n)
{
    if(!f0 || !f1 || !src || !dst || n == 0 || n > (1u << 20)) return Fail(OALGPU_ERR_INVALID, "oalgpu_biquad_dual_process: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    DevBuf<BiquadState> dF;
    DevBuf<float> dSrc, dDst;
    BiquadState st[2];
    std::memcpy(&st[0], f0, sizeof(BiquadState)); std::memcpy(&st[1], f1, sizeof(BiquadState));
    HIP_TRY(dF.alloc(2)); HIP_TRY(dF.upload(st, 2));
    HIP_TRY(dSrc.alloc(n)); HIP_TRY(dSrc.upload(src, n));
    HIP_TRY(dDst.alloc(n));
    LaunchBiquadDual(nullptr, dF.p, dF.p + 1, dSrc.p, dDst.p, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dDst.download(dst, n));
    HIP_TRY(dF.download(st, 2));
    std::memcpy(f0, &st[0], sizeof(BiquadState)); std::memcpy(f1, &st[1], sizeof(BiquadState));
    return OALGPU_OK;
}

// ---------------------------------------------------------------- context
static int AllocStreamRows(oalgpu_context *c);

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
static int AllocStreamRows(oalgpu_context *c)
{
    DeviceLayout &L = c->L;
    const size_t nv = L.numVoices;
    HIP_TRY(c->streams.alloc(nv * L.streamsPerVoice * kLine)); HIP_TRY(c->streams.zero()); L.streams = c->streams.p;
    HIP_TRY(c->lineGains.alloc(nv * L.streamsPerVoice * LineBlockDwords(L.lineStride))); HIP_TRY(c->lineGains.zero());
    L.lineGains = c->lineGains.p;
    return OALGPU_OK;
}

// the parsed (or handed-over) store becomes the context's: host copy, HBM copy, voice filter arrays
static int InstallHrtfData(oalgpu_context *c, HrtfData &&parsed)
{
    // a data set at another rate than the device's is brought to the device's rate as GetLoadedHrtf does
    // (core/hrtf.cpp:539-606: every HRIR through the polyphase resampler, delays and IrSize rescaled)
    if(parsed.sampleRate != c->desc.sample_rate) ResampleHrtfData(parsed, c->desc.sample_rate);
    if(int rc = oalgpu_sync(c)) return rc;           // a second load replaces buffers the streams may still read
    c->hrtfHost = std::move(parsed);
    const HrtfData &h = c->hrtfHost;
    HIP_TRY(c->hFieldDist.alloc(h.fieldDistance.size())); HIP_TRY(c->hFieldDist.upload(h.fieldDistance.data(), h.fieldDistance.size()));
    HIP_TRY(c->hEvCount.alloc(h.fieldEvCount.size())); HIP_TRY(c->hEvCount.upload(h.fieldEvCount.data(), h.fieldEvCount.size()));
    HIP_TRY(c->hAzCount.alloc(h.elevAzCount.size())); HIP_TRY(c->hAzCount.upload(h.elevAzCount.data(), h.elevAzCount.size()));
    HIP_TRY(c->hIrOffset.alloc(h.elevIrOffset.size())); HIP_TRY(c->hIrOffset.upload(h.elevIrOffset.data(), h.elevIrOffset.size()));
    HIP_TRY(c->hCoeffs.alloc(h.coeffs.size())); HIP_TRY(c->hCoeffs.upload(h.coeffs.data(), h.coeffs.size()));
    HIP_TRY(c->hDelays.alloc(h.delays.size())); HIP_TRY(c->hDelays.upload(h.delays.data(), h.delays.size()));
    HrtfStoreDev &d = c->hrtfDev;
    d.irSize = h.irSize; d.numFields = uint32_t(h.fieldDistance.size()); d.numElevs = uint32_t(h.elevAzCount.size());
    d.numIrs = h.numIrs();
    d.fieldDistance = c->hFieldDist.p; d.fieldEvCount = c->hEvCount.p; d.elevAzCount = c->hAzCount.p;
    d.elevIrOffset = c->hIrOffset.p; d.coeffs = c->hCoeffs.p; d.delays = c->hDelays.p;
    c->hrtfLoaded = true;
    ++c->hrtfGeneration;

    DeviceLayout &L = c->L;
    L.hrirs = c->hCoeffs.p;
    L.irSize = h.irSize;
    L.irStride = (h.irSize + 15u) & ~15u;
    if(L.hrtf)
    {
        const size_t n = size_t{L.numVoices} * L.irStride * 2;
        HIP_TRY(c->hrtfOld.alloc(n)); HIP_TRY(c->hrtfOld.zero()); L.hrtfOld = c->hrtfOld.p;
        HIP_TRY(c->hrtfTgt.alloc(n)); HIP_TRY(c->hrtfTgt.zero()); L.hrtfTgt = c->hrtfTgt.p;
        if(!c->directSet) c->dIrSize = h.irSize;
    }
    // (a set with more than 64 taps: the send rows of such an HRTF context go through stream rows)
    if(L.accLines && WaveKernelAccLines(L) == 0)
    {
        L.accLines = 0;
        if(int rc = AllocStreamRows(c)) return rc;
    }
    // one voice per wavefront, sixteen per workgroup (voice_wave16.hip): its grid is voices / 16 workgroups -- never more partial
    // buses than the context's buffers were sized for (the wavefront-per-voice kernel's grid has at least twice as many)
    if(c->useWave)
    {
        const bool want16 = !(c->desc.flags & (OALGPU_CTX_WAVE_PAIRS | OALGPU_CTX_RESIDENT)) && Wave16Applies(L);
        uint32_t cus = 256u;
        {
            hipDeviceProp_t prop{};
            if(hipGetDeviceProperties(&prop, c->desc.device) == hipSuccess && prop.multiProcessorCount > 0) cus = uint32_t(prop.multiProcessorCount);
            else (void)hipGetLastError();
        }
        L.wave16 = want16 ? Wave16WavesFor(L.numVoices, cus) : 0u;
        const uint32_t groups = std::max<uint32_t>(1u, WaveKernelGroups(L));
        if(groups > c->groupsAllocated) return Fail(OALGPU_ERR_INVALID, "internal: the voice kernel's grid outgrew the partial buses");
        L.numGroups = groups; L.numLineGroups = groups;
        if(L.wave16) c->res.enabled = false;
    }
    return OALGPU_OK;
}

int oalgpu_hrtf_load_mhr(oalgpu_context *c, const void *data, size_t size)
{
    if(!c || !data) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;
    HrtfData parsed;
    const std::string err = ParseMhr(data, size, parsed);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    return InstallHrtfData(c, std::move(parsed));
}

/* An HrtfStore already in memory (core/hrtf.h:22-59), as the device holds it in DeviceBase::mHrtf once
 * GetLoadedHrtf (core/hrtf.cpp:471-620) has loaded -- and resampled -- a data set: the same arrays oalgpu_hrtf_raw
 * hands out. */
int oalgpu_hrtf_load_store(oalgpu_context *c, uint32_t sample_rate, uint32_t ir_size, const float *field_distance,
    const uint8_t *field_evcount, uint32_t num_fields, const uint16_t *elev_azcount, const uint16_t *elev_iroffset,
    uint32_t num_elevs, const float *coeffs, const uint8_t *delays, uint32_t num_irs)
{
    if(!c || !field_distance || !field_evcount || !elev_azcount || !elev_iroffset || !coeffs || !delays)
        return Fail(OALGPU_ERR_INVALID, "null argument");
    if(sample_rate == 0 || ir_size < 8 || ir_size > kHrirLen || num_fields == 0 || num_elevs == 0 || num_irs == 0)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: bad sizes");
    uint32_t evs = 0;
    for(uint32_t f = 0; f < num_fields; ++f)
    {
        if(field_evcount[f] < 2) return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: a field needs at least two elevations");
        evs += field_evcount[f];
    }
    if(evs != num_elevs) return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: the fields' elevation counts do not add up to num_elevs");
    for(uint32_t e = 0; e < num_elevs; ++e)
        if(elev_azcount[e] == 0 || uint32_t(elev_iroffset[e]) + elev_azcount[e] > num_irs)
            return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: an elevation's HRIRs lie outside the store");
    if(int rc = UseCtx(c)) return rc;
    HrtfData h;
    h.sampleRate = sample_rate; h.irSize = ir_size;
    h.fieldDistance.assign(field_distance, field_distance + num_fields);
    h.fieldEvCount.assign(field_evcount, field_evcount + num_fields);
    h.elevAzCount.assign(elev_azcount, elev_azcount + num_elevs);
    h.elevIrOffset.assign(elev_iroffset, elev_iroffset + num_elevs);
    h.coeffs.assign(coeffs, coeffs + size_t{num_irs} * kHrirLen * 2);
    h.delays.assign(delays, delays + size_t{num_irs} * 2);
    return InstallHrtfData(c, std::move(h));
}

int oalgpu_hrtf_info_get(oalgpu_context *c, oalgpu_hrtf_info *out)
{
    if(!c || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "no HRTF data set loaded");
    const HrtfData &h = c->hrtfHost;
    out->sample_rate = h.sampleRate; out->ir_size = h.irSize;
    out->num_fields = uint32_t(h.fieldDistance.size()); out->num_elevs = uint32_t(h.elevAzCount.size());
    out->num_irs = h.numIrs();
    return OALGPU_OK;
}

int oalgpu_hrtf_raw(oalgpu_context *c, float *field_distance, uint8_t *field_evcount, uint16_t *elev_azcount,
    uint16_t *elev_iroffset, float *coeffs, uint8_t *delays)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "no HRTF data set loaded");
    const HrtfData &h = c->hrtfHost;
    std::copy(h.fieldDistance.begin(), h.fieldDistance.end(), field_distance);
    std::copy(h.fieldEvCount.begin(), h.fieldEvCount.end(), field_evcount);
    std::copy(h.elevAzCount.begin(), h.elevAzCount.end(), elev_azcount);
    std::copy(h.elevIrOffset.begin(), h.elevIrOffset.end(), elev_iroffset);
    std::copy(h.coeffs.begin(), h.coeffs.end(), coeffs);
    std::copy(h.delays.begin(), h.delays.end(), delays);
    return OALGPU_OK;
}

int oalgpu_hrtf_get_coeffs(oalgpu_context *c, const float *dirs, size_t count, float *coeffs, uint32_t *delays)
{
    if(!c || !dirs || !coeffs || !delays || count == 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_get_coeffs: bad arguments");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "no HRTF data set loaded");
    if(int rc = UseCtx(c)) return rc;
    DevBuf<float> dDirs, dCo;
    DevBuf<uint32_t> dDel;
    HIP_TRY(dDirs.alloc(count * 4)); HIP_TRY(dDirs.upload(dirs, count * 4));
    HIP_TRY(dCo.alloc(count * kHrirLen * 2));
    HIP_TRY(dDel.alloc(count * 2));
    LaunchGetCoeffs(c->stream, c->hrtfDev, dDirs.p, uint32_t(count), dCo.p, dDel.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(dCo.download(coeffs, count * kHrirLen * 2));
    HIP_TRY(dDel.download(delays, count * 2));
    return OALGPU_OK;
}

int oalgpu_set_direct_hrtf(oalgpu_context *c, const float *chan_coeffs, const float *hfscales, float xover_norm,
    uint32_t irsize)
{
    if(!c || !chan_coeffs || !hfscales || irsize < 8 || irsize > kHrirLen) return Fail(OALGPU_ERR_INVALID, "oalgpu_set_direct_hrtf: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    const uint32_t nd = c->L.numDry;
    std::vector<SplitterState> sp(nd);
    for(auto &s : sp) { s.coeff = SplitterCoeff(xover_norm); s.lpZ1 = s.lpZ2 = s.apZ1 = 0.0f; }
    if(int rc = oalgpu_sync(c)) return rc;
    HIP_TRY(c->dSplit.upload(sp.data(), nd));
    HIP_TRY(c->dSplit2.upload(sp.data(), nd));
    c->dSplitCur = 0;
    c->dSplitCoeff = SplitterCoeff(xover_norm); c->runPowerSeg = 0;
    HIP_TRY(c->dHfScale.upload(hfscales, nd));
    {   // MixDirectHrtf applies IrSize taps (rounded up to even: ApplyCoeffs works on pairs); the decoder of a resampled
        // data set carries non-zero taps beyond that, which the fixed-length FIR of the FAST post-process must not see
        std::vector<float> cc(chan_coeffs, chan_coeffs + size_t{nd} * kHrirLen * 2);
        const uint32_t live = (irsize + 1u) & ~1u;
        for(uint32_t ch = 0; ch < nd; ++ch)
            for(uint32_t k = live; k < uint32_t(kHrirLen); ++k) { cc[(size_t{ch} * kHrirLen + k) * 2] = 0.0f; cc[(size_t{ch} * kHrirLen + k) * 2 + 1] = 0.0f; }
        HIP_TRY(c->dCoeffs.upload(cc.data(), cc.size()));
    }
    c->dIrSize = irsize;
    c->directSet = true;
    return OALGPU_OK;
}

/* DirectHrtfState::build (core/hrtf.cpp:266-366) on the context's data set: the ambisonic-to-binaural decoder of the
 * HRTF post-process from the host's virtual-speaker layout (alc/panning.cpp:861-1038 holds the reference's: AmbiPoints,
 * AmbiMatrix rows of 16, AmbiOrderHFGain[5]; InitHrtfPanning passes device->mIrSize = the store's IrSize and
 * perHrirMin for orders >= 3).  One-time host work in double precision, then oalgpu_set_direct_hrtf. */
int oalgpu_set_direct_hrtf_from_store(oalgpu_context *c, const float *points, const float *matrix, uint32_t num_points,
    const float *order_hf_gain, float xover_freq, uint32_t ir_size, int per_hrir_min)
{
    if(!c || !points || !matrix || !order_hf_gain || num_points == 0 || c->L.numDry > 16)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_set_direct_hrtf_from_store: bad arguments");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "oalgpu_set_direct_hrtf_from_store: no data set loaded");
    const DirectHrtfBuild b = BuildDirectHrtf(c->hrtfHost, ir_size ? ir_size : c->hrtfHost.irSize, per_hrir_min != 0, points, matrix,
        num_points, c->L.numDry, xover_freq, order_hf_gain);
    return oalgpu_set_direct_hrtf(c, b.coeffs.data(), b.hfScale.data(), b.xoverNorm, std::max<uint32_t>(b.irSize, 8u));
}

/* The same two pieces of one-time HRTF set-up as pure host functions (no device): a data set brought to `device_rate`
 * (0: its own) -- info, and coeffs (num_irs x 128 x 2) / delays (num_irs x 2) when not NULL -- and the decoder build. */
int oalgpu_hrtf_parse_host(const void *mhr, size_t size, uint32_t device_rate, oalgpu_hrtf_info *info, float *coeffs, uint8_t *delays)
{
    if(!mhr || !info) return Fail(OALGPU_ERR_INVALID, "null argument");
    HrtfData h;
    const std::string err = ParseMhr(mhr, size, h);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    if(device_rate) ResampleHrtfData(h, device_rate);
    info->sample_rate = h.sampleRate; info->ir_size = h.irSize; info->num_fields = uint32_t(h.fieldDistance.size());
    info->num_elevs = uint32_t(h.elevAzCount.size()); info->num_irs = h.numIrs();
    if(coeffs) std::memcpy(coeffs, h.coeffs.data(), h.coeffs.size() * sizeof(float));
    if(delays) std::memcpy(delays, h.delays.data(), h.delays.size());
    return OALGPU_OK;
}

// the host's view of the loaded store, for HrtfBlendFor
static HrtfStoreDev HostStoreView(const HrtfData &h)
{
    HrtfStoreDev st{};
    st.irSize = h.irSize; st.numFields = uint32_t(h.fieldDistance.size());
    st.numElevs = uint32_t(h.elevAzCount.size()); st.numIrs = h.numIrs();
    st.fieldDistance = h.fieldDistance.data(); st.fieldEvCount = h.fieldEvCount.data();
    st.elevAzCount = h.elevAzCount.data(); st.elevIrOffset = h.elevIrOffset.data();
    st.coeffs = h.coeffs.data(); st.delays = h.delays.data();
    return st;
}

int oalgpu_hrtf_blend_host(const void *mhr, size_t size, uint32_t device_rate, const float *dirs, size_t count,
    uint32_t *idx, float *w, float *pass, uint32_t *delays)
{
    if(!mhr || !dirs || !idx || !w || !pass || !delays) return Fail(OALGPU_ERR_INVALID, "null argument");
    HrtfData h;
    const std::string err = ParseMhr(mhr, size, h);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    if(device_rate) ResampleHrtfData(h, device_rate);
    const HrtfStoreDev store = HostStoreView(h);
    for(size_t i = 0; i < count; ++i)
    {
        const HrirBlend b = HrtfBlendFor(store, dirs[4 * i], dirs[4 * i + 1], dirs[4 * i + 2], dirs[4 * i + 3]);
        for(int k = 0; k < 4; ++k) { idx[4 * i + k] = b.idx[k]; w[4 * i + k] = b.w[k]; }
        pass[i] = b.passthru;
        delays[2 * i] = b.delay[0]; delays[2 * i + 1] = b.delay[1];
    }
    return OALGPU_OK;
}

int oalgpu_hrtf_build_direct_host(const void *mhr, size_t size, uint32_t device_rate, uint32_t ir_size, int per_hrir_min,
    const float *points, const float *matrix, uint32_t num_points, uint32_t num_chans, float xover_freq, const float *order_hf_gain,
    float *out_coeffs, float *out_hfscales, float *out_xover_norm, uint32_t *out_irsize)
{
    if(!mhr || !points || !matrix || !order_hf_gain || !out_coeffs || !out_hfscales || num_chans < 1 || num_chans > 16 || num_points == 0)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_build_direct_host: bad arguments");
    HrtfData h;
    const std::string err = ParseMhr(mhr, size, h);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    if(device_rate) ResampleHrtfData(h, device_rate);
    const DirectHrtfBuild b = BuildDirectHrtf(h, ir_size ? ir_size : h.irSize, per_hrir_min != 0, points, matrix, num_points,
        num_chans, xover_freq, order_hf_gain);
    std::memcpy(out_coeffs, b.coeffs.data(), b.coeffs.size() * sizeof(float));
    std::memcpy(out_hfscales, b.hfScale.data(), b.hfScale.size() * sizeof(float));
    if(out_xover_norm) *out_xover_norm = b.xoverNorm;
    if(out_irsize) *out_irsize = b.irSize;
    return OALGPU_OK;
}

// ---- buffer handles: allocation, references, release ----
static bool BufferLive(const oalgpu_context *c, int h) { return h >= 0 && uint32_t(h) < c->numBuffers && c->bufHost[size_t(h)].live; }
static int AllocBufferHandle(oalgpu_context *c, uint32_t *out)
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
static bool HostStoresReachDevice(oalgpu_context *c)
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
static int JoinPost(oalgpu_context *c)
{
    if(!c->postPending) return OALGPU_OK;
    HIP_TRY(hipStreamWaitEvent(c->stream, c->lastPostEvent ? c->lastPostEvent : c->evPostDone, 0));
    c->postPending = false;
    return OALGPU_OK;
}


// ---- the resident voice kernel: host side (device side and protocol: kernels.hpp ResidentDoor, voice_wave.hip, post_wave.hip) ----
static bool ResidentWanted(const oalgpu_context *c, int post_process)
{
    const auto &R = c->res;
    return R.enabled && !R.failed && WaveKernelHasResident(c->L) && post_process && c->hrtfLoaded && c->directSet && !c->timing && c->cbVoices.empty()
        && c->initPending.empty() && c->carryAccum && !c->comm && !c->pendingMix.active && c->useWave && c->ownStream && !c->serialOnly
        && c->L.numReal >= 2 && c->L.numSlots == 0;
}

// a parameter block that was waiting for a resident update is applied the launched way (the caller has parked the kernel)
static int FlushResidentBlock(oalgpu_context *c)
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
    R.groupsPerCu = uint32_t(std::max(0, WaveResidentGroupsPerCu()));
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

} // namespace

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

static int FlushPendingMix(oalgpu_context *c, oalgpu_param_block *next)
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

} // extern "C"
