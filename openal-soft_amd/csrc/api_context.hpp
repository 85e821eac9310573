// The device context and what the C-ABI translation units share (api.hip: context life, the update, slots; api_comm.hip: the
// sharded update's transports; api_percall.hip: the per-call operators; api_hrtf.hip: HRTF data sets; api_voices.hip: buffers, voices,
// parameters; api_output.hip: what comes back; api_callback.hip: callback sources).  Host logic only; no CPU fallback anywhere.
#pragma once
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
inline const TableBlob &Blob() { static const TableBlob b; return b; }

} // namespace oalgpu

struct BusTransport;
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

namespace oalgpu { extern thread_local std::string gLastError; }

// ---- shared between the translation units (definitions: api.hip unless noted) ----
int FlushPendingMix(oalgpu_context *c, struct oalgpu_param_block *next = nullptr);
bool ResidentWanted(const oalgpu_context *c, int post_process);
int FlushResidentBlock(oalgpu_context *c);
int UseCtx(oalgpu_context *c);
int UseCtxResident(oalgpu_context *c);
void RetireCallbackVoice(oalgpu_context *c, uint32_t voice);
void NoteCallbackSteps(oalgpu_context *c, const uint32_t *voices, const oalgpu_voice_params *params, size_t count);
int FlushInits(oalgpu_context *c);
int AllocStreamRows(oalgpu_context *c);
int JoinPost(oalgpu_context *c);
bool HostStoresReachDevice(oalgpu_context *c);                     // api_voices.hip
int AllocBufferHandle(oalgpu_context *c, uint32_t *out);           // api_voices.hip
oalgpu::HrtfStoreDev HostStoreView(const oalgpu::HrtfData &h);     // api_hrtf.hip
int ServiceCallbacks(oalgpu_context *c, uint32_t samplesToDo);     // api_callback.hip
int CommReduceBus(oalgpu_context *c, hipStream_t s);               // api_comm.hip

