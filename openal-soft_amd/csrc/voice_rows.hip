// Dry lines AND sends (BASELINE configs[3]: 8192 voices, 5 dry lines + 4 x 4 wet lines): every voice's rows stay on the
// compute unit, and a voice's fixed work is done ONCE.
//
//   "for every Playing|Stopping voice: voice->mix(...)"   alc/alu.cpp:2201-2206
//      -> Voice::mix                                       core/voice.cpp:988-1233
//      -> LoadResampledSamples / Resample_*                voice.cpp:642-824
//      -> DoFilters, direct path and per send              voice.cpp:255-267, :966-983
//      -> MixSamples onto the dry lines and the slots' wet lines   voice.cpp:934-984, core/mixer/mixer_c.cpp:150-259
//
// The two forms this replaces both pay for 21 mix lines.  voice_wave.hip's stream rows: a wavefront per voice leaves a 4 KB row
// per mixed signal in HBM and the workgroup's tail reads them back -- 6.5 x the algorithmic bytes.  voice_slice.hip: a wavefront
// per 256-frame slice keeps the lines in registers and never writes a row -- and resamples, filters and resolves every voice
// four times (instruction-bound: twice the time).  Here the cut is along BOTH axes:
//   * PRODUCE: a wavefront owns a voice -- request, resampler (the register form of voice_wave16.hip: outputs lane + 64 j in 16
//     registers) -- once per voice; what MixSamples would read goes into a 4 KB row slot in LDS together with the resolved
//     gains of the lines it feeds (line = lane);
//   * CONSUME: a wavefront owns a 128-frame SLICE of every mix line -- 32 lines x 2 frames per lane in 64 registers for the
//     whole launch -- and adds its slice of all eight slots' rows, gains through v_readlane, four lines at a time, only the
//     blocks a row feeds.  No accumulator is shared, dumped or summed across wavefronts; each stores its slice of the
//     workgroup's partial bus at the end.  Deterministic: rows are added in voice order, filtered ones in (voice, kind) order.
// A round = eight voices.  A voice has up to 2 + sends signals: the unfiltered samples (for every line whose filter is inactive:
// ONE row with the merged gains), the direct-filtered samples, a send's own filtered copy.  The unfiltered rows are slot A of
// the round; every filtered signal is a JOB that any wavefront of the workgroup can run (the owner posts the pair's state and
// the resolved gains, the executor scans a copy of the owner's slot-A row in its own second row): the round's DoFilters work is
// spread over all eight wavefronts whichever voices it belongs to -- one batch of scans for up to eight filtered signals.
// Measured on BASELINE configs[3] (profiles/r6/config4_rows_ab.txt): 127 us per launch against 138 us of the stream rows, a
// step of 135 against 148 us; the default form of these contexts (api.hip).
#define OALGPU_WAVE_NO_LAUNCHER
#include "voice_wave.hip"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kRowsWaves = 8;
constexpr int kRowsThreads = kRowsWaves * 64;
constexpr int kRowsOuts = kLine / 64;
constexpr int kRowsLines = 32;                          // line accumulators per wavefront: 8 blocks of 4
constexpr int kRowsSlice = kLine / kRowsWaves;          // 128 frames: two per lane

constexpr int kRowsRd = 1184, kRowsRd2 = kMaxEdge + kPre * 64 + 8;
constexpr int kRowsArea = kResampleDataSize + kLine;    // the generic loader's window + sample line (voice_wave16.hip)
static_assert(kRowsRd + kRowsRd2 <= kRowsArea, "the twice-parked window fits");
struct alignas(16) RowsLds {
    union {
        struct { float rd[kRowsRd]; float rd2[kRowsRd2]; } a;
        struct { float rd[kResampleDataSize]; float smp[kLine]; } g;
    };
    float row[kLine];                                   // the slot: what MixSamples reads of this wavefront's voice, one signal at a time
    float gain[kRowsLines], rampA[kRowsLines], rampB[kRowsLines];   // the slot's resolved gains, line = index (StoreRowBlock's form)
    uint32_t live, blocks, maxFade, pad0;               // live: the slot holds a row; blocks: which 4-line blocks it feeds
    int32_t best;
    uint32_t pad[3];
};
// a filter job: a signal of some voice of the round that needs DoFilters (gain/rampA/rampB consecutive: the consumers index them
// from gain)
struct alignas(16) RowsJob {
    float fst[32];                                      // the pair's two BiquadStates
    float gain[kRowsLines], rampA[kRowsLines], rampB[kRowsLines];
    uint32_t live, blocks, maxFade, src, outPos, slotsLo, slotsHi, pad;     // src: the owner's wavefront (its slot-A row is the input)
};
struct RowsWg {
    alignas(16) f2 tabF[12 * 32];
    f2 tabP[12 * 32];
    uint32_t tabKey, tabM, tabL, pad;
    uint32_t cnt[2][kRowsWaves];                // the owners' pending jobs, per batch parity
    RowsJob jobs[kRowsWaves];
    RowsLds w[kRowsWaves];
};
static_assert(kResampleDataSize >= kLine, "a wavefront's second row lives in the resampler's window area");
static_assert(offsetof(RowsLds, rampA) == offsetof(RowsLds, gain) + sizeof(float) * kRowsLines
    && offsetof(RowsLds, rampB) == offsetof(RowsLds, gain) + 2 * sizeof(float) * kRowsLines, "gain, rampA, rampB are one array to the consumers");
static_assert(sizeof(RowsWg) <= 131072, "one workgroup per compute unit, 32 KB left");

struct RowsGenView { float *rd, *in, *rd2; int32_t &best; uint32_t *pad; };

// ResampleRingRegs of voice_wave16.hip (the same statements; that file's anonymous namespace is not this one's)
template<int M, bool DUAL>
__device__ __forceinline__ void RowsResample(const f2 *tabF, const f2 *tabP, const float *rdb, uint32_t frac0, uint32_t increment,
    uint32_t bdst, float (&outs)[kRowsOuts], uint32_t lane, const float *rd2b, uint32_t rdbIndex)
{
    constexpr int NP = (M / 2 >= 6) ? 6 : M / 2;
    constexpr int G = (M / 2) / NP;
    f2 F[2][NP], P[2][NP], S[2][NP];
    const uint32_t tstep = 64u * increment;
    const uint32_t tlast = frac0 + (bdst - 1u) * increment;
    auto load = [&](int set, uint32_t tt, int g)
    {
        tt = tt < tlast ? tt : tlast;
        const uint32_t pi = (tt >> 11) & 31u;
        const f2 *tf = tabF + pi, *tp = tabP + pi;
        const uint32_t pos = tt >> kFracBits;
#pragma unroll
        for(int q = 0; q < NP; ++q)
        {
            F[set][q] = tf[(g * NP + q) * 32];
            P[set][q] = tp[(g * NP + q) * 32];
        }
        if constexpr (DUAL)
        {
            const bool odd = ((rdbIndex + pos) & 1u) != 0u;
            const f2 *sp = reinterpret_cast<const f2*>(odd ? rd2b + pos - 1u : rdb + pos);
#pragma unroll
            for(int q = 0; q < NP; ++q) S[set][q] = sp[g * NP + q];
        }
        else
        {
            const float *s = rdb + pos;
#pragma unroll
            for(int q = 0; q < NP; ++q) S[set][q] = f2{s[2 * (g * NP + q)], s[2 * (g * NP + q) + 1]};
        }
    };
    const uint32_t tb = frac0 + lane * increment;
    load(0, tb, 0);
#pragma unroll
    for(int j = 0; j < kRowsOuts; ++j)
    {
        outs[j] = 0.0f;
        if(uint32_t(64 * j) < bdst)
        {
            const uint32_t tt0 = tb + uint32_t(j) * tstep;
            const uint32_t tt = tt0 < tlast ? tt0 : tlast;
            const f2 pf = splat(float(tt & 2047u) * (1.0f / 2048.0f));
            f2 r0 = {0.0f, 0.0f}, r1 = {0.0f, 0.0f};
#pragma unroll
            for(int g = 0; g < G; ++g)
            {
                const int u = j * G + g, nx = u + 1;
                if(nx < kRowsOuts * G) load(nx & 1, tb + uint32_t(nx / G) * tstep, nx % G);
#pragma unroll
                for(int q = 0; q < NP; ++q)
                {
                    const f2 c = pkfma(pf, P[u & 1][q], F[u & 1][q]);
                    if(q & 1) r1 = pkfma(c, S[u & 1][q], r1);
                    else r0 = pkfma(c, S[u & 1][q], r0);
                }
            }
            outs[j] = (r0.x + r0.y) + (r1.x + r1.y);
        }
        // (an output at a time: with 256 registers to fill -- two wavefronts per SIMD -- the scheduler otherwise pulls the reads of
        // many outputs ahead and the allocator spills what the accumulators and the request in flight leave no room for)
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ float RowsReadLaneF(float v, int l)
{ return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// the slot's gains (StoreRowBlock's form: a constant per line, and for the first maxFade frames the ramp's distance from it)
template<typename Slot>
__device__ __forceinline__ void RowsPublishGains(Slot &w, const RowLineGain &r, uint32_t lane)
{
    uint32_t maxFade = r.fadeLen;
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = uint32_t(__shfl_xor(int(maxFade), d)); maxFade = o > maxFade ? o : maxFade; }
    const float ra = r.fadeLen ? r.cur - r.gain : 0.0f, rb = r.fadeLen ? r.step : 0.0f;
    if(lane < uint32_t(kRowsLines)) { w.gain[lane] = r.gain; w.rampA[lane] = ra; w.rampB[lane] = rb; }
    const bool nz = lane < uint32_t(kRowsLines) && (r.gain != 0.0f || ra != 0.0f || rb != 0.0f);
    const unsigned long long m = __ballot(nz);
    uint32_t blocks = 0u;
#pragma unroll
    for(int b = 0; b < kRowsLines / 4; ++b) blocks |= ((m >> (4 * b)) & 0xFull) ? (1u << b) : 0u;
    if(lane == 0) { w.live = 1u; w.blocks = blocks; w.maxFade = maxFade; }
}

// PROF: the measurement variant (tools/phase_times_rows.py): s_memtime stamps per workgroup, round and wavefront
template<bool PROF>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(kRowsThreads) VoiceRowsKernel(DeviceLayout L, uint32_t samplesToDo, uint32_t vpg, NextBlock next, WaveProf prof)
{
    __shared__ RowsWg sm;
    const uint32_t t = threadIdx.x;
    uint32_t lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t group = blockIdx.x;
    const uint32_t N = samplesToDo;
    RowsLds &w = sm.w[wave];
    const uint32_t numDry = L.numDry, numSends = L.numSends, wetCh = L.wetChannels;
    const uint32_t vBegin = group * vpg;
    const uint32_t vEnd = (vBegin + vpg < L.numVoices) ? vBegin + vpg : L.numVoices;
    const uint32_t lastVoice = L.numVoices - 1u;

    f2 acc[kRowsLines];                         // (a line's two frames are ONE packed operand: v_pk_fma_f32 with the gain in a scalar pair)
#pragma unroll
    for(int c = 0; c < kRowsLines; ++c) acc[c] = f2{0.0f, 0.0f};

    // ---- the workgroup's resampler rows: those of its first voice (LDS-DMA, voice_wave16.hip); a voice on other rows takes the
    // generic loader ----
    const uint32_t keyVoice = vBegin < L.numVoices ? vBegin : lastVoice;
    const VoiceHead headK = LoadHeadScalar(L.ctl + keyVoice);
    const int psK = headK.playState, kK = headK.rsKind;
    const uint32_t mK = kK == 2 ? 4u : headK.rsM, lK = kK == 2 ? 1u : headK.rsL, offK = headK.rsFilterOffset;
    const bool eligK = vBegin < L.numVoices && (kK == 2 || (kK == 3 && (mK == 12 || mK == 24)))
        && (psK == OALGPU_VOICE_PLAYING || psK == OALGPU_VOICE_STOPPING);
    if(eligK)
    {
        typedef const __attribute__((address_space(1))) void *gvoidp;
        typedef __attribute__((address_space(3))) void *lvoidp;
        const float *src = L.tables + offK + size_t{lane >> 1} * (2u * mK) + (lane & 1u);
        for(uint32_t pp = wave; pp < mK / 2u; pp += uint32_t(kRowsWaves))
        {
            __builtin_amdgcn_global_load_lds((gvoidp)(src + 2u * pp), (lvoidp)&sm.tabF[32u * pp], 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoidp)(src + mK + 2u * pp), (lvoidp)&sm.tabP[32u * pp], 4, 0, 0);
        }
    }
    if(t == 0) { sm.tabKey = eligK ? offK * 8u + uint32_t(kK) : 0xffffffffu; sm.tabM = eligK ? mK : 0u; sm.tabL = eligK ? lK : 0u; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const uint32_t rounds = (vEnd > vBegin) ? (vEnd - vBegin + uint32_t(kRowsWaves) - 1u) / uint32_t(kRowsWaves) : 0u;
    const uint32_t wetBase = numDry;

    // ---- a voice's request, one round ahead: control line, plan, the source window in registers, mPrevSamples, the dry gains ----
    struct Req {
        bool haveVoice, mixes, playing, active, looping, regPath;
        uint32_t v, sM, sL;
        VoiceHead head; BufferItem buf; SrcPlan plan;
        int32_t sendSlots[6];
    };
    Req rq{};
    float preN[kPre];
    float prevN = 0.0f, dryTgN = 0.0f, dryCuN = 0.0f;
    // the sends' gains (lane = send x wet channel, two registers: <= 128 of them) and every filter pair's first 16 dwords per slot
    // (lane = 32 x pair + dword; pair 0 = the direct path's, 1 + si = send si's: up to seven pairs in four registers) come with the
    // request too: fetched where they are used they were two dependent round trips to L2 per send and round
    float sTgN[2] = {0.0f, 0.0f}, sCuN[2] = {0.0f, 0.0f}, fltN[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const uint32_t sendLanes = numSends * wetCh;
#pragma unroll
    for(int i = 0; i < kPre; ++i) preN[i] = 0.0f;
    // (the scalar half runs twice per voice: when the voice's window is requested, and again at the start of its round -- the
    // control line comes from L2 in a few hundred cycles, where some forty scalar values carried across a round went through
    // spill lanes at every use)
    auto requestScalars = [&](uint32_t round) __attribute__((always_inline))
    {
        const uint32_t vRaw = vBegin + round * uint32_t(kRowsWaves) + wave;
        rq.haveVoice = round < rounds && vRaw < vEnd;
        rq.v = rq.haveVoice ? vRaw : lastVoice;
        rq.head = LoadHeadScalar(L.ctl + rq.v);
        rq.buf = LoadCtlBufferScalar(L.ctl + rq.v);
        {
            static_assert(offsetof(VoiceCtl, sendSlot) == 48, "VoiceCtl::sendSlot follows the head");
            cu4 *src = (cu4*)(uintptr_t)(L.ctl + rq.v);
            const u4 a = src[3], b = src[4];
            rq.sendSlots[0] = int32_t(a.x); rq.sendSlots[1] = int32_t(a.y); rq.sendSlots[2] = int32_t(a.z); rq.sendSlots[3] = int32_t(a.w);
            rq.sendSlots[4] = int32_t(b.x); rq.sendSlots[5] = int32_t(b.y);
        }
        const int vstate = rq.head.playState;
        rq.mixes = rq.haveVoice && (vstate == OALGPU_VOICE_PLAYING || vstate == OALGPU_VOICE_STOPPING);
        rq.playing = rq.haveVoice && vstate == OALGPU_VOICE_PLAYING;
        rq.active = rq.mixes && rq.head.step >= 1u;
        rq.plan = PlanSource(rq.head, N);
        rq.looping = false;
        if(rq.head.curBuffer >= 0)
        {   // voice.cpp:1015-1019
            rq.looping = rq.head.loopBuffer >= 0 && ((rq.head.flags & kFlagQueue) || !(rq.head.position >= 0 && uint32_t(rq.head.position) >= rq.buf.loopEnd));
            rq.plan.prefetch = rq.plan.prefetch && GatherCovers(rq.plan.bsrc, rq.buf, rq.looping, uint32_t(rq.head.position));
        }
        if(rq.head.flags & (kFlagDelayed | kFlagQueue)) rq.plan.prefetch = false;
        rq.sM = rq.head.rsKind == 2 ? 4u : rq.head.rsM; rq.sL = rq.head.rsKind == 2 ? 1u : rq.head.rsL;
        rq.regPath = rq.active && rq.plan.prefetch && eligK && (rq.head.rsKind == 2 || rq.head.rsKind == 3)
            && rq.head.rsFilterOffset * 8u + uint32_t(rq.head.rsKind) == offK * 8u + uint32_t(kK) && rq.sM == mK
            && !(rq.head.step == kFracOne && rq.head.positionFrac == 0u);
    };
    auto request = [&](uint32_t round) __attribute__((always_inline))
    {
        requestScalars(round);
        if(rq.regPath)
        {
            GatherStatic(preN, rq.plan.bsrc, rq.buf, rq.looping, uint32_t(rq.head.position), lane);
            prevN = (lane < kMaxPad) ? L.prev[size_t{rq.v} * kMaxPad + lane] : 0.0f;
        }
        dryTgN = dryCuN = 0.0f;
        if(rq.active && lane < numDry) { dryTgN = L.gainTgt[size_t{rq.v} * numDry + lane]; dryCuN = L.gainCur[size_t{rq.v} * numDry + lane]; }
#pragma unroll
        for(int h = 0; h < 2; ++h)
        {
            const uint32_t k = lane + 64u * uint32_t(h);
            sTgN[h] = (rq.active && k < sendLanes) ? L.sendTgt[size_t{rq.v} * sendLanes + k] : 0.0f;
            sCuN[h] = (rq.active && k < sendLanes) ? L.sendCur[size_t{rq.v} * sendLanes + k] : 0.0f;
        }
#pragma unroll
        for(int h = 0; h < 4; ++h)
        {
            const uint32_t pair = 2u * uint32_t(h) + (lane >> 5);          // 0: direct, 1 + si: send si
            const float *src = pair == 0u ? reinterpret_cast<const float*>(L.dfilt + size_t{rq.v} * 2)
                : reinterpret_cast<const float*>(L.sfilt + (size_t{rq.v} * numSends + (pair - 1u)) * 2);
            fltN[h] = (rq.active && pair <= numSends) ? src[lane & 31u] : 0.0f;
        }
    };
    request(0u);

    for(uint32_t round = 0; round < rounds; ++round)
    {
        asm volatile("" : "+v"(lane));
        auto stamp = [&](int slot)
        {
            if constexpr (PROF)
            {
                if(prof.times && (t & 63u) == 0u && round < 8u)
                    prof.times[((size_t{group} * 8u + round) * 8u + wave) * 8u + slot] = __builtin_readcyclecounter();
            }
        };
        stamp(0);
        // ================= PRODUCE: this wavefront's voice (requested one round ago) =================
        requestScalars(round);
        const Req cur = rq;
        // (the send slots as six values of their own: picked through an address, the whole request would live in scratch)
        const int32_t ss0 = rq.sendSlots[0], ss1 = rq.sendSlots[1], ss2 = rq.sendSlots[2], ss3 = rq.sendSlots[3], ss4 = rq.sendSlots[4], ss5 = rq.sendSlots[5];
        auto slotOf = [&](uint32_t k) __attribute__((always_inline))
        { return k == 1u ? ss1 : k == 2u ? ss2 : k == 3u ? ss3 : k == 4u ? ss4 : k == 5u ? ss5 : ss0; };
        const uint32_t v = cur.v;
        const VoiceHead head = cur.head;
        const BufferItem buf = cur.buf;
        const bool haveVoice = cur.haveVoice, mixes = cur.mixes, playing = cur.playing, looping = cur.looping, regPath = cur.regPath;
        bool active = cur.active;
        const float dryTg = dryTgN, dryCu = dryCuN;
        const float sTg[2] = {sTgN[0], sTgN[1]}, sCu[2] = {sCuN[0], sCuN[1]}, flt[4] = {fltN[0], fltN[1], fltN[2], fltN[3]};
        // send si's gain of wet channel (lane - base) to the lanes of the slot's lines; a filter pair's 32 dwords into the wavefront's scratch
        auto sendGain = [&](const float (&g)[2], uint32_t si, uint32_t base)
        {
            const uint32_t from = si * wetCh + (lane - base);
            const float a = __shfl(g[0], int(from & 63u)), b = __shfl(g[1], int(from & 63u));
            return from < 64u ? a : b;
        };
        // which pairs are at rest -- states and counter zero, current == target coefficients: what WaveDoFilters checks before it clears
        // an INACTIVE pair (voice.cpp:264-265) -- decided on the words in registers (voice_wave.hip's parking does the same): an
        // inactive pair that already is clear costs nothing
        uint32_t restMask = 0u;
#pragma unroll
        for(int h = 0; h < 4; ++h)
        {
            const uint32_t li = lane & 15u, bits = __builtin_bit_cast(uint32_t, flt[h]);
            const float ahead = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, int(bits), 0x105, 0xF, 0xF, true));   // row_shl:5
            bool ok = true;                         // (a pair is two slots of 16 dwords: every row of 16 lanes is one BiquadState)
            if(li < 2u) ok = flt[h] == 0.0f;
            else if(li < 7u) ok = flt[h] == ahead;
            else if(li == 12u) ok = bits == 0u;
            const unsigned long long bad = __ballot(!ok);
            if((bad & 0xffffffffull) == 0ull) restMask |= 1u << (2 * h);
            if((bad >> 32) == 0ull) restMask |= 2u << (2 * h);
        }
        auto stagePair = [&](uint32_t pair)
        {
            float v = 0.0f;
#pragma unroll
            for(int h = 0; h < 4; ++h) { const float x = __shfl(flt[h], int(((pair & 1u) << 5) | (lane & 31u))); if((pair >> 1) == uint32_t(h)) v = x; }
            WaveSync();
            if(lane < 32u) sm.jobs[wave].fst[lane] = v;     // (the wavefront's own job record: nothing is posted before the round's first barrier)
            WaveSync();
        };
        if(regPath)
        {   // the window is parked twice (rd2[i] = rd[i + 1]: aligned pair reads, voice_wave.hip)
            const bool isShort = buf.fmt == OALGPU_FMT_SHORT;
            if(lane < kMaxPad) { w.a.rd[lane] = prevN; if(lane) w.a.rd2[lane - 1u] = prevN; }
#pragma unroll
            for(int i = 0; i < kPre; ++i)
            {
                const float sv = GatherDecode(preN[i], isShort);
                w.a.rd[kMaxEdge + lane + 64u * uint32_t(i)] = sv;
                w.a.rd2[kMaxEdge - 1u + lane + 64u * uint32_t(i)] = sv;
            }
        }
        // voice.cpp:1002-1046: what does not mix; delayed starts
        uint32_t outPos = 0;
        if(mixes && !active && !playing && lane == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
        if(active && (head.flags & kFlagDelayed))
        {
            const uint32_t d = L.startDelay[v];
            if(!playing)
            {
                if(lane == 0) { L.ctl[v].playState = OALGPU_VOICE_STOPPED; L.ctl[v].flags = head.flags & ~kFlagDelayed; L.startDelay[v] = 0u; }
                active = false;
            }
            else if(d >= N) { if(lane == 0) L.startDelay[v] = d - N; active = false; }
            else
            {
                outPos = d;
                if(lane == 0) L.startDelay[v] = 0u;
                for(uint32_t k = lane; k < outPos; k += 64) w.g.smp[k] = 0.0f;
            }
        }
        float outs[kRowsOuts];
        {   // (element by element behind an opaque value: written as a plain loop this is a memset of the whole array, and an array
            // that is touched as a whole becomes ONE 16-register vector value -- copied tuple by tuple at every element update)
            float zero = 0.0f;
            asm volatile("" : "+v"(zero));
#pragma unroll
            for(int j = 0; j < kRowsOuts; ++j) outs[j] = zero;
        }
        int32_t bufferItem = head.curBuffer;
        uint32_t counter = 0;
        uint32_t order = 0u;                        // the signals this voice has: bit si = send si's filtered copy, numSends = unfiltered, numSends + 1 = direct-filtered
        bool directFilter = false;
        RowLineGain row0;                           // the unfiltered signal's merged gains, line = lane
        if(active)
        {
            if(regPath)
            {
                WaveSync();
                const uint32_t sM = cur.sM, sL = cur.sL;
                const float *rdb = w.a.rd + (kMaxEdge - sL), *rd2b = w.a.rd2 + (kMaxEdge - sL);
                const uint32_t rdbIndex = uint32_t(kMaxEdge) - sL;
                if(sM == 24u) RowsResample<24, true>(sm.tabF, sm.tabP, rdb, head.positionFrac, head.step, N, outs, lane, rd2b, rdbIndex);
                else if(sM == 12u) RowsResample<12, true>(sm.tabF, sm.tabP, rdb, head.positionFrac, head.step, N, outs, lane, rd2b, rdbIndex);
                else RowsResample<4, false>(sm.tabF, sm.tabP, rdb, head.positionFrac, head.step, N, outs, lane, nullptr, 0u);
                asm volatile("" : "+v"(lane));
                if(playing)
                {   // voice.cpp:772-785
                    const uint32_t srcOffset = uint32_t((uint64_t{N} * head.step + head.positionFrac) >> kFracBits);
                    if(lane < kMaxPad) L.prev[size_t{v} * kMaxPad + lane] = w.a.rd[srcOffset + lane];
                }
            }
            else
            {
                RowsGenView gv{w.g.rd, w.g.smp - kHist, nullptr, w.best, w.pad};
                LoadResampledWave<true, false>(sm, gv, L, v, lane, head, playing, N - outPos, N - outPos, bufferItem, looping,
                    SrcPlan{false, cur.plan.bdst, cur.plan.bsrc}, outPos);
                asm volatile("" : "+v"(lane));
                WaveSync();
#pragma unroll
                for(int j = 0; j < kRowsOuts; ++j) outs[j] = w.g.smp[lane + 64u * uint32_t(j)];
            }
            if(N < uint32_t(kLine))
            {
#pragma unroll
                for(int j = 0; j < kRowsOuts; ++j) if(lane + 64u * uint32_t(j) >= N) outs[j] = 0.0f;
            }
            WaveSync();
        }
        stamp(1);
        // ---- the next round's voice: its window leaves HBM now, beside this round's filters and mixing
        request(round + 1u);
        if(active)
        {
            if(head.flags & kFlagAmbiScale)
            {   // VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091 (through the slot: nothing is published yet)
                const AmbiScaleState a = L.ambi[v];
                SplitterState sp{a.coeff, a.lpZ1, a.lpZ2, a.apZ1};
#pragma unroll
                for(int j = 0; j < kRowsOuts; ++j) w.row[lane + 64u * uint32_t(j)] = outs[j];
                WaveSync();
                SplitterScan<false>(sp, w.row + outPos, N - outPos, a.hfScale, a.lfScale, lane);
                WaveSync();
#pragma unroll
                for(int j = 0; j < kRowsOuts; ++j) outs[j] = w.row[lane + 64u * uint32_t(j)];
                WaveSync();
                if(lane == 0) { L.ambi[v].lpZ1 = sp.lpZ1; L.ambi[v].lpZ2 = sp.lpZ2; L.ambi[v].apZ1 = sp.apZ1; }
            }
            counter = (head.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093
            directFilter = (head.flags & kFlagDirectFilter) != 0;
            // Everything that needs no slot of its own is resolved here, whatever the round's other voices have: the sends whose
            // filters are inactive ride on the unfiltered signal (their gains are merged into row0; the pair is cleared,
            // voice.cpp:264-265), an inactive direct pair is cleared.
            for(uint32_t si = 0; si < numSends; ++si)
            {
                const int32_t slot = slotOf(si);
                if(slot < 0) continue;
                if((head.flags >> (kFlagSendFilterShift + si)) & 1u) { order |= 1u << si; continue; }
                const uint32_t base = wetBase + uint32_t(slot) * wetCh;
                const bool mine = lane >= base && lane < base + wetCh;
                const size_t vs = size_t{v} * numSends + si;
                float tg = 0.0f, cu = 0.0f;
                float *curp = L.sendCur + vs * wetCh + (lane - base);
                const float tgS = sendGain(sTg, si, base), cuS = sendGain(sCu, si, base);
                if(mine)
                {
                    tg = playing ? tgS : 0.0f;
                    cu = counter ? cuS : tg;
                }
                const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                if(mine) { *curp = g.newCur; row0.add(g); }
                BiquadSlot *slots = &L.sfilt[vs * 2];
                if(!((restMask >> (1u + si)) & 1u))
                {
                    stagePair(1u + si);
                    WaveDoFilters(sm.jobs[wave].fst, slots, false, w.row, N, lane);
                }
            }
            if(!directFilter)
            {
                BiquadSlot *slots = &L.dfilt[size_t{v} * 2];
                if(!(restMask & 1u))
                {
                    stagePair(0u);
                    WaveDoFilters(sm.jobs[wave].fst, slots, false, w.row, N, lane);
                }
                // MixSamples onto the dry lines rides on the unfiltered signal (voice.cpp:962-963)
                float tg = 0.0f, cu = 0.0f;
                if(lane < numDry) { tg = playing ? dryTg : 0.0f; cu = counter ? dryCu : tg; }
                const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                if(lane < numDry) { L.gainCur[size_t{v} * numDry + lane] = g.newCur; row0.add(g); }
            }
            else order |= 2u << numSends;
            order |= 1u << numSends;
        }
        // ---- slot A: the unfiltered signal with the merged gains of everything that rides on it (a voice that feeds nothing
        // unfiltered publishes an all-zero block mask: its samples are there for the filter jobs to read)
        uint32_t pending = order & ~(1u << numSends);      // the signals that need DoFilters: they become JOBS
        if(active)
        {
#pragma unroll
            for(int j = 0; j < kRowsOuts; ++j) w.row[lane + 64u * uint32_t(j)] = outs[j];
            RowsPublishGains(w, row0, lane);
        }
        else if(lane == 0) w.live = 0u;
        if(lane == 0) sm.cnt[0][wave] = uint32_t(__builtin_popcount(pending));
        stamp(2);
        __syncthreads();
        stamp(3);

        // consume: this wavefront's 128-frame slice of up to eight rows, in index order.  The rows' words come in ONE read (lane =
        // index), every row's operands are requested before the first is used.  (Instruction-bound -- v_readlane, two FMAs per line and
        // row; the gain through the FMA's DPP operand instead, quad_perm:[i,i,i,i] on a register per 4-line block, measured slower: 9.6 K
        // against 7.3 K cycles, DPP operands run at half rate.)
        auto consume = [&](uint32_t mLive, uint32_t mBlocks, uint32_t mFade, auto rowOf, auto gainsOf)
        {
            float x0[kRowsWaves], x1[kRowsWaves], gv[kRowsWaves];
#pragma unroll
            for(int ww = 0; ww < kRowsWaves; ++ww)
            {
                const float *row = rowOf(ww);
                x0[ww] = row[uint32_t(kRowsSlice) * wave + lane]; x1[ww] = row[uint32_t(kRowsSlice) * wave + 64u + lane];
                gv[ww] = gainsOf(ww)[lane & 31u];
            }
#pragma unroll
            for(int ww = 0; ww < kRowsWaves; ++ww)
            {
                if(__builtin_amdgcn_readlane(int(mLive), ww) == 0) continue;
                const uint32_t blocks = uint32_t(__builtin_amdgcn_readlane(int(mBlocks), ww)), maxFade = uint32_t(__builtin_amdgcn_readlane(int(mFade), ww));
#pragma unroll
                for(int b = 0; b < kRowsLines / 4; ++b)
                {
                    if(!((blocks >> b) & 1u)) continue;
#pragma unroll
                    for(int i = 0; i < 4; ++i)
                    {
                        const float g = RowsReadLaneF(gv[ww], 4 * b + i);
                        acc[4 * b + i] = pkfma(f2{x0[ww], x1[ww]}, splat(g), acc[4 * b + i]);
                    }
                }
                if(maxFade != 0u && wave == 0u)
                {   // MixLine's ramp (Counter <= 64 frames, voice.cpp:1093): the first slice's first 64 frames, frame = lane
                    const float *gp = gainsOf(ww);
                    const float av = gp[kRowsLines + (lane & 31u)], bv = gp[2 * kRowsLines + (lane & 31u)];
                    const float xr = (lane < maxFade) ? x0[ww] : 0.0f, fl = float(lane);
#pragma unroll
                    for(int b = 0; b < kRowsLines / 4; ++b)
                    {
                        if(!((blocks >> b) & 1u)) continue;
#pragma unroll
                        for(int i = 0; i < 4; ++i)
                        {
                            const float a = RowsReadLaneF(av, 4 * b + i), bb = RowsReadLaneF(bv, 4 * b + i);
                            acc[4 * b + i].x = __builtin_fmaf(xr, __builtin_fmaf(bb, fl, a), acc[4 * b + i].x);
                        }
                    }
                }
            }
        };
        {
            const uint32_t mLive = lane < uint32_t(kRowsWaves) ? sm.w[lane].live : 0u;
            const uint32_t mBlocks = lane < uint32_t(kRowsWaves) ? sm.w[lane].blocks : 0u;
            const uint32_t mFade = lane < uint32_t(kRowsWaves) ? sm.w[lane].maxFade : 0u;
            consume(mLive, mBlocks, mFade, [&](int ww) { return (const float*)sm.w[ww].row; }, [&](int ww) { return (const float*)sm.w[ww].gain; });
        }
        stamp(4);

        // ================= the round's filter jobs, eight at a time =================
        // A signal that needs DoFilters -- the direct path's, a send's own copy (voice.cpp:255-267, :966-983) -- is a JOB: its owner
        // posts the pair's state, the resolved gains of the lines it feeds and where its samples are; wavefront j of the workgroup
        // EXECUTES job j -- copies the owner's slot-A row into its own second row (the resampler's window area, idle until the next
        // round parks), scans it, stores the pair's state -- and everybody consumes the results in job order.  The round's filters
        // are spread over all eight wavefronts whichever voices they belong to: a round costs ceil(jobs / 8) scans, not as many as
        // its busiest voice has filtered signals.  Job order = (owner, kind) through a prefix over the owners' counts: deterministic.
        for(uint32_t batch = 0;; ++batch)
        {
            const uint32_t cw = lane < uint32_t(kRowsWaves) ? sm.cnt[batch & 1u][lane] : 0u;
            uint32_t total = 0u, myStart = 0u;
#pragma unroll
            for(int ww = 0; ww < kRowsWaves; ++ww)
            {
                const uint32_t c = uint32_t(__builtin_amdgcn_readlane(int(cw), ww));
                myStart += uint32_t(ww) < wave ? c : 0u;
                total += c;
            }
            if(total == 0u) break;
            uint32_t idx = myStart;
            for(uint32_t q = 0; q <= numSends && pending; ++q)
            {
                const uint32_t k = q == 0u ? numSends + 1u : q - 1u;        // the direct path's first, then the sends
                if(!((pending >> k) & 1u)) continue;
                if(idx < uint32_t(kRowsWaves))
                {
                    RowsJob &jb = sm.jobs[idx];
                    RowLineGain r;
                    BiquadSlot *slots;
                    uint32_t pair;
                    if(k < numSends)
                    {
                        const int32_t slot = slotOf(k);
                        const uint32_t base = wetBase + uint32_t(slot) * wetCh;
                        const bool mine = lane >= base && lane < base + wetCh;
                        const size_t vs = size_t{v} * numSends + k;
                        float tg = 0.0f, cu = 0.0f;
                        float *curp = L.sendCur + vs * wetCh + (lane - base);
                        const float tgS = sendGain(sTg, k, base), cuS = sendGain(sCu, k, base);
                        if(mine)
                        {
                            tg = playing ? tgS : 0.0f;
                            cu = counter ? cuS : tg;
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        if(mine) { *curp = g.newCur; r.add(g); }
                        slots = &L.sfilt[vs * 2];
                        pair = 1u + k;
                    }
                    else
                    {
                        float tg = 0.0f, cu = 0.0f;
                        if(lane < numDry) { tg = playing ? dryTg : 0.0f; cu = counter ? dryCu : tg; }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        if(lane < numDry) { L.gainCur[size_t{v} * numDry + lane] = g.newCur; r.add(g); }
                        slots = &L.dfilt[size_t{v} * 2];
                        pair = 0u;
                    }
                    float fv = 0.0f;
#pragma unroll
                    for(int h = 0; h < 4; ++h) { const float x = __shfl(flt[h], int(((pair & 1u) << 5) | (lane & 31u))); if((pair >> 1) == uint32_t(h)) fv = x; }
                    if(lane < 32u) jb.fst[lane] = fv;
                    RowsPublishGains(jb, r, lane);
                    if(lane == 0)
                    {
                        const uint64_t sp = uint64_t(uintptr_t(slots));
                        jb.src = wave; jb.outPos = outPos; jb.slotsLo = uint32_t(sp); jb.slotsHi = uint32_t(sp >> 32);
                    }
                    pending &= ~(1u << k);
                }
                ++idx;
            }
            if(lane == 0) sm.cnt[(batch + 1u) & 1u][wave] = uint32_t(__builtin_popcount(pending));
            __syncthreads();
            const uint32_t nJobs = total < uint32_t(kRowsWaves) ? total : uint32_t(kRowsWaves);
            if(wave < nJobs)
            {
                RowsJob &jb = sm.jobs[wave];
                const uint32_t src = uint32_t(__builtin_amdgcn_readfirstlane(int(jb.src))), op = uint32_t(__builtin_amdgcn_readfirstlane(int(jb.outPos)));
                BiquadSlot *slots = reinterpret_cast<BiquadSlot*>(uintptr_t(uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(jb.slotsLo))))
                    | (uint64_t(uint32_t(__builtin_amdgcn_readfirstlane(int(jb.slotsHi)))) << 32)));
                const float *from = sm.w[src].row;
                float xs[kRowsOuts];
#pragma unroll
                for(int j = 0; j < kRowsOuts; ++j) xs[j] = from[lane + 64u * uint32_t(j)];
#pragma unroll
                for(int j = 0; j < kRowsOuts; ++j) w.g.rd[lane + 64u * uint32_t(j)] = xs[j];
                WaveSync();
                WaveDoFilters(jb.fst, slots, true, w.g.rd + op, N - op, lane);
            }
            __syncthreads();
            {
                const uint32_t mLive = lane < nJobs ? sm.jobs[lane].live : 0u;
                const uint32_t mBlocks = lane < nJobs ? sm.jobs[lane].blocks : 0u;
                const uint32_t mFade = lane < nJobs ? sm.jobs[lane].maxFade : 0u;
                consume(mLive, mBlocks, mFade, [&](int ww) { return (const float*)sm.w[ww].g.rd; }, [&](int ww) { return (const float*)sm.jobs[ww].gain; });
            }
            __syncthreads();
        }
        stamp(5);

        // ================= state write-back (voice.cpp:1116-1232) =================
        if(active && lane == 0)
        {
            VoiceCtl &c = L.ctl[v];
            c.flags = ((head.flags | kFlagFading) & ~kFlagDelayed) & ~kFlagHrtfDirty;
            if(!playing) c.playState = OALGPU_VOICE_STOPPED;
            else
            {
                int32_t bufPosInt = head.position;
                uint32_t bufPosFrac = head.positionFrac + head.step * (N - outPos);
                const uint32_t samplesDone = bufPosFrac >> kFracBits;
                bufPosInt = AddSat(bufPosInt, int32_t(samplesDone));
                bufPosFrac &= kFracMask;
                if(bufferItem >= 0 && bufPosInt > 0 && (head.flags & kFlagQueue))
                {
                    uint32_t buffersDone = 0;
                    const int32_t before = bufferItem;
                    AdvanceQueue(L.buffers, bufferItem, head.loopBuffer, bufPosInt, buffersDone);
                    if(buffersDone) L.queueDone[v] += buffersDone;
                    if(bufferItem >= 0 && bufferItem != before) c.buf = L.buffers[bufferItem];
                }
                else if(bufferItem >= 0 && bufPosInt > 0)
                {
                    if(looping)
                    {
                        uint32_t pos = uint32_t(bufPosInt);
                        if(pos >= buf.loopEnd)
                        {
                            pos = ((pos - buf.loopStart) % (buf.loopEnd - buf.loopStart)) + buf.loopStart;
                            bufPosInt = int32_t(pos);
                        }
                    }
                    else if(uint32_t(bufPosInt) >= buf.sampleLen)
                        bufferItem = -1;
                }
                c.position = bufPosInt;
                c.positionFrac = bufPosFrac;
                c.curBuffer = bufferItem;
                if(bufferItem < 0)
                {
                    c.loopBuffer = -1;
                    c.playState = OALGPU_VOICE_STOPPING;
                }
            }
        }
        // the next update's parameter block: the wavefront installs the record of the voice it has just produced
        if(next.map && haveVoice)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const int32_t ri = __builtin_amdgcn_readfirstlane(next.map[v]);
            if(ri >= 0) ApplyNextRecord(L, next.recs[ri], lane);
        }
        stamp(6);
        __syncthreads();                            // (the slots, the job records and the counts are the next round's)
        stamp(7);
    }

    // ---- the wavefront's slice of the workgroup's partial bus
    {
        float *pl = L.partLines + size_t{group} * L.mixLines * kLine + uint32_t(kRowsSlice) * wave + lane;
#pragma unroll
        for(int c = 0; c < kRowsLines; ++c)
        {   // (no early exit: the loop must unroll, or the accumulators are indexed at run time and live in scratch)
            if(uint32_t(c) < L.mixLines)
            {
                StorePartial(pl + size_t(c) * kLine, acc[c].x);
                StorePartial(pl + size_t(c) * kLine + 64, acc[c].y);
            }
        }
    }
}

} // namespace

// which layouts: dry lines AND sends (or more lines than the wavefront-per-voice kernel holds in registers), no near-field control
bool RowsKernelApplies(const DeviceLayout &L)
{
    return !L.hrtf && L.numDry >= 1 && L.mixLines >= 1 && L.mixLines <= uint32_t(kRowsLines) && L.numSends <= 6 && L.nfc == nullptr
        && L.numDry + L.numSlots * L.wetChannels <= uint32_t(kRowsLines) && L.wetChannels <= 32u && L.numSends * L.wetChannels <= 128u;
}
const char *RowsKernelName() { return "VoiceRowsKernel"; }
uint32_t RowsWavesPerGroup() { return uint32_t(kRowsWaves); }

hipError_t LaunchVoiceRows(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const WaveProf *prof, hipEvent_t evStart, hipEvent_t evStop,
    const ParamRecord *nextRecs, const int32_t *nextMap)
{
    const NextBlock next{nextRecs, nextMap, nullptr, ResidentArgs{}};
    const WaveProf none{nullptr, 0u};
    if(prof) hipExtLaunchKernelGGL(VoiceRowsKernel<true>, dim3(L.numGroups), dim3(kRowsThreads), 0, s, evStart, evStop, 0u, L, samplesToDo, L.rowsVpg, next, *prof);
    else hipExtLaunchKernelGGL(VoiceRowsKernel<false>, dim3(L.numGroups), dim3(kRowsThreads), 0, s, evStart, evStop, 0u, L, samplesToDo, L.rowsVpg, next, none);
    return hipGetLastError();
}

} // namespace oalgpu
