// The FAST HRTF hot path at FOUR wavefronts per SIMD: one voice per wavefront, sixteen wavefronts per workgroup, one
// workgroup per compute unit (OALGPU_CTX_WAVE16; BASELINE configs[2]).
//
//   "for every Playing|Stopping voice: voice->mix(...)"   alc/alu.cpp:2201-2206
//      -> Voice::mix                                       core/voice.cpp:988-1233
//      -> LoadResampledSamples / Resample_*                voice.cpp:642-824, mixer_sse.cpp:199-329
//      -> DoFilters (dual biquad)                          voice.cpp:255-267
//      -> DoHrtfMix -> MixHrtf / MixHrtfBlend              voice.cpp:827-902, mixer/hrtfbase.h:17-89
//
// Why this shape.  VoiceWaveKernel (voice_wave.hip) keeps two voices per wavefront and two wavefronts per SIMD (217
// registers: two complete resampler outputs in flight, the accumulator tiles live across both voices, the next voice's
// window requested over the FIR).  Its counters name no saturated unit: a wavefront issues one instruction per ~10
// cycles, two of them cannot keep a SIMD's issue ports busy, and the phases in which they could (all eight wavefronts of
// a CU resampling: the LDS pipe) they spend in step.  Here a wavefront owns ONE voice, so
//   * nothing is carried from voice to voice: the accumulator tiles live from the FIR to the dump only, no request-ahead
//     registers, no parking -- the peaks are the ring resampler (two sets of 6 tap pairs) and the FIR's fragments: <= 112
//     registers, four wavefronts per SIMD, with the 64 registers the post-stream kernels need left over;
//   * the resampler's outputs stay in registers (16 per lane, output = lane + 64 j) -- there is no LDS copy of the sample
//     line: the FIR's inputs (x', split half precision, one ear at a time) are built straight from them, so the
//     wavefront's LDS is the window (parked twice, aligned pair reads) OR one ear's inputs OR the dump: 9.4 KB, sixteen
//     of them + the workgroup's resampler rows + the post-process's 5 KB on one CU;
//   * 4096 voices are exactly one wavefront per voice slot of the machine (256 CUs x 16): one round, one partial bus per
//     CU (256 instead of 512).
// Voices the register path does not cover (a window that does not fit, queues, delayed starts, other formats and
// resamplers) take voice_wave.hip's generic loader (LoadResampledWave) into an LDS sample line and join the common path
// from there.  Arithmetic is FAST mode; all integer state is bit-exact (the same statements as voice_wave.hip).
#define OALGPU_WAVE_NO_LAUNCHER
#include "voice_wave.hip"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kW16Outs = kLine / 64;                    // outputs per lane: output k = lane + 64 j

// ---- the wavefront's LDS: four views of one area of 2352 dwords ----
constexpr int kW16Rd = 1184;                            // fast path: the window (48 + 17 x 64 + 8 -> 32 mod 64: rd2 half the banks away)
constexpr int kW16Rd2 = kMaxEdge + kPre * 64 + 8;       // the same window one sample on
constexpr int kW16Area = kResampleDataSize + kLine;     // generic path: DeviceBase::mResampleData + the sample line
static_assert(kW16Rd % 64 == 32 && kW16Rd >= kMaxPad + kPre * 64 + 8, "rd2 sits half the banks away from rd");
static_assert(kW16Rd + kW16Rd2 <= kW16Area, "the twice-parked window fits");
static_assert(kResampleDataSize >= kLine + 64, "a send's filter scan parks the unfiltered samples in the window's area");
constexpr int kW16DumpF2 = 64 * 17 + 68;                // FirMfmaH's tiles as frames, 17 entries per 16 frames
static_assert(kW16DumpF2 * 2 <= kW16Area, "the dump fits");
struct W16PhaseB {
    uint32_t xh[2][kXhDw];                              // ONE ear's FIR inputs [hi | lo], two frames per dword (FirMfmaH)
    uint32_t hr[2][kHrDw];                              // that ear's reversed response [hi | lo]
    uint32_t hro[2][kHrDw];                             // a replaced filter's old response
    float in0[2 * kHist];                               // [Hrtf.History | the first 64 samples]: what the delayed inputs' head reads
};
static_assert(sizeof(W16PhaseB) <= size_t(kW16Area) * 4, "phase B fits the area (nothing of it is written before the sample line is back in registers)");
struct alignas(16) W16Lds {
    union {
        struct { float rd[kW16Rd]; float rd2[kW16Rd2]; } a;                    // register path: the window, twice
        struct { float rd[kResampleDataSize]; float smp[kLine]; } g;           // generic loader: window + sample line
        W16PhaseB b;
        f2 dump[kW16DumpF2];
        float raw[kW16Area];
    };
    int32_t best;
    uint32_t pad[3];
};
// (the rows FIRST: a DS instruction's immediate offset has 16 bits, and with the rows behind 150 KB of wavefront areas every one
// of the resampler's 24 row reads per output needed a VALU operation of its own to build its address)
template<int WAVES>
struct W16Wg {
    alignas(16) f2 tabF[12 * 32];                       // [tap pair][phase] = fil[2p], fil[2p+1]   (up to 24 taps)
    f2 tabP[12 * 32];
    uint32_t tabKey, tabM, tabL, pad;
    uint32_t res[12];                                   // RES: what thread 0 learned at the door, for the workgroup
    W16Lds w[WAVES];
};
static_assert(sizeof(W16Wg<16>) <= 157440, "123 LDS granules: the post-process (4) and the reduction (1) fit beside it");

// what LoadResampledWave (wave_common.hpp) reaches through its wavefront-LDS argument
struct W16GenView { float *rd, *in, *rd2; int32_t &best; uint32_t *pad; };

// ---- resampler: ResampleRunRing (wave_common.hpp) with the outputs left in registers ----
// Outputs lane, lane + 64, ...: outs[j] = output lane + 64 j.  Two register sets of NP tap pairs; while one group is
// multiplied the next group's 3 NP reads are in flight (four wavefronts per SIMD cover the rest of the latency).
template<int M, bool DUAL>
__device__ __forceinline__ void ResampleRingRegs(const f2 *tabF, const f2 *tabP, const float *rdb, uint32_t frac0, uint32_t increment,
    uint32_t bdst, float (&outs)[kW16Outs], uint32_t lane, const float *rd2b, uint32_t rdbIndex)
{
    constexpr int NP = (M / 2 >= 6) ? 6 : M / 2;
    constexpr int G = (M / 2) / NP;               // 1 (cubic, bsinc12), 2 (bsinc24)
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
    for(int j = 0; j < kW16Outs; ++j)
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
                if(nx < kW16Outs * G) load(nx & 1, tb + uint32_t(nx / G) * tstep, nx % G);
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
    }
}

// ---- the dual-ear FIR on the matrix pipe, ONE ear: FirMfmaH (dev_wave.hpp) without the ear loop ----
// (fragments are fetched when they are needed: the other three wavefronts of the SIMD cover the LDS latency)
template<int TILES, bool LOW8>
__device__ __forceinline__ void FirMfmaEar(f4 (&acc)[5], const uint32_t (&xh)[2][kXhDw], const uint32_t (&hr)[2][kHrDw], float inv, uint32_t lane)
{
    const uint32_t i = lane & 15u, g = lane >> 4;
    const uint32_t u0 = 16u - i + 8u * g, shift = (u0 & 1u) * 2u, ud = u0 >> 1;
    const f4 vinv = {inv, inv, inv, inv};
    h8 A[2][3];
#pragma unroll
    for(int s = 0; s < 2; ++s)
#pragma unroll
        for(int c = 0; c < 3; ++c)
        {
            const uint32_t *p = hr[s] + ud + 16 * c;
            const uint32_t q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
            const u4 q = {__builtin_amdgcn_alignbyte(q1, q0, shift), __builtin_amdgcn_alignbyte(q2, q1, shift),
                __builtin_amdgcn_alignbyte(q3, q2, shift), __builtin_amdgcn_alignbyte(q4, q3, shift)};
            A[s][c] = __builtin_bit_cast(h8, q);
        }
#pragma unroll
    for(int T = 0; T < TILES; ++T)
    {
        __builtin_amdgcn_sched_barrier(0);                       // (a tile at a time: the scheduler must not hold five tiles' fragments at once)
        const uint32_t j = (T == 4 && i > 3u) ? 3u : i;          // the ring-out tile has four columns
        h8 B[2][3];
#pragma unroll
        for(int s = 0; s < 2; ++s)
        {
            const u4 *p = reinterpret_cast<const u4*>(xh[s]) + 32 * T + 2u * j + g;
#pragma unroll
            for(int c = 0; c < 3; ++c) B[s][c] = __builtin_bit_cast(h8, p[4 * c]);
        }
        f4 ta = {0.0f, 0.0f, 0.0f, 0.0f}, tb = ta, tc = ta;
#pragma unroll
        for(int c = 0; c < 3; ++c)
        {
            ta = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[1][c], B[0][c], ta, 0, 0, 0);
            tb = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][c], B[1][c], tb, 0, 0, 0);
            tc = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[0][c], B[0][c], tc, 0, 0, 0);
        }
        f4 sum = __builtin_elementwise_fma((ta + tb) + tc, vinv, acc[T]);
        asm volatile("" : "+v"(sum));                            // (summed HERE: left to itself the compiler sinks the three partial tiles' sum to the dump)
        if(!LOW8 || i < 8u) acc[T] = sum;
        __builtin_amdgcn_sched_barrier(0);                       // (nor three partial tiles per tile until the end)
    }
}

__device__ __forceinline__ float W16ReadLaneF(float v, int l)
{ return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

// a response (tap = lane, one ear) as split halves r[80 - lane] in hr[hi | lo] (FirMfmaH's operand A); returns 1 / scale
__device__ __forceinline__ float W16StageResponse(uint32_t (&hr)[2][kHrDw], float h, uint32_t lane)
{
    float sh, inv;
    HalfScale(WaveMaxBits(__builtin_bit_cast(uint32_t, __builtin_fabsf(h))), sh, inv);
    uint32_t hi, lo;
    SplitHalf2(h * sh, 0.0f, hi, lo);
    // zero padding: the halves outside u in [17, 80]
    uint32_t *hz32 = &hr[0][0];
    for(uint32_t k = lane; k < uint32_t(2 * kHrDw); k += 64) hz32[k] = 0u;
    uint16_t *hz = reinterpret_cast<uint16_t*>(&hr[0][0]);
    const uint32_t u = 80u - lane;
    hz[0 * kHrHalves + u] = uint16_t(hi); hz[1 * kHrHalves + u] = uint16_t(lo);
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, inv)));
}

// The voices of a workgroup's sixteen wavefronts.  Wavefronts w, w + 4, w + 8, w + 12 share a SIMD; voices that cost more --
// an active filter, a replaced HRIR -- tend to come in regular patterns (every n-th source of a scene), so the slot within
// a group of four rotates with the group: a period-4 pattern puts one voice of each kind on every SIMD.
__device__ __forceinline__ uint32_t W16VoiceOf(uint32_t group, uint32_t wave, uint32_t waves)
{
    const uint32_t a = wave >> 2, b = wave & 3u;
    return group * waves + 4u * a + ((a + b) & 3u);
}

struct Next16 { const ParamRecord *recs; const int32_t *map; const float *rows; ResidentArgs res; };   // (res: the resident launch's, RES)
// SENDS: what the voice's auxiliary sends need of the context (voice.cpp:966-983).  The send's signal -- the resampled samples, through the
// send's own filter pair when that is active -- leaves as ONE 4 KB stream row per voice and send with the resolved gains of the slot's
// wet lines (voice_wave.hip's StoreRowBlock); StreamRowsMixKernel, launched behind this kernel, turns a workgroup's rows into its
// partial wet lines.  The HRTF path keeps its 110 registers: no line accumulator lives beside the FIR.
struct W16Sends {
    BiquadSlot *sfilt; const float *sendTgt; float *sendCur; float *streams; uint32_t *lineGains;
    uint32_t numSends, wetCh, lineStride, spv;
};

// PROF: the measurement variant (tools/phase_times16.py): s_memtime stamps per phase; the product variant carries none of it
// WAVES: wavefronts (= voices) per workgroup -- 16 where the scene fills the machine that way (one workgroup per compute unit),
// 8 or 4 for smaller scenes, so that every compute unit gets its share of them
// RES: the resident launch (OALGPU_CTX_RESIDENT; the protocol: kernels.hpp ResidentDoor, voice_wave.hip) -- the kernel's body is one
// UPDATE of a loop that ends when the host says so; everything a launch does per update it does per turn of that loop, in the same order
// with the same operations: the bits are those of one launch per update (tests/test_gpu_resident.py).
template<bool PROF, int WAVES, bool SENDS = false, bool RES = false>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(WAVES * 64) VoiceWave16Kernel(WaveArgsHrtf L, uint32_t samplesToDo, Next16 next, WaveProf prof, W16Sends S)
{
    unsigned long long tEntry = 0;
    if constexpr (PROF) tEntry = __builtin_readcyclecounter();
    constexpr uint32_t kW16Waves = uint32_t(WAVES), kW16Threads = uint32_t(WAVES) * 64u;
    __shared__ W16Wg<WAVES> sm;
    const uint32_t t = threadIdx.x;
    uint32_t lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t group = blockIdx.x;
    const uint32_t irStride = L.irStride;
    uint32_t N = samplesToDo;                   // (RES: every update brings its own length)
    W16Lds &w = sm.w[wave];
    asm volatile("; argument block resident" :: "s"(L.tables), "s"(L.buffers), "s"(L.ctl), "s"(L.prev), "s"(L.dfilt), "s"(L.hrtfOld),
        "s"(L.hrtfTgt), "s"(L.hist), "s"(L.ambi), "s"(L.startDelay), "s"(L.queueDone), "s"(L.partHrtf), "s"(L.hrirs),
        "s"(next.recs), "s"(next.map), "s"(L.numVoices), "s"(L.irStride), "s"(L.pad), "s"(samplesToDo));

    const uint32_t vRaw = W16VoiceOf(group, wave, kW16Waves);
    const bool haveVoice = vRaw < L.numVoices;
    const uint32_t lastVoice = L.numVoices - 1u;
    const uint32_t v = haveVoice ? vRaw : lastVoice;                 // (a wavefront without a voice reads a valid line and mixes nothing)
    const uint32_t keyVoice = group * uint32_t(kW16Waves);           // the voice whose resampler rows the workgroup stages

    // ---- RES: this launch's turn of the update loop (one pass through the kernel's body otherwise) ----
    uint32_t upd = 0u, updBase = 0u;
    if constexpr (RES)
    {
        const ResidentArgs &RA = next.res;
        upd = updBase = RA.base;
        if(t == 0)
        {   // the host launches nothing that waits for this kernel before every workgroup of it has a CU (see ResidentSubmit)
            const uint32_t old = __hip_atomic_fetch_add(RA.counters + 16u * kRcStarted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(old + 1u == RA.startedTarget)
                __hip_atomic_store(RA.hostFlags + 16u * kRhResident, RA.launchId, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    for(;;)
    {
    // RES: thread and lane index are re-derived per update behind an opaque move: what the body computes from them then is a few VALU
    // per update, not a set of loop invariants kept in registers across the resampler and the FIR, whose peaks decide the register count
    uint32_t tU = threadIdx.x;
    if constexpr (RES) { asm volatile("" : "+v"(tU)); lane = tU & 63u; }
    const uint32_t t = tU;
    if constexpr (RES)
    {
        const ResidentArgs &RA = next.res;
        // Thread 0 waits for the update: the doorbell (or the word to leave), and -- kResidentSets updates on -- for the reduction
        // that still reads the partial set this update's sums go into (voice_wave.hip's loop, statement for statement).
        if(t == 0)
        {
            const unsigned long long tTop = __builtin_amdgcn_s_memrealtime();
            uint32_t go = 0u, fault = 0u;
            unsigned long long recs = 0ull, map = 0ull, rowsP = 0ull;
            uint32_t smp = 0u;
            if(upd != RA.endSeq)
            {
                const ResidentDoor *door = RA.door;
                const uint32_t *exitp = &door->exitSeq[RA.launchId & 3u];
                const uint32_t *rr = RA.counters + 16u * kRcRedRead;
                const bool needRed = int32_t(upd - kResidentSets) >= 0;
                const uint32_t want = (upd - kResidentSets + 1u) * RA.redPerUpdate;
                unsigned long long t0 = 0ull;
                uint32_t waitedFor = 0u;                     // 1: the doorbell, 2: the reduction
                for(uint32_t spins = 0;; ++spins)
                {
                    const uint32_t ex = ResLoadSys(exitp), sq = ResLoadSys(&door->seq), rd = needRed ? ResLoadDev(rr) : want;
                    if(int32_t(ex - upd) <= 0) break;
                    const bool rung = int32_t(sq - upd) > 0, redOk = int32_t(rd - want) >= 0;
                    if(rung && redOk) { go = 1u; break; }
                    if(spins == 0u) { t0 = __builtin_amdgcn_s_memrealtime(); waitedFor = rung ? 2u : 1u; }
                    else if((spins & 15u) == 15u && __builtin_amdgcn_s_memrealtime() - t0 > kResidentWatchdogTicks)
                    {
                        fault = 1u;
                        uint32_t *fi = RA.hostFlags + 16u * kRhFault;
                        fi[0] = upd; fi[1] = sq; fi[2] = ex; fi[3] = rd; fi[4] = want; fi[5] = group; fi[6] = RA.launchId;
                        fi[7] = uint32_t(__builtin_amdgcn_s_memrealtime() - t0);
                        break;
                    }
                    if(spins < 32u) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(24);
                }
                if(waitedFor && (group & 127u) == 0u)
                    __hip_atomic_fetch_add(RA.counters + 16u * (waitedFor == 1u ? kRcWaitDoor : kRcWaitRed), uint32_t(__builtin_amdgcn_s_memrealtime() - t0),
                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if(go)
                {
                    const ResidentSlot *sl = &door->slot[upd % kResidentSlots];
                    recs = ResLoadSys(&sl->recs); map = ResLoadSys(&sl->map); rowsP = ResLoadSys(&sl->rows); smp = ResLoadSys(&sl->samples);
                }
            }
            if(fault) __hip_atomic_store(RA.hostFlags + 16u * kRhError, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if((group & 127u) == 0u)
                __hip_atomic_fetch_add(RA.counters + 16u * kRcTop, uint32_t(__builtin_amdgcn_s_memrealtime() - tTop), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sm.res[0] = go; sm.res[1] = smp;
            sm.res[2] = uint32_t(recs); sm.res[3] = uint32_t(recs >> 32); sm.res[4] = uint32_t(map); sm.res[5] = uint32_t(map >> 32);
            sm.res[8] = uint32_t(rowsP); sm.res[9] = uint32_t(rowsP >> 32);
        }
        __syncthreads();
        const unsigned long long tSeen = __builtin_amdgcn_s_memrealtime();
        if(__builtin_amdgcn_readfirstlane(sm.res[0]) == 0u) break;
        N = __builtin_amdgcn_readfirstlane(sm.res[1]);
        const unsigned long long recsU = (uint64_t{uint32_t(__builtin_amdgcn_readfirstlane(sm.res[3]))} << 32) | uint32_t(__builtin_amdgcn_readfirstlane(sm.res[2]));
        const unsigned long long mapU = (uint64_t{uint32_t(__builtin_amdgcn_readfirstlane(sm.res[5]))} << 32) | uint32_t(__builtin_amdgcn_readfirstlane(sm.res[4]));
        // the update's parameter block: every wavefront installs the record of its own voice, as ApplyParamsKernel would in front of a
        // launch (the voice's state of the update before was written back by this very wavefront, in program order)
        if(mapU && haveVoice)
        {
            const unsigned long long rowsU = (uint64_t{uint32_t(__builtin_amdgcn_readfirstlane(sm.res[9]))} << 32) | uint32_t(__builtin_amdgcn_readfirstlane(sm.res[8]));
            InstallPair(L, reinterpret_cast<const int32_t*>(mapU), reinterpret_cast<const ParamRecord*>(recsU), reinterpret_cast<const float*>(rowsU), v, v, false, lane);
        }
        // What the update reads through the scalar cache -- the voices' control lines -- was written with vector stores, by this
        // wavefront and (the key voice's head) by the key voice's: the stores are in L2 before the barrier, the scalar cache forgets behind it.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_s_dcache_inv();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if(t == 0)
        {
            const unsigned long long tIn = __builtin_amdgcn_s_memrealtime();
            if((group & 127u) == 0u)
                __hip_atomic_fetch_add(RA.counters + 16u * kRcInstall, uint32_t(tIn - tSeen), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sm.res[6] = uint32_t(tIn);
        }
    }
    auto stamp = [&](int slot)
    {
        if constexpr (PROF) { if(prof.times && haveVoice && (t & 63u) == 0u) prof.times[size_t{v} * 8 + slot] = __builtin_readcyclecounter(); }
    };
    if constexpr (PROF) { if(prof.times && haveVoice && (t & 63u) == 0u) prof.times[size_t{v} * 8 + 0] = tEntry; }
    // (eight more stamps per voice behind the first block: the inside of the first ear's pass)
    auto stamp2 = [&](int slot)
    {
        if constexpr (PROF) { if(prof.times && haveVoice && (t & 63u) == 0u) prof.times[size_t{L.numVoices} * 8 + size_t{v} * 8 + slot] = __builtin_readcyclecounter(); }
    };
    const VoiceHead head = LoadHeadScalar(L.ctl + v);
    const BufferItem buf = LoadCtlBufferScalar(L.ctl + v);
    const VoiceTail tail = LoadTailScalar(L.ctl + v);
    asm volatile("" ::: "memory");
    const VoiceHead headK = LoadHeadScalar(L.ctl + (keyVoice < L.numVoices ? keyVoice : lastVoice));
    asm volatile("" ::: "memory");
    const int psK = headK.playState, kK = headK.rsKind;
    const uint32_t mK = kK == 2 ? 4u : headK.rsM, lK = kK == 2 ? 1u : headK.rsL, offK = headK.rsFilterOffset;

    // ---- the workgroup's resampler rows: LDS-DMA gathers, a tap pair per wavefront (voice_wave.hip's pass 0) ----
    const bool eligK = keyVoice < L.numVoices && (kK == 2 || (kK == 3 && (mK == 12 || mK == 24)))
        && (psK == OALGPU_VOICE_PLAYING || psK == OALGPU_VOICE_STOPPING);
    // (RES: the rows staged by an earlier update of this launch are still there unless the key voice's resampler changed)
    bool stageRows = true;
    if constexpr (RES) { if(upd != updBase) stageRows = !(eligK && sm.tabKey == offK * 8u + uint32_t(kK) && sm.tabM == mK); }
    if constexpr (RES) __syncthreads();         // (everybody has compared before thread 0 rewrites the key)
    if(eligK && stageRows)
    {
        typedef const __attribute__((address_space(1))) void *gvoidp;
        typedef __attribute__((address_space(3))) void *lvoidp;
        const float *src = L.tables + offK + size_t{lane >> 1} * (2u * mK) + (lane & 1u);
        for(uint32_t pp = wave; pp < mK / 2u; pp += uint32_t(kW16Waves))
        {
            __builtin_amdgcn_global_load_lds((gvoidp)(src + 2u * pp), (lvoidp)&sm.tabF[32u * pp], 4, 0, 0);
            __builtin_amdgcn_global_load_lds((gvoidp)(src + mK + 2u * pp), (lvoidp)&sm.tabP[32u * pp], 4, 0, 0);
        }
        if(t == 0) { sm.tabKey = offK * 8u + uint32_t(kK); sm.tabM = mK; sm.tabL = lK; }
    }

    // ---- what the voice starts from: the source window (register path), mPrevSamples, Hrtf.History, the direct filter
    // pair, the target response (tap = lane), a replaced filter's old one ----
    const int vstate = head.playState;
    const bool mixes = haveVoice && (vstate == OALGPU_VOICE_PLAYING || vstate == OALGPU_VOICE_STOPPING);
    const bool playing = haveVoice && vstate == OALGPU_VOICE_PLAYING;
    bool active = mixes && head.step >= 1u;
    const bool dirty = (head.flags & kFlagHrtfDirty) != 0;
    SrcPlan plan = PlanSource(head, N);
    bool looping = false;
    if(head.curBuffer >= 0)
    {   // voice.cpp:1015-1019: a position at or past the loop end plays on without looping
        looping = head.loopBuffer >= 0 && ((head.flags & kFlagQueue) || !(head.position >= 0 && uint32_t(head.position) >= buf.loopEnd));
        plan.prefetch = plan.prefetch && GatherCovers(plan.bsrc, buf, looping, uint32_t(head.position));
    }
    if(head.flags & (kFlagDelayed | kFlagQueue)) plan.prefetch = false;
    // the register path: a prefetched window resampled from the workgroup's staged rows with two sets of 6 tap pairs
    const uint32_t sM = head.rsKind == 2 ? 4u : head.rsM, sL = head.rsKind == 2 ? 1u : head.rsL;
    const bool regPath = active && plan.prefetch && eligK && (head.rsKind == 2 || head.rsKind == 3)
        && head.rsFilterOffset * 8u + uint32_t(head.rsKind) == offK * 8u + uint32_t(kK) && sM == mK
        && !(head.step == kFracOne && head.positionFrac == 0u);
    float pre[kPre];
#pragma unroll
    for(int i = 0; i < kPre; ++i) pre[i] = 0.0f;
    float prevv = 0.0f;
    if(regPath)
    {
        GatherStatic(pre, plan.bsrc, buf, looping, uint32_t(head.position), lane);
        prevv = (lane < kMaxPad) ? L.prev[size_t{v} * kMaxPad + lane] : 0.0f;
    }

    // the key voice does not qualify: wavefront 0 looks for one that does (voice_wave.hip's prologue)
    if(!eligK)
    {
        if(wave == 0)
        {
            const uint32_t cand = keyVoice + lane;
            bool eligible = false;
            uint32_t off = 0, m = 0, l = 0;
            int kind = 0;
            if(lane < uint32_t(kW16Waves) && cand < L.numVoices)
            {
                const VoiceCtl &c = L.ctl[cand];
                kind = c.rsKind; off = c.rsFilterOffset; m = c.rsM; l = c.rsL;
                if(kind == 2) { m = 4; l = 1; }
                eligible = (kind == 2 || (kind == 3 && (m == 12 || m == 24)))
                    && (c.playState == OALGPU_VOICE_PLAYING || c.playState == OALGPU_VOICE_STOPPING);
            }
            const unsigned long long mask = __ballot(eligible);
            if(mask)
            {
                const int firstLane = __ffsll((long long)mask) - 1;
                const uint32_t key = uint32_t(__shfl(int(off * 8u + uint32_t(kind)), firstLane));
                const uint32_t fm = uint32_t(__shfl(int(m), firstLane)), fl = uint32_t(__shfl(int(l), firstLane));
                if(lane == 0) { sm.tabKey = key; sm.tabM = fm; sm.tabL = fl; }
            }
            else if(lane == 0) { sm.tabKey = 0xffffffffu; sm.tabM = 0; sm.tabL = 0; }
        }
        __syncthreads();
        const uint32_t key = sm.tabKey, m = sm.tabM;
        if(key != 0xffffffffu)
        {
            const float *filter = L.tables + (key >> 3);
            for(uint32_t idx = t; idx < (m / 2u) * 32u; idx += uint32_t(kW16Threads))
            {
                const uint32_t p = idx >> 5, pi = idx & 31u;
                const float *row = filter + pi * 2u * m;
                sm.tabF[idx] = f2{row[2u * p], row[2u * p + 1u]};
                sm.tabP[idx] = f2{row[m + 2u * p], row[m + 2u * p + 1u]};
            }
        }
    }
    // the rows (LDS-DMA) and this wavefront's window are in; the window is parked twice (rd2[i] = rd[i + 1])
    if(regPath)
    {
        const bool isShort = buf.fmt == OALGPU_FMT_SHORT;
        if(lane < kMaxPad) { w.a.rd[lane] = prevv; if(lane) w.a.rd2[lane - 1u] = prevv; }
#pragma unroll
        for(int i = 0; i < kPre; ++i)
        {
            const float sv = GatherDecode(pre[i], isShort);
            w.a.rd[kMaxEdge + lane + 64u * uint32_t(i)] = sv;
            w.a.rd2[kMaxEdge - 1u + lane + 64u * uint32_t(i)] = sv;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(7);
    // Instruction arbitration between the four wavefronts of a SIMD goes by priority, then by age: left alone the oldest runs
    // through its phases first and the youngest runs its last ones alone, at a quarter of the issue rate (measured: the
    // resampler took 11 K cycles for a workgroup's first wavefront and 38 K for its last).  A wavefront therefore gives
    // priority away as it advances: whoever is behind is served first.  (Level 3 is left to the post-stream kernels that run beside
    // this one: the reduction and the post-process of the update before are short chains the step waits for.)
    __builtin_amdgcn_s_setprio(2);

    // ---- voice.cpp:1002-1046: what does not mix; delayed starts ----
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

    f4 accM[2][5];
#pragma unroll
    for(int e = 0; e < 2; ++e)
#pragma unroll
        for(int b = 0; b < 5; ++b) accM[e][b] = f4{0.0f, 0.0f, 0.0f, 0.0f};

    uint32_t counter = 0, fademix = 0, todo = 0;
    float endGain = 0.0f, gainAfterBlend = 0.0f;
    float fstv = 0.0f, histv = 0.0f;
    f2 hT = {0.0f, 0.0f};
    int32_t bufferItem = head.curBuffer;
    if constexpr (SENDS)
    {   // nothing to mix for this voice in this update: the rows it left last time are dead (voice_wave.hip does the same)
        if(haveVoice && !active && lane < S.spv)
            S.lineGains[(size_t{v} * S.spv + lane) * LineBlockDwords(S.lineStride) + 3u * S.lineStride] = 0u;
    }
    if(active)
    {
        float outs[kW16Outs];
        // ---------------- LoadResampledSamples ----------------
        if(regPath)
        {
            WaveSync();
            const float *rdb = w.a.rd + (kMaxEdge - sL), *rd2b = w.a.rd2 + (kMaxEdge - sL);
            const uint32_t rdbIndex = uint32_t(kMaxEdge) - sL;
            if(sM == 24u) ResampleRingRegs<24, true>(sm.tabF, sm.tabP, rdb, head.positionFrac, head.step, N, outs, lane, rd2b, rdbIndex);
            else if(sM == 12u) ResampleRingRegs<12, true>(sm.tabF, sm.tabP, rdb, head.positionFrac, head.step, N, outs, lane, rd2b, rdbIndex);
            else ResampleRingRegs<4, false>(sm.tabF, sm.tabP, rdb, head.positionFrac, head.step, N, outs, lane, nullptr, 0u);
            asm volatile("" : "+v"(lane));
            // voice.cpp:772-785: history for the next update, taken at the end-of-mix position
            if(playing)
            {
                const uint32_t srcOffset = uint32_t((uint64_t{N} * head.step + head.positionFrac) >> kFracBits);
                if(lane < kMaxPad) L.prev[size_t{v} * kMaxPad + lane] = w.a.rd[srcOffset + lane];
            }
        }
        else
        {
            W16GenView gv{w.g.rd, w.g.smp - kHist, nullptr, w.best, w.pad};
            LoadResampledWave<true, false>(sm, gv, L, v, lane, head, playing, N - outPos, N - outPos, bufferItem, looping,
                SrcPlan{false, plan.bdst, plan.bsrc}, outPos);
            asm volatile("" : "+v"(lane));
            WaveSync();
#pragma unroll
            for(int j = 0; j < kW16Outs; ++j) outs[j] = w.g.smp[lane + 64u * uint32_t(j)];
        }
        if(N < uint32_t(kLine))
        {   // (a short block: outputs from N on are whatever the clamped reads produced)
#pragma unroll
            for(int j = 0; j < kW16Outs; ++j) if(lane + 64u * uint32_t(j) >= N) outs[j] = 0.0f;
        }
        stamp(1);
        __builtin_amdgcn_s_setprio(1);
        // the direct filter pair, Hrtf.History and the target response (tap = lane): requested now, behind the resampler --
        // held across it they would be four registers at the kernel's register peak; the SIMD's other wavefronts cover the trip
        fstv = (lane < 32u) ? reinterpret_cast<const float*>(L.dfilt + size_t{v} * 2)[lane] : 0.0f;
        histv = L.hist[size_t{v} * kHist + lane];
        hT = (lane < irStride) ? reinterpret_cast<const f2*>(L.hrtfTgt + size_t{v} * irStride * 2)[lane] : f2{0.0f, 0.0f};
        WaveSync();                                 // (the window is dead from here on: its area is the sample line's / phase B's)

        auto toLine = [&]()
        {
#pragma unroll
            for(int j = 0; j < kW16Outs; ++j) w.g.smp[lane + 64u * uint32_t(j)] = outs[j];
            WaveSync();
        };
        auto fromLine = [&]()
        {
            WaveSync();
#pragma unroll
            for(int j = 0; j < kW16Outs; ++j) outs[j] = w.g.smp[lane + 64u * uint32_t(j)];
            WaveSync();
        };
        // DoFilters (voice.cpp:255-267) for one filter pair whose 2 x 16 words are held lane-wise in `fv` (the states go through
        // v_readlane into scalar registers: read back from LDS they were 26 vector registers at the scan's peak), over the sample line
        auto doPair = [&](float fv, BiquadSlot *slots, bool on)
        {
            BiquadState f0, f1;
            {
                float a[13], b[13];
#pragma unroll
                for(int k = 0; k < 13; ++k) { a[k] = W16ReadLaneF(fv, k); b[k] = W16ReadLaneF(fv, 16 + k); }
                f0 = BiquadState{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], __builtin_bit_cast(int32_t, a[12])};
                f1 = BiquadState{b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], __builtin_bit_cast(int32_t, b[12])};
            }
            if(on)
            {
                if(f0.counter <= 0 && f1.counter <= 0)
                {
                    BiquadDualWaveScan(f0, f1, w.g.smp + outPos, N - outPos, lane);
                    if(lane == 0) { slots[0].f.z1 = f0.z1; slots[0].f.z2 = f0.z2; slots[1].f.z1 = f1.z1; slots[1].f.z2 = f1.z2; }
                }
                else
                {
                    if(lane == 0) BiquadDualInterp(f0, f1, w.g.smp + outPos, w.g.smp + outPos, N - outPos);
                    if(lane == 0) { slots[0].f = f0; slots[1].f = f1; }
                }
            }
            else
            {   // an inactive pair is cleared (voice.cpp:264-265); the store is skipped when it already is clear
                const bool clean0 = f0.z1 == 0.0f && f0.z2 == 0.0f && f0.counter == 0 && f0.b0 == f0.tb0 && f0.b1 == f0.tb1
                    && f0.b2 == f0.tb2 && f0.a1 == f0.ta1 && f0.a2 == f0.ta2;
                const bool clean1 = f1.z1 == 0.0f && f1.z2 == 0.0f && f1.counter == 0 && f1.b0 == f1.tb0 && f1.b1 == f1.tb1
                    && f1.b2 == f1.tb2 && f1.a1 == f1.ta1 && f1.a2 == f1.ta2;
                if(!(clean0 && clean1) && lane == 0)
                {
                    BiquadClear(f0); BiquadClear(f1);
                    slots[0].f = f0; slots[1].f = f1;
                }
            }
        };
        if(head.flags & kFlagAmbiScale)
        {   // VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091
            const AmbiScaleState a = L.ambi[v];
            SplitterState sp{a.coeff, a.lpZ1, a.lpZ2, a.apZ1};
            toLine();
            SplitterScan<false>(sp, w.g.smp + outPos, N - outPos, a.hfScale, a.lfScale, lane);
            fromLine();
            if(lane == 0) { L.ambi[v].lpZ1 = sp.lpZ1; L.ambi[v].lpZ2 = sp.lpZ2; L.ambi[v].apZ1 = sp.apZ1; }
        }
        counter = (head.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093

        if constexpr (SENDS)
        {   // ---------------- the auxiliary sends (voice.cpp:966-983): their rows leave before the direct filter runs ----------------
            const uint32_t ls = S.lineStride, spv = S.spv, numSends = S.numSends, wetCh = S.wetCh;
            float *rowsV = S.streams + size_t{v} * spv * kLine;
            uint32_t *blkV = S.lineGains + size_t{v} * spv * LineBlockDwords(ls);
            int32_t sendSlots[6];
            {   // VoiceCtl::sendSlot, bytes 48..71 of the voice's control line: through the scalar cache
                static_assert(offsetof(VoiceCtl, sendSlot) == 48, "VoiceCtl::sendSlot follows the head");
                cu4 *src = (cu4*)(uintptr_t)(L.ctl + v);
                const u4 a = src[3], b = src[4];
                sendSlots[0] = int32_t(a.x); sendSlots[1] = int32_t(a.y); sendSlots[2] = int32_t(a.z); sendSlots[3] = int32_t(a.w);
                sendSlots[4] = int32_t(b.x); sendSlots[5] = int32_t(b.y);
            }
            RowLineGain row0;                           // the unfiltered row's merged gains, wet line = lane
            bool row0Live = false;
            for(uint32_t si = 0; si < numSends; ++si)
            {
                int32_t slot = sendSlots[0];
#pragma unroll
                for(int k = 1; k < 6; ++k) slot = (si == uint32_t(k)) ? sendSlots[k] : slot;
                uint32_t *blkS = blkV + size_t{2u + si} * LineBlockDwords(ls);
                if(slot < 0) { if(lane == 0) blkS[3u * ls] = 0u; continue; }
                const size_t vs = size_t{v} * numSends + si;
                const bool sendFilter = (head.flags >> (kFlagSendFilterShift + si)) & 1u;
                BiquadSlot *slots = &S.sfilt[vs * 2];
                {   // DoFilters with the send's pair FIRST, with nothing of the gains alive across the scan (its registers are the
                    // section's peak): the filtered copy stays in the sample line's place, the unfiltered samples wait in the
                    // window's area (dead behind the resampler) -- in registers across the scan they were 16 more
                    const float sfC = (lane < 32u) ? reinterpret_cast<const float*>(&S.sfilt[vs * 2])[lane] : 0.0f;
                    if(sendFilter)
                    {
                        float *bak = w.g.rd;
#pragma unroll
                        for(int j = 0; j < kW16Outs; ++j) bak[lane + 64u * uint32_t(j)] = outs[j];
                        toLine();
                        doPair(sfC, slots, true);
                        WaveSync();
#pragma unroll
                        for(int j = 0; j < kW16Outs; ++j) outs[j] = bak[lane + 64u * uint32_t(j)];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    else doPair(sfC, slots, false);
                }
                // the send's gains onto its slot's wet lines (voice.cpp:978-979); an HRTF context's mix lines ARE the wet lines
                float stC = 0.0f, scC = 0.0f;
                if(lane < wetCh) { stC = S.sendTgt[vs * wetCh + lane]; scC = S.sendCur[vs * wetCh + lane]; }
                const uint32_t base = uint32_t(slot) * wetCh;
                const bool mine = lane >= base && lane < base + wetCh;
                float tg = 0.0f, cu = 0.0f;
                float *curp = S.sendCur + vs * wetCh + (lane - base);
                {
                    const float tgS = __shfl(stC, int(lane - base)), cuS = __shfl(scC, int(lane - base));
                    if(mine)
                    {
                        tg = playing ? tgS : 0.0f;
                        cu = counter ? cuS : tg;
                    }
                }
                const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                if(mine) *curp = g.newCur;
                if(sendFilter)
                {
                    RowLineGain r;
                    if(mine) r.add(g);
                    float *dst = rowsV + size_t{2u + si} * kLine;
#pragma unroll
                    for(int j = 0; j < kW16Outs; ++j)
                    {
                        const uint32_t k = lane + 64u * uint32_t(j);
                        dst[k] = (k < N) ? w.g.smp[k] : 0.0f;
                    }
                    StoreRowBlock(blkS, ls, lane, r, true);
                    WaveSync();
                }
                else
                {
                    if(mine) row0.add(g);
                    row0Live = true;
                    if(lane == 0) blkS[3u * ls] = 0u;
                }
            }
            if(row0Live)
            {
#pragma unroll
                for(int j = 0; j < kW16Outs; ++j) rowsV[lane + 64u * uint32_t(j)] = outs[j];
            }
            StoreRowBlock(blkV, ls, lane, row0, row0Live);
            WaveSync();
        }

        // ---------------- DoFilters, direct path (voice.cpp:255-267) ----------------
        {
            const bool directFilter = (head.flags & kFlagDirectFilter) != 0;
            BiquadSlot *slots = &L.dfilt[size_t{v} * 2];
            if(directFilter)
            {
                toLine();
                doPair(fstv, slots, true);
                fromLine();
            }
            else doPair(fstv, slots, false);
        }

        stamp(2);
        __builtin_amdgcn_s_setprio(0);
        // ---------------- DoHrtfMix, voice.cpp:827-902 ----------------
        // Hrtf.History for the next update: in[N .. N + 63] of [history | samples]
        {
            const uint32_t q = N + lane;                        // index into [history(64) | samples]
            float hv;
            if(N == uint32_t(kLine)) hv = outs[kW16Outs - 1];   // samples 960 + lane
            else
            {
                const uint32_t k = q - uint32_t(kHist);         // sample index where q >= 64
                const uint32_t j0 = (N - uint32_t(kHist)) >> 6; // (N >= 64 here when used; uniform)
                float lo = 0.0f, hi2 = 0.0f;
#pragma unroll
                for(int j = 0; j < kW16Outs; ++j) { if(uint32_t(j) == j0) lo = outs[j]; if(uint32_t(j) == j0 + 1u) hi2 = outs[j]; }
                const float sLo = __shfl(lo, int(k & 63u)), sHi = __shfl(hi2, int(k & 63u));
                const float smpv = (N >= uint32_t(kHist)) ? (((k >> 6) == j0) ? sLo : sHi) : __shfl(outs[0], int(k & 63u));
                const float old = __shfl(histv, int(q & 63u));
                hv = (q >= uint32_t(kHist)) ? smpv : old;
            }
            if(playing) __builtin_nontemporal_store(hv, &L.hist[size_t{v} * kHist + lane]);
        }
        w.b.in0[lane] = histv;
        w.b.in0[kHist + lane] = outs[0];

        const float targetGain = tail.tgtGain * (playing ? 1.0f : 0.0f);
        const float oldGain = counter ? tail.oldGain : tail.tgtGain;   // voice.cpp:1100
        float blendGain = targetGain;
        if(counter)
        {
            fademix = N < counter ? N : counter;
            if(counter > fademix)
                blendGain = lerpf(oldGain, targetGain, float(fademix) / float(counter));
        }
        const float newStep = fademix ? blendGain / float(fademix) : 0.0f;
        gainAfterBlend = fademix ? blendGain : oldGain;
        todo = N - fademix;
        endGain = targetGain;
        if(todo && counter > N)
            endGain = lerpf(gainAfterBlend, targetGain, float(todo) / float(counter - fademix));
        const float mainStep = todo ? (endGain - gainAfterBlend) / float(todo) : 0.0f;
        const bool oldOn = fademix && oldGain > kGainSilence;
        const bool newOn = fademix && newStep * float(fademix) > kGainSilence;
        const float oldStep = fademix ? oldGain / float(fademix) : 0.0f;
        const bool merged = !dirty;
        const bool oldPass = !merged && oldOn;
        // (a replaced filter's old response, tap = lane: requested here, used behind the first ear's main FIR)
        const f2 hO = (oldPass && lane < irStride) ? reinterpret_cast<const f2*>(L.hrtfOld + size_t{v} * irStride * 2)[lane] : f2{0.0f, 0.0f};
        const float gbase = gainAfterBlend - mainStep * float(fademix);
        // g(i): the gain of input frame i (MixHrtfBlend's two ramps summed while Old == Target, hrtfbase.h:54-88)
        auto gainAt = [&](uint32_t i, bool mayFade)
        {
            float g = __builtin_fmaf(mainStep, float(i), gbase);
            if(mayFade && i < fademix)
            {
                float gf = newOn ? newStep * float(i) : 0.0f;
                if(merged && oldOn) gf += oldStep * float(fademix - i);
                g = gf;
            }
            return g;
        };
        WaveSync();
        stamp2(0);

        // The inputs' scale -- the power of two that puts their largest magnitude into [2^14, 2^15) -- comes from a bound, once for both
        // ears: the largest sample (or history value) times the largest gain of the block.  (The products' own maximum was a
        // second pass over them per ear, for a bit or two of headroom the split does not need: a value's halves keep 22
        // significant bits wherever it lies below the block's maximum.)
        float sx, invX;
        {
            float mxo = __builtin_fabsf(histv);
#pragma unroll
            for(int j = 0; j < kW16Outs; ++j) mxo = __builtin_fmaxf(__builtin_fabsf(outs[j]), mxo);
            const float mxw = __builtin_bit_cast(float, WaveMaxBits(__builtin_bit_cast(uint32_t, mxo)));
            const float gmax = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(blendGain) + __builtin_fabsf(oldGain), __builtin_fabsf(gainAfterBlend)),
                __builtin_fabsf(endGain));
            HalfScale(__builtin_bit_cast(uint32_t, mxw * gmax), sx, invX);
            invX = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, invX)));
            sx = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sx)));
        }
        const float step64 = mainStep * 64.0f * sx;
        // x'[i] = in[64 - d + i] * g(i), i < N: frames i >= d are sample i - d (this lane: i = lane + 64 j + d), frames below d come
        // out of the history (lane = frame); split into halves and stored at half index 64 + i.  FULL: a whole line -- only the last
        // 64 samples can land at or beyond frame N.
        auto buildEar = [&](uint32_t d, auto fullTag)
        {
            constexpr bool FULL = decltype(fullTag)::value;
            // zero pads: frames -64 .. -1 and N .. 1103 (whole dwords: N is even or the odd half is rewritten below)
            if(lane < 32u) { w.b.xh[0][lane] = 0u; w.b.xh[1][lane] = 0u; }
            for(uint32_t dw = 32u + (N >> 1) + lane; dw < uint32_t(kXhDw); dw += 64u) { w.b.xh[0][dw] = 0u; w.b.xh[1][dw] = 0u; }
            WaveSync();                                         // (the halves below go through a pointer of another type)
            uint16_t *xz = reinterpret_cast<uint16_t*>(&w.b.xh[0][0]) + 64u + lane + d;       // frame lane + d
            if(lane < d)
            {
                const float headP = (lane < N) ? w.b.in0[kHist - d + lane] * gainAt(lane, true) : 0.0f;
                uint32_t hi, lo;
                SplitHalf2(headP * sx, 0.0f, hi, lo);
                uint16_t *xh0 = reinterpret_cast<uint16_t*>(&w.b.xh[0][0]) + 64u + lane;
                xh0[0] = uint16_t(hi); xh0[kXhHalves] = uint16_t(lo);
            }
            const uint32_t i00 = lane + d;
            const float g0s = gainAt(i00, true) * sx;                                  // frame lane + d: the only row inside the fade
            const float gls = __builtin_fmaf(mainStep, float(i00), gbase) * sx;        // the main ramp at that frame, scaled
#pragma unroll
            for(int j = 0; j < kW16Outs; j += 2)
            {
                const float ga = (j == 0) ? g0s : __builtin_fmaf(step64, float(j), gls);
                const float gb = __builtin_fmaf(step64, float(j + 1), gls);
                float p0 = outs[j] * ga, p1 = outs[j + 1] * gb;
                if(!FULL || j + 2 == kW16Outs)
                {
                    if(i00 + 64u * uint32_t(j) >= N) p0 = 0.0f;
                    if(i00 + 64u * uint32_t(j + 1) >= N) p1 = 0.0f;
                }
                uint32_t hi, lo;
                SplitHalf2(p0, p1, hi, lo);
                // (frames up to 1023 + 63 < 1104: inside the array for any delay)
                xz[64 * j] = uint16_t(hi); xz[kXhHalves + 64 * j] = uint16_t(lo);
                xz[64 * (j + 1)] = uint16_t(hi >> 16); xz[kXhHalves + 64 * (j + 1)] = uint16_t(lo >> 16);
            }
        };

#pragma unroll
        for(int e = 0; e < 2; ++e)
        {
            const uint32_t d = tail.tgtDelay[e];
            if(N == uint32_t(kLine)) buildEar(d, std::true_type{}); else buildEar(d, std::false_type{});
            uint16_t *xz = reinterpret_cast<uint16_t*>(&w.b.xh[0][0]);
            if(e == 0) stamp2(1);
            const float invH = W16StageResponse(w.b.hr, e == 0 ? hT.x : hT.y, lane);
            WaveSync();
            if(e == 0) stamp2(2);
            FirMfmaEar<5, false>(accM[e], w.b.xh, w.b.hr, invX * invH, lane);
            if(e == 0) stamp2(3);
            if(oldPass)
            {   // the replaced filter's fade-out (MixHrtfBlend, hrtfbase.h:54-70): 64 inputs x IrSize taps land in frames 0..126 --
                // the first eight columns of tile 0.  Its inputs go over the main inputs' frames -64..143, frame = lane.
                const uint32_t od = tail.oldDelay[e];
                float xo = 0.0f;
                if(lane < fademix) xo = w.b.in0[kHist - od + lane] * (oldStep * float(fademix - lane));
                float sxo, invXo;
                HalfScale(WaveMaxBits(__builtin_bit_cast(uint32_t, __builtin_fabsf(xo))), sxo, invXo);
                invXo = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, invXo)));
                uint32_t hi, lo;
                SplitHalf2(xo * sxo, 0.0f, hi, lo);
                WaveSync();
                xz[0 * kXhHalves + 64u + lane] = uint16_t(hi); xz[1 * kXhHalves + 64u + lane] = uint16_t(lo);
                if(lane < 40u) { w.b.xh[0][64u + lane] = 0u; w.b.xh[1][64u + lane] = 0u; }
                const float invHO = W16StageResponse(w.b.hro, e == 0 ? hO.x : hO.y, lane);
                WaveSync();
                FirMfmaEar<1, true>(accM[e], w.b.xh, w.b.hro, invXo * invHO, lane);
            }
            WaveSync();
            stamp(3 + e);
        }
    }

    // ---------------- state write-back (voice.cpp:1094-1232) ----------------
    if(active)
    {
        if(dirty && (counter == 0 || fademix))
        {   // Old <- Target
            if(lane < irStride) reinterpret_cast<f2*>(L.hrtfOld + size_t{v} * irStride * 2)[lane] = hT;
        }
        if(lane == 0)
        {
            VoiceCtl &c = L.ctl[v];
            if(counter == 0 || fademix) { c.hrtfOldDelay[0] = tail.tgtDelay[0]; c.hrtfOldDelay[1] = tail.tgtDelay[1]; }
            c.hrtfOldGain = todo ? endGain : gainAfterBlend;
            uint32_t flags = (head.flags | kFlagFading) & ~kFlagDelayed;
            if(counter == 0 || fademix) flags &= ~kFlagHrtfDirty;
            c.flags = flags;
            if(!playing) c.playState = OALGPU_VOICE_STOPPED;
            else
            {
                int32_t bufPosInt = head.position;
                uint32_t bufPosFrac = head.positionFrac + head.step * (N - outPos);
                const uint32_t samplesDone = bufPosFrac >> kFracBits;
                bufPosInt = AddSat(bufPosInt, int32_t(samplesDone));
                bufPosFrac &= kFracMask;
                if(bufferItem >= 0 && bufPosInt > 0 && (head.flags & kFlagQueue))
                {   // a streaming source: buffers the position ran past are done (voice.cpp:1182-1194)
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
    }

    // ---- the next update's parameter block: every wavefront installs the record of the voice it has just mixed.  Here, in front
    // of the dump's barrier: the install is a chain of dependent round trips (map entry -> record -> stores), and all but the
    // workgroup's last wavefront spend it waiting for that one anyway.  The voice's state was written back by this very wavefront,
    // in program order.
    if(!RES && next.map && haveVoice)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if(next.rows && irStride <= 64u) InstallPair(L, next.map, next.recs, next.rows, v, v, false, lane);
        else
        {
            const int32_t ri = __builtin_amdgcn_readfirstlane(next.map[v]);
            if(ri >= 0) ApplyNextRecord(L, next.recs[ri], lane);
        }
    }
    stamp(5);
    // ---------------- one partial bus per workgroup: the wavefronts dump their tiles, then a fixed-order sum ----------------
    {
        WaveSync();
        f2 *dump = w.dump;
        const uint32_t jc = lane & 15u, q4 = lane >> 4;
#pragma unroll
        for(int T = 0; T < 4; ++T)
#pragma unroll
            for(int r = 0; r < 4; ++r)
                dump[17u * (16u * uint32_t(T) + jc) + 4u * q4 + uint32_t(r)] = f2{accM[0][T][r], accM[1][T][r]};
        if(jc < 4u)
        {
#pragma unroll
            for(int r = 0; r < 4; ++r) dump[17u * (64u + jc) + 4u * q4 + uint32_t(r)] = f2{accM[0][4][r], accM[1][4][r]};
        }
        __syncthreads();
        f2 *ph = reinterpret_cast<f2*>(L.partHrtf) + size_t{group} * (kLine + kHrirLen);
        if constexpr (RES)      // the update's set of partial buses
            ph = reinterpret_cast<f2*>(next.res.partBase + size_t{upd % kResidentSets} * next.res.setStride) + size_t{group} * (kLine + kHrirLen);
        for(uint32_t k = t; k < uint32_t(kLine + kHrirLen); k += uint32_t(kW16Threads))
        {
            f2 s = {0.0f, 0.0f};
            if(k < 64u * 17u)
            {
                const uint32_t at = k + (k >> 4);
                s = sm.w[0].dump[at];
#pragma unroll
                for(int ww = 1; ww < WAVES; ++ww) { const f2 o = sm.w[ww].dump[at]; s.x += o.x; s.y += o.y; }
            }
            if constexpr (RES) StorePartialCoherent(&ph[k], s);       // (read by the reduction's launch while this one runs on)
            else StorePartial(&ph[k], s);
        }
    }
    stamp(6);
    if constexpr (!RES) break;
    else
    {   // the workgroup's partial is where the reduction will read it (written through, every thread's stores acknowledged)
        // before the workgroup counts as arrived for this update; then on to the next one
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if(t == 0)
        {
            __hip_atomic_fetch_add(next.res.counters + 16u * (kRcArrive0 + upd % kResidentSets), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if((group & 127u) == 0u)
                __hip_atomic_fetch_add(next.res.counters + 16u * kRcBusy, uint32_t(__builtin_amdgcn_s_memrealtime()) - sm.res[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        ++upd;
    }
    }
}

} // namespace

// FAST HRTF contexts on the matrix-pipe FIR, without sends or -- `sends` -- with them (their rows leave as stream rows, see W16Sends)
bool Wave16Applies(const DeviceLayout &L)
{
    if(!(L.hrtf && L.firMfma && L.irStride >= 8 && L.irStride <= 64 && L.sliceLines == 0 && L.nfc == nullptr)) return false;
    if(L.numSends == 0) return L.accLines == 0;
    return L.numSends <= 6 && L.mixLines >= 1 && L.mixLines <= 32 && L.wetChannels <= 32u;
}
// wavefronts per workgroup for a scene of `voices` on a device of `cus` compute units: the smallest of 4 / 8 / 16 whose grid fits the
// machine in one round (16 beyond that: the rounds follow each other out of phase)
uint32_t Wave16WavesFor(uint32_t voices, uint32_t cus)
{
    for(uint32_t w : {4u, 8u}) if((voices + w - 1u) / w <= cus) return w;
    return 16u;
}
uint32_t Wave16Groups(const DeviceLayout &L) { return (L.numVoices + L.wave16 - 1u) / L.wave16; }
const char *Wave16KernelName(const DeviceLayout &L)
{
    if(L.numSends) return L.wave16 == 16u ? "VoiceWave16Kernel<16, sends>" : (L.wave16 == 8u ? "VoiceWave16Kernel<8, sends>" : "VoiceWave16Kernel<4, sends>");
    return L.wave16 == 16u ? "VoiceWave16Kernel<16>" : (L.wave16 == 8u ? "VoiceWave16Kernel<8>" : "VoiceWave16Kernel<4>");
}

hipError_t LaunchVoiceWave16(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const WaveProf *prof, hipEvent_t evStart, hipEvent_t evStop,
    const ParamRecord *nextRecs, const int32_t *nextMap, const float *nextRows)
{
    const Next16 next{nextRecs, nextMap, nextRows, ResidentArgs{}};
    const WaveProf none{nullptr, 0u};
    const dim3 grid(Wave16Groups(L)), block(L.wave16 * 64u);
    const W16Sends S{L.sfilt, L.sendTgt, L.sendCur, L.streams, L.lineGains, L.numSends, L.wetChannels, L.lineStride, L.streamsPerVoice};
#define OALGPU_W16_LAUNCH(P, W, SD, PA) hipExtLaunchKernelGGL((VoiceWave16Kernel<P, W, SD>), grid, block, 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, samplesToDo, next, PA, S)
    if(L.numSends)
    {   // (no measurement variant and no parameter block installed by the kernel: WaveKernelAppliesRecords says so to the host)
        // (the launch's stop event -- what the post stream waits for, and the end of what the host times -- belongs to the LAST kernel)
        const Next16 noNext{nullptr, nullptr, nullptr, ResidentArgs{}};
        if(L.wave16 == 16u) hipExtLaunchKernelGGL((VoiceWave16Kernel<false, 16, true>), grid, block, 0, s, evStart, nullptr, 0u, WaveArgsHrtf{L}, samplesToDo, noNext, none, S);
        else if(L.wave16 == 8u) hipExtLaunchKernelGGL((VoiceWave16Kernel<false, 8, true>), grid, block, 0, s, evStart, nullptr, 0u, WaveArgsHrtf{L}, samplesToDo, noNext, none, S);
        else hipExtLaunchKernelGGL((VoiceWave16Kernel<false, 4, true>), grid, block, 0, s, evStart, nullptr, 0u, WaveArgsHrtf{L}, samplesToDo, noNext, none, S);
        if(hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        return LaunchStreamRowsMix(s, L, samplesToDo, L.wave16, evStop);     // (voice_wave.hip: the workgroups' rows onto their partial wet lines)
    }
    if(prof)
    {
        if(L.wave16 == 16u) OALGPU_W16_LAUNCH(true, 16, false, *prof); else if(L.wave16 == 8u) OALGPU_W16_LAUNCH(true, 8, false, *prof); else OALGPU_W16_LAUNCH(true, 4, false, *prof);
    }
    else
    {
        if(L.wave16 == 16u) OALGPU_W16_LAUNCH(false, 16, false, none); else if(L.wave16 == 8u) OALGPU_W16_LAUNCH(false, 8, false, none); else OALGPU_W16_LAUNCH(false, 4, false, none);
    }
#undef OALGPU_W16_LAUNCH
    return hipGetLastError();
}

// the resident launch of the 16-wavefront form (OALGPU_CTX_RESIDENT; voice_wave_res.hip dispatches here)
hipError_t LaunchVoiceWave16Resident(hipStream_t s, const DeviceLayout &L, const ResidentArgs &args, hipEvent_t evStart, hipEvent_t evStop)
{
    const Next16 next{nullptr, nullptr, nullptr, args};
    const WaveProf none{nullptr, 0u};
    hipExtLaunchKernelGGL((VoiceWave16Kernel<false, 16, false, true>), dim3(Wave16Groups(L)), dim3(1024), 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, 0u, next, none, W16Sends{});
    return hipGetLastError();
}
int Wave16ResidentGroupsPerCu()
{
    int n = 0;
    if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, VoiceWave16Kernel<false, 16, false, true>, 1024, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

} // namespace oalgpu
