// The FAST HRTF hot path, second form: one WORKGROUP of four wavefronts per voice (BASELINE
// configs[2], the headline metric).
//
//   "for every Playing|Stopping voice: voice->mix(...)"   alc/alu.cpp:2201-2206
//      -> Voice::mix                                       core/voice.cpp:988-1233
//      -> LoadResampledSamples / Resample_*                voice.cpp:642-824, mixer_sse.cpp:199-329
//      -> DoFilters (dual biquad)                          voice.cpp:255-267
//      -> DoHrtfMix -> MixHrtf / MixHrtfBlend              voice.cpp:827-902, mixer/hrtfbase.h:17-89
//
// Why a workgroup per voice.  The wavefront-per-voice kernel (voice_wave.hip) keeps a voice's 1088
// output frames in one wavefront's registers because its packed-VALU FIR needs a sliding input
// window per lane; that costs 233 VGPRs, i.e. two wavefronts per SIMD, and every voice is a chain
// of ~30 K cycles of mostly latency (measured, DESIGN.md 3.4: VALU 45 % and LDS 36 % busy).  With the
// dual-ear FIR on the matrix pipe (FirMfmaTile, dev_wave.hpp) nothing needs a per-lane window any
// more, so all phases of a voice spread over 256 lanes:
//   * resampler: 4 outputs per lane instead of 16 (staged rows as in voice_wave.hip, taps in two
//     groups of 6 pairs, the next group's LDS reads in flight while one is multiplied);
//   * dual biquad: a block scan over 256 runs of 4 samples (in-wave Kogge-Stone + a 4-step hand-over
//     between the wavefronts through LDS);
//   * FIR: wavefront w owns output tile b = w (frames 64 a + 16 w + r) of both ears and a quarter of
//     the ring-out tile: 48 v_mfma_f32_16x16x4_f32 per voice and wavefront, accumulators in 20 VGPRs
//     for the whole launch.
// The kernel needs < 128 VGPRs and 29 KB of LDS per workgroup: four workgroups (16 wavefronts) per CU,
// whose phases interleave -- one workgroup's MFMAs run beside another's LDS-bound resampler and a
// third's VALU-bound filter scan.  Voices of a workgroup follow each other; the next voice's source
// window, history and HRIR are requested from HBM right after the current voice's were parked in
// LDS, so their latency hides behind a whole voice.
//
// Everything uncommon (a chunked or delayed source window, a missing buffer, interpolating filter
// coefficients, B-Format prescale) runs on wavefront 0 through the wave-level routines of
// wave_common.hpp, on the same LDS buffers.  Arithmetic is FAST mode (FMA, own summation order:
// tolerance in DESIGN.md); all integer state is bit-exact.
#include "wave_common.hpp"

#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kBT = 256;                      // threads per workgroup
constexpr int kBW = kBT / 64;
constexpr int kPreB = 5;                      // prefetched source samples per lane (5*256 = 1280 >= 1024 + 48 at pitch 1)

struct alignas(16) BlockLds {
    union {
        float xp[2][kXpFloats];               // FIR inputs per ear, planar, 17 rows of stride 68 (wave_common.hpp)
        float rd[kResampleDataSize + 8];      // DeviceBase::mResampleData (dead before xp is built)
        f2 dump[kLine + kHrirLen];            // epilogue: [frame] = (L, R)
    };
    float in[kHist + kLine];                  // [Hrtf.History | resampled, filtered samples]
    float coldP[2][64 + 128];                 // coldP[e][64 + j] = Hrtf.Old.Coeffs[j][e], zero padded
    float xoP[2][64];                         // old-filter fade-out inputs (i < 64) per ear
    float hp[2][kHpLen];                      // hp[e][16 + j] = Hrtf.Target.Coeffs[j][e], zero padded
    float fst[32];                            // the voice's two BiquadSlots (2 x 16 dwords)
    float xch[16];                            // filter scan: wave totals [4][2] | final state [2]
    f2 acc[kLine];                            // the workgroup's stereo accumulator, frames 0..1023 (L, R)
    f2 accTail[kBW][64];                      // ring-out frames 1024..1087: one part per wavefront
    int32_t best;
    uint32_t pad[3];
    f2 tabF[kTabPairs * 32];                  // [tap pair][phase] = fil[2p], fil[2p+1]
    f2 tabP[kTabPairs * 32];                  //                    = phd[2p], phd[2p+1]
    uint32_t tabKey, tabM, tabL;
    uint32_t pad2;
};

// ---- source window ---------------------------------------------------------------------------------
// The first chunk of LoadBufferStatic (core/voice.cpp:500-544) as a register gather by the whole
// workgroup: thread t requests elements t, t + 256, ... of the `count` source samples that start at
// buffer position dataPos.  One index formula covers the three shapes -- inside the buffer, past the
// end of a non-looping buffer (holds the last sample), one wrap at the loop end -- so the loads are
// 32-bit offsets off the buffer's (scalar) base address; elements past `count` are loaded (clamped
// into the buffer) and never used.  Mono float / int16 buffers; anything else is filled by the generic
// LoadBufferStatic loop when the voice starts (GatherWindowCovers() == false).
__device__ __forceinline__ bool GatherWindowCovers(uint32_t count, const BufferItem &b, bool looping, uint32_t dataPos)
{
    if((b.fmt != OALGPU_FMT_FLOAT && b.fmt != OALGPU_FMT_SHORT) || b.frameStep != 1u || b.sampleLen == 0u) return false;
    if(!looping) return dataPos < b.sampleLen;
    return dataPos < b.loopEnd && b.loopEnd <= b.sampleLen
        && count <= (b.loopEnd - dataPos) + (b.loopEnd - b.loopStart);
}

template<int NPRE, int NT>
__device__ __forceinline__ void GatherWindow(float (&pre)[NPRE], const BufferItem &b, bool looping, uint32_t dataPos, uint32_t tid)
{
    const uint32_t lastIdx = b.sampleLen - 1u;
    const uint32_t first = looping ? b.loopEnd - dataPos : 0xffffffffu;     // elements before the wrap
    const uint32_t wrapTo = b.loopStart;
    const bool isShort = b.fmt == OALGPU_FMT_SHORT;
    typedef const __attribute__((address_space(1))) float *gf;
    typedef const __attribute__((address_space(1))) int16_t *gs;
    const gf pf = reinterpret_cast<gf>((const __attribute__((address_space(1))) void*)b.data);
    const gs ps = reinterpret_cast<gs>((const __attribute__((address_space(1))) void*)b.data);
#pragma unroll
    for(int i = 0; i < NPRE; ++i)
    {
        const uint32_t k = tid + uint32_t(NT) * uint32_t(i);
        uint32_t idx = (k < first) ? dataPos + k : wrapTo + (k - first);
        idx = idx < lastIdx ? idx : lastIdx;
        pre[i] = isShort ? __builtin_bit_cast(float, int32_t(ps[idx])) : pf[idx];
    }
}

// ---- dual biquad as a 256-lane block scan (time-invariant coefficients) ---------------------------
// BiquadFilter::dualProcess (core/filters/biquad.cpp:254-282): two transposed-direct-form-II sections
// in cascade, each linear in its state s = (z1, z2): s' = A s + B x.  Thread t owns samples
// [4t, 4t+4) for both sections (one ds_read_b128 / ds_write_b128).  Per section: M = A^4; forced
// response q_t of the run; an inclusive Kogge-Stone scan inside each wavefront (powers M, M^2 .. M^32);
// the wavefronts' totals and W = M^64 hand the run-start state from wavefront to wavefront through
// LDS (three steps at most); then the true recurrence from the run-start state.  The reference's
// serial loop differs from this by rounding only.  All kBT threads call this (two barriers per
// section); `xch` = 16 floats of LDS scratch.
__device__ __forceinline__ void BiquadBlockSection(float (&x)[4], uint32_t cnt, const BiquadState &f, float &z1, float &z2,
    uint32_t tid, uint32_t lastTid, float *xch)
{
    const uint32_t lane = tid & 63u, wave = tid >> 6;
    const float b0 = f.b0, b1 = f.b1, b2 = f.b2, a1 = f.a1, a2 = f.a2;
    S2 m0{1.0f, 0.0f}, m1{0.0f, 1.0f};                 // columns of M = A^4
#pragma unroll
    for(int i = 0; i < 4; ++i)
    {
        m0 = S2{__builtin_fmaf(-a1, m0.a, m0.b), -a2 * m0.a};
        m1 = S2{__builtin_fmaf(-a1, m1.a, m1.b), -a2 * m1.a};
    }
    S2 e{0.0f, 0.0f};
#pragma unroll
    for(int i = 0; i < 4; ++i) BqStep(e, x[i], b0, b1, b2, a1, a2);     // x[i] = 0 past the end
    S2 ua{1.0f, 0.0f}, ub{0.0f, 1.0f};                // -> columns of M^lane
    S2 p0 = m0, p1 = m1;                              // M^(2^step)
#pragma unroll
    for(int step = 0; step < 6; ++step)
    {
        const int d = 1 << step;
        const S2 o{__shfl_up(e.a, d), __shfl_up(e.b, d)};
        const S2 mo = Mv2(p0, p1, o);
        if(int(lane) >= d) { e.a += mo.a; e.b += mo.b; }
        const S2 na = Mv2(p0, p1, ua), nb = Mv2(p0, p1, ub);
        if(lane & uint32_t(d)) { ua = na; ub = nb; }
        const S2 n0 = Mv2(p0, p1, p0), n1 = Mv2(p0, p1, p1); p0 = n0; p1 = n1;
    }
    // p0, p1 = W = M^64; e of lane 63 = the wavefront's forced response from a zero state
    if(lane == 63u) { xch[2u * wave] = e.a; xch[2u * wave + 1u] = e.b; }
    __syncthreads();
    S2 sw{z1, z2};                                    // state at the start of this wavefront's samples
    for(uint32_t w = 0; w < wave; ++w)
    {
        const S2 n = Mv2(p0, p1, sw);
        sw = S2{n.a + xch[2u * w], n.b + xch[2u * w + 1u]};
    }
    const S2 prevE{__shfl_up(e.a, 1), __shfl_up(e.b, 1)};
    S2 st{__builtin_fmaf(ub.a, sw.b, ua.a * sw.a), __builtin_fmaf(ub.b, sw.b, ua.b * sw.a)};   // M^lane sw
    if(lane > 0) { st.a += prevE.a; st.b += prevE.b; }
#pragma unroll
    for(int i = 0; i < 4; ++i)
        if(uint32_t(i) < cnt) x[i] = BqStep(st, x[i], b0, b1, b2, a1, a2);
    if(tid == lastTid) { xch[8] = st.a; xch[9] = st.b; }
    __syncthreads();
    z1 = xch[8]; z2 = xch[9];
    __syncthreads();                                  // xch is reused by the next section
}

__device__ __forceinline__ void BiquadDualBlockScan(BiquadState &f0, BiquadState &f1, float *buf /* in place, 16-byte aligned */,
    uint32_t n, uint32_t tid, float *xch)
{
    const uint32_t begin = 4u * tid;
    const uint32_t cnt = begin < n ? (n - begin < 4u ? n - begin : 4u) : 0u;
    const uint32_t lastTid = (n - 1u) / 4u;
    const f4 xv = *reinterpret_cast<const f4*>(buf + begin);          // 4 tid + 3 <= 1023: inside the line
    float x[4];
#pragma unroll
    for(int i = 0; i < 4; ++i) x[i] = (uint32_t(i) < cnt) ? xv[i] : 0.0f;
    BiquadBlockSection(x, cnt, f0, f0.z1, f0.z2, tid, lastTid, xch);
    BiquadBlockSection(x, cnt, f1, f1.z1, f1.z2, tid, lastTid, xch);
    if(cnt == 4u) *reinterpret_cast<f4*>(buf + begin) = f4{x[0], x[1], x[2], x[3]};
    else
    {
#pragma unroll
        for(int i = 0; i < 4; ++i) if(uint32_t(i) < cnt) buf[begin + i] = x[i];
    }
}

// (out of line: see the note on the uncommon paths below)
__device__ __forceinline__ void RareBiquadInterp(BiquadState *f0, BiquadState *f1, float *buf, uint32_t n)
{ BiquadDualInterp(*f0, *f1, buf, buf, n); }

// DoFilters (voice.cpp:255-267) for the direct path, in place over buf[0..n), by the whole workgroup.
// (`aligned`: buf is 16-byte aligned -- a delayed start's line is not, and takes the serial loop)
__device__ __forceinline__ void BlockDoFilters(BlockLds &sm, BiquadSlot *slots, bool filterActive, float *buf, uint32_t n,
    uint32_t tid, bool aligned)
{
    if(!filterActive && tid != 0) return;               // voice.cpp:264-265 is one thread's work
    BiquadState f0, f1;
    {
        const float *a = sm.fst, *b = sm.fst + 16;
        f0 = BiquadState{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], __builtin_bit_cast(int32_t, a[12])};
        f1 = BiquadState{b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], __builtin_bit_cast(int32_t, b[12])};
    }
    if(filterActive)
    {
        if(f0.counter <= 0 && f1.counter <= 0 && aligned)
        {
            BiquadDualBlockScan(f0, f1, buf, n, tid, sm.xch);
            if(tid == 0) { slots[0].f.z1 = f0.z1; slots[0].f.z2 = f0.z2; slots[1].f.z1 = f1.z1; slots[1].f.z2 = f1.z2; }
        }
        else if(tid == 0)
        {   // coefficients still moving (biquad.cpp:284-343): the reference's serial loop
            BiquadState g0 = f0, g1 = f1;             // (copies: f0 / f1 stay in registers on the common path)
            RareBiquadInterp(&g0, &g1, buf, n);
            slots[0].f = g0; slots[1].f = g1;
        }
    }
    else
    {   // voice.cpp:264-265; skip the store when the pair already is clear
        const bool clean0 = f0.z1 == 0.0f && f0.z2 == 0.0f && f0.counter == 0 && f0.b0 == f0.tb0 && f0.b1 == f0.tb1
            && f0.b2 == f0.tb2 && f0.a1 == f0.ta1 && f0.a2 == f0.ta2;
        const bool clean1 = f1.z1 == 0.0f && f1.z2 == 0.0f && f1.counter == 0 && f1.b0 == f1.tb0 && f1.b1 == f1.tb1
            && f1.b2 == f1.tb2 && f1.a1 == f1.ta1 && f1.a2 == f1.ta2;
        if(!(clean0 && clean1) && tid == 0)
        {
            BiquadClear(f0); BiquadClear(f1);
            slots[0].f = f0; slots[1].f = f1;
        }
    }
}

// ---- the uncommon paths (inlined; kept apart so that the common path reads straight).  NOTE: never take
// the address of the kernel's DeviceLayout or of a voice head: that moves them from SGPRs to the stack.
// (arguments BY VALUE: taking the address of the kernel's DeviceLayout or of a voice head would move them
// from SGPRs to the stack for the whole kernel)
__device__ __forceinline__ void RareLoadResampled(BlockLds *sm, DeviceLayout L, uint32_t v, uint32_t lane,
    VoiceHead head, bool playing, uint32_t N, int32_t bufferItem, bool looping, SrcPlan plan, uint32_t mixOffset)
{ LoadResampledWave<true>(*sm, *sm, L, v, lane, head, playing, N, N, bufferItem, looping, plan, mixOffset); }

__device__ __forceinline__ void RareResampleAt(BlockLds *sm, const float *tables, VoiceHead head, uint32_t N,
    uint32_t t)
{
    const float *filter = tables + head.rsFilterOffset;
    const TabLayout lay = ReferenceTabLayout(head.rsM);
    float *mixing = sm->in + kHist;
    for(uint32_t k = t; k < N; k += kBT)
        mixing[k] = ResampleAt<false, false>(head.rsKind, head.rsM, head.rsL, head.rsSf, filter, lay, sm->rd,
            head.positionFrac, head.step, k, N);
}

__device__ __forceinline__ void RareAmbiScale(BlockLds *sm, AmbiScaleState *ambi, uint32_t N, uint32_t lane, uint32_t mixOffset)
{
    const AmbiScaleState a = *ambi;
    SplitterState sp{a.coeff, a.lpZ1, a.lpZ2, a.apZ1};
    SplitterScan<false>(sp, sm->in + kHist + mixOffset, N, a.hfScale, a.lfScale, lane);
    if(lane == 0) { ambi->lpZ1 = sp.lpZ1; ambi->lpZ2 = sp.lpZ2; ambi->apZ1 = sp.apZ1; }
}

// One workgroup mixes voices [group * blockVoices, ...) one after the other; see the file comment.
// MINW: workgroups (= wavefronts per SIMD) the register allocation leaves room for on a CU.
template<int MINW>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(kBT, MINW) VoiceBlockKernel(DeviceLayout L, uint32_t samplesToDo)
{
    __shared__ BlockLds sm;
    const uint32_t t0 = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t0 >> 6);
    const uint32_t group = blockIdx.x;
    const uint32_t irStride = L.irStride;
    const uint32_t N = samplesToDo;
    const uint32_t vBegin = group * L.blockVoices;
    const uint32_t vEnd = (vBegin + L.blockVoices < L.numVoices) ? vBegin + L.blockVoices : L.numVoices;

    // The accumulated stereo signal of the workgroup's voices lives in LDS (sm.acc / sm.accTail): a
    // voice's FIR tiles are folded into it right after its MFMAs, so no accumulator register is live
    // across the resampler and the filter scan (the kernel is held to 128 VGPRs).
    float accO = 0.0f;                        // old-filter fade-out: frame t & 127 of ear t >> 7

    // ---- the next voice's request: registers from the request to the parking one voice later
    VoiceHead headN{};
    VoiceTail tailN{};
    BufferItem bufN{};
    SrcPlan planN{false, 0u, 0u};
    bool loopingN = false, dirtyN = false;
    float preN[kPreB];
    float prevN = 0.0f, histN = 0.0f, fstN = 0.0f;
    f2 hN = {0.0f, 0.0f}, oldN = {0.0f, 0.0f};
#pragma unroll
    for(int i = 0; i < kPreB; ++i) preN[i] = 0.0f;
    auto request = [&](uint32_t vn, uint32_t t)
    {   // headN / bufN / tailN hold voice vn's control line
        planN = PlanSource<kPreB * kBT>(headN, N);
        loopingN = false;
        if(headN.curBuffer >= 0)
        {   // voice.cpp:1015-1019: a position at or past the loop end plays on without looping
            loopingN = headN.loopBuffer >= 0 && ((headN.flags & kFlagQueue)
                || !(headN.position >= 0 && uint32_t(headN.position) >= bufN.loopEnd));
            planN.prefetch = planN.prefetch && GatherWindowCovers(planN.bsrc, bufN, loopingN, uint32_t(headN.position));
        }
        if(headN.flags & (kFlagDelayed | kFlagQueue)) planN.prefetch = false;   // (see voice_wave.hip)
        if(planN.prefetch)
        {
            GatherWindow<kPreB, kBT>(preN, bufN, loopingN, uint32_t(headN.position), t);
            prevN = (t < uint32_t(kMaxPad)) ? L.prev[size_t{vn} * kMaxPad + t] : 0.0f;
        }
        fstN = (t < 32u) ? reinterpret_cast<const float*>(L.dfilt + size_t{vn} * 2)[t] : 0.0f;
        histN = (t < uint32_t(kHist)) ? L.hist[size_t{vn} * kHist + t] : 0.0f;
        hN = (t < L.irSize) ? reinterpret_cast<const f2*>(L.hrtfTgt + size_t{vn} * irStride * 2)[t] : f2{0.0f, 0.0f};
        dirtyN = (headN.flags & kFlagHrtfDirty) != 0;
        oldN = (dirtyN && t < irStride) ? reinterpret_cast<const f2*>(L.hrtfOld + size_t{vn} * irStride * 2)[t] : f2{0.0f, 0.0f};
    };

    // ---- prologue: first voice's control line, the workgroup's resampler rows, zero pads
    if(vBegin < vEnd)
    {
        headN = LoadHeadScalar(L.ctl + vBegin); bufN = LoadCtlBufferScalar(L.ctl + vBegin); tailN = LoadTailScalar(L.ctl + vBegin);
        request(vBegin, t0);
    }
    {
        const uint32_t t = t0;
        // rows of the first voice that resamples through staged rows (cubic / fast bsinc)
        uint32_t key = 0xffffffffu, m = 0, l = 0;
        for(uint32_t c = vBegin; c < vEnd; ++c)
        {
            const VoiceHead hc = (c == vBegin) ? headN : LoadHeadScalar(L.ctl + c);
            const int kK = hc.rsKind;
            const uint32_t mK = kK == 2 ? 4u : hc.rsM, lK = kK == 2 ? 1u : hc.rsL;
            const bool elig = (kK == 2 || (kK == 3 && (mK == 12 || mK == 24 || mK == 48)))
                && (hc.playState == OALGPU_VOICE_PLAYING || hc.playState == OALGPU_VOICE_STOPPING);
            if(elig) { key = hc.rsFilterOffset * 8u + uint32_t(kK); m = mK; l = lK; break; }
        }
        if(t == 0) { sm.tabKey = key; sm.tabM = m; sm.tabL = l; }
        if(key != 0xffffffffu)
        {
            const float *filter = L.tables + (key >> 3);
            for(uint32_t idx = t; idx < (m / 2u) * 32u; idx += kBT)
            {
                const uint32_t p = idx >> 5, pi = idx & 31u;
                const float *row = filter + pi * 2u * m;
                sm.tabF[idx] = f2{row[2u * p], row[2u * p + 1u]};
                sm.tabP[idx] = f2{row[m + 2u * p], row[m + 2u * p + 1u]};
            }
        }
        for(uint32_t k = t; k < 2u * 192u; k += kBT) (&sm.coldP[0][0])[k] = 0.0f;
        for(uint32_t k = t; k < uint32_t(kLine); k += kBT) sm.acc[k] = f2{0.0f, 0.0f};
        (&sm.accTail[0][0])[t] = f2{0.0f, 0.0f};
        if(t < 16u) { sm.hp[0][t] = 0.0f; sm.hp[1][t] = 0.0f; sm.hp[0][80u + t] = 0.0f; sm.hp[1][80u + t] = 0.0f; }
    }
    __syncthreads();

    for(uint32_t v = vBegin; v < vEnd; ++v)
    {
        // The thread index is re-derived per voice (and again per phase) behind an opaque move:
        // addresses and frame numbers built from it are then a few VALU per voice instead of loop
        // invariants that the allocator, held to 128 registers, would keep in scratch.
        uint32_t t = t0;
        asm volatile("" : "+v"(t));
        uint32_t lane = t & 63u;
        // profiling aid (OALGPU_PHASE_TIMES): s_memtime stamps per voice and phase
        auto stamp = [&](int slot)
        {
            if(L.phaseTimes && t0 == 0) L.phaseTimes[size_t{v} * 8 + slot] = __builtin_readcyclecounter();
        };
        stamp(0);
        // ---------------- this voice's state, requested one voice ago, is parked in LDS ----------------
        const VoiceHead head = headN;
        const VoiceTail tail = tailN;
        const BufferItem buf = bufN;
        const SrcPlan plan = planN;
        const bool looping = loopingN, dirty = dirtyN;
        if(plan.prefetch)
        {   // mPrevSamples first, the window over its second half (same wavefront for both: LDS order)
            const bool isShort = buf.fmt == OALGPU_FMT_SHORT;
            if(t < uint32_t(kMaxPad)) sm.rd[t] = prevN;
#pragma unroll
            for(int i = 0; i < kPreB; ++i) sm.rd[kMaxEdge + t + uint32_t(kBT) * uint32_t(i)] = GatherDecode(preN[i], isShort);
        }
        if(t < uint32_t(kHist))
        {
            sm.in[t] = histN;
            sm.hp[0][16u + t] = hN.x; sm.hp[1][16u + t] = hN.y;
            if(dirty) { sm.coldP[0][64u + t] = oldN.x; sm.coldP[1][64u + t] = oldN.y; }
        }
        if(t < 32u) sm.fst[t] = fstN;
        const bool haveNext = v + 1u < vEnd;
        if(haveNext)
        {   // the next voice's control line: in flight while this one resamples
            headN = LoadHeadScalar(L.ctl + v + 1u); bufN = LoadCtlBufferScalar(L.ctl + v + 1u); tailN = LoadTailScalar(L.ctl + v + 1u);
        }
        __syncthreads();
        stamp(1);

        const int vstate = head.playState;
        const bool mixes = vstate == OALGPU_VOICE_PLAYING || vstate == OALGPU_VOICE_STOPPING;
        const bool playing = vstate == OALGPU_VOICE_PLAYING;
        bool active = mixes && head.step >= 1u;
        // voice.cpp:1002-1010
        if(mixes && !active && !playing && t == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
        uint32_t outPos = 0;
        if(active && (head.flags & kFlagDelayed))
        {   // delayed start, voice.cpp:1023-1046 (see voice_wave.hip): samples at offset outPos behind zeros
            const uint32_t d = L.startDelay[v];
            __syncthreads();
            if(!playing)
            {
                if(t == 0) { L.ctl[v].playState = OALGPU_VOICE_STOPPED; L.ctl[v].flags = head.flags & ~kFlagDelayed; L.startDelay[v] = 0u; }
                active = false;
            }
            else if(d >= N) { if(t == 0) L.startDelay[v] = d - N; active = false; }
            else
            {
                outPos = d;
                if(t == 0) L.startDelay[v] = 0u;
                for(uint32_t k = t; k < outPos; k += kBT) sm.in[kHist + k] = 0.0f;
                __syncthreads();
            }
        }

        int32_t bufferItem = head.curBuffer;
        uint32_t counter = 0, fademix = 0, todo = 0;
        float endGain = 0.0f, gainAfterBlend = 0.0f;
        bool oldPass = false;
        if(active)
        {
            // ---- LoadResampledSamples, voice.cpp:642-824
            const uint32_t increment = head.step, fracPos = head.positionFrac;
            const bool single = plan.prefetch && plan.bdst == N;          // one chunk, its window parked above
            if(single)
            {
                float *mixing = sm.in + kHist;
                const int kind = head.rsKind;
                const uint32_t tableKey = head.rsFilterOffset * 8u + uint32_t(kind);
                const bool staged = (kind == 2 || kind == 3) && sm.tabKey == tableKey;
                const uint32_t sM = kind == 2 ? 4u : head.rsM, sL = kind == 2 ? 1u : head.rsL;
                if(increment == kFracOne && fracPos == 0)
                {   // voice.cpp:764-769
                    for(uint32_t k = t; k < N; k += kBT) mixing[k] = sm.rd[kMaxEdge + k];
                }
                else if(staged)
                {
                    ResampleRunBlockM<kBT>(sm, sm.rd + (kMaxEdge - sL), sM, fracPos, increment, N, mixing, t);
                }
                else RareResampleAt(&sm, L.tables, head, N, t);
                // voice.cpp:772-785: history for the next update, taken at the end-of-mix position
                if(playing && t < uint32_t(kMaxPad))
                {
                    const uint32_t srcOffset = uint32_t((uint64_t{N} * increment + fracPos) >> kFracBits);
                    L.prev[size_t{v} * kMaxPad + t] = sm.rd[srcOffset + t];
                }
            }
            else if(wave == 0)
                RareLoadResampled(&sm, L, v, lane, head, playing, N - outPos, bufferItem, looping, plan, outPos);
        }
        __syncthreads();                      // sm.in[64..] complete; sm.rd dead
        stamp(2);

        asm volatile("" : "+v"(t));
        lane = t & 63u;

        if(active)
        {
            if(head.flags & kFlagAmbiScale)
            {   // VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091
                if(wave == 0) RareAmbiScale(&sm, &L.ambi[v], N - outPos, lane, outPos);
                __syncthreads();
            }
            counter = (head.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093

            // ---- DoFilters, direct path (voice.cpp:255-267): in place on sm.in[kHist..]
            BlockDoFilters(sm, &L.dfilt[size_t{v} * 2], (head.flags & kFlagDirectFilter) != 0, sm.in + kHist + outPos, N - outPos, t,
                outPos == 0u);
            __syncthreads();
            stamp(3);

            // ---- DoHrtfMix, voice.cpp:827-902
            asm volatile("" : "+v"(t));
            if(playing && t < uint32_t(kHist)) L.hist[size_t{v} * kHist + t] = sm.in[N + t];

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
            const uint32_t odL = tail.oldDelay[0], odR = tail.oldDelay[1];
            const uint32_t dL = tail.tgtDelay[0], dR = tail.tgtDelay[1];
            const float oldStep = fademix ? oldGain / float(fademix) : 0.0f;
            // Old == Target (coefficients and delays) unless the parameter side replaced the target
            // since the last mix: the fade-out and fade-in of MixHrtfBlend then share taps and
            // their gains are summed (voice_wave.hip)
            const bool merged = !dirty;
            oldPass = !merged && oldOn;

            // x'[i] = (In[64 - dL + i], In[64 - dR + i]) * g(i), planar per ear: frame i at
            // xp[e][68 (i/64 + 1) + i%64]; row 0 = the 64 zero frames in front
            {
                const float *inL = sm.in + (kHist - dL), *inR = sm.in + (kHist - dR);
                const float gbase = gainAfterBlend - mainStep * float(fademix);
                float xl[4], xr[4], xol = 0.0f, xor_ = 0.0f;
#pragma unroll
                for(int j = 0; j < 4; ++j)
                {
                    const uint32_t i = t + uint32_t(kBT) * uint32_t(j);
                    xl[j] = inL[i]; xr[j] = inR[i];            // i <= 1023: inside sm.in for any N
                }
                if(oldPass && t < fademix)
                {
                    const float g = oldStep * float(fademix - t);
                    xol = sm.in[kHist - odL + t] * g; xor_ = sm.in[kHist - odR + t] * g;
                }
#pragma unroll
                for(int j = 0; j < 4; ++j)
                {
                    const uint32_t i = t + uint32_t(kBT) * uint32_t(j);
                    float g;
                    if(i < 64u)
                    {   // the only frames the fade can touch (fademix <= 64)
                        if(i < fademix)
                        {
                            g = newOn ? newStep * float(i) : 0.0f;
                            if(merged && oldOn) g += oldStep * float(fademix - i);
                        }
                        else g = gainAfterBlend + mainStep * float(i - fademix);
                    }
                    else g = __builtin_fmaf(mainStep, float(i), gbase);
                    const bool in = i < N;
                    const uint32_t at = uint32_t(kXpStride) * ((i >> 6) + 1u) + (i & 63u);
                    sm.xp[0][at] = in ? xl[j] * g : 0.0f;
                    sm.xp[1][at] = in ? xr[j] * g : 0.0f;
                }
                if(t < 64u)
                {
                    sm.xp[0][t] = 0.0f; sm.xp[1][t] = 0.0f;
                    if(oldPass) { sm.xoP[0][t] = xol; sm.xoP[1][t] = xor_; }
                }
            }
            __syncthreads();

        }
        // ---------------- the next voice's source window, history and HRIR leave HBM now ----------------
        // (registers from here to the parking at the top of the next pass: only the matrix-pipe phase,
        // whose register needs are small, runs in between)
        stamp(4);
        asm volatile("" : "+v"(t));
        if(haveNext) request(v + 1u, t);
        if(active)
        {
            // ---- the FIR: tile `wave` of both ears on the matrix pipe, folded into the LDS accumulator
            asm volatile("" : "+v"(t));
            lane = t & 63u;
            {
                f4 accM[2][2], accT[2];
#pragma unroll
                for(int e = 0; e < 2; ++e)
                {
                    accM[e][0] = f4{0.0f, 0.0f, 0.0f, 0.0f}; accM[e][1] = f4{0.0f, 0.0f, 0.0f, 0.0f};
                    accT[e] = f4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                FirMfmaTile<kXpStride>(accM, accT, sm.xp[0], sm.xp[1], sm.hp[0], sm.hp[1], lane, wave);
                const uint32_t rr = lane & 15u, q4 = lane >> 4;
                f2 *am = &sm.acc[256u * q4 + 16u * wave + rr];       // element i: frame 64 (4 q4 + i) + 16 wave + rr
                f2 cur[4];
#pragma unroll
                for(int i = 0; i < 4; ++i) cur[i] = am[64 * i];
#pragma unroll
                for(int i = 0; i < 4; ++i)
                    am[64 * i] = f2{cur[i].x + (accM[0][0][i] + accM[0][1][i]), cur[i].y + (accM[1][0][i] + accM[1][1][i])};
                if(lane < 16u)
                {
                    f2 *at = &sm.accTail[wave][lane];
#pragma unroll
                    for(int i = 0; i < 4; ++i) { const f2 c = at[16 * i]; at[16 * i] = f2{c.x + accT[0][i], c.y + accT[1][i]}; }
                }
            }
            stamp(5);
            if(oldPass)
            {   // frame f of ear e receives cOld[f - i] * xo[i], i < 64 (a replaced filter's fade-out)
                const uint32_t e = wave >> 1, f = t & 127u;
                const float *co = &sm.coldP[e][64u + f];
                const float *xo = sm.xoP[e];
#pragma unroll 8
                for(int i = 0; i < 64; ++i) accO = __builtin_fmaf(co[-i], xo[i], accO);
            }

            // ---- voice.cpp:1094-1101 / :869-873,900: Old <- Target, Old.Gain <- reached gain
            if(dirty && (counter == 0 || fademix) && t < irStride)
            {   // (the target's taps are in sm.hp; zero past IrSize like the stored filter)
                f2 *od = reinterpret_cast<f2*>(L.hrtfOld + size_t{v} * irStride * 2);
                od[t] = f2{sm.hp[0][16u + t], sm.hp[1][16u + t]};
            }
            // ---- voice.cpp:1116-1232: flags, position, loop wrap / end of buffer ----
            if(t == 0)
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
        stamp(6);
        __syncthreads();                      // xp / hp / coldP / xoP free for the next voice's parking
        stamp(7);
    }

    // ---- one partial per workgroup: tiles -> frames, the four ring-out parts in wavefront order
    {
        const uint32_t t = t0, lane = t0 & 63u;
        (void)lane;
        float *oo = sm.in;                               // [ear][frame < 128] old-filter sums
        oo[t] = accO;
        __syncthreads();
        f2 *ph = reinterpret_cast<f2*>(L.partHrtf) + size_t{group} * (kLine + kHrirLen);
        for(uint32_t k = t; k < uint32_t(kLine + kHrirLen); k += kBT)
        {
            f2 s = {0.0f, 0.0f};
            if(k < uint32_t(kLine)) s = sm.acc[k];
            else if(k < uint32_t(kLine) + 64u)
            {   // the four ring-out parts in wavefront order
                s = sm.accTail[0][k - kLine];
#pragma unroll
                for(int w = 1; w < kBW; ++w) { const f2 o = sm.accTail[w][k - kLine]; s.x += o.x; s.y += o.y; }
            }
            if(k < 128u) { s.x += oo[k]; s.y += oo[128u + k]; }
            ph[k] = s;
        }
    }
}

} // namespace




uint32_t BlockKernelGroups(const DeviceLayout &L)
{ return (L.numVoices + L.blockVoices - 1u) / L.blockVoices; }

int BlockKernelOccupancy(int minw)
{   // resident workgroups per CU according to the runtime (profiling aid)
    int n = -1;
    if(minw == 3) (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, VoiceBlockKernel<3>, kBT, 0);
    else (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, VoiceBlockKernel<4>, kBT, 0);
    return n;
}

hipError_t LaunchVoiceBlock(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo)
{
    static const bool report = std::getenv("OALGPU_REPORT_OCCUPANCY") != nullptr;
    static bool done = false;
    if(report && !done)
    {
        done = true;
        std::fprintf(stderr, "VoiceBlockKernel<%u>: %d workgroups per CU, %u workgroups, %u voices each\n", L.blockWaves,
            BlockKernelOccupancy(int(L.blockWaves)), BlockKernelGroups(L), L.blockVoices);
    }
    if(L.blockWaves == 3) hipLaunchKernelGGL(VoiceBlockKernel<3>, dim3(BlockKernelGroups(L)), dim3(kBT), 0, s, L, samplesToDo);
    else hipLaunchKernelGGL(VoiceBlockKernel<4>, dim3(BlockKernelGroups(L)), dim3(kBT), 0, s, L, samplesToDo);
    return hipGetLastError();
}

} // namespace oalgpu
