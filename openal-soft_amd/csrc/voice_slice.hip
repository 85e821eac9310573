// Dry lines and sends with MANY mix lines (BASELINE configs[3]: 5 dry + 4 x 4 wet = 21 lines): a wavefront per FRAME SLICE.
//
//   "for every Playing|Stopping voice: voice->mix(...)"   alc/alu.cpp:2201-2206
//      -> Voice::mix                                       core/voice.cpp:988-1233
//      -> LoadResampledSamples / Resample_*                voice.cpp:642-824
//      -> DoFilters (direct path, every send)              voice.cpp:255-267, :966-983
//      -> MixSamples onto the dry lines / a slot's lines   voice.cpp:934-984, mixer/mixer_c.cpp:150-186
//
// The wavefront-per-voice kernel (voice_wave.hip) keeps a voice's 1024 output frames in one wavefront; N mix lines then need
// N x 1024 frames of accumulator per wavefront -- 16 N registers per lane -- and beyond six lines that does not exist beside the
// resampler, so that kernel leaves a 4 KB stream row per mixed signal in HBM and mixes the rows in its tail (config 4: 64 MB of
// rows written and read back per launch, 6.6 x the algorithmic traffic).  This kernel cuts the update the other way: the four
// wavefronts of a workgroup own the four 256-frame SLICES of the update, and every wavefront walks ALL of the workgroup's
// voices for its slice.  Lane l owns frames 4 l .. 4 l + 3 of its slice on every line: 4 registers per line, 24 lines in 96
// registers, for the whole launch -- nothing is dumped, summed across wavefronts or stored until the very end, when each
// wavefront stores its own quarter of the workgroup's partial bus.  What makes the cut possible:
//   * the resampler is a pure function of the source and the output's position: output frame f of the update reads the source
//     at position + ((frac + f step) >> 16), so a slice is resampled from its own window of the source (the window of the
//     update, core/voice.cpp:662-753, is contiguous; a slice's window is a part of it -- FillWindow below restates what every
//     element of the update's window is, for static, looping, queued, ended and not-yet-started sources, chunk by chunk where
//     the reference loads the window in chunks: what lies past a source's END depends on the chunk that loaded it);
//   * the filters (DoFilters, the ambisonic splitter) are recurrences over the frames: their state goes from the wavefront of
//     slice k to the wavefront of slice k + 1 through an LDS mailbox (the four wavefronts become a pipeline, skewed by one
//     filter run; three voices in four have no active filter and need no hand-over at all);
//   * gains ramp over the first <= 64 frames only (MixLine's Counter, voice.cpp:1093): slice 0 alone sees the ramp;
//   * what does not depend on the slice -- a voice's plan, the gains of all its lines (ONE PrepareMixLine per voice, line =
//     lane), Gains.Current as the update leaves it -- is worked out once per workgroup, by one wavefront per voice, in front of
//     every block of 16 voices, and read by the four slice wavefronts from LDS records (SliceRec);
//   * everything that changes a voice's state for the NEXT update -- position, play state, mPrevSamples, flags -- is written
//     after a barrier at the end, when no wavefront reads this update's state any more.
// The four wavefronts do the same amount of work by construction (the same voices, a quarter of the frames each), and the kernel's
// time is linear in the voices per workgroup.  Opt-in (OALGPU_CTX_SLICE_LINES): a third of the stream-row kernel's HBM traffic
// and half again its instructions -- both kernels are bound by instruction issue.
// FAST arithmetic (FMA, own summation order); integer state bit-exact.  Near-field control and HRTF stay with voice_wave.hip.
#include <hip/hip_ext.h>
#include "wave_common.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kSl = 256;                      // frames per slice
constexpr int kSlPre = 9;                     // prefetched window: registers per lane
constexpr int kSlWin = kSlPre * 64;           // source samples a slice's window holds (pitch <= ~2.1 in one piece)
constexpr int kSlRd = kMaxEdge + kSlWin + 8;  // 608 floats = 9.5 x 64: rd2 starts half the banks away from rd
static_assert(kSlRd % 64 == 32, "rd2 sits half the banks away from rd");
constexpr int kSlBlock = 16;                  // voices between two workgroup barriers (mailbox slots)
constexpr int kSlStages = 8;                  // recurrences with a hand-over: direct filter, six sends, ambisonic splitter
constexpr int kStageAmbi = 7;

struct alignas(16) SliceWaveLds {
    float rd[kSlRd];                          // [24 samples in front of the slice | its window]
    float rd2[kSlRd];                         // rd2[i] = rd[i + 1] (aligned pair reads at odd positions, see wave_common.hpp)
    float in[kSl + 8];                        // the slice of the voice's row (resampled, then direct-filtered in place)
    float tmp[kSl + 8];                       // a send's own filtered copy
    float fst[32];                            // the filter pair being run (2 x 16 dwords)
    float st[kSlStages - 1][32];              // the voice's filter pairs as memory holds them (its first slice's wavefront): direct, sends
    float misc[64];
    int32_t best;
    uint32_t pad[3];
};

// What a voice's update is, worked out ONCE per workgroup (by one wavefront, in front of the block's voices) for the four
// wavefronts that mix its slices: the plan, and the gains of all lines resolved (line = lane)
struct SliceRec {
    uint32_t w[16];                           // see RecWord
    float gain[32], cur[32], step[32];        // per line: the gain behind the ramp; the ramp's start and step (slice 0 only)
    uint32_t fade[32];                        // per line: frames the ramp covers (0: none)
    int32_t rowId[32];                        // per line: the signal it is mixed from (-1 none, 0 unfiltered row, 1 direct-filtered, 2 + send)
};
enum RecWord : int { kRwFlags = 0, kRwOutPos, kRwBufferItem, kRwBsrcFull, kRwSfBits, kRwHaveSend, kRwPending, kRwOrder };
enum RecFlag : uint32_t { kRfActive = 1u, kRfPlaying = 2u, kRfLooping = 4u, kRfQueue = 8u, kRfAnyFull = 16u, kRfMulti = 32u, kRfDirectFilter = 64u };

struct SliceWgLds {
    SliceWaveLds w[kWWaves];
    SliceRec rec[kSlBlock];
    alignas(16) f2 tabF[kTabPairs * 32];      // [tap pair][phase], as in voice_wave.hip
    f2 tabP[kTabPairs * 32];
    uint32_t tabKey, tabM, tabL, pad;
    float mail[kSlBlock][kSlStages][32];      // a recurrence's state behind slice k of voice j, for slice k + 1
    uint32_t flag[kSlBlock][kSlStages];       // 8 x block + slices done
};

struct RowGain {                              // a row's merged gains over the mix lines, line = lane (RowLineGain of voice_wave.hip)
    float gain{0.0f}, cur{0.0f}, step{0.0f};
    uint32_t fadeLen{0};
    __device__ __forceinline__ void add(const MixLineGain &g)
    {
        const float constant = g.steady ? g.tgt : 0.0f;
        gain += constant;
        if(g.fadeLen) { cur += g.cur; step += g.step; fadeLen = g.fadeLen; }
        else cur += constant;
    }
};

// MixSamples of one row's slice onto the lines (mixer_c.cpp:150-186): lane l holds frames 4 l .. 4 l + 3 of the slice.
// row: kSl floats, zeros where the voice has no samples.  ramp: this is slice 0 (frames below a line's fadeLen use cur + step f).
template<int ACCN>
__device__ __forceinline__ void MixSlice(float (&acc)[ACCN][4], const float *row, const RowGain &rg, bool ramp, uint32_t lane)
{
    const bool hasRamp = ramp && rg.fadeLen != 0u && lane < uint32_t(ACCN);
    const bool nz = lane < uint32_t(ACCN) && (rg.gain != 0.0f || (hasRamp && (rg.cur != 0.0f || rg.step != 0.0f)));
    const unsigned long long live = __ballot(nz);
    if(live == 0ull) return;
    const unsigned long long ramped = __ballot(hasRamp);
    const f4 xv = *reinterpret_cast<const f4*>(row + 4u * lane);
    const float x[4] = {xv.x, xv.y, xv.z, xv.w};
    static_assert(ACCN % 8 == 0, "lines come in blocks of 8");
#pragma unroll
    for(int blk = 0; blk < ACCN / 8; ++blk)
    {
        if(!((live >> (8 * blk)) & 0xffull)) continue;      // (a signal feeds few lines: whole blocks of 8 are skipped)
#pragma unroll
        for(int cc = 0; cc < 8; ++cc)
        {
            constexpr int dummy = 0; (void)dummy;
            const int c = 8 * blk + cc;
            if(!((live >> c) & 1ull)) continue;             // (uniform)
            const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rg.gain), c));
            if((ramped >> c) & 1ull)
            {
                const uint32_t fl = uint32_t(__builtin_amdgcn_readlane(int(rg.fadeLen), c));
                const float cu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rg.cur), c));
                const float st = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rg.step), c));
#pragma unroll
                for(int r = 0; r < 4; ++r)
                {
                    const uint32_t f = 4u * lane + uint32_t(r);
                    const float gr = (f < fl) ? __builtin_fmaf(st, float(f), cu) : g;
                    acc[c][r] = __builtin_fmaf(x[r], gr, acc[c][r]);
                }
            }
            else
            {
#pragma unroll
                for(int r = 0; r < 4; ++r) acc[c][r] = __builtin_fmaf(x[r], g, acc[c][r]);
            }
        }
    }
}

// CalculateBufferSize (voice.cpp:600-640) for a piece of a slice: how many outputs the window's capacity allows at once
__device__ __forceinline__ void SliceBufferSize(uint32_t fracPos, uint32_t increment, uint32_t dstRemaining, uint32_t &dst, uint32_t &src)
{
    constexpr uint32_t srcMax = uint32_t(kSlWin - kMaxEdge);          // (the window holds the 24 samples in front of it too)
    const uint32_t ext = increment <= kFracOne ? 1u : 0u;
    const uint64_t srcSize = ((uint64_t{dstRemaining - ext} * increment + fracPos) >> kFracBits) + ext + kMaxEdge;
    if(srcSize <= srcMax) { dst = dstRemaining; src = uint32_t(srcSize); return; }
    uint64_t dstSize = ((uint64_t{srcMax - kMaxEdge} << kFracBits) - fracPos) / increment;
    if(dstSize >= dstRemaining) { dst = dstRemaining; src = srcMax; return; }
    dstSize &= ~uint64_t{3};
    dst = dstSize ? uint32_t(dstSize) : 1u;
    src = srcMax;
}

// What a voice's update is made of, decided from its control line alone (uniform over the workgroup: every wavefront reads the
// same line and nothing writes it before the final barrier).
struct VoicePlan {
    bool active, playing, looping, queue, anyFull;
    uint32_t outPos;                          // a delayed start's first output frame (voice.cpp:1023-1046)
    int32_t bufferItem;
    uint32_t bsrcFull;                        // the update's first chunk as the reference sizes it (the hold rule reads it)
    bool multi;                               // the reference loads the update's window in more than one chunk
};

// ---- the update's source window, element by element -----------------------------------------------------------------
// W(A), A >= 0: element A of mResampleData (relative to the update's start) as LoadResampledSamples builds it
// (voice.cpp:662-753): A < 24 -> mPrevSamples[A]; then the source from the voice's position on -- zeros in front of a
// negative position (:679-697), the buffer with its loop (LoadBufferStatic, :500-544), the queue (LoadBufferQueue, :563-594),
// the last sample held past the end; a voice without a buffer holds the sample nearest zero of what mPrevSamples has ahead of
// the position (:704-719).
// The reference loads the window in chunks of at most kResampleDataSize samples (CalculateBufferSize, :600-640: pitches above
// ~1.27 at 1024 frames), and what a chunk puts past the END of a source depends on the chunk: LoadBufferStatic / LoadBufferQueue
// hold the last sample they loaded IN THAT CALL -- a chunk that begins at or past the end holds 0.  So the outputs of chunk c
// see: A < S_c + 24 what chunk c - 1 left there (the 24 samples carried over, :807-810), from there on what chunk c's own load
// put there.  anchor = S_c, the chunk's source offset from the update's start; anchorPrev = S_(c-1) (unused for chunk 0, where
// A < 24 is mPrevSamples).  dst[i] = W(A0 + i) as chunk c's outputs see it, i < cnt, by one wavefront.
template<class LT>
__device__ __forceinline__ void FillWindow(float *dst, uint32_t A0, uint32_t cnt, const LT &L, uint32_t v, const VoiceHead &h,
    const BufferItem &buf, const VoicePlan &p, SliceWaveLds &w, uint32_t lane, uint32_t anchor = 0u, uint32_t anchorPrev = 0u)
{
    const float *prev = L.prev + size_t{v} * kMaxPad;
    const uint32_t nPrev = A0 < uint32_t(kMaxEdge) ? ((uint32_t(kMaxEdge) - A0 < cnt) ? uint32_t(kMaxEdge) - A0 : cnt) : 0u;
    for(uint32_t i = lane; i < nPrev; i += 64u) dst[i] = prev[A0 + i];
    if(nPrev == cnt) return;
    if(p.bufferItem < 0)
    {   // voice.cpp:704-719 (decided by the update's first chunk: later ones find the held sample everywhere they look)
        const uint32_t avail = p.bsrcFull < uint32_t(kMaxEdge) ? p.bsrcFull : uint32_t(kMaxEdge);
        if(lane < uint32_t(kMaxEdge)) w.misc[lane] = prev[kMaxEdge + lane];
        WaveSync();
        if(lane == 0)
        {
            uint32_t best = 0;
            for(uint32_t i = 1; i < avail; ++i)
                if(fabsf(w.misc[i]) < fabsf(w.misc[best])) best = i;
            w.best = int32_t(best);
        }
        WaveSync();
        const uint32_t best = uint32_t(w.best);
        const float hold = w.misc[best];
        for(uint32_t i = nPrev + lane; i < cnt; i += 64u)
        {
            const uint32_t A = A0 + i;
            dst[i] = (A <= uint32_t(kMaxEdge) + best) ? prev[A] : hold;
        }
        WaveSync();
        return;
    }
    // [nPrev, cnt): first what the chunk before left (A < anchor + 24), then the chunk's own load
    uint32_t i0 = nPrev;
    for(int part = 0; part < 2 && i0 < cnt; ++part)
    {
        uint32_t i1 = cnt, from = anchor;
        if(part == 0)
        {
            const uint64_t edge = uint64_t{anchor} + uint32_t(kMaxEdge);
            if(anchor == 0u || uint64_t{A0} + i0 >= edge) continue;
            i1 = (edge - A0 < cnt) ? uint32_t(edge - A0) : cnt;
            from = anchorPrev;
        }
        const uint32_t e0 = A0 + i0 - uint32_t(kMaxEdge);              // source samples from the voice's position
        const uint32_t m = i1 - i0;
        float *d = dst + i0;
        i0 = i1;
        const int64_t p0 = int64_t{h.position} + int64_t{e0};
        uint32_t z = 0;
        if(p0 < 0) { const uint64_t need = uint64_t(-p0); z = need < m ? uint32_t(need) : m; }
        for(uint32_t i = lane; i < z; i += 64u) d[i] = 0.0f;
        if(z == m) continue;
        const uint64_t upos64 = uint64_t(p0 + int64_t{z});
        const uint32_t upos = upos64 > 0xffffffffull ? 0xffffffffu : uint32_t(upos64);
        d += z;
        const uint32_t n = m - z;
        // where the load that put these samples there began (a negative position: at the buffer's start, voice.cpp:679-697)
        const int64_t c0 = int64_t{h.position} + int64_t{from};
        const uint64_t callPos = c0 > 0 ? uint64_t(c0) : 0ull;
        if(p.queue)
        {   // LoadBufferQueue: crawl from the voice's current item; past the queue's end the last sample THAT call loaded is held
            int32_t item = p.bufferItem;
            uint64_t total = 0;                         // samples of the items visited (a queue that ends: of the whole rest of it)
            float last = 0.0f;
            uint32_t dataPos = upos, done = 0;
            for(uint32_t guard = 0; item >= 0 && done < n && guard < 4096u; ++guard)
            {
                const BufferItem b = L.buffers[item];
                total += b.sampleLen;
                if(b.sampleLen) last = LoadSampleAny(b.fmt, b.data, size_t{b.sampleLen - 1u} * b.frameStep);
                if(dataPos >= b.sampleLen) dataPos -= b.sampleLen;
                else
                {
                    const uint32_t rem = (n - done < b.sampleLen - dataPos) ? n - done : b.sampleLen - dataPos;
                    for(uint32_t k = lane; k < rem; k += 64u) d[done + k] = LoadSampleAny(b.fmt, b.data, size_t{dataPos + k} * b.frameStep);
                    done += rem;
                    dataPos = 0;
                }
                item = b.next > 0 ? b.next - 1 : h.loopBuffer;
            }
            if(done < n)
            {   // the queue ended: the call held its last sample if it had loaded any, i.e. if it began in front of the end
                const float hold = callPos < total ? last : 0.0f;
                for(uint32_t k = done + lane; k < n; k += 64u) d[k] = hold;
            }
            continue;
        }
        if(p.looping) { FillFromBuffer<64>(d, n, buf, true, upos, lane); continue; }
        // LoadBufferStatic without a loop: past the end the buffer's last sample, if the call began in front of the end
        const bool any = uint64_t{buf.sampleLen} > callPos;
        const uint32_t lastIdx = buf.sampleLen - 1u;
        const float hold = any ? LoadSampleAny(buf.fmt, buf.data, size_t{lastIdx} * buf.frameStep) : 0.0f;
        for(uint32_t k = lane; k < n; k += 64u)
        {
            const uint64_t idx64 = uint64_t{upos} + k;
            d[k] = idx64 < buf.sampleLen ? LoadSampleAny(buf.fmt, buf.data, size_t(idx64) * buf.frameStep) : hold;
        }
    }
}

// a slice's share of the update for one voice
struct SlicePlan {
    uint32_t cnt, dstOff;                     // voice samples in the slice; where they start in it
    uint32_t srcOff, frac;                    // the slice's first output: source offset from the update's position, fraction
    uint32_t bdst, bsrc;                      // its first piece
    bool pref;                                // that piece is gathered into registers one voice ahead
    uint32_t nPrev, upos;                     // pref: elements out of mPrevSamples; the buffer position of the first gathered element
};

template<class LT>
__device__ __forceinline__ VoicePlan PlanVoice(const LT &L, uint32_t v, const VoiceHead &h, const BufferItem &buf, uint32_t N)
{
    VoicePlan p{};
    const bool mixes = h.playState == OALGPU_VOICE_PLAYING || h.playState == OALGPU_VOICE_STOPPING;
    p.playing = h.playState == OALGPU_VOICE_PLAYING;
    p.active = mixes && h.step >= 1u;
    p.outPos = 0;
    if(p.active && (h.flags & kFlagDelayed))
    {
        const uint32_t d = __builtin_amdgcn_readfirstlane(L.startDelay[v]);
        if(!p.playing || d >= N) p.active = false;
        else p.outPos = d;
    }
    p.bufferItem = h.curBuffer;
    p.queue = (h.flags & kFlagQueue) != 0;
    p.looping = false;
    p.anyFull = false;
    if(h.curBuffer >= 0)
    {   // voice.cpp:1015-1019: a position at or past the loop end plays on without looping
        p.looping = h.loopBuffer >= 0 && (p.queue || !(h.position >= 0 && uint32_t(h.position) >= buf.loopEnd));
        p.anyFull = buf.sampleLen > (h.position > 0 ? uint32_t(h.position) : 0u);
    }
    uint32_t bdst = 0;
    p.bsrcFull = 0;
    if(p.active) CalcBufferSize(h.positionFrac, h.step, N - p.outPos, bdst, p.bsrcFull);
    p.multi = p.active && bdst < N - p.outPos;
    return p;
}

__device__ __forceinline__ SlicePlan PlanSlice(const VoiceHead &h, const BufferItem &buf, const VoicePlan &p, uint32_t f0, uint32_t nS)
{
    SlicePlan s{};
    if(!p.active || nS == 0u) return s;
    const uint32_t lo = f0 > p.outPos ? f0 : p.outPos;
    if(lo >= f0 + nS) return s;
    s.cnt = f0 + nS - lo;
    s.dstOff = lo - f0;
    const uint64_t t0 = uint64_t{h.positionFrac} + uint64_t{lo - p.outPos} * h.step;
    s.srcOff = uint32_t(t0 >> kFracBits);
    s.frac = uint32_t(t0) & kFracMask;
    SliceBufferSize(s.frac, h.step, s.cnt, s.bdst, s.bsrc);
    // the plain shape: a static float / int16 buffer, the whole slice in one piece, at most one loop wrap inside the window
    s.pref = false;
    if(p.bufferItem >= 0 && !p.queue && !p.multi && h.position >= 0 && s.bdst == s.cnt && (buf.fmt == OALGPU_FMT_FLOAT || buf.fmt == OALGPU_FMT_SHORT))
    {
        s.nPrev = s.srcOff < uint32_t(kMaxEdge) ? uint32_t(kMaxEdge) - s.srcOff : 0u;
        const uint32_t total = uint32_t(kMaxEdge) + s.bsrc;
        if(s.nPrev > total) s.nPrev = total;
        const uint64_t up = uint64_t(uint32_t(h.position)) + (s.srcOff + s.nPrev - uint32_t(kMaxEdge));
        if(up <= 0x7fffffffull)
        {
            uint32_t upos = uint32_t(up);
            const uint32_t m = total - s.nPrev;
            bool ok = true;
            if(p.looping)
            {
                const uint32_t ls = buf.loopStart, le = buf.loopEnd, size = le - ls;
                if(upos >= le) upos = ((upos - ls) % size) + ls;
                ok = m <= (le - upos) + size;
            }
            else ok = p.anyFull;
            s.upos = upos;
            s.pref = ok;
        }
    }
    return s;
}

// the gather of a plain slice's window into registers (GatherStatic of wave_common.hpp with the update's `any`)
__device__ __forceinline__ void GatherSlice(float (&pre)[kSlPre], uint32_t count, const BufferItem &b, bool looping, uint32_t dataPos, uint32_t lane)
{
    if(GatherIsLinear<kSlPre, 64>(count, b, looping, dataPos))
    {
        if(b.fmt == OALGPU_FMT_FLOAT) GatherLinearT<OALGPU_FMT_FLOAT, kSlPre, 64>(pre, b, dataPos, lane);
        else GatherLinearT<OALGPU_FMT_SHORT, kSlPre, 64>(pre, b, dataPos, lane);
        return;
    }
    const uint32_t fs = b.frameStep;
    const bool isShort = b.fmt == OALGPU_FMT_SHORT;
    const uint32_t lastIdx = b.sampleLen - 1u;
    const uint32_t ls = b.loopStart, first = looping ? b.loopEnd - dataPos : 0xffffffffu;
#pragma unroll
    for(int i = 0; i < kSlPre; ++i)
    {
        const uint32_t k = lane + 64u * uint32_t(i);
        uint32_t idx;
        if(looping) idx = (k < first) ? dataPos + k : ls + (k - first);
        else { const uint64_t q = uint64_t{dataPos} + k; idx = q < lastIdx ? uint32_t(q) : lastIdx; }
        float r = 0.0f;
        if(k < count)
            r = isShort ? LoadRawGlobal<OALGPU_FMT_SHORT>(b.data, size_t{idx} * fs) : LoadRawGlobal<OALGPU_FMT_FLOAT>(b.data, size_t{idx} * fs);
        pre[i] = r;
        __builtin_amdgcn_sched_barrier(0);          // (one element's address arithmetic at a time: ten at once cost ten register pairs)
    }
}

// BiquadWaveScan / BiquadDualWaveScan of wave_common.hpp with runs of SEG samples per lane instead of 17: a slice has 256
// samples, so runs of 5 (odd: the lanes' runs start on distinct banks) do the per-sample part of the scan in 5 steps
// instead of 17.  Same scheme: M = A^SEG, the run's forced response, run-start states by ScanLinear2, the recurrence itself.
template<int SEG>
__device__ __forceinline__ void BiquadWaveScanT(float (&x)[SEG], uint32_t cnt, const BiquadState &f, float &z1, float &z2, uint32_t lane, int lastLane)
{
    const float b0 = f.b0, b1 = f.b1, b2 = f.b2, a1 = f.a1, a2 = f.a2;
    S2 m0{1.0f, 0.0f}, m1{0.0f, 1.0f};                // columns of A^j
    S2 c{__builtin_fmaf(-a1, b0, b1), __builtin_fmaf(-a2, b0, b2)};      // A^j Bv
    S2 ea{0.0f, 0.0f}, eb{0.0f, 0.0f};
#pragma unroll
    for(int j = 0; j < SEG; ++j)
    {
        const float xv = x[SEG - 1 - j];
        if(j & 1) { eb.a = __builtin_fmaf(xv, c.a, eb.a); eb.b = __builtin_fmaf(xv, c.b, eb.b); }
        else { ea.a = __builtin_fmaf(xv, c.a, ea.a); ea.b = __builtin_fmaf(xv, c.b, ea.b); }
        c = S2{__builtin_fmaf(-a1, c.a, c.b), -a2 * c.a};
        m0 = S2{__builtin_fmaf(-a1, m0.a, m0.b), -a2 * m0.a};
        m1 = S2{__builtin_fmaf(-a1, m1.a, m1.b), -a2 * m1.a};
    }
    S2 e{ea.a + eb.a, ea.b + eb.b};
    {
        const S2 mz = Mv2(m0, m1, S2{z1, z2});
        if(lane == 0u) { e.a += mz.a; e.b += mz.b; }
    }
    e = ScanLinear2(e, m0, m1, lane);
    S2 st;
    st.a = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z1), __builtin_bit_cast(int, e.a), 0x138, 0xF, 0xF, false));
    st.b = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, z2), __builtin_bit_cast(int, e.b), 0x138, 0xF, 0xF, false));
    S2 zc = st;
#pragma unroll
    for(int i = 0; i < SEG; ++i)
    {
        const float xv = x[i];
        const float t1 = __builtin_fmaf(xv, b1, st.b);
        const float y = __builtin_fmaf(xv, b0, st.a);
        st.a = __builtin_fmaf(-y, a1, t1);
        st.b = __builtin_fmaf(xv, b2, -y * a2);
        x[i] = y;
        if(uint32_t(i) + 1u == cnt) zc = st;
    }
    z1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc.a), lastLane));
    z2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, zc.b), lastLane));
}

template<int SEG>
__device__ __forceinline__ void BiquadDualWaveScanT(BiquadState &f0, BiquadState &f1, float *buf /* in place */, uint32_t n, uint32_t lane)
{
    static_assert(SEG * 64 >= kSl && (SEG & 1), "the runs cover a slice; odd stride");
    const uint32_t begin = lane * uint32_t(SEG) < n ? lane * uint32_t(SEG) : n;
    const uint32_t cnt = (begin + uint32_t(SEG) < n) ? uint32_t(SEG) : n - begin;
    const int lastLane = int((n - 1u) / uint32_t(SEG));
    float x[SEG];
#pragma unroll
    for(int i = 0; i < SEG; ++i) x[i] = (uint32_t(i) < cnt) ? buf[begin + i] : 0.0f;
    BiquadWaveScanT<SEG>(x, cnt, f0, f0.z1, f0.z2, lane, lastLane);
    BiquadWaveScanT<SEG>(x, cnt, f1, f1.z1, f1.z2, lane, lastLane);
#pragma unroll
    for(int i = 0; i < SEG; ++i) if(uint32_t(i) < cnt) buf[begin + i] = x[i];
}

// DoFilters (voice.cpp:255-267) on one slice with the pair's state in fst (LDS, in AND out: the whole BiquadState pair)
__device__ __forceinline__ void SliceFilterRun(float *fst, float *buf, uint32_t n, uint32_t lane)
{
    BiquadState f0, f1;
    {
        const float *a = fst, *b = fst + 16;
        f0 = BiquadState{a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], __builtin_bit_cast(int32_t, a[12])};
        f1 = BiquadState{b[0], b[1], b[2], b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11], __builtin_bit_cast(int32_t, b[12])};
    }
    WaveSync();
    if(f0.counter <= 0 && f1.counter <= 0)
    {
        BiquadDualWaveScanT<5>(f0, f1, buf, n, lane);
        if(lane == 0) { fst[0] = f0.z1; fst[1] = f0.z2; fst[16] = f1.z1; fst[17] = f1.z2; }
    }
    else
    {
        if(lane == 0)
        {
            BiquadDualInterp(f0, f1, buf, buf, n);
            float *a = fst, *b = fst + 16;
            a[0] = f0.z1; a[1] = f0.z2; a[2] = f0.b0; a[3] = f0.b1; a[4] = f0.b2; a[5] = f0.a1; a[6] = f0.a2;
            a[12] = __builtin_bit_cast(float, f0.counter);
            b[0] = f1.z1; b[1] = f1.z2; b[2] = f1.b0; b[3] = f1.b1; b[4] = f1.b2; b[5] = f1.a1; b[6] = f1.a2;
            b[12] = __builtin_bit_cast(float, f1.counter);
        }
    }
    WaveSync();
}

// The kernel's argument block: the fields of DeviceLayout it reads (a by-value DeviceLayout is ~70 dwords of SGPRs, most of
// them spilled into VGPR lanes around every call -- see WaveArgsHrtf, voice_wave.hip)
struct SliceArgs {
    uint32_t numVoices, waveVoices, numDry, numSends, wetChannels, mixLines;
    const float *tables;
    const BufferItem *buffers;
    VoiceCtl *ctl;
    float *prev;
    BiquadSlot *dfilt, *sfilt;
    float *gainCur, *gainTgt, *sendCur, *sendTgt;
    AmbiScaleState *ambi;
    uint32_t *startDelay, *queueDone;
    float *partLines;
    explicit SliceArgs(const DeviceLayout &L) : numVoices{L.numVoices}, waveVoices{L.waveVoices}, numDry{L.numDry}, numSends{L.numSends},
        wetChannels{L.wetChannels}, mixLines{L.mixLines}, tables{L.tables}, buffers{L.buffers}, ctl{L.ctl}, prev{L.prev}, dfilt{L.dfilt},
        sfilt{L.sfilt}, gainCur{L.gainCur}, gainTgt{L.gainTgt}, sendCur{L.sendCur}, sendTgt{L.sendTgt}, ambi{L.ambi},
        startDelay{L.startDelay}, queueDone{L.queueDone}, partLines{L.partLines} { }
};

// Is a filter pair at rest -- states and counter zero, current == target coefficients (what WaveDoFilters checks before it
// clears an INACTIVE pair, voice.cpp:264-265)?  Decided on the pair's words in flight (lane = word, 2 x 16), as in voice_wave.hip.
// Two pairs per register (lanes 0..31, 32..63): bit 0 / bit 1 of the result = the first / second pair is at rest.
__device__ __forceinline__ uint32_t PairsAtRest(float word, uint32_t lane)
{
    const uint32_t li = lane & 15u;
    const uint32_t bits = __builtin_bit_cast(uint32_t, word);
    const float ahead = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, int(bits), 0x105, 0xF, 0xF, true));   // row_shl:5
    bool ok = true;
    if(li < 2u) ok = word == 0.0f;
    else if(li < 7u) ok = word == ahead;
    else if(li == 12u) ok = bits == 0u;
    const unsigned long long bad = __ballot(!ok);
    return ((bad & 0xffffffffull) ? 0u : 1u) | ((bad >> 32) ? 0u : 2u);
}

// A voice's control line (VoiceCtl, 128 bytes) rides in ONE vector register, lane = dword, requested two voices ahead; its fields
// come out with v_readlane where they are used.  (As scalar loads two voices ahead the lines were 56 SGPRs live across the whole
// loop body, and the body's straight-line part was 680 spill moves of 1650 VALU instructions.)
static_assert(offsetof(VoiceCtl, sendSlot) == 48 && offsetof(VoiceCtl, buf) == 96, "VoiceCtl layout");
__device__ __forceinline__ uint32_t CtlWord(uint32_t ctlv, int k) { return uint32_t(__builtin_amdgcn_readlane(int(ctlv), k)); }
__device__ __forceinline__ VoiceHead HeadFrom(uint32_t c)
{
    VoiceHead h;
    h.playState = int32_t(CtlWord(c, 0)); h.position = int32_t(CtlWord(c, 1)); h.positionFrac = CtlWord(c, 2);
    h.curBuffer = int32_t(CtlWord(c, 3)); h.loopBuffer = int32_t(CtlWord(c, 4)); h.step = CtlWord(c, 5);
    h.rsKind = int32_t(CtlWord(c, 6)); h.rsM = CtlWord(c, 7); h.rsL = CtlWord(c, 8);
    h.rsSf = __builtin_bit_cast(float, CtlWord(c, 9)); h.rsFilterOffset = CtlWord(c, 10); h.flags = CtlWord(c, 11);
    return h;
}
__device__ __forceinline__ BufferItem BufFrom(uint32_t c)
{
    BufferItem b;
    b.data = reinterpret_cast<const void*>((uint64_t{CtlWord(c, 25)} << 32) | CtlWord(c, 24));
    b.fmt = int32_t(CtlWord(c, 26)); b.frameStep = CtlWord(c, 27); b.sampleLen = CtlWord(c, 28);
    b.loopStart = CtlWord(c, 29); b.loopEnd = CtlWord(c, 30); b.next = int32_t(CtlWord(c, 31));
    return b;
}

template<int ACCN>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(kWThreads, 2) VoiceSliceKernel(SliceArgs L, uint32_t N)
{
    __shared__ SliceWgLds sm;
    const uint32_t t = threadIdx.x;
    const uint32_t lane0 = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t group = blockIdx.x;
    SliceWaveLds &w = sm.w[wave];
    const uint32_t nvWg = kWWaves * L.waveVoices;
    const uint32_t v0 = group * nvWg;
    const uint32_t nv = v0 < L.numVoices ? ((v0 + nvWg < L.numVoices) ? nvWg : L.numVoices - v0) : 0u;
    const uint32_t f0 = uint32_t(kSl) * wave;
    const uint32_t nS = f0 < N ? ((N - f0 < uint32_t(kSl)) ? N - f0 : uint32_t(kSl)) : 0u;
    const uint32_t kLast = (N - 1u) / uint32_t(kSl);
    const uint32_t lastVoice = L.numVoices - 1u;

    // ---- the workgroup's resampler rows (voice_wave.hip's prologue without the shortcuts): those of its first voice that mixes
    // through a staged resampler
    {
        if(wave == 0)
        {
            uint32_t key = 0xffffffffu, fm = 0, fl = 0;
            for(uint32_t j0 = 0; j0 < nv && key == 0xffffffffu; j0 += 64u)
            {
                const uint32_t cand = v0 + j0 + lane0;
                bool eligible = false;
                uint32_t off = 0, m = 0, l = 0;
                int kind = 0;
                if(j0 + lane0 < nv)
                {
                    const VoiceCtl &c = L.ctl[cand];
                    kind = c.rsKind; off = c.rsFilterOffset; m = c.rsM; l = c.rsL;
                    if(kind == 2) { m = 4; l = 1; }
                    eligible = (kind == 2 || (kind == 3 && (m == 12 || m == 24 || m == 48)))
                        && (c.playState == OALGPU_VOICE_PLAYING || c.playState == OALGPU_VOICE_STOPPING);
                }
                const unsigned long long mask = __ballot(eligible);
                if(mask)
                {
                    const int firstLane = __ffsll((long long)mask) - 1;
                    key = uint32_t(__shfl(int(off * 8u + uint32_t(kind)), firstLane));
                    fm = uint32_t(__shfl(int(m), firstLane)); fl = uint32_t(__shfl(int(l), firstLane));
                }
            }
            if(lane0 == 0) { sm.tabKey = key; sm.tabM = fm; sm.tabL = fl; }
        }
        for(uint32_t i = t; i < uint32_t(kSlBlock * kSlStages); i += kWThreads) (&sm.flag[0][0])[i] = 0u;
        __syncthreads();
        const uint32_t key = sm.tabKey, m = sm.tabM;
        if(key != 0xffffffffu)
        {
            const float *filter = L.tables + (key >> 3);
            for(uint32_t idx = t; idx < (m / 2u) * 32u; idx += kWThreads)
            {
                const uint32_t p = idx >> 5, pi = idx & 31u;
                const float *row = filter + pi * 2u * m;
                sm.tabF[idx] = f2{row[2u * p], row[2u * p + 1u]};
                sm.tabP[idx] = f2{row[m + 2u * p], row[m + 2u * p + 1u]};
            }
        }
        __syncthreads();
    }

    float acc[ACCN][4];
#pragma unroll
    for(int c = 0; c < ACCN; ++c)
#pragma unroll
        for(int r = 0; r < 4; ++r) acc[c][r] = 0.0f;

    const uint32_t numDry = L.numDry, numSends = L.numSends, wetCh = L.wetChannels;
    const uint32_t sendLanes = numSends * wetCh;
    // which slot's which wet line a lane stands for in the gain vectors (line = lane: dry lines, then the slots' wet lines)
    const bool laneWet = lane0 >= numDry && lane0 < L.mixLines;
    const uint32_t laneSlot = laneWet ? (lane0 - numDry) / (wetCh ? wetCh : 1u) : 0u;
    const uint32_t laneCh = laneWet ? (lane0 - numDry) % (wetCh ? wetCh : 1u) : 0u;

    auto loadCtlV = [&](uint32_t vx, uint32_t lane) -> uint32_t
    { return lane < 32u ? reinterpret_cast<const uint32_t*>(L.ctl + (vx < L.numVoices ? vx : lastVoice))[lane] : 0u; };

    // ---- a voice's record (SliceRec): its plan; the gains of every line it feeds, resolved as MixSamples will use them
    // (voice.cpp:934-984, :1094-1112 -- ONE PrepareMixLine per voice, line = lane: a dry line takes the dry gains, a slot's wet
    // line those of the first send into the slot; further sends into the same slot are left `pending`); Gains.Current as the
    // update leaves it; which filter pairs are at rest, and the inactive ones that are not cleared (voice.cpp:264-265).
    // By ONE wavefront per voice, before the four slice wavefronts start on the block.
    struct ProLoads { uint32_t ctl; float dryTg, dryCu, sTg0, sCu0, sTg1, sCu1; };
    auto proRequest = [&](uint32_t vx, ProLoads &q, uint32_t lane)
    {
        q.ctl = loadCtlV(vx, lane);
        const size_t sb = size_t{vx} * sendLanes;
        q.dryTg = q.dryCu = q.sTg0 = q.sCu0 = q.sTg1 = q.sCu1 = 0.0f;
        if(lane < numDry) { q.dryTg = L.gainTgt[size_t{vx} * numDry + lane]; q.dryCu = L.gainCur[size_t{vx} * numDry + lane]; }
        if(lane < sendLanes) { q.sTg0 = L.sendTgt[sb + lane]; q.sCu0 = L.sendCur[sb + lane]; }
        if(lane + 64u < sendLanes) { q.sTg1 = L.sendTgt[sb + lane + 64u]; q.sCu1 = L.sendCur[sb + lane + 64u]; }
    };
    auto prepare = [&](uint32_t vx, const ProLoads &q, SliceRec &rec, uint32_t lane)
    {
        const VoiceHead h = HeadFrom(q.ctl);
        const BufferItem b = BufFrom(q.ctl);
        const VoicePlan vp = PlanVoice(L, vx, h, b, N);
        if(!vp.active) { if(lane < 16u) rec.w[lane] = 0u; return; }
        const bool playing = vp.playing;
        const bool directFilter = (h.flags & kFlagDirectFilter) != 0;
        const uint32_t counter = (h.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093
        const uint32_t sfBits = (h.flags >> kFlagSendFilterShift) & 0x3fu;
        int32_t sendSlots[6];
#pragma unroll
        for(int k = 0; k < 6; ++k) sendSlots[k] = int32_t(CtlWord(q.ctl, 12 + k));
        uint32_t haveSend = 0u;
#pragma unroll
        for(int k = 0; k < 6; ++k) if(uint32_t(k) < numSends && sendSlots[k] >= 0) haveSend |= 1u << k;
        // every slot's first send; the lanes of the slot's lines learn which
        int32_t mySend = -1;
        uint32_t taken = 0u, slotsTaken = 0u;
#pragma unroll
        for(int k = 0; k < 6; ++k)
        {
            if(!((haveSend >> k) & 1u)) continue;
            const uint32_t slot = uint32_t(sendSlots[k]) & 31u;
            if((slotsTaken >> slot) & 1u) continue;
            slotsTaken |= 1u << slot;
            taken |= 1u << k;
            if(laneWet && laneSlot == slot) mySend = k;
        }
        const bool isDry = lane < numDry;
        const bool mine = isDry || mySend >= 0;
        const uint32_t from = uint32_t(mySend) * wetCh + laneCh;        // (the sends' gains came in packed, lane = send x wet channel)
        float tgS = __shfl(q.sTg0, int(from & 63u)), cuS = __shfl(q.sCu0, int(from & 63u));
        if(sendLanes > 64u)
        {
            const float tB = __shfl(q.sTg1, int(from & 63u)), cB = __shfl(q.sCu1, int(from & 63u));
            if(from >= 64u) { tgS = tB; cuS = cB; }
        }
        const float tg = (mine && playing) ? (isDry ? q.dryTg : tgS) : 0.0f;              // SilentCoeffs when Stopping
        const float cu = (mine && counter) ? (isDry ? q.dryCu : cuS) : tg;                // voice.cpp:1094-1112
        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
        RowGain all;
        if(mine)
        {
            float *curStore = isDry ? L.gainCur + size_t{vx} * numDry + lane : L.sendCur + (size_t{vx} * numSends + uint32_t(mySend)) * wetCh + laneCh;
            *curStore = g.newCur;
            all.add(g);
        }
        int32_t rowId = -1;
        if(isDry) rowId = directFilter ? 1 : 0;
        else if(mySend >= 0) rowId = ((sfBits >> uint32_t(mySend)) & 1u) ? 2 + mySend : 0;
        // the signals: bits 0..5 the filtered sends, 6 the unfiltered row, 7 the direct-filtered row (in this order: the direct
        // filter runs in place)
        uint32_t order = taken & sfBits;
        if(__ballot(rowId == 0) != 0ull) order |= 1u << 6;
        if(directFilter) order |= 1u << 7;
        const uint32_t flags = kRfActive | (playing ? kRfPlaying : 0u) | (vp.looping ? kRfLooping : 0u) | (vp.queue ? kRfQueue : 0u)
            | (vp.anyFull ? kRfAnyFull : 0u) | (vp.multi ? kRfMulti : 0u) | (directFilter ? kRfDirectFilter : 0u);
        if(lane < 16u)
        {
            uint32_t wv = 0u;
            wv = lane == uint32_t(kRwFlags) ? flags : wv;
            wv = lane == uint32_t(kRwOutPos) ? vp.outPos : wv;
            wv = lane == uint32_t(kRwBufferItem) ? uint32_t(vp.bufferItem) : wv;
            wv = lane == uint32_t(kRwBsrcFull) ? vp.bsrcFull : wv;
            wv = lane == uint32_t(kRwSfBits) ? sfBits : wv;
            wv = lane == uint32_t(kRwHaveSend) ? haveSend : wv;
            wv = lane == uint32_t(kRwPending) ? (haveSend & ~taken) : wv;
            wv = lane == uint32_t(kRwOrder) ? order : wv;
            rec.w[lane] = wv;
        }
        if(lane < 32u) { rec.gain[lane] = all.gain; rec.cur[lane] = all.cur; rec.step[lane] = all.step; rec.fade[lane] = all.fadeLen; rec.rowId[lane] = rowId; }
    };
    auto planOf = [&](uint32_t rw) -> VoicePlan
    {   // (a record's words, lane = word, back into the plan the slice code reads)
        VoicePlan p{};
        const uint32_t fl = CtlWord(rw, kRwFlags);
        p.active = (fl & kRfActive) != 0u; p.playing = (fl & kRfPlaying) != 0u; p.looping = (fl & kRfLooping) != 0u; p.queue = (fl & kRfQueue) != 0u;
        p.anyFull = (fl & kRfAnyFull) != 0u; p.multi = (fl & kRfMulti) != 0u;
        p.outPos = CtlWord(rw, kRwOutPos); p.bufferItem = int32_t(CtlWord(rw, kRwBufferItem)); p.bsrcFull = CtlWord(rw, kRwBsrcFull);
        return p;
    };

    float preN[kSlPre];
    float prevN = 0.0f;
#pragma unroll
    for(int i = 0; i < kSlPre; ++i) preN[i] = 0.0f;
    // requested one voice ahead with the window, by the wavefront of the voice's FIRST slice: the filter pairs as memory holds them
    float dfN = 0.0f;
    float sfN[3] = {0.0f, 0.0f, 0.0f};            // the sends' pairs, two per register: sends 2 q (lanes 0..31) and 2 q + 1
    VoicePlan vpN{};
    SlicePlan spN{};
    uint32_t rwN = 0u;                            // the next voice's record words, lane = word
    auto request = [&](uint32_t vn, uint32_t jbn, uint32_t ctlv, uint32_t lane)
    {
        rwN = sm.rec[jbn].w[lane & 15u];
        vpN = planOf(rwN);
        spN = SlicePlan{};
        if(!vpN.active) return;
        const VoiceHead h = HeadFrom(ctlv);
        const BufferItem b = BufFrom(ctlv);
        spN = PlanSlice(h, b, vpN, f0, nS);
        if(spN.pref)
        {
            const uint32_t total = uint32_t(kMaxEdge) + spN.bsrc;
            GatherSlice(preN, total - spN.nPrev, b, vpN.looping, spN.upos, lane);
            prevN = (lane < spN.nPrev) ? L.prev[size_t{vn} * kMaxPad + spN.srcOff + lane] : 0.0f;
        }
        // the voice's filter pairs as memory holds them, for the wavefront of its first slice: it starts the ones that run and
        // clears the inactive ones that are not at rest
        if(spN.cnt && wave == vpN.outPos / uint32_t(kSl))
        {
            if(lane < 32u) dfN = reinterpret_cast<const float*>(L.dfilt + size_t{vn} * 2)[lane];
            const float *sf = reinterpret_cast<const float*>(L.sfilt + size_t{vn} * numSends * 2);      // [send][2 x 16 dwords]
#pragma unroll
            for(int q = 0; q < 3; ++q)
                if(64u * uint32_t(q) + lane < 32u * numSends) sfN[q] = sf[64u * uint32_t(q) + lane];
        }
    };

    for(uint32_t b0 = 0; b0 < nv; b0 += uint32_t(kSlBlock))
    {
    const uint32_t nb = (nv - b0 < uint32_t(kSlBlock)) ? nv - b0 : uint32_t(kSlBlock);
    __syncthreads();                              // the block before is through: its records and mailboxes are free
    // (the lane index behind an opaque move per block: addresses built from it are a few VALU here, not loop invariants in scratch)
    uint32_t laneB = lane0;
    asm volatile("" : "+v"(laneB));
    {   // ---- the block's records: four voices per wavefront, the next one's loads in flight while one is worked out
        ProLoads qa, qb;
        if(wave < nb) proRequest(v0 + b0 + wave, qa, laneB);
#pragma unroll
        for(int q = 0; q < kSlBlock / kWWaves; ++q)
        {
            const uint32_t jbq = wave + uint32_t(kWWaves * q);
            if(jbq >= nb) break;
            if(jbq + uint32_t(kWWaves) < nb) proRequest(v0 + b0 + jbq + uint32_t(kWWaves), qb, laneB);
            prepare(v0 + b0 + jbq, qa, sm.rec[jbq], laneB);
            qa = qb;
        }
    }
    __syncthreads();
    uint32_t ctlA = loadCtlV(v0 + b0, laneB), ctlB = loadCtlV(v0 + b0 + 1u, laneB);
    // (nothing of the block before is carried: the request registers begin again, so that they are not live across the prologue)
#pragma unroll
    for(int i = 0; i < kSlPre; ++i) preN[i] = 0.0f;
    prevN = 0.0f; dfN = 0.0f; sfN[0] = sfN[1] = sfN[2] = 0.0f;
    request(v0 + b0, 0u, ctlA, laneB);

    for(uint32_t jb = 0; jb < nb; ++jb)
    {
        uint32_t lane = lane0;
        asm volatile("" : "+v"(lane));
        const uint32_t j = b0 + jb;
        const uint32_t v = v0 + j;
        const uint32_t tagBase = (j / uint32_t(kSlBlock)) * 8u;
        const uint32_t ctlCur = ctlA, rw = rwN;
        const VoicePlan vp = vpN;
        const SlicePlan sp = spN;

        // this voice's window leaves the registers; the next voice's is requested
        if(sp.pref)
        {
            const bool isShort = int32_t(CtlWord(ctlCur, 26)) == OALGPU_FMT_SHORT;
            if(lane < sp.nPrev) { w.rd[lane] = prevN; if(lane) w.rd2[lane - 1u] = prevN; }
#pragma unroll
            for(int i = 0; i < kSlPre; ++i)
            {
                const float sv = GatherDecode(preN[i], isShort);
                const uint32_t at = sp.nPrev + lane + 64u * uint32_t(i);
                w.rd[at] = sv;
                if(at) w.rd2[at - 1u] = sv;
            }
        }
        const uint32_t kFirst = vp.outPos / uint32_t(kSl);
        const bool isFirst = wave == kFirst, isLast = wave == kLast;
        uint32_t restMask = 0u;                 // isFirst: which of the voice's filter pairs are at rest (bit 0 direct, 1 + i send i)
        if(sp.cnt && isFirst)
        {
            if(lane < 32u) w.st[0][lane] = dfN;
            restMask = PairsAtRest(dfN, lane) & 1u;
#pragma unroll
            for(int q = 0; q < 3; ++q)
                if(2u * uint32_t(q) < numSends)
                {
                    if(64u * uint32_t(q) + lane < 32u * numSends) (&w.st[1][0])[64u * uint32_t(q) + lane] = sfN[q];
                    restMask |= PairsAtRest(sfN[q], lane) << (1u + 2u * uint32_t(q));
                }
        }
        ctlA = ctlB;
        if(jb + 1u < nb)
        {
            request(v + 1u, jb + 1u, ctlA, lane);
            ctlB = loadCtlV(v + 2u, lane);
        }
        if(!vp.active) continue;
        if(wave < kFirst || wave > kLast) continue;
        const VoiceHead head = HeadFrom(ctlCur);
        const BufferItem buf = BufFrom(ctlCur);
        const bool playing = vp.playing;
        const uint32_t cnt = sp.cnt, dstOff = sp.dstOff;
        float *row = w.in;

        // ---- LoadResampledSamples for the slice: row[dstOff .. dstOff + cnt)
        if(cnt != uint32_t(kSl)) for(uint32_t k = lane; k < uint32_t(kSl); k += 64u) if(k < dstOff || k >= dstOff + cnt) row[k] = 0.0f;
        {
            const int kind = head.rsKind;
            const uint32_t increment = head.step;
            const uint32_t tableKey = head.rsFilterOffset * 8u + uint32_t(kind);
            const bool staged = (kind == 2 || kind == 3) && sm.tabKey == tableKey;
            const uint32_t sM = kind == 2 ? 4u : head.rsM, sL = kind == 2 ? 1u : head.rsL;
            const float *filter = L.tables + head.rsFilterOffset;
            // The reference's chunks of the voice's update (CalculateBufferSize with ITS capacity: one chunk up to a pitch of ~1.27),
            // cut to the slice; each cut in pieces the slice's window holds.  What lies past a source's end depends on the chunk
            // that loaded it (FillWindow).
            const uint32_t Nv = N - vp.outPos;
            const uint32_t l0 = f0 + dstOff - vp.outPos, l1 = l0 + cnt;      // the slice's voice samples
            uint32_t D = 0u, S = 0u, Sprev = 0u, fc = head.positionFrac;
            bool firstPiece = true;
            while(D < l1)
            {
                uint32_t cd = Nv - D, cs = 0u;
                if(vp.multi) CalcBufferSize(fc, increment, Nv - D, cd, cs);
                const uint32_t a = D > l0 ? D : l0, e = (D + cd < l1) ? D + cd : l1;
                if(a < e)
                {
                    const uint64_t ta = uint64_t{fc} + uint64_t{a - D} * increment;
                    uint32_t srcOff = S + uint32_t(ta >> kFracBits), frac = uint32_t(ta) & kFracMask;
                    for(uint32_t loaded = a; loaded < e;)
                    {
                        uint32_t bdst, bsrc;
                        SliceBufferSize(frac, increment, e - loaded, bdst, bsrc);
                        const bool dual = firstPiece && sp.pref;
                        if(!dual)
                        {
                            WaveSync();
                            FillWindow(w.rd, srcOff, uint32_t(kMaxEdge) + bsrc, L, v, head, buf, vp, w, lane, S, Sprev);
                        }
                        WaveSync();
                        float *out = row + dstOff + (loaded - l0);
                        if(increment == kFracOne && frac == 0u)
                        {
                            for(uint32_t k = lane; k < bdst; k += 64u) out[k] = w.rd[kMaxEdge + k];
                        }
                        else if(staged)
                            ResampleRunRingM(sm, w.rd + (kMaxEdge - sL), sM, frac, increment, bdst, out, reinterpret_cast<float*>(&w.pad[0]), lane,
                                dual ? w.rd2 + (kMaxEdge - sL) : nullptr, uint32_t(kMaxEdge) - sL);
                        else
                        {
                            const TabLayout lay = ReferenceTabLayout(head.rsM);
                            for(uint32_t k = lane; k < bdst; k += 64u)
                                out[k] = ResampleAt<false, false>(kind, head.rsM, head.rsL, head.rsSf, filter, lay, w.rd, frac, increment, k, bdst);
                        }
                        loaded += bdst;
                        const uint64_t tn = uint64_t{frac} + uint64_t{bdst} * increment;
                        srcOff += uint32_t(tn >> kFracBits);
                        frac = uint32_t(tn) & kFracMask;
                        firstPiece = false;
                    }
                }
                const uint64_t tc = uint64_t{fc} + uint64_t{cd} * increment;
                Sprev = S;
                S += uint32_t(tc >> kFracBits);
                fc = uint32_t(tc) & kFracMask;
                D += cd;
            }
        }
        asm volatile("" : "+v"(lane));
        WaveSync();

        // ---- a recurrence's state: from memory (the voice's first slice) or from the wavefront of the slice before
        auto acquire = [&](uint32_t stage, const float *global32, uint32_t words = 32u)
        {
            if(isFirst)
            {
                if(stage == uint32_t(kStageAmbi)) { if(lane < words) w.fst[lane] = global32[lane]; }
                else if(lane < 32u) w.fst[lane] = w.st[stage][lane];
            }
            else
            {
                const uint32_t want = tagBase + wave;
                for(uint32_t spins = 0;; ++spins)
                {
                    const uint32_t f = __hip_atomic_load(&sm.flag[jb][stage], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if(f == want) break;
                    __builtin_amdgcn_s_sleep(8);                        // (~0.2 us a look: a waiting wavefront's polls are instructions the SIMD's other wavefront could issue)
                    if(spins > (1u << 24)) break;                      // (never: a wavefront waits only for lower slices, and slice 0 for nobody)
                }
                // (mailbox and flag are LDS, which serves a wavefront's operations in order: nothing of wider scope to wait for --
                // a workgroup-scope fence would also wait for the next voice's window, in flight from HBM)
                WaveSync();
                if(lane < 32u) w.fst[lane] = sm.mail[jb][stage][lane];
            }
            WaveSync();
        };
        auto release = [&](uint32_t stage, float *global32)
        {
            WaveSync();
            if(isLast) { if(lane < 32u && (lane & 15u) < 13u) global32[lane] = w.fst[lane]; }
            else
            {
                if(lane < 32u) sm.mail[jb][stage][lane] = w.fst[lane];
                WaveSync();
                if(lane == 0) __hip_atomic_store(&sm.flag[jb][stage], tagBase + wave + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            WaveSync();
        };

        if(head.flags & kFlagAmbiScale)
        {   // VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091
            acquire(kStageAmbi, reinterpret_cast<const float*>(L.ambi + v), 8u);
            const float coeff = w.fst[0], hfS = w.fst[4], lfS = w.fst[5];
            SplitterState spl{coeff, w.fst[1], w.fst[2], w.fst[3]};
            WaveSync();
            SplitterScan<false>(spl, row + dstOff, cnt, hfS, lfS, lane);
            WaveSync();
            if(lane == 0) { w.fst[1] = spl.lpZ1; w.fst[2] = spl.lpZ2; w.fst[3] = spl.apZ1; }
            WaveSync();
            if(isLast) { if(lane >= 1u && lane < 4u) reinterpret_cast<float*>(L.ambi + v)[lane] = w.fst[lane]; }
            else release(kStageAmbi, nullptr);
            WaveSync();
        }

        // ---- DoFilters and MixSamples (voice.cpp:934-984).  Every line is mixed from exactly one of the voice's signals -- the
        // unfiltered row (0: dry lines without a direct filter, sends without a filter), the direct-filtered row (1), a send's
        // own filtered copy (2 + send) -- with the gains the voice's record holds (line = lane).  The signals that exist go one
        // after the other through ONE filter run and ONE mix: a single site of each keeps the 96 accumulator registers in one
        // allocation across the loop.
        const bool ramp = wave == 0u;
        const SliceRec &rec = sm.rec[jb];
        RowGain all;
        int32_t rowId = -1;
        if(lane < 32u)
        {
            all.gain = rec.gain[lane]; rowId = rec.rowId[lane];
            if(ramp) { all.cur = rec.cur[lane]; all.step = rec.step[lane]; all.fadeLen = rec.fade[lane]; }
        }
        uint32_t order = CtlWord(rw, kRwOrder);
        uint32_t pending = CtlWord(rw, kRwPending);
        for(;;)
        {
            while(order)
            {
                const uint32_t bit = uint32_t(__builtin_ctz(order));
                order &= order - 1u;
                const int32_t k = bit < 6u ? int32_t(2u + bit) : int32_t(bit - 6u);
                RowGain rg;
                if(rowId == k) rg = all;
                const float *src = row;
                uint32_t stage = 0u;
                float *slots32 = reinterpret_cast<float*>(&L.dfilt[size_t{v} * 2]);
                if(k >= 2)
                {   // the send's own filtered copy
                    for(uint32_t i = lane; i < uint32_t(kSl); i += 64u) w.tmp[i] = row[i];
                    src = w.tmp;
                    stage = 1u + bit;
                    slots32 = reinterpret_cast<float*>(&L.sfilt[(size_t{v} * numSends + bit) * 2]);
                }
                if(k != 0)
                {
                    acquire(stage, slots32);
                    SliceFilterRun(w.fst, (k >= 2 ? w.tmp : row) + dstOff, cnt, lane);
                    release(stage, slots32);
                }
                WaveSync();
                MixSlice<ACCN>(acc, src, rg, ramp, lane);
                WaveSync();
            }
            if(!pending) break;
            // ---- (rare) further sends into a slot that already has one: another pass, resolved here -- every slot's first pending send
            const uint32_t sfBits = CtlWord(rw, kRwSfBits);
            int32_t mySend = -1;
            uint32_t taken = 0u, slotsTaken = 0u;
#pragma unroll
            for(int q = 0; q < 6; ++q)
            {
                if(!((pending >> q) & 1u)) continue;
                const uint32_t slot = CtlWord(ctlCur, 12 + q) & 31u;
                if((slotsTaken >> slot) & 1u) continue;
                slotsTaken |= 1u << slot;
                taken |= 1u << q;
                if(laneWet && laneSlot == slot) mySend = q;
            }
            pending &= ~taken;
            const bool mine = mySend >= 0;
            const size_t gi = (size_t{v} * numSends + uint32_t(mine ? mySend : 0)) * wetCh + laneCh;
            const float tgS = mine ? L.sendTgt[gi] : 0.0f;
            const float tg = (mine && playing) ? tgS : 0.0f;
            const uint32_t counter = (head.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;
            all = RowGain{};
            if(isFirst)
            {
                const float cu = (mine && counter) ? L.sendCur[gi] : tg;
                const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                if(mine) { L.sendCur[gi] = g.newCur; all.add(g); }
            }
            else if(mine)
            {
                const float c = (__builtin_fabsf(tg) > kGainSilence) ? tg : 0.0f;
                all.gain = c; all.cur = c;
            }
            rowId = mine ? (((sfBits >> uint32_t(mySend)) & 1u) ? 2 + mySend : 0) : -1;
            order = taken & sfBits;
            if(__ballot(rowId == 0) != 0ull) order |= 1u << 6;
        }
        if(isFirst)
        {   // an inactive pair that is not at rest is cleared (voice.cpp:264-265): bit 0 the direct pair, 1 + i send i's
            const uint32_t sfBits = CtlWord(rw, kRwSfBits);
            uint32_t clr = (((CtlWord(rw, kRwHaveSend) & ~sfBits) << 1) | ((CtlWord(rw, kRwFlags) & kRfDirectFilter) ? 0u : 1u)) & ~restMask;
            while(clr)
            {
                const uint32_t stage = uint32_t(__builtin_ctz(clr));
                clr &= clr - 1u;
                float *slots32 = stage ? reinterpret_cast<float*>(&L.sfilt[(size_t{v} * numSends + (stage - 1u)) * 2]) : reinterpret_cast<float*>(&L.dfilt[size_t{v} * 2]);
                WaveSync();
                if(lane < 32u) w.fst[lane] = w.st[stage][lane];
                WaveSync();
                WaveDoFilters(w.fst, reinterpret_cast<BiquadSlot*>(slots32), false, row, 0u, lane);
                WaveSync();
            }
        }
    }
    }

    // ---- the wavefront's quarter of the workgroup's partial bus (the reduction reads all 1024 frames of every line)
    {
        float *pl = L.partLines + size_t{group} * L.mixLines * kLine + f0 + 4u * lane0;
#pragma unroll
        for(int c = 0; c < ACCN; ++c)
        {
            if(uint32_t(c) < L.mixLines)
                __builtin_nontemporal_store(f4{acc[c][0], acc[c][1], acc[c][2], acc[c][3]}, reinterpret_cast<f4*>(pl + size_t(c) * kLine));
        }
    }

    // ---- the voices' state for the next update (voice.cpp:772-785, :1116-1232): nobody reads this update's any more
    __syncthreads();
    for(uint32_t j = wave; j < nv; j += kWWaves)
    {
        const uint32_t lane = lane0;
        const uint32_t v = v0 + j;
        const VoiceHead head = LoadHeadScalar(L.ctl + v);
        const BufferItem buf = LoadCtlBufferScalar(L.ctl + v);
        const bool mixes = head.playState == OALGPU_VOICE_PLAYING || head.playState == OALGPU_VOICE_STOPPING;
        const bool playing = head.playState == OALGPU_VOICE_PLAYING;
        bool active = mixes && head.step >= 1u;
        VoicePlan vq = PlanVoice(L, v, head, buf, N);            // (reads startDelay, which lane 0 rewrites below)
        // voice.cpp:1002-1010
        if(mixes && !active && !playing && lane == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
        uint32_t outPos = 0;
        if(active && (head.flags & kFlagDelayed))
        {   // voice.cpp:1023-1046
            const uint32_t d = __builtin_amdgcn_readfirstlane(L.startDelay[v]);
            if(!playing)
            {
                if(lane == 0) { L.ctl[v].playState = OALGPU_VOICE_STOPPED; L.ctl[v].flags = head.flags & ~kFlagDelayed; L.startDelay[v] = 0u; }
                active = false;
            }
            else if(d >= N) { if(lane == 0) L.startDelay[v] = d - N; active = false; }
            else { outPos = d; if(lane == 0) L.startDelay[v] = 0u; }
        }
        if(!active) continue;
        int32_t bufferItem = head.curBuffer;
        if(playing)
        {   // mPrevSamples: the 48 samples around the position the update ends at
            const uint32_t Nv = N - outPos;
            uint32_t D = 0u, S = 0u, Sprev = 0u, fc = head.positionFrac;
            while(vq.multi)
            {   // (the reference takes them out of the window of the update's LAST chunk)
                uint32_t cd, cs;
                CalcBufferSize(fc, head.step, Nv - D, cd, cs);
                if(D + cd >= Nv) break;
                const uint64_t tc = uint64_t{fc} + uint64_t{cd} * head.step;
                Sprev = S; S += uint32_t(tc >> kFracBits); fc = uint32_t(tc) & kFracMask; D += cd;
            }
            const uint32_t srcOffEnd = S + uint32_t((uint64_t{Nv - D} * head.step + fc) >> kFracBits);
            WaveSync();
            FillWindow(w.rd, srcOffEnd, uint32_t(kMaxPad), L, v, head, buf, vq, w, lane, S, Sprev);
            WaveSync();
            if(lane < uint32_t(kMaxPad)) L.prev[size_t{v} * kMaxPad + lane] = w.rd[lane];
            WaveSync();
        }
        if(lane == 0)
        {
            VoiceCtl &c = L.ctl[v];
            c.flags = (head.flags | kFlagFading) & ~(kFlagDelayed | kFlagHrtfDirty);
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
                    if(vq.looping)
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
}

} // namespace

// Which layouts the slice kernel takes: dry-line contexts (no HRTF, no near-field control) whose lines do not fit the
// wavefront-per-voice kernel's registers -- sends, or more than six lines -- up to 24 lines.
uint32_t SliceKernelLines(const DeviceLayout &L)
{
    if(L.hrtf || L.nfc || L.mixLines == 0 || L.mixLines > 24 || L.numDry < 1 || L.numSends > 6) return 0;
    if(L.numSends * L.wetChannels > 128u) return 0;
    if(L.numSends == 0 && L.mixLines <= 6) return 0;                 // (the register-accumulator form of voice_wave.hip)
    return 24u;
}

const char *SliceKernelName() { return "VoiceSliceKernel<24>"; }

hipError_t LaunchVoiceSlice(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, hipEvent_t evStart, hipEvent_t evStop)
{
    const dim3 grid(WaveKernelGroups(L)), block(kWThreads);
    hipExtLaunchKernelGGL((VoiceSliceKernel<24>), grid, block, 0, s, evStart, evStop, 0u, SliceArgs{L}, samplesToDo);
    return hipGetLastError();
}

} // namespace oalgpu
