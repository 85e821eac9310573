// The FAST HRTF hot path: one WAVEFRONT per voice (BASELINE configs[2], the headline metric).
//
//   "for every Playing|Stopping voice: voice->mix(...)"   alc/alu.cpp:2201-2206
//      -> Voice::mix                                       core/voice.cpp:988-1233
//      -> LoadResampledSamples / Resample_*                voice.cpp:642-824, mixer_sse.cpp:199-329
//      -> DoFilters (dual biquad)                          voice.cpp:255-267
//      -> DoHrtfMix -> MixHrtf / MixHrtfBlend              voice.cpp:827-902, mixer/hrtfbase.h:17-89
//
// Why a wavefront per voice.  fp32 FMA throughput on gfx950 is only reached by v_pk_fma_f32
// (tools/ubench_fma.hip on MI355X: 124-130 TFLOP/s packed vs 63-100 scalar), and the dual-ear FIR
// is 128 of the 176 FMAs per output sample, so the kernel is built around packed FMAs whose two
// halves are the two EARS: the accumulator pair (L,R) of a frame, the coefficient pair
// (Coeffs[j][0], Coeffs[j][1]) exactly as HrirArray stores it -- fetched through the scalar
// cache into an SGPR pair -- and the input pair x'[i] = (In[64-delayL+i], In[64-delayR+i])*g(i)
// staged once per voice in LDS as float2.  Every operand is a naturally aligned register pair;
// nothing is broadcast, shuffled or re-packed.  Lane l owns R consecutive output frames
// [R*l, R*l+R) of the 64*R-frame accumulator for the whole launch (R = 17: 1088 frames =
// 1024 + IrSize 64; R = 18: 1152 = 1024 + HrirLength for IrSize <= 128), keeps a sliding window
// of R+3 inputs in registers and needs 4 new ds_read_b64 per 4 taps x R frames x 2 ears
// (= 136 FMAs); the lane stride R is odd/coprime to the LDS banking, so those reads are
// conflict-free.  Voices of a wavefront follow each other through the same registers, so bus
// traffic is one partial per WORKGROUP per launch.
//
// The resampler reads its coefficient rows from an LDS copy transposed to [tap pair][phase] so
// that the 32 phases map onto the 64 LDS banks: any per-lane phase pattern is conflict-free
// (ds_read_b64), and (fil + pf*phd) and the tap MACs are packed over tap pairs.
//
// Arithmetic is FAST mode (FMA, own summation order): results agree with the reference within
// the tolerance stated in DESIGN.md; all integer state (positions, loop wrap, play state,
// delays, fade counters) is bit-exact.  EXACT mode and every other configuration (sends, non-HRTF
// buses) run voice_kernel.hip.
#include <hip/hip_ext.h>
#include "wave_common.hpp"

#pragma clang fp contract(off)

#ifndef OALGPU_WAVE_MIN_WG
#define OALGPU_WAVE_MIN_WG 2              // workgroups per CU the kernels are built for (experiments: 1 shows the unconstrained budget)
#endif

namespace oalgpu {
namespace {
// One line's share of a stream row's gain block (kernels.hpp LineBlockDwords): contributions of
// several MixSamples calls onto the same row and line add up -- the constant gains, and for
// the ramped frames the per-frame values (a contribution without a ramp adds its constant there)
struct RowLineGain {
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
__device__ __forceinline__ void StoreRowBlock(uint32_t *blk, uint32_t ls, uint32_t lane, const RowLineGain &r, bool live)
{
    uint32_t maxFade = r.fadeLen;
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = uint32_t(__shfl_xor(int(maxFade), d)); maxFade = o > maxFade ? o : maxFade; }
    if(lane < ls)
    {
        blk[lane] = __builtin_bit_cast(uint32_t, r.gain);
        blk[ls + lane] = __builtin_bit_cast(uint32_t, r.fadeLen ? r.cur - r.gain : 0.0f);
        blk[2u * ls + lane] = __builtin_bit_cast(uint32_t, r.fadeLen ? r.step : 0.0f);
    }
    // flags[0]: which 8-line blocks of the gain vector carry anything (LinesMixKernel runs per block and
    // skips rows that give nothing to its lines); flags[1]: frames covered by the longest ramp
    const bool nz = lane < ls && (r.gain != 0.0f || (r.fadeLen && (r.cur != r.gain || r.step != 0.0f)));
    const unsigned long long m = __ballot(nz);
    const uint32_t mask = ((m & 0xFFull) ? 1u : 0u) | ((m & 0xFF00ull) ? 2u : 0u) | ((m & 0xFF0000ull) ? 4u : 0u)
        | ((m & 0xFF000000ull) ? 8u : 0u);
    if(lane < 8u) blk[3u * ls + lane] = lane == 0u ? (live ? mask : 0u) : (lane == 1u ? maxFade : 0u);
}

// A workgroup's partial bus is written once and read once, by the reduction, from whatever XCD its workgroup lands on:
// stored non-temporally it does not wait, dirty, in this XCD's L2 for the write-back at the kernel's end (0.45 us off the
// launch, 0.5-0.85 off the step: profiles/r4/nt_partials_ab.txt).
template<class T>
__device__ __forceinline__ void StorePartial(T *p, T v)
{
    __builtin_nontemporal_store(v, p);
}

// ---- MixSamples straight out of LDS into the wavefront's own line accumulators (contexts with <= 8 mix lines) ----
// The resampled / filtered samples of a voice are in LDS when DoFilters ends; with few enough lines the wavefront keeps
// N lines x 64R frames of accumulator in registers (lane l owns frames [R l, R l + R) of every line, like the HRTF
// accumulator) and MixSamples (core/mixer/mixer_c.cpp:150-186, as Voice::mix calls it, voice.cpp:934-984) becomes R x N
// FMAs per lane with the line's gain in an SGPR: no stream row, no gain block and no row mix in the kernel's tail --
// the row never leaves the CU.  rg: the row's merged gains, lane = line (RowLineGain); frames below the ramp's length
// use cur + step * frame instead of the constant (MixLine with Counter <= 64: only lanes 0..3 can hold such frames).
template<int ACCL, int R>
__device__ __forceinline__ void MixRowAcc(float (&accL)[ACCL][R], float (&accRamp)[ACCL], const float *row, uint32_t N, const RowLineGain &rg,
    uint32_t lane)
{
    const bool hasRamp = rg.fadeLen != 0u && lane < uint32_t(ACCL);
    const float rampA = hasRamp ? rg.cur - rg.gain : 0.0f, rampB = hasRamp ? rg.step : 0.0f;
    const bool nz = lane < uint32_t(ACCL) && (rg.gain != 0.0f || rampA != 0.0f || rampB != 0.0f);
    const unsigned long long live = __ballot(nz);
    if(live == 0ull) return;
    uint32_t maxFade = hasRamp ? rg.fadeLen : 0u;
#pragma unroll
    for(int d = 32; d >= 1; d >>= 1) { const uint32_t o = uint32_t(__shfl_xor(int(maxFade), d)); maxFade = o > maxFade ? o : maxFade; }
    {
        float x[R];
#pragma unroll
        for(int r = 0; r < R; ++r)
        {
            const uint32_t f = uint32_t(R) * lane + uint32_t(r);
            x[r] = (f < N) ? row[f] : 0.0f;
        }
#pragma unroll
        for(int c = 0; c < ACCL; ++c)
        {
            if(!((live >> c) & 1ull)) continue;                 // (uniform: the ballot is a scalar)
            const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rg.gain), c));
#pragma unroll
            for(int r = 0; r < R; ++r) accL[c][r] = __builtin_fmaf(x[r], g, accL[c][r]);
        }
    }
    if(maxFade != 0u)
    {   // the ramp's distance from the constant, frames < maxFade <= 64: frame = lane, in accumulators of their own
        // (added to the lines when the accumulators are dumped)
        const float xf = (lane < maxFade && lane < N) ? row[lane] : 0.0f;
        const float fl = float(lane);
#pragma unroll
        for(int c = 0; c < ACCL; ++c)
        {
            const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rampA), c));
            const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rampB), c));
            accRamp[c] = __builtin_fmaf(xf, __builtin_fmaf(b, fl, a), accRamp[c]);
        }
    }
}

// ---- MixSamples of the workgroup's stream rows onto the mix lines (core/mixer/mixer_c.cpp:183-215 as
// Voice::mix calls it, voice.cpp:934-984): out[c][f] += row_r[f] * gain_{r,c}(f) over the rows of the
// workgroup's voices in row order, into ONE partial bus per workgroup (BusReduceKernel sums those in
// workgroup order: deterministic).  Runs as the kernel's tail, when all four wavefronts are through their
// voices: the rows were written moments ago by this workgroup, so they come back out of this XCD's L2, and
// the phase overlaps with the other workgroups' resamplers instead of being a launch of its own.
// Thread t owns frames 4t .. 4t+3 of every line.  The rows' gain blocks are staged in LDS (the voices' LDS
// is free by now); a row feeds few lines -- the dry lines, or one slot's wet lines -- and its flags say
// which 8-line blocks (StoreRowBlock), so only those are read and multiplied.  A gain ramp (MixLine with
// Counter <= 64, voice.cpp:1093) only ever covers the first 64 frames: wavefront 0 adds
// s * (rampA + rampB * f), the ramp's distance from the constant, behind the row's constant term.
constexpr uint32_t kMixListMax = 512;             // live rows per staged chunk (>= kMixGainDwords / 32)
constexpr uint32_t kMixGainDwords = 12288;        // staged gain blocks: 48 KB
template<int S, uint32_t kRowBatch = 4>
__device__ __forceinline__ void WgMixRows(uint32_t *lds, const DeviceLayout &L, uint32_t group, uint32_t v0, uint32_t nv,
    uint32_t t, uint32_t N)
{
    constexpr uint32_t kBlk = 3u * S + 8u;
    const uint32_t lane = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t spv = L.streamsPerVoice;
    uint32_t *list = lds, *count = lds + kMixListMax, *gl = lds + kMixListMax + 4u;
    float acc[S][4];
#pragma unroll
    for(int c = 0; c < S; ++c)
#pragma unroll
        for(int j = 0; j < 4; ++j) acc[c][j] = 0.0f;
    uint32_t chunkVoices = kMixGainDwords / (spv * kBlk);
    chunkVoices = chunkVoices < 1u ? 1u : (chunkVoices > kMixListMax / spv ? kMixListMax / spv : chunkVoices);
    for(uint32_t c0 = 0; c0 < nv; c0 += chunkVoices)
    {
        const uint32_t cv = (nv - c0 < chunkVoices) ? nv - c0 : chunkVoices;
        const uint32_t nrows = cv * spv;
        const uint32_t *src = L.lineGains + size_t{v0 + c0} * spv * kBlk;
        __syncthreads();
        for(uint32_t i = t; i < nrows * kBlk; i += kWThreads) gl[i] = src[i];
        __syncthreads();
        if(wave == 0u)
        {   // the chunk's live rows, in row order
            uint32_t n = 0;
            for(uint32_t r0 = 0; r0 < nrows; r0 += 64u)
            {
                const uint32_t r = r0 + lane;
                const bool live = r < nrows && gl[r * kBlk + 3u * S] != 0u;
                const unsigned long long ml = __ballot(live);
                if(live) list[n + uint32_t(__popcll(ml & ((1ull << lane) - 1ull)))] = r;
                n += uint32_t(__popcll(ml));
            }
            if(lane == 0u) *count = n;
        }
        __syncthreads();
        const uint32_t nLive = __builtin_amdgcn_readfirstlane(*count);
        const float *rows = L.streams + size_t{v0 + c0} * spv * kLine + 4u * t;
        // the rows come back from L2 / the fabric (~2 K cycles a round trip): the next batch's loads are in flight while
        // one batch is multiplied
        uint32_t rrN[kRowBatch];
        float4 sN[kRowBatch];
        auto loadBatch = [&](uint32_t b)
        {
#pragma unroll
            for(uint32_t k = 0; k < kRowBatch; ++k)
            {
                rrN[k] = __builtin_amdgcn_readfirstlane(list[(b + k < nLive) ? b + k : (nLive ? nLive - 1u : 0u)]);
                sN[k] = *reinterpret_cast<const float4*>(rows + size_t{rrN[k]} * kLine);
            }
        };
        if(nLive) loadBatch(0u);
        for(uint32_t b = 0; b < nLive; b += kRowBatch)
        {
            uint32_t rr[kRowBatch];
            float4 s[kRowBatch];
#pragma unroll
            for(uint32_t k = 0; k < kRowBatch; ++k) { rr[k] = rrN[k]; s[k] = sN[k]; }
            if(b + kRowBatch < nLive) loadBatch(b + kRowBatch);
#pragma unroll
            for(uint32_t k = 0; k < kRowBatch; ++k)
            {
                if(b + k >= nLive) break;
                const uint32_t *blk = gl + rr[k] * kBlk;
                const uint2 fl = *reinterpret_cast<const uint2*>(blk + 3u * S);
                const uint32_t mask = __builtin_amdgcn_readfirstlane(fl.x), len = __builtin_amdgcn_readfirstlane(fl.y);
                const float sv[4] = {s[k].x, s[k].y, s[k].z, s[k].w};
#pragma unroll
                for(int z = 0; z < S / 8; ++z)
                {
                    if(!((mask >> z) & 1u)) continue;
                    const float4 g0 = *reinterpret_cast<const float4*>(blk + 8 * z), g1 = *reinterpret_cast<const float4*>(blk + 8 * z + 4);
                    const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                    for(int c = 0; c < 8; ++c)
#pragma unroll
                        for(int j = 0; j < 4; ++j) acc[8 * z + c][j] = __builtin_fmaf(sv[j], gv[c], acc[8 * z + c][j]);
                    if(wave == 0u && len != 0u)
                    {
                        const float4 a0 = *reinterpret_cast<const float4*>(blk + S + 8 * z), a1 = *reinterpret_cast<const float4*>(blk + S + 8 * z + 4);
                        const float4 b0 = *reinterpret_cast<const float4*>(blk + 2 * S + 8 * z), b1 = *reinterpret_cast<const float4*>(blk + 2 * S + 8 * z + 4);
                        const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                        const float bv[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                        for(int j = 0; j < 4; ++j)
                        {
                            const uint32_t fr = 4u * t + uint32_t(j);
                            const float sr = (fr < len) ? sv[j] : 0.0f;
                            const float ff = float(fr);
#pragma unroll
                            for(int c = 0; c < 8; ++c) acc[8 * z + c][j] = __builtin_fmaf(sr, av[c] + bv[c] * ff, acc[8 * z + c][j]);
                        }
                    }
                }
            }
        }
    }
    float *pl = L.partLines + size_t{group} * L.mixLines * kLine + 4u * t;
#pragma unroll
    for(int c = 0; c < S; ++c)
    {
        if(uint32_t(c) >= L.mixLines) continue;
        float4 o;
        o.x = (4u * t < N) ? acc[c][0] : 0.0f; o.y = (4u * t + 1u < N) ? acc[c][1] : 0.0f;
        o.z = (4u * t + 2u < N) ? acc[c][2] : 0.0f; o.w = (4u * t + 3u < N) ? acc[c][3] : 0.0f;
        StorePartial(reinterpret_cast<f4*>(pl + size_t(c) * kLine), f4{o.x, o.y, o.z, o.w});
    }
}

// WgMixRows as a launch of its own (behind voice_wave16.hip's kernel with sends): workgroup g mixes the rows of voices
// [g vpg, (g + 1) vpg) into partial bus g
template<int S>
__global__ void __launch_bounds__(kWThreads) StreamRowsMixKernel(DeviceLayout L, uint32_t samplesToDo, uint32_t vpg)
{
    __shared__ uint32_t lds[kMixListMax + 4u + kMixGainDwords];
    const uint32_t group = blockIdx.x, v0 = group * vpg;
    const uint32_t nv = v0 < L.numVoices ? ((L.numVoices - v0 < vpg) ? L.numVoices - v0 : vpg) : 0u;
    // (a launch of its own has no resampler beside it to hide the rows' trip: a workgroup's sixteen live rows are requested at once)
    WgMixRows<S, 4u>(lds, L, group, v0, nv, threadIdx.x, samplesToDo);
}

// NL == 0: HRTF voices (DoHrtfMix into the wave's register accumulator).
// NL == 1: voices panned onto dry lines (MixSamples, voice.cpp:962-963).  Holding N lines x 1024
// frames of accumulator per wavefront would take 16 N VGPRs per lane on top of the resampler's
// ~230 (measured: 209 spilled registers at 2 waves/SIMD, or 1 wave/SIMD and a latency-bound
// resampler), so this variant stops after DoFilters: it leaves the voice's 1024 filtered
// samples in HBM (L.streams, 4 KB per voice) together with the resolved MixLine gain of every
// line (L.lineGains), and LinesMixKernel below turns those into partial buses.
// SENDS: the context has auxiliary sends (voice.cpp:966-983): every send with a slot leaves a
// stream row too -- the unfiltered resampled samples shared by all sends (and the direct path)
// whose filter is inactive, or its own filtered copy -- with a gain block over the wet lines.
// Stream rows of a voice: [0] unfiltered, [1] direct-filtered (dry-line contexts), [2+i] send i filtered.
// MF: the dual-ear FIR of HRTF voices (IrSize <= 64) on the matrix pipe in split half precision (FirMfmaH,
// dev_wave.hpp) instead of packed VALU FMAs: 90 MFMAs (1.4 K cycles of the matrix pipe) per voice in the place
// of 1088 v_pk_fma_f32 (8.7 K cycles of the VALU), beside the other wavefront's resampler and filters.
// PROF: the measurement variant (tools/phase_times.py): s_memtime stamps per phase and stage ablation; the
// product variants carry none of it.
// LT: the kernel's argument block.  The dry-line and send variants take the whole DeviceLayout; the HRTF variants
// without sends -- the hot path -- take WaveArgsHrtf, the 16 fields they read: a by-value DeviceLayout is ~70 dwords
// of kernel arguments, preloaded into SGPRs at entry and spilled from there into VGPR lanes one pair at a time
// (the pass loop used to be entered 4 K cycles into a 90 K-cycle wavefront).
struct WaveArgsHrtf {
    uint32_t numVoices, waveVoices, irStride, pad;
    const float *tables;
    const BufferItem *buffers;
    VoiceCtl *ctl;
    float *prev;
    BiquadSlot *dfilt;
    float *hrtfOld, *hrtfTgt, *hist;
    AmbiScaleState *ambi;
    uint32_t *startDelay, *queueDone;
    float *partHrtf;
    const float *hrirs;                         // the store's HRIRs: what a parameter record's blend indices point into
    explicit WaveArgsHrtf(const DeviceLayout &L) : numVoices{L.numVoices}, waveVoices{L.waveVoices}, irStride{L.irStride}, pad{L.irSize},
        tables{L.tables}, buffers{L.buffers}, ctl{L.ctl}, prev{L.prev}, dfilt{L.dfilt}, hrtfOld{L.hrtfOld}, hrtfTgt{L.hrtfTgt},
        hist{L.hist}, ambi{L.ambi}, startDelay{L.startDelay}, queueDone{L.queueDone}, partHrtf{L.partHrtf}, hrirs{L.hrirs} { }
};

// The next update's parameter block, installed by the wavefront that has just mixed the voice (the kernel's epilogue) instead
// of by a kernel of its own between two voice kernels: ApplyRecordWave (kernels.hpp) for the kernels that carry the whole
// DeviceLayout, the same operations on the lean argument block for the HRTF kernels without sends (`pad` holds IrSize there).
struct NextBlock { const ParamRecord *recs; const int32_t *map; const float *rows; ResidentArgs res; };   // (rows: the records' blended target HRIRs,
                                                                                                          // null: blend at install; res: the resident launch's, RES)

// ---- the resident launch (RES, OALGPU_CTX_RESIDENT; the protocol: kernels.hpp ResidentDoor) ----
// What the host writes is read with system-scope loads (they go to memory: nothing a cache holds of the door can be trusted
// inside one launch); what other launches read while this one runs -- the partial buses -- is stored written-through
// (device-scope relaxed atomics: sc1 stores), the idiom of post_wave.hip.
[[maybe_unused]] __device__ __forceinline__ uint32_t ResLoadSys(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
[[maybe_unused]] __device__ __forceinline__ unsigned long long ResLoadSys(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
[[maybe_unused]] __device__ __forceinline__ uint32_t ResLoadDev(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
[[maybe_unused]] __device__ __forceinline__ void StorePartialCoherent(f2 *p, f2 v)
{
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One poll loop for every wait of the resident launch: `ready` is asked until it says yes or the watchdog's time is up
// (false: the caller flags the error and leaves -- a wait never hangs the GPU).  Short naps first, longer ones once the
// wait has lasted.
// waited: what the wait lasted, in s_memrealtime ticks, is added to this counter (null: nowhere)
template<class F>
__device__ __forceinline__ bool ResWait(F ready, uint32_t *waited = nullptr)
{
    if(ready()) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    for(uint32_t spins = 0;; ++spins)
    {
        if(spins < 32u) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(24);
        if(ready()) { ok = true; break; }
        if((spins & 15u) == 15u && __builtin_amdgcn_s_memrealtime() - t0 > kResidentWatchdogTicks) break;
    }
    if(waited) __hip_atomic_fetch_add(waited, uint32_t(__builtin_amdgcn_s_memrealtime() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}
__device__ __forceinline__ void ApplyRecordLean(const WaveArgsHrtf &L, const ParamRecord &r, uint32_t lane)
{
    const uint32_t v = r.voice;
    VoiceCtl &ctl = L.ctl[v];
    if(lane == 0)
    {
        ctl.step = r.step;
        ctl.rsKind = r.rsKind; ctl.rsM = r.rsM; ctl.rsL = r.rsL; ctl.rsSf = r.rsSf;
        ctl.rsFilterOffset = r.rsFilterOffset;
        const uint32_t keep = ctl.flags & (kFlagFading | kFlagHasHrtf | kFlagAmbiScale | kFlagNfc | kFlagDelayed | kFlagQueue
            | (r.keepHrtf ? uint32_t(kFlagHrtfDirty) : 0u));
        ctl.flags = keep | (r.flags & ~(kFlagFading | kFlagHasHrtf | kFlagHrtfDirty | kFlagAmbiScale | kFlagNfc | kFlagDelayed | kFlagQueue))
            | kFlagHasHrtf | (r.keepHrtf ? 0u : uint32_t(kFlagHrtfDirty));
        for(int i = 0; i < 6; ++i) ctl.sendSlot[i] = -1;
        if(!r.keepHrtf)
        {
            ctl.hrtfTgtDelay[0] = r.hrtfDelay[0]; ctl.hrtfTgtDelay[1] = r.hrtfDelay[1];
            ctl.hrtfTgtGain = r.hrtfGain;
        }
    }
    if(lane == 1) BiquadSetTarget(L.dfilt[size_t{v} * 2 + 0].f, r.dirLp);
    if(lane == 2) BiquadSetTarget(L.dfilt[size_t{v} * 2 + 1].f, r.dirHp);
    if(!r.keepHrtf)
    {   // ApplyHrtfTargetWave (kernels.hpp), on the lean block
        const uint32_t i0 = r.hrtfIdx[0], i1 = r.hrtfIdx[1], i2 = r.hrtfIdx[2], i3 = r.hrtfIdx[3];
        const float w0 = r.hrtfW[0], w1 = r.hrtfW[1], w2 = r.hrtfW[2], w3 = r.hrtfW[3], pass = r.hrtfPass;
        const uint32_t live = ((L.pad + 1u) & ~1u) * 2u;
        for(uint32_t e = lane; e < L.irStride * 2; e += 64)
        {
            float x = (e < 2) ? pass : 0.0f;
            x = L.hrirs[size_t{i0} * (kHrirLen * 2) + e] * w0 + x;
            x = L.hrirs[size_t{i1} * (kHrirLen * 2) + e] * w1 + x;
            x = L.hrirs[size_t{i2} * (kHrirLen * 2) + e] * w2 + x;
            x = L.hrirs[size_t{i3} * (kHrirLen * 2) + e] * w3 + x;
            L.hrtfTgt[size_t{v} * L.irStride * 2 + e] = (e < live) ? x : 0.0f;
        }
    }
}
__device__ __forceinline__ void ApplyNextRecord(const WaveArgsHrtf &L, const ParamRecord &r, uint32_t lane) { ApplyRecordLean(L, r, lane); }
__device__ __forceinline__ void ApplyNextRecord(const DeviceLayout &L, const ParamRecord &r, uint32_t lane) { ApplyRecordWave(L, r, lane); }

// ---- RES: ApplyRecordLean for TWO voices at once, as three rounds of loads instead of a dozen dependent ones ----
// The resident kernel installs an update's records in front of the update, on every workgroup's critical path: record by record
// (map entry -> record -> four HRIRs -> control line -> filter slots, each waiting for the one before, per voice) that was as long
// as mixing a voice.  Here a wavefront reads both map entries first; then everything both installs READ -- the records' scalars,
// the voices' control flags, the filter slots' targets and the records' blended target HRIRs, which the block carries
// precomputed (BlendRowsKernel: the same weighted sum as ApplyHrtfTargetWave, evaluated once when the block is created) --;
// then it stores.  Same values as ApplyRecordLean, so the same bits as ApplyParamsKernel.
struct RecScalars {
    uint32_t step; int32_t rsKind; uint32_t rsM, rsL; float rsSf; uint32_t rsFilterOffset, flags, keepHrtf, delay0, delay1; float gain;
    float lp[5], hp[5];
};
__device__ __forceinline__ RecScalars LoadRecScalars(const ParamRecord &r)
{
    RecScalars o;
    o.step = r.step; o.rsKind = r.rsKind; o.rsM = r.rsM; o.rsL = r.rsL; o.rsSf = r.rsSf; o.rsFilterOffset = r.rsFilterOffset;
    o.flags = r.flags; o.keepHrtf = r.keepHrtf; o.delay0 = r.hrtfDelay[0]; o.delay1 = r.hrtfDelay[1]; o.gain = r.hrtfGain;
#pragma unroll
    for(int k = 0; k < 5; ++k) { o.lp[k] = r.dirLp[k]; o.hp[k] = r.dirHp[k]; }
    return o;
}
__device__ __forceinline__ void InstallPair(const WaveArgsHrtf &L, const int32_t *__restrict__ map, const ParamRecord *__restrict__ recs,
    const float *__restrict__ rows, uint32_t vA, uint32_t vB, bool hasB, uint32_t lane)
{
    // ---- round 1: which records
    const int32_t mA = map[vA], mB = map[hasB ? vB : vA];
    const int32_t riA = __builtin_amdgcn_readfirstlane(mA), riBr = __builtin_amdgcn_readfirstlane(mB);
    const int32_t riB = hasB ? riBr : -1;
    if(riA < 0 && riB < 0) return;
    const bool doA = riA >= 0, doB = riB >= 0;
    const uint32_t irStride = L.irStride;
    // ---- round 2: everything the installs read
    RecScalars sA{}, sB{};
    f2 rowA = {0.0f, 0.0f}, rowB = {0.0f, 0.0f};
    uint32_t flA = 0u, flB = 0u;
    if(doA)
    {
        sA = LoadRecScalars(recs[riA]);
        if(lane < irStride) rowA = reinterpret_cast<const f2*>(rows + size_t{uint32_t(riA)} * irStride * 2u)[lane];
        if(lane == 0) flA = L.ctl[vA].flags;
    }
    if(doB)
    {
        sB = LoadRecScalars(recs[riB]);
        if(lane < irStride) rowB = reinterpret_cast<const f2*>(rows + size_t{uint32_t(riB)} * irStride * 2u)[lane];
        if(lane == 0) flB = L.ctl[vB].flags;
    }
    // the direct filter pairs: lanes 1, 2 = voice A's (low-pass, high-pass), lanes 3, 4 = voice B's
    const bool fLane = lane >= 1u && lane <= 4u;
    const bool fB = lane >= 3u, fHp = ((lane - 1u) & 1u) != 0u;
    const bool fOn = fLane && (fB ? doB : doA);
    BiquadState *fp = &L.dfilt[size_t{fB ? vB : vA} * 2u + (fHp ? 1u : 0u)].f;
    float tb0 = 0.0f, tb1 = 0.0f, tb2 = 0.0f, ta1 = 0.0f, ta2 = 0.0f;
    int32_t fcounter = 0;
    if(fOn) { tb0 = fp->tb0; tb1 = fp->tb1; tb2 = fp->tb2; ta1 = fp->ta1; ta2 = fp->ta2; fcounter = fp->counter; }
    // ---- the stores
    auto installCtl = [&](uint32_t v, const RecScalars &r, uint32_t flagsOld, f2 row)
    {
        VoiceCtl &ctl = L.ctl[v];
        if(lane == 0)
        {
            ctl.step = r.step;
            ctl.rsKind = r.rsKind; ctl.rsM = r.rsM; ctl.rsL = r.rsL; ctl.rsSf = r.rsSf;
            ctl.rsFilterOffset = r.rsFilterOffset;
            const uint32_t keep = flagsOld & (kFlagFading | kFlagHasHrtf | kFlagAmbiScale | kFlagNfc | kFlagDelayed | kFlagQueue
                | (r.keepHrtf ? uint32_t(kFlagHrtfDirty) : 0u));
            ctl.flags = keep | (r.flags & ~(kFlagFading | kFlagHasHrtf | kFlagHrtfDirty | kFlagAmbiScale | kFlagNfc | kFlagDelayed | kFlagQueue))
                | kFlagHasHrtf | (r.keepHrtf ? 0u : uint32_t(kFlagHrtfDirty));
            for(int i = 0; i < 6; ++i) ctl.sendSlot[i] = -1;
            if(!r.keepHrtf)
            {
                ctl.hrtfTgtDelay[0] = r.delay0; ctl.hrtfTgtDelay[1] = r.delay1;
                ctl.hrtfTgtGain = r.gain;
            }
        }
        if(!r.keepHrtf && lane < irStride) reinterpret_cast<f2*>(L.hrtfTgt + size_t{v} * irStride * 2u)[lane] = row;
    };
    if(doA) installCtl(vA, sA, flA, rowA);
    if(doB) installCtl(vB, sB, flB, rowB);
    if(fOn)
    {   // BiquadSetTarget (dev_mix.hpp) on the values read above
        float c[5];
#pragma unroll
        for(int k = 0; k < 5; ++k) c[k] = fB ? (fHp ? sB.hp[k] : sB.lp[k]) : (fHp ? sA.hp[k] : sA.lp[k]);
        bool changed = !(fabsf(c[0] - tb0) <= 0.015625f);
        changed |= !(fabsf(c[1] - tb1) <= 0.015625f);
        changed |= !(fabsf(c[2] - tb2) <= 0.015625f);
        changed |= !(fabsf(c[3] - ta1) <= 0.015625f);
        changed |= !(fabsf(c[4] - ta2) <= 0.015625f);
        fp->tb0 = c[0]; fp->tb1 = c[1]; fp->tb2 = c[2]; fp->ta1 = c[3]; fp->ta2 = c[4];
        const bool snap = changed ? !(fcounter >= 0) : (fcounter <= 0);
        if(changed && fcounter >= 0) fp->counter = 256;
        if(snap) { fp->counter = 0; fp->b0 = c[0]; fp->b1 = c[1]; fp->b2 = c[2]; fp->a1 = c[3]; fp->a2 = c[4]; }
    }
}

// ACCL > 0: the context's mix lines (dry lines and / or the slots' wet lines, <= ACCL of them) accumulate in the
// wavefront's registers (MixRowAcc) instead of leaving stream rows; such kernels run the register-lean resampler.
// RES: the resident launch -- the kernel's body is one UPDATE of a loop that ends when the host says so (see NextBlock above);
// everything a launch does per update it does per turn of that loop, in the same order with the same operations: the bits are
// those of one launch per update (tests/test_gpu_pipeline.py).
template<int R, int TAPS, int NL, bool SENDS, bool MF = false, bool PROF = false, class LT = DeviceLayout, int ACCL = 0, bool RES = false>
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(kWThreads, OALGPU_WAVE_MIN_WG) VoiceWaveKernel(LT L, uint32_t samplesToDo, WaveProf prof, NextBlock next)
{
    static_assert(!RES || (MF && NL == 0 && !SENDS && !PROF && ACCL == 0 && std::is_same<LT, WaveArgsHrtf>::value), "the resident launch is the HRTF hot path's");
    static_assert(std::is_same<LT, DeviceLayout>::value || (NL == 0 && !SENDS), "the lean argument block is the HRTF variants'");
    static_assert(ACCL == 0 || NL > 0 || SENDS, "line accumulators belong to kernels that mix onto lines");
    static_assert(!MF || (R == 17 && TAPS == 64 && NL == 0), "the matrix-pipe FIR is the 64-tap HRTF form");
    using WL = WaveLds<R, TAPS, MF>;
    __shared__ WgLds<R, TAPS, MF> sm;
    const uint32_t t = threadIdx.x;
    const uint32_t lane0 = t & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint32_t group = blockIdx.x;
    const uint32_t vpw = L.waveVoices;
    const uint32_t irStride = L.irStride;
    WL &w = sm.w[wave];
    uint32_t N = samplesToDo;                   // (RES: every update brings its own length)
    // The argument block's pointers are needed in SGPRs all at once here, so that their kernarg loads are issued together and
    // waited for ONCE: left to itself the compiler loads each pointer where it spills it to a VGPR lane -- one scalar load and
    // one s_waitcnt after the other, a dozen dependent round trips in front of every wavefront's first useful instruction.
    if constexpr (std::is_same<LT, WaveArgsHrtf>::value)
        asm volatile("; argument block resident" :: "s"(L.tables), "s"(L.buffers), "s"(L.ctl), "s"(L.prev), "s"(L.dfilt), "s"(L.hrtfOld),
            "s"(L.hrtfTgt), "s"(L.hist), "s"(L.ambi), "s"(L.startDelay), "s"(L.queueDone), "s"(L.partHrtf), "s"(L.hrirs),
            "s"(next.recs), "s"(next.map), "s"(L.numVoices), "s"(L.waveVoices), "s"(L.irStride), "s"(L.pad), "s"(samplesToDo), "s"(gridDim.x));

    // The workgroup owns kWWaves*vpw consecutive voices; a wavefront takes every SECOND one of its
    // half of them (wave 0: v, v+2, ..; wave 1: v+1, v+3, ..).  Voices that cost more -- an active
    // filter, a replaced HRIR -- tend to come in regular patterns (every n-th source of a scene);
    // with consecutive voices per wavefront a period-4 pattern puts both expensive voices of a
    // group of four on the same wavefront, and the launch lasts as long as its slowest wavefront.
    const uint32_t vBegin = group * kWWaves * vpw + (wave & 1u) + 2u * (wave >> 1) * vpw;
    const uint32_t vEnd = (vBegin + 2u * vpw < L.numVoices) ? vBegin + 2u * vpw : L.numVoices;   // v = vBegin + 2k < vEnd
    const uint32_t vCount = vBegin < vEnd ? (vEnd - vBegin + 1u) / 2u : 0u;
    // PROF: per-wavefront stamps behind the per-voice ones
    auto waveStamp = [&](int slot)
    {
        if constexpr (PROF)
        {
            if(prof.times && lane0 == 0)
                prof.times[size_t{L.numVoices} * 8 + size_t{group * kWWaves + wave} * 8 + slot] = __builtin_readcyclecounter();
        }
    };
    const uint32_t ablate = PROF ? prof.ablate : 0u;
    waveStamp(0);
    // Two workgroups share a CU, and the launch fills the machine exactly once: workgroup g and
    // g + gridDim/2 land on the same CU (the dispatcher deals the first half one per CU, then the
    // second half).  Run in the same order, their wavefronts sit in the same phase at the same
    // time -- all eight in the LDS-bound resampler, then all eight in the VALU-bound FIR.  The
    // second half therefore takes its voices in reverse order: voices differ in cost, so the two
    // workgroups of a CU drift out of phase within the first voice and stay complementary.
    const bool rev = group >= (gridDim.x + 1u) / 2u;
    auto voiceAt = [&](uint32_t j) { return vBegin + 2u * (rev ? vCount - 1u - j : j); };

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
    // RES: thread and lane index are re-derived per update behind an opaque move, like the lane index per pass below: what the
    // body computes from them (the dump's and the partial's addresses) then is a few VALU per update, not a set of loop
    // invariants kept in registers across the resampler and the FIR, whose peaks decide the kernel's register count.
    uint32_t tU = t, laneU = lane0;
    if constexpr (RES) asm volatile("" : "+v"(tU), "+v"(laneU));
    const uint32_t t = tU, lane0 = laneU;
    bool stageRows = true;                      // pass 0 stages the workgroup's resampler rows (RES: only when they are not there yet)
    if constexpr (RES)
    {
        const ResidentArgs &RA = next.res;
        // Thread 0 waits for the update: the doorbell (or the word to leave), and -- kResidentSets updates on -- for the reduction
        // that still reads the partial set this update's sums go into.  The others wait at the barrier; what thread 0 learned
        // goes round through LDS.  The decision to leave is the host's alone (exitSeq, or the launch's own bound endSeq), so
        // that every workgroup takes it for the same update; the watchdog only turns a host that went away into an error.
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
                // one round of loads per look: the word to leave, the doorbell and the reduction's counter are in flight together
                unsigned long long t0 = 0ull;
                uint32_t waitedFor = 0u;                     // 1: the doorbell, 2: the reduction
                for(uint32_t spins = 0;; ++spins)
                {
                    const uint32_t ex = ResLoadSys(exitp), sq = ResLoadSys(&door->seq), rd = needRed ? ResLoadDev(rr) : want;
                    // (the word to leave first: a parked launch ends in front of the update the NEXT launch is being rung for)
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
            if((group & 127u) == 0u)       // (every 128th workgroup keeps the books: 32-bit sums of ticks over all of them wrap within seconds)
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
        // the update's parameter block: every wavefront installs the records of its own voices, as ApplyParamsKernel would in
        // front of a launch (the voices' state of the update before was written back by this very wavefront, in program order)
        if(mapU)
        {
            const int32_t *map = reinterpret_cast<const int32_t*>(mapU);
            const ParamRecord *recs = reinterpret_cast<const ParamRecord*>(recsU);
            const unsigned long long rowsU = (uint64_t{uint32_t(__builtin_amdgcn_readfirstlane(sm.res[9]))} << 32) | uint32_t(__builtin_amdgcn_readfirstlane(sm.res[8]));
            const float *rows = reinterpret_cast<const float*>(rowsU);
            for(uint32_t j = 0; j < vCount; j += 2u)
                InstallPair(L, map, recs, rows, vBegin + 2u * j, vBegin + 2u * j + 2u, j + 1u < vCount, lane0);
        }
        // What the update reads through the scalar cache -- the voices' control lines -- was written with vector stores, by this
        // wavefront and (the key voice's head) by wavefront 0: the stores are in L2 before the barrier, the scalar cache forgets
        // behind it.  (Nothing crosses workgroups: no fence of wider scope, which would write back and invalidate a whole L2.)
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
    // Voices are processed in passes; pass 0 only requests the first voice's source window and
    // stages the workgroup's resampler rows.  The request for the NEXT voice's window sits at one
    // point of the pass -- after this voice's FIR inputs are built, before its FIR runs -- so the
    // HBM latency of every window but the first is covered by ~1100 packed FMAs.
    VoiceHead headN{};
    VoiceTail tailN{};
    SrcPlan planN{false, 0u, 0u};
    BufferItem bufN{};
    bool loopingN = false;
    float fstC = 0.0f;                      // SENDS: the direct filter's state words, carried to the next pass
    // the voice whose resampler rows the workgroup stages (see the prologue below): every wavefront
    // reads its head itself, so that the choice needs no barrier.  (Requested first: both control lines are then in
    // flight together -- scalar loads return out of order, so the first wait for any of them waits for all.)
    // (Unconditional loads of a clamped index: behind a branch each request would be waited for where the branch joins.)
    const uint32_t keyVoice = group * kWWaves * vpw;
    const uint32_t lastVoice = L.numVoices - 1u;
    {
        const uint32_t v0 = vCount ? voiceAt(0) : lastVoice;        // (a wavefront without voices never uses what it reads here)
        headN = LoadHeadScalar(L.ctl + v0); bufN = LoadCtlBufferScalar(L.ctl + v0);
        if constexpr (NL == 0) tailN = LoadTailScalar(L.ctl + v0);
    }
    asm volatile("" ::: "memory");          // (the first voice's request is issued above this line ...)
    const VoiceHead headK = LoadHeadScalar(L.ctl + (keyVoice < L.numVoices ? keyVoice : lastVoice));
    asm volatile("" ::: "memory");          // (... the key voice's above this one: both are in flight at the first wait for either)
    const int psK = headK.playState, kK = headK.rsKind;
    const uint32_t mK = kK == 2 ? 4u : headK.rsM, lK = kK == 2 ? 1u : headK.rsL, offK = headK.rsFilterOffset;
    if constexpr (PROF) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); waveStamp(4); }

    f2 acc[MF ? 1 : R];
#pragma unroll
    for(int r = 0; r < (MF ? 1 : R); ++r) acc[r] = f2{0.0f, 0.0f};
    float invH = 1.0f;                               // MF: 1 / scale of the parked (next) voice's response
    float invHO = 1.0f;                              // MF: and of its replaced (old) response, if it has one
    bool fstClean = false;                           // the parked voice's direct filter pair is at rest (nothing to clear)
    float xoL = 0.0f, xoR = 0.0f;                    // MF: this voice's old-filter fade-out inputs, frame = lane
    f4 accM[2][5];                                   // MF: FirMfmaH's tiles, ear x (4 x 256 frames + ring-out)
#pragma unroll
    for(int e = 0; e < 2; ++e)
#pragma unroll
        for(int b = 0; b < 5; ++b) accM[e][b] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    f2 accO[WL::kQ];
#pragma unroll
    for(int q = 0; q < WL::kQ; ++q) accO[q] = f2{0.0f, 0.0f};
    float accRamp[ACCL > 0 ? ACCL : 1];                // ACCL: the gain ramps' share of the lines' first 64 frames, frame = lane
#pragma unroll
    for(int c = 0; c < (ACCL > 0 ? ACCL : 1); ++c) accRamp[c] = 0.0f;
    float accL[ACCL > 0 ? ACCL : 1][R];                // ACCL: the mix lines, lane l owns frames [R l, R l + R)
#pragma unroll
    for(int c = 0; c < (ACCL > 0 ? ACCL : 1); ++c)
#pragma unroll
        for(int r = 0; r < R; ++r) accL[c][r] = 0.0f;

    for(uint32_t pass = 0; pass <= vCount; ++pass)
    {
        // The lane index is re-derived per pass behind an opaque move: addresses built from it are
        // then a few VALU per voice instead of loop invariants that the allocator, already at the
        // 256-register limit of 2 waves/SIMD, would keep in scratch.
        uint32_t lane = lane0;
        asm volatile("" : "+v"(lane));
        const bool first = pass == 0;
        // Instruction arbitration between the two wavefronts of a SIMD favours the older one: the
        // workgroup that arrived first runs at full speed, its partner on what is left (measured:
        // 62 K against 78 K cycles for the same work, and the launch lasts until the slower one
        // ends, running alone for the last 16 K).  The younger workgroup therefore takes priority
        // for the first half of its voices and hands it back for the second.
        if(rev)
        {
            if(pass == 0) __builtin_amdgcn_s_setprio(1);
            else if(pass == vCount / 2u + 1u) __builtin_amdgcn_s_setprio(0);
        }
        const uint32_t v = first ? 0u : voiceAt(pass - 1u);  // meaningless in pass 0
        const bool haveNext = pass < vCount;
        const uint32_t vn = haveNext ? voiceAt(pass) : 0u;  // the voice to request (the first one in pass 0)
        // The next voice's request: these registers live from the request to the end of the pass,
        // where they are parked in LDS.
        float preN[kPre];
        float prevN = 0.0f, histN = 0.0f, fstN = 0.0f;
        f2 oldN[TAPS / 64];
        f2 hN = {0.0f, 0.0f};                        // MF: the next voice's target HRIR, tap = lane
        bool dirtyN = false;
#pragma unroll
        for(int q = 0; q < TAPS / 64; ++q) oldN[q] = f2{0.0f, 0.0f};
#pragma unroll
        for(int i = 0; i < kPre; ++i) preN[i] = 0.0f;
        auto requestNext = [&]()
        {
            if(haveNext)
            {   // its head is in headN
                planN = PlanSource(headN, N);
                loopingN = false;
                if(headN.curBuffer >= 0)
                {
                    // voice.cpp:1015-1019: a position at or past the loop end plays on without looping
                    loopingN = headN.loopBuffer >= 0 && ((headN.flags & kFlagQueue)
                        || !(headN.position >= 0 && uint32_t(headN.position) >= bufN.loopEnd));
                    planN.prefetch = planN.prefetch && GatherCovers(planN.bsrc, bufN, loopingN, uint32_t(headN.position));
                }
                if(headN.flags & (kFlagDelayed | kFlagQueue)) planN.prefetch = false;   // a delayed start's window depends on where in
                                                                                   // the update it starts; a queue's spans several buffers
                // (the window first: each gather variant starts by waiting for older loads into its
                // registers -- the variants share them -- and must not find a fresh one in front of it)
                if(planN.prefetch)
                {
                    GatherStatic(preN, planN.bsrc, bufN, loopingN, uint32_t(headN.position), lane);
                    prevN = (lane < kMaxPad) ? L.prev[size_t{vn} * kMaxPad + lane] : 0.0f;
                }
                fstN = (lane < 32u) ? reinterpret_cast<const float*>(L.dfilt + size_t{vn} * 2)[lane] : 0.0f;
                if constexpr (NL == 0)
                {
                    histN = L.hist[size_t{vn} * kHist + lane];
                    if constexpr (MF)
                        hN = (lane < irStride) ? reinterpret_cast<const f2*>(L.hrtfTgt + size_t{vn} * irStride * 2)[lane] : f2{0.0f, 0.0f};    // (taps beyond IrSize are stored as zeros)
                    dirtyN = (headN.flags & kFlagHrtfDirty) != 0;
                    const f2 *oc = reinterpret_cast<const f2*>(L.hrtfOld + size_t{vn} * irStride * 2);
    #pragma unroll
                    for(int q = 0; q < TAPS / 64; ++q)
                        oldN[q] = (dirtyN && lane + 64u * q < irStride) ? oc[lane + 64u * q] : f2{0.0f, 0.0f};
                }
            }
        };

        // ---------------- part 1: this voice up to its FIR inputs ----------------
        bool active = false;
        VoiceHead head{};
        BufferItem buf{};
        bool looping = false, playing = false, dirty = false, oldPass = false;
        VoiceTail tail{};
        uint32_t counter = 0, fademix = 0, todo = 0;
        float endGain = 0.0f, gainAfterBlend = 0.0f;
        int32_t bufferItem = -1;
        uint32_t outPos = 0;                    // delayed start (voice.cpp:1023-1046): the voice's samples begin here
        if(!first)
        {
            // head, plan, buffer and the gathered window were requested one pass ago
            head = headN;
            tail = tailN;
            buf = bufN;
            looping = loopingN;
            // next voice's head (and the HRTF delays/gains behind it): in flight while this voice resamples
            if(haveNext)
            {
                headN = LoadHeadScalar(L.ctl + vn); bufN = LoadCtlBufferScalar(L.ctl + vn);
                if constexpr (NL == 0) tailN = LoadTailScalar(L.ctl + vn);
            }

            const int vstate = head.playState;
            const bool mixes = vstate == OALGPU_VOICE_PLAYING || vstate == OALGPU_VOICE_STOPPING;
            playing = vstate == OALGPU_VOICE_PLAYING;
            active = mixes && head.step >= 1u;
            // voice.cpp:1002-1010
            if(mixes && !active && !playing && lane == 0) L.ctl[v].playState = OALGPU_VOICE_STOPPED;
            if(active && (head.flags & kFlagDelayed))
            {   // voice.cpp:1023-1046.  A delayed voice is on its first mix (no fade, empty history, cleared
                // filters): its N - outPos samples are produced at offset outPos of the sample line behind
                // zeros and everything downstream mixes a full line.
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
                    for(uint32_t k = lane; k < outPos; k += 64) w.in[kHist + k] = 0.0f;
                }
            }
            if constexpr ((SENDS || NL > 0) && ACCL == 0)
            {   // nothing to mix for this voice in this update
                if(!active && lane < L.streamsPerVoice)
                    L.lineGains[(size_t{v} * L.streamsPerVoice + lane) * LineBlockDwords(L.lineStride) + 3u * L.lineStride] = 0u;
            }
        }
        auto stamp = [&](int slot)
        {
            if constexpr (PROF) { if(prof.times && lane == 0) prof.times[size_t{v} * 8 + slot] = __builtin_readcyclecounter(); }
        };
        float invX = 1.0f;                            // MF: 1 / scale of this voice's FIR inputs
        if(active)
        {
            stamp(0);
            bufferItem = head.curBuffer;
            dirty = (head.flags & kFlagHrtfDirty) != 0;

            // Hrtf.History (w.in[0..63]), the direct filter state (w.fst; SENDS: fstC), a replaced
            // filter's old coefficients (w.cold) and the source window were parked by the last pass
            const float fstv = fstC;
            const SrcPlan plan = planN;
            // What the mix onto the lines starts from -- Gains.Target / Gains.Current of the dry lines (line = lane), the first
            // send's filter pair and gains (wet line = lane) and the sends' slots -- is requested HERE, in front of the
            // resampler: each was a dependent round trip to L2 of its own in the middle of the voice (1.5 K cycles a piece).
            float dryTgP = 0.0f, dryCuP = 0.0f, sfP = 0.0f, stP = 0.0f, scP = 0.0f;
            int32_t sendSlots[6] = {-1, -1, -1, -1, -1, -1};
            auto sendPrefetch = [&](uint32_t si)
            {
                if constexpr (SENDS)
                {
                    if(si < L.numSends)
                    {
                        const size_t vs = size_t{v} * L.numSends + si;
                        sfP = (lane < 32u) ? reinterpret_cast<const float*>(&L.sfilt[vs * 2])[lane] : 0.0f;
                        if(lane < L.wetChannels) { stP = L.sendTgt[vs * L.wetChannels + lane]; scP = L.sendCur[vs * L.wetChannels + lane]; }
                    }
                }
            };
            if constexpr (NL > 0)
            {
                if(lane < L.numDry) { dryTgP = L.gainTgt[size_t{v} * L.numDry + lane]; dryCuP = L.gainCur[size_t{v} * L.numDry + lane]; }
            }
            if constexpr (SENDS)
            {   // VoiceCtl::sendSlot, bytes 48..71 of the voice's control line: through the scalar cache
                static_assert(offsetof(VoiceCtl, sendSlot) == 48, "VoiceCtl::sendSlot follows the head");
                cu4 *src = (cu4*)(uintptr_t)(L.ctl + v);
                const u4 a = src[3], b = src[4];
                sendSlots[0] = int32_t(a.x); sendSlots[1] = int32_t(a.y); sendSlots[2] = int32_t(a.z); sendSlots[3] = int32_t(a.w);
                sendSlots[4] = int32_t(b.x); sendSlots[5] = int32_t(b.y);
                sendPrefetch(0);
            }
            LoadResampledWave<(ACCL > 0), PROF>(sm, w, L, v, lane, head, playing, N - outPos, N - outPos, bufferItem, looping, plan, outPos, prof);
            asm volatile("" : "+v"(lane));      // addresses used from here on are rebuilt, not carried across the resampler
            if constexpr (NL > 0) requestNext();
            if(head.flags & kFlagAmbiScale)
            {   // VoiceFlag::IsAmbisonic: mAmbiSplitter.processScale, voice.cpp:1082-1091
                const AmbiScaleState a = L.ambi[v];
                SplitterState sp{a.coeff, a.lpZ1, a.lpZ2, a.apZ1};
                WaveSync();
                SplitterScan<false>(sp, w.in + kHist + outPos, N - outPos, a.hfScale, a.lfScale, lane);
                WaveSync();
                if(lane == 0) { L.ambi[v].lpZ1 = sp.lpZ1; L.ambi[v].lpZ2 = sp.lpZ2; L.ambi[v].apZ1 = sp.apZ1; }
            }

            stamp(1);
            counter = (head.flags & kFlagFading) ? (N < 64u ? N : 64u) : 0u;     // voice.cpp:1093

            const bool directFilter = (head.flags & kFlagDirectFilter) && !(ablate & 8u);
            if constexpr (SENDS || NL > 0)
            {   // ---- stream rows that must leave before the direct filter overwrites w.in
                const uint32_t ls = L.lineStride, spv = L.streamsPerVoice, numSends = L.numSends, wetCh = L.wetChannels;
                const uint32_t wetBase = L.hrtf ? 0u : L.numDry;
                float *rowsV = ACCL ? nullptr : L.streams + size_t{v} * spv * kLine;
                uint32_t *blkV = ACCL ? nullptr : L.lineGains + size_t{v} * spv * LineBlockDwords(ls);
                RowLineGain row0;                       // the unfiltered row's merged gains, line = lane
                bool row0Live = false;
                if constexpr (NL > 0)
                {
                    if(!directFilter)
                    {   // MixSamples onto the dry lines rides on the unfiltered row (DoNfcMix: only
                        // the W line is mixed from the voice's own samples, voice.cpp:908-909)
                        const uint32_t nd = (L.nfc && (head.flags & kFlagNfc)) ? 1u : L.numDry, ndAll = L.numDry;
                        float tg = 0.0f, cu = 0.0f;
                        if(lane < nd)
                        {
                            tg = playing ? dryTgP : 0.0f;    // SilentCoeffs when Stopping
                            cu = counter ? dryCuP : tg;      // voice.cpp:1094-1112
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        if(lane < nd) { L.gainCur[size_t{v} * ndAll + lane] = g.newCur; row0.add(g); }
                        row0Live = true;
                    }
                }
                if constexpr (SENDS)
                {
                    for(uint32_t si = 0; si < numSends; ++si)
                    {
                        int32_t slot = sendSlots[0];
#pragma unroll
                        for(int k = 1; k < 6; ++k) slot = (si == uint32_t(k)) ? sendSlots[k] : slot;
                        const float sfC = sfP, stC = stP, scC = scP;     // this send's state, requested one send (or the resampler) ago
                        sendPrefetch(si + 1u);
                        uint32_t *blkS = ACCL ? nullptr : blkV + size_t{2u + si} * LineBlockDwords(ls);
                        if(slot < 0) { if(ACCL == 0 && lane == 0) blkS[3u * ls] = 0u; continue; }
                        const bool sendFilter = (head.flags >> (kFlagSendFilterShift + si)) & 1u;
                        // the send's gains onto its slot's wet lines (voice.cpp:978-979)
                        const uint32_t base = wetBase + uint32_t(slot) * wetCh;
                        const bool mine = lane >= base && lane < base + wetCh;
                        float tg = 0.0f, cu = 0.0f;
                        float *curp = L.sendCur + (size_t{v} * numSends + si) * wetCh + (lane - base);
                        {   // the wet lines' gains came in with wet line = lane: to the lanes of the slot's lines
                            const float tgS = __shfl(stC, int(lane - base)), cuS = __shfl(scC, int(lane - base));
                            if(mine)
                            {
                                tg = playing ? tgS : 0.0f;
                                cu = counter ? cuS : tg;
                            }
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        if(mine) *curp = g.newCur;
                        // filter state of this send
                        BiquadSlot *slots = &L.sfilt[(size_t{v} * numSends + si) * 2];
                        WaveSync();
                        if(lane < 32u) w.fst[lane] = sfC;
                        WaveSync();
                        if(sendFilter)
                        {   // its own filtered copy, built in the (currently unused) resampler scratch
                            float *tmp = w.rd;
                            for(uint32_t k = lane; k < N; k += 64) tmp[k] = w.in[kHist + k];
                            WaveSync();
                            WaveDoFilters(w.fst, slots, true, tmp + outPos, N - outPos, lane);
                            WaveSync();
                            RowLineGain r;
                            if(mine) r.add(g);
                            if constexpr (ACCL > 0) MixRowAcc<(ACCL > 0 ? ACCL : 1), R>(accL, accRamp, tmp, N, r, lane);
                            else
                            {
                            float *dst = rowsV + size_t{2u + si} * kLine;
                            for(uint32_t k = lane; k < uint32_t(kLine); k += 64) dst[k] = (k < N) ? tmp[k] : 0.0f;
                            StoreRowBlock(blkS, ls, lane, r, true);
                            }
                        }
                        else
                        {
                            WaveDoFilters(w.fst, slots, false, w.in + kHist, N, lane);
                            if(mine) row0.add(g);
                            row0Live = true;
                            if(ACCL == 0 && lane == 0) blkS[3u * ls] = 0u;
                        }
                    }
                }
                if constexpr (ACCL > 0)
                {
                    if(row0Live) { WaveSync(); MixRowAcc<(ACCL > 0 ? ACCL : 1), R>(accL, accRamp, w.in + kHist, N, row0, lane); }
                }
                else
                {
                if(row0Live)
                    for(uint32_t k = lane; k < uint32_t(kLine); k += 64) rowsV[k] = (k < N) ? w.in[kHist + k] : 0.0f;
                StoreRowBlock(blkV, ls, lane, row0, row0Live);
                }
                WaveSync();
            }

            // ---- DoFilters, direct path (voice.cpp:255-267): in place on w.in[kHist..]
            {
                if constexpr (SENDS) { if(lane < 32u) w.fst[lane] = fstv; }
                WaveSync();
                if(directFilter || !fstClean)
                    WaveDoFilters(w.fst, &L.dfilt[size_t{v} * 2], directFilter, w.in + kHist + outPos, N - outPos, lane);
                WaveSync();
            }
            if constexpr (NL > 0)
            {   // the direct-filtered row (voice.cpp:962-963 after an active DoFilters)
                const bool nfcV = L.nfc && (head.flags & kFlagNfc);
                const uint32_t ls = L.lineStride, spv = L.streamsPerVoice, nd = nfcV ? 1u : L.numDry;
                uint32_t *blk1 = ACCL ? nullptr : L.lineGains + (size_t{v} * spv + 1u) * LineBlockDwords(ls);
                if(directFilter)
                {
                    if constexpr (ACCL == 0)
                    {
                    float *dst = L.streams + (size_t{v} * spv + 1u) * kLine;
                    for(uint32_t k = lane; k < uint32_t(kLine); k += 64) dst[k] = (k < N) ? w.in[kHist + k] : 0.0f;
                    }
                    float tg = 0.0f, cu = 0.0f;
                    if(lane < nd)
                    {
                        tg = playing ? dryTgP : 0.0f;
                        cu = counter ? dryCuP : tg;
                    }
                    const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                    RowLineGain r;
                    if(lane < nd) { L.gainCur[size_t{v} * L.numDry + lane] = g.newCur; r.add(g); }
                    if constexpr (ACCL > 0) MixRowAcc<(ACCL > 0 ? ACCL : 1), R>(accL, accRamp, w.in + kHist, N, r, lane);
                    else StoreRowBlock(blk1, ls, lane, r, true);
                }
                else if(ACCL == 0 && lane == 0) blk1[3u * ls] = 0u;
                if(ACCL == 0 && L.nfc)
                {   // DoNfcMix, voice.cpp:911-931: one row per ambisonic order above 0, the voice's
                    // (direct-filtered) samples through that order's NFC section
                    const uint32_t rowBase = 2u + L.numSends, ndAll = L.numDry;
                    uint32_t line = 1;
                    for(uint32_t o = 1; o <= L.nfcOrders; ++o)
                    {
                        uint32_t *blkN = L.lineGains + (size_t{v} * spv + rowBase + o - 1u) * LineBlockDwords(ls);
                        if(!nfcV) { if(lane == 0) blkN[3u * ls] = 0u; continue; }
                        const uint32_t cnt = L.chansPerOrder[o];
                        WaveSync();
                        NfcWaveScan(L.nfc[v], o, w.in + kHist, w.rd, N, lane);
                        WaveSync();
                        float *dst = L.streams + (size_t{v} * spv + rowBase + o - 1u) * kLine;
                        for(uint32_t k = lane; k < uint32_t(kLine); k += 64) dst[k] = (k < N) ? w.rd[k] : 0.0f;
                        const bool mine = lane >= line && lane < line + cnt;
                        float tg = 0.0f, cu = 0.0f;
                        if(mine)
                        {
                            tg = playing ? dryTgP : 0.0f;
                            cu = counter ? dryCuP : tg;
                        }
                        const MixLineGain g = PrepareMixLine(cu, tg, counter, N);
                        RowLineGain r;
                        if(mine) { L.gainCur[size_t{v} * ndAll + lane] = g.newCur; r.add(g); }
                        StoreRowBlock(blkN, ls, lane, r, true);
                        line += cnt;
                    }
                }
            }

            stamp(2);
            // (matrix-pipe FIR: the next voice's request leaves now -- the FIR is too short to cover it, the
            // build of its inputs in front of it makes up for that)
            if constexpr (MF) requestNext();
            if constexpr (NL == 0)
            {
            // ---- DoHrtfMix, voice.cpp:827-902
            WaveSync();
            if(playing) __builtin_nontemporal_store(w.in[N + lane], &L.hist[size_t{v} * kHist + lane]);    // (read next by the next launch)

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
            // A voice whose target filter was not replaced since its last mix has Old == Target
            // (coefficients and delays; voice.cpp:869 / :1100), so the old-filter fade-out and
            // the new-filter fade-in of MixHrtfBlend act on the same taps and their gains are
            // summed.
            const bool merged = !dirty;

            // x'[i] = (In[64 - dL + i], In[64 - dR + i]) * g(i); zero pads on both sides
            const float *inL = w.in + (kHist - dL), *inR = w.in + (kHist - dR);
            const float gbase = gainAfterBlend - mainStep * float(fademix);
            if constexpr (MF)
            {   // per ear, as packed f16 pairs: lane l owns the frame pairs l + 64 j (frames 2 (l + 64 j), + 1), so
                // that a pair is one dword.  First the products and their largest magnitude, then the power-of-
                // two scale that puts that magnitude into [2^14, 2^15), the split and the stores.
                // Two frames are one register pair all the way: gains, products and scaling are packed operations, the
                // block maximum a v_max3 with |.| modifiers per two values (the scalar form was 5 VALU operations per
                // value here and 4 more in the split; this is 1.5 and 2.5).
                f2 xl[kLine / 128], xr[kLine / 128];
#pragma unroll
                for(int j = 0; j < kLine / 128; ++j)
                {   // 2 (lane + 64 j) + 1 <= 1023: inside w.in for any N
                    const uint32_t i = 2u * (lane + 64u * uint32_t(j));
                    xl[j] = f2{inL[i], inL[i + 1u]}; xr[j] = f2{inR[i], inR[i + 1u]};
                }
                const float fi0 = float(2u * lane);
                const f2 step2 = splat(mainStep), base2 = splat(gbase);
                float mxf = 0.0f;
#pragma unroll
                for(int j = 0; j < kLine / 128; ++j)
                {
                    const uint32_t i = 2u * (lane + 64u * uint32_t(j));
                    const f2 fi = f2{fi0 + float(128 * j), fi0 + float(128 * j + 1)};       // exact
                    f2 g = pkfma(step2, fi, base2);
                    if(j == 0)
                    {   // only the first 64 frames can lie inside the fade
#pragma unroll
                        for(int k = 0; k < 2; ++k)
                            if(i + uint32_t(k) < fademix)
                            {
                                float gf = newOn ? newStep * float(i + uint32_t(k)) : 0.0f;
                                if(merged && oldOn) gf += oldStep * float(fademix - (i + uint32_t(k)));
                                g[k] = gf;
                            }
                    }
                    if(N < uint32_t(kLine))
                    {   // (a short block: frames from N on are silence.  Selected, not multiplied away: w.in beyond N is
                        // whatever an earlier voice left there, and 0 * Inf or 0 * NaN would reach the ring-out frames)
                        if(i >= N) { g.x = 0.0f; xl[j].x = 0.0f; xr[j].x = 0.0f; }
                        if(i + 1u >= N) { g.y = 0.0f; xl[j].y = 0.0f; xr[j].y = 0.0f; }
                    }
                    const f2 a = xl[j] * g, b = xr[j] * g;
                    xl[j] = a; xr[j] = b;
                    mxf = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), mxf);
                    mxf = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(b.x), __builtin_fabsf(b.y)), mxf);
                }
                float sx;
                HalfScale(WaveMaxBits(__builtin_bit_cast(uint32_t, mxf)), sx, invX);
                if(lane < 32u)
                {   // frames -64 .. -1
                    w.xh[0][0][lane] = 0u; w.xh[0][1][lane] = 0u; w.xh[1][0][lane] = 0u; w.xh[1][1][lane] = 0u;
                }
                if(lane < uint32_t(kXhDw - 32 - kLine / 2))
                {   // frames 1024 .. 1103
                    const uint32_t d = 32u + kLine / 2 + lane;
                    w.xh[0][0][d] = 0u; w.xh[0][1][d] = 0u; w.xh[1][0][d] = 0u; w.xh[1][1][d] = 0u;
                }
                const f2 sx2 = splat(sx);
#pragma unroll
                for(int j = 0; j < kLine / 128; ++j)
                {
                    const uint32_t d = 32u + lane + 64u * uint32_t(j);
                    uint32_t hi, lo;
                    const f2 a = xl[j] * sx2, b = xr[j] * sx2;
                    SplitHalf2(a.x, a.y, hi, lo);
                    w.xh[0][0][d] = hi; w.xh[0][1][d] = lo;
                    SplitHalf2(b.x, b.y, hi, lo);
                    w.xh[1][0][d] = hi; w.xh[1][1][d] = lo;
                }
            }
            else
            {
            w.x2[lane] = f2{0.0f, 0.0f};
            if(TAPS > 64) w.x2[64 + lane] = f2{0.0f, 0.0f};
            for(uint32_t k = TAPS + N + lane; k < uint32_t(WL::kX); k += 64) w.x2[k] = f2{0.0f, 0.0f};
            if(!(ablate & 16u))
            {
                if(lane < N)
                {   // i = lane: the only pass that can touch the fade (fademix <= 64)
                    const uint32_t i = lane;
                    float g;
                    if(i < fademix)
                    {
                        g = newOn ? newStep * float(i) : 0.0f;
                        if(merged && oldOn) g += oldStep * float(fademix - i);
                    }
                    else g = gainAfterBlend + mainStep * float(i - fademix);
                    w.x2[TAPS + lane] = f2{inL[i] * g, inR[i] * g};
                }
                // all reads first, then all writes: in and x2 are members of one LDS object, so
                // the compiler keeps a read behind every earlier write (one LDS round trip per
                // frame row otherwise).  lane + 64 j <= 1023: inside w.in for any N.
                float xl[kLine / 64 - 1], xr[kLine / 64 - 1];
#pragma unroll
                for(int j = 0; j < kLine / 64 - 1; ++j)
                {
                    const uint32_t i = lane + 64u * uint32_t(j + 1);
                    xl[j] = inL[i]; xr[j] = inR[i];
                }
#pragma unroll
                for(int j = 0; j < kLine / 64 - 1; ++j)
                {
                    const uint32_t i = lane + 64u * uint32_t(j + 1);
                    const float g = __builtin_fmaf(mainStep, float(i), gbase);
                    if(i < N) w.x2[TAPS + lane + 64u * uint32_t(j + 1)] = f2{xl[j] * g, xr[j] * g};
                }
            }
            }
            // old-filter fade-out inputs (one per lane) and coefficients, replaced filters only
            oldPass = !merged && oldOn;
            if(oldPass)
            {
                f2 xo = {0.0f, 0.0f};
                if(lane < fademix)
                {
                    const float g = oldStep * float(fademix - lane);
                    xo = f2{w.in[kHist - odL + lane] * g, w.in[kHist - odR + lane] * g};
                }
                if constexpr (MF) { xoL = xo.x; xoR = xo.y; }      // staged over the main inputs once the main FIR is through
                else w.xo[lane] = xo;
            }
            WaveSync();
            }
            stamp(3);
        }

        // ---------------- the next voice's source window leaves HBM now ----------------
        // NL == 0: before the FIR, whose ~1100 packed FMAs cover the latency (matrix-pipe FIR: before the build
        // of its inputs, above).  NL > 0 (no FIR in this kernel): right after the resampler, ahead of the
        // filters and the stream-row stores.
        if constexpr (PROF) { if(first) waveStamp(7); }
        // pass 0: the resampler rows the workgroup stages (see the prologue below) are requested BEFORE the first voice's
        // window, as LDS-DMA gathers (global_load_lds_dword: lane l of the load for tap pair p fetches fil / phd element
        // 2 p + (l & 1) of phase row l >> 1 straight into tabF / tabP [p][l >> 1] -- no registers hold them, no ds_write, and
        // their round trip lies behind the issue of the window's loads instead of in front of the staging barrier)
        constexpr uint32_t kMaxM = 2u * uint32_t(WgLds<R, TAPS, MF>::kPairs);
        const bool eligK = keyVoice < L.numVoices && (kK == 2 || (kK == 3 && (mK == 12 || mK == 24 || mK == 48) && mK <= kMaxM))
            && (psK == OALGPU_VOICE_PLAYING || psK == OALGPU_VOICE_STOPPING);
        // RES: the rows staged by an earlier update of this launch are still there unless the key voice's resampler changed
        // (every wavefront reads the same head behind the update's barrier and the same LDS words: a uniform decision)
        if constexpr (RES) { if(first && upd != updBase) stageRows = !(eligK && sm.tabKey == offK * 8u + uint32_t(kK) && sm.tabM == mK); }
        if(first && eligK && stageRows)
        {
            typedef const __attribute__((address_space(1))) void *gvoidp;
            typedef __attribute__((address_space(3))) void *lvoidp;
            const float *src = L.tables + offK + size_t{lane >> 1} * (2u * mK) + (lane & 1u);
            for(uint32_t pp = wave; pp < mK / 2u; pp += kWWaves)
            {
                __builtin_amdgcn_global_load_lds((gvoidp)(src + 2u * pp), (lvoidp)&sm.tabF[32u * pp], 4, 0, 0);
                __builtin_amdgcn_global_load_lds((gvoidp)(src + mK + 2u * pp), (lvoidp)&sm.tabP[32u * pp], 4, 0, 0);
            }
        }
        if constexpr (NL > 0 || MF) { if(!active) requestNext(); }
        else requestNext();
        if constexpr (PROF) { if(first) waveStamp(5); }

        if(first && stageRows)
        {   // ---- workgroup prologue: pick and stage the resampler rows most voices will use
            // Normally those of the workgroup's first voice, whose head every wavefront has read
            // itself (no barrier, no second round trip at kernel start); if that voice does not
            // qualify, wavefront 0 looks for one that does.
            const uint32_t gBegin = group * kWWaves * vpw;
            uint32_t key = offK * 8u + uint32_t(kK), m = mK;
            if(eligK)
            {   // (the values to store are made here, behind an opaque move: as loop invariants the compiler kept them
                // in VGPRs across every pass for the one store of pass 0)
                uint32_t k0 = key, k1 = mK, k2 = lK;
                asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2));
                if(t == 0) { sm.tabKey = k0; sm.tabM = k1; sm.tabL = k2; }
            }
            else
            {
            if(wave == 0)
            {
                const uint32_t cand = gBegin + lane;
                bool eligible = false;
                uint32_t off = 0, m = 0, l = 0;
                int kind = 0;
                if(lane < kWWaves * vpw && cand < L.numVoices)
                {
                    const VoiceCtl &c = L.ctl[cand];
                    kind = c.rsKind; off = c.rsFilterOffset; m = c.rsM; l = c.rsL;
                    if(kind == 2) { m = 4; l = 1; }
                    eligible = (kind == 2 || (kind == 3 && (m == 12 || m == 24 || m == 48) && m <= kMaxM))
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
            key = sm.tabKey; m = sm.tabM;
            }
            if(eligK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the rows requested above are in LDS (and so is this wavefront's window in its registers)
            else
            if(key != 0xffffffffu)
            {
                const float *filter = L.tables + (key >> 3);
                for(uint32_t idx = t; idx < (m / 2u) * 32u; idx += kWThreads)
                {
                    const uint32_t p = idx >> 5, pi = idx & 31u;
                    const float *row = filter + pi * 2u * m;
                    {
                    sm.tabF[idx] = f2{row[2u * p], row[2u * p + 1u]};
                    sm.tabP[idx] = f2{row[m + 2u * p], row[m + 2u * p + 1u]};
                    }
                }
            }
            // zero padding of the old-filter coefficient array (never overwritten)
            if constexpr (MF)
            {   // zero padding of the reversed responses (every voice rewrites u in [17, 80] only)
                uint32_t *hz = &w.hr[0][0][0], *ho = &w.hro[0][0][0];
                for(uint32_t k = lane; k < uint32_t(4 * kHrDw); k += 64) { hz[k] = 0u; ho[k] = 0u; }
            }
            else
                for(uint32_t k = lane; k < uint32_t(TAPS + 128); k += 64) w.cold[k] = f2{0.0f, 0.0f};
            __syncthreads();
            waveStamp(6);
        }

        // ---------------- part 2: FIR and state write-back ----------------
        if(active)
        {
            stamp(4);
            cf16 *co = (cf16*)(uintptr_t)(L.hrtfTgt + size_t{v} * irStride * 2);
            if(NL > 0 || (ablate & 1u)) {}
            else if constexpr (MF)
            {
                FirMfmaH<5, false, (ACCL == 0)>(accM, w.xh, w.hr, invX * invH, lane);
                if(oldPass)
                {   // the replaced filter's fade-out (MixHrtfBlend, hrtfbase.h:54-70): 64 inputs x IrSize taps land in
                    // frames 0..126 -- the first eight columns of tile 0.  Its inputs go over the main inputs' first
                    // 104 dwords (frames -64..143; the main FIR has read them), as halves like those: frame = lane.
                    float sxo, invXo;
                    HalfScale(WaveMaxBits(__builtin_bit_cast(uint32_t, __builtin_fmaxf(__builtin_fabsf(xoL), __builtin_fabsf(xoR)))), sxo, invXo);
                    uint32_t hi, lo;
                    SplitHalf2(xoL * sxo, xoR * sxo, hi, lo);
                    WaveSync();
                    uint16_t *xz = reinterpret_cast<uint16_t*>(&w.xh[0][0][0]);
                    xz[0 * kXhHalves + 64u + lane] = uint16_t(hi); xz[1 * kXhHalves + 64u + lane] = uint16_t(lo);
                    xz[2 * kXhHalves + 64u + lane] = uint16_t(hi >> 16); xz[3 * kXhHalves + 64u + lane] = uint16_t(lo >> 16);
                    if(lane < 40u)
                    {
                        w.xh[0][0][64u + lane] = 0u; w.xh[0][1][64u + lane] = 0u; w.xh[1][0][64u + lane] = 0u; w.xh[1][1][64u + lane] = 0u;
                    }
                    WaveSync();
                    FirMfmaH<1, true, (ACCL == 0)>(accM, w.xh, w.hro, invXo * invHO, lane);
                }
            }
            else if(irStride == uint32_t(TAPS))
                FirMainPk<R, TAPS>(acc, &w.x2[TAPS + R * lane], co);
            else
            {   // other HRIR lengths: 16-tap segments (irStride is a multiple of 16, zero padded)
                const f2 *xw = &w.x2[TAPS + R * lane];
                for(uint32_t seg = 0; seg * 16u < irStride; ++seg)
                    FirMainPk<R, 16>(acc, xw - 16 * seg, co + 2 * seg);
            }
            if(NL == 0 && !MF && oldPass && !(ablate & 1u))
            {   // frames lane + 64q receive cOld[lane + 64q - i] * xo[i], i < 64
#pragma unroll 1
                for(int i0 = 0; i0 < 64; i0 += 8)
                {
#pragma unroll
                    for(int ii = 0; ii < 8; ++ii)
                    {
                        const int i = i0 + ii;
                        const f2 xi = w.xo[i];           // same address in every lane: LDS broadcast
#pragma unroll
                        for(int q = 0; q < WL::kQ; ++q) accO[q] = pkfma(w.cold[64 + lane + 64 * q - i], xi, accO[q]);
                    }
                }
            }
        }

        // ---------------- the next voice's state is parked in LDS ----------------
        // After the FIR every LDS word the next pass starts from is free (rd shares x2; in[0..63],
        // fst and cold were last read above), and what was requested before the FIR has landed.
        // Parked here, ahead of this voice's write-back stores, no wait of the next pass has to
        // sit behind those stores, and the ~21 registers of the request are dead outside the FIR.
        if(haveNext)
        {
            WaveSync();
            if(planN.prefetch)
            {
                const bool isShort = bufN.fmt == OALGPU_FMT_SHORT;
                if(lane < kMaxPad) { w.rd[lane] = prevN; if(lane) w.rd2[lane - 1u] = prevN; }
#pragma unroll
                for(int i = 0; i < kPre; ++i)
                {
                    // all kPre*64 words (rd has room; the resampler never reads past bsrc + padding), and the
                    // same one sample on (rd2[i] = rd[i + 1])
                    const float sv = GatherDecode(preN[i], isShort);
                    w.rd[kMaxEdge + lane + 64u * uint32_t(i)] = sv;
                    w.rd2[kMaxEdge - 1u + lane + 64u * uint32_t(i)] = sv;
                }
            }
            if constexpr (NL == 0)
            {
                w.in[lane] = histN;
                if constexpr (MF)
                {   // the response, tap = lane, as r[80 - lane] (FirMfmaH)
                    float sh;
                    HalfScale(WaveMaxBits(__builtin_bit_cast(uint32_t, __builtin_fmaxf(__builtin_fabsf(hN.x), __builtin_fabsf(hN.y)))), sh, invH);
                    invH = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, invH)));    // (uniform: stays in an SGPR across the pass)
                    uint32_t hi, lo;
                    SplitHalf2(hN.x * sh, hN.y * sh, hi, lo);       // (left, right) leading halves / remainders
                    uint16_t *hz = reinterpret_cast<uint16_t*>(&w.hr[0][0][0]);
                    const uint32_t u = 80u - lane;
                    hz[0 * kHrHalves + u] = uint16_t(hi); hz[1 * kHrHalves + u] = uint16_t(lo);
                    hz[2 * kHrHalves + u] = uint16_t(hi >> 16); hz[3 * kHrHalves + u] = uint16_t(lo >> 16);
                }
                if(dirtyN)
                {
                    if constexpr (MF)
                    {   // the replaced response, like the target's above
                        float sh;
                        HalfScale(WaveMaxBits(__builtin_bit_cast(uint32_t, __builtin_fmaxf(__builtin_fabsf(oldN[0].x), __builtin_fabsf(oldN[0].y)))), sh, invHO);
                        invHO = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, invHO)));
                        uint32_t hi, lo;
                        SplitHalf2(oldN[0].x * sh, oldN[0].y * sh, hi, lo);
                        uint16_t *hz = reinterpret_cast<uint16_t*>(&w.hro[0][0][0]);
                        const uint32_t u = 80u - lane;
                        hz[0 * kHrHalves + u] = uint16_t(hi); hz[1 * kHrHalves + u] = uint16_t(lo);
                        hz[2 * kHrHalves + u] = uint16_t(hi >> 16); hz[3 * kHrHalves + u] = uint16_t(lo >> 16);
                    }
                    else
                    {
#pragma unroll
                    for(int q = 0; q < TAPS / 64; ++q) w.cold[64 + lane + 64 * q] = oldN[q];
                    }
                }
            }
            if constexpr (SENDS) fstC = fstN;
            else if(lane < 32u) w.fst[lane] = fstN;
            {   // Is the pair at rest -- states and counter zero, current == target coefficients (what WaveDoFilters checks
                // before it clears an INACTIVE pair, voice.cpp:264-265)?  Decided here, on the words in flight, so that the
                // three voices in four without an active filter do not pay an LDS round trip to find nothing to do.
                const uint32_t li = lane & 15u;
                const uint32_t bits = __builtin_bit_cast(uint32_t, fstN);
                const float ahead = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, int(bits), 0x105, 0xF, 0xF, true));   // row_shl:5
                bool ok = true;
                if(li < 2u) ok = fstN == 0.0f;
                else if(li < 7u) ok = fstN == ahead;
                else if(li == 12u) ok = bits == 0u;
                fstClean = __ballot(!ok && lane < 32u) == 0ull;
            }
            WaveSync();
        }

        if(first) waveStamp(1);
        if(active)
        {
            stamp(5);
            // voice.cpp:1094-1101 / :869-873,900: Old <- Target, Old.Gain <- reached gain
            if(NL == 0 && dirty && (counter == 0 || fademix))
            {
                const float *tg = L.hrtfTgt + size_t{v} * irStride * 2;
                float *od = L.hrtfOld + size_t{v} * irStride * 2;
                for(uint32_t k = lane; k < irStride * 2; k += 64) od[k] = tg[k];
            }
            // ---- voice.cpp:1116-1232: flags, position, loop wrap / end of buffer ----
            if(lane == 0)
            {
                VoiceCtl &c = L.ctl[v];
                if(NL == 0)
                {
                    if(counter == 0 || fademix) { c.hrtfOldDelay[0] = tail.tgtDelay[0]; c.hrtfOldDelay[1] = tail.tgtDelay[1]; }
                    c.hrtfOldGain = todo ? endGain : gainAfterBlend;
                }
                uint32_t flags = (head.flags | kFlagFading) & ~kFlagDelayed;
                if(NL > 0 || counter == 0 || fademix) flags &= ~kFlagHrtfDirty;
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
            WaveSync();
            stamp(6);
        }
    }

    waveStamp(2);
    // the workgroup's stream rows onto the mix lines: one partial bus per workgroup (HRTF voices with sends:
    // behind the accumulator dump, whose registers are free by then)
    auto mixRows = [&]()
    {
        if constexpr (NL > 0 || SENDS)
        {
        static_assert(sizeof(sm) >= (kMixListMax + 4u + kMixGainDwords) * sizeof(uint32_t), "the row mix stages its gain blocks in the voices' LDS");
        uint32_t *lds = reinterpret_cast<uint32_t*>(&sm);
        const uint32_t vWg = group * kWWaves * vpw;
        const uint32_t nvWg = vWg < L.numVoices ? ((vWg + kWWaves * vpw < L.numVoices) ? kWWaves * vpw : L.numVoices - vWg) : 0u;
        if(L.lineStride <= 8u) WgMixRows<8>(lds, L, group, vWg, nvWg, t, N);
        else if(L.lineStride <= 16u) WgMixRows<16>(lds, L, group, vWg, nvWg, t, N);
        else WgMixRows<32>(lds, L, group, vWg, nvWg, t, N);
        waveStamp(3);
        }
    };
    // ACCL: the wavefronts' line accumulators -> one partial bus per workgroup, two lines at a time through the (now free)
    // FIR-input area of every wavefront's LDS, summed in wavefront order like the HRTF accumulator below
    auto dumpLines = [&]()
    {
        if constexpr (ACCL > 0)
        {
        auto areaOf = [&](int ww) { return MF ? reinterpret_cast<f2*>(&sm.w[ww].xh[0][0][0]) : sm.w[ww].x2; };
        static_assert(!MF || sizeof(w.xh) >= size_t(WL::kFrames) * sizeof(f2), "a line pair fits in the FIR-input area");
        const uint32_t nlines = L.mixLines;
        float *pl = L.partLines + size_t{group} * nlines * kLine;
#pragma unroll
        for(int c = 0; c < ACCL; c += 2)
        {
            if(uint32_t(c) >= nlines) break;
            __syncthreads();
            f2 *area = areaOf(int(wave));
#pragma unroll
            for(int r = 0; r < R; ++r) area[R * lane0 + r] = f2{accL[c][r], (c + 1 < ACCL) ? accL[c + 1 < ACCL ? c + 1 : c][r] : 0.0f};
            WaveSync();
            {   // the ramps' share: frame = lane
                const f2 cur = area[lane0];
                area[lane0] = f2{cur.x + accRamp[c], cur.y + ((c + 1 < ACCL) ? accRamp[c + 1 < ACCL ? c + 1 : c] : 0.0f)};
            }
            __syncthreads();
            for(uint32_t k = t; k < uint32_t(kLine); k += kWThreads)
            {
                f2 sum = areaOf(0)[k];
#pragma unroll
                for(int ww = 1; ww < kWWaves; ++ww) { const f2 o = areaOf(ww)[k]; sum.x += o.x; sum.y += o.y; }
                StorePartial(&pl[size_t(c) * kLine + k], (k < N) ? sum.x : 0.0f);
                if(uint32_t(c) + 1u < nlines) StorePartial(&pl[size_t(c + 1) * kLine + k], (k < N) ? sum.y : 0.0f);
            }
        }
        waveStamp(3);
        }
    };
    if constexpr (NL > 0) { if constexpr (ACCL > 0) dumpLines(); else mixRows(); }
    else
    {
    // ---- one partial per workgroup: waves dump their accumulators, then a fixed-order sum
    {
        // [frame] = (L, R), frames < 64R (the matrix-pipe kernels: 17 entries per 16 frames, in the place of xh)
        auto dumpOf = [&](int ww) { return MF ? reinterpret_cast<f2*>(&sm.w[ww].xh[0][0][0]) : sm.w[ww].x2; };
        static_assert(!MF || sizeof(w.xh) >= (WL::kFrames + WL::kFrames / 16) * sizeof(f2), "the padded dump fits in xh");
        f2 *dump = dumpOf(int(wave));
        WaveSync();
        if constexpr (MF)
        {   // FirMfmaH's tiles -> frames; frame f at dump[f + f / 16]: lanes of one column then are 17 entries
            // (34 banks) apart instead of 16 (all on the same two banks)
            const uint32_t jc = lane0 & 15u, q4 = lane0 >> 4;
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
        }
        else
        {
#pragma unroll
        for(int r = 0; r < R; ++r) dump[R * lane0 + r] = acc[r];
        }
        WaveSync();
        auto dumpAt = [](uint32_t f) { return MF ? f + (f >> 4) : f; };
        if constexpr (!MF)
        {   // (the matrix-pipe kernels add a replaced filter's fade-out straight into tile 0)
#pragma unroll
        for(int q = 0; q < WL::kQ; ++q)
        {
            const f2 cur = dump[dumpAt(lane0 + 64 * q)];
            dump[dumpAt(lane0 + 64 * q)] = f2{cur.x + accO[q].x, cur.y + accO[q].y};
        }
        }
        __syncthreads();
        f2 *ph = reinterpret_cast<f2*>(L.partHrtf) + size_t{group} * (kLine + kHrirLen);
        if constexpr (RES)      // the update's set of partial buses
            ph = reinterpret_cast<f2*>(next.res.partBase + size_t{upd % kResidentSets} * next.res.setStride) + size_t{group} * (kLine + kHrirLen);
        for(uint32_t k = t; k < uint32_t(kLine + kHrirLen); k += kWThreads)
        {
            f2 s = {0.0f, 0.0f};
            if(k < uint32_t(WL::kFrames))
            {
                s = dumpOf(0)[dumpAt(k)];
#pragma unroll
                for(int ww = 1; ww < kWWaves; ++ww) { const f2 o = dumpOf(ww)[dumpAt(k)]; s.x += o.x; s.y += o.y; }
            }
            if constexpr (RES) StorePartialCoherent(&ph[k], s);       // (read by the reduction's launch while this one runs on)
            else StorePartial(&ph[k], s);
        }
    }
    waveStamp(3);
    if constexpr (SENDS) { if constexpr (ACCL > 0) dumpLines(); else mixRows(); }
    }
    // ---- the next update's parameter block: every wavefront installs the records of the voices it has just mixed.
    // The voices' state was written back by this very wavefront, in program order, so its loads see it; the kernel boundary makes
    // the result visible to the next launch, as it did for the parameter kernel this replaces.  (The dry-line kernels too: what
    // their tail reads -- stream rows and gain blocks -- is none of what a record writes.)
    if(!RES && next.map)
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        bool done = false;
        if constexpr (std::is_same<LT, WaveArgsHrtf>::value)
        {   // the block carries its records' blended HRIRs: two voices per round of loads (InstallPair)
            if(next.rows && L.irStride <= 64u)       // (InstallPair moves ONE tap pair per lane: responses of up to 64 taps)
            {
                for(uint32_t j = 0; j < vCount; j += 2u)
                    InstallPair(L, next.map, next.recs, next.rows, vBegin + 2u * j, vBegin + 2u * j + 2u, j + 1u < vCount, lane0);
                done = true;
            }
        }
        if(!done)
        for(uint32_t j = 0; j < vCount; ++j)
        {
            const uint32_t v = vBegin + 2u * j;
            const int32_t ri = __builtin_amdgcn_readfirstlane(next.map[v]);
            if(ri >= 0) ApplyNextRecord(L, next.recs[ri], lane0);
        }
    }
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

#ifndef OALGPU_WAVE_NO_LAUNCHER          // (register-budget experiments instantiate single variants of the kernel: tools/kernel_regs.sh)
bool WaveKernelApplies(bool exact, const DeviceLayout &L)
{
    if(exact || L.mixLines > 32 || L.numSends > 6) return false;
    if(L.hrtf) return L.irStride >= 8 && L.irStride <= 128;
    return L.numDry >= 1;
}

// Few enough mix lines (the dry lines of a non-HRTF context plus every slot's wet lines) accumulate in registers.  Near-field
// control adds rows per ambisonic order and stays on the stream-row path (oalgpu_context_set_nfc clears accLines).
uint32_t WaveKernelAccLines(const DeviceLayout &L)
{
    // The register budget (256 per lane at two wavefronts per SIMD) decides which shapes exist: 6 lines beside the
    // resampler and the filter scans of a dry-line context without sends; 4 wet lines -- one first-order slot -- beside the
    // HRTF accumulator tiles.  More lines, or dry lines AND sends, go through stream rows.
    if(L.nfc || L.mixLines == 0) return 0;
    if(!L.hrtf) return (L.numSends == 0 && L.mixLines <= 6) ? 6u : 0u;
    if(L.numSends && L.irStride <= 64 && L.firMfma && L.mixLines <= 4) return 4u;
    return 0;
}

const char *WaveKernelName(const DeviceLayout &L)
{
    if(L.wave16) return Wave16KernelName(L);
    if(L.rows8) return RowsKernelName();
    if(L.sliceLines) return SliceKernelName();
    const bool sends = L.numSends != 0;
    if(L.accLines)
        return !L.hrtf ? "VoiceWaveKernel<17, 64, 1, false, false, false, DeviceLayout, 6>" : "VoiceWaveKernel<17, 64, 0, true, true, false, DeviceLayout, 4>";
    if(!L.hrtf) return sends ? "VoiceWaveKernel<17, 64, 1, true>" : "VoiceWaveKernel<17, 64, 1, false>";
    if(L.irStride <= 64 && L.firMfma) return sends ? "VoiceWaveKernel<17, 64, 0, true, true>" : "VoiceWaveKernel<17, 64, 0, false, true>";
    if(L.irStride <= 64) return sends ? "VoiceWaveKernel<17, 64, 0, true>" : "VoiceWaveKernel<17, 64, 0, false>";
    return sends ? "VoiceWaveKernel<18, 128, 0, true>" : "VoiceWaveKernel<18, 128, 0, false>";
}

uint32_t WaveKernelGroups(const DeviceLayout &L)
{
    if(L.wave16) return Wave16Groups(L);
    if(L.rows8) return (L.numVoices + L.rowsVpg - 1u) / L.rowsVpg;
    return (L.numVoices + kWWaves * L.waveVoices - 1u) / (kWWaves * L.waveVoices);
}

// prof: null in production; the measurement variants exist for the HRTF kernels without sends only
// evStart / evStop (both or neither): HIP events bound to the DISPATCH (hipExtLaunchKernel) -- the kernel's own start and end,
// what rocprofv3's kernel trace reports, without the command-processor time an event recorded around the launch includes
// the kernels whose wavefronts install a parameter block behind their voices: every VoiceWaveKernel (voice_slice.hip has no such epilogue)
bool WaveKernelAppliesRecords(const DeviceLayout &L) { return L.sliceLines == 0 && !(L.wave16 && L.numSends); }      // (every VoiceWaveKernel; not the slice kernel, not voice_wave16.hip's with sends)

hipError_t LaunchStreamRowsMix(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, uint32_t vpg, hipEvent_t evStop)
{
    const dim3 grid((L.numVoices + vpg - 1u) / vpg), block(kWThreads);
    if(L.lineStride <= 8u) hipExtLaunchKernelGGL(StreamRowsMixKernel<8>, grid, block, 0, s, nullptr, evStop, 0u, L, samplesToDo, vpg);
    else if(L.lineStride <= 16u) hipExtLaunchKernelGGL(StreamRowsMixKernel<16>, grid, block, 0, s, nullptr, evStop, 0u, L, samplesToDo, vpg);
    else hipExtLaunchKernelGGL(StreamRowsMixKernel<32>, grid, block, 0, s, nullptr, evStop, 0u, L, samplesToDo, vpg);
    return hipGetLastError();
}

hipError_t LaunchVoiceWave(hipStream_t s, const DeviceLayout &L, uint32_t samplesToDo, const WaveProf *prof, hipEvent_t evStart, hipEvent_t evStop,
    const ParamRecord *nextRecs, const int32_t *nextMap, const float *nextRows)
{
    if(L.wave16) return LaunchVoiceWave16(s, L, samplesToDo, prof, evStart, evStop, nextRecs, nextMap, L.hrtf ? nextRows : nullptr);   // (voice_wave16.hip)
    if(L.rows8) return LaunchVoiceRows(s, L, samplesToDo, prof, evStart, evStop, nextRecs, nextMap);      // (voice_rows.hip)
    if(L.sliceLines) return LaunchVoiceSlice(s, L, samplesToDo, evStart, evStop);       // (voice_slice.hip)
    const NextBlock next{nextRecs, nextMap, L.hrtf ? nextRows : nullptr, ResidentArgs{}};
    const uint32_t groups = WaveKernelGroups(L);
    const bool sends = L.numSends != 0;
    const dim3 grid(groups), block(kWThreads);
    const WaveProf none{nullptr, 0u};
    if(prof && L.hrtf && !sends && L.irStride <= 64)
    {
        if(L.firMfma) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, false, true, true, WaveArgsHrtf>), grid, block, 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, samplesToDo, *prof, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, false, false, true, WaveArgsHrtf>), grid, block, 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, samplesToDo, *prof, next);
    }
    else if(L.accLines && !L.hrtf)
    {   // (the measurement variant of these: the same kernel with PROF)
        if(prof) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, false, false, true, DeviceLayout, 6>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, *prof, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, false, false, false, DeviceLayout, 6>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
    }
    else if(L.accLines && L.hrtf)
    {
        if(prof) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, true, true, true, DeviceLayout, 4>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, *prof, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, true, true, false, DeviceLayout, 4>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
    }
    else if(prof && !L.hrtf)
    {
        if(sends) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, true, false, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, *prof, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, false, false, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, *prof, next);
    }
    else if(prof && sends && L.irStride <= 64 && L.firMfma)
        hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, true, true, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, *prof, next);
    else if(!L.hrtf)
    {
        if(sends) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 1, false>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
    }
    else if(L.irStride <= 64 && L.firMfma)
    {
        if(sends) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, true, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, false, true, false, WaveArgsHrtf>), grid, block, 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, samplesToDo, none, next);
    }
    else if(L.irStride <= 64)
    {
        if(sends) hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, false, false, false, WaveArgsHrtf>), grid, block, 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, samplesToDo, none, next);
    }
    else
    {
        if(sends) hipExtLaunchKernelGGL((VoiceWaveKernel<18, 128, 0, true>), grid, block, 0, s, evStart, evStop, 0u, L, samplesToDo, none, next);
        else hipExtLaunchKernelGGL((VoiceWaveKernel<18, 128, 0, false, false, false, WaveArgsHrtf>), grid, block, 0, s, evStart, evStop, 0u, WaveArgsHrtf{L}, samplesToDo, none, next);
    }
    return hipGetLastError();
}

#endif

} // namespace oalgpu
