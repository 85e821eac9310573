// HRTF post-process of one update, FAST mode: DeviceBase::Process(HrtfPostProcess),
// alc/alu.cpp:289-298 -> MixDirectHrtf_* (MixDirectHrtfBase, core/mixer/hrtfbase.h:91-133).
//
// The post-process runs on the context's post stream BESIDE the next update's voice kernel, which fills the machine
// exactly once (two workgroups per CU, 2 x 240 of 512 VGPRs per SIMD lane, 2 x 58 of 128 LDS granules): whatever
// does not fit into what is left of a CU waits until a voice workgroup leaves one and then holds that CU against
// the next launch.  Round 2's form -- ONE workgroup of a wavefront per channel with 147 KB of LDS and a
// register-tiled FIR -- was such an intruder (16 us with a 64-tap decoder, 38 us with the 87 taps InitHrtfPanning's
// real decoder has).  It is now three launches of small pieces that fit into the gaps (<= 32 VGPRs, <= 6.4 KB LDS):
//   PostSplitKernel   one wavefront per dry (ambisonic) channel: BandSplitter::processHfScale
//                     (core/filters/splitter.cpp:65-97), a 3-state linear recurrence, as a block scan;
//   PostFirKernel     one wavefront per 16 output frames: every channel's decoder HRIR (the same input for both
//                     ears, hrtfbase.h:104-117) from LDS copies of the channels' windows and coefficients;
//                     lane = (frame, ear, half of the taps);
//   PostShiftKernel   carried HrtfAccumData + the channels' sum -> RealOut L/R, and the accumulator's shift
//                     (hrtfbase.h:119-132).
// EXACT mode keeps the term-by-term kernel in percall_kernels.hip.
#include <hip/hip_ext.h>
#include "dev_wave.hpp"
#include "dev_reduce.hpp"

#pragma clang fp contract(off)

namespace oalgpu {
namespace {

constexpr int kPostFrames = kLine + kHrirLen;             // HrtfAccumData: 1152 frames
constexpr int kPostGroup = 4;                               // channels staged at a time

__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) PostSplitKernel(const float *__restrict__ in, float *__restrict__ xf,
    SplitterState *__restrict__ splitters, const float *__restrict__ hfscales, uint32_t n)
{
    __shared__ float x[kLine + 64];
    const uint32_t c = blockIdx.x, lane = threadIdx.x;
    for(uint32_t k = lane; k < uint32_t(kLine); k += 64) x[k] = (k < n) ? in[size_t{c} * kLine + k] : 0.0f;
    WaveSync();
    SplitterState st = splitters[c];
    SplitterScanHfTri(st, x, n, hfscales[c], lane);
    if(lane == 0) splitters[c] = st;
    WaveSync();
    for(uint32_t k = lane; k < uint32_t(kLine); k += 64) xf[size_t{c} * kLine + k] = (k < n) ? x[k] : 0.0f;
}

// tmp[o] = sum over the channels c and the taps t < taps of coeffs[c][t] * x_c[o - t], o in [16 b, 16 b + 16) (kPostBlock = 16)
template<int kPostBlock /* output frames per workgroup: 8 or 16 */>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(32))) PostFirKernel(const float *__restrict__ xf, uint32_t nch, const float *__restrict__ chanCoeffs,
    uint32_t taps /* multiple of 16, <= 128 */, float *__restrict__ tmp)
{
    __shared__ float xs[kPostGroup][kHrirLen + kPostBlock];            // x_c[16 b - 128 + j]
    __shared__ float cs[kPostGroup][kHrirLen][2];
    constexpr uint32_t kSplit = 32u / uint32_t(kPostBlock);       // the taps go over 2 or 4 groups of lanes
    const uint32_t lane = threadIdx.x, f = lane & uint32_t(kPostBlock - 1), e = (lane / uint32_t(kPostBlock)) & 1u,
        hq = lane / uint32_t(2 * kPostBlock);                      // frame, ear, part of the taps
    const int32_t base = int32_t(blockIdx.x) * kPostBlock - kHrirLen;
    const uint32_t half = taps / kSplit;
    float acc = 0.0f;
    for(uint32_t c0 = 0; c0 < nch; c0 += kPostGroup)
    {
        const uint32_t gc = (nch - c0 < uint32_t(kPostGroup)) ? nch - c0 : uint32_t(kPostGroup);
        __syncthreads();
        for(uint32_t c = 0; c < gc; ++c)
        {   // (no index arithmetic with divisions: the kernel has to stay inside 32 VGPRs)
            for(uint32_t j = lane; j < uint32_t(kHrirLen + kPostBlock); j += 64)
            {
                const int32_t fr = base + int32_t(j);
                xs[c][j] = (fr >= 0 && fr < kLine) ? xf[size_t{c0 + c} * kLine + uint32_t(fr)] : 0.0f;
            }
            for(uint32_t r = lane; r < taps * 2u; r += 64) (&cs[c][0][0])[r] = chanCoeffs[size_t{c0 + c} * kHrirLen * 2 + r];
        }
        __syncthreads();
        for(uint32_t c = 0; c < gc; ++c)
        {
            const float *xw = &xs[c][kHrirLen + f] - hq * half;       // x_c[o - t], t = hq * half + k
            const float *cw = &cs[c][hq * half][e];
            // (four taps in flight: the kernel has to fit into the 32 VGPRs per SIMD lane two voice wavefronts leave)
#pragma unroll 1
            for(uint32_t k = 0; k < half; k += 4)
            {
                const float c0 = cw[2u * k], c1 = cw[2u * k + 2u], c2 = cw[2u * k + 4u], c3 = cw[2u * k + 6u];
                const float x0 = xw[-int32_t(k)], x1 = xw[-int32_t(k) - 1], x2 = xw[-int32_t(k) - 2], x3 = xw[-int32_t(k) - 3];
                acc = __builtin_fmaf(c0, x0, acc); acc = __builtin_fmaf(c1, x1, acc);
                acc = __builtin_fmaf(c2, x2, acc); acc = __builtin_fmaf(c3, x3, acc);
            }
        }
    }
    if constexpr (kSplit == 4u) acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if(lane < uint32_t(2 * kPostBlock)) tmp[(size_t{blockIdx.x} * kPostBlock + f) * 2u + e] = acc;
}

__global__ void __launch_bounds__(256) PostShiftKernel(float *__restrict__ left, float *__restrict__ right, float *__restrict__ accum,
    const float *__restrict__ tmp, uint32_t n)
{
    const uint32_t t = threadIdx.x;
    const f2 *acc2 = reinterpret_cast<const f2*>(accum), *tmp2 = reinterpret_cast<const f2*>(tmp);
    f2 *accOut = reinterpret_cast<f2*>(accum);
    constexpr int kPer = (kPostFrames + 255) / 256;
    f2 tot[kPer];
#pragma unroll
    for(int k = 0; k < kPer; ++k)
    {
        const uint32_t o = t + 256u * uint32_t(k);
        f2 s = {0.0f, 0.0f};
        if(o < uint32_t(kPostFrames))
        {
            const f2 a = acc2[o], b = tmp2[o];
            s = f2{a.x + b.x, a.y + b.y};
            if(o < n) { left[o] = left[o] + s.x; right[o] = right[o] + s.y; }
        }
        tot[k] = s;
    }
    __syncthreads();                           // every thread has read the old accumulator
    // hrtfbase.h:127-132: frames [n, n+128) move to the front, the following n frames are cleared, anything beyond stays
#pragma unroll
    for(int k = 0; k < kPer; ++k)
    {
        const uint32_t o = t + 256u * uint32_t(k);
        if(o >= n && o < n + kHrirLen && o < uint32_t(kPostFrames)) accOut[o - n] = tot[k];
    }
    __syncthreads();
#pragma unroll
    for(int k = 0; k < kPer; ++k)
    {
        const uint32_t o = t + 256u * uint32_t(k);
        if(o >= uint32_t(kHrirLen) && o < kHrirLen + n && o < uint32_t(kPostFrames)) accOut[o] = f2{0.0f, 0.0f};
    }
}

// The three steps above as ONE launch (the post stream of a pipelined context: every launch there costs ~4 us of dispatch
// floor plus its wait for a CU slot beside the voice kernel, and the chain of four -- reduction, split, FIR, shift -- was as
// long as the voice kernel it runs beside).  Workgroups [0, nch): PostSplitKernel's work, one dry channel each; they hand
// the filtered channel over through device-scope relaxed atomics (write-through stores, L2-coherent loads of exactly those
// words -- no fence, which at device scope would write back and invalidate a whole L2 under the voice kernel; the idiom of
// the convolution kernel) and then raise a counter.  Workgroups [nch, nch + 18): PostFirKernel's work, one wavefront per 64
// output frames, once the counter says that every channel of THIS update has arrived (`epoch` = what the counter reads
// then; it only ever grows) -- they sleep-poll until then, which is safe because the dispatcher places a grid's
// workgroups in index order: the ones that are waited for are resident before the ones that wait.  The shift rides on the
// FIR's result: frame o of HrtfAccumData + the channels' sum goes to RealOut (o < n) and to its place in the carried
// accumulator (hrtfbase.h:119-132), which is a buffer of its own (`carryOut`; the reduction of the next update adds it to
// the voices' partial sums), so that no workgroup writes what another still reads.  The splitter states are read from one
// buffer and filed in the other for the same reason (a workgroup that starts late must not see the new state).
__device__ __forceinline__ void PostStoreCoherent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float PostLoadCoherent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

constexpr int kFusedFrames = 64;                            // output frames per FIR workgroup: one per lane, both ears

// Every wait inside these kernels is bounded: a counter that never arrives (a launch that failed on the host's side, a voice
// kernel that gave up) becomes an error word the host reads, not a hung GPU.  `ready` is asked until it says yes or 2 s have
// passed (s_memrealtime: 100 MHz).
// waited (resident contexts' measurement words, kernels.hpp kRcWait*): the wait's length in ticks is added there (null: nowhere)
template<class F>
__device__ __forceinline__ bool PostWait(F ready, int nap, uint32_t *waited = nullptr)
{
    if(ready()) return true;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = false;
    for(uint32_t spins = 0;; ++spins)
    {
        if(nap <= 4) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(8);
        if(ready()) { ok = true; break; }
        if((spins & 63u) == 63u && __builtin_amdgcn_s_memrealtime() - t0 > kResidentWatchdogTicks) break;
    }
    if(waited) __hip_atomic_fetch_add(waited, uint32_t(__builtin_amdgcn_s_memrealtime() - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ok;
}
__device__ __forceinline__ bool PostWaitCounter(const uint32_t *counter, uint32_t target, int nap, uint32_t *waited = nullptr)
{
    return PostWait([&]() { return int32_t(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0; }, nap, waited);
}
// One wavefront's share of the fused post-process: wg < nch = the split of dry channel wg, else FIR block wg - nch; outTarget: what `outArrived` reads when this launch's FIR workgroups are all through.
// MODE 1 (FUSED): the bus block was reduced by workgroups of the SAME launch (ReducePostFusedKernel): `reduced` reaches
// `reducedEpoch` when they are all through, and what they wrote is read with L2-coherent loads.
// MODE 2 (resident contexts): the reduction is a launch of its own on another stream (BusReduceResidentKernel) that may still be
// running, and the NEXT update's reduction is waiting for this launch's FIR workgroups (`postDone`) before it touches the bus
// block and reads the carried accumulator: besides MODE 1's loads, everything another launch reads or overwrites while this one
// still runs -- the output lines, the carried accumulator -- is stored written-through, and a FIR workgroup counts itself in
// behind its stores.  The last one tells the host how far the post-process has come (`progress`, pinned).
struct PostResident { uint32_t *postDone; uint32_t postDoneTarget; uint32_t *progress; uint32_t progressValue; uint32_t *error; uint32_t *counters; };
template<int MODE>
__device__ __forceinline__ void PostFusedBlock(float *xs, uint32_t wg, uint32_t tid, uint32_t outTarget, const float *__restrict__ in, uint32_t nch,
    const SplitterState *__restrict__ spIn, SplitterState *__restrict__ spOut, const float *__restrict__ hfscales,
    const float *__restrict__ chanCoeffs, uint32_t taps, const float *__restrict__ accIn, float *__restrict__ carryOut,
    float *__restrict__ left, float *__restrict__ right, uint32_t n, float *__restrict__ xf, uint32_t *__restrict__ arrived, uint32_t epoch,
    Tri3 runPower, float *__restrict__ hostOut, uint32_t *__restrict__ hostFlag, uint32_t hostSeq, uint32_t *__restrict__ outArrived,
    const uint32_t *__restrict__ reduced, uint32_t reducedEpoch, PostResident PR = PostResident{nullptr, 0u, nullptr, 0u, nullptr, nullptr})
{
    constexpr bool FUSED = MODE != 0, RESID = MODE == 2;
    const uint32_t lane = tid;
    if(wg < nch)
    {   // ---- BandSplitter::processHfScale of dry channel c (PostSplitKernel)
        const uint32_t c = wg;
        if constexpr (FUSED)
        {   // the dry lines are this launch's own reduction workgroups' sums
            if(!PostWaitCounter(reduced, reducedEpoch, 4, (RESID && wg == 0 && lane == 0) ? PR.counters + 16u * kRcWaitRedDone : nullptr) && PR.error && lane == 0) __hip_atomic_store(PR.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll 8
            for(uint32_t k = lane; k < uint32_t(kLine); k += 64) xs[k] = (k < n) ? PostLoadCoherent(in + size_t{c} * kLine + k) : 0.0f;
        }
        else
        {
#pragma unroll 8
            for(uint32_t k = lane; k < uint32_t(kLine); k += 64) xs[k] = (k < n) ? in[size_t{c} * kLine + k] : 0.0f;
        }
        WaveSync();
        SplitterState st = spIn[c];
        SplitterScanHfDpp(st, xs, n, hfscales[c], lane, runPower);
        if(lane == 0) spOut[c] = st;
        WaveSync();
#pragma unroll 8
        for(uint32_t k = lane; k < uint32_t(kLine); k += 64) PostStoreCoherent(xf + size_t{c} * kLine + k, (k < n) ? xs[k] : 0.0f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the channel is where the others will read it ...
        if(lane == 0) __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ... before it counts as arrived
        return;
    }
    // ---- the decoder FIR over 64 output frames -- lane = frame, both ears in the lane; the channels' windows AND their (left,
    // right) tap pairs come out of LDS: a window read per lane and a broadcast read of the pair per tap, two multiply-adds.  The
    // taps do not depend on the update, so the first two channels' are staged BEFORE the wait for the splits (until round 5 they
    // came through the scalar cache eight pairs at a time, each load waited for where it was used: 44 dependent scalar loads were
    // most of what lay between the splits' arrival and the workgroup's end -- and that interval is on the post chain's critical
    // path, which bounds the step).  Two channels at a time: windows and tap pairs of two fit the 4352 bytes the split's line
    // buffer has anyway -- with a second array for four channels' taps (8448 bytes, seven LDS granules) the kernel no longer
    // started beside a resident voice kernel and its reduction (round 5: the post-process of an update then was what everything
    // else waited for).  Then the shift (PostFirKernel + PostShiftKernel).
    const uint32_t blk = wg - nch;
    const uint32_t o = blk * uint32_t(kFusedFrames) + lane;
    const int32_t base = int32_t(blk) * kFusedFrames - kHrirLen;
    constexpr int kCoefDw = kHrirLen * 2;                        // one channel's staged tap pairs
    constexpr int kWin = kHrirLen + kFusedFrames;
    constexpr int kPair = 2;                                     // channels per pass
    static_assert(sizeof(float) * (kLine + 64) >= sizeof(float) * kPair * (kWin + kCoefDw), "two windows and two channels' taps fit into the split's line buffer");
    float (*xw4)[kWin] = reinterpret_cast<float (*)[kWin]>(xs);
    float *cst = xs + kPair * kWin;
    auto stageCoeffs = [&](uint32_t c0, uint32_t gc)
    {   // (the channels' tap pairs lie 128 pairs apart in memory as in LDS: four loads in flight off one base per channel -- left
        // to the compiler's unrolling the copy alone took the kernel from 26 to 48 registers)
        typedef const __attribute__((address_space(1))) float *gfloatp;
        gfloatp src = reinterpret_cast<gfloatp>((const __attribute__((address_space(1))) void*)(chanCoeffs + size_t{c0} * kCoefDw)) + lane;
        float *dst = cst + lane;
#pragma unroll 1
        for(uint32_t q = 0; q < gc; ++q)
        {
            const float a = src[0], b = src[64], c2 = src[128], d = src[192];
            dst[0] = a; dst[64] = b; dst[128] = c2; dst[192] = d;
            src += kCoefDw; dst += kCoefDw;
        }
    };
    stageCoeffs(0u, nch < uint32_t(kPair) ? nch : uint32_t(kPair));
    f2 accOld = {0.0f, 0.0f};
    if constexpr (!FUSED) accOld = reinterpret_cast<const f2*>(accIn)[o];           // (requested before the wait)
    if(!PostWaitCounter(arrived, epoch, 8, (RESID && wg == nch && lane == 0) ? PR.counters + 16u * kRcWaitSplit : nullptr) && PR.error && lane == 0) __hip_atomic_store(PR.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if constexpr (FUSED)      // (the channels have arrived, so the reduction they waited for is through: its accumulator columns)
        accOld = f2{PostLoadCoherent(accIn + 2u * o), PostLoadCoherent(accIn + 2u * o + 1u)};
    float accL = 0.0f, accR = 0.0f;
    for(uint32_t c0 = 0; c0 < nch; c0 += kPair)
    {
        const uint32_t gc = (nch - c0 < uint32_t(kPair)) ? nch - c0 : uint32_t(kPair);
        WaveSync();
        if(c0) stageCoeffs(c0, gc);
        for(uint32_t c = 0; c < gc; ++c)
        {
#pragma unroll
            for(int j3 = 0; j3 < kWin / 64; ++j3)
            {
                const uint32_t j = lane + 64u * uint32_t(j3);
                const int32_t fr = base + int32_t(j);
                xw4[c][j] = (fr >= 0 && fr < kLine) ? PostLoadCoherent(xf + size_t{c0 + c} * kLine + uint32_t(fr)) : 0.0f;
            }
        }
        WaveSync();
        for(uint32_t c = 0; c < gc; ++c)
        {
            const float *xw = &xw4[c][kHrirLen + lane];                  // x_c[o - t] = xw[-t]
            const f2 *cw = reinterpret_cast<const f2*>(cst + c * uint32_t(kCoefDw));
#pragma unroll 1
            for(uint32_t t4 = 0; t4 < taps; t4 += 4)
            {   // (four taps in flight: the kernel stays within the 32 registers that fit beside the voice kernels' wavefronts)
                float x[4]; f2 cc[4];
#pragma unroll
                for(int j = 0; j < 4; ++j) { x[j] = xw[-int32_t(t4) - j]; cc[j] = cw[t4 + uint32_t(j)]; }
#pragma unroll
                for(int j = 0; j < 4; ++j)
                {
                    accL = __builtin_fmaf(cc[j].x, x[j], accL);
                    accR = __builtin_fmaf(cc[j].y, x[j], accR);
                }
            }
        }
    }
    {   // PostShiftKernel's arithmetic: s = accumulator + channels' sum
        const f2 s = f2{accOld.x + accL, accOld.y + accR};
        if(o < uint32_t(kLine))
        {
            float l, r;
            if constexpr (FUSED) { l = PostLoadCoherent(left + o); r = PostLoadCoherent(right + o); }
            else { l = left[o]; r = right[o]; }
            if(o < n)
            {
                l = l + s.x; r = r + s.y;
                if constexpr (RESID) { PostStoreCoherent(left + o, l); PostStoreCoherent(right + o, r); }
                else { left[o] = l; right[o] = r; }
            }
            // the pipelined host boundary (oalgpu_read_output_async): the two output lines also go straight into the host's
            // pinned ring slot -- no copy launch behind this kernel, no runtime call on the host
            if(hostOut) { hostOut[o] = l; hostOut[uint32_t(kLine) + o] = r; }
        }
        // hrtfbase.h:127-132: frames [n, n + 128) move to the front, the following n frames are cleared, anything beyond stays
        f2 *carry2 = reinterpret_cast<f2*>(carryOut);
        auto storeCarry = [&](uint32_t at, f2 v)
        {
            if constexpr (RESID) { PostStoreCoherent(carryOut + 2u * at, v.x); PostStoreCoherent(carryOut + 2u * at + 1u, v.y); }
            else carry2[at] = v;
        };
        if(o >= n && o < n + uint32_t(kHrirLen)) storeCarry(o - n, s);
        if(o >= uint32_t(kHrirLen)) storeCarry(o, (o < uint32_t(kHrirLen) + n) ? f2{0.0f, 0.0f} : s);
    }
    if constexpr (RESID)
    {   // behind this workgroup's stores (acknowledged: they are written through): it counts as through; the last one of the
        // launch hands the ring slot's sequence number (behind everybody's lines) and the progress word to the host
        if(hostOut) __threadfence_system();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if(lane == 0)
        {
            const uint32_t t = __hip_atomic_fetch_add(PR.postDone, 1u, hostOut ? __ATOMIC_ACQ_REL : __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(t + 1u == PR.postDoneTarget)
            {
                if(hostOut)
                {
                    __threadfence_system();
                    __hip_atomic_store(hostFlag, hostSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                __hip_atomic_store(PR.progress, PR.progressValue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    else if(hostOut)
    {   // the slot's sequence number goes out behind the last FIR workgroup's lines (system scope: the reader is the host)
        // (outArrived only ever grows; `outTarget` is what it reads when this launch's FIR workgroups are all through)
        __threadfence_system();
        if(lane == 0)
        {
            const uint32_t t = __hip_atomic_fetch_add(outArrived, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if(t + 1u == outTarget)
            {
                __threadfence_system();
                __hip_atomic_store(hostFlag, hostSeq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(48))) PostFusedKernel(const float *__restrict__ in, uint32_t nch,
    const SplitterState *__restrict__ spIn, SplitterState *__restrict__ spOut, const float *__restrict__ hfscales,
    const float *__restrict__ chanCoeffs, uint32_t taps, const float *__restrict__ accIn, float *__restrict__ carryOut,
    float *__restrict__ left, float *__restrict__ right, uint32_t n, float *__restrict__ xf, uint32_t *__restrict__ arrived, uint32_t epoch,
    Tri3 runPower, float *__restrict__ hostOut, uint32_t *__restrict__ hostFlag, uint32_t hostSeq, uint32_t *__restrict__ outArrived, uint32_t outTarget)
{
    __shared__ float xs[kLine + 64];                                   // split: the channel; FIR: [kPostGroup][128 + 64] windows
    __builtin_amdgcn_s_setprio(3);                                     // (a short chain the step waits for, beside a voice kernel that fills the SIMDs)
    PostFusedBlock<0>(xs, blockIdx.x, threadIdx.x, outTarget, in, nch, spIn, spOut, hfscales, chanCoeffs, taps, accIn, carryOut, left, right, n,
        xf, arrived, epoch, runPower, hostOut, hostFlag, hostSeq, outArrived, nullptr, 0u);
}

// Reduction AND post-process of a context without effect slots and without a collective as ONE launch: workgroups
// [0, nReduce) are BusReduceKernel<4>'s (four wavefronts, 64 bus columns each; they store their sums written-through and
// count themselves in), the rest are PostFusedKernel's workgroups with three idle wavefronts that leave at once -- the
// splits wait for the reduction's counter, the FIR blocks for the splits'.  One dispatch gap and one launch less on the post
// stream than the two kernels; the same sums in the same order, so the same bits.
// (four partial sums of each run per step instead of BusReduceKernel<4>'s eight: beside the post-process's registers that is
// what keeps the launch within the 48 registers two voice wavefronts leave of a SIMD)
#ifndef OALGPU_FUSED_KF
#define OALGPU_FUSED_KF 4
#endif
struct ReducePostArgs {
    const float *carry; uint32_t nReduce;
    const float *in; uint32_t nch;
    const SplitterState *spIn; SplitterState *spOut; const float *hfscales, *chanCoeffs; uint32_t taps;
    const float *accIn; float *carryOut, *left, *right; uint32_t n; float *xf; uint32_t *arrived; uint32_t epoch; Tri3 runPower;
    float *hostOut; uint32_t *hostFlag; uint32_t hostSeq; uint32_t *outArrived; uint32_t outTarget;
    uint32_t *reduced; uint32_t reducedEpoch;
};
__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) ReducePostFusedKernel(DeviceLayout L, ReducePostArgs A)
{
    __shared__ float xs[kLine + 64];
    if(blockIdx.x < A.nReduce)
    {
        BusReduceBlock<4, true, OALGPU_FUSED_KF>(L, A.carry, reinterpret_cast<float (*)[64]>(xs), blockIdx.x, threadIdx.x);
        if(threadIdx.x < 64)
        {   // the sums are where the others will read them before this workgroup counts as through
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if(threadIdx.x == 0) __hip_atomic_fetch_add(A.reduced, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    if(threadIdx.x >= 64) return;
    PostFusedBlock<1>(xs, blockIdx.x - A.nReduce, threadIdx.x, A.outTarget, A.in, A.nch, A.spIn, A.spOut, A.hfscales, A.chanCoeffs, A.taps,
        A.accIn, A.carryOut, A.left, A.right, A.n, A.xf, A.arrived, A.epoch, A.runPower, A.hostOut, A.hostFlag, A.hostSeq, A.outArrived, A.reduced, A.reducedEpoch);
}

// ---- resident contexts (OALGPU_CTX_RESIDENT; the protocol: kernels.hpp ResidentDoor) ----
// The reduction of update u of a resident context: BusReduceKernel<4>'s sums in the same order, launched when the update is
// SUBMITTED, on the context's reduce stream.  It waits for the voice workgroups' arrival on the update's partial set (the voice
// kernel is one launch that never ends in between), reads the set with L2-coherent loads, tells the voice kernel that the set
// may be written again (kRcRedRead) and only then waits for the post-process of update u - 1 (its FIR workgroups have read the
// bus block and written the carried accumulator: kRcPostDone) -- so that only the carry's add and the stores, not the 4.7 MB of
// partial sums, lie between two post-processes.  Its sums are stored written-through; it counts itself in on kRcRedDone.
struct ResidentReduceArgs {
    uint32_t *counters, *hostFlags;
    uint32_t set, arriveTarget, postDoneTarget;
};
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(48))) BusReduceResidentKernel(DeviceLayout L, const float *__restrict__ carry,
    ResidentReduceArgs A)
{
    __shared__ float slice[4][64];
    const uint32_t tid = threadIdx.x;
    __builtin_amdgcn_s_setprio(3);              // (a short chain the resident voice wavefronts wait for: four of them share this SIMD, at priorities 2 .. 0)
    if(tid == 0)
    {
        if(!PostWaitCounter(A.counters + 16u * (kRcArrive0 + A.set), A.arriveTarget, 4, blockIdx.x == 0 ? A.counters + 16u * kRcWaitArrive : nullptr))
            __hip_atomic_store(A.hostFlags + 16u * kRhError, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    auto mid = [&]()
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // every partial sum this thread asked for has arrived
        __syncthreads();
        if(tid == 0)
        {
            __hip_atomic_fetch_add(A.counters + 16u * kRcRedRead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if(!PostWaitCounter(A.counters + 16u * kRcPostDone, A.postDoneTarget, 4, blockIdx.x == 0 ? A.counters + 16u * kRcWaitPost : nullptr))
                __hip_atomic_store(A.hostFlags + 16u * kRhError, 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
    };
    BusReduceBlock<4, true, 8, true>(L, carry, slice, blockIdx.x, tid, mid);
    if(tid < 64)
    {   // the sums are where the post-process will read them before this workgroup counts as through
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if(tid == 0) __hip_atomic_fetch_add(A.counters + 16u * kRcRedDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ void OALGPU_SINGLE_DS_OPS __launch_bounds__(64) __attribute__((amdgpu_num_vgpr(48))) PostResidentKernel(const float *__restrict__ in, uint32_t nch,
    const SplitterState *__restrict__ spIn, SplitterState *__restrict__ spOut, const float *__restrict__ hfscales,
    const float *__restrict__ chanCoeffs, uint32_t taps, const float *__restrict__ accIn, float *__restrict__ carryOut,
    float *__restrict__ left, float *__restrict__ right, uint32_t n, float *__restrict__ xf, uint32_t *__restrict__ arrived, uint32_t epoch,
    Tri3 runPower, float *__restrict__ hostOut, uint32_t *__restrict__ hostFlag, uint32_t hostSeq, const uint32_t *__restrict__ reduced,
    uint32_t reducedEpoch, PostResident PR)
{
    __shared__ float xs[kLine + 64];
    __builtin_amdgcn_s_setprio(3);
    PostFusedBlock<2>(xs, blockIdx.x, threadIdx.x, 0u, in, nch, spIn, spOut, hfscales, chanCoeffs, taps, accIn, carryOut, left, right, n,
        xf, arrived, epoch, runPower, hostOut, hostFlag, hostSeq, nullptr, reduced, reducedEpoch, PR);
}

} // namespace

// update u's reduction of a resident context (see BusReduceResidentKernel): L.partHrtf = the update's partial set; counters /
// hostFlags: the context's (kernels.hpp ResidentCounter / ResidentHostFlag); arriveTarget: what the set's arrival counter reads
// when every voice workgroup has stored update u's partial; postDoneTarget: what kRcPostDone reads when the post-process of
// update u - 1 is through
void LaunchBusReduceResident(hipStream_t s, const DeviceLayout &L, const float *carry, uint32_t *counters, uint32_t *hostFlags, uint32_t set,
    uint32_t arriveTarget, uint32_t postDoneTarget)
{
    const uint32_t total = uint32_t(BusFloats(L));
    hipLaunchKernelGGL(BusReduceResidentKernel, dim3((total + 63u) / 64u), dim3(256), 0, s, L, carry, ResidentReduceArgs{counters, hostFlags, set, arriveTarget, postDoneTarget});
}

// update u's HRTF post-process of a resident context (PostFusedBlock<2>); redDoneTarget: what kRcRedDone reads when update u's
// reduction is through; postDoneTarget: what kRcPostDone reads when this launch's FIR workgroups are; progressValue: what the
// last of them stores into the host's progress word
uint32_t PostResidentFirGroups() { return uint32_t(kPostFrames / kFusedFrames); }
void LaunchPostResident(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, const float *accIn, float *carryOut,
    const SplitterState *spIn, SplitterState *spOut, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n,
    float *xf, uint32_t *arrived, uint32_t epoch, const float runPower[4], hipEvent_t evDone, float *hostOut, uint32_t *hostFlag, uint32_t hostSeq,
    uint32_t *counters, uint32_t *hostFlags, uint32_t redDoneTarget, uint32_t postDoneTarget, uint32_t progressValue)
{
    const uint32_t taps = irsize <= 16u ? 16u : ((irsize + 15u) & ~15u);
    const Tri3 P{runPower[0], runPower[1], runPower[2], runPower[3]};
    const PostResident PR{counters + 16u * kRcPostDone, postDoneTarget, hostFlags + 16u * kRhProgress, progressValue, hostFlags + 16u * kRhError, counters};
    hipExtLaunchKernelGGL(PostResidentKernel, dim3(nch + kPostFrames / kFusedFrames), dim3(64), 0, s, nullptr, evDone, 0u, in, nch, spIn, spOut, hfscales,
        chanCoeffs, taps, accIn, carryOut, left, right, n, xf, arrived, epoch, P, hostOut, hostFlag, hostSeq, counters + 16u * kRcRedDone, redDoneTarget, PR);
}

// the whole FAST post-process in one launch; spIn / spOut: the two splitter-state buffers of the context (the caller swaps them),
// carryOut: the carried accumulator the next reduction adds (1152 x 2); xf: nch x 1024 floats of scratch; arrived / epoch: the
// context's channel counter and the value it reaches when this update's channels are all there (nch more than before);
// runPower: the splitter's transition over a run of ((n + 63) / 64) | 1 samples as the scan wants it (SplitterRunPowers, api.hip);
// hostOut (null: none): 2 x 1024 floats of pinned host memory that receive the two output lines, hostFlag: where hostSeq is
// stored when they are all there, outArrived / outTarget: the context's counter of FIR workgroups (it only ever grows) and what
// it reads when this launch's are all through
void LaunchPostDirectHrtfFused(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, const float *accIn, float *carryOut,
    const SplitterState *spIn, SplitterState *spOut, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n,
    float *xf, uint32_t *arrived, uint32_t epoch, const float runPower[4], hipEvent_t evDone,
    float *hostOut, uint32_t *hostFlag, uint32_t hostSeq, uint32_t *outArrived, uint32_t outTarget)
{
    const uint32_t taps = irsize <= 16u ? 16u : ((irsize + 15u) & ~15u);
    const Tri3 P{runPower[0], runPower[1], runPower[2], runPower[3]};
    static_assert(kPostFrames % kFusedFrames == 0, "whole workgroups");
    hipExtLaunchKernelGGL(PostFusedKernel, dim3(nch + kPostFrames / kFusedFrames), dim3(64), 0, s, nullptr, evDone, 0u, in, nch, spIn, spOut, hfscales,
        chanCoeffs, taps, accIn, carryOut, left, right, n, xf, arrived, epoch, P, hostOut, hostFlag, hostSeq, outArrived, outTarget);
}

// The reduction of update k and its HRTF post-process in one launch (see ReducePostFusedKernel); reduced / reducedEpoch: the
// context's counter of reduction workgroups and what it reads when this update's are all through
void LaunchReducePostFused(hipStream_t s, const DeviceLayout &L, const float *carry, float *left, float *right, const float *in, uint32_t nch,
    const float *accIn, float *carryOut, const SplitterState *spIn, SplitterState *spOut, const float *hfscales, const float *chanCoeffs,
    uint32_t irsize, uint32_t n, float *xf, uint32_t *arrived, uint32_t epoch, const float runPower[4], hipEvent_t evDone,
    float *hostOut, uint32_t *hostFlag, uint32_t hostSeq, uint32_t *outArrived, uint32_t outTarget, uint32_t *reduced, uint32_t reducedEpoch)
{
    ReducePostArgs A{};
    A.carry = carry; A.nReduce = ReducePostReduceGroups(L);
    A.in = in; A.nch = nch; A.spIn = spIn; A.spOut = spOut; A.hfscales = hfscales; A.chanCoeffs = chanCoeffs;
    A.taps = irsize <= 16u ? 16u : ((irsize + 15u) & ~15u);
    A.accIn = accIn; A.carryOut = carryOut; A.left = left; A.right = right; A.n = n; A.xf = xf; A.arrived = arrived; A.epoch = epoch;
    A.runPower = Tri3{runPower[0], runPower[1], runPower[2], runPower[3]};
    A.hostOut = hostOut; A.hostFlag = hostFlag; A.hostSeq = hostSeq; A.outArrived = outArrived; A.outTarget = outTarget; A.reduced = reduced; A.reducedEpoch = reducedEpoch;
    hipExtLaunchKernelGGL(ReducePostFusedKernel, dim3(A.nReduce + nch + kPostFrames / kFusedFrames), dim3(256), 0, s, nullptr, evDone, 0u, L, A);
}

// temp: nch x 1024 filtered channels, then 1152 x 2 channel sums
// evDone: an event bound to the completion of the last of the three dispatches (null: none)
void LaunchPostDirectHrtfFast(hipStream_t s, float *left, float *right, const float *in, uint32_t nch, float *accum,
    SplitterState *splitters, const float *hfscales, const float *chanCoeffs, uint32_t irsize, uint32_t n, float *temp, hipEvent_t evDone)
{
    float *xf = temp, *tmp = temp + size_t{nch} * kLine;
    const uint32_t taps = irsize <= 16u ? 16u : ((irsize + 15u) & ~15u);
    hipLaunchKernelGGL(PostSplitKernel, dim3(nch), dim3(64), 0, s, in, xf, splitters, hfscales, n);
    // (16 frames per workgroup: 47.5 us per step against 49.0 with 8 and 50.5 with round 2's single workgroup, one box)
    hipLaunchKernelGGL(PostFirKernel<16>, dim3(kPostFrames / 16), dim3(64), 0, s, xf, nch, chanCoeffs, taps, tmp);
    hipExtLaunchKernelGGL(PostShiftKernel, dim3(1), dim3(256), 0, s, nullptr, evDone, 0u, left, right, accum, tmp, n);
}

} // namespace oalgpu
