// BusReduceKernel<4>'s work (voice_kernel.hip) for ONE block of 64 bus columns, restated for the
// launch that fuses the reduction with the HRTF post-process (post_wave.hip): the same sums in the same order.  (The kernel of
// that name keeps its own copy: moved into a shared function its register allocation came out four registers higher, past the
// 48 that fit beside two voice wavefronts.)
#pragma once
#include "kernels.hpp"

namespace oalgpu {

constexpr int kReduceSegs = 16;

// COHERENT: the sums are read by other workgroups of the SAME launch (device-scope relaxed atomic store: written through)
template<bool COHERENT>
__device__ __forceinline__ void ReduceStore(float *p, float v)
{
    if constexpr (COHERENT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// COHLOAD: the partial sums (and the carry) were written by launches that are still RUNNING (the resident voice kernel, the
// post-process of the update before): read them with L2-coherent loads (device-scope relaxed atomics), whatever an L2 holds of
// those lines from the last time they were read is not to be trusted
template<bool COHLOAD, class P>
__device__ __forceinline__ float ReduceLoad(P p)
{
    if constexpr (COHLOAD) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
struct ReduceNoHook { __device__ __forceinline__ void operator()() const { } };

// slice: kReduceWaves == 4 ? [4][64] : [kReduceSegs][64] floats of LDS; block: which 64 columns; tid: thread of the workgroup
// (kReduceWaves x 64 of them, all of which must call this -- it synchronises the workgroup)
// KF: partial sums of each of a wavefront's four runs requested per step (the post-stream shape: 4 x KF loads in flight)
// mid (the four-wavefront shape): called by every thread between the runs' sums and the carry's load -- the partial buses have
// been read, the bus block has not been touched
template<int kReduceWaves, bool COHERENT, int KF = 8, bool COHLOAD = false, class MID = ReduceNoHook>
__device__ __forceinline__ void BusReduceBlock(const DeviceLayout &L, const float *__restrict__ carry, float (*slice)[64], uint32_t block, uint32_t tid,
    MID mid = MID{})
{
    // (the post-stream shape hands its runs over four at a time: 1 KB, ONE allocation granule of LDS -- beside two voice
    // workgroups of the dry-line kernels a CU has four granules to spare, and a launch of more reduction workgroups than
    // CUs (config 4: 336) must not keep a voice workgroup of the next update waiting for LDS)
    const uint32_t wave0 = tid >> 6, lane = tid & 63;
    const uint32_t idx = block * 64u + lane;
    const uint32_t dryLines = L.numDry + L.numReal;
    const uint32_t wetLines = L.numSlots * L.wetChannels;
    const uint32_t lineFloats = (dryLines + wetLines) * kLine;
    const uint32_t total = lineFloats + (kLine + kHrirLen) * 2;

    // 32-bit element offsets off one uniform base per source (the partial buses are far below
    // 2^32 floats): one address register per load keeps the wave inside its 32 VGPRs
    const bool fromLines = block * 64u < lineFloats;          // lineFloats is a multiple of 1024: uniform per workgroup
    bool have = false;
    uint32_t off = 0, stride = 0;
    if(fromLines)
    {
        const uint32_t line = idx / kLine, p = idx % kLine;
        int32_t from = -1;
        if(line < dryLines) { if(!L.hrtf && line < L.numDry) from = int32_t(line); }
        else from = int32_t((L.hrtf ? 0u : L.numDry) + (line - dryLines));
        if(from >= 0) { have = true; off = uint32_t(from) * kLine + p; stride = L.mixLines * kLine; }
    }
    else if(L.hrtf && idx < total)
    {
        have = true;
        off = idx - lineFloats;
        stride = uint32_t(kLine + kHrirLen) * 2u;
    }
    // uniform per wavefront (64 | lineFloats) and kept in SGPRs as a GLOBAL pointer, so that a
    // load's address is one 32-bit VGPR offset
    typedef const __attribute__((address_space(1))) float *gfloatp;
    gfloatp base;
    {
        const uint64_t b = reinterpret_cast<uint64_t>(fromLines ? L.partLines : L.partHrtf);
        const uint32_t lo = __builtin_amdgcn_readfirstlane(uint32_t(b)), hi = __builtin_amdgcn_readfirstlane(uint32_t(b >> 32));
        base = reinterpret_cast<gfloatp>((uint64_t{hi} << 32) | lo);
    }

    const uint32_t ngroups = fromLines ? L.numLineGroups : L.numGroups;
    const uint32_t per = (ngroups + kReduceSegs - 1) / kReduceSegs;
    if constexpr (kReduceWaves == 4)
    {   // The post-stream shape: a wavefront's four runs advance TOGETHER, eight groups of each per step -- 32 loads in flight
        // instead of 8, a quarter of the dependent round trips (the kernel is nothing but their latency: 16 of them per
        // wavefront, 8.4 us, before).  Every run is still summed in group order, so the result does not change by a bit.
        typedef const __attribute__((address_space(1))) char *gcharp;
        const uint32_t sb = stride * 4u;
        uint32_t gA[4], gE[4], oo[4];
        float sum[4];
        uint32_t common = per;
#pragma unroll
        for(int q = 0; q < 4; ++q)
        {
            const uint32_t seg = wave0 + 4u * uint32_t(q);
            gA[q] = seg * per < ngroups ? seg * per : ngroups;
            gE[q] = (gA[q] + per < ngroups) ? gA[q] + per : ngroups;
            oo[q] = (off + gA[q] * stride) * 4u;
            sum[q] = 0.0f;
            common = (gE[q] - gA[q]) < common ? gE[q] - gA[q] : common;
        }
        if(have)
        {
#pragma unroll 1
            for(uint32_t j = 0; j + uint32_t(KF) <= common; j += uint32_t(KF))
            {
                float v[4][KF];
#pragma unroll
                for(int q = 0; q < 4; ++q)
#pragma unroll
                    for(int k = 0; k < KF; ++k) v[q][k] = ReduceLoad<COHLOAD>(reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + (oo[q] + uint32_t(k) * sb)));
#pragma unroll
                for(int q = 0; q < 4; ++q)
                {
#pragma unroll
                    for(int k = 0; k < KF; ++k) sum[q] = sum[q] + v[q][k];
                    oo[q] += uint32_t(KF) * sb; gA[q] += uint32_t(KF);
                }
            }
#pragma unroll
            for(int q = 0; q < 4; ++q)
                for(; gA[q] < gE[q]; ++gA[q]) { sum[q] = sum[q] + ReduceLoad<COHLOAD>(reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + oo[q])); oo[q] += sb; }
        }
        mid();
        // segments in order 0..15 (segment = wavefront + 4 q), exactly as the 16-wavefront shape sums them
        float t = (carry && idx >= lineFloats && idx < total) ? ReduceLoad<COHLOAD>(carry + (idx - lineFloats)) : 0.0f;
#pragma unroll
        for(int q = 0; q < 4; ++q)
        {
            slice[wave0][lane] = sum[q];
            __syncthreads();
            if(wave0 == 0) { t = t + slice[0][lane]; t = t + slice[1][lane]; t = t + slice[2][lane]; t = t + slice[3][lane]; }
            __syncthreads();
        }
        if(wave0 == 0 && idx < total && (idx < lineFloats || L.hrtf)) ReduceStore<COHERENT>(&L.bus[idx], t);
        return;
    }
    else
#pragma unroll 1
    for(uint32_t seg = wave0; seg < uint32_t(kReduceSegs); seg += kReduceWaves)
    {
        const uint32_t g0 = seg * per < ngroups ? seg * per : ngroups;
        const uint32_t g1 = (g0 + per < ngroups) ? g0 + per : ngroups;
        float sum = 0.0f;
        if(have)
        {
            // byte offsets (the partial buses are far below 4 GB)
            typedef const __attribute__((address_space(1))) char *gcharp;
            const uint32_t sb = stride * 4u;
            uint32_t g = g0, o = (off + g0 * stride) * 4u;
#pragma unroll 1
            for(; g + 8 <= g1; g += 8)
            {
                float v[8];
#pragma unroll
                for(int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + (o + uint32_t(k) * sb));
#pragma unroll
                for(int k = 0; k < 8; ++k) sum = sum + v[k];
                o += 8u * sb;
            }
            for(; g < g1; ++g) { sum = sum + *reinterpret_cast<gfloatp>(reinterpret_cast<gcharp>(base) + o); o += sb; }
        }
        slice[seg][lane] = sum;
    }
    __syncthreads();
    if(wave0 == 0 && idx < total)
    {
        float t = (carry && idx >= lineFloats) ? carry[idx - lineFloats] : 0.0f;
        t = t + slice[0][lane];
#pragma unroll 5
        for(int w = 1; w < kReduceSegs; ++w) t = t + slice[w][lane];
        if(idx < lineFloats || L.hrtf) ReduceStore<COHERENT>(&L.bus[idx], t);
    }
}

} // namespace oalgpu
