// Device functions shared by the two batched voice kernels (voice_kernel.hip: one workgroup per
// voice group, every resampler/format/send combination, EXACT and FAST; voice_wave.hip: one
// wavefront per voice, the FAST HRTF hot path): sample decode, static-buffer gather, the
// CalculateBufferSize integer logic and saturating position arithmetic of core/voice.cpp.
#pragma once
#include "dev_math.hpp"
#include "../../include/oalgpu.h"

#pragma clang fp contract(off)

namespace oalgpu {

struct alignas(16) BufferItem {     // VoiceBufferItem, core/voice.h:84-98
    const void *data;
    int32_t fmt;
    uint32_t frameStep, sampleLen, loopStart, loopEnd;
    int32_t next;                   // mNext: the buffer queued behind this one (table index + 1; 0 = none)
};

// ---- SampleInfo<T>::to_float, core/fmt_traits.h:91-139 (+ the mu-law/A-law tables :12-80 in
// closed form) ----
__device__ __forceinline__ float DecodeMulaw(uint32_t u)
{
    u = ~u & 0xffu;
    const int exponent = (u >> 4) & 7, mantissa = u & 15;
    const int seg = ((33 << exponent) - 33) << 2;             // 0,132,396,924,...
    const int mag = seg + (mantissa << (exponent + 3));
    return float((u & 0x80u) ? -mag : mag) * (1.0f / 32768.0f);
}
__device__ __forceinline__ float DecodeAlaw(uint32_t a)
{
    a = (a ^ 0x55u) & 0xffu;
    const int exponent = (a >> 4) & 7, mantissa = a & 15;
    const int mag = (exponent == 0) ? ((mantissa << 4) + 8) : (((mantissa << 4) + 0x108) << (exponent - 1));
    return float((a & 0x80u) ? mag : -mag) * (1.0f / 32768.0f);
}

template<int FMT>
__device__ __forceinline__ float LoadSample(const void *data, size_t idx)
{
    if constexpr(FMT == OALGPU_FMT_UBYTE) return (float(static_cast<const uint8_t*>(data)[idx]) - 128.0f) * (1.0f / 128.0f);
    else if constexpr(FMT == OALGPU_FMT_SHORT) return float(static_cast<const int16_t*>(data)[idx]) * (1.0f / 32768.0f);
    else if constexpr(FMT == OALGPU_FMT_INT) return float(static_cast<const int32_t*>(data)[idx]) * (1.0f / 2147483648.0f);
    else if constexpr(FMT == OALGPU_FMT_FLOAT) return static_cast<const float*>(data)[idx];
    else if constexpr(FMT == OALGPU_FMT_DOUBLE) return float(static_cast<const double*>(data)[idx]);
    else if constexpr(FMT == OALGPU_FMT_MULAW) return DecodeMulaw(static_cast<const uint8_t*>(data)[idx]);
    else return DecodeAlaw(static_cast<const uint8_t*>(data)[idx]);
}

// LoadBufferStatic, core/voice.cpp:500-544: element k of the `count` source samples starting
// at buffer position dataPos (loop wrap by modulo; past-the-end holds the last sample).
template<int FMT, int NT>
__device__ __forceinline__ void FillFromStatic(float *dst, uint32_t count, const BufferItem &b, bool looping,
    uint32_t dataPos, uint32_t tid)
{
    const uint32_t fs = b.frameStep;
    if(!looping)
    {
        const bool any = b.sampleLen > dataPos;
        const uint32_t avail = any ? b.sampleLen - dataPos : 0u;
        const float last = any ? LoadSample<FMT>(b.data, size_t{b.sampleLen - 1u} * fs) : 0.0f;
        for(uint32_t k = tid; k < count; k += NT)
            dst[k] = (k < avail) ? LoadSample<FMT>(b.data, size_t{dataPos + k} * fs) : last;
    }
    else
    {
        const uint32_t ls = b.loopStart, le = b.loopEnd, size = le - ls;
        const uint32_t intPos = (dataPos < le) ? dataPos : ((dataPos - ls) % size) + ls;
        const uint32_t first = le - intPos;
        for(uint32_t k = tid; k < count; k += NT)
        {
            const uint32_t idx = (k < first) ? intPos + k : ls + ((k - first) % size);
            dst[k] = LoadSample<FMT>(b.data, size_t{idx} * fs);
        }
    }
}

template<int NT>
__device__ __forceinline__ void FillFromBuffer(float *dst, uint32_t count, const BufferItem &b, bool looping,
    uint32_t dataPos, uint32_t tid)
{
    switch(b.fmt)
    {
    case OALGPU_FMT_UBYTE: FillFromStatic<OALGPU_FMT_UBYTE, NT>(dst, count, b, looping, dataPos, tid); break;
    case OALGPU_FMT_SHORT: FillFromStatic<OALGPU_FMT_SHORT, NT>(dst, count, b, looping, dataPos, tid); break;
    case OALGPU_FMT_INT: FillFromStatic<OALGPU_FMT_INT, NT>(dst, count, b, looping, dataPos, tid); break;
    case OALGPU_FMT_FLOAT: FillFromStatic<OALGPU_FMT_FLOAT, NT>(dst, count, b, looping, dataPos, tid); break;
    case OALGPU_FMT_DOUBLE: FillFromStatic<OALGPU_FMT_DOUBLE, NT>(dst, count, b, looping, dataPos, tid); break;
    case OALGPU_FMT_MULAW: FillFromStatic<OALGPU_FMT_MULAW, NT>(dst, count, b, looping, dataPos, tid); break;
    default: FillFromStatic<OALGPU_FMT_ALAW, NT>(dst, count, b, looping, dataPos, tid); break;
    }
}

__device__ __forceinline__ float LoadSampleAny(int fmt, const void *data, size_t idx)
{
    switch(fmt)
    {
    case OALGPU_FMT_UBYTE: return LoadSample<OALGPU_FMT_UBYTE>(data, idx);
    case OALGPU_FMT_SHORT: return LoadSample<OALGPU_FMT_SHORT>(data, idx);
    case OALGPU_FMT_INT: return LoadSample<OALGPU_FMT_INT>(data, idx);
    case OALGPU_FMT_FLOAT: return LoadSample<OALGPU_FMT_FLOAT>(data, idx);
    case OALGPU_FMT_DOUBLE: return LoadSample<OALGPU_FMT_DOUBLE>(data, idx);
    case OALGPU_FMT_MULAW: return LoadSample<OALGPU_FMT_MULAW>(data, idx);
    default: return LoadSample<OALGPU_FMT_ALAW>(data, idx);
    }
}

// LoadBufferQueue, core/voice.cpp:563-594: `count` source samples from position dataPos of queue item
// `item`, crawling the queue (mNext, then the loop item) and holding the last sample when it ends.
// Every thread walks the queue itself (uniform control flow); `sync` orders the threads' LDS accesses.
template<int NT, typename Sync>
__device__ __forceinline__ void FillFromQueue(float *dst, uint32_t count, const BufferItem *buffers, int32_t item, int32_t loopItem,
    uint32_t dataPos, uint32_t tid, Sync sync)
{
    uint32_t done = 0;
    for(uint32_t guard = 0; item >= 0 && done < count && guard < 4096u; ++guard)
    {
        const BufferItem b = buffers[item];
        const int32_t nextItem = b.next > 0 ? b.next - 1 : loopItem;
        if(dataPos >= b.sampleLen) { dataPos -= b.sampleLen; item = nextItem; continue; }
        const uint32_t rem = (count - done < b.sampleLen - dataPos) ? count - done : b.sampleLen - dataPos;
        for(uint32_t k = tid; k < rem; k += NT) dst[done + k] = LoadSampleAny(b.fmt, b.data, size_t{dataPos + k} * b.frameStep);
        done += rem;
        dataPos = 0;
        item = nextItem;
    }
    if(done < count)
    {
        sync();
        const float last = done ? dst[done - 1u] : 0.0f;
        sync();
        for(uint32_t k = done + tid; k < count; k += NT) dst[k] = last;
    }
}

// Voice::mix's "streaming source" position update (core/voice.cpp:1182-1194): buffers the position ran
// past are left behind; `item` < 0 afterwards = the queue ended.
__device__ __forceinline__ void AdvanceQueue(const BufferItem *buffers, int32_t &item, int32_t loopItem, int32_t &pos, uint32_t &buffersDone)
{
    for(uint32_t guard = 0; item >= 0 && guard < 65536u; ++guard)
    {
        const BufferItem b = buffers[item];
        if(b.sampleLen > uint32_t(pos)) break;
        pos -= int32_t(b.sampleLen);
        ++buffersDone;
        item = b.next > 0 ? b.next - 1 : loopItem;
    }
}

// CalculateBufferSize, core/voice.cpp:600-640 (integer, bit-exact).
__device__ __forceinline__ void CalcBufferSize(uint32_t fracPos, uint32_t increment, uint32_t dstRemaining,
    uint32_t &dst, uint32_t &src)
{
    constexpr uint32_t srcMax = kResampleDataSize - kMaxEdge;
    const uint32_t ext = increment <= kFracOne ? 1u : 0u;
    const uint64_t srcSize = ((uint64_t{dstRemaining - ext} * increment + fracPos) >> kFracBits) + ext + kMaxEdge;
    if(srcSize <= srcMax) { dst = dstRemaining; src = uint32_t(srcSize); return; }
    const uint64_t dstSize = ((uint64_t{srcMax - kMaxEdge} << kFracBits) - fracPos) / increment;
    if(dstSize < dstRemaining) { dst = uint32_t(dstSize) & ~3u; src = srcMax; return; }
    dst = dstRemaining; src = srcMax;
}

__device__ __forceinline__ int32_t AddSat(int32_t a, int32_t b)
{
    const int64_t r = int64_t{a} + b;
    return r > 2147483647ll ? 2147483647 : (r < -2147483648ll ? int32_t(-2147483647 - 1) : int32_t(r));
}

} // namespace oalgpu
