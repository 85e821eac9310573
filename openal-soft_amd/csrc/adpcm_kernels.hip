// IMA4 and MS ADPCM source buffers: LoadSamples<IMA4Data> / LoadSamples<MSADPCMData>,
// core/voice.cpp:288-484.
//
// The reference decodes the compressed blocks on the fly in every mix.  A static buffer never changes, the
// blocks are independent of each other and HBM is plentiful: here a buffer is decoded ONCE when it is
// registered -- one thread per block and channel, the reference's integer arithmetic step for step -- into
// 16-bit PCM, which the voice kernels then read like any AL_FORMAT_*16 buffer.  Every decoded value is an
// integer in [-32768, 32767] that the reference turns into sample / 32768.0f; the kernels' int16 load does
// the same, so the samples a voice sees are bit-identical to the reference's.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oalgpu {
namespace {

// IMA ADPCM step sizes (IMA Digital Audio Focus and Technical Working Groups, "Recommended Practices for
// Enhancing Digital Audio Compatibility in Multimedia Systems", rev. 3.00, 1992); core/voice.cpp:199-209
__constant__ int32_t kImaStep[89] = {
    7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107, 118, 130,
    143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876, 963, 1060, 1166,
    1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871, 5358, 5894, 6484, 7132, 7845,
    8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385, 24623, 27086, 29794, 32767};
// MS ADPCM adaption table and predictor coefficients (Microsoft "New Multimedia Data Types and Data Techniques",
// 1994); core/voice.cpp:224-241
__constant__ int32_t kMsAdaption[16] = {230, 230, 230, 230, 307, 409, 512, 614, 768, 614, 512, 409, 307, 230, 230, 230};
__constant__ int32_t kMsCoeff[7][2] = {{256, 0}, {512, -256}, {0, 0}, {192, 64}, {240, 0}, {460, -208}, {392, -232}};

__device__ __forceinline__ int32_t Clamp16(int32_t v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }

// one thread = one block of one channel; dst[(block * samplesPerBlock + i) * channels + chan]
__global__ void __launch_bounds__(64) DecodeIma4Kernel(const uint8_t *src, int16_t *dst, uint32_t numBlocks, uint32_t samplesPerBlock,
    uint32_t channels, uint32_t sampleLen)
{
    const uint32_t id = blockIdx.x * 64u + threadIdx.x;
    if(id >= numBlocks * channels) return;
    const uint32_t block = id / channels, chan = id % channels;
    const uint32_t blockBytes = ((samplesPerBlock - 1u) / 2u + 4u) * channels;
    const uint8_t *b = src + size_t{block} * blockBytes;
    // voice.cpp:314-323: a signed 16-bit sample and a clamped table index start the block
    int32_t sample = int16_t(uint16_t(b[chan * 4 + 0]) | (uint16_t(b[chan * 4 + 1]) << 8));
    int32_t index = int16_t(uint16_t(b[chan * 4 + 2]) | (uint16_t(b[chan * 4 + 3]) << 8));
    index = index < 0 ? 0 : (index > 88 ? 88 : index);
    const uint8_t *nibbles = b + size_t{channels + chan} * 4;
    const uint32_t base = block * samplesPerBlock;
    if(base < sampleLen) dst[size_t{base} * channels + chan] = int16_t(sample);
    for(uint32_t n = 0; n + 1u < samplesPerBlock; ++n)
    {   // voice.cpp:333-350: 8 nibbles per channel in 4 bytes, the channels' words interleaved
        const uint32_t byteShift = (n & 1u) * 4u;
        const uint32_t wordOffset = (n >> 1) & ~3u;
        const uint32_t byteOffset = wordOffset * channels + ((n >> 1) & 3u);
        const uint32_t code = (nibbles[byteOffset] >> byteShift) & 15u;
        const int32_t codeword = int32_t(2u * (code & 7u) + 1u) * ((code & 8u) ? -1 : 1);
        sample = Clamp16(sample + codeword * kImaStep[index] / 8);
        const int32_t adjust = (code & 4u) ? int32_t(2u * (code & 3u) + 2u) : -1;
        index += adjust;
        index = index < 0 ? 0 : (index > 88 ? 88 : index);
        if(base + n + 1u < sampleLen) dst[size_t{base + n + 1u} * channels + chan] = int16_t(sample);
    }
}

__global__ void __launch_bounds__(64) DecodeMsAdpcmKernel(const uint8_t *src, int16_t *dst, uint32_t numBlocks, uint32_t samplesPerBlock,
    uint32_t channels, uint32_t sampleLen)
{
    const uint32_t id = blockIdx.x * 64u + threadIdx.x;
    if(id >= numBlocks * channels) return;
    const uint32_t block = id / channels, chan = id % channels;
    const uint32_t blockBytes = ((samplesPerBlock - 2u) / 2u + 7u) * channels;
    const uint8_t *b = src + size_t{block} * blockBytes;
    // voice.cpp:393-412: predictor index, scale, two history samples (the second one is the older)
    uint32_t pred = b[chan];
    pred = pred > 6u ? 6u : pred;
    int32_t scale = int16_t(uint16_t(b[channels + 2 * chan]) | (uint16_t(b[channels + 2 * chan + 1]) << 8));
    int32_t h0 = int16_t(uint16_t(b[3 * channels + 2 * chan]) | (uint16_t(b[3 * channels + 2 * chan + 1]) << 8));
    int32_t h1 = int16_t(uint16_t(b[5 * channels + 2 * chan]) | (uint16_t(b[5 * channels + 2 * chan + 1]) << 8));
    const uint8_t *nibbles = b + size_t{7} * channels;
    const int32_t c0 = kMsCoeff[pred][0], c1 = kMsCoeff[pred][1];
    const uint32_t base = block * samplesPerBlock;
    if(base < sampleLen) dst[size_t{base} * channels + chan] = int16_t(h1);
    if(base + 1u < sampleLen) dst[size_t{base + 1u} * channels + chan] = int16_t(h0);
    uint32_t nibbleOffset = chan;
    for(uint32_t n = 2; n < samplesPerBlock; ++n, nibbleOffset += channels)
    {   // voice.cpp:439-458
        const uint32_t byteOffset = nibbleOffset >> 1;
        const uint32_t byteShift = ((nibbleOffset & 1u) ^ 1u) * 4u;
        const uint32_t nval = (nibbles[byteOffset] >> byteShift) & 15u;
        const int32_t p = ((int32_t(nval) ^ 0x08) - 0x08) * scale;
        const int32_t diff = (h0 * c0 + h1 * c1) / 256;
        const int32_t sample = Clamp16(p + diff);
        h1 = h0; h0 = sample;
        scale = kMsAdaption[nval] * scale / 256;
        scale = scale < 16 ? 16 : scale;
        if(base + n < sampleLen) dst[size_t{base + n} * channels + chan] = int16_t(sample);
    }
}

} // namespace

void LaunchDecodeAdpcm(hipStream_t s, bool msadpcm, const uint8_t *src, int16_t *dst, uint32_t numBlocks, uint32_t samplesPerBlock,
    uint32_t channels, uint32_t sampleLen)
{
    const dim3 grid((numBlocks * channels + 63u) / 64u), block(64);
    if(msadpcm) hipLaunchKernelGGL(DecodeMsAdpcmKernel, grid, block, 0, s, src, dst, numBlocks, samplesPerBlock, channels, sampleLen);
    else hipLaunchKernelGGL(DecodeIma4Kernel, grid, block, 0, s, src, dst, numBlocks, samplesPerBlock, channels, sampleLen);
}

} // namespace oalgpu
