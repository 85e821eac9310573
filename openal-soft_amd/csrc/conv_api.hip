// C-ABI of the convolution reverb (include/oalgpu.h): host-side setup that the reference does
// once per impulse response in ConvolutionState::deviceUpdate (alc/effects/convolution.cpp:
// 318-471) -- segmenting, the double-precision forward FFT of every segment, the 1/256
// pre-scale -- and the per-update launch sequence.
#include "api_util.hpp"
#include "kernels.hpp"
#include "../host/polyphase.hpp"
#include "../host/params.hpp"

#include <cmath>
#include <complex>
#include <memory>
#include <vector>

using namespace oalgpu;

struct oalgpu_convolution {
    int device{0};
    uint32_t nlines{0}, numSegs{0}, ringSlots{0}, numChunks{0};
    uint32_t channels{1};                      // mChans.size()
    bool upsample{false};                      // mMix == UpsampleMix
    DevBuf<float> chanOut, hfScale, lfScale;
    DevBuf<SplitterState> split;
    uint32_t fifoPos{0}, curSeg{0};            // mFifoPos, mCurrentSegment
    DevBuf<float> xhist, ring, filt, fir, outFifo, partial, cur, tgt, tw128, tw256;
    DevBuf<float> hostIn, hostOut;             // staging for the host-buffer entry point
    DevBuf<uint32_t> ticket;                   // ConvFusedKernel's delivery counter
    DevBuf<float> firOut;                      // ... and its time-domain FIR slices
};

namespace {

constexpr uint32_t kSegLen = 128, kFftLen = 256, kSegsPerChunk = 16, kMaxBlocksPerUpdate = 8;

// radix-2 complex FFT in double (setup only)
void FftDouble(std::vector<std::complex<double>> &a)
{
    const size_t n = a.size();
    for(size_t i = 1, j = 0; i < n; ++i)
    {
        size_t bit = n >> 1;
        for(; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if(i < j) std::swap(a[i], a[j]);
    }
    for(size_t len = 2; len <= n; len <<= 1)
    {
        const double ang = -2.0 * 3.14159265358979323846 / double(len);
        for(size_t i = 0; i < n; i += len)
            for(size_t k = 0; k < len / 2; ++k)
            {
                const std::complex<double> w{std::cos(ang * double(k)), std::sin(ang * double(k))};
                const auto u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

// (PPhaseResampler: host/polyphase.hpp)

} // namespace

extern "C" {

/* PPhaseResampler::init + process on host memory: n_out = ceil(n_in * dst_rate / src_rate) is what
 * ConvolutionState::deviceUpdate asks for (convolution.cpp:360-362).  Pure host code. */
int oalgpu_polyphase_resample(uint32_t src_rate, uint32_t dst_rate, const double *in, size_t n_in, double *out, size_t n_out)
{
    if(!in || !out || src_rate == 0 || dst_rate == 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_polyphase_resample: bad arguments");
    Polyphase pp;
    pp.init(src_rate, dst_rate);
    pp.process(in, n_in, out, n_out);
    return OALGPU_OK;
}

int oalgpu_convolution_create(int device, uint32_t num_out_lines, const float *ir, uint32_t ir_len,
    oalgpu_convolution **out)
{ return oalgpu_convolution_create_ex(device, num_out_lines, ir, ir_len, 1, 0, 0, out); }

int oalgpu_convolution_create_ex(int device, uint32_t num_out_lines, const float *ir_in, uint32_t ir_frames, uint32_t channels,
    uint32_t ir_rate, uint32_t device_rate, oalgpu_convolution **out)
{
    if(!ir_in || !out || ir_frames < 1 || ir_frames > (1u << 22) || num_out_lines < 1 || num_out_lines > OALGPU_MAX_OUTPUT_CHANNELS
        || channels < 1 || channels > 8 || (ir_rate == 0) != (device_rate == 0))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_convolution_create: bad arguments");
    *out = nullptr;
    if(int rc = UseDevice(device)) return rc;
    auto c = std::make_unique<oalgpu_convolution>();
    c->device = device;
    c->nlines = num_out_lines;
    c->channels = channels;
    // the response at the device's rate, planar: resp[ch][ir_len] (:360-362, :412-422)
    uint32_t ir_len = ir_frames;
    std::vector<float> resp;
    if(ir_rate != device_rate)
    {
        ir_len = uint32_t((uint64_t{ir_frames} * device_rate + (ir_rate - 1)) / ir_rate);
        if(ir_len < 1 || ir_len > (1u << 22)) return Fail(OALGPU_ERR_INVALID, "oalgpu_convolution_create: resampled response too long");
        Polyphase pp;
        pp.init(ir_rate, device_rate);
        std::vector<double> src(ir_frames), dst(ir_len);
        resp.resize(size_t{channels} * ir_len);
        for(uint32_t ch = 0; ch < channels; ++ch)
        {
            for(uint32_t i = 0; i < ir_frames; ++i) src[i] = double(ir_in[size_t{i} * channels + ch]);
            pp.process(src.data(), ir_frames, dst.data(), ir_len);
            for(uint32_t i = 0; i < ir_len; ++i) resp[size_t{ch} * ir_len + i] = float(dst[i]);
        }
    }
    else
    {
        resp.resize(size_t{channels} * ir_len);
        for(uint32_t ch = 0; ch < channels; ++ch)
            for(uint32_t i = 0; i < ir_len; ++i) resp[size_t{ch} * ir_len + i] = ir_in[size_t{i} * channels + ch];
    }
    // convolution.cpp:377-378: segments after the first; at least one.  The spectrum ring gets 8
    // slots more than there are segments: one update completes up to 8 input blocks at once, and
    // their spectra must not replace ones an earlier block of the same update still reads.
    uint32_t segs = (ir_len + kSegLen - 1) / kSegLen;
    segs = std::max<uint32_t>(segs, 2u) - 1u;
    c->numSegs = segs;
    c->ringSlots = segs + kMaxBlocksPerUpdate;
    c->numChunks = (c->numSegs + kSegsPerChunk - 1) / kSegsPerChunk;

    std::vector<float> fir(size_t{channels} * kSegLen, 0.0f);
    // filter spectra, packed float2[128] per segment: bin 0 = (DC, Nyquist), bins 1..127 complex,
    // scaled by 1/256 (:444-457)
    std::vector<float> filt(size_t{channels} * c->numSegs * kFftLen, 0.0f);
    std::vector<std::complex<double>> buf(kFftLen);
    for(uint32_t ch = 0; ch < channels; ++ch)
    {
        const float *ir = resp.data() + size_t{ch} * ir_len;
        for(uint32_t k = 0; k < std::min(ir_len, kSegLen); ++k) fir[size_t{ch} * kSegLen + k] = ir[k];
        for(uint32_t s = 0; s < segs; ++s)
        {
            const size_t base = size_t{s + 1} * kSegLen;
            if(base >= ir_len) break;
            const size_t todo = std::min<size_t>(ir_len - base, kSegLen);
            for(size_t i = 0; i < kFftLen; ++i) buf[i] = (i < todo) ? std::complex<double>{double(ir[base + i]), 0.0} : 0.0;
            FftDouble(buf);
            float *dst = &filt[(size_t{ch} * c->numSegs + s) * kFftLen];
            constexpr double scale = 1.0 / double(kFftLen);
            dst[0] = float(buf[0].real() * scale);
            dst[1] = float(buf[kSegLen].real() * scale);
            for(uint32_t k = 1; k < kSegLen; ++k)
            {
                dst[2 * k] = float(buf[k].real() * scale);
                dst[2 * k + 1] = float(buf[k].imag() * scale);
            }
        }
    }
    std::vector<float> tw128(128), tw256(256);
    for(uint32_t k = 0; k < 64; ++k)
    {
        const double a = -2.0 * 3.14159265358979323846 * double(k) / 128.0;
        tw128[2 * k] = float(std::cos(a)); tw128[2 * k + 1] = float(std::sin(a));
    }
    for(uint32_t k = 0; k < 128; ++k)
    {
        const double a = -2.0 * 3.14159265358979323846 * double(k) / 256.0;
        tw256[2 * k] = float(std::cos(a)); tw256[2 * k + 1] = float(std::sin(a));
    }
    HIP_TRY(c->fir.alloc(fir.size())); HIP_TRY(c->fir.upload(fir.data(), fir.size()));
    HIP_TRY(c->filt.alloc(filt.size())); HIP_TRY(c->filt.upload(filt.data(), filt.size()));
    HIP_TRY(c->tw128.alloc(128)); HIP_TRY(c->tw128.upload(tw128.data(), 128));
    HIP_TRY(c->tw256.alloc(256)); HIP_TRY(c->tw256.upload(tw256.data(), 256));
    HIP_TRY(c->xhist.alloc(kFftLen)); HIP_TRY(c->xhist.zero());
    HIP_TRY(c->ring.alloc(size_t{c->ringSlots} * kFftLen)); HIP_TRY(c->ring.zero());
    HIP_TRY(c->outFifo.alloc(size_t{channels} * kFftLen)); HIP_TRY(c->outFifo.zero());
    HIP_TRY(c->partial.alloc(size_t{channels} * c->numChunks * 8 * kFftLen)); HIP_TRY(c->partial.zero());
    HIP_TRY(c->ticket.alloc(16)); HIP_TRY(c->ticket.zero());
    HIP_TRY(c->firOut.alloc(size_t{channels} * OALGPU_BUFFER_LINE_SIZE)); HIP_TRY(c->firOut.zero());
    HIP_TRY(c->chanOut.alloc(size_t{channels} * OALGPU_BUFFER_LINE_SIZE)); HIP_TRY(c->chanOut.zero());
    HIP_TRY(c->cur.alloc(size_t{channels} * num_out_lines)); HIP_TRY(c->cur.zero());
    HIP_TRY(c->tgt.alloc(size_t{channels} * num_out_lines)); HIP_TRY(c->tgt.zero());
    HIP_TRY(c->split.alloc(channels)); HIP_TRY(c->split.zero());
    HIP_TRY(c->hfScale.alloc(channels)); HIP_TRY(c->hfScale.zero());
    HIP_TRY(c->lfScale.alloc(channels)); HIP_TRY(c->lfScale.zero());
    HIP_TRY(c->hostIn.alloc(OALGPU_BUFFER_LINE_SIZE));
    HIP_TRY(c->hostOut.alloc(size_t{num_out_lines} * OALGPU_BUFFER_LINE_SIZE));
    *out = c.release();
    return OALGPU_OK;
}

void oalgpu_convolution_destroy(oalgpu_convolution *c)
{
    if(!c) return;
    (void)UseDevice(c->device);          // (a resident voice kernel on the device is told to leave first: it would sit out the synchronisation until its watchdog)
    (void)hipDeviceSynchronize();
    delete c;
}

int oalgpu_convolution_set_target_gains(oalgpu_convolution *c, const float *gains)
{
    if(!c || !gains) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(c->channels != 1) return Fail(OALGPU_ERR_INVALID, "oalgpu_convolution_set_target_gains: a multi-channel response takes oalgpu_convolution_set_channel_gains");
    return oalgpu_convolution_set_channel_gains(c, gains);
}

int oalgpu_convolution_set_channel_gains(oalgpu_convolution *c, const float *gains)
{
    if(!c || !gains) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseDevice(c->device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(c->tgt.p, gains, size_t{c->channels} * c->nlines * sizeof(float), hipMemcpyHostToDevice));
    return OALGPU_OK;
}

int oalgpu_convolution_set_upsample(oalgpu_convolution *c, const float *hf_scales, const float *lf_scales, float xover_norm)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseDevice(c->device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    if(!hf_scales || !lf_scales) { c->upsample = false; return OALGPU_OK; }
    if(!(xover_norm > 0.0f && xover_norm < 0.5f)) return Fail(OALGPU_ERR_INVALID, "oalgpu_convolution_set_upsample: 0 < xover_norm < 0.5");
    // the splitters are initialised by deviceUpdate (:364-365) and keep their state across update() calls
    if(!c->upsample)
    {
        std::vector<SplitterState> sp(c->channels, SplitterState{SplitterCoeff(xover_norm), 0.0f, 0.0f, 0.0f});
        HIP_TRY(c->split.upload(sp.data(), sp.size()));
    }
    HIP_TRY(c->hfScale.upload(hf_scales, c->channels));
    HIP_TRY(c->lfScale.upload(lf_scales, c->channels));
    c->upsample = true;
    return OALGPU_OK;
}

int oalgpu_convolution_process_device(oalgpu_convolution *c, void *hip_stream, const float *wet_in_dev,
    float *out_lines_dev, uint32_t n)
{
    if(!c || !wet_in_dev || !out_lines_dev || n == 0 || n > OALGPU_BUFFER_LINE_SIZE)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_convolution_process: bad arguments");
    if(int rc = UseDevice(c->device)) return rc;
    ConvLayoutHost h{};
    h.numSegs = c->numSegs; h.ringSlots = c->ringSlots; h.nlines = c->nlines; h.n = n;
    h.fifoPos = c->fifoPos; h.curSeg = c->curSeg;
    h.numBlocks = (c->fifoPos + n) / kSegLen;
    h.numChunks = c->numChunks; h.segsPerChunk = kSegsPerChunk;
    h.wetIn = wet_in_dev; h.xhist = c->xhist.p; h.ring = c->ring.p; h.filt = c->filt.p; h.fir = c->fir.p;
    h.outFifo = c->outFifo.p; h.partial = c->partial.p; h.cur = c->cur.p; h.tgt = c->tgt.p;
    h.outLines = out_lines_dev; h.tw128 = c->tw128.p; h.tw256 = c->tw256.p; h.ticket = c->ticket.p; h.firOut = c->firOut.p;
    h.channels = c->channels; h.upsample = c->upsample ? 1u : 0u; h.chanOut = c->chanOut.p;
    h.split = c->split.p; h.hfScale = c->hfScale.p; h.lfScale = c->lfScale.p;
    LaunchConvolution(static_cast<hipStream_t>(hip_stream), h);
    HIP_TRY(hipGetLastError());
    c->fifoPos = (c->fifoPos + n) % kSegLen;
    c->curSeg = (c->curSeg + c->ringSlots - h.numBlocks) % c->ringSlots;
    return OALGPU_OK;
}

int oalgpu_convolution_process(oalgpu_convolution *c, const float *wet_in, float *out_lines, uint32_t n)
{
    if(!c || !wet_in || !out_lines || n == 0 || n > OALGPU_BUFFER_LINE_SIZE)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_convolution_process: bad arguments");
    if(int rc = UseDevice(c->device)) return rc;
    const size_t outFloats = size_t{c->nlines} * OALGPU_BUFFER_LINE_SIZE;
    HIP_TRY(hipMemcpy(c->hostIn.p, wet_in, n * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->hostOut.p, out_lines, outFloats * sizeof(float), hipMemcpyHostToDevice));
    if(int rc = oalgpu_convolution_process_device(c, nullptr, c->hostIn.p, c->hostOut.p, n)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_lines, c->hostOut.p, outFloats * sizeof(float), hipMemcpyDeviceToHost));
    return OALGPU_OK;
}

} // extern "C"

namespace oalgpu { uint32_t ConvOutLines(const oalgpu_convolution *c) { return c ? c->nlines : 0u; } }
