// The per-call operators of include/oalgpu.h: version, tables, oalgpu_resample / _mix / _mix_hrtf*, the sample converter.
#include "api_context.hpp"

const char *oalgpu_version(void) { return "oalgpu 0.1 (gfx950)"; }
const char *oalgpu_last_error(void) { return gLastError.c_str(); }

int oalgpu_device_count(void)
{
    int count = 0;
    if(hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count < 0 ? 0 : count;
}

// ---------------------------------------------------------------- host-side tables / params
int oalgpu_bsinc_table_get(int which, oalgpu_bsinc_table *out)
{
    const BsincTable *t = GetBsincTable(which);
    if(!t || !out) return Fail(OALGPU_ERR_INVALID, "bsinc table: which must be 12, 24 or 48");
    out->scaleBase = t->scaleBase; out->scaleRange = t->scaleRange;
    std::memcpy(out->m, t->m, sizeof(out->m));
    std::memcpy(out->filterOffset, t->filterOffset, sizeof(out->filterOffset));
    out->tab = t->tab.data(); out->tablen = t->tab.size();
    return OALGPU_OK;
}

int oalgpu_cubic_table_get(int which, float *out)
{
    const CubicTable *t = GetCubicTable(which);
    if(!t || !out) return Fail(OALGPU_ERR_INVALID, "cubic table: which must be 0 or 1");
    std::memcpy(out, t->phase, sizeof(t->phase));
    return OALGPU_OK;
}

int oalgpu_prepare_resampler(int resampler, uint32_t increment, oalgpu_interp_state *out)
{
    if(!out || resampler < 0 || resampler > OALGPU_RESAMPLER_BSINC48) return Fail(OALGPU_ERR_INVALID, "bad resampler");
    PrepareResampler(resampler, increment, out);
    return OALGPU_OK;
}

void oalgpu_biquad_reset(oalgpu_biquad *f)
{
    std::memset(f, 0, sizeof(*f));
    f->b0 = 1.0f; f->tb0 = 1.0f; f->counter = -1;
}

void oalgpu_biquad_set_params_from_slope(oalgpu_biquad *f, int type, float f0norm, float gain, float slope)
{
    float c[5];
    DesignBiquadFromSlope(type, f0norm, gain, slope, c);
    ApplyBiquadTarget(f, c);
}

void oalgpu_splitter_init(oalgpu_splitter *s, float f0norm)
{
    s->coeff = SplitterCoeff(f0norm);
    s->lp_z1 = s->lp_z2 = s->ap_z1 = 0.0f;
}

// ---------------------------------------------------------------- per-call kernels
int oalgpu_resample(int device, int mode, int resampler, uint32_t increment, const float *src, size_t srclen,
    uint32_t frac, float *dst, size_t n)
{
    if(!src || !dst || n == 0 || n > 4096 || frac >= kFracOne || increment < 1) return Fail(OALGPU_ERR_INVALID, "oalgpu_resample: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    oalgpu_interp_state st;
    if(int rc = oalgpu_prepare_resampler(resampler, increment, &st)) return rc;
    const TableBlob &blob = Blob();
    DevBuf<float> dTab, dSrc, dDst;
    HIP_TRY(dTab.alloc(blob.data.size())); HIP_TRY(dTab.upload(blob.data.data(), blob.data.size()));
    HIP_TRY(dSrc.alloc(srclen)); HIP_TRY(dSrc.upload(src, srclen));
    HIP_TRY(dDst.alloc(n));
    ResampleSpec spec{st.kind, st.m, st.l, st.sf, dTab.p + blob.filterBase(st)};
    LaunchResample(nullptr, mode == OALGPU_MATH_EXACT, spec, dSrc.p, frac, increment, dDst.p, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dDst.download(dst, n));
    return OALGPU_OK;
}

/* ---- SampleConverter (core/converter.h:15-59, core/converter.cpp:175-330): the second consumer of the resamplers ----
 * The integer bookkeeping of convert() / availableOut() runs here on the host, exactly as the reference's; the
 * float work of a call -- LoadSample<T>, the resampler, StoreSample<T> for every channel and chunk -- is one launch
 * (SampleConvertKernel, output_kernels.hip). */
struct oalgpu_converter {
    int device{0};
    int srcType{0}, dstType{0};
    uint32_t channels{1}, increment{kFracOne};
    uint32_t srcPrepCount{kMaxPad}, fracOffset{0};          // mSrcPrepCount, mFracOffset
    oalgpu_interp_state st{};
    DevBuf<float> tables, prev, prev2;
    DevBuf<unsigned char> src, dst;
    DevBuf<ConvertChunk> chunks;
    bool flip{false};                                       // which of prev / prev2 holds the current PrevSamples
};

static size_t DevFmtBytes(int type) { static const size_t b[7] = {1, 1, 2, 2, 4, 4, 4}; return b[type]; }

int oalgpu_converter_create(int device, int src_type, int dst_type, uint32_t channels, uint32_t src_rate, uint32_t dst_rate,
    int resampler, oalgpu_converter **out)
{
    if(!out || src_type < OALGPU_OUT_I8 || src_type > OALGPU_OUT_F32 || dst_type < OALGPU_OUT_I8 || dst_type > OALGPU_OUT_F32
        || channels < 1 || channels > 64 || src_rate < 1 || dst_rate < 1 || resampler < 0 || resampler > OALGPU_RESAMPLER_BSINC48)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_converter_create: bad arguments");
    *out = nullptr;
    if(int rc = UseDevice(device)) return rc;
    auto c = std::make_unique<oalgpu_converter>();
    c->device = device; c->srcType = src_type; c->dstType = dst_type; c->channels = channels;
    // converter.cpp:199-201: step = clamp(round(srcRate * MixerFracOne / dstRate), 1, MaxPitch * MixerFracOne)
    const double step = std::min(std::max(std::round(double(src_rate) * double(kFracOne) / double(dst_rate)), 1.0), 10.0 * double(kFracOne));
    c->increment = uint32_t(step);
    if(c->increment != kFracOne)
        if(int rc = oalgpu_prepare_resampler(resampler, c->increment, &c->st)) return rc;
    const TableBlob &blob = Blob();
    HIP_TRY(c->tables.alloc(blob.data.size())); HIP_TRY(c->tables.upload(blob.data.data(), blob.data.size()));
    HIP_TRY(c->prev.alloc(size_t{channels} * kMaxPad)); HIP_TRY(c->prev.zero());
    HIP_TRY(c->prev2.alloc(size_t{channels} * kMaxPad)); HIP_TRY(c->prev2.zero());
    *out = c.release();
    return OALGPU_OK;
}

void oalgpu_converter_destroy(oalgpu_converter *c)
{
    if(!c) return;
    (void)UseDevice(c->device);          // (a resident voice kernel on the device is told to leave first: it would sit out the synchronisation until its watchdog)
    (void)hipDeviceSynchronize();
    delete c;
}

/* SampleConverter::availableOut, converter.cpp:216-234 */
uint32_t oalgpu_converter_available_out(const oalgpu_converter *c, uint32_t src_frames)
{
    if(!c || src_frames < 1) return 0;
    const uint32_t prep = c->srcPrepCount;
    if(prep < kMaxPad && kMaxPad - prep >= src_frames) return 0;
    uint64_t size = uint64_t{prep} + src_frames - kMaxPad;
    size <<= kFracBits;
    size -= c->fracOffset;
    const uint64_t n = (size + c->increment - 1) / c->increment;
    return uint32_t(std::min<uint64_t>(std::max<uint64_t>(n, 1), 2147483647ull));
}

int oalgpu_converter_convert(oalgpu_converter *c, const void **src, uint32_t *src_frames, void *dst, uint32_t dst_frames)
{
    if(!c || !src || !src_frames || (!*src && *src_frames) || (!dst && dst_frames))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_converter_convert: bad arguments");
    if(int rc = UseDevice(c->device)) return rc;
    const uint32_t inc = c->increment, total = *src_frames;
    // ---- the chunk loop of convert(), integers only (converter.cpp:248-327)
    std::vector<ConvertChunk> chunks;
    uint32_t pos = 0, left = total, consumed = 0, prep = c->srcPrepCount, frac = c->fracOffset, base = 0;
    const uint32_t prep0 = prep;
    while(pos < dst_frames && left > 0)
    {
        const uint32_t readable = std::min(left, uint32_t(kLine) - prep);
        if(prep < kMaxPad && kMaxPad - prep >= readable)
        {   // not enough input for one output: keep what came (the reference reports it all as taken)
            prep += readable;
            left = 0;
            break;
        }
        uint64_t size = uint64_t{prep} + readable - kMaxPad;
        size <<= kFracBits;
        size -= frac;
        uint32_t dstSize = uint32_t(std::min<uint64_t>(std::max<uint64_t>((size + inc - 1) / inc, 1), uint64_t(kLine)));
        dstSize = std::min(dstSize, dst_frames - pos);
        const uint32_t posEnd = dstSize * inc + frac;
        const uint32_t srcEnd = posEnd >> kFracBits;
        const uint32_t nextPrep = std::min(prep + readable - srcEnd, uint32_t(kMaxPad));
        chunks.push_back(ConvertChunk{base, frac, pos, dstSize});
        const uint32_t srcRead = std::min(left, srcEnd + nextPrep - prep);
        base += srcEnd;
        prep = nextPrep;
        frac = posEnd & kFracMask;
        consumed += srcRead; left -= srcRead;
        pos += dstSize;
    }
    if(prep != c->srcPrepCount || !chunks.empty())
    {
        const size_t srcBytes = size_t{total} * c->channels * DevFmtBytes(c->srcType);
        const size_t dstBytes = size_t{pos} * c->channels * DevFmtBytes(c->dstType);
        if(c->src.n < srcBytes) HIP_TRY(c->src.alloc(srcBytes));
        if(c->dst.n < dstBytes) HIP_TRY(c->dst.alloc(std::max<size_t>(dstBytes, 16)));
        if(c->chunks.n < chunks.size()) HIP_TRY(c->chunks.alloc(std::max<size_t>(chunks.size(), 1)));
        HIP_TRY(hipMemcpy(c->src.p, *src, srcBytes, hipMemcpyHostToDevice));
        if(!chunks.empty()) HIP_TRY(c->chunks.upload(chunks.data(), chunks.size()));
        ConvertJob J{};
        const TableBlob &blob = Blob();
        J.spec = ResampleSpec{c->st.kind, c->st.m, c->st.l, c->st.sf, inc != kFracOne ? c->tables.p + blob.filterBase(c->st) : c->tables.p};
        J.increment = inc;
        J.prev = c->flip ? c->prev2.p : c->prev.p; J.newPrev = c->flip ? c->prev.p : c->prev2.p;
        J.src = c->src.p; J.dst = c->dst.p;
        J.prep0 = prep0; J.srcFrames = total; J.channels = c->channels; J.numChunks = uint32_t(chunks.size());
        J.endBase = base; J.nextPrep = prep;
        J.srcType = c->srcType; J.dstType = c->dstType;
        LaunchSampleConvert(nullptr, J, c->chunks.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        if(dstBytes) HIP_TRY(hipMemcpy(dst, c->dst.p, dstBytes, hipMemcpyDeviceToHost));
        c->flip = !c->flip;
    }
    c->srcPrepCount = prep; c->fracOffset = frac;
    *src = static_cast<const unsigned char*>(*src) + size_t{consumed} * c->channels * DevFmtBytes(c->srcType);
    *src_frames = left;
    return int(pos);
}

int oalgpu_mix(int device, const float *in, size_t n, float *out, size_t nlines, float *current_gains,
    const float *target_gains, size_t counter, size_t outpos)
{
    if(!in || !out || !current_gains || !target_gains || n == 0 || n + outpos > kLine || nlines == 0 || nlines > 64)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    DevBuf<float> dIn, dOut, dCur, dTgt;
    HIP_TRY(dIn.alloc(n)); HIP_TRY(dIn.upload(in, n));
    HIP_TRY(dOut.alloc(nlines * kLine)); HIP_TRY(dOut.upload(out, nlines * kLine));
    HIP_TRY(dCur.alloc(nlines)); HIP_TRY(dCur.upload(current_gains, nlines));
    HIP_TRY(dTgt.alloc(nlines)); HIP_TRY(dTgt.upload(target_gains, nlines));
    LaunchMix(nullptr, dIn.p, uint32_t(n), dOut.p, uint32_t(nlines), dCur.p, dTgt.p, uint32_t(counter), uint32_t(outpos));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dOut.download(out, nlines * kLine));
    HIP_TRY(dCur.download(current_gains, nlines));
    return OALGPU_OK;
}

static int MixHrtfCommon(int device, int mode, const float *in, float *accum, uint32_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float step, const float *oldcoeffs, const uint32_t olddelay[2], float oldgain,
    int blend, size_t n)
{
    if(!in || !accum || !coeffs || !delay || n == 0 || n > kLine || irsize < 8 || irsize > kHrirLen
        || delay[0] > 63 || delay[1] > 63)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_hrtf: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    constexpr size_t accLen = (kLine + kHrirLen) * 2;
    DevBuf<float> dIn, dAcc, dCo, dOld;
    HIP_TRY(dIn.alloc(n + kHist)); HIP_TRY(dIn.upload(in, n + kHist));
    HIP_TRY(dAcc.alloc(accLen)); HIP_TRY(dAcc.upload(accum, accLen));
    HIP_TRY(dCo.alloc(kHrirLen * 2)); HIP_TRY(dCo.upload(coeffs, kHrirLen * 2));
    HIP_TRY(dOld.alloc(kHrirLen * 2));
    if(blend) HIP_TRY(dOld.upload(oldcoeffs, kHrirLen * 2));
    LaunchMixHrtf(nullptr, mode == OALGPU_MATH_EXACT, dIn.p, dAcc.p, irsize, dCo.p, delay[0], delay[1], gain, step,
        dOld.p, blend ? olddelay[0] : 0, blend ? olddelay[1] : 0, oldgain, blend, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dAcc.download(accum, accLen));
    return OALGPU_OK;
}

int oalgpu_mix_hrtf(int device, int mode, const float *in, float *accum, uint32_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float gainstep, size_t n)
{ return MixHrtfCommon(device, mode, in, accum, irsize, coeffs, delay, gain, gainstep, nullptr, nullptr, 0.0f, 0, n); }

int oalgpu_mix_hrtf_blend(int device, int mode, const float *in, float *accum, uint32_t irsize,
    const float *oldcoeffs, const uint32_t olddelay[2], float oldgain, const float *newcoeffs,
    const uint32_t newdelay[2], float newgainstep, size_t n)
{
    if(!oldcoeffs || !olddelay || olddelay[0] > 63 || olddelay[1] > 63) return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_hrtf_blend: bad arguments");
    return MixHrtfCommon(device, mode, in, accum, irsize, newcoeffs, newdelay, 0.0f, newgainstep, oldcoeffs, olddelay,
        oldgain, 1, n);
}

int oalgpu_mix_direct_hrtf(int device, int mode, float *left, float *right, const float *in, size_t nch,
    float *accum, oalgpu_splitter *splitters, const float *hfscales, const float *chan_coeffs, size_t irsize, size_t n)
{
    if(!left || !right || !in || !accum || !splitters || !hfscales || !chan_coeffs || nch == 0 || nch > 64 || n == 0
        || n > kLine || irsize < 8 || irsize > kHrirLen)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_mix_direct_hrtf: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    constexpr size_t accLen = (kLine + kHrirLen) * 2;
    DevBuf<float> dL, dR, dIn, dAcc, dHf, dCo, dTemp;
    DevBuf<SplitterState> dSp;
    HIP_TRY(dL.alloc(kLine)); HIP_TRY(dL.upload(left, kLine));
    HIP_TRY(dR.alloc(kLine)); HIP_TRY(dR.upload(right, kLine));
    HIP_TRY(dIn.alloc(nch * kLine)); HIP_TRY(dIn.upload(in, nch * kLine));
    HIP_TRY(dAcc.alloc(accLen)); HIP_TRY(dAcc.upload(accum, accLen));
    HIP_TRY(dHf.alloc(nch)); HIP_TRY(dHf.upload(hfscales, nch));
    HIP_TRY(dCo.alloc(nch * kHrirLen * 2)); HIP_TRY(dCo.upload(chan_coeffs, nch * kHrirLen * 2));
    HIP_TRY(dTemp.alloc(nch * kLine + accLen));
    HIP_TRY(dSp.alloc(nch)); HIP_TRY(dSp.upload(reinterpret_cast<const SplitterState*>(splitters), nch));
    if(mode == OALGPU_MATH_EXACT)
        LaunchMixDirectHrtf(nullptr, true, dL.p, dR.p, dIn.p, uint32_t(nch), dAcc.p, dSp.p, dHf.p, dCo.p,
            uint32_t(irsize), uint32_t(n), dTemp.p);
    else
        LaunchPostDirectHrtfFast(nullptr, dL.p, dR.p, dIn.p, uint32_t(nch), dAcc.p, dSp.p, dHf.p, dCo.p,
            uint32_t(irsize), uint32_t(n), dTemp.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dL.download(left, kLine)); HIP_TRY(dR.download(right, kLine));
    HIP_TRY(dAcc.download(accum, accLen));
    HIP_TRY(dSp.download(reinterpret_cast<SplitterState*>(splitters), nch));
    return OALGPU_OK;
}

int oalgpu_biquad_dual_process(int device, oalgpu_biquad *f0, oalgpu_biquad *f1, const float *src, float *dst, size_t n)
{
    if(!f0 || !f1 || !src || !dst || n == 0 || n > (1u << 20)) return Fail(OALGPU_ERR_INVALID, "oalgpu_biquad_dual_process: bad arguments");
    if(int rc = UseDevice(device)) return rc;
    DevBuf<BiquadState> dF;
    DevBuf<float> dSrc, dDst;
    BiquadState st[2];
    std::memcpy(&st[0], f0, sizeof(BiquadState)); std::memcpy(&st[1], f1, sizeof(BiquadState));
    HIP_TRY(dF.alloc(2)); HIP_TRY(dF.upload(st, 2));
    HIP_TRY(dSrc.alloc(n)); HIP_TRY(dSrc.upload(src, n));
    HIP_TRY(dDst.alloc(n));
    LaunchBiquadDual(nullptr, dF.p, dF.p + 1, dSrc.p, dDst.p, uint32_t(n));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(dDst.download(dst, n));
    HIP_TRY(dF.download(st, 2));
    std::memcpy(f0, &st[0], sizeof(BiquadState)); std::memcpy(f1, &st[1], sizeof(BiquadState));
    return OALGPU_OK;
}

// ---------------------------------------------------------------- context

