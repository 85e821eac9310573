// HRTF data sets: .mhr / store hand-over, getCoeffs on the device, the direct-HRTF decoder, host-side helpers.
#include "api_context.hpp"

// the parsed (or handed-over) store becomes the context's: host copy, HBM copy, voice filter arrays
static int InstallHrtfData(oalgpu_context *c, HrtfData &&parsed)
{
    // a data set at another rate than the device's is brought to the device's rate as GetLoadedHrtf does
    // (core/hrtf.cpp:539-606: every HRIR through the polyphase resampler, delays and IrSize rescaled)
    if(parsed.sampleRate != c->desc.sample_rate) ResampleHrtfData(parsed, c->desc.sample_rate);
    if(int rc = oalgpu_sync(c)) return rc;           // a second load replaces buffers the streams may still read
    c->hrtfHost = std::move(parsed);
    const HrtfData &h = c->hrtfHost;
    HIP_TRY(c->hFieldDist.alloc(h.fieldDistance.size())); HIP_TRY(c->hFieldDist.upload(h.fieldDistance.data(), h.fieldDistance.size()));
    HIP_TRY(c->hEvCount.alloc(h.fieldEvCount.size())); HIP_TRY(c->hEvCount.upload(h.fieldEvCount.data(), h.fieldEvCount.size()));
    HIP_TRY(c->hAzCount.alloc(h.elevAzCount.size())); HIP_TRY(c->hAzCount.upload(h.elevAzCount.data(), h.elevAzCount.size()));
    HIP_TRY(c->hIrOffset.alloc(h.elevIrOffset.size())); HIP_TRY(c->hIrOffset.upload(h.elevIrOffset.data(), h.elevIrOffset.size()));
    HIP_TRY(c->hCoeffs.alloc(h.coeffs.size())); HIP_TRY(c->hCoeffs.upload(h.coeffs.data(), h.coeffs.size()));
    HIP_TRY(c->hDelays.alloc(h.delays.size())); HIP_TRY(c->hDelays.upload(h.delays.data(), h.delays.size()));
    HrtfStoreDev &d = c->hrtfDev;
    d.irSize = h.irSize; d.numFields = uint32_t(h.fieldDistance.size()); d.numElevs = uint32_t(h.elevAzCount.size());
    d.numIrs = h.numIrs();
    d.fieldDistance = c->hFieldDist.p; d.fieldEvCount = c->hEvCount.p; d.elevAzCount = c->hAzCount.p;
    d.elevIrOffset = c->hIrOffset.p; d.coeffs = c->hCoeffs.p; d.delays = c->hDelays.p;
    c->hrtfLoaded = true;
    ++c->hrtfGeneration;

    DeviceLayout &L = c->L;
    L.hrirs = c->hCoeffs.p;
    L.irSize = h.irSize;
    L.irStride = (h.irSize + 15u) & ~15u;
    if(L.hrtf)
    {
        const size_t n = size_t{L.numVoices} * L.irStride * 2;
        HIP_TRY(c->hrtfOld.alloc(n)); HIP_TRY(c->hrtfOld.zero()); L.hrtfOld = c->hrtfOld.p;
        HIP_TRY(c->hrtfTgt.alloc(n)); HIP_TRY(c->hrtfTgt.zero()); L.hrtfTgt = c->hrtfTgt.p;
        if(!c->directSet) c->dIrSize = h.irSize;
    }
    // (a set with more than 64 taps: the send rows of such an HRTF context go through stream rows)
    if(L.accLines && WaveKernelAccLines(L) == 0)
    {
        L.accLines = 0;
        if(int rc = AllocStreamRows(c)) return rc;
    }
    // one voice per wavefront, sixteen per workgroup (voice_wave16.hip): its grid is voices / 16 workgroups -- never more partial
    // buses than the context's buffers were sized for (the wavefront-per-voice kernel's grid has at least twice as many)
    if(c->useWave)
    {
        // (with sends: the voice-per-wavefront kernel leaves the sends' signals as stream rows and a small kernel behind it mixes them --
        // the HRTF path keeps its four wavefronts per SIMD; OALGPU_CTX_WAVE_PAIRS keeps the wet lines in the registers of the
        // two-voices-per-wavefront kernel, OALGPU_CTX_STREAM_ROWS its stream rows)
        const bool want16 = !(c->desc.flags & OALGPU_CTX_WAVE_PAIRS) && Wave16Applies(L)
            && (L.numSends == 0 || !(c->desc.flags & (OALGPU_CTX_STREAM_ROWS | OALGPU_CTX_PROFILE | OALGPU_CTX_SLICE_LINES)));
        uint32_t cus = 256u;
        {
            hipDeviceProp_t prop{};
            if(hipGetDeviceProperties(&prop, c->desc.device) == hipSuccess && prop.multiProcessorCount > 0) cus = uint32_t(prop.multiProcessorCount);
            else (void)hipGetLastError();
        }
        L.wave16 = want16 ? Wave16WavesFor(L.numVoices, cus) : 0u;
        if(L.wave16 && L.numSends)
        {
            L.accLines = 0;
            if(!L.streams) { if(int rc = AllocStreamRows(c)) return rc; }
        }
        const uint32_t groups = std::max<uint32_t>(1u, WaveKernelGroups(L));
        if(groups > c->groupsAllocated) return Fail(OALGPU_ERR_INVALID, "internal: the voice kernel's grid outgrew the partial buses");
        L.numGroups = groups; L.numLineGroups = groups;
        if(L.wave16 && (L.wave16 != 16u || L.numSends)) c->res.enabled = false;     // (the resident launch: the 16-wavefront form, or OALGPU_CTX_WAVE_PAIRS' kernel)
    }
    return OALGPU_OK;
}

int oalgpu_hrtf_load_mhr(oalgpu_context *c, const void *data, size_t size)
{
    if(!c || !data) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseCtx(c)) return rc;
    HrtfData parsed;
    const std::string err = ParseMhr(data, size, parsed);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    return InstallHrtfData(c, std::move(parsed));
}

/* An HrtfStore already in memory (core/hrtf.h:22-59), as the device holds it in DeviceBase::mHrtf once
 * GetLoadedHrtf (core/hrtf.cpp:471-620) has loaded -- and resampled -- a data set: the same arrays oalgpu_hrtf_raw
 * hands out. */
int oalgpu_hrtf_load_store(oalgpu_context *c, uint32_t sample_rate, uint32_t ir_size, const float *field_distance,
    const uint8_t *field_evcount, uint32_t num_fields, const uint16_t *elev_azcount, const uint16_t *elev_iroffset,
    uint32_t num_elevs, const float *coeffs, const uint8_t *delays, uint32_t num_irs)
{
    if(!c || !field_distance || !field_evcount || !elev_azcount || !elev_iroffset || !coeffs || !delays)
        return Fail(OALGPU_ERR_INVALID, "null argument");
    if(sample_rate == 0 || ir_size < 8 || ir_size > kHrirLen || num_fields == 0 || num_elevs == 0 || num_irs == 0)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: bad sizes");
    uint32_t evs = 0;
    for(uint32_t f = 0; f < num_fields; ++f)
    {
        if(field_evcount[f] < 2) return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: a field needs at least two elevations");
        evs += field_evcount[f];
    }
    if(evs != num_elevs) return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: the fields' elevation counts do not add up to num_elevs");
    for(uint32_t e = 0; e < num_elevs; ++e)
        if(elev_azcount[e] == 0 || uint32_t(elev_iroffset[e]) + elev_azcount[e] > num_irs)
            return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_load_store: an elevation's HRIRs lie outside the store");
    if(int rc = UseCtx(c)) return rc;
    HrtfData h;
    h.sampleRate = sample_rate; h.irSize = ir_size;
    h.fieldDistance.assign(field_distance, field_distance + num_fields);
    h.fieldEvCount.assign(field_evcount, field_evcount + num_fields);
    h.elevAzCount.assign(elev_azcount, elev_azcount + num_elevs);
    h.elevIrOffset.assign(elev_iroffset, elev_iroffset + num_elevs);
    h.coeffs.assign(coeffs, coeffs + size_t{num_irs} * kHrirLen * 2);
    h.delays.assign(delays, delays + size_t{num_irs} * 2);
    return InstallHrtfData(c, std::move(h));
}

int oalgpu_hrtf_info_get(oalgpu_context *c, oalgpu_hrtf_info *out)
{
    if(!c || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "no HRTF data set loaded");
    const HrtfData &h = c->hrtfHost;
    out->sample_rate = h.sampleRate; out->ir_size = h.irSize;
    out->num_fields = uint32_t(h.fieldDistance.size()); out->num_elevs = uint32_t(h.elevAzCount.size());
    out->num_irs = h.numIrs();
    return OALGPU_OK;
}

int oalgpu_hrtf_raw(oalgpu_context *c, float *field_distance, uint8_t *field_evcount, uint16_t *elev_azcount,
    uint16_t *elev_iroffset, float *coeffs, uint8_t *delays)
{
    if(!c) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "no HRTF data set loaded");
    const HrtfData &h = c->hrtfHost;
    std::copy(h.fieldDistance.begin(), h.fieldDistance.end(), field_distance);
    std::copy(h.fieldEvCount.begin(), h.fieldEvCount.end(), field_evcount);
    std::copy(h.elevAzCount.begin(), h.elevAzCount.end(), elev_azcount);
    std::copy(h.elevIrOffset.begin(), h.elevIrOffset.end(), elev_iroffset);
    std::copy(h.coeffs.begin(), h.coeffs.end(), coeffs);
    std::copy(h.delays.begin(), h.delays.end(), delays);
    return OALGPU_OK;
}

int oalgpu_hrtf_get_coeffs(oalgpu_context *c, const float *dirs, size_t count, float *coeffs, uint32_t *delays)
{
    if(!c || !dirs || !coeffs || !delays || count == 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_get_coeffs: bad arguments");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "no HRTF data set loaded");
    if(int rc = UseCtx(c)) return rc;
    DevBuf<float> dDirs, dCo;
    DevBuf<uint32_t> dDel;
    HIP_TRY(dDirs.alloc(count * 4)); HIP_TRY(dDirs.upload(dirs, count * 4));
    HIP_TRY(dCo.alloc(count * kHrirLen * 2));
    HIP_TRY(dDel.alloc(count * 2));
    LaunchGetCoeffs(c->stream, c->hrtfDev, dDirs.p, uint32_t(count), dCo.p, dDel.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(dCo.download(coeffs, count * kHrirLen * 2));
    HIP_TRY(dDel.download(delays, count * 2));
    return OALGPU_OK;
}

int oalgpu_set_direct_hrtf(oalgpu_context *c, const float *chan_coeffs, const float *hfscales, float xover_norm,
    uint32_t irsize)
{
    if(!c || !chan_coeffs || !hfscales || irsize < 8 || irsize > kHrirLen) return Fail(OALGPU_ERR_INVALID, "oalgpu_set_direct_hrtf: bad arguments");
    if(int rc = UseCtx(c)) return rc;
    const uint32_t nd = c->L.numDry;
    std::vector<SplitterState> sp(nd);
    for(auto &s : sp) { s.coeff = SplitterCoeff(xover_norm); s.lpZ1 = s.lpZ2 = s.apZ1 = 0.0f; }
    if(int rc = oalgpu_sync(c)) return rc;
    HIP_TRY(c->dSplit.upload(sp.data(), nd));
    HIP_TRY(c->dSplit2.upload(sp.data(), nd));
    c->dSplitCur = 0;
    c->dSplitCoeff = SplitterCoeff(xover_norm); c->runPowerSeg = 0;
    HIP_TRY(c->dHfScale.upload(hfscales, nd));
    {   // MixDirectHrtf applies IrSize taps (rounded up to even: ApplyCoeffs works on pairs); the decoder of a resampled
        // data set carries non-zero taps beyond that, which the fixed-length FIR of the FAST post-process must not see
        std::vector<float> cc(chan_coeffs, chan_coeffs + size_t{nd} * kHrirLen * 2);
        const uint32_t live = (irsize + 1u) & ~1u;
        for(uint32_t ch = 0; ch < nd; ++ch)
            for(uint32_t k = live; k < uint32_t(kHrirLen); ++k) { cc[(size_t{ch} * kHrirLen + k) * 2] = 0.0f; cc[(size_t{ch} * kHrirLen + k) * 2 + 1] = 0.0f; }
        HIP_TRY(c->dCoeffs.upload(cc.data(), cc.size()));
    }
    c->dIrSize = irsize;
    c->directSet = true;
    return OALGPU_OK;
}

/* DirectHrtfState::build (core/hrtf.cpp:266-366) on the context's data set: the ambisonic-to-binaural decoder of the
 * HRTF post-process from the host's virtual-speaker layout (alc/panning.cpp:861-1038 holds the reference's: AmbiPoints,
 * AmbiMatrix rows of 16, AmbiOrderHFGain[5]; InitHrtfPanning passes device->mIrSize = the store's IrSize and
 * perHrirMin for orders >= 3).  One-time host work in double precision, then oalgpu_set_direct_hrtf. */
int oalgpu_set_direct_hrtf_from_store(oalgpu_context *c, const float *points, const float *matrix, uint32_t num_points,
    const float *order_hf_gain, float xover_freq, uint32_t ir_size, int per_hrir_min)
{
    if(!c || !points || !matrix || !order_hf_gain || num_points == 0 || c->L.numDry > 16)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_set_direct_hrtf_from_store: bad arguments");
    if(!c->hrtfLoaded) return Fail(OALGPU_ERR_NO_HRTF, "oalgpu_set_direct_hrtf_from_store: no data set loaded");
    const DirectHrtfBuild b = BuildDirectHrtf(c->hrtfHost, ir_size ? ir_size : c->hrtfHost.irSize, per_hrir_min != 0, points, matrix,
        num_points, c->L.numDry, xover_freq, order_hf_gain);
    return oalgpu_set_direct_hrtf(c, b.coeffs.data(), b.hfScale.data(), b.xoverNorm, std::max<uint32_t>(b.irSize, 8u));
}

/* The same two pieces of one-time HRTF set-up as pure host functions (no device): a data set brought to `device_rate`
 * (0: its own) -- info, and coeffs (num_irs x 128 x 2) / delays (num_irs x 2) when not NULL -- and the decoder build. */
int oalgpu_hrtf_parse_host(const void *mhr, size_t size, uint32_t device_rate, oalgpu_hrtf_info *info, float *coeffs, uint8_t *delays)
{
    if(!mhr || !info) return Fail(OALGPU_ERR_INVALID, "null argument");
    HrtfData h;
    const std::string err = ParseMhr(mhr, size, h);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    if(device_rate) ResampleHrtfData(h, device_rate);
    info->sample_rate = h.sampleRate; info->ir_size = h.irSize; info->num_fields = uint32_t(h.fieldDistance.size());
    info->num_elevs = uint32_t(h.elevAzCount.size()); info->num_irs = h.numIrs();
    if(coeffs) std::memcpy(coeffs, h.coeffs.data(), h.coeffs.size() * sizeof(float));
    if(delays) std::memcpy(delays, h.delays.data(), h.delays.size());
    return OALGPU_OK;
}

// the host's view of the loaded store, for HrtfBlendFor
HrtfStoreDev HostStoreView(const HrtfData &h)
{
    HrtfStoreDev st{};
    st.irSize = h.irSize; st.numFields = uint32_t(h.fieldDistance.size());
    st.numElevs = uint32_t(h.elevAzCount.size()); st.numIrs = h.numIrs();
    st.fieldDistance = h.fieldDistance.data(); st.fieldEvCount = h.fieldEvCount.data();
    st.elevAzCount = h.elevAzCount.data(); st.elevIrOffset = h.elevIrOffset.data();
    st.coeffs = h.coeffs.data(); st.delays = h.delays.data();
    return st;
}

int oalgpu_hrtf_blend_host(const void *mhr, size_t size, uint32_t device_rate, const float *dirs, size_t count,
    uint32_t *idx, float *w, float *pass, uint32_t *delays)
{
    if(!mhr || !dirs || !idx || !w || !pass || !delays) return Fail(OALGPU_ERR_INVALID, "null argument");
    HrtfData h;
    const std::string err = ParseMhr(mhr, size, h);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    if(device_rate) ResampleHrtfData(h, device_rate);
    const HrtfStoreDev store = HostStoreView(h);
    for(size_t i = 0; i < count; ++i)
    {
        const HrirBlend b = HrtfBlendFor(store, dirs[4 * i], dirs[4 * i + 1], dirs[4 * i + 2], dirs[4 * i + 3]);
        for(int k = 0; k < 4; ++k) { idx[4 * i + k] = b.idx[k]; w[4 * i + k] = b.w[k]; }
        pass[i] = b.passthru;
        delays[2 * i] = b.delay[0]; delays[2 * i + 1] = b.delay[1];
    }
    return OALGPU_OK;
}

int oalgpu_hrtf_build_direct_host(const void *mhr, size_t size, uint32_t device_rate, uint32_t ir_size, int per_hrir_min,
    const float *points, const float *matrix, uint32_t num_points, uint32_t num_chans, float xover_freq, const float *order_hf_gain,
    float *out_coeffs, float *out_hfscales, float *out_xover_norm, uint32_t *out_irsize)
{
    if(!mhr || !points || !matrix || !order_hf_gain || !out_coeffs || !out_hfscales || num_chans < 1 || num_chans > 16 || num_points == 0)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_hrtf_build_direct_host: bad arguments");
    HrtfData h;
    const std::string err = ParseMhr(mhr, size, h);
    if(!err.empty()) return Fail(OALGPU_ERR_INVALID, "mhr: " + err);
    if(device_rate) ResampleHrtfData(h, device_rate);
    const DirectHrtfBuild b = BuildDirectHrtf(h, ir_size ? ir_size : h.irSize, per_hrir_min != 0, points, matrix, num_points,
        num_chans, xover_freq, order_hf_gain);
    std::memcpy(out_coeffs, b.coeffs.data(), b.coeffs.size() * sizeof(float));
    std::memcpy(out_hfscales, b.hfScale.data(), b.hfScale.size() * sizeof(float));
    if(out_xover_norm) *out_xover_norm = b.xoverNorm;
    if(out_irsize) *out_irsize = b.irSize;
    return OALGPU_OK;
}


