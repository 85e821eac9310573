/* oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Thin C-ABI harness over the REAL reference mixer (kcat/openal-soft, compiled in
 * place from /root/reference by oracle/Makefile).  It implements oracle/oalref.h by
 * calling the reference's own functions:
 *   Resample_*_{C,SSE,SSE2,SSE4}   core/mixer/defs.h:78-141
 *   Mix_*, MixHrtf_*, MixHrtfBlend_*, MixDirectHrtf_*   core/mixer/mixer_{c,sse}.cpp
 *   BiquadInterpFilter, BandSplitter   core/filters/{biquad,splitter}.cpp
 *   PrepareResampler   alc/alu.cpp:253-281
 *   LoadHrtf / HrtfStore::getCoeffs   core/hrtf_loader.cpp:726, core/hrtf.cpp:192-260
 *   Voice::prepare / Voice::mix   core/voice.cpp:1235,988
 * No reference source is copied; this file only drives it.
 */
#include "config.h"
#include "config_simd.h"

/* The harness needs to read/write filter state that the reference keeps private.
 * Everything those two headers include is pulled in first so the keyword
 * redefinitions only touch biquad.h / splitter.h themselves. */
#include <algorithm>
#include <array>
#include <cmath>
#include <numbers>
#include <span>
#include "alnumeric.h"
#include "opthelpers.h"
#define private public
#define protected public
#define class struct
#include "core/filters/biquad.h"
#include "core/filters/splitter.h"
#undef class
#undef private
#undef protected

#include <algorithm>
#include <array>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <vector>

#include "alnumeric.h"
#include "core/bformatdec.h"
#include "core/bs2b.h"
#include "core/bsinc_tables.h"
#include "core/encoderbase.hpp"
#include "core/front_stablizer.h"
#include "core/mastering.h"
#include "core/buffer_storage.h"
#include "core/context.h"
#include "core/cpu_caps.h"
#include "core/cubic_tables.h"
#include "core/device.h"
#include "core/effectslot.h"
#include "core/fpu_ctrl.h"
#include "core/hrtf.h"
#include "core/hrtf_loader.hpp"
#include "core/mixer.h"
#include "core/mixer/defs.h"
#include "core/mixer/hrtfdefs.h"
#include "core/voice.h"
#include "core/async_event.h"
#include "ringbuffer.h"
#include "polyphase_resampler.h"
#include "core/converter.h"

#include "oalref.h"

#include "alc/alu.h"
#include "alc/effects/base.h"
#include "core/effects/base.h"

namespace {

bool gSimd = true;
bool gInit = false;

void EnsureInit()
{
    if(gInit) return;
    gInit = true;
    if(auto info = GetCPUInfo())
        CPUCapFlags = info->mCaps;
}

void ApplySimd()
{
    EnsureInit();
    static CPUCapBitset detected = CPUCapFlags;
    CPUCapFlags = gSimd ? detected : CPUCapBitset{};
    aluInit({}, 1.0f);
    Voice::InitMixer(std::nullopt);
}

} // namespace
/* for ref_conv.cpp: the mixer function pointers follow oal_set_simd there too */
void oalref_apply_simd() { ApplySimd(); }
namespace {

std::unique_ptr<HrtfStore> gHrtfOwner;
HrtfStorePtr gHrtfLoaded;                   /* a store GetLoadedHrtf returned (oal_hrtf_load_for_rate) */
HrtfStore *gHrtf = nullptr;

BSincTable const *GetBsinc(int which)
{
    switch(which)
    {
    case 12: return &gBSinc12;
    case 24: return &gBSinc24;
    case 48: return &gBSinc48;
    }
    return nullptr;
}

void ToBiquad(BiquadInterpFilter const &f, oal_biquad *o)
{
    o->z1 = f.mZ1; o->z2 = f.mZ2;
    o->b0 = f.mCoeffs.mB0; o->b1 = f.mCoeffs.mB1; o->b2 = f.mCoeffs.mB2;
    o->a1 = f.mCoeffs.mA1; o->a2 = f.mCoeffs.mA2;
    o->tb0 = f.mTargetCoeffs.mB0; o->tb1 = f.mTargetCoeffs.mB1; o->tb2 = f.mTargetCoeffs.mB2;
    o->ta1 = f.mTargetCoeffs.mA1; o->ta2 = f.mTargetCoeffs.mA2;
    o->counter = f.mCounter;
}
void FromBiquad(oal_biquad const *o, BiquadInterpFilter &f)
{
    f.mZ1 = o->z1; f.mZ2 = o->z2;
    f.mCoeffs.mB0 = o->b0; f.mCoeffs.mB1 = o->b1; f.mCoeffs.mB2 = o->b2;
    f.mCoeffs.mA1 = o->a1; f.mCoeffs.mA2 = o->a2;
    f.mTargetCoeffs.mB0 = o->tb0; f.mTargetCoeffs.mB1 = o->tb1; f.mTargetCoeffs.mB2 = o->tb2;
    f.mTargetCoeffs.mA1 = o->ta1; f.mTargetCoeffs.mA2 = o->ta2;
    f.mCounter = o->counter;
}
void ToSplitter(BandSplitter const &b, oal_splitter *s)
{ s->coeff = b.mCoeff; s->lp_z1 = b.mLpZ1; s->lp_z2 = b.mLpZ2; s->ap_z1 = b.mApZ1; }
void FromSplitter(oal_splitter const *s, BandSplitter &b)
{ b.mCoeff = s->coeff; b.mLpZ1 = s->lp_z1; b.mLpZ2 = s->lp_z2; b.mApZ1 = s->ap_z1; }

struct Dev final : DeviceBase { Dev() : DeviceBase{DeviceType::Loopback} { } };
struct Ctx final : ContextBase { explicit Ctx(DeviceBase *d) : ContextBase{d} { } };
struct Item final : VoiceBufferItem { };

struct BufferData {
    std::vector<std::byte> bytes;
    Item item;
};

/* what a callback buffer's user function reads from: a byte stream that simply ends */
struct CbStream {
    std::vector<std::byte> bytes;
    size_t pos{0};
    unsigned calls{0};
};
int CbRead(void *user, void *data, int nbytes) noexcept
{
    auto *st = static_cast<CbStream*>(user);
    ++st->calls;
    auto const n = std::min(static_cast<size_t>(std::max(nbytes, 0)), st->bytes.size() - st->pos);
    std::memcpy(data, st->bytes.data() + st->pos, n);
    st->pos += n;
    return static_cast<int>(n);
}

} // namespace

struct oal_scene {
    oal_device_desc desc{};
    std::unique_ptr<Dev> dev;
    std::unique_ptr<Ctx> ctx;
    std::deque<BufferData> buffers;
    std::deque<Voice> voices;
    std::vector<int> vstate;
    std::vector<unsigned> buffersDone;     /* summed AsyncBufferCompleteEvent counts per voice */
    std::deque<EffectSlotBase> slots;
    std::deque<CbStream> streams;
    std::vector<int> streamOfVoice;        /* [voice] index into streams, -1 = not a callback source */
    HrtfStorePtr hrtf;
};

extern "C" {

const char *oal_kind(void) { return "reference"; }

void oal_set_simd(int enable) { gSimd = enable != 0; ApplySimd(); }

int oal_bsinc_table_get(int which, oal_bsinc_table *out)
{
    auto const *t = GetBsinc(which);
    if(!t) return -1;
    out->scaleBase = t->scaleBase.c_val;
    out->scaleRange = t->scaleRange.c_val;
    for(size_t i{0};i < 16;++i)
    {
        out->m[i] = t->m[i].c_val;
        out->filterOffset[i] = t->filterOffset[i].c_val;
    }
    out->tab = t->Tab.data();
    out->tablen = t->Tab.size();
    return 0;
}

int oal_cubic_table_get(int which, float *out)
{
    CubicTable const *t = (which == 0) ? static_cast<CubicTable const*>(&gSplineFilter)
        : static_cast<CubicTable const*>(&gGaussianFilter);
    for(size_t pi{0};pi < CubicPhaseCount;++pi)
    {
        std::copy_n(t->mTable[pi].mCoeffs.data(), 4, out + pi*8);
        std::copy_n(t->mTable[pi].mDeltas.data(), 4, out + pi*8 + 4);
    }
    return 0;
}

static ResamplerFunc PrepareFor(int resampler, uint32_t increment, InterpState *state)
{
    ApplySimd();
    return PrepareResampler(static_cast<Resampler>(resampler), increment, state);
}

void oal_prepare_resampler(int resampler, uint32_t increment, oal_interp_state *out)
{
    auto state = InterpState{};
    std::ignore = PrepareFor(resampler, increment, &state);
    std::memset(out, 0, sizeof(*out));
    auto const r = static_cast<Resampler>(resampler);
    switch(r)
    {
    case Resampler::Point: out->kind = 0; break;
    case Resampler::Linear: out->kind = 1; break;
    case Resampler::Spline: out->kind = 2; out->table = 0; break;
    case Resampler::Gaussian: out->kind = 2; out->table = 1; break;
    default:
        {
            auto const &bs = std::get<BsincState>(state);
            auto const which = (r == Resampler::FastBSinc12 || r == Resampler::BSinc12) ? 12
                : (r == Resampler::FastBSinc24 || r == Resampler::BSinc24) ? 24 : 48;
            auto const isfast = (r == Resampler::FastBSinc12 || r == Resampler::FastBSinc24
                || r == Resampler::FastBSinc48);
            out->kind = (!isfast && increment > MixerFracOne) ? 4 : 3;
            out->table = which;
            out->sf = bs.sf;
            out->m = bs.m.c_val;
            out->l = bs.l.c_val;
            out->filter_offset = static_cast<uint32_t>(bs.filter.data() - GetBsinc(which)->Tab.data());
        }
    }
}

void oal_resample(int resampler, uint32_t increment, const float *src, size_t srclen,
    uint32_t frac, float *dst, size_t n)
{
    auto state = InterpState{};
    auto const func = PrepareFor(resampler, increment, &state);
    auto const fpuctl = FPUCtl{};
    /* SSE variants want a 16-byte aligned destination (voice.cpp:636 keeps multiples of 4). */
    alignas(16) static thread_local std::array<float, 2048> tmp;
    func(&state, std::span{src, srclen}, frac, increment, std::span{tmp.data(), n});
    std::copy_n(tmp.data(), n, dst);
}

void oal_mix(const float *in, size_t n, float *out, size_t nlines, float *cur_gains,
    const float *target_gains, size_t counter, size_t outpos)
{
    ApplySimd();
    auto const fpuctl = FPUCtl{};
    auto lines = al::vector<FloatBufferLine,16>(nlines);
    for(size_t c{0};c < nlines;++c)
        std::copy_n(out + c*BufferLineSize, BufferLineSize, lines[c].data());
    alignas(16) std::array<float, BufferLineSize> inbuf{};
    std::copy_n(in, n, inbuf.data());
    MixSamples(std::span{inbuf}.first(n), std::span{lines}, std::span{cur_gains, nlines},
        std::span{target_gains, nlines}, counter, outpos);
    for(size_t c{0};c < nlines;++c)
        std::copy_n(lines[c].data(), BufferLineSize, out + c*BufferLineSize);
}

void oal_mix_one(const float *in, size_t n, float *out, float *cur_gain, float target_gain,
    size_t counter)
{
    ApplySimd();
    auto const fpuctl = FPUCtl{};
    alignas(16) std::array<float, BufferLineSize> inbuf{};
    alignas(16) std::array<float, BufferLineSize> outbuf{};
    std::copy_n(in, n, inbuf.data());
    std::copy_n(out, n, outbuf.data());
    MixSamples(std::span{inbuf}.first(n), std::span{outbuf}.first(n), *cur_gain, target_gain,
        counter);
    std::copy_n(outbuf.data(), n, out);
}

/* MixHrtf_/MixHrtfBlend_ are file-static selections in voice.cpp; call the variants directly. */
void oal_mix_hrtf(const float *in, float *accum, uint32_t irsize, const float *coeffs,
    const uint32_t delay[2], float gain, float gainstep, size_t n)
{
    EnsureInit();
    auto const fpuctl = FPUCtl{};
    alignas(16) HrirArray hc{};
    std::memcpy(hc.data(), coeffs, sizeof(hc));
    alignas(16) std::array<float, BufferLineSize+HrtfHistoryLength> inbuf{};
    std::copy_n(in, n+HrtfHistoryLength, inbuf.data());
    alignas(16) std::array<f32x2, BufferLineSize+HrirLength> acc{};
    std::memcpy(acc.data(), accum, sizeof(acc));
    auto const parms = MixHrtfFilter{hc, {delay[0], delay[1]}, gain, gainstep};
    if(gSimd) MixHrtf_SSE(inbuf, acc, irsize, &parms, n);
    else MixHrtf_C(inbuf, acc, irsize, &parms, n);
    std::memcpy(accum, acc.data(), sizeof(acc));
}

void oal_mix_hrtf_blend(const float *in, float *accum, uint32_t irsize, const float *oldcoeffs,
    const uint32_t olddelay[2], float oldgain, const float *newcoeffs,
    const uint32_t newdelay[2], float newgainstep, size_t n)
{
    EnsureInit();
    auto const fpuctl = FPUCtl{};
    auto oldf = HrtfFilter{};
    std::memcpy(oldf.Coeffs.data(), oldcoeffs, sizeof(oldf.Coeffs));
    oldf.Delay = {olddelay[0], olddelay[1]};
    oldf.Gain = oldgain;
    alignas(16) HrirArray hc{};
    std::memcpy(hc.data(), newcoeffs, sizeof(hc));
    alignas(16) std::array<float, BufferLineSize+HrtfHistoryLength> inbuf{};
    std::copy_n(in, n+HrtfHistoryLength, inbuf.data());
    alignas(16) std::array<f32x2, BufferLineSize+HrirLength> acc{};
    std::memcpy(acc.data(), accum, sizeof(acc));
    auto const parms = MixHrtfFilter{hc, {newdelay[0], newdelay[1]}, 0.0f, newgainstep};
    if(gSimd) MixHrtfBlend_SSE(inbuf, acc, irsize, &oldf, &parms, n);
    else MixHrtfBlend_C(inbuf, acc, irsize, &oldf, &parms, n);
    std::memcpy(accum, acc.data(), sizeof(acc));
}

void oal_splitter_init(oal_splitter *s, float f0norm)
{ auto b = BandSplitter{}; b.init(f0norm); ToSplitter(b, s); }

void oal_splitter_process_hfscale(oal_splitter *s, const float *in, float *out, size_t n,
    float hfscale)
{
    auto const fpuctl = FPUCtl{};
    auto b = BandSplitter{}; FromSplitter(s, b);
    b.processHfScale(std::span{in, n}, std::span{out, n}, hfscale);
    ToSplitter(b, s);
}

void oal_splitter_process_scale(oal_splitter *s, float *samples, size_t n, float hfscale,
    float lfscale)
{
    auto const fpuctl = FPUCtl{};
    auto b = BandSplitter{}; FromSplitter(s, b);
    b.processScale(std::span{samples, n}, hfscale, lfscale);
    ToSplitter(b, s);
}

void oal_mix_direct_hrtf(float *left, float *right, const float *in, size_t nch, float *accum,
    oal_splitter *splitters, const float *hfscales, const float *chan_coeffs, size_t irsize,
    size_t n)
{
    EnsureInit();
    auto const fpuctl = FPUCtl{};
    auto lines = al::vector<FloatBufferLine,16>(nch+2);
    for(size_t c{0};c < nch;++c)
        std::copy_n(in + c*BufferLineSize, BufferLineSize, lines[c].data());
    std::copy_n(left, BufferLineSize, lines[nch].data());
    std::copy_n(right, BufferLineSize, lines[nch+1].data());
    auto state = DirectHrtfState::Create(nch);
    for(size_t c{0};c < nch;++c)
    {
        FromSplitter(&splitters[c], state->mChannels[c].mSplitter);
        state->mChannels[c].mHfScale = hfscales[c];
        std::memcpy(state->mChannels[c].mCoeffs.data(), chan_coeffs + c*HrirLength*2,
            sizeof(HrirArray));
    }
    alignas(16) std::array<f32x2, BufferLineSize+HrirLength> acc{};
    std::memcpy(acc.data(), accum, sizeof(acc));
    auto const ins = std::span<FloatBufferLine const>{lines.data(), nch};
    if(gSimd)
        MixDirectHrtf_SSE(lines[nch], lines[nch+1], ins, acc, state->mTemp, state->mChannels,
            irsize, n);
    else
        MixDirectHrtf_C(lines[nch], lines[nch+1], ins, acc, state->mTemp, state->mChannels,
            irsize, n);
    std::memcpy(accum, acc.data(), sizeof(acc));
    std::copy_n(lines[nch].data(), BufferLineSize, left);
    std::copy_n(lines[nch+1].data(), BufferLineSize, right);
    for(size_t c{0};c < nch;++c)
        ToSplitter(state->mChannels[c].mSplitter, &splitters[c]);
}

void oal_biquad_reset(oal_biquad *f) { auto b = BiquadInterpFilter{}; ToBiquad(b, f); }
void oal_biquad_clear(oal_biquad *f)
{ auto b = BiquadInterpFilter{}; FromBiquad(f, b); b.clear(); ToBiquad(b, f); }
void oal_biquad_set_params_from_slope(oal_biquad *f, int type, float f0norm, float gain,
    float slope)
{
    auto b = BiquadInterpFilter{}; FromBiquad(f, b);
    b.setParamsFromSlope(static_cast<BiquadType>(type), f0norm, gain, slope);
    ToBiquad(b, f);
}
void oal_biquad_dual_process(oal_biquad *f0, oal_biquad *f1, const float *src, float *dst,
    size_t n)
{
    auto const fpuctl = FPUCtl{};
    auto b0 = BiquadInterpFilter{}; FromBiquad(f0, b0);
    auto b1 = BiquadInterpFilter{}; FromBiquad(f1, b1);
    DualBiquadInterp{b0, b1}.process(std::span{src, n}, std::span{dst, n});
    ToBiquad(b0, f0); ToBiquad(b1, f1);
}

int oal_hrtf_load(const char *path)
{
    auto f = std::ifstream{path, std::ios::binary};
    if(!f.is_open()) return -1;
    try {
        auto store = LoadHrtf(f);
        if(!store) return -2;
        gHrtfOwner = std::move(store);
        gHrtf = gHrtfOwner.get();
    }
    catch(...) { return -3; }
    return 0;
}

/* GetLoadedHrtf (core/hrtf.cpp:471-620) as it is, on a data set found the way the reference finds one:
 * EnumerateHrtf lists the .mhr files under ALSOFT_LOCAL_PATH (`dir`), GetLoadedHrtf loads the first and -- when the
 * set's rate differs from devrate -- resamples every HRIR with PPhaseResampler and rescales delays and IrSize
 * (:539-606).  The loaded store becomes the harness's current one (oal_hrtf_raw reads it out). */
int oal_hrtf_load_for_rate(const char *dir, uint32_t devrate)
{
    setenv("ALSOFT_LOCAL_PATH", dir, 1);
    try {
        auto const names = EnumerateHrtf(std::nullopt);
        if(names.empty()) return -1;
        gHrtfLoaded = GetLoadedHrtf(names[0], devrate);
        if(!gHrtfLoaded) return -2;
        gHrtfOwner.reset();
        gHrtf = gHrtfLoaded.get();
    }
    catch(...) { return -3; }
    return 0;
}

/* DirectHrtfState::build (core/hrtf.cpp:266-366) on the current store, with the caller's AmbiPoints / AmbiMatrix /
 * AmbiOrderHFGain (alc/panning.cpp:861-1038 holds the reference's own): points = npoints x {elevation, azimuth}
 * radians, matrix = npoints x 16 (the first 16 of a ChannelCoeffs row), out_coeffs = nchans x 128 x 2, out_hfscale = nchans. */
int oal_direct_hrtf_build(uint32_t irsize, int per_hrir_min, const float *points, const float *matrix, uint32_t npoints,
    uint32_t nchans, float xover_freq, const float *order_hf_gain, float *out_coeffs, float *out_hfscale, uint32_t *out_irsize)
{
    if(!gHrtf || nchans < 1 || nchans > 16) return -1;
    auto pts = std::vector<AngularPoint>{};
    auto mat = std::vector<std::array<float, MaxAmbiChannels>>(npoints);
    for(uint32_t i{0}; i < npoints; ++i)
    {
        pts.push_back(AngularPoint{EvRadians{points[i*2]}, AzRadians{points[i*2 + 1]}});
        mat[i].fill(0.0f);
        for(size_t c{0}; c < 16; ++c) mat[i][c] = matrix[i*16 + c];      /* rows of 16: orders up to 3 */
    }
    auto gains = std::array<float, MaxAmbiOrder+1>{};
    for(size_t o{0}; o <= MaxAmbiOrder; ++o) gains[o] = order_hf_gain[o];
    auto state = DirectHrtfState::Create(nchans);
    state->build(gHrtf, irsize, per_hrir_min != 0, pts, mat, xover_freq, gains);
    for(uint32_t c{0}; c < nchans; ++c)
    {
        out_hfscale[c] = state->mChannels[c].mHfScale;
        for(size_t k{0}; k < HrirLength; ++k)
        {
            out_coeffs[(c*HrirLength + k)*2 + 0] = state->mChannels[c].mCoeffs[k][0];
            out_coeffs[(c*HrirLength + k)*2 + 1] = state->mChannels[c].mCoeffs[k][1];
        }
    }
    *out_irsize = state->mIrSize;
    return 0;
}

int oal_hrtf_info_get(oal_hrtf_info *out)
{
    if(!gHrtf) return -1;
    out->sample_rate = gHrtf->mSampleRate;
    out->ir_size = gHrtf->mIrSize;
    out->num_fields = static_cast<uint32_t>(gHrtf->mFields.size());
    out->num_elevs = static_cast<uint32_t>(gHrtf->mElev.size());
    out->num_irs = static_cast<uint32_t>(gHrtf->mCoeffs.size());
    return 0;
}

int oal_hrtf_raw(float *field_distance, uint8_t *field_evcount, uint16_t *elev_azcount,
    uint16_t *elev_iroffset, float *coeffs, uint8_t *delays)
{
    if(!gHrtf) return -1;
    for(size_t i{0};i < gHrtf->mFields.size();++i)
    {
        field_distance[i] = gHrtf->mFields[i].distance;
        field_evcount[i] = gHrtf->mFields[i].evCount.c_val;
    }
    for(size_t i{0};i < gHrtf->mElev.size();++i)
    {
        elev_azcount[i] = gHrtf->mElev[i].azCount.c_val;
        elev_iroffset[i] = gHrtf->mElev[i].irOffset.c_val;
    }
    std::memcpy(coeffs, gHrtf->mCoeffs.data(), gHrtf->mCoeffs.size()*sizeof(HrirArray));
    for(size_t i{0};i < gHrtf->mDelays.size();++i)
    {
        delays[i*2+0] = gHrtf->mDelays[i][0].c_val;
        delays[i*2+1] = gHrtf->mDelays[i][1].c_val;
    }
    return 0;
}

void oal_hrtf_get_coeffs(float elevation, float azimuth, float distance, float spread,
    float *coeffs, uint32_t delays[2])
{
    auto const fpuctl = FPUCtl{};
    alignas(16) HrirArray hc{};
    auto d = std::array<unsigned,2>{};
    gHrtf->getCoeffs(elevation, azimuth, distance, spread, hc, d);
    std::memcpy(coeffs, hc.data(), sizeof(hc));
    delays[0] = d[0]; delays[1] = d[1];
}


oal_scene *oal_scene_create(const oal_device_desc *desc)
{
    ApplySimd();
    auto s = std::make_unique<oal_scene>();
    s->desc = *desc;
    s->dev = std::make_unique<Dev>();
    auto &dev = *s->dev;
    dev.mSampleRate = desc->sample_rate;
    dev.mUpdateSize = BufferLineSize;
    dev.mBufferSize = BufferLineSize;
    dev.FmtType = DevFmtFloat;
    dev.NumAuxSends = desc->num_aux_sends;
    dev.MixBuffer.resize(desc->num_dry_channels + desc->num_real_channels);
    dev.Dry.Buffer = std::span{dev.MixBuffer}.first(desc->num_dry_channels);
    dev.RealOut.Buffer = desc->num_real_channels
        ? std::span{dev.MixBuffer}.subspan(desc->num_dry_channels) : dev.Dry.Buffer;
    if(desc->hrtf)
    {
        if(!gHrtf) return nullptr;
        dev.mRenderMode = RenderMode::Hrtf;
        dev.mIrSize = gHrtf->mIrSize;
        dev.RealOut.ChannelIndex[FrontLeft] = 0_u8;
        dev.RealOut.ChannelIndex[FrontRight] = 1_u8;
        auto state = DirectHrtfState::Create(desc->num_dry_channels);
        state->mIrSize = gHrtf->mIrSize;
        dev.mPostProcess.emplace<HrtfPostProcess>(HrtfPostProcess{std::move(state)});
    }
    s->ctx = std::make_unique<Ctx>(s->dev.get());
    s->ctx->mEnabledEvts.store({}, std::memory_order_relaxed);
    s->ctx->mAsyncEvents = FifoBuffer<AsyncEvent>::Create(64, false);
    {
        auto evts = ContextBase::AsyncEventBitset{};
        evts.set(AsyncEnableBits::BufferCompleted);
        s->ctx->mEnabledEvts.store(evts, std::memory_order_relaxed);
    }
    for(uint32_t i{0};i < desc->num_slots;++i)
    {
        auto &slot = s->slots.emplace_back();
        slot.mWetBuffer.resize(desc->wet_channels);
        slot.Wet.Buffer = slot.mWetBuffer;
        slot.EffectType = EffectSlotType::Reverb;
    }
    return s.release();
}

void oal_scene_destroy(oal_scene *s) { delete s; }

int oal_scene_add_buffer(oal_scene *s, const void *data, int fmt_type, uint32_t frame_step,
    uint32_t sample_len, uint32_t loop_start, uint32_t loop_end)
{
    static constexpr std::array<size_t,7> bps{1, 2, 4, 4, 8, 1, 1};
    auto const nbytes = size_t{sample_len} * frame_step * bps.at(static_cast<size_t>(fmt_type));
    auto &b = s->buffers.emplace_back();
    b.bytes.resize(nbytes + 16);
    std::memcpy(b.bytes.data(), data, nbytes);
    auto const count = size_t{sample_len} * frame_step;
    auto *p = b.bytes.data();
    /* NOLINTBEGIN */
    switch(fmt_type)
    {
    case OAL_FMT_UBYTE: b.item.mSamples = std::span{reinterpret_cast<u8*>(p), count}; break;
    case OAL_FMT_SHORT: b.item.mSamples = std::span{reinterpret_cast<i16*>(p), count}; break;
    case OAL_FMT_INT: b.item.mSamples = std::span{reinterpret_cast<i32*>(p), count}; break;
    case OAL_FMT_FLOAT: b.item.mSamples = std::span{reinterpret_cast<f32*>(p), count}; break;
    case OAL_FMT_DOUBLE: b.item.mSamples = std::span{reinterpret_cast<f64*>(p), count}; break;
    case OAL_FMT_MULAW: b.item.mSamples = std::span{reinterpret_cast<MulawSample*>(p), count}; break;
    case OAL_FMT_ALAW: b.item.mSamples = std::span{reinterpret_cast<AlawSample*>(p), count}; break;
    default: s->buffers.pop_back(); return -1;
    }
    /* NOLINTEND */
    b.item.mBlockAlign = 1;
    b.item.mSampleLen = sample_len;
    b.item.mLoopStart = loop_start;
    b.item.mLoopEnd = loop_end;
    /* stash the frame step where add_voice can find it */
    b.bytes.back() = static_cast<std::byte>(frame_step);
    b.bytes[b.bytes.size()-2] = static_cast<std::byte>(fmt_type);
    return static_cast<int>(s->buffers.size()-1);
}

/* IMA4 / MS ADPCM data as the buffer layer hands it to the voices (al/buffer.cpp LoadData:
 * mBlockAlign = samples per block, mSamples = the compressed bytes) */
int oal_scene_add_buffer_adpcm(oal_scene *s, const void *data, int adpcm_type, uint32_t channels,
    uint32_t samples_per_block, uint32_t sample_len, uint32_t loop_start, uint32_t loop_end)
{
    auto const nblocks = size_t{(sample_len + samples_per_block - 1u) / samples_per_block};
    auto const block_bytes = adpcm_type == 1 ? size_t{(samples_per_block-2u)/2u + 7u}*channels
        : size_t{(samples_per_block-1u)/2u + 4u}*channels;
    auto const nbytes = nblocks * block_bytes;
    auto &b = s->buffers.emplace_back();
    b.bytes.resize(nbytes + 16);
    std::memcpy(b.bytes.data(), data, nbytes);
    auto *p = b.bytes.data();
    /* NOLINTBEGIN */
    if(adpcm_type == 1) b.item.mSamples = std::span{reinterpret_cast<MSADPCMData*>(p), nbytes};
    else b.item.mSamples = std::span{reinterpret_cast<IMA4Data*>(p), nbytes};
    /* NOLINTEND */
    b.item.mBlockAlign = samples_per_block;
    b.item.mSampleLen = sample_len;
    b.item.mLoopStart = loop_start;
    b.item.mLoopEnd = loop_end;
    b.bytes.back() = static_cast<std::byte>(channels);
    b.bytes[b.bytes.size()-2] = static_cast<std::byte>(7 + adpcm_type);
    return static_cast<int>(s->buffers.size()-1);
}

/* alSourceQueueBuffers' linking (al/source.cpp): `next` plays after `buffer`; next < 0 ends the queue */
int oal_scene_link_buffers(oal_scene *s, int buffer, int next)
{
    auto &b = s->buffers.at(static_cast<size_t>(buffer));
    b.item.mNext.store(next < 0 ? nullptr : &s->buffers.at(static_cast<size_t>(next)).item,
        std::memory_order_relaxed);
    return 0;
}

/* a streaming voice: not VoiceFlag::IsStatic, so Voice::mix loads through LoadBufferQueue
 * (voice.cpp:563-594) and walks mCurrentBuffer along mNext (voice.cpp:1182-1194) */
int oal_scene_add_queue_voice(oal_scene *s, const oal_voice_desc *desc)
{
    auto const vi = oal_scene_add_voice(s, desc);
    if(vi < 0) return vi;
    s->voices.at(static_cast<size_t>(vi)).mFlags.reset(VoiceFlag::IsStatic);
    return vi;
}

/* a callback source (AL_SOFT_callback_buffer): a buffer whose mCallback reads `stream` (al/buffer.cpp PrepareCallback:
 * storage of MixerLineSize*MaxPitch + MaxResamplerEdge frames, mSampleLen 0) and a voice with VoiceFlag::IsCallback */
int oal_scene_add_callback_voice(oal_scene *s, const void *stream, size_t stream_bytes, int fmt_type,
    uint32_t position_frac, uint32_t frequency)
{
    static constexpr std::array<size_t,7> bps{1, 2, 4, 4, 8, 1, 1};
    constexpr auto line_size = size_t{DeviceBase::MixerLineSize}*MaxPitch + MaxResamplerEdge;
    auto &st = s->streams.emplace_back();
    st.bytes.assign(static_cast<const std::byte*>(stream), static_cast<const std::byte*>(stream) + stream_bytes);
    auto const dummy = std::vector<std::byte>(bps.at(static_cast<size_t>(fmt_type)));
    auto const bi = oal_scene_add_buffer(s, dummy.data(), fmt_type, 1u, 1u, 0u, 0u);
    if(bi < 0) return -1;
    auto &b = s->buffers.at(static_cast<size_t>(bi));
    b.bytes.assign(line_size*bps.at(static_cast<size_t>(fmt_type)) + 16, std::byte{});
    auto *p = b.bytes.data();
    /* NOLINTBEGIN */
    switch(fmt_type)
    {
    case OAL_FMT_UBYTE: b.item.mSamples = std::span{reinterpret_cast<u8*>(p), line_size}; break;
    case OAL_FMT_SHORT: b.item.mSamples = std::span{reinterpret_cast<i16*>(p), line_size}; break;
    case OAL_FMT_INT: b.item.mSamples = std::span{reinterpret_cast<i32*>(p), line_size}; break;
    case OAL_FMT_FLOAT: b.item.mSamples = std::span{reinterpret_cast<f32*>(p), line_size}; break;
    case OAL_FMT_DOUBLE: b.item.mSamples = std::span{reinterpret_cast<f64*>(p), line_size}; break;
    case OAL_FMT_MULAW: b.item.mSamples = std::span{reinterpret_cast<MulawSample*>(p), line_size}; break;
    default: b.item.mSamples = std::span{reinterpret_cast<AlawSample*>(p), line_size}; break;
    }
    /* NOLINTEND */
    b.bytes.back() = static_cast<std::byte>(1);
    b.bytes[b.bytes.size()-2] = static_cast<std::byte>(fmt_type);
    b.item.mSampleLen = 0;
    b.item.mCallback = CbRead;
    b.item.mUserData = &st;
    auto const desc = oal_voice_desc{bi, 0, 0, position_frac, frequency};
    auto const vi = oal_scene_add_voice(s, &desc);
    if(vi < 0) return vi;
    auto &v = s->voices.at(static_cast<size_t>(vi));
    v.mFlags.reset(VoiceFlag::IsStatic);
    v.mFlags.set(VoiceFlag::IsCallback);
    s->streamOfVoice.resize(s->voices.size(), -1);
    s->streamOfVoice[static_cast<size_t>(vi)] = static_cast<int>(s->streams.size()) - 1;
    return vi;
}

/* mNumCallbackBlocks, mCallbackBlockOffset, CallbackStopped, and how often the user function was called */
int oal_scene_callback_state(oal_scene *s, int voice, uint32_t out[4])
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    out[0] = v.mNumCallbackBlocks; out[1] = v.mCallbackBlockOffset;
    out[2] = v.mFlags.test(VoiceFlag::CallbackStopped) ? 1u : 0u;
    out[3] = 0u;
    if(static_cast<size_t>(voice) < s->streamOfVoice.size() && s->streamOfVoice[static_cast<size_t>(voice)] >= 0)
        out[3] = s->streams[static_cast<size_t>(s->streamOfVoice[static_cast<size_t>(voice)])].calls;
    return 0;
}

unsigned oal_scene_voice_buffers_done(oal_scene *s, int voice)
{ return static_cast<size_t>(voice) < s->buffersDone.size() ? s->buffersDone[static_cast<size_t>(voice)] : 0u; }

/* index of the voice's mCurrentBuffer in the scene's buffer list, -1 = none */
int oal_scene_voice_current_buffer(oal_scene *s, int voice)
{
    auto const *cur = s->voices.at(static_cast<size_t>(voice)).mCurrentBuffer.load();
    for(size_t i{0};i < s->buffers.size();++i)
        if(&s->buffers[i].item == cur) return static_cast<int>(i);
    return -1;
}

int oal_scene_add_voice(oal_scene *s, const oal_voice_desc *desc)
{
    static constexpr std::array<unsigned,9> bps{1, 2, 4, 4, 8, 1, 1, 1, 1};
    auto &buf = s->buffers.at(static_cast<size_t>(desc->buffer));
    auto const frame_step = static_cast<unsigned>(buf.bytes.back());
    auto const fmt = static_cast<size_t>(buf.bytes[buf.bytes.size()-2]);
    auto &v = s->voices.emplace_back();
    s->vstate.push_back(OAL_VOICE_PLAYING);
    /* InitVoice, al/source.cpp:639-670 */
    v.mLoopBuffer.store(desc->looping ? &buf.item : nullptr, std::memory_order_relaxed);
    v.mFmtChannels = FmtMono;
    v.mFrequency = desc->frequency;
    v.mFrameStep = frame_step;
    v.mBytesPerBlock = frame_step * bps.at(fmt);
    v.mSamplesPerBlock = 1;
    if(fmt >= 7)
    {   /* InitVoice: BufferStorage::blockSizeFromFmt / mBlockAlign */
        v.mSamplesPerBlock = buf.item.mBlockAlign;
        v.mBytesPerBlock = (fmt == 8 ? (buf.item.mBlockAlign-2u)/2u + 7u : (buf.item.mBlockAlign-1u)/2u + 4u)
            * frame_step;
    }
    v.mAmbiOrder = 0;
    v.mFlags.reset();
    v.mFlags.set(VoiceFlag::IsStatic);
    v.mNumCallbackBlocks = 0;
    v.mCallbackBlockOffset = 0;
    v.prepare(s->dev.get());
    v.mPosition.store(desc->position, std::memory_order_relaxed);
    v.mPositionFrac.store(desc->position_frac, std::memory_order_relaxed);
    v.mCurrentBuffer.store(&buf.item, std::memory_order_relaxed);
    v.mStartTime = {};
    v.mSourceID.store(static_cast<unsigned>(s->voices.size()), std::memory_order_relaxed);
    v.mPlayState.store(Voice::Playing, std::memory_order_relaxed);
    if(s->desc.hrtf) v.mFlags.set(VoiceFlag::HasHrtf);
    v.mDuplicateMono = false;
    return static_cast<int>(s->voices.size()-1);
}

int oal_scene_set_voice_params(oal_scene *s, int voice, const oal_voice_params *p)
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    auto &dev = *s->dev;
    auto const fpuctl = FPUCtl{};
    /* CalcNonAttnVoiceParams, alc/alu.cpp:1664-1686 */
    v.mDirect.Buffer = dev.Dry.Buffer;
    for(size_t i{0};i < dev.NumAuxSends;++i)
    {
        if(p->send_slot[i] < 0) v.mSend[i].Buffer = {};
        else v.mSend[i].Buffer = s->slots.at(static_cast<size_t>(p->send_slot[i])).Wet.Buffer;
    }
    v.mStep = p->step;
    v.mResampler = PrepareResampler(static_cast<Resampler>(p->resampler), v.mStep,
        &v.mResampleState);

    auto &chan = v.mChans[0];
    if(s->desc.hrtf)
    {
        /* CalcPanningAndFilters HRTF branch, alc/alu.cpp:1600-1611; CalcHrtfPanning :1207-1217 */
        v.mDirect.Buffer = dev.RealOut.Buffer;
        dev.mHrtf = nullptr;
        gHrtf->getCoeffs(p->hrtf_ev, p->hrtf_az, p->hrtf_dist, p->hrtf_spread,
            chan.mDryParams.Hrtf.Target.Coeffs, chan.mDryParams.Hrtf.Target.Delay);
        chan.mDryParams.Hrtf.Target.Gain = p->hrtf_gain;
        v.mFlags.set(VoiceFlag::HasHrtf);
    }
    else
        std::copy_n(p->dry_gains, MaxOutputChannels, chan.mDryParams.Gains.Target.begin());
    for(size_t i{0};i < dev.NumAuxSends;++i)
        std::copy_n(p->send_gains[i], MaxAmbiChannels, chan.mWetParams[i].Gains.Target.begin());

    /* alc/alu.cpp:1619-1656 */
    v.mDirect.FilterActive = p->direct_filter.active != 0;
    chan.mDryParams.LowPass.setParamsFromSlope(BiquadType::HighShelf, p->direct_filter.hf_norm,
        p->direct_filter.gain_hf, 1.0f);
    chan.mDryParams.HighPass.setParamsFromSlope(BiquadType::LowShelf, p->direct_filter.lf_norm,
        p->direct_filter.gain_lf, 1.0f);
    for(size_t i{0};i < dev.NumAuxSends;++i)
    {
        v.mSend[i].FilterActive = p->send_filter[i].active != 0;
        chan.mWetParams[i].LowPass.setParamsFromSlope(BiquadType::HighShelf,
            p->send_filter[i].hf_norm, p->send_filter[i].gain_hf, 1.0f);
        chan.mWetParams[i].HighPass.setParamsFromSlope(BiquadType::LowShelf,
            p->send_filter[i].lf_norm, p->send_filter[i].gain_lf, 1.0f);
    }
    return 0;
}

int oal_scene_add_voice_multi(oal_scene *s, const oal_voice_desc *desc, uint32_t num_channels)
{
    auto const vi = oal_scene_add_voice(s, desc);
    auto &v = s->voices.at(static_cast<size_t>(vi));
    if(num_channels == 2 && v.mFrameStep == 2)
    {   /* a plain stereo source: two ChannelData, no ambisonic handling */
        v.mFmtChannels = FmtStereo;
        auto const pos = v.mPosition.load(std::memory_order_relaxed);
        auto const frac = v.mPositionFrac.load(std::memory_order_relaxed);
        v.prepare(s->dev.get());
        v.mPosition.store(pos, std::memory_order_relaxed);
        v.mPositionFrac.store(frac, std::memory_order_relaxed);
        return v.mChans.size() == 2 ? vi : -1;
    }
    if(num_channels != 4 || v.mFrameStep != 4) return -1;
    /* a first-order B-Format source; the device has to be at least first order for prepare()
     * to keep all four channels (voice.cpp:1247-1249) */
    if(s->dev->mAmbiOrder < 1) s->dev->mAmbiOrder = 1;
    v.mFmtChannels = FmtBFormat3D;
    v.mAmbiOrder = 1;
    auto const pos = v.mPosition.load(std::memory_order_relaxed);
    auto const frac = v.mPositionFrac.load(std::memory_order_relaxed);
    v.prepare(s->dev.get());
    v.mPosition.store(pos, std::memory_order_relaxed);
    v.mPositionFrac.store(frac, std::memory_order_relaxed);
    if(v.mChans.size() != 4) return -1;
    return vi;
}

int oal_scene_set_channel_params(oal_scene *s, int voice, uint32_t channel, const oal_voice_params *p)
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    auto &dev = *s->dev;
    auto const fpuctl = FPUCtl{};
    if(channel >= v.mChans.size() || s->desc.hrtf) return -1;
    v.mDirect.Buffer = dev.Dry.Buffer;
    for(size_t i{0};i < dev.NumAuxSends;++i)
    {
        if(p->send_slot[i] < 0) v.mSend[i].Buffer = {};
        else v.mSend[i].Buffer = s->slots.at(static_cast<size_t>(p->send_slot[i])).Wet.Buffer;
    }
    v.mStep = p->step;
    v.mResampler = PrepareResampler(static_cast<Resampler>(p->resampler), v.mStep, &v.mResampleState);
    auto &chan = v.mChans[channel];
    std::copy_n(p->dry_gains, MaxOutputChannels, chan.mDryParams.Gains.Target.begin());
    for(size_t i{0};i < dev.NumAuxSends;++i)
        std::copy_n(p->send_gains[i], MaxAmbiChannels, chan.mWetParams[i].Gains.Target.begin());
    v.mDirect.FilterActive = p->direct_filter.active != 0;
    chan.mDryParams.LowPass.setParamsFromSlope(BiquadType::HighShelf, p->direct_filter.hf_norm,
        p->direct_filter.gain_hf, 1.0f);
    chan.mDryParams.HighPass.setParamsFromSlope(BiquadType::LowShelf, p->direct_filter.lf_norm,
        p->direct_filter.gain_lf, 1.0f);
    for(size_t i{0};i < dev.NumAuxSends;++i)
    {
        v.mSend[i].FilterActive = p->send_filter[i].active != 0;
        chan.mWetParams[i].LowPass.setParamsFromSlope(BiquadType::HighShelf,
            p->send_filter[i].hf_norm, p->send_filter[i].gain_hf, 1.0f);
        chan.mWetParams[i].HighPass.setParamsFromSlope(BiquadType::LowShelf,
            p->send_filter[i].lf_norm, p->send_filter[i].gain_lf, 1.0f);
    }
    return 0;
}

int oal_scene_set_channel_ambi_scale(oal_scene *s, int voice, uint32_t channel, float xover_norm,
    float hf_scale, float lf_scale)
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    if(channel >= v.mChans.size()) return -1;
    auto &chan = v.mChans[channel];
    chan.mAmbiSplitter.init(xover_norm);
    chan.mAmbiHFScale = hf_scale;
    chan.mAmbiLFScale = lf_scale;
    v.mFlags.set(VoiceFlag::IsAmbisonic);
    return 0;
}

int oal_scene_set_nfc(oal_scene *s, float w1, const uint32_t channels_per_order[5])
{
    auto &dev = *s->dev;
    dev.AvgSpeakerDist = 1.0f;          /* > 0: only gates the parameter side (alu.cpp:919) */
    dev.mNFCtrlFilter.init(w1);
    for(size_t i{0};i < 5;++i) dev.NumChannelsPerOrder[i] = channels_per_order[i];
    return 0;
}

int oal_scene_set_voice_nfc(oal_scene *s, int voice, float w0)
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    auto const fpuctl = FPUCtl{};
    v.mChans[0].mDryParams.NFCtrlFilter.adjust(w0);
    v.mFlags.set(VoiceFlag::HasNfc);
    return 0;
}

/* Voice::mStartTime = device clock + `samples` output samples (delayed start, voice.cpp:1023-1046) */
int oal_scene_set_voice_start_delay(oal_scene *s, int voice, uint32_t samples)
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    auto const ns = std::chrono::nanoseconds{static_cast<long long>(
        std::llround(double(samples) * 1.0e9 / double(s->dev->mSampleRate)))};
    v.mStartTime = s->dev->getClockTime() + ns;
    return 0;
}

int oal_scene_set_voice_state(oal_scene *s, int voice, int vstate)
{
    s->vstate.at(static_cast<size_t>(voice)) = vstate;
    auto &v = s->voices.at(static_cast<size_t>(voice));
    v.mPlayState.store(static_cast<Voice::State>(vstate), std::memory_order_relaxed);
    return 0;
}

int oal_scene_mix(oal_scene *s, uint32_t samples_to_do, int post_process)
{
    auto &dev = *s->dev;
    auto const fpuctl = FPUCtl{};
    /* DeviceBase::renderSamples(unsigned), alc/alu.cpp:2412-2443 */
    for(auto &line : dev.MixBuffer) line.fill(0.0f);
    /* ProcessContexts, alc/alu.cpp:2196-2206 */
    for(auto &slot : s->slots)
        for(auto &line : slot.mWetBuffer) line.fill(0.0f);
    auto const curtime = dev.getClockTime();
    for(auto &v : s->voices)
    {
        auto const vstate = v.mPlayState.load(std::memory_order_acquire);
        if(vstate != Voice::Stopped && vstate != Voice::Pending)
            v.mix(vstate, s->ctx.get(), curtime, samples_to_do);
        /* what the event thread would receive (voice.cpp:1207-1218) */
        auto *ring = s->ctx->mAsyncEvents.get();
        auto const evts = ring->getReadVector();
        auto count = 0_uz;
        for(auto const &part : evts)
        {
            for(auto &evt : part)
            {
                if(auto const *done = std::get_if<AsyncBufferCompleteEvent>(&evt))
                {
                    if(s->buffersDone.size() < s->voices.size()) s->buffersDone.resize(s->voices.size());
                    s->buffersDone.at(done->mId - 1u) += done->mCount;
                }
                ++count;
            }
        }
        if(count) ring->readAdvance(count);
    }
    if(post_process && s->desc.hrtf)
    {
        auto &proc = std::get<HrtfPostProcess>(dev.mPostProcess);
        dev.Process(proc, samples_to_do);
    }
    /* the device clock moves on as in DeviceBase::renderSamples (alc/alu.cpp:2426-2435) */
    {
        auto const samplesDone = dev.mSamplesDone.load(std::memory_order_relaxed) + samples_to_do;
        auto const clockBaseSec = dev.mClockBaseSec.load(std::memory_order_relaxed)
            + DeviceBase::seconds32{samplesDone/dev.mSampleRate};
        dev.mSamplesDone.store(samplesDone%dev.mSampleRate, std::memory_order_relaxed);
        dev.mClockBaseSec.store(clockBaseSec, std::memory_order_relaxed);
    }
    return 0;
}

int oal_scene_post_process(oal_scene *s, uint32_t samples_to_do)
{
    if(!s->desc.hrtf) return -1;
    auto &dev = *s->dev;
    auto const fpuctl = FPUCtl{};
    auto &proc = std::get<HrtfPostProcess>(dev.mPostProcess);
    dev.Process(proc, samples_to_do);
    return 0;
}

const float *oal_scene_dry(oal_scene *s) { return s->dev->MixBuffer[0].data(); }
const float *oal_scene_wet(oal_scene *s, int slot)
{ return s->slots.at(static_cast<size_t>(slot)).mWetBuffer[0].data(); }
const float *oal_scene_hrtf_accum(oal_scene *s) { return s->dev->HrtfAccumData[0].data(); }

int oal_scene_voice_state(oal_scene *s, int voice, oal_voice_state *out)
{
    auto &v = s->voices.at(static_cast<size_t>(voice));
    auto &chan = v.mChans[0];
    std::memset(out, 0, sizeof(*out));
    out->play_state = static_cast<int>(v.mPlayState.load());
    out->position = v.mPosition.load();
    out->position_frac = v.mPositionFrac.load();
    out->has_buffer = v.mCurrentBuffer.load() != nullptr;
    out->fading = v.mFlags.test(VoiceFlag::IsFading);
    std::copy_n(v.mPrevSamples[0].data(), MaxResamplerPadding, out->prev_samples);
    std::copy_n(chan.mDryParams.Gains.Current.data(), MaxOutputChannels, out->dry_current);
    out->hrtf_old_gain = chan.mDryParams.Hrtf.Old.Gain;
    out->hrtf_old_delay[0] = chan.mDryParams.Hrtf.Old.Delay[0];
    out->hrtf_old_delay[1] = chan.mDryParams.Hrtf.Old.Delay[1];
    std::copy_n(chan.mDryParams.Hrtf.History.data(), HrtfHistoryLength, out->hrtf_history);
    ToBiquad(chan.mDryParams.LowPass, &out->direct_lp);
    ToBiquad(chan.mDryParams.HighPass, &out->direct_hp);
    for(size_t i{0};i < MaxSendCount;++i)
    {
        std::copy_n(chan.mWetParams[i].Gains.Current.data(), MaxAmbiChannels,
            out->send_current[i]);
        ToBiquad(chan.mWetParams[i].LowPass, &out->send_lp[i]);
        ToBiquad(chan.mWetParams[i].HighPass, &out->send_hp[i]);
    }
    return 0;
}

int oal_scene_set_direct_hrtf(oal_scene *s, const float *chan_coeffs, const float *hfscales,
    float xover_norm, uint32_t irsize)
{
    if(!s->desc.hrtf) return -1;
    auto &proc = std::get<HrtfPostProcess>(s->dev->mPostProcess);
    auto &state = *proc.mHrtfState;
    state.mIrSize = irsize;
    for(size_t c{0};c < state.mChannels.size();++c)
    {
        state.mChannels[c].mSplitter.init(xover_norm);
        state.mChannels[c].mHfScale = hfscales[c];
        std::memcpy(state.mChannels[c].mCoeffs.data(), chan_coeffs + c*HrirLength*2,
            sizeof(HrirArray));
    }
    return 0;
}

/* (the convolution reverb lives in ref_conv.cpp, which compiles alc/effects/convolution.cpp itself to
 * reach its file-local ConvolutionState) */

/* ---- the small EffectStates: alc/effects/{equalizer,modulator,echo,dedicated}.cpp behind their factories ---- */
struct oal_effect {
    std::unique_ptr<Dev> dev;
    std::unique_ptr<Ctx> ctx;
    EffectSlotBase slot;
    al::intrusive_ptr<EffectState> state;
    EffectProps props;
    std::array<FloatBufferLine, 9> wet{};
    size_t numWet{4};
    int kind{};
};

/* kind: 0 equalizer, 1 modulator, 2 echo, 3 dedicated, 4 compressor, 5 chorus / flanger, 6 distortion, 7 autowah,
 * 8 vocal morpher, 9 frequency shifter, 10 pitch shifter; a device with num_out_lines dry lines (identity AmbiMap:
 * line i = ACN i, scale 1) and, for the dedicated effect, num_real real output lines whose FrontCenter sits at
 * front_center (< 0: none).  ambi_order / horizontal / xover_freq: the device's mAmbiOrder, m2DMixing and mXOverFreq
 * (what the A-Format effects' deviceUpdate builds its up-sampler from); wet_channels: the slot's wet lines */
oal_effect *oal_effect_create_ex(int kind, uint32_t sample_rate, uint32_t num_out_lines, uint32_t num_real, int front_center,
    uint32_t ambi_order, int horizontal, float xover_freq, uint32_t wet_channels)
{
    ApplySimd();
    if(wet_channels < 1 || wet_channels > 9) return nullptr;
    auto e = std::make_unique<oal_effect>();
    e->kind = kind;
    e->numWet = wet_channels;
    e->dev = std::make_unique<Dev>();
    auto &dev = *e->dev;
    dev.mSampleRate = sample_rate;
    dev.mUpdateSize = BufferLineSize;
    dev.mBufferSize = BufferLineSize;
    dev.FmtType = DevFmtFloat;
    dev.mAmbiOrder = ambi_order;
    dev.m2DMixing = horizontal != 0;
    dev.mXOverFreq = xover_freq;
    dev.MixBuffer.resize(num_out_lines + num_real);
    dev.Dry.Buffer = std::span{dev.MixBuffer}.first(num_out_lines);
    dev.RealOut.Buffer = num_real ? std::span{dev.MixBuffer}.subspan(num_out_lines) : dev.Dry.Buffer;
    dev.RealOut.ChannelIndex.fill(InvalidChannelIndex);
    if(front_center >= 0) dev.RealOut.ChannelIndex[FrontCenter] = u8{static_cast<u8::value_t>(front_center)};
    for(uint32_t i{0};i < num_out_lines;++i) dev.Dry.AmbiMap[i] = BFChannelConfig{1.0f, i};
    e->ctx = std::make_unique<Ctx>(e->dev.get());
    e->slot.mWetBuffer.resize(wet_channels);
    e->slot.Wet.Buffer = e->slot.mWetBuffer;
    for(uint32_t i{0};i < wet_channels;++i) e->slot.Wet.AmbiMap[i] = BFChannelConfig{1.0f, i};
    switch(kind)
    {
    case 0: e->state = EqualizerStateFactory_getFactory()->create(); break;
    case 1: e->state = ModulatorStateFactory_getFactory()->create(); break;
    case 2: e->state = EchoStateFactory_getFactory()->create(); break;
    case 3: e->state = DedicatedStateFactory_getFactory()->create(); break;
    case 4: e->state = CompressorStateFactory_getFactory()->create(); break;
    case 5: e->state = ChorusStateFactory_getFactory()->create(); break;
    case 6: e->state = DistortionStateFactory_getFactory()->create(); break;
    case 7: e->state = AutowahStateFactory_getFactory()->create(); break;
    case 8: e->state = VmorpherStateFactory_getFactory()->create(); break;
    case 9: e->state = FshifterStateFactory_getFactory()->create(); break;
    case 10: e->state = PshifterStateFactory_getFactory()->create(); break;
    default: return nullptr;
    }
    e->state->deviceUpdate(e->dev.get(), nullptr);
    return e.release();
}

oal_effect *oal_effect_create(int kind, uint32_t sample_rate, uint32_t num_out_lines, uint32_t num_real, int front_center)
{ return oal_effect_create_ex(kind, sample_rate, num_out_lines, num_real, front_center, 1u, 0, 400.0f, 4u); }

/* AmbiScale::SecondOrderUp (9 x MaxAmbiChannels) and GetHFOrderScales(2, device_order, horizontal)[0..1]: what the pitch
 * shifter's up-sampler is built from (pshifter.cpp:148-165, :188-198) */
void oal_ambi_upmix_info2(uint32_t device_order, int horizontal, float *order_scales2, float *second_order_up)
{
    auto const scales = AmbiScale::GetHFOrderScales(2, device_order, horizontal != 0);
    order_scales2[0] = scales[0]; order_scales2[1] = scales[1];
    for(size_t i{0};i < 9;++i)
        std::copy_n(AmbiScale::SecondOrderUp[i].begin(), MaxAmbiChannels, second_order_up + i*MaxAmbiChannels);
}

/* props: the floats of the effect's property struct in declaration order (core/effects/base.h:116-169; the
 * modulator's waveform and the dedicated effect's target as a float-coded integer) */
void oal_effect_update(oal_effect *e, const float *p, float slot_gain)
{
    switch(e->kind)
    {
    case 0: e->props = EqualizerProps{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9]}; break;
    case 1: e->props = ModulatorProps{p[0], p[1], static_cast<ModulatorWaveform>(static_cast<int>(p[2]))}; break;
    case 2: e->props = EchoProps{p[0], p[1], p[2], p[3], p[4]}; break;
    case 3: e->props = DedicatedProps{static_cast<int>(p[0]) ? DedicatedProps::Lfe : DedicatedProps::Dialog, p[1]}; break;
    case 4: e->props = CompressorProps{p[0] != 0.0f}; break;
    case 5: e->props = ChorusProps{static_cast<ChorusWaveform>(static_cast<int>(p[0])), static_cast<int>(p[1]), p[2], p[3], p[4], p[5]}; break;
    case 6: e->props = DistortionProps{p[0], p[1], p[2], p[3], p[4]}; break;
    case 7: e->props = AutowahProps{p[0], p[1], p[2], p[3]}; break;
    case 8: e->props = VmorpherProps{p[0], static_cast<VMorpherPhenome>(static_cast<int>(p[1])),
        static_cast<VMorpherPhenome>(static_cast<int>(p[2])), static_cast<int>(p[3]), static_cast<int>(p[4]),
        static_cast<VMorpherWaveform>(static_cast<int>(p[5]))}; break;
    case 9: e->props = FshifterProps{p[0], static_cast<FShifterDirection>(static_cast<int>(p[1])),
        static_cast<FShifterDirection>(static_cast<int>(p[2]))}; break;
    default: e->props = PshifterProps{static_cast<int>(p[0]), static_cast<int>(p[1])}; break;
    }
    e->slot.Gain = slot_gain;
    e->state->update(e->ctx.get(), &e->slot, &e->props, EffectTarget{&e->dev->Dry, &e->dev->RealOut});
}

/* wet_in: wet_channels x 1024; lines: (num_out_lines + num_real) x 1024, added to */
void oal_effect_process(oal_effect *e, const float *wet_in, float *lines, uint32_t n)
{
    auto const fpuctl = FPUCtl{};
    auto &dev = *e->dev;
    for(size_t c{0};c < e->numWet;++c) std::copy_n(wet_in + c*BufferLineSize, BufferLineSize, e->wet[c].begin());
    for(size_t l{0};l < dev.MixBuffer.size();++l)
        std::copy_n(lines + l*BufferLineSize, BufferLineSize, dev.MixBuffer[l].begin());
    e->state->process(n, std::span{e->wet}.first(e->numWet), e->state->mOutTarget);
    for(size_t l{0};l < dev.MixBuffer.size();++l)
        std::copy_n(dev.MixBuffer[l].begin(), BufferLineSize, lines + l*BufferLineSize);
}

/* 1 when the state's output target is the real output lines (a dedicated effect with a FrontCenter / LFE line) */
int oal_effect_targets_real(oal_effect *e)
{ return e->state->mOutTarget.data() == e->dev->RealOut.Buffer.data() && e->dev->RealOut.Buffer.data() != e->dev->Dry.Buffer.data(); }

void oal_effect_destroy(oal_effect *e) { delete e; }

/* SampleConverter (core/converter.cpp:175-330): Create / availableOut / convert as they are */
struct oal_converter { SampleConverterPtr conv; };
oal_converter *oal_converter_create(int src_type, int dst_type, uint32_t channels, uint32_t src_rate, uint32_t dst_rate,
    int resampler)
{
    ApplySimd();
    auto c = std::make_unique<oal_converter>();
    c->conv = SampleConverter::Create(static_cast<DevFmtType>(src_type), static_cast<DevFmtType>(dst_type), channels,
        src_rate, dst_rate, static_cast<Resampler>(resampler));
    return c->conv ? c.release() : nullptr;
}
uint32_t oal_converter_available_out(oal_converter *c, uint32_t src_frames) { return c->conv->availableOut(src_frames); }
/* returns the frames written; *consumed_bytes = how far convert() advanced the source pointer */
uint32_t oal_converter_convert(oal_converter *c, const void *src, uint32_t *src_frames, void *dst, uint32_t dst_frames,
    uint64_t *consumed_bytes)
{
    const void *p = src;
    auto const n = c->conv->convert(&p, src_frames, dst, dst_frames);
    *consumed_bytes = static_cast<uint64_t>(static_cast<const char*>(p) - static_cast<const char*>(src));
    return n;
}
void oal_converter_destroy(oal_converter *c) { delete c; }

/* PPhaseResampler (common/polyphase_resampler.cpp): init(src_rate, dst_rate) + process(in, out) */
void oal_pphase_resample(uint32_t src_rate, uint32_t dst_rate, const double *in, size_t n_in, double *out,
    size_t n_out)
{
    auto rs = PPhaseResampler{};
    rs.init(src_rate, dst_rate);
    rs.process(std::span{in, n_in}, std::span{out, n_out});
}

/* ---- BFormatDec (core/bformatdec.cpp:27-95), the AmbiDecPostProcess of non-HRTF devices ---- */
struct oal_bformatdec { std::unique_ptr<BFormatDec> dec; size_t inchans, nout; };

oal_bformatdec *oal_bformatdec_create(uint32_t inchans, uint32_t nout, const float *coeffs_hf,
    const float *coeffs_lf, float xover_norm)
{
    ApplySimd();
    auto d = std::make_unique<oal_bformatdec>();
    d->inchans = inchans; d->nout = nout;
    auto hf = std::vector<ChannelDec>(nout), lf = std::vector<ChannelDec>(coeffs_lf ? nout : 0u);
    for(size_t o{0};o < nout;++o)
    {
        hf[o] = ChannelDec{};
        std::copy_n(coeffs_hf + o*MaxAmbiChannels, MaxAmbiChannels, hf[o].begin());
        if(coeffs_lf)
        {
            lf[o] = ChannelDec{};
            std::copy_n(coeffs_lf + o*MaxAmbiChannels, MaxAmbiChannels, lf[o].begin());
        }
    }
    d->dec = std::make_unique<BFormatDec>(inchans, hf, lf, xover_norm);
    return d.release();
}

/* out: nout x 1024 (accumulated into, like DeviceBase::RealOut), in: inchans x 1024 */
void oal_bformatdec_process(oal_bformatdec *d, float *out, const float *in, uint32_t n)
{
    auto const fpuctl = FPUCtl{};
    auto outl = al::vector<FloatBufferLine,16>(d->nout);
    auto inl = al::vector<FloatBufferLine,16>(d->inchans);
    for(size_t c{0};c < d->nout;++c) std::copy_n(out + c*BufferLineSize, BufferLineSize, outl[c].data());
    for(size_t c{0};c < d->inchans;++c) std::copy_n(in + c*BufferLineSize, BufferLineSize, inl[c].data());
    d->dec->process(std::span{outl}, std::span<const FloatBufferLine>{inl.data(), inl.size()}, n);
    for(size_t c{0};c < d->nout;++c) std::copy_n(outl[c].data(), BufferLineSize, out + c*BufferLineSize);
}

void oal_bformatdec_destroy(oal_bformatdec *d) { delete d; }

void oal_calc_direction_coeffs(const float dir[3], float spread, float *out25)
{
    auto const coeffs = CalcDirectionCoeffs(std::span<const float,3>{dir, 3}, spread);
    std::copy(coeffs.begin(), coeffs.end(), out25);
}

} // extern "C"
