/* oracle/ref_reverb.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference's EAX reverb behind the oal_reverb_* C ABI of oracle/oalref.h.  ReverbState and
 * its pipelines are file-local types of alc/effects/reverb.cpp, so that translation unit is
 * compiled HERE, in place from /root/reference, by including it (nothing is copied into this
 * repository); the harness below can then read the fields ReverbState::update() computes. */
#include "config.h"
#include "alc/effects/reverb.cpp"

#include <cstring>
#include <memory>

#include "core/fpu_ctrl.h"
#include "oalref.h"

namespace {

struct RDev final : DeviceBase { RDev() : DeviceBase{DeviceType::Loopback} { } };
struct RCtx final : ContextBase { explicit RCtx(DeviceBase *d) : ContextBase{d} { } };

void GetBq(BiquadFilter const &f, oal_bq *o)
{
    /* BiquadFilter keeps its coefficients protected; read them through a layout-compatible view */
    struct View { float z1, z2, b0, b1, b2, a1, a2; };
    static_assert(sizeof(View) == sizeof(BiquadFilter));
    View v;
    std::memcpy(&v, &f, sizeof(v));
    *o = oal_bq{v.b0, v.b1, v.b2, v.a1, v.a2};
}

} // namespace

struct oal_reverb {
    std::unique_ptr<RDev> dev;
    std::unique_ptr<RCtx> ctx;
    EffectSlotBase slot;
    al::intrusive_ptr<EffectState> state;
    EffectProps props;
    std::array<FloatBufferLine, 4> wet{};
    ReverbState *rs() { return static_cast<ReverbState*>(state.get()); }
};

extern "C" {

/* AmbiScale::GetHFOrderScales(1, device_order, horizontal)[0..1], AmbiScale::FirstOrderUp (4 x 25) and the
 * default crossover (DeviceBase::mXOverFreq) relative to sample_rate: what a device of that order hands the
 * effect in deviceUpdate (reverb.cpp:835-851) */
void oal_ambi_upmix_info(uint32_t device_order, int horizontal, uint32_t sample_rate, float *order_scales2,
    float *first_order_up, float *xover_norm)
{
    auto const scales = AmbiScale::GetHFOrderScales(1, device_order, horizontal != 0);
    order_scales2[0] = scales[0]; order_scales2[1] = scales[1];
    for(size_t i{0};i < 4;++i)
        std::copy_n(AmbiScale::FirstOrderUp[i].begin(), MaxAmbiChannels, first_order_up + i*MaxAmbiChannels);
    *xover_norm = RDev{}.mXOverFreq / static_cast<float>(sample_rate);
}

oal_reverb *oal_reverb_create(uint32_t sample_rate, uint32_t num_out_lines)
{ return oal_reverb_create_ex(sample_rate, num_out_lines, 1); }

/* device_order > 1: the reference's ReverbState up-mixes (mUpmixOutput); num_out_lines = (order+1)^2 */
oal_reverb *oal_reverb_create_ex(uint32_t sample_rate, uint32_t num_out_lines, uint32_t device_order)
{
    auto r = std::make_unique<oal_reverb>();
    r->dev = std::make_unique<RDev>();
    auto &dev = *r->dev;
    dev.mSampleRate = sample_rate;
    dev.mUpdateSize = BufferLineSize;
    dev.mBufferSize = BufferLineSize;
    dev.FmtType = DevFmtFloat;
    dev.mAmbiOrder = device_order;
    dev.MixBuffer.resize(num_out_lines);
    dev.Dry.Buffer = std::span{dev.MixBuffer};
    dev.RealOut.Buffer = dev.Dry.Buffer;
    for(uint32_t i{0};i < num_out_lines;++i) dev.Dry.AmbiMap[i] = BFChannelConfig{1.0f, i};
    r->ctx = std::make_unique<RCtx>(r->dev.get());
    r->state = ReverbStateFactory_getFactory()->create();
    r->state->deviceUpdate(r->dev.get(), nullptr);
    return r.release();
}

void oal_reverb_destroy(oal_reverb *r) { delete r; }

int oal_reverb_update(oal_reverb *r, const oal_reverb_props *p, float slot_gain)
{
    auto rp = ReverbProps{};
    rp.Density = p->density; rp.Diffusion = p->diffusion; rp.Gain = p->gain; rp.GainHF = p->gain_hf;
    rp.GainLF = p->gain_lf; rp.DecayTime = p->decay_time; rp.DecayHFRatio = p->decay_hf_ratio;
    rp.DecayLFRatio = p->decay_lf_ratio; rp.ReflectionsGain = p->reflections_gain;
    rp.ReflectionsDelay = p->reflections_delay;
    rp.ReflectionsPan = {p->reflections_pan[0], p->reflections_pan[1], p->reflections_pan[2]};
    rp.LateReverbGain = p->late_reverb_gain; rp.LateReverbDelay = p->late_reverb_delay;
    rp.LateReverbPan = {p->late_reverb_pan[0], p->late_reverb_pan[1], p->late_reverb_pan[2]};
    rp.EchoTime = p->echo_time; rp.EchoDepth = p->echo_depth; rp.ModulationTime = p->modulation_time;
    rp.ModulationDepth = p->modulation_depth; rp.AirAbsorptionGainHF = p->air_absorption_gain_hf;
    rp.HFReference = p->hf_reference; rp.LFReference = p->lf_reference;
    rp.RoomRolloffFactor = p->room_rolloff_factor; rp.DecayHFLimit = p->decay_hf_limit != 0;
    r->props = rp;
    r->slot.Gain = slot_gain;
    r->state->update(r->ctx.get(), &r->slot, &r->props, EffectTarget{&r->dev->Dry, &r->dev->RealOut});
    return 0;
}

int oal_reverb_get_params(oal_reverb *r, oal_reverb_params *out)
{
    auto *rs = r->rs();
    std::memset(out, 0, sizeof(*out));
    out->pipeline_state = static_cast<int32_t>(rs->mPipelineState);
    out->current_pipeline = rs->mCurrentPipeline ? 1 : 0;
    for(int pi{0};pi < 2;++pi)
    {
        auto &P = rs->mPipelines[static_cast<size_t>(pi)];
        auto &o = out->pipe[pi];
        GetBq(P.mFilter[0].Lp, &o.filter_lp);
        GetBq(P.mFilter[0].Hp, &o.filter_hp);
        for(size_t j{0};j < 4;++j)
        {
            o.early_delay_tap[j][0] = static_cast<uint32_t>(P.mEarlyDelayTap[j][0]);
            o.early_delay_tap[j][1] = static_cast<uint32_t>(P.mEarlyDelayTap[j][1]);
            o.late_delay_tap[j][0] = static_cast<uint32_t>(P.mLateDelayTap[j][0]);
            o.late_delay_tap[j][1] = static_cast<uint32_t>(P.mLateDelayTap[j][1]);
            o.early_ap_offset[j] = static_cast<uint32_t>(P.mEarly.Allpass.Offset[j]);
            o.early_offset[j] = static_cast<uint32_t>(P.mEarly.Offset[j]);
            o.late_offset[j] = static_cast<uint32_t>(P.mLate.Offset[j]);
            o.late_ap_offset[j] = static_cast<uint32_t>(P.mLate.VecAp.Offset[j]);
            o.t60_mid_gain[j] = P.mLate.T60[j].mMidGain;
            GetBq(P.mLate.T60[j].mHFFilter, &o.t60_hf[j]);
            GetBq(P.mLate.T60[j].mLFFilter, &o.t60_lf[j]);
            for(size_t c{0};c < 25;++c)
            {
                o.early_gains_target[j][c] = P.mEarly.Gains[j].Target[c];
                o.late_gains_target[j][c] = P.mLate.Gains[j].Target[c];
            }
        }
        o.early_delay_coeff[0] = P.mEarlyDelayCoeff[0];
        o.early_delay_coeff[1] = P.mEarlyDelayCoeff[1];
        o.mix_x = P.mMixX; o.mix_y = P.mMixY;
        o.early_ap_coeff = P.mEarly.Allpass.Coeff;
        o.early_coeff = P.mEarly.Coeff;
        o.late_density_gain = P.mLate.DensityGain;
        o.mod_step = P.mLate.Mod.Step;
        o.mod_depth = P.mLate.Mod.Depth;
        o.late_ap_coeff = P.mLate.VecAp.Coeff;
        o.fade_sample_count = static_cast<uint32_t>(P.mFadeSampleCount);
    }
    return 0;
}

int oal_reverb_set_params(oal_reverb*, const oal_reverb_params*) { return -1; }

void oal_reverb_process(oal_reverb *r, const float *wet_in, float *out_lines, uint32_t n)
{
    auto const fpuctl = FPUCtl{};
    auto &dev = *r->dev;
    for(size_t c{0};c < 4;++c) std::copy_n(wet_in + c*BufferLineSize, n, r->wet[c].begin());
    for(size_t l{0};l < dev.MixBuffer.size();++l)
        std::copy_n(out_lines + l*BufferLineSize, BufferLineSize, dev.MixBuffer[l].begin());
    r->state->process(n, r->wet, r->state->mOutTarget);
    for(size_t l{0};l < dev.MixBuffer.size();++l)
        std::copy_n(dev.MixBuffer[l].begin(), BufferLineSize, out_lines + l*BufferLineSize);
}

int oal_reverb_line_lengths(oal_reverb *r, uint32_t *lengths11)
{
    auto *rs = r->rs();
    size_t i{0};
    lengths11[i++] = static_cast<uint32_t>(rs->mMainDelay.mLine.size());
    for(auto &P : rs->mPipelines)
    {
        lengths11[i++] = static_cast<uint32_t>(P.mLateDelayIn.mLine.size());
        lengths11[i++] = static_cast<uint32_t>(P.mEarly.Allpass.Delay.mLine.size());
        lengths11[i++] = static_cast<uint32_t>(P.mEarly.Delay.mLine.size());
        lengths11[i++] = static_cast<uint32_t>(P.mLate.VecAp.Delay.mLine.size());
        lengths11[i++] = static_cast<uint32_t>(P.mLate.Delay.mLine.size());
    }
    return static_cast<int>(rs->mSampleBuffer.size());
}

} // extern "C"
