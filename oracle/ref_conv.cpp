/* oracle/ref_conv.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The reference's convolution reverb behind the oal_conv_* C ABI of oracle/oalref.h.  ConvolutionState is a
 * file-local type of alc/effects/convolution.cpp, so that translation unit is compiled HERE, in place from
 * /root/reference, by including it (nothing is copied into this repository); the harness below can then
 * read what ConvolutionState::update() computes per channel -- the Target gains, the HF/LF scales and
 * whether UpsampleMix was chosen -- which is what the product's C-ABI takes from its caller. */
#include "config.h"
#include "alc/effects/convolution.cpp"

#include <cstring>
#include <memory>
#include <vector>

#include "core/fpu_ctrl.h"
#include "oalref.h"

void oalref_apply_simd();

namespace {
struct CDev final : DeviceBase { CDev() : DeviceBase{DeviceType::Loopback} { } };
struct CCtx final : ContextBase { explicit CCtx(DeviceBase *d) : ContextBase{d} { } };
} // namespace

struct oal_conv {
    std::unique_ptr<CDev> dev;
    std::unique_ptr<CCtx> ctx;
    EffectSlotBase slot;
    al::intrusive_ptr<EffectState> state;
    std::vector<float> ir;
    BufferStorage storage;
    EffectProps props;
    std::array<FloatBufferLine, 1> wet{};
    ConvolutionState *cs() { return static_cast<ConvolutionState*>(state.get()); }
};

extern "C" {

/* `ir`: ir_len interleaved frames of `channels` samples (1 = mono, 2 = stereo, 4 = first-order B-Format in
 * ACN order with N3D scaling) at ir_rate; the device has ambisonic order `device_order` and
 * num_out_lines = (device_order + 1)^2 dry lines with the identity AmbiMap. */
oal_conv *oal_conv_create_ex(uint32_t sample_rate, uint32_t num_out_lines, uint32_t device_order, const float *ir,
    uint32_t ir_len, uint32_t channels, uint32_t ir_rate)
{
    oalref_apply_simd();
    if(channels != 1 && channels != 2 && channels != 4) return nullptr;
    auto c = std::make_unique<oal_conv>();
    c->dev = std::make_unique<CDev>();
    auto &dev = *c->dev;
    dev.mSampleRate = sample_rate;
    dev.mUpdateSize = BufferLineSize;
    dev.mBufferSize = BufferLineSize;
    dev.FmtType = DevFmtFloat;
    dev.mAmbiOrder = device_order;
    dev.MixBuffer.resize(num_out_lines);
    dev.Dry.Buffer = std::span{dev.MixBuffer};
    dev.RealOut.Buffer = dev.Dry.Buffer;
    for(uint32_t i{0};i < num_out_lines;++i) dev.Dry.AmbiMap[i] = BFChannelConfig{1.0f, i};
    c->ctx = std::make_unique<CCtx>(c->dev.get());
    c->ir.assign(ir, ir + size_t{ir_len}*channels);
    c->storage.mData = std::span<f32>{reinterpret_cast<f32*>(c->ir.data()), c->ir.size()};
    c->storage.mSampleRate = ir_rate;
    c->storage.mChannels = channels == 1 ? FmtMono : (channels == 2 ? FmtStereo : FmtBFormat3D);
    c->storage.mType = FmtFloat;
    c->storage.mSampleLen = ir_len;
    c->storage.mAmbiLayout = AmbiLayout::ACN;
    c->storage.mAmbiScaling = AmbiScaling::N3D;
    c->storage.mAmbiOrder = channels == 4 ? 1u : 0u;
    c->state = ConvolutionStateFactory_getFactory()->create();
    c->state->deviceUpdate(c->dev.get(), &c->storage);
    c->props = ConvolutionProps{{0.0f, 0.0f, -1.0f}, {0.0f, 1.0f, 0.0f}};
    return c.release();
}

oal_conv *oal_conv_create(uint32_t sample_rate, uint32_t num_out_lines, const float *ir,
    uint32_t ir_len, uint32_t ir_rate)
{ return oal_conv_create_ex(sample_rate, num_out_lines, 1, ir, ir_len, 1, ir_rate); }

void oal_conv_set_orientation(oal_conv *c, const float at[3], const float up[3])
{ c->props = ConvolutionProps{{at[0], at[1], at[2]}, {up[0], up[1], up[2]}}; }

void oal_conv_update(oal_conv *c, float slot_gain)
{
    c->slot.Gain = slot_gain;
    c->state->update(c->ctx.get(), &c->slot, &c->props, EffectTarget{&c->dev->Dry, &c->dev->RealOut});
}

/* what update() left in mChans: targets = channels x 25 (MaxAmbiChannels), hf / lf = channels; returns the
 * number of channels; *upsample = (mMix == UpsampleMix); *xover_norm = the splitters' f0norm */
uint32_t oal_conv_channel_info(oal_conv *c, float *targets, float *hf, float *lf, int *upsample, float *xover_norm)
{
    auto &st = *c->cs();
    for(size_t i{0};i < st.mChans.size();++i)
    {
        std::copy_n(st.mChans[i].Target.begin(), MaxAmbiChannels, targets + i*MaxAmbiChannels);
        hf[i] = st.mChans[i].mHfScale;
        lf[i] = st.mChans[i].mLfScale;
    }
    *upsample = st.mMix == &ConvolutionState::UpsampleMix;
    *xover_norm = c->dev->mXOverFreq / static_cast<float>(c->dev->mSampleRate);
    return static_cast<uint32_t>(st.mChans.size());
}

void oal_conv_process(oal_conv *c, const float *wet_in, float *out_lines, uint32_t n)
{
    auto const fpuctl = FPUCtl{};
    auto &dev = *c->dev;
    std::copy_n(wet_in, n, c->wet[0].begin());
    for(size_t l{0};l < dev.MixBuffer.size();++l)
        std::copy_n(out_lines + l*BufferLineSize, BufferLineSize, dev.MixBuffer[l].begin());
    c->state->process(n, c->wet, c->state->mOutTarget);
    for(size_t l{0};l < dev.MixBuffer.size();++l)
        std::copy_n(dev.MixBuffer[l].begin(), BufferLineSize, out_lines + l*BufferLineSize);
}

void oal_conv_destroy(oal_conv *c) { delete c; }

} // extern "C"
