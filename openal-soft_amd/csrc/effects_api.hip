// C-ABI of the small EffectStates (include/oalgpu.h: oalgpu_effect_*): deviceUpdate / update as the reference does
// them on its mixer thread with libm -- biquad design, the echo's delays, the modulator's carrier period -- and the
// scalar bookkeeping process() performs (delay-line offset, carrier index); the block itself is one launch
// (effects_kernels.hip).
#include "api_util.hpp"
#include "kernels.hpp"
#include "../host/params.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <vector>

using namespace oalgpu;

struct oalgpu_effect {
    int device{0}, kind{0};
    bool exact{false};
    uint32_t sampleRate{48000}, numIn{1}, nlines{1};
    bool updated{false};
    DevBuf<FxState> st;
    DevBuf<float> tgtGains, delay, hostIn, hostOut;
    FxLaunch F{};
    // ModulatorState::mIndex / mRange (modulator.cpp:75-77), EchoState::mOffset (echo.cpp:55)
    uint32_t modIndex{0}, modRange{1};
    uint32_t echoOffset{0};
};

namespace {

int UploadBiquad(oalgpu_effect *e, uint32_t chan, uint32_t which, const float c[5])
{   // copyParamsFrom / setParams: the coefficients change, the filter's history stays
    float five[5] = {c[0], c[1], c[2], c[3], c[4]};
    char *dst = reinterpret_cast<char*>(e->st.p) + offsetof(FxState, bq) + (size_t{chan} * 4 + which) * sizeof(BiquadState)
        + offsetof(BiquadState, b0);
    HIP_TRY(hipMemcpy(dst, five, sizeof(five), hipMemcpyHostToDevice));
    return OALGPU_OK;
}

uint32_t NextPow2(uint32_t v) { uint32_t p = 1; while(p < v) p <<= 1; return p; }

} // namespace

extern "C" {

int oalgpu_effect_create(int device, int math_mode, int kind, uint32_t sample_rate, uint32_t num_in_channels,
    uint32_t num_out_lines, oalgpu_effect **out)
{
    if(!out || kind < OALGPU_EFFECT_EQUALIZER || kind > OALGPU_EFFECT_COMPRESSOR || sample_rate < 8000 || num_in_channels < 1
        || num_in_channels > kFxMaxIn || num_out_lines < 1 || num_out_lines > OALGPU_MAX_OUTPUT_CHANNELS)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_create: bad arguments");
    *out = nullptr;
    if(int rc = UseDevice(device)) return rc;
    auto e = std::make_unique<oalgpu_effect>();
    e->device = device; e->kind = kind; e->exact = math_mode == OALGPU_MATH_EXACT;
    e->sampleRate = sample_rate; e->numIn = num_in_channels; e->nlines = num_out_lines;
    HIP_TRY(e->st.alloc(1)); HIP_TRY(e->st.zero());
    HIP_TRY(e->tgtGains.alloc(2 * OALGPU_MAX_OUTPUT_CHANNELS)); HIP_TRY(e->tgtGains.zero());
    HIP_TRY(e->hostIn.alloc(size_t{num_in_channels} * OALGPU_BUFFER_LINE_SIZE));
    HIP_TRY(e->hostOut.alloc(size_t{num_out_lines} * OALGPU_BUFFER_LINE_SIZE));
    FxLaunch &F = e->F;
    F.kind = kind; F.exact = e->exact ? 1 : 0; F.numIn = num_in_channels; F.nlines = num_out_lines;
    F.st = e->st.p; F.tgtGains = e->tgtGains.p;
    for(uint32_t c = 0; c < kFxMaxIn; ++c) { F.target[c] = OALGPU_INVALID_CHANNEL; F.tgtGain[c] = 0.0f; }
    F.modRange = 1; F.modWave = 0;
    if(kind == OALGPU_EFFECT_ECHO)
    {   // EchoState::deviceUpdate, echo.cpp:77-91: EchoMaxDelay 0.207 s + EchoMaxLRDelay 0.404 s, next power of two
        const float f = float(sample_rate);
        const uint32_t len = NextPow2(uint32_t(0.207f * f + 0.5f) + uint32_t(0.404f * f + 0.5f));
        HIP_TRY(e->delay.alloc(len)); HIP_TRY(e->delay.zero());
        F.delay = e->delay.p; F.delayMask = len - 1u;
    }
    if(kind == OALGPU_EFFECT_COMPRESSOR)
    {   // CompressorState::deviceUpdate, compressor.cpp:87-101: 100 ms from 0.5 to 2, 200 ms back; mEnvFollower = 1
        F.attackMult = std::pow(2.0f / 0.5f, 1.0f / (float(sample_rate) * 0.1f));
        F.releaseMult = std::pow(0.5f / 2.0f, 1.0f / (float(sample_rate) * 0.2f));
        const float one = 1.0f;
        HIP_TRY(hipMemcpy(reinterpret_cast<char*>(e->st.p) + offsetof(FxState, env), &one, sizeof(one), hipMemcpyHostToDevice));
    }
    // a BiquadFilter starts as the identity (mB0 = 1)
    for(uint32_t c = 0; c < kFxMaxIn; ++c)
        for(uint32_t k = 0; k < 4; ++k)
        {
            const float ident[5] = {1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            if(int rc = UploadBiquad(e.get(), c, k, ident)) return rc;
        }
    *out = e.release();
    return OALGPU_OK;
}

void oalgpu_effect_destroy(oalgpu_effect *e)
{
    if(!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    delete e;
}

int oalgpu_effect_update(oalgpu_effect *e, const void *props, const uint32_t *target_channels, const float *gains)
{
    if(!e || !gains) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: null argument");
    if(int rc = UseDevice(e->device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const float rate = float(e->sampleRate);
    FxLaunch &F = e->F;
    switch(e->kind)
    {
    case OALGPU_EFFECT_EQUALIZER:
        {   // EqualizerState::update, equalizer.cpp:115-165
            if(!props || !target_channels) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: equalizer needs props and targets");
            const auto &p = *static_cast<const oalgpu_equalizer_props*>(props);
            float c[4][5];
            DesignBiquadFromSlope(OALGPU_BIQUAD_LOWSHELF, p.low_cutoff / rate, std::sqrt(p.low_gain), 0.75f, c[0]);
            DesignBiquadFromBandwidth(OALGPU_BIQUAD_PEAKING, p.mid1_center / rate, std::sqrt(p.mid1_gain), p.mid1_width, c[1]);
            DesignBiquadFromBandwidth(OALGPU_BIQUAD_PEAKING, p.mid2_center / rate, std::sqrt(p.mid2_gain), p.mid2_width, c[2]);
            DesignBiquadFromSlope(OALGPU_BIQUAD_HIGHSHELF, p.high_cutoff / rate, std::sqrt(p.high_gain), 0.75f, c[3]);
            for(uint32_t ch = 0; ch < e->numIn; ++ch)
                for(uint32_t k = 0; k < 4; ++k)
                    if(int rc = UploadBiquad(e, ch, k, c[k])) return rc;
        }
        break;
    case OALGPU_EFFECT_MODULATOR:
        {   // ModulatorState::update, modulator.cpp:103-163
            if(!props || !target_channels) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: modulator needs props and targets");
            const auto &p = *static_cast<const oalgpu_modulator_props*>(props);
            const float perCycle = p.frequency > 0.0f ? rate / p.frequency + 0.5f : 1.0f;
            const uint32_t range = uint32_t(std::min(std::max(perCycle, 1.0f), rate));
            e->modIndex = uint32_t(uint64_t{e->modIndex} * range / e->modRange);
            e->modRange = range;
            F.modScale = 0.0f; F.modWave = 0;
            if(e->modRange == 1) {}
            else if(p.waveform == OALGPU_MODULATOR_SINUSOID) { F.modScale = 3.14159265358979323846f * 2.0f / float(e->modRange); F.modWave = 1; }
            else if(p.waveform == OALGPU_MODULATOR_SAWTOOTH) { F.modScale = 2.0f / float(e->modRange - 1u); F.modWave = 2; }
            else
            {
                e->modRange = (e->modRange + 1u) & ~1u;
                F.modScale = 1.0f / float(e->modRange - 1u); F.modWave = 3;
            }
            const float f0norm = std::min(std::max(p.high_pass_cutoff / rate, 1.0f / 512.0f), 0.49f);
            float c[5];
            DesignBiquadFromBandwidth(OALGPU_BIQUAD_HIGHPASS, f0norm, 1.0f, 0.75f, c);
            for(uint32_t ch = 0; ch < e->numIn; ++ch)
                if(int rc = UploadBiquad(e, ch, 0, c)) return rc;
        }
        break;
    case OALGPU_EFFECT_COMPRESSOR:
        if(!props || !target_channels) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: compressor needs props and targets");
        F.compOn = static_cast<const oalgpu_compressor_props*>(props)->on_off ? 1 : 0;
        break;
    case OALGPU_EFFECT_ECHO:
        {   // EchoState::update, echo.cpp:93-117 (the two taps' panned gains come from the caller)
            if(!props) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: echo needs props");
            const auto &p = *static_cast<const oalgpu_echo_props*>(props);
            F.tap[0] = std::max(uint32_t(std::round(p.delay * rate)), 1u);
            F.tap[1] = uint32_t(std::round(p.lr_delay * rate)) + F.tap[0];
            if(F.tap[1] > F.delayMask) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: echo delays beyond AL_ECHO_MAX_DELAY + AL_ECHO_MAX_LRDELAY");
            const float gainhf = std::max(1.0f - p.damping, 0.0625f);
            float c[5];
            DesignBiquadFromSlope(OALGPU_BIQUAD_HIGHSHELF, 5000.0f / rate, gainhf, 1.0f, c);
            if(int rc = UploadBiquad(e, 0, 0, c)) return rc;
            F.feedGain = p.feedback;
            HIP_TRY(e->tgtGains.upload(gains, size_t{2} * e->nlines));
        }
        break;
    default:    // DedicatedState::update, dedicated.cpp:66-100: the gains of the target line(s), resolved by the caller
        HIP_TRY(e->tgtGains.upload(gains, e->nlines));
        break;
    }
    if(e->kind == OALGPU_EFFECT_EQUALIZER || e->kind == OALGPU_EFFECT_MODULATOR || e->kind == OALGPU_EFFECT_COMPRESSOR)
        for(uint32_t ch = 0; ch < e->numIn; ++ch) { F.target[ch] = target_channels[ch]; F.tgtGain[ch] = gains[ch]; }
    e->updated = true;
    return OALGPU_OK;
}

} // extern "C"

int oalgpu_effect_process_device(oalgpu_effect *e, void *hip_stream, const float *wet_in_dev, float *out_lines_dev, uint32_t n)
{
    if(!e || !wet_in_dev || !out_lines_dev || n == 0 || n > OALGPU_BUFFER_LINE_SIZE)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_process: bad arguments");
    if(!e->updated) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_process: no update() yet");
    if(int rc = UseDevice(e->device)) return rc;
    FxLaunch F = e->F;
    F.wetIn = wet_in_dev; F.outLines = out_lines_dev; F.n = n;
    F.modIndex = e->modIndex; F.modRange = e->modRange;
    F.offset = e->echoOffset;
    LaunchEffect(static_cast<hipStream_t>(hip_stream), F);
    HIP_TRY(hipGetLastError());
    // what process() does to the scalars (modulator.cpp:176-189, echo.cpp:127-157)
    if(e->kind == OALGPU_EFFECT_MODULATOR && e->modRange > 1) e->modIndex = (e->modIndex + n) % e->modRange;
    if(e->kind == OALGPU_EFFECT_ECHO) e->echoOffset = (e->echoOffset + n) & F.delayMask;
    return OALGPU_OK;
}

extern "C" int oalgpu_effect_process(oalgpu_effect *e, const float *wet_in, float *out_lines, uint32_t n)
{
    if(!e || !wet_in || !out_lines) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(int rc = UseDevice(e->device)) return rc;
    const size_t inFloats = size_t{e->numIn} * OALGPU_BUFFER_LINE_SIZE, outFloats = size_t{e->nlines} * OALGPU_BUFFER_LINE_SIZE;
    HIP_TRY(hipMemcpy(e->hostIn.p, wet_in, inFloats * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(e->hostOut.p, out_lines, outFloats * sizeof(float), hipMemcpyHostToDevice));
    if(int rc = oalgpu_effect_process_device(e, nullptr, e->hostIn.p, e->hostOut.p, n)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out_lines, e->hostOut.p, outFloats * sizeof(float), hipMemcpyDeviceToHost));
    return OALGPU_OK;
}

namespace oalgpu {
uint32_t EffectOutLines(const oalgpu_effect *e) { return e ? e->nlines : 0u; }
uint32_t EffectInChannels(const oalgpu_effect *e) { return e ? e->numIn : 0u; }
}
