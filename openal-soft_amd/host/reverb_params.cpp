// ReverbState::update and friends, alc/effects/reverb.cpp:722-1351, restated for the host side
// of liboalgpu.  Arithmetic follows the reference operation by operation (single precision,
// libm calls in the same places) so that the block matches the compiled reference bit for bit;
// tests/test_reverb.py checks exactly that.
#include "reverb_params.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "params.hpp"

namespace oalgpu {
namespace {

constexpr float kMinDecayTime = 0.1f, kMaxDecayTime = 20.0f;           // reverb.cpp:54-55
constexpr float kMaxModulationTime = 4.0f, kDefaultModulationTime = 0.25f, kMaxHFReference = 20000.0f;
constexpr float kModulationDepthCoeff = 0.05f;                        // :84
constexpr uint32_t kModFracOne = 1u << 24;                            // :60-61
constexpr uint32_t kMaxUpdateSamples = 256;                           // :68
constexpr float kDensityScale = 1000.0f;                              // :139
constexpr float kDecayGain = 0.001f;                                  // ReverbDecayGain, core/effects/base.h:22
constexpr float kMaxReflectionsDelay = 0.3f, kMaxLateReverbDelay = 0.1f;   // base.h:24-25
constexpr float kSpeedOfSound = 343.3f;                               // core/context.h:32
constexpr float kReverbBoost = 1.0f;                                  // alc/effects/base.h:11
constexpr float kSqrt3 = 1.732050807568877293527446341505872367f;
constexpr float kInvSqrt2 = static_cast<float>(1.0 / 1.414213562373095048801688724209698079);

constexpr float kEarlyTap[4] = {0.000000e+0f, 1.010676e-3f, 2.126553e-3f, 3.358580e-3f};   // :185-187
constexpr float kEarlyAllpass[4] = {4.854840e-4f, 5.360178e-4f, 5.918117e-4f, 6.534130e-4f}; // :195-197
constexpr float kEarlyLine[4] = {2.992520e-3f, 5.456575e-3f, 7.688329e-3f, 9.709681e-3f};  // :221-223
constexpr float kLateAllpass[4] = {8.091400e-4f, 1.019453e-3f, 1.407968e-3f, 1.618280e-3f}; // :229-231
constexpr float kLateLine[4] = {9.709681e-3f, 1.223343e-2f, 1.689561e-2f, 1.941936e-2f};   // :248-250

// std::reduce over four floats as libstdc++ evaluates it: init + ((a0+a1) + (a2+a3))
constexpr float Reduce4(const float (&a)[4]) { return 0.0f + ((a[0] + a[1]) + (a[2] + a[3])); }
constexpr float kLateAllpassAverage = Reduce4(kLateAllpass) / 4.0f;                        // :252-253
constexpr float kLateDelayAverage = Reduce4(kLateLine) / 4.0f + kLateAllpassAverage;       // :254-255

constexpr float kEarlyA2B[4][4] = {{0.5f, 0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f, -0.5f},   // :102-108
    {0.5f, -0.5f, -0.5f, 0.5f}, {0.5f, 0.5f, -0.5f, -0.5f}};
constexpr float kLateA2B[4][4] = {{0.5f, 0.5f, 0.5f, 0.5f}, {kInvSqrt2, -kInvSqrt2, 0.0f, 0.0f}, // :116-122
    {0.0f, 0.0f, -kInvSqrt2, kInvSqrt2}, {0.5f, 0.5f, -0.5f, -0.5f}};

// float2uint, common/alnumeric.h:223-240
uint32_t Trunc(float f)
{
    int32_t bits;
    std::memcpy(&bits, &f, sizeof(bits));
    const uint32_t keep = static_cast<uint32_t>(bits >> 31) ^ 0xffffffffu;
    const int shift = ((bits >> 23) & 0xff) - (127 + 23);
    if(shift < -23) return 0;
    if(shift > 8) return 0xffffffffu & keep;
    const uint32_t mant = (static_cast<uint32_t>(bits) & 0x7fffffu) | 0x800000u;
    return ((shift < 0) ? (mant >> -shift) : (mant << shift)) & keep;
}
// fastf2u: round to nearest even (cvtss2si), alnumeric.h:163-189
uint32_t RoundU(float f) { return static_cast<uint32_t>(static_cast<int32_t>(std::lrintf(f))); }

uint32_t NextPow2(uint32_t v)
{
    if(v > 0) { --v; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; }
    return v + 1;
}
// DelayLineU/I::calcLineLength, :273-284
uint32_t LineLength(float length, float frequency, uint32_t extra)
{ return NextPow2(Trunc(std::ceil(length * frequency)) + extra) * 4u; }

float DelayLengthMult(float density) { return std::max(1.0f, std::cbrt(density * kDensityScale)); }   // :722
float DecayCoeff(float length, float decayTime) { return std::pow(kDecayGain, length / decayTime); } // :861
float DecayLength(float coeff, float decayTime) { return std::log10(coeff) * decayTime / -3.0f; }    // :867
float Lerp(float a, float b, float mu) { return a + (b - a) * mu; }

void ShelfCoeffs(int type, float f0norm, float gain, oalgpu_bq_coeffs *out)
{
    float c[5];
    DesignBiquadFromSlope(type, f0norm, gain, 1.0f, c);
    *out = oalgpu_bq_coeffs{c[0], c[1], c[2], c[3], c[4]};
}

// GetTransformFromVector, :1111-1148 (transposed: row 0 carries the direction)
void TransformFromVector(const float vec[3], float m[4][4])
{
    float norm[3] = {vec[0], vec[1], vec[2]};
    float mag = std::sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
    if(mag > 1.0f)
    {
        const float scale = kSqrt3 / mag;
        norm[0] *= -scale; norm[1] *= scale; norm[2] *= scale;
        mag = 1.0f;
    }
    else
    {
        norm[0] *= -kSqrt3; norm[1] *= kSqrt3; norm[2] *= kSqrt3;
    }
    const float r[4][4] = {{1.0f, norm[0], norm[1], norm[2]}, {0.0f, 1.0f - mag, 0.0f, 0.0f},
        {0.0f, 0.0f, 1.0f - mag, 0.0f}, {0.0f, 0.0f, 0.0f, 1.0f - mag}};
    std::memcpy(m, r, sizeof(r));
}

// update3DPanning (:1151-1220) + ComputePanGains over an identity map.  Without up-mixing the A-to-B-Format
// conversion is combined with the panning transform (:1186-1203); with it (firstOrderUp != nullptr) the
// transform is combined with the first-order upsample matrix (:1166-1184) -- MixOutAmbiUp does the A-to-B
// conversion itself.
void PanGains(const float a2b[4][4], const float matrix[4][4], float gain, uint32_t numLines,
    const float (*firstOrderUp)[OALGPU_MAX_AMBI_CHANNELS], float target[4][OALGPU_MAX_AMBI_CHANNELS])
{
    for(int i = 0; i < 4; ++i)
    {
        float coeffs[OALGPU_MAX_AMBI_CHANNELS] = {};
        for(int j = 0; j < 4; ++j)
        {
            if(firstOrderUp)
            {
                const float a = matrix[i][j];
                for(int k = 0; k < OALGPU_MAX_AMBI_CHANNELS; ++k) coeffs[k] = a * firstOrderUp[j][k] + coeffs[k];
                continue;
            }
            const float a = a2b[j][i];
            for(int k = 0; k < 4; ++k) coeffs[k] = a * matrix[j][k] + coeffs[k];
        }
        for(uint32_t c = 0; c < OALGPU_MAX_AMBI_CHANNELS; ++c)
            target[i][c] = (c < numLines) ? 1.0f * coeffs[c] * gain : 0.0f;
    }
}

} // namespace

size_t ReverbLineLengths(float frequency, uint32_t lengths[11])
{
    const float multiplier = DelayLengthMult(1.0f);
    constexpr float maxModDelay = kMaxModulationTime * kModulationDepthCoeff / 2.0f;
    const uint32_t lateVecapExtra = Trunc(std::ceil(kLateAllpass[0] * multiplier * frequency));
    constexpr float lateDiffAvg = (kLateLine[3] - kLateLine[0]) / 4.0f;
    size_t k = 0, total = 0;
    lengths[k++] = LineLength(kMaxReflectionsDelay + kEarlyTap[3] * multiplier, frequency, OALGPU_BUFFER_LINE_SIZE);
    for(int p = 0; p < 2; ++p)
    {
        lengths[k++] = LineLength(kMaxLateReverbDelay + lateDiffAvg * multiplier, frequency, OALGPU_BUFFER_LINE_SIZE);
        lengths[k++] = LineLength(kEarlyAllpass[3] * multiplier, frequency, 0);
        lengths[k++] = LineLength(kEarlyLine[3] * multiplier, frequency, kMaxUpdateSamples);
        lengths[k++] = LineLength(kLateAllpass[3] * multiplier, frequency, lateVecapExtra);
        lengths[k++] = LineLength(kLateLine[3] * multiplier + maxModDelay, frequency, 4);
    }
    for(k = 0; k < 11; ++k) total += lengths[k];
    return total;
}

ReverbHost::ReverbHost()
{
    params.pipeline_state = OALGPU_REVERB_DEVICE_CLEAR;
    params.current_pipeline = 0;
    for(int p = 0; p < 2; ++p)
    {
        auto &P = params.pipe[p];
        // constructor defaults: BiquadFilter passes through, mMixX{1}, mMixY{0}, Mod.Step{1},
        // mFadeSampleCount{1}
        P.filter_lp = P.filter_hp = oalgpu_bq_coeffs{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for(int j = 0; j < 4; ++j) P.t60_hf[j] = P.t60_lf[j] = oalgpu_bq_coeffs{1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        P.mix_x = 1.0f; P.mix_y = 0.0f;
        P.mod_step = 1u;
        P.fade_sample_count = 1u;
    }
}

void ReverbHost::clearPipeline(int which)
{
    auto &P = params.pipe[which];
    std::memset(P.early_delay_tap, 0, sizeof(P.early_delay_tap));
    P.early_delay_coeff[0] = P.early_delay_coeff[1] = 0.0f;
    std::memset(P.late_delay_tap, 0, sizeof(P.late_delay_tap));
    std::memset(P.early_gains_target, 0, sizeof(P.early_gains_target));
    std::memset(P.late_gains_target, 0, sizeof(P.late_gains_target));
    P.mod_step = 1u; P.mod_depth = 0.0f;
    modIndex[which] = 0u;
}

bool ReverbHost::update(const oalgpu_reverb_props &props, float slotGain)
{
    const float frequency = static_cast<float>(sampleRate);

    float hfRatio = props.decay_hf_ratio;
    if(props.decay_hf_limit && props.air_absorption_gain_hf < 1.0f)
    {   // CalcLimitedHfRatio, :910-923
        const float limitRatio = 1.0f / kSpeedOfSound / DecayLength(props.air_absorption_gain_hf, props.decay_time);
        hfRatio = std::min(limitRatio, hfRatio);
    }
    const float lfDecayTime = std::clamp(props.decay_time * props.decay_lf_ratio, kMinDecayTime, kMaxDecayTime);
    const float hfDecayTime = std::clamp(props.decay_time * hfRatio, kMinDecayTime, kMaxDecayTime);

    const bool fullUpdate = params.pipeline_state == OALGPU_REVERB_DEVICE_CLEAR
        || last.density != props.density || last.diffusion != props.diffusion
        || last.decayTime != props.decay_time || last.hfDecayTime != hfDecayTime
        || last.lfDecayTime != lfDecayTime || last.modulationTime != props.modulation_time
        || last.modulationDepth != props.modulation_depth || last.hfReference != props.hf_reference
        || last.lfReference != props.lf_reference;
    if(fullUpdate)
    {
        last.density = props.density; last.diffusion = props.diffusion; last.decayTime = props.decay_time;
        last.hfDecayTime = hfDecayTime; last.lfDecayTime = lfDecayTime;
        last.modulationTime = props.modulation_time; last.modulationDepth = props.modulation_depth;
        last.hfReference = props.hf_reference; last.lfReference = props.lf_reference;

        params.pipeline_state = (params.pipeline_state != OALGPU_REVERB_DEVICE_CLEAR)
            ? OALGPU_REVERB_START_FADE : OALGPU_REVERB_NORMAL;
        params.current_pipeline = !params.current_pipeline;
        params.pipe[!params.current_pipeline].early_delay_coeff[1] = 0.0f;
    }
    auto &P = params.pipe[params.current_pipeline];

    const float densityMult = DelayLengthMult(props.density);

    // ReverbPipeline::updateDelayLine, :1074-1103
    P.early_delay_coeff[1] = props.gain;
    for(int j = 0; j < 4; ++j)
        P.early_delay_tap[j][1] = Trunc((kEarlyTap[j] * densityMult + props.reflections_delay) * frequency);
    for(int j = 0; j < 4; ++j)
    {
        float length = kLateLine[j];
        length -= kLateLine[0];
        length = length * float{1.0f / 4.0f} * densityMult + props.late_reverb_delay;
        P.late_delay_tap[j][1] = Trunc(length * frequency);
    }

    // update3DPanning, :1151-1220
    {
        const float gain = slotGain * kReverbBoost;
        float earlymat[4][4], latemat[4][4];
        TransformFromVector(props.reflections_pan, earlymat);
        TransformFromVector(props.late_reverb_pan, latemat);
        const float (*up)[OALGPU_MAX_AMBI_CHANNELS] = upmix ? firstOrderUp : nullptr;
        PanGains(kEarlyA2B, earlymat, props.reflections_gain * gain, numLines, up, P.early_gains_target);
        PanGains(kLateA2B, latemat, props.late_reverb_gain * gain, numLines, up, P.late_gains_target);
    }

    // master filters, :1298-1309
    const float hf0norm = std::min(props.hf_reference / frequency, 0.49f);
    const float lf0norm = std::min(props.lf_reference / frequency, 0.49f);
    ShelfCoeffs(OALGPU_BIQUAD_HIGHSHELF, hf0norm, props.gain_hf, &P.filter_lp);
    ShelfCoeffs(OALGPU_BIQUAD_LOWSHELF, lf0norm, props.gain_lf, &P.filter_hp);

    if(fullUpdate)
    {
        // EarlyReflections::updateLines, :944-966
        P.early_ap_coeff = props.diffusion * props.diffusion * kInvSqrt2;
        for(int j = 0; j < 4; ++j) P.early_ap_offset[j] = Trunc(kEarlyAllpass[j] * densityMult * frequency);
        for(int j = 0; j < 4; ++j) P.early_offset[j] = Trunc(kEarlyLine[j] * densityMult * frequency);
        {
            const float length = Reduce4(kEarlyLine) / 4.0f * densityMult;
            P.early_coeff = DecayCoeff(length, props.decay_time);
        }
        // CalcMatrixCoeffs, :895-905
        {
            const float t = props.diffusion * std::atan(kSqrt3);
            P.mix_x = std::cos(t);
            P.mix_y = std::sin(t) / kSqrt3;
        }
        // Modulation::updateModulator, :972-1002
        P.mod_step = std::max(RoundU(float(kModFracOne) / (frequency * props.modulation_time)), 1u);
        if(props.modulation_time >= kDefaultModulationTime)
            P.mod_depth = kModulationDepthCoeff / 4.0f * kDefaultModulationTime * props.modulation_depth * frequency;
        else
            P.mod_depth = kModulationDepthCoeff / 4.0f * props.modulation_time * props.modulation_depth * frequency;
        // LateReverb::updateLines, :1005-1070
        {
            const float normWeight = frequency / kMaxHFReference;
            const float decayTimeWeighted = lf0norm * normWeight * lfDecayTime
                + (hf0norm - lf0norm) * normWeight * props.decay_time
                + (1.0f - hf0norm * normWeight) * hfDecayTime;
            const float a = DecayCoeff(kLateDelayAverage * densityMult, decayTimeWeighted);
            P.late_density_gain = std::sqrt(1.0f - a * a);                  // CalcDensityGain :876-892
            P.late_ap_coeff = props.diffusion * props.diffusion * kInvSqrt2;
            for(int j = 0; j < 4; ++j) P.late_ap_offset[j] = Trunc(kLateAllpass[j] * densityMult * frequency);
            float lengths[4];
            for(int j = 0; j < 4; ++j) lengths[j] = kLateLine[j] * densityMult;
            for(int j = 0; j < 4; ++j) P.late_offset[j] = std::max(Trunc(lengths[j] * frequency + 0.5f), 1u) - 1u;
            const float moddepth = P.mod_depth / frequency;
            for(int j = 0; j < 4; ++j)
                lengths[j] = Lerp(kLateAllpass[j], kLateAllpassAverage, props.diffusion) * densityMult + moddepth + lengths[j];
            for(int j = 0; j < 4; ++j)
            {   // T60Filter::calcCoeffs, :930-941
                const float mfGain = DecayCoeff(lengths[j], props.decay_time);
                const float lfGain = DecayCoeff(lengths[j], lfDecayTime) / mfGain;
                const float hfGain = DecayCoeff(lengths[j], hfDecayTime) / mfGain;
                P.t60_mid_gain[j] = mfGain;
                ShelfCoeffs(OALGPU_BIQUAD_LOWSHELF, lf0norm, lfGain, &P.t60_lf[j]);
                ShelfCoeffs(OALGPU_BIQUAD_HIGHSHELF, hf0norm, hfGain, &P.t60_hf[j]);
            }
        }
    }

    // fade length, :1329-1350
    const float decayBase = slotGain * props.gain * props.late_reverb_gain;
    const float decayDiff = kDecayGain / std::max(decayBase, kDecayGain);
    const float diffTime = !(decayDiff < 1.0f) ? 0.0f : (std::log10(decayDiff) * (20.0f / -60.0f) * props.decay_time);
    const float decaySamples = (props.reflections_delay + props.late_reverb_delay + diffTime) * frequency;
    P.fade_sample_count = static_cast<uint32_t>(static_cast<size_t>(std::min(decaySamples, 100000.0f)));
    return fullUpdate;
}

bool ReverbHost::install(const oalgpu_reverb_params &src)
{
    bool full = false;
    if(src.current_pipeline != params.current_pipeline)
    {
        full = true;
        params.pipeline_state = src.pipeline_state;
        params.current_pipeline = src.current_pipeline;
        params.pipe[!params.current_pipeline].early_delay_coeff[1] = 0.0f;
    }
    auto &P = params.pipe[params.current_pipeline];
    const auto &S = src.pipe[params.current_pipeline];
    // keep what process() owns: the current taps / coefficient
    uint32_t etap0[4], ltap0[4];
    for(int j = 0; j < 4; ++j) { etap0[j] = P.early_delay_tap[j][0]; ltap0[j] = P.late_delay_tap[j][0]; }
    const float ecoeff0 = P.early_delay_coeff[0];
    P = S;
    for(int j = 0; j < 4; ++j) { P.early_delay_tap[j][0] = etap0[j]; P.late_delay_tap[j][0] = ltap0[j]; }
    P.early_delay_coeff[0] = ecoeff0;
    return full;
}

ReverbHost::Step ReverbHost::begin(uint32_t n)
{
    Step s{};
    s.current = params.current_pipeline;
    s.offset = offset;
    s.modIndex[0] = modIndex[0]; s.modIndex[1] = modIndex[1];
    const int old = !params.current_pipeline;
    params.pipeline_state = std::max<int32_t>(OALGPU_REVERB_FADING, params.pipeline_state);   // :1840
    s.oldMode = 0;
    if(params.pipeline_state != OALGPU_REVERB_NORMAL)
    {   // :1847-1880
        if(params.pipeline_state == OALGPU_REVERB_CLEANUP)
        {
            clearPipeline(old);
            params.pipeline_state = OALGPU_REVERB_NORMAL;
            s.oldMode = 3;
        }
        else
        {
            auto &O = params.pipe[old];
            if(n >= O.fade_sample_count)
            {
                std::memset(O.early_gains_target, 0, sizeof(O.early_gains_target));
                std::memset(O.late_gains_target, 0, sizeof(O.late_gains_target));
                O.fade_sample_count = 0;
                params.pipeline_state = OALGPU_REVERB_CLEANUP;
                s.oldMode = 2;
            }
            else
            {
                O.fade_sample_count -= n;
                s.oldMode = 1;
            }
        }
    }
    return s;
}

void ReverbHost::finish(const Step &s, uint32_t n)
{
    auto handOver = [this, n](int which)
    {   // processEarly :1577-1585, processLate :1757-1759, calcDelays :1680
        auto &P = params.pipe[which];
        P.early_delay_coeff[0] = P.early_delay_coeff[1];
        for(int j = 0; j < 4; ++j)
        {
            P.early_delay_tap[j][0] = P.early_delay_tap[j][1];
            P.late_delay_tap[j][0] = P.late_delay_tap[j][1];
        }
        modIndex[which] += P.mod_step * n;
    };
    handOver(s.current);
    if(s.oldMode == 1 || s.oldMode == 2) handOver(!s.current);
    offset += n;
}

} // namespace oalgpu
