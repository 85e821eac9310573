// C-ABI of the small EffectStates (include/oalgpu.h: oalgpu_effect_*): deviceUpdate / update as the reference does
// them on its mixer thread with libm -- biquad design, the echo's delays, the modulator's carrier period -- and the
// scalar bookkeeping process() performs (delay-line offset, carrier index); the block itself is one launch
// (effects_kernels.hip).
#include "api_util.hpp"
#include "kernels.hpp"
#include "../host/params.hpp"
#include "../host/tables.hpp"

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
    // ---- chorus .. frequency shifter (effects2_kernels.hip) ----
    DevBuf<Fx2State> st2;
    DevBuf<float> cubic, upTgt, window;
    DevBuf<double> fsIn;
    DevBuf<FsPair> fsOutFifo, fsAccum, fsOutdata, fsTw, fsPhase;
    DevBuf<float> psRing, psPhase, psAccum, psOutFifo, psRows, psTw;     // psRing / psPhase: two copies, read [parity], written [parity ^ 1]
    Fx2Launch G{};
    // ChorusState::mOffset / mLfoOffset / mLfoRange / mLfoDisp (chorus.cpp:83-88)
    uint32_t chOffset{0}, lfoOffset{0}, lfoRange{1}, lfoDisp{0};
    // VmorpherState::mIndex (vmorpher.cpp:152); FshifterState::mCount / mPos / mChans[c].mPhase (fshifter.cpp:100-113)
    uint32_t vmIndex{0};
    uint32_t fsCount{0}, fsPos{768}, fsPhase4[4]{};
    // PshifterState::mCount / mPos (pshifter.cpp:86-87)
    uint32_t psCount{0}, psPos{1024 - 128}, psParity{0};
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

bool IsFx2(int kind) { return kind >= OALGPU_EFFECT_CHORUS && kind <= OALGPU_EFFECT_PSHIFTER; }

// float2int / float2uint (common/alnumeric.h): truncation
int32_t TruncI(float f) { return static_cast<int32_t>(f); }
uint32_t TruncU(float f) { return static_cast<uint32_t>(static_cast<int64_t>(f)); }
// fastf2u: round to nearest even (cvtss2si)
uint32_t RoundU(float f) { return static_cast<uint32_t>(static_cast<int32_t>(std::lrintf(f))); }

template<typename T>
int UploadAt(void *base, size_t offset, const T *src, size_t count)
{
    HIP_TRY(hipMemcpy(static_cast<char*>(base) + offset, src, count * sizeof(T), hipMemcpyHostToDevice));
    return OALGPU_OK;
}

// gHannWindow<1024>, common/hann_window.hpp: sin^2 through double, mirrored
std::vector<float> HannWindow1024()
{
    std::vector<float> win(1024);
    const double scale = 3.14159265358979323846 / double(1024 + 1);
    for(uint32_t i = 0; i < 512; ++i)
    {
        const double v = std::sin((i + 1.0) * scale);
        win[i] = win[1023 - i] = static_cast<float>(v * v);
    }
    return win;
}

#pragma clang fp contract(off)
// cos(pi / 2^i), sin(pi / 2^i) correctly rounded to double (tools/gen_fft_twiddle_roots.py derives them with 60-digit
// arithmetic): what complex_fft's table gArgAngle (common/alcomplex.cpp:84-99) holds
constexpr double kFftRoots[10][2] = {
    {-0x1.0000000000000p+0, 0x0.0p+0},
    {0x0.0p+0, 0x1.0000000000000p+0},
    {0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1},
    {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},
    {0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3},
    {0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4},
    {0x1.ff621e3796d7ep-1, 0x1.91f65f10dd814p-5},
    {0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6},
    {0x1.fff62169b92dbp-1, 0x1.921d1fcdec784p-7},
    {0x1.fffd8858e8a92p-1, 0x1.921f0fe670071p-8},
};

// the twiddle factors complex_fft (common/alcomplex.cpp:105-148) runs through: stage i starts from
// w = polar(1, pi / 2^i) and multiplies u by w once per j
void BuildTwiddles(std::vector<FsPair> &tw)
{
    tw.assign(1024, FsPair{1.0, 0.0});
    for(uint32_t i = 0; i < 10; ++i)
    {
        const uint32_t step2 = 1u << i;
        const FsPair w{kFftRoots[i][0], kFftRoots[i][1]};
        FsPair u = w;
        for(uint32_t j = 1; j < step2; ++j)
        {
            tw[step2 + j] = u;
            const double re = u.x * w.x - u.y * w.y, im = u.x * w.y + u.y * w.x;
            u = FsPair{re, im};
        }
    }
}


// update() of the chorus / distortion / autowah / vocal morpher / frequency shifter states
int UpdateFx2(oalgpu_effect *e, const void *props, const uint32_t *target_channels, const float *gains)
{
    if(!props || !target_channels) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: this effect needs props and targets");
    const float rate = float(e->sampleRate);
    Fx2Launch &G = e->G;
    const bool aformat = e->kind == OALGPU_EFFECT_CHORUS || e->kind == OALGPU_EFFECT_DISTORTION || e->kind == OALGPU_EFFECT_FSHIFTER;
    switch(e->kind)
    {
    case OALGPU_EFFECT_CHORUS:
        {   // ChorusState::update, chorus.cpp:165-251
            const auto &p = *static_cast<const oalgpu_chorus_props*>(props);
            const int32_t mindelay = 24 << 8;                       // MaxResamplerEdge << gCubicTable.sTableBits
            const float stepscale = rate * 256.0f;
            G.chWave = p.waveform;
            G.chDelay = std::max(TruncI(std::round(p.delay * stepscale)), mindelay);
            G.chDepth = std::min(float(G.chDelay) * p.depth, float(G.chDelay - mindelay));
            G.chFeedback = p.feedback;
            if(!(p.rate > 0.0f)) { e->lfoOffset = 0; e->lfoRange = 1; G.lfoScale = 0.0f; e->lfoDisp = 0; }
            else
            {
                const int32_t rangeLimit = 2147483647 / 360 - 180;
                const float range = std::round(rate / p.rate);
                const uint32_t lfoRange = TruncU(std::min(range, float(rangeLimit)));
                e->lfoOffset = e->lfoOffset * lfoRange / e->lfoRange;
                e->lfoRange = lfoRange;
                G.lfoScale = (p.waveform == OALGPU_CHORUS_TRIANGLE) ? 4.0f / float(lfoRange)
                    : 3.14159265358979323846f * 2.0f / float(lfoRange);
                int32_t phase = p.phase;
                if(phase < 0) phase += 360;
                e->lfoDisp = (lfoRange * uint32_t(phase) + 180u) / 360u;
            }
            G.lfoRange = e->lfoRange;
            G.chAvgDelay = (uint32_t(G.chDelay) + 32768u) >> 16;         // (mDelay + MixerFracHalf) >> MixerFracBits
            // how far behind the write position a tap or the feedback reads
            const uint32_t maxTap = (uint32_t(G.chDelay) + uint32_t(std::ceil(G.chDepth)) + 1u) >> 8;
            G.chHist = std::max(maxTap + 3u, G.chAvgDelay);
            if(G.chHist > G.delayMask) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: chorus delay beyond the delay line");
        }
        break;
    case OALGPU_EFFECT_DISTORTION:
        {   // DistortionState::update, distortion.cpp:141-195 (the filters work on the 4x oversampled signal)
            const auto &p = *static_cast<const oalgpu_distortion_props*>(props);
            const float edge = std::min(std::sin(3.14159265358979323846f * 0.5f * p.edge), 0.99f);
            G.edgeCoeff = 2.0f * edge / (1.0f - edge);
            float lp[5], bp[5];
            DesignBiquadFromBandwidth(OALGPU_BIQUAD_LOWPASS, p.lowpass_cutoff / rate * 0.25f, 1.0f, 0.746268656716f, lp);
            const float bandwidth = p.eq_bandwidth / (p.eq_center * 0.67f);
            DesignBiquadFromBandwidth(OALGPU_BIQUAD_BANDPASS, p.eq_center / rate * 0.25f, 1.0f, bandwidth, bp);
            for(uint32_t c = 0; c < 4; ++c)
            {
                if(int rc = UploadAt(e->st2.p, offsetof(Fx2State, lp) + c * sizeof(BiquadState) + offsetof(BiquadState, b0), lp, 5)) return rc;
                if(int rc = UploadAt(e->st2.p, offsetof(Fx2State, bp) + c * sizeof(BiquadState) + offsetof(BiquadState, b0), bp, 5)) return rc;
            }
        }
        break;
    case OALGPU_EFFECT_AUTOWAH:
        {   // AutowahState::update, autowah.cpp:100-122
            const auto &p = *static_cast<const oalgpu_autowah_props*>(props);
            const float release = std::min(std::max(p.release_time, 0.001f), 1.0f);
            G.attackRate = std::exp(-1.0f / (p.attack_time * rate));
            G.releaseRate = std::exp(-1.0f / (release * rate));
            G.resonanceGain = std::sqrt(std::log10(p.resonance) * 10.0f / 3.0f);
            G.peakGain = 1.0f - std::log10(p.peak_gain / 31621.0f);
            G.freqMinNorm = 20.0f / rate;
            G.bandwidthNorm = (2500.0f - 20.0f) / rate;
        }
        break;
    case OALGPU_EFFECT_VMORPHER:
        {   // VmorpherState::update, vmorpher.cpp:234-277; getFiltersByPhoneme :164-226
            const auto &p = *static_cast<const oalgpu_vmorpher_props*>(props);
            const float step = p.rate / rate;
            G.vmStep = RoundU(std::min(std::max(step * 16777216.0f, 0.0f), 16777216.0f - 1.0f));
            G.vmWave = G.vmStep == 0 ? 0 : p.waveform == OALGPU_VMORPHER_SINUSOID ? 1 : p.waveform == OALGPU_VMORPHER_TRIANGLE ? 2 : 3;
            static const float kFreq[5][4] = {{800, 1150, 2900, 3900}, {350, 2000, 2800, 3600}, {270, 2140, 2950, 3900},
                {450, 800, 2830, 3800}, {325, 700, 2700, 3800}};
            static const float kGain[5][4] = {{1.000000f, 0.501187f, 0.025118f, 0.100000f}, {1.000000f, 0.100000f, 0.177827f, 0.009999f},
                {1.000000f, 0.251188f, 0.050118f, 0.050118f}, {1.000000f, 0.281838f, 0.079432f, 0.079432f},
                {1.000000f, 0.158489f, 0.017782f, 0.009999f}};
            const int32_t ph[2] = {p.phoneme_a, p.phoneme_b}, tune[2] = {p.phoneme_a_coarse_tuning, p.phoneme_b_coarse_tuning};
            for(int v = 0; v < 2; ++v)
            {
                const float pitch = std::pow(2.0f, float(tune[v]) / 12.0f);
                for(int k = 0; k < 4; ++k)
                {
                    if(ph[v] >= 0 && ph[v] < 5)
                    {   // FormantFilter(f0norm, gain): mCoeff = tan(pi * f0norm)
                        G.vmG[v * 4 + k] = std::tan(3.14159265358979323846f * ((kFreq[ph[v]][k] * pitch) / rate));
                        G.vmGain[v * 4 + k] = kGain[ph[v]][k];
                    }
                    else { G.vmG[v * 4 + k] = 0.0f; G.vmGain[v * 4 + k] = 1.0f; }       // the other phonemes: FormantFilter{}
                }
            }
            // the copies of the new filters start with cleared histories, for every wet channel
            std::vector<float> zeros(size_t{e->numIn} * 16, 0.0f);
            if(int rc = UploadAt(e->st2.p, offsetof(Fx2State, vmS), zeros.data(), zeros.size())) return rc;
        }
        break;
    case OALGPU_EFFECT_PSHIFTER:
        {   // PshifterState::update, pshifter.cpp:168-199
            const auto &p = *static_cast<const oalgpu_pshifter_props*>(props);
            const int32_t tune = p.coarse_tune * 100 + p.fine_tune;
            const float pitch = std::pow(2.0f, float(tune) / 1200.0f);
            G.psPitchI = RoundU(std::min(std::max(pitch, 0.5f), 2.0f) * 65536.0f);
            G.psPitch = float(G.psPitchI) * (1.0f / 65536.0f);
        }
        break;
    default:
        {   // FshifterState::update, fshifter.cpp:162-214
            const auto &p = *static_cast<const oalgpu_fshifter_props*>(props);
            const float step = p.frequency / rate;
            const uint32_t phaseStep = RoundU(std::min(step, 1.0f) * 65536.0f);
            for(int c = 0; c < 4; ++c) G.fsPhaseStep[c] = phaseStep;
            const int32_t dir[2] = {p.left_direction, p.right_direction};
            for(int side = 0; side < 2; ++side)
                for(int c = side * 2; c < side * 2 + 2; ++c)
                {
                    if(dir[side] == OALGPU_FSHIFTER_DOWN) G.fsSign[c] = -1.0;
                    else if(dir[side] == OALGPU_FSHIFTER_UP) G.fsSign[c] = 1.0;
                    else { e->fsPhase4[c] = 0; G.fsPhaseStep[c] = 0; }
                }
        }
        break;
    }
    for(uint32_t ch = 0; ch < kFxMaxIn; ++ch) { G.target[ch] = OALGPU_INVALID_CHANNEL; G.tgtGain[ch] = 0.0f; }
    const bool pshift = e->kind == OALGPU_EFFECT_PSHIFTER;
    const uint32_t chans = aformat ? std::min(e->numIn, 4u) : pshift ? std::min(e->numIn, 9u) : e->numIn;
    for(uint32_t ch = 0; ch < chans; ++ch) G.target[ch] = target_channels[ch];
    if((aformat || pshift) && G.upsample)
    {   // UpsampleParams::mTargetGains = ComputePanGains(target.Main, AmbiScale::FirstOrderUp[ch] (pitch shifter: SecondOrderUp[ch]),
        // gain): gains[4 or 9][num_out_lines]
        const uint32_t rows = pshift ? 9u : 4u;
        std::vector<float> up(9 * 32, 0.0f);
        for(uint32_t ch = 0; ch < rows; ++ch)
            for(uint32_t l = 0; l < e->nlines; ++l) up[ch * 32 + l] = gains[size_t{ch} * e->nlines + l];
        HIP_TRY(e->upTgt.upload(up.data(), up.size()));
    }
    else
        for(uint32_t ch = 0; ch < chans; ++ch) G.tgtGain[ch] = gains[ch];
    e->updated = true;
    return OALGPU_OK;
}

} // namespace

extern "C" {

int oalgpu_effect_create(int device, int math_mode, int kind, uint32_t sample_rate, uint32_t num_in_channels,
    uint32_t num_out_lines, oalgpu_effect **out)
{
    if(!out || kind < OALGPU_EFFECT_EQUALIZER || kind > OALGPU_EFFECT_PSHIFTER || sample_rate < 8000 || num_in_channels < 1
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
    if(IsFx2(kind))
    {
        const float f = float(sample_rate);
        HIP_TRY(e->st2.alloc(1)); HIP_TRY(e->st2.zero());
        HIP_TRY(e->upTgt.alloc(9 * 32)); HIP_TRY(e->upTgt.zero());
        Fx2Launch &G = e->G;
        G.kind = kind; G.numIn = num_in_channels; G.nlines = num_out_lines; G.st = e->st2.p; G.upTgt = e->upTgt.p;
        for(uint32_t c = 0; c < kFxMaxIn; ++c) { G.target[c] = OALGPU_INVALID_CHANNEL; G.tgtGain[c] = 0.0f; }
        G.hfScale[0] = G.hfScale[1] = 1.0f;
        G.lfoRange = 1;
        if(kind == OALGPU_EFFECT_CHORUS)
        {   // ChorusState::deviceUpdate, chorus.cpp:129-163: four lines of NextPowerOf2(2 * max(ChorusMaxDelay, FlangerMaxDelay) * rate + 1)
            const uint32_t len = NextPow2(TruncU(0.016f * 2.0f * f) + 1u);
            HIP_TRY(e->delay.alloc(size_t{len} * 4)); HIP_TRY(e->delay.zero());
            HIP_TRY(e->cubic.alloc(kFineCubicSteps * 2 + 1));
            HIP_TRY(e->cubic.upload(GetFineCubicFilter(), kFineCubicSteps * 2 + 1));
            G.delay = e->delay.p; G.delayMask = len - 1u; G.cubic = e->cubic.p;
        }
        if(kind == OALGPU_EFFECT_DISTORTION)
        {
            BiquadState ident{}; ident.b0 = 1.0f;
            BiquadState eight[8]; for(auto &b : eight) b = ident;
            if(int rc = UploadAt(e->st2.p, offsetof(Fx2State, lp), eight, 8)) return rc;
        }
        if(kind == OALGPU_EFFECT_AUTOWAH)
        {   // AutowahState::deviceUpdate, autowah.cpp:86-98
            G.attackRate = 1.0f; G.releaseRate = 1.0f; G.resonanceGain = 10.0f; G.peakGain = 4.5f;
            G.freqMinNorm = 4.5e-4f; G.bandwidthNorm = 0.05f;
        }
        if(kind == OALGPU_EFFECT_VMORPHER)
            for(int k = 0; k < 8; ++k) { G.vmG[k] = 0.0f; G.vmGain[k] = 1.0f; }       // FormantFilter{}
        if(kind == OALGPU_EFFECT_FSHIFTER)
        {   // FshifterState::deviceUpdate, fshifter.cpp:133-160; the Hann window of common/hann_window.hpp
            HIP_TRY(e->fsIn.alloc(4 * 1024)); HIP_TRY(e->fsIn.zero());
            HIP_TRY(e->fsOutFifo.alloc(4 * 256)); HIP_TRY(e->fsOutFifo.zero());
            HIP_TRY(e->fsAccum.alloc(4 * 1024)); HIP_TRY(e->fsAccum.zero());
            HIP_TRY(e->fsOutdata.alloc(4 * OALGPU_BUFFER_LINE_SIZE)); HIP_TRY(e->fsOutdata.zero());
            std::vector<FsPair> tw; BuildTwiddles(tw);
            HIP_TRY(e->fsTw.alloc(tw.size()));
            HIP_TRY(e->fsTw.upload(tw.data(), tw.size()));
            // cos / sin of phase_idx * (pi*2 / MixerFracOne), fshifter.cpp:323-326: one entry per phase index
            std::vector<FsPair> ph(65536);
            for(uint32_t i = 0; i < 65536; ++i)
            {
                const double phase = i * (3.14159265358979323846 * 2.0 / 65536.0);
                ph[i] = FsPair{std::cos(phase), std::sin(phase)};
            }
            HIP_TRY(e->fsPhase.alloc(ph.size()));
            HIP_TRY(e->fsPhase.upload(ph.data(), ph.size()));
            const std::vector<float> win = HannWindow1024();
            HIP_TRY(e->window.alloc(win.size()));
            HIP_TRY(e->window.upload(win.data(), win.size()));
            G.fsIn = e->fsIn.p; G.fsOutFifo = e->fsOutFifo.p; G.fsAccum = e->fsAccum.p; G.fsOutdata = e->fsOutdata.p;
            G.fsTw = e->fsTw.p; G.fsPhase = e->fsPhase.p; G.fsWindow = e->window.p;
            for(int c = 0; c < 4; ++c) G.fsSign[c] = 1.0;
        }
        if(kind == OALGPU_EFFECT_PSHIFTER)
        {   // PshifterState::deviceUpdate, pshifter.cpp:128-166: up to second order; pitch 1
            if(num_in_channels > 9) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_create: the pitch shifter works on up to 9 channels (second order)");
            HIP_TRY(e->psRing.alloc(2 * 9 * 1024)); HIP_TRY(e->psRing.zero());
            HIP_TRY(e->psPhase.alloc(2 * 2 * 513)); HIP_TRY(e->psPhase.zero());
            HIP_TRY(e->psAccum.alloc(9 * 1024)); HIP_TRY(e->psAccum.zero());
            HIP_TRY(e->psOutFifo.alloc(9 * 128)); HIP_TRY(e->psOutFifo.zero());
            HIP_TRY(e->psRows.alloc(9 * OALGPU_BUFFER_LINE_SIZE)); HIP_TRY(e->psRows.zero());
            std::vector<float> tw(2 * 512);
            for(uint32_t m = 0; m < 512; ++m)
            {
                const double a = -2.0 * 3.14159265358979323846 * double(m) / 1024.0;
                tw[2 * m] = static_cast<float>(std::cos(a)); tw[2 * m + 1] = static_cast<float>(std::sin(a));
            }
            HIP_TRY(e->psTw.alloc(tw.size())); HIP_TRY(e->psTw.upload(tw.data(), tw.size()));
            const std::vector<float> win = HannWindow1024();
            HIP_TRY(e->window.alloc(win.size())); HIP_TRY(e->window.upload(win.data(), win.size()));
            G.psAccum = e->psAccum.p; G.psOutFifo = e->psOutFifo.p; G.psRows = e->psRows.p;
            G.psTw = e->psTw.p; G.psWindow = e->window.p;
            G.psPitchI = 65536u; G.psPitch = 1.0f;
        }
        *out = e.release();
        return OALGPU_OK;
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
    (void)UseDevice(e->device);          // (a resident voice kernel on the device is told to leave first: it would sit out the synchronisation until its watchdog)
    (void)hipDeviceSynchronize();
    delete e;
}

int oalgpu_effect_update(oalgpu_effect *e, const void *props, const uint32_t *target_channels, const float *gains)
{
    if(!e || !gains) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_update: null argument");
    if(int rc = UseDevice(e->device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    const float rate = float(e->sampleRate);
    if(IsFx2(e->kind)) return UpdateFx2(e, props, target_channels, gains);
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

/* deviceUpdate on a device above first order (chorus.cpp:143-162 and alike): mUpsampler */
int oalgpu_effect_set_upsampler(oalgpu_effect *e, const float order_scales[2], float xover_norm)
{
    if(!e) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(!(e->kind == OALGPU_EFFECT_CHORUS || e->kind == OALGPU_EFFECT_DISTORTION || e->kind == OALGPU_EFFECT_FSHIFTER
        || e->kind == OALGPU_EFFECT_PSHIFTER))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_set_upsampler: only the chorus, the distortion and the frequency / pitch shifters up-sample");
    if(e->nlines > 32) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_set_upsampler: at most 32 output lines");
    if(int rc = UseDevice(e->device)) return rc;
    HIP_TRY(hipDeviceSynchronize());
    Fx2Launch &G = e->G;
    G.upsample = order_scales ? 1 : 0;
    if(order_scales)
    {
        G.hfScale[0] = order_scales[0]; G.hfScale[1] = order_scales[1];
        G.splitCoeff = SplitterCoeff(xover_norm);
    }
    // a fresh BandSplitter and zeroed gains, as deviceUpdate leaves them
    std::vector<float> zeros(9 * 32 + 9 * 3, 0.0f);
    if(int rc = UploadAt(e->st2.p, offsetof(Fx2State, upCur), zeros.data(), zeros.size())) return rc;
    e->updated = false;
    return OALGPU_OK;
}

} // extern "C"

namespace {
int ProcessFx2(oalgpu_effect *e, hipStream_t stream, const float *wet_in_dev, float *out_lines_dev, uint32_t n)
{
    Fx2Launch G = e->G;
    G.wetIn = wet_in_dev; G.outLines = out_lines_dev; G.n = n;
    uint32_t lds = 0;
    if(e->kind == OALGPU_EFFECT_CHORUS)
    {
        G.offset = e->chOffset;
        G.lfoStart[0] = e->lfoOffset;
        G.lfoStart[1] = (e->lfoOffset + e->lfoDisp) % e->lfoRange;
        lds = (6u * OALGPU_BUFFER_LINE_SIZE + G.chHist + OALGPU_BUFFER_LINE_SIZE) * sizeof(float);
        if(lds > 65536u) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_process: chorus delay too long for the workgroup's LDS");
    }
    if(e->kind == OALGPU_EFFECT_VMORPHER) G.vmIndex = e->vmIndex;
    if(e->kind == OALGPU_EFFECT_FSHIFTER)
    {
        G.fsCount = e->fsCount; G.fsPos = e->fsPos;
        for(int c = 0; c < 4; ++c) G.fsPhaseIdx[c] = e->fsPhase4[c];
    }
    if(e->kind == OALGPU_EFFECT_PSHIFTER)
    {
        const uint32_t p = e->psParity;
        G.psRingIn = e->psRing.p + size_t{p} * 9 * 1024; G.psRingOut = e->psRing.p + size_t{p ^ 1u} * 9 * 1024;
        G.psPhaseIn = e->psPhase.p + size_t{p} * 2 * 513; G.psPhaseOut = e->psPhase.p + size_t{p ^ 1u} * 2 * 513;
        G.psCount = e->psCount; G.psPos = e->psPos;
    }
    LaunchEffect2(stream, G, lds);
    HIP_TRY(hipGetLastError());
    // the scalars process() moves on (chorus.cpp:283,391; vmorpher.cpp:293-294; fshifter.cpp:228-259,330-336)
    if(e->kind == OALGPU_EFFECT_CHORUS)
    {
        e->chOffset += n;
        e->lfoOffset = (e->lfoOffset + n) % e->lfoRange;
    }
    if(e->kind == OALGPU_EFFECT_PSHIFTER)
    {
        e->psParity ^= 1u;
        for(uint32_t base = 0; base < n;)
        {
            const uint32_t todo = std::min(128u - e->psCount, n - base);
            e->psCount += todo; base += todo;
            if(e->psCount < 128u) break;
            e->psCount = 0; e->psPos = (e->psPos + 128u) & 1023u;
        }
    }
    if(e->kind == OALGPU_EFFECT_VMORPHER)
        for(uint32_t base = 0; base < n; base += 256u)
            e->vmIndex = (e->vmIndex + G.vmStep * std::min(256u, n - base)) & 0xffffffu;
    if(e->kind == OALGPU_EFFECT_FSHIFTER)
    {
        for(uint32_t base = 0; base < n;)
        {
            const uint32_t todo = std::min(256u - e->fsCount, n - base);
            e->fsCount += todo; base += todo;
            if(e->fsCount < 256u) break;
            e->fsCount = 0; e->fsPos = (e->fsPos + 256u) & 1023u;
        }
        for(int c = 0; c < 4; ++c) e->fsPhase4[c] = (e->fsPhase4[c] + n * G.fsPhaseStep[c]) & 65535u;
    }
    return OALGPU_OK;
}
} // namespace

int oalgpu_effect_process_device(oalgpu_effect *e, void *hip_stream, const float *wet_in_dev, float *out_lines_dev, uint32_t n)
{
    if(!e || !wet_in_dev || !out_lines_dev || n == 0 || n > OALGPU_BUFFER_LINE_SIZE)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_process: bad arguments");
    if(!e->updated) return Fail(OALGPU_ERR_INVALID, "oalgpu_effect_process: no update() yet");
    if(int rc = UseDevice(e->device)) return rc;
    if(IsFx2(e->kind)) return ProcessFx2(e, static_cast<hipStream_t>(hip_stream), wet_in_dev, out_lines_dev, n);
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
