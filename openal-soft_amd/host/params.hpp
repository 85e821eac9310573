// Host-side parameter preparation: the scalar per-voice work the reference's ALU does with
// libm before a voice mixes (resampler state, biquad coefficient design).  See params.cpp.
#pragma once
#include "../../include/oalgpu.h"

namespace oalgpu {

// PrepareResampler / BsincPrepare / SelectResampler, alc/alu.cpp:140-281.
void PrepareResampler(int resampler, uint32_t increment, oalgpu_interp_state *out);

// BiquadFilter::SetParams coefficient design, core/filters/biquad.cpp:48-129, for
// setParamsFromSlope (biquad.h:172-177).  Writes b0,b1,b2,a1,a2 (a0-normalised) to coeffs[5].
void DesignBiquadFromSlope(int type, float f0norm, float gain, float slope, float coeffs[5]);
// setParamsFromBandwidth (biquad.h:111-113) and SetParams itself (rcpQ given)
void DesignBiquadFromBandwidth(int type, float f0norm, float gain, float bandwidth, float coeffs[5]);
void DesignBiquad(int type, float f0norm, float gain, float rcpQ, float coeffs[5]);

// check_set (biquad.cpp:38-43) over the 5 coefficients + BiquadInterpFilter::setParams state
// machine (:131-149), applied to a host-side filter image.
void ApplyBiquadTarget(oalgpu_biquad *f, const float coeffs[5]);

// BandSplitter::init, core/filters/splitter.cpp:14-26.
float SplitterCoeff(float f0norm);

// NfcFilter (core/filters/nfc.cpp:56-219): the sections of order 1..4 after init(w1) and
// adjust(w0).  a[o] = {a0, a1..ao}, b[o] = {unused, b1..bo}, baseGain[o] = mBaseGain.
struct NfcDesign { float a[5][5], b[5][5], baseGain[5]; };
void NfcInit(float w1, NfcDesign &d);
void NfcAdjust(float w0, NfcDesign &d);

} // namespace oalgpu
