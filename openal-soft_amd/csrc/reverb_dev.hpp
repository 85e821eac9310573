// Device-side layout of one EAX reverb instance, shared by reverb_kernels.hip and reverb_api.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/oalgpu.h"

namespace oalgpu {

// The all-pass stages run in LDS windows sized for sample rates up to 48 kHz: the longest early
// all-pass delay is 6.534130e-4 s x 10 (density multiplier) x 48000 = 313 samples, the longest
// late vector all-pass delay 1.618280e-3 x 10 x 48000 = 776 (reverb.cpp:195-197,229-231).
constexpr uint32_t kRvMaxSampleRate = 48000;
constexpr uint32_t kRvMaxEarlyApOffset = 320;
constexpr uint32_t kRvMaxLateApOffset = 784;

// what process() keeps per pipeline besides the delay lines and the scalars of
// oalgpu_reverb_pipeline: Gains[j].Current and the biquad delay elements
struct RvPipeState {
    float earlyCur[4][OALGPU_MAX_AMBI_CHANNELS];
    float lateCur[4][OALGPU_MAX_AMBI_CHANNELS];
    float z[4][8];                 // per line: mFilter Lp z1,z2 / Hp z1,z2, T60 HF z1,z2 / LF z1,z2
    float split[2][4][4];          // mAmbiSplitter[early | late][row]: lpZ1, lpZ2, apZ1 (MixOutAmbiUp)
};

struct RvLines {                   // per pipeline; strides = samples per line (powers of two)
    float *lateIn, *eap, *edelay, *vap, *ldelay;
    uint32_t lateInStride, eapStride, edelayStride, vapStride, ldelayStride;
};

struct RvLayout {
    oalgpu_reverb_pipeline *pipe;  // [2]
    RvPipeState *state;            // [2]
    float *mainDelay;
    uint32_t mainStride;
    RvLines lines[2];
    float *earlyOut, *lateOut;     // [2][4][1024] scratch: mEarlySamples / mLateSamples per pipeline
    const float *cubic;            // gCubicTable, 513 floats
    const float *wetIn;            // 4 x 1024
    float *outLines;               // nlines x 1024
    uint32_t nlines, n;
    uint32_t offset;               // mOffset (low 32 bits: every line length is a power of two)
    uint32_t modIndex[2];          // mLate.Mod.Index at the start of this block
    int current;                   // mCurrentPipeline
    int oldMode;                   // ReverbHost::Step::oldMode
    uint32_t upmix;                // mUpmixOutput: MixOutAmbiUp instead of MixOutPlain
    float orderScale[2];           // mOrderScales[0], [1]
    float splitCoeff;              // BandSplitter{device->mXOverFreq / frequency}.mCoeff
    uint32_t fast;                 // OALGPU_MATH_FAST: the dual-biquad sections as block scans (rounding-level differences)
    unsigned long long *stamps;    // measurement aid (oalgpu_reverb_debug_enable_phase_times), null in production: [4 roles][8 sub-blocks][8]
};

// several instances adding into the same target lines in one launch (see ReverbProcessBody)
constexpr uint32_t kRvBatchMax = 8;
struct RvBatch {
    RvLayout r[kRvBatchMax];
    uint32_t count;
    uint32_t *ticket;              // device word, 0 between launches
};
static_assert(sizeof(RvBatch) <= 3800, "RvBatch travels as a kernel argument");

void LaunchReverbProcess(hipStream_t s, const RvLayout &L);
void LaunchReverbBatch(hipStream_t s, const RvBatch &B);
void LaunchReverbInstall(hipStream_t s, oalgpu_reverb_pipeline *dst, const oalgpu_reverb_pipeline &src);

} // namespace oalgpu

// api.hip: process `count` reverbs (<= kRvBatchMax) that share `out_lines_dev` and a stream as one
// launch; `ticket_dev` is a zeroed device word owned by the caller
int oalgpu_reverb_process_batch_device(oalgpu_reverb *const *revs, const float *const *wet_in_dev, uint32_t count,
    float *out_lines_dev, uint32_t n, void *hip_stream, uint32_t *ticket_dev);
