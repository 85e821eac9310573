// C-ABI of the EAX reverb (include/oalgpu.h): instance set-up (ReverbState::deviceUpdate,
// alc/effects/reverb.cpp:822-852), the host-side update() (host/reverb_params.cpp) and the
// per-block launch.
#include "api_util.hpp"
#include "reverb_dev.hpp"
#ifdef OALGPU_MEASUREMENT
#include "../../tools/measure/oalgpu_measure.h"
#endif

#include <cstdlib>
#include <cstring>
#include <memory>

#include "../host/params.hpp"
#include "../host/reverb_params.hpp"
#include "../host/tables.hpp"

using namespace oalgpu;

struct oalgpu_reverb {
    int device{-1};
    hipStream_t stream{nullptr};
    ReverbHost host;
    bool dirty[2]{true, true};              // host mirror of pipeline p differs from the device copy
    uint32_t lengths[11]{};
    size_t total{0};
    DevBuf<float> samples, scratch, cubic, hostIn, hostOut;
    DevBuf<oalgpu_reverb_pipeline> pipe;
    DevBuf<RvPipeState> state;
    DevBuf<unsigned long long> stamps;      // measurement aid (oalgpu_reverb_debug_enable_phase_times)
    bool upmixStale{false};                 // set_upmix changed the output mode: the panning gains in use were designed for the other one
    RvLayout L{};
};

namespace {

int CheckOffsets(const oalgpu_reverb_pipeline &P)
{
    for(int j = 0; j < 4; ++j)
    {
        if(P.early_ap_offset[j] < 1 || P.early_ap_offset[j] > kRvMaxEarlyApOffset || P.late_ap_offset[j] < 1
            || P.late_ap_offset[j] > kRvMaxLateApOffset || P.late_offset[j] < 1)
            return Fail(OALGPU_ERR_INVALID, "reverb: delay offsets outside what the kernel's LDS windows hold");
    }
    if(P.late_ap_offset[0] > P.late_ap_offset[1] || P.late_ap_offset[0] > P.late_ap_offset[2]
        || P.late_ap_offset[0] > P.late_ap_offset[3])
        return Fail(OALGPU_ERR_INVALID, "reverb: the first late all-pass offset must be the shortest");
    return OALGPU_OK;
}

} // namespace

extern "C" {

int oalgpu_reverb_create(int device, uint32_t sample_rate, uint32_t num_out_lines, oalgpu_reverb **out)
{
    if(!out || sample_rate < 8000 || num_out_lines < 1 || num_out_lines > OALGPU_MAX_AMBI_CHANNELS)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_create: bad arguments");
    *out = nullptr;
    auto r = std::make_unique<oalgpu_reverb>();
    r->device = device;
    r->host.sampleRate = sample_rate;
    r->host.numLines = num_out_lines;
    r->total = ReverbLineLengths(static_cast<float>(sample_rate), r->lengths);
    if(device < 0) { *out = r.release(); return OALGPU_OK; }       // parameter-only instance

    if(sample_rate > kRvMaxSampleRate)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_create: sample rates above 48 kHz are not supported");
    if(int rc = UseDevice(device)) return rc;
    HIP_TRY(r->samples.alloc(r->total)); HIP_TRY(r->samples.zero());
    HIP_TRY(r->scratch.alloc(size_t{2} * 2 * 4 * OALGPU_BUFFER_LINE_SIZE)); HIP_TRY(r->scratch.zero());
    HIP_TRY(r->cubic.alloc(kFineCubicSteps * 2 + 1));
    HIP_TRY(r->cubic.upload(GetFineCubicFilter(), kFineCubicSteps * 2 + 1));
    HIP_TRY(r->pipe.alloc(2)); HIP_TRY(r->pipe.upload(r->host.params.pipe, 2));
    HIP_TRY(r->state.alloc(2)); HIP_TRY(r->state.zero());
    HIP_TRY(r->hostIn.alloc(size_t{4} * OALGPU_BUFFER_LINE_SIZE));
    HIP_TRY(r->hostOut.alloc(size_t{num_out_lines} * OALGPU_BUFFER_LINE_SIZE));
    r->dirty[0] = r->dirty[1] = false;

    RvLayout &L = r->L;
    L.pipe = r->pipe.p; L.state = r->state.p;
    float *at = r->samples.p;
    size_t k = 0;
    L.mainDelay = at; L.mainStride = r->lengths[k] / 4u; at += r->lengths[k++];
    for(int p = 0; p < 2; ++p)
    {
        RvLines &ln = L.lines[p];
        ln.lateIn = at; ln.lateInStride = r->lengths[k] / 4u; at += r->lengths[k++];
        ln.eap = at; ln.eapStride = r->lengths[k] / 4u; at += r->lengths[k++];
        ln.edelay = at; ln.edelayStride = r->lengths[k] / 4u; at += r->lengths[k++];
        ln.vap = at; ln.vapStride = r->lengths[k] / 4u; at += r->lengths[k++];
        ln.ldelay = at; ln.ldelayStride = r->lengths[k] / 4u; at += r->lengths[k++];
    }
    L.earlyOut = r->scratch.p;
    L.lateOut = r->scratch.p + size_t{2} * 4 * OALGPU_BUFFER_LINE_SIZE;
    L.cubic = r->cubic.p;
    L.nlines = num_out_lines;
    L.stamps = nullptr;
    *out = r.release();
    return OALGPU_OK;
}

/* deviceUpdate on a device above first order (reverb.cpp:835-851): mUpmixOutput, mOrderScales, the pipelines'
 * band splitters.  The constants of the ambisonic layer come from the caller: order_scales[0..1] =
 * AmbiScale::GetHFOrderScales(1, device order, 2D mixing), first_order_up = AmbiScale::FirstOrderUp (4 x 25,
 * core/ambidefs.cpp), xover_norm = device->mXOverFreq / frequency.  NULL order_scales: back to MixOutPlain.
 * Takes effect with the next oalgpu_reverb_update (which designs the panning gains). */
int oalgpu_reverb_set_upmix(oalgpu_reverb *r, const float order_scales[2], const float *first_order_up, float xover_norm)
{
    if(!r) return Fail(OALGPU_ERR_INVALID, "null argument");
    r->upmixStale = true;                   // process() is refused until an update() has designed gains for this mode
    if(!order_scales || !first_order_up)
    {
        r->host.upmix = false; r->L.upmix = 0u;
        return OALGPU_OK;
    }
    if(!(xover_norm > 0.0f && xover_norm < 0.5f)) return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_set_upmix: 0 < xover_norm < 0.5");
    r->host.upmix = true;
    std::memcpy(r->host.firstOrderUp, first_order_up, sizeof(r->host.firstOrderUp));
    r->L.upmix = 1u;
    r->L.orderScale[0] = order_scales[0]; r->L.orderScale[1] = order_scales[1];
    r->L.splitCoeff = SplitterCoeff(xover_norm);
    return OALGPU_OK;
}

void oalgpu_reverb_destroy(oalgpu_reverb *r)
{
    if(!r) return;
    if(r->device >= 0)
    {
        (void)UseDevice(r->device);       // (a resident voice kernel on the device is told to leave first)
        (void)hipDeviceSynchronize();
    }
    delete r;
}

int oalgpu_reverb_set_stream(oalgpu_reverb *r, void *hip_stream)
{
    if(!r) return Fail(OALGPU_ERR_INVALID, "null argument");
    r->stream = static_cast<hipStream_t>(hip_stream);
    return OALGPU_OK;
}

int oalgpu_reverb_update(oalgpu_reverb *r, const oalgpu_reverb_props *props, float slot_gain)
{
    if(!r || !props) return Fail(OALGPU_ERR_INVALID, "null argument");
    // validate before committing: a block the kernel cannot run must not stay installed
    auto trial = r->host;
    const bool full = trial.update(*props, slot_gain);
    if(r->device >= 0)
    {
        if(int rc = CheckOffsets(trial.params.pipe[trial.params.current_pipeline])) return rc;
        // (the other pipeline is either the one that was current -- validated then -- or still unused)
    }
    r->host = trial;
    r->dirty[r->host.params.current_pipeline] = true;
    if(full) r->dirty[!r->host.params.current_pipeline] = true;
    r->upmixStale = false;
    return OALGPU_OK;
}

int oalgpu_reverb_get_params(oalgpu_reverb *r, oalgpu_reverb_params *out)
{
    if(!r || !out) return Fail(OALGPU_ERR_INVALID, "null argument");
    *out = r->host.params;
    return OALGPU_OK;
}

int oalgpu_reverb_set_params(oalgpu_reverb *r, const oalgpu_reverb_params *params)
{
    if(!r || !params) return Fail(OALGPU_ERR_INVALID, "null argument");
    if(params->current_pipeline != 0 && params->current_pipeline != 1)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_set_params: current_pipeline must be 0 or 1");
    if(r->device >= 0)
    {   // a caller-supplied block: checked before anything is installed
        const int cur = params->current_pipeline;
        if(int rc = CheckOffsets(params->pipe[cur])) return rc;
        // the other pipeline runs only while a parameter change cross-fades (reverb.cpp:1840-1882)
        if(params->pipeline_state == OALGPU_REVERB_START_FADE || params->pipeline_state == OALGPU_REVERB_FADING)
            if(int rc = CheckOffsets(params->pipe[!cur])) return rc;
    }
    const bool full = r->host.install(*params);
    r->dirty[r->host.params.current_pipeline] = true;
    if(full) r->dirty[!r->host.params.current_pipeline] = true;
    r->upmixStale = false;
    return OALGPU_OK;
}

} // extern "C"
namespace oalgpu { uint32_t ReverbOutLines(const oalgpu_reverb *r) { return r ? r->L.nlines : 0u; } }
extern "C" {

int oalgpu_reverb_line_lengths(oalgpu_reverb *r, uint32_t lengths[11])
{
    if(!r || !lengths) return Fail(OALGPU_ERR_INVALID, "null argument");
    std::memcpy(lengths, r->lengths, sizeof(r->lengths));
    return static_cast<int>(r->total);
}

} // extern "C"

// validates, advances the host mirror by one block, installs what changed, and returns the launch
// layout of this block
static int PrepareBlock(oalgpu_reverb *r, const float *wet_in_dev, float *out_lines_dev, uint32_t n, RvLayout *out)
{
    if(!r || !wet_in_dev || !out_lines_dev || n == 0 || n > OALGPU_BUFFER_LINE_SIZE)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_process: bad arguments");
    if(r->device < 0)
        return Fail(OALGPU_ERR_NO_DEVICE, "oalgpu_reverb_process: parameter-only instance (created with device < 0)");
    if(r->host.params.pipeline_state == OALGPU_REVERB_DEVICE_CLEAR)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_process: no update() yet (the delay offsets are still zero)");
    if(r->upmixStale)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_process: oalgpu_reverb_set_upmix takes effect with an update(); none has followed it");
    if(int rc = UseDevice(r->device)) return rc;
    const ReverbHost::Step st = r->host.begin(n);
    if(st.oldMode == 2 || st.oldMode == 3) r->dirty[!st.current] = true;   // targets dropped / scalars cleared
    for(int p = 0; p < 2; ++p)
    {
        if(!r->dirty[p]) continue;
        LaunchReverbInstall(r->stream, r->pipe.p + p, r->host.params.pipe[p]);
        r->dirty[p] = false;
    }
    RvLayout L = r->L;
    L.wetIn = wet_in_dev; L.outLines = out_lines_dev; L.n = n;
    L.offset = static_cast<uint32_t>(st.offset);
    L.modIndex[0] = st.modIndex[0]; L.modIndex[1] = st.modIndex[1];
    L.current = st.current; L.oldMode = st.oldMode;
    r->host.finish(st, n);
    *out = L;
    return OALGPU_OK;
}

int oalgpu_reverb_process_batch_device(oalgpu_reverb *const *revs, const float *const *wet_in_dev, uint32_t count,
    float *out_lines_dev, uint32_t n, void *hip_stream, uint32_t *ticket_dev)
{
    if(!revs || !wet_in_dev || count == 0 || count > kRvBatchMax || (count > 1 && !ticket_dev))
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_process_batch: bad arguments");
    RvBatch B{};
    B.count = count;
    B.ticket = ticket_dev;
    for(uint32_t i = 0; i < count; ++i)
    {
        if(!revs[i]) return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_process_batch: null instance");
        revs[i]->stream = static_cast<hipStream_t>(hip_stream);
        if(int rc = PrepareBlock(revs[i], wet_in_dev[i], out_lines_dev, n, &B.r[i])) return rc;
    }
    LaunchReverbBatch(static_cast<hipStream_t>(hip_stream), B);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

extern "C" {

int oalgpu_reverb_process_device(oalgpu_reverb *r, const float *wet_in_dev, float *out_lines_dev, uint32_t n)
{
    RvLayout L{};
    if(int rc = PrepareBlock(r, wet_in_dev, out_lines_dev, n, &L)) return rc;
    LaunchReverbProcess(r->stream, L);
    HIP_TRY(hipGetLastError());
    return OALGPU_OK;
}

int oalgpu_reverb_set_math_mode(oalgpu_reverb *r, int math_mode)
{
    if(!r || (math_mode != OALGPU_MATH_EXACT && math_mode != OALGPU_MATH_FAST)) return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_set_math_mode: bad arguments");
    r->L.fast = math_mode == OALGPU_MATH_FAST ? 1u : 0u;
    return OALGPU_OK;
}

#ifdef OALGPU_MEASUREMENT      // (liboalgpu_measure.so, `make measure`: tools/measure/oalgpu_measure.h)
/* measurement aid: cycle-counter stamps of the launches from here on, [4 roles][8 sub-blocks][8]
 * (tools/reverb_phase_times.py; a device instance only) */
int oalgpu_reverb_debug_enable_phase_times(oalgpu_reverb *r)
{
    if(!r || r->device < 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_debug_enable_phase_times: needs a device instance");
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(r->stamps.alloc(4 * 8 * 8)); HIP_TRY(r->stamps.zero());
    r->L.stamps = r->stamps.p;
    return OALGPU_OK;
}

/* measurement aid: the stamps of the last launch */
int oalgpu_reverb_debug_phase_times(oalgpu_reverb *r, unsigned long long *out)
{
    if(!r || !out || !r->stamps.p) return Fail(OALGPU_ERR_INVALID, "phase times were not enabled");
    HIP_TRY(hipStreamSynchronize(r->stream));
    HIP_TRY(r->stamps.download(out, 4 * 8 * 8));
    return OALGPU_OK;
}
#endif // OALGPU_MEASUREMENT

int oalgpu_reverb_skip(oalgpu_reverb *r, uint32_t n)
{
    if(!r || n == 0 || n > OALGPU_BUFFER_LINE_SIZE) return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_skip: bad arguments");
    if(r->device >= 0) return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_skip: only for parameter-only instances");
    if(r->host.params.pipeline_state == OALGPU_REVERB_DEVICE_CLEAR)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_skip: no update() yet");
    const ReverbHost::Step st = r->host.begin(n);
    r->host.finish(st, n);
    return OALGPU_OK;
}

int oalgpu_reverb_process(oalgpu_reverb *r, const float *wet_in, float *out_lines, uint32_t n)
{
    if(!r || !wet_in || !out_lines || n == 0 || n > OALGPU_BUFFER_LINE_SIZE)
        return Fail(OALGPU_ERR_INVALID, "oalgpu_reverb_process: bad arguments");
    if(r->device < 0)
        return Fail(OALGPU_ERR_NO_DEVICE, "oalgpu_reverb_process: parameter-only instance (created with device < 0)");
    if(int rc = UseDevice(r->device)) return rc;
    const size_t outFloats = size_t{r->L.nlines} * OALGPU_BUFFER_LINE_SIZE;
    HIP_TRY(hipMemcpyAsync(r->hostIn.p, wet_in, size_t{4} * OALGPU_BUFFER_LINE_SIZE * sizeof(float),
        hipMemcpyHostToDevice, r->stream));
    HIP_TRY(hipMemcpyAsync(r->hostOut.p, out_lines, outFloats * sizeof(float), hipMemcpyHostToDevice, r->stream));
    if(int rc = oalgpu_reverb_process_device(r, r->hostIn.p, r->hostOut.p, n)) return rc;
    HIP_TRY(hipMemcpyAsync(out_lines, r->hostOut.p, outFloats * sizeof(float), hipMemcpyDeviceToHost, r->stream));
    HIP_TRY(hipStreamSynchronize(r->stream));
    return OALGPU_OK;
}

} // extern "C"
