/* oalgpu.h -- C ABI of the MI355X-native voice-mixing path for OpenAL Soft.
 *
 * This is the drop-in boundary: plain C, plain pointers and sizes, no C++ or torch types.
 * It replaces, for the per-voice mixing hot path only, what alc/alu.cpp's voice loop
 * (ProcessContexts, alc/alu.cpp:2201-2206) reaches through the reference's function-pointer
 * surface.  Each entry point cites the reference interface it stands in for (paths relative
 * to kcat/openal-soft @ 2026-08-21).  INTEGRATION.md shows the reference-side binding.
 *
 * Two granularities:
 *   1. per-call mirrors of the reference kernels (Resample_*, Mix_*, MixHrtf_*, MixHrtfBlend_*,
 *      MixDirectHrtf_*, BiquadInterpFilter::dualProcess, HrtfStore::getCoeffs) operating on
 *      host buffers -- one launch each; these exist for parity tests and small callers;
 *   2. the batched path: a device context holding source buffers, tables, the HRIR set and all
 *      per-voice mixing state in HBM, and oalgpu_mix_update() = "for every Playing|Stopping
 *      voice: voice->mix(...)" for one update of <= 1024 samples, then the HRTF post-process.
 *
 * All functions return 0 on success or a negative oalgpu_error.  Nothing here throws.  A
 * context is driven by one thread at a time (the reference's mixer thread); different contexts
 * are independent.  There is NO CPU fallback: without a usable HIP device every call fails
 * with OALGPU_ERR_NO_DEVICE.
 */
#ifndef OALGPU_H
#define OALGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built -fvisibility=hidden: these declarations are its whole export list */
#endif

#define OALGPU_BUFFER_LINE_SIZE 1024      /* BufferLineSize, core/bufferline.h:11 */
#define OALGPU_MAX_RESAMPLER_PADDING 48   /* core/resampler_limits.h:8 */
#define OALGPU_MAX_RESAMPLER_EDGE 24      /* core/resampler_limits.h:10 */
#define OALGPU_HRTF_HISTORY_LENGTH 64     /* core/mixer/hrtfdefs.h:16 */
#define OALGPU_HRIR_LENGTH 128            /* core/mixer/hrtfdefs.h:20 */
#define OALGPU_MAX_SENDS 6                /* MaxSendCount, core/voice.h:31 */
#define OALGPU_MAX_OUTPUT_CHANNELS 32     /* core/devformat.h:81 */
#define OALGPU_MAX_AMBI_CHANNELS 25       /* core/ambidefs.h:19 */
#define OALGPU_MIXER_FRAC_BITS 16         /* core/mixer/defs.h:23 */

typedef enum oalgpu_error {
    OALGPU_OK = 0,
    OALGPU_ERR_NO_DEVICE = -1,     /* no HIP device / HIP runtime failure at init */
    OALGPU_ERR_INVALID = -2,       /* bad argument */
    OALGPU_ERR_HIP = -3,           /* a HIP call failed; see oalgpu_last_error() */
    OALGPU_ERR_NO_HRTF = -4,       /* HRTF requested but no data set loaded */
    OALGPU_ERR_CAPACITY = -5       /* more voices/buffers than the context was created for */
} oalgpu_error;

/* enum class Resampler, core/mixer/defs.h:31-45 (same numeric values) */
typedef enum oalgpu_resampler {
    OALGPU_RESAMPLER_POINT, OALGPU_RESAMPLER_LINEAR, OALGPU_RESAMPLER_SPLINE,
    OALGPU_RESAMPLER_GAUSSIAN, OALGPU_RESAMPLER_FAST_BSINC12, OALGPU_RESAMPLER_BSINC12,
    OALGPU_RESAMPLER_FAST_BSINC24, OALGPU_RESAMPLER_BSINC24, OALGPU_RESAMPLER_FAST_BSINC48,
    OALGPU_RESAMPLER_BSINC48
} oalgpu_resampler;

/* FmtType, core/storage_formats.h:9-19 (PCM types; ADPCM is a "next" row) */
typedef enum oalgpu_fmt_type {
    OALGPU_FMT_UBYTE, OALGPU_FMT_SHORT, OALGPU_FMT_INT, OALGPU_FMT_FLOAT, OALGPU_FMT_DOUBLE,
    OALGPU_FMT_MULAW, OALGPU_FMT_ALAW
} oalgpu_fmt_type;

/* Voice::State, core/voice.h:178-183 */
typedef enum oalgpu_play_state {
    OALGPU_VOICE_STOPPED, OALGPU_VOICE_PLAYING, OALGPU_VOICE_STOPPING, OALGPU_VOICE_PENDING
} oalgpu_play_state;

/* BiquadType, core/filters/biquad.h:24-39 */
typedef enum oalgpu_biquad_type {
    OALGPU_BIQUAD_HIGHSHELF, OALGPU_BIQUAD_LOWSHELF, OALGPU_BIQUAD_PEAKING, OALGPU_BIQUAD_LOWPASS,
    OALGPU_BIQUAD_HIGHPASS, OALGPU_BIQUAD_BANDPASS
} oalgpu_biquad_type;

/* Arithmetic mode of a context / per-call kernel.
 * EXACT: every kernel reproduces the operation order and rounding (separate mul/add, FTZ) of
 *        the reference's x86 SSE build, so single-voice results are bit-identical to it.
 * FAST:  fused multiply-add in the FIR inner loops (resampler taps, HRTF taps); results agree
 *        with the reference within the fp32 tolerance stated in DESIGN.md. */
typedef enum oalgpu_math_mode { OALGPU_MATH_EXACT = 0, OALGPU_MATH_FAST = 1 } oalgpu_math_mode;

const char *oalgpu_version(void);
/* Human-readable text of the last failure on this thread ("" if none). */
const char *oalgpu_last_error(void);
/* Number of usable HIP devices (0 when there is none; never negative). */
int oalgpu_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Tables and parameter preparation (host side, no GPU needed)
 * ---------------------------------------------------------------------------------------- */
/* BSincTable, core/bsinc_tables.h:11-20, generated as core/bsinc_tables.cpp:147-371 does. */
typedef struct oalgpu_bsinc_table {
    float scaleBase, scaleRange;
    uint32_t m[16];
    uint32_t filterOffset[16];
    const float *tab;
    size_t tablen;
} oalgpu_bsinc_table;
int oalgpu_bsinc_table_get(int which /* 12, 24, 48 */, oalgpu_bsinc_table *out);
/* gSplineFilter / gGaussianFilter, core/cubic_tables.cpp:39-106; out = float[32][8]. */
int oalgpu_cubic_table_get(int which /* 0 spline, 1 gaussian */, float *out);

/* InterpState after PrepareResampler(resampler, increment, &state), alc/alu.cpp:253-281
 * (BsincPrepare :140-164, kernel selection :167-238). */
typedef struct oalgpu_interp_state {
    int32_t kind;           /* 0 point, 1 linear, 2 cubic, 3 fast bsinc, 4 bsinc */
    int32_t table;          /* cubic: 0 spline / 1 gaussian; bsinc: 12/24/48 */
    float sf;               /* BsincState::sf */
    uint32_t m, l;          /* BsincState::m, ::l */
    uint32_t filter_offset; /* BsincState::filter as an offset (floats) into the table */
} oalgpu_interp_state;
int oalgpu_prepare_resampler(int resampler, uint32_t increment, oalgpu_interp_state *out);

/* BiquadInterpFilter, core/filters/biquad.h:136-217 (coefficients, targets, counter, z). */
typedef struct oalgpu_biquad {
    float z1, z2;
    float b0, b1, b2, a1, a2;
    float tb0, tb1, tb2, ta1, ta2;
    int32_t counter;
} oalgpu_biquad;
void oalgpu_biquad_reset(oalgpu_biquad *f);
/* BiquadInterpFilter::setParamsFromSlope, biquad.h:172-177 -> biquad.cpp:48-149 (host libm). */
void oalgpu_biquad_set_params_from_slope(oalgpu_biquad *f, int type, float f0norm, float gain,
    float slope);

/* ------------------------------------------------------------------------------------------
 * Per-call mirrors of the reference kernels.  Host pointers in, host pointers out; each call
 * uploads, runs ONE hand-written HIP kernel on device `device`, downloads.  `mode` is an
 * oalgpu_math_mode.
 * ---------------------------------------------------------------------------------------- */
/* ResamplerFunc, core/mixer/defs.h:71-72 (Resample_{Point,Linear,Cubic,FastBSinc,BSinc}_*).
 * src = the reference's `src` span (mResampleData: starts MaxResamplerEdge samples before the
 * first source sample). */
int oalgpu_resample(int device, int mode, int resampler, uint32_t increment, const float *src,
    size_t srclen, uint32_t frac, float *dst, size_t n);
/* MixerOutFunc, core/mixer.h:22-25 (Mix_*, N lines): out = nlines x 1024 floats. */
int oalgpu_mix(int device, const float *in, size_t n, float *out, size_t nlines,
    float *current_gains, const float *target_gains, size_t counter, size_t outpos);
/* HrtfMixerFunc, core/voice.cpp:73-75 (MixHrtf_*): in = n+64 floats, accum = (1024+128) f32x2. */
int oalgpu_mix_hrtf(int device, int mode, const float *in, float *accum, uint32_t irsize,
    const float *coeffs, const uint32_t delay[2], float gain, float gainstep, size_t n);
/* HrtfMixerBlendFunc, core/voice.cpp:76-78 (MixHrtfBlend_*). */
int oalgpu_mix_hrtf_blend(int device, int mode, const float *in, float *accum, uint32_t irsize,
    const float *oldcoeffs, const uint32_t olddelay[2], float oldgain, const float *newcoeffs,
    const uint32_t newdelay[2], float newgainstep, size_t n);
/* BandSplitter state, core/filters/splitter.h:10-14 */
typedef struct oalgpu_splitter { float coeff, lp_z1, lp_z2, ap_z1; } oalgpu_splitter;
void oalgpu_splitter_init(oalgpu_splitter *s, float f0norm); /* splitter.cpp:14-26 */
/* HrtfDirectMixerFunc, alc/alu.cpp:117-120 (MixDirectHrtf_*). */
int oalgpu_mix_direct_hrtf(int device, int mode, float *left, float *right, const float *in,
    size_t nch, float *accum, oalgpu_splitter *splitters, const float *hfscales,
    const float *chan_coeffs, size_t irsize, size_t n);
/* DualBiquadInterp::process, core/filters/biquad.h:211-217 -> biquad.cpp:284-343. */
int oalgpu_biquad_dual_process(int device, oalgpu_biquad *f0, oalgpu_biquad *f1, const float *src,
    float *dst, size_t n);

/* ------------------------------------------------------------------------------------------
 * Batched path: device context
 * ---------------------------------------------------------------------------------------- */
typedef struct oalgpu_context oalgpu_context;

/* The part of DeviceBase (core/device.h:225-416) the voice path reads. */
typedef struct oalgpu_context_desc {
    int32_t  device;              /* HIP device ordinal */
    int32_t  math_mode;           /* oalgpu_math_mode */
    uint32_t sample_rate;         /* DeviceBase::mSampleRate */
    uint32_t num_dry_channels;    /* Dry.Buffer lines (3 stereo, 5 for 7.1, 4 HRTF) */
    uint32_t num_real_channels;   /* RealOut lines after Dry in MixBuffer (HRTF device: 2) */
    uint32_t num_aux_sends;       /* DeviceBase::NumAuxSends (<= 6) */
    uint32_t num_slots;           /* effect slots = wet buses */
    uint32_t wet_channels;        /* lines per wet bus (4 = 1st order, 9 = 2nd order) */
    int32_t  hrtf;                /* RenderMode::Hrtf: voices use the dual-ear FIR */
    uint32_t max_voices;
    uint32_t max_buffers;
    uint32_t voices_per_group;    /* 0 = choose automatically (tuning knob) */
    uint32_t flags;               /* OALGPU_CTX_* below; 0 = the product configuration */
} oalgpu_context_desc;
/* oalgpu_context_desc::flags -- kernel variants are chosen here, never from the environment */
#define OALGPU_CTX_FIR_VALU 1u    /* FAST HRTF voices (IrSize <= 64): the dual-ear FIR as packed fp32 VALU FMAs
                                   * instead of the matrix pipe in split half precision */
#define OALGPU_CTX_PROFILE  2u    /* the voice kernel's measurement variant: per-phase cycle stamps and stage ablation,
                                   * read and set through the measurement build only (tools/measure/oalgpu_measure.h) */
#define OALGPU_CTX_STREAM_ROWS 8u  /* FAST contexts leave a 4 KB stream row per mixed signal in HBM and mix the rows in the tail of the
                                   * two-voices-per-wavefront kernel (csrc/voice_wave.hip; the only form of contexts with near-field control and
                                   * sends) instead of their default form -- the lines in the wavefronts' registers (dry-line contexts without
                                   * sends and <= 6 lines), the rows in LDS (dry-line contexts with sends or 7 .. 32 lines, csrc/voice_rows.hip),
                                   * the voice-per-wavefront kernel with a row mixer behind it (HRTF contexts with sends, csrc/voice_wave16.hip):
                                   * for A/B runs and tests of the row path */
#define OALGPU_CTX_APPLY_IN_VOICE_KERNEL 16u /* pipelined HRTF contexts: oalgpu_mix_update is submitted with the NEXT library call on the
                                   * context, and when that call is oalgpu_param_block_apply the update's own voice kernel installs the
                                   * block (every wavefront the records of the voices it mixed) instead of a parameter kernel between
                                   * two voice kernels.  Measured slower than the parameter kernel on three of five boxes (DESIGN.md
                                   * 3.10): an opt-in variant, for A/B runs */
#define OALGPU_CTX_FUSED_REDUCE 32u /* pipelined HRTF contexts without effect slots and without a collective: the bus reduction and the
                                   * post-process as ONE launch.  One launch less for the host, and measured 1.2-1.9 us per update
                                   * slower than the two launches: an opt-in variant, for A/B runs */
#define OALGPU_CTX_RESIDENT 64u   /* pipelined FAST HRTF contexts without sends, effect slots or a collective (IrSize <= 64; a scene of more than 8 voices
                                   * per compute unit, or OALGPU_CTX_WAVE_PAIRS): ONE launch of the voice kernel stays on the machine over many updates.  Every oalgpu_mix_update (with its post-process) still
                                   * submits one update and produces its own output: the call writes the update's doorbell slot -- the length and
                                   * the parameter block of the oalgpu_param_block_apply in front of it, which the kernel's wavefronts install for
                                   * their own voices -- and launches the update's reduction and post-process, which wait for device counters
                                   * instead of for a kernel's end.  A workgroup starts update u + 1 when IT is through with u.  The kernel stays
                                   * through oalgpu_param_block_apply, oalgpu_mix_update, oalgpu_read_output_async and oalgpu_output_wait; any
                                   * other entry point of the library (on any context of the device) first tells it to leave -- it finishes the
                                   * updates that have been rung and ends, and the next oalgpu_mix_update launches a new one -- so results are
                                   * those of a launch per update, bit for bit (tests/test_gpu_pipeline.py).  A host that stops calling for long
                                   * should call oalgpu_sync: the kernel holds the device's compute units while it waits, and gives up (the
                                   * context then reports an error and launches per update) after 2 s without a doorbell inside a wait.
                                   * Contexts the mode does not cover ignore the flag. */
#define OALGPU_CTX_SLICE_LINES 128u /* FAST dry-line contexts whose lines do not fit the wavefronts' registers (sends, or 7 .. 24 mix lines; no
                                   * near-field control): instead of leaving a 4 KB stream row per mixed signal in HBM and mixing the rows in
                                   * the voice kernel's tail, cut the update into four 256-frame slices, one per wavefront of a workgroup, each
                                   * of which walks all of the workgroup's voices and keeps its 24 lines x 4 frames per lane in registers: no
                                   * row ever leaves the CU (csrc/voice_slice.hip).  Measured on BASELINE configs[3]: the voice kernel's
                                   * HBM traffic falls to a third and its time nearly doubles -- the per-voice work is done four times and the
                                   * kernel is instruction-bound: an opt-in variant, for A/B runs.  Other contexts ignore it. */
#define OALGPU_CTX_WAVE_PAIRS 256u /* FAST HRTF contexts (IrSize <= 64) mix one voice per wavefront at four wavefronts per SIMD -- 16, 8 or 4 wavefronts
                                   * per workgroup by the scene's size; the resampler's outputs in registers, one ear's FIR inputs at a time
                                   * (csrc/voice_wave16.hip); with auxiliary sends the send's signal leaves as one stream row per
                                   * voice and send, and a small kernel behind the voice kernel mixes the rows onto the slots' wet lines.  This flag
                                   * selects the form of rounds 1-5 instead: two voices per wavefront, two wavefronts per SIMD, one first-order
                                   * slot's wet lines in registers (csrc/voice_wave.hip) -- for A/B runs.  OALGPU_CTX_RESIDENT has a resident launch of
                                   * either form: this kernel's with the flag, the 16-wavefront form of the voice-per-wavefront kernel (scenes that
                                   * fill the machine) without.  Other contexts ignore the flag. */
#define OALGPU_CTX_ROW_SLICES 512u /* (the DEFAULT form of these contexts since round 6; the flag is accepted and names it.)  FAST dry-line contexts
                                   * with sends (or 7 .. 32 mix lines; no near-field control): a voice's signals never leave the compute unit AND
                                   * its fixed work is done once -- a wavefront per voice resamples into a 4 KB slot of LDS, the round's filtered
                                   * signals are jobs dealt to all eight wavefronts of the workgroup, and each wavefront adds its own 128-frame
                                   * slice of every row to 32 line accumulators it keeps in registers for the whole launch
                                   * (csrc/voice_rows.hip).  OALGPU_CTX_STREAM_ROWS selects the rows-in-HBM form instead. */
#define OALGPU_CTX_SERIAL   4u    /* oalgpu_mix_update on one stream (no overlap of an update's reduction and
                                   * post-process with the next update's voices): a measurement aid */

int  oalgpu_context_create(const oalgpu_context_desc *desc, oalgpu_context **out);
void oalgpu_context_destroy(oalgpu_context *ctx);

/* LoadHrtf(std::istream&), core/hrtf_loader.hpp:10 (format v3 "MinPHR03",
 * core/hrtf_loader.cpp:583-721); the store is uploaded to HBM.  Must precede any HRTF voice. */
int oalgpu_hrtf_load_mhr(oalgpu_context *ctx, const void *data, size_t size);
/* The same from an HrtfStore already in memory (core/hrtf.h:22-59): what DeviceBase::mHrtf holds once GetLoadedHrtf
 * (core/hrtf.cpp:471-620) has loaded -- and resampled -- a data set; the arrays are those oalgpu_hrtf_raw hands out
 * (mFields: distance / evCount, mElev: azCount / irOffset, mCoeffs: num_irs x 128 x 2, mDelays: num_irs x 2). */
int oalgpu_hrtf_load_store(oalgpu_context *ctx, uint32_t sample_rate, uint32_t ir_size, const float *field_distance,
    const uint8_t *field_evcount, uint32_t num_fields, const uint16_t *elev_azcount, const uint16_t *elev_iroffset,
    uint32_t num_elevs, const float *coeffs, const uint8_t *delays, uint32_t num_irs);
typedef struct oalgpu_hrtf_info {
    uint32_t sample_rate, ir_size, num_fields, num_elevs, num_irs;
} oalgpu_hrtf_info;
int oalgpu_hrtf_info_get(oalgpu_context *ctx, oalgpu_hrtf_info *out);
/* Host copy of the parsed store, HrtfStore layout (core/hrtf.h:22-44). */
int oalgpu_hrtf_raw(oalgpu_context *ctx, float *field_distance, uint8_t *field_evcount,
    uint16_t *elev_azcount, uint16_t *elev_iroffset, float *coeffs, uint8_t *delays);
/* HrtfStore::getCoeffs, core/hrtf.cpp:192-260, evaluated ON THE GPU for `count` directions.
 * dirs = count x {elevation, azimuth, distance, spread}; coeffs = count x 128 x 2;
 * delays = count x 2. */
int oalgpu_hrtf_get_coeffs(oalgpu_context *ctx, const float *dirs, size_t count, float *coeffs,
    uint32_t *delays);
/* DirectHrtfState (core/hrtf.h:85-109) used by the post-process: per dry channel decoder IR,
 * HF scale and the shared crossover (the decoder design itself, alc/panning.cpp:847-1138, is
 * one-time init and stays with the caller). */
int oalgpu_set_direct_hrtf(oalgpu_context *ctx, const float *chan_coeffs /* ndry x 128 x 2 */,
    const float *hfscales, float xover_norm, uint32_t irsize);
/* DirectHrtfState::build (core/hrtf.cpp:266-366), the decoder design itself, on the context's data set: from the
 * host's virtual-speaker layout -- points = num_points x {elevation, azimuth} (radians), matrix = num_points x 16
 * (the AmbiMatrix rows), order_hf_gain[5] (alc/panning.cpp:861-1038 holds the reference's tables; InitHrtfPanning
 * passes ir_size = the store's IrSize (0 here) and per_hrir_min for orders >= 3).  One-time host work in double
 * precision; ends in oalgpu_set_direct_hrtf. */
int oalgpu_set_direct_hrtf_from_store(oalgpu_context *ctx, const float *points, const float *matrix, uint32_t num_points,
    const float *order_hf_gain, float xover_freq, uint32_t ir_size, int per_hrir_min);
/* Host-only (no device needed): a .mhr brought to device_rate as GetLoadedHrtf does (core/hrtf.cpp:539-606; 0 = the
 * set's own rate) -- info, and coeffs (num_irs x 128 x 2) / delays (num_irs x 2) when not NULL -- and the decoder
 * build on it.  oalgpu_hrtf_load_mhr does the same resampling when the context's rate differs from the set's. */
int oalgpu_hrtf_parse_host(const void *mhr, size_t size, uint32_t device_rate, oalgpu_hrtf_info *info, float *coeffs, uint8_t *delays);
/* Host-only: the index half of HrtfStore::getCoeffs (core/hrtf.cpp:192-245) as oalgpu_voice_set_params and
 * oalgpu_voice_move_async evaluate it when they build a record -- for `count` directions (elevation, azimuth, distance,
 * spread each) the four HRIR indices, their weights, the pass-through tap and the two blended delays
 * (idx / w: count x 4, pass: count, delays: count x 2). */
int oalgpu_hrtf_blend_host(const void *mhr, size_t size, uint32_t device_rate, const float *dirs, size_t count,
    uint32_t *idx, float *w, float *pass, uint32_t *delays);
int oalgpu_hrtf_build_direct_host(const void *mhr, size_t size, uint32_t device_rate, uint32_t ir_size, int per_hrir_min,
    const float *points, const float *matrix, uint32_t num_points, uint32_t num_chans, float xover_freq, const float *order_hf_gain,
    float *out_coeffs, float *out_hfscales, float *out_xover_norm, uint32_t *out_irsize);

/* al::Buffer storage / VoiceBufferItem (core/voice.h:84-98): sample data is copied to HBM once.
 * frame_step = interleaved samples per frame; mono voices read channel 0.  Returns the handle
 * (>= 0) or a negative error. */
int oalgpu_buffer_register(oalgpu_context *ctx, const void *data, int fmt_type,
    uint32_t frame_step, uint32_t sample_len, uint32_t loop_start, uint32_t loop_end);
/* alDeleteBuffers / a buffer's storage replaced (core/buffer_storage.h:47-77: the storage is freed and reused; VoiceBufferItems
 * are reused too, core/voice.h:84-98): the host gives the handle up.  The HBM copy is freed, and the handle handed out again by a
 * later registration, once nothing refers to it: no voice slot that was initialised on it -- a slot lets go when it is
 * initialised again or set to OALGPU_VOICE_STOPPED --, no queue link of a live buffer (oalgpu_buffer_queue_link), no channel
 * view (oalgpu_buffer_channel_view).  Until then the voices that play it keep playing it; the handle can no longer be named
 * in oalgpu_voice_init / oalgpu_buffer_queue_link / oalgpu_buffer_channel_view.  max_buffers bounds the LIVE handles. */
int oalgpu_buffer_release(oalgpu_context *ctx, int buffer);
/* (tests) whether the handle is live, whether its release is waiting for references to go, and how many hold it */
int oalgpu_buffer_info(oalgpu_context *ctx, int buffer, int32_t *live, int32_t *release_pending, uint32_t *references);

/* One channel of an interleaved multi-channel buffer as a buffer of its own (same frames and
 * loop points, no copy): multi-channel sources -- B-Format in particular -- are mixed as one
 * voice per channel, which is what Voice::mix's per-ChannelData loop does (DoMix,
 * core/voice.cpp:934-984).  Returns the new handle. */
int oalgpu_buffer_channel_view(oalgpu_context *ctx, int buffer, uint32_t channel);

/* Voice::prepare + the source attach of InitVoice (al/source.cpp:639-670) for a static mono
 * voice: mixing state cleared, position set, state Playing, not fading. */
typedef struct oalgpu_voice_desc {
    int32_t  buffer;
    int32_t  looping;             /* mLoopBuffer != nullptr */
    int32_t  position;            /* mPosition */
    uint32_t position_frac;       /* mPositionFrac */
    uint32_t frequency;           /* mFrequency */
} oalgpu_voice_desc;
int oalgpu_voice_init(oalgpu_context *ctx, uint32_t voice, const oalgpu_voice_desc *desc);
/* Near-field control (DoNfcMix, core/voice.cpp:904-932; NfcFilter, core/filters/nfc.cpp).
 * Context side (alc/panning.cpp:285-299): the control filter NfcFilter::init(w1), w1 =
 * speed_of_sound / (control_distance * sample_rate), and DeviceBase::NumChannelsPerOrder[0..4]
 * (a 0 ends the list; [0] is the W line); call before the voices are initialised.  FAST-mode
 * dry-line contexts only.  Voice side (alc/alu.cpp:919-941, 1328-1341): NFCtrlFilter.adjust(w0)
 * and VoiceFlag::HasNfc; the W line is then mixed from the voice's samples, every order's lines
 * from that order's NFC-filtered copy. */
int oalgpu_context_set_nfc(oalgpu_context *ctx, float w1, const uint32_t channels_per_order[5]);
int oalgpu_voice_set_nfc(oalgpu_context *ctx, uint32_t voice, float w0);

/* VoiceFlag::IsAmbisonic for this (channel) voice: ChannelData::mAmbiSplitter.init(xover_norm),
 * mAmbiHFScale, mAmbiLFScale as Voice::prepare sets them up for a B-Format source on a
 * higher-order device (core/voice.cpp:1353-1380); every mix then runs
 * BandSplitter::processScale over the resampled samples ahead of DoFilters (:1082-1091).
 * Call after oalgpu_voice_init (which clears it). */
int oalgpu_voice_set_ambi_scale(oalgpu_context *ctx, uint32_t voice, float xover_norm, float hf_scale,
    float lf_scale);

typedef struct oalgpu_filter_params {
    int32_t active;               /* TargetData::FilterActive */
    float gain_hf, hf_norm;       /* HighShelf gain, HFReference / sample_rate */
    float gain_lf, lf_norm;       /* LowShelf gain, LFReference / sample_rate */
} oalgpu_filter_params;

/* What CalcVoiceParams (alc/alu.cpp:1512-1710,2012-2031) leaves in the Voice for mix() to use. */
typedef struct oalgpu_voice_params {
    uint32_t step;                                   /* mStep */
    int32_t  resampler;                              /* props.mResampler */
    oalgpu_filter_params direct_filter;              /* alc/alu.cpp:1619-1637 */
    float    dry_gains[OALGPU_MAX_OUTPUT_CHANNELS];  /* mDryParams.Gains.Target */
    float    hrtf_ev, hrtf_az, hrtf_dist, hrtf_spread; /* getCoeffs arguments, alu.cpp:1214.  A distance is never negative
                                                      * in the reference (a vector norm, alu.cpp:1761, scaled: :1214); NEGATIVE values of hrtf_dist
                                                      * are reserved: OALGPU_HRTF_KEEP_TARGET (-1) = the voice's HRTF target,
                                                      * delays and gain stay as they are (see oalgpu_voice_set_hrtf_targets),
                                                      * any other negative value is rejected with OALGPU_ERR_INVALID */
    float    hrtf_gain;                              /* Hrtf.Target.Gain */
    int32_t  send_slot[OALGPU_MAX_SENDS];            /* -1: mSend[i].Buffer empty */
    oalgpu_filter_params send_filter[OALGPU_MAX_SENDS];
    float    send_gains[OALGPU_MAX_SENDS][OALGPU_MAX_AMBI_CHANNELS]; /* mWetParams[i].Gains.Target */
} oalgpu_voice_params;
#define OALGPU_HRTF_KEEP_TARGET (-1.0f)
/* For a host that keeps the reference's own parameter stage (CalcVoiceParams as it is): Hrtf.Target of `count` voices as
 * CalcHrtfPanning left it in the Voice (HrtfFilter, core/mixer/hrtfdefs.h:36-40; alc/alu.cpp:1214-1216, :1256-1258,
 * :1296-1298) -- coeffs = count x 128 x 2 (the blended HrirArray), delays = count x 2, gains = count -- installed as the
 * voices' target filter (marked as replaced: the next mix cross-fades from Hrtf.Old, voice.cpp:846-873).  The voices'
 * other parameters go through oalgpu_voice_set_params with hrtf_dist = OALGPU_HRTF_KEEP_TARGET. */
int oalgpu_voice_set_hrtf_targets(oalgpu_context *ctx, const uint32_t *voices, const float *coeffs, const uint32_t *delays,
    const float *gains, size_t count);

/* ---- panning on the GPU: CalcDirectionCoeffs + ComputePanGains -----------------------------
 * What CalcPanningAndFilters does for a point source (alc/alu.cpp; core/mixer.h:68-73, core/ambidefs.h:219-271,
 * core/mixer.cpp:16-102): coeffs = CalcDirectionCoeffs(dir, spread), then ComputePanGains(&Device->Dry, coeffs,
 * dry_gain, mDryParams.Gains.Target) and, per send with a slot, ComputePanGains(&Slot->Wet, coeffs, send_gain[i],
 * mWetParams[i].Gains.Target) -- gains[line] = AmbiMap[line].Scale * coeffs[AmbiMap[line].Index] * gain.  The
 * device's and the slots' AmbiMaps (core/device.h MixParams) are handed over once; until then they are the
 * identity (line i = ACN i, scale 1).  oalgpu_voice_set_pan REPLACES the dry_gains / send_gains of the voices'
 * last oalgpu_voice_params (an HRTF context has no dry gains: only the sends'); `dir` is the normalized
 * OpenAL-space direction.  Without spread the gains are bit-identical to the reference's. */
typedef struct oalgpu_voice_pan {
    float dir[3];
    float spread;                                    /* 0 .. tau */
    float dry_gain;                                  /* DryGain.Base * downmix gain */
    float send_gain[OALGPU_MAX_SENDS];               /* WetGain[i].Base * downmix gain */
} oalgpu_voice_pan;
int oalgpu_context_set_ambi_map(oalgpu_context *ctx, const uint8_t *index, const float *scale);        /* num_dry entries */
int oalgpu_slot_set_ambi_map(oalgpu_context *ctx, uint32_t slot, const uint8_t *index, const float *scale); /* wet_channels */
int oalgpu_voice_set_pan(oalgpu_context *ctx, const uint32_t *voices, const oalgpu_voice_pan *pans, size_t count);

/* ---- streaming sources: a queue of buffers (VoiceBufferItem::mNext, core/voice.h:85) ---------------------
 * oalgpu_buffer_queue_link(buffer, next) links `next` behind `buffer` (alSourceQueueBuffers; next < 0 ends
 * the queue there).  oalgpu_voice_init_queue starts a voice that is NOT VoiceFlag::IsStatic on the queue's
 * first buffer: LoadBufferQueue (core/voice.cpp:563-594) crawls the queue -- a looping voice returns to
 * first_buffer when it ends, a non-looping one holds the last sample and stops -- and Voice::mix leaves
 * finished buffers behind (voice.cpp:1182-1194).  oalgpu_voice_queue_state reports the current buffer
 * (-1 once the queue ended) and how many buffers the voice has played through (the counts of
 * AsyncBufferCompleteEvent, voice.cpp:1207-1218). */
int oalgpu_buffer_queue_link(oalgpu_context *ctx, int buffer, int next_buffer);
int oalgpu_voice_init_queue(oalgpu_context *ctx, uint32_t voice, int first_buffer, int looping, int32_t position,
    uint32_t position_frac);
int oalgpu_voice_queue_state(oalgpu_context *ctx, uint32_t voice, int32_t *current_buffer, uint32_t *buffers_done);
/* alSourceUnqueueBuffers: the first `count` buffers of the voice's queue -- processed ones: checked against the buffers_done the
 * host has read back (oalgpu_voice_queue_state, oalgpu_voices_readback) -- leave the queue; the voice's hold moves to the buffer
 * behind them, so that a released buffer among them is freed while the source plays on. */
int oalgpu_voice_queue_unqueue(oalgpu_context *ctx, uint32_t voice, uint32_t count);

/* ---- callback sources (AL_SOFT_callback_buffer) ----------------------------------------------------------------
 * BufferStorage::mCallback / VoiceBufferItem::mCallback (core/buffer_storage.h:48, core/voice.h:85): the source's
 * samples come from a function the mixer calls in the middle of Voice::mix, for exactly the blocks the voice's
 * position needs (voice.cpp:726-752), into a storage of MixerLineSize*MaxPitch + MaxResamplerEdge frames
 * (al/buffer.cpp:474) whose consumed blocks Voice::mix drops afterwards (:1155-1180).  Calling user code is host
 * work by definition: oalgpu_mix_update calls `fn` on the calling thread BEFORE the update's voice kernel is
 * launched, with the byte counts the reference would ask for (the library mirrors the voice's integer state --
 * mPositionFrac, mStep, mNumCallbackBlocks, mCallbackBlockOffset, VoiceFlag::CallbackStopped -- on the host), uploads
 * the storage in stream order and hands it to the voice kernels as a static, non-looping buffer of
 * mNumCallbackBlocks samples read from mCallbackBlockOffset: LoadBufferCallback (:546-561) and the non-looping
 * LoadBufferStatic (:504-519) are the same function of (length, position), and both kinds of voice end when the
 * position reaches the length.  Mono PCM formats (one sample per block).  A callback that returns fewer bytes than
 * asked stops being called (CallbackStopped); the voice plays what it has, holds the last sample, and ends.
 * fn: `int32_t fn(void *userptr, void *data, int32_t num_bytes)` -> bytes written. */
typedef int32_t (*oalgpu_callback_fn)(void *userptr, void *data, int32_t num_bytes);
int oalgpu_voice_init_callback(oalgpu_context *ctx, uint32_t voice, int fmt_type, uint32_t position_frac,
    oalgpu_callback_fn fn, void *userptr);
typedef struct oalgpu_callback_state {
    int32_t position;                   /* Voice::mPosition (the device-side position is relative to the storage) */
    uint32_t position_frac, num_blocks, block_offset;
    int32_t stopped, play_state, has_buffer;
} oalgpu_callback_state;
int oalgpu_voice_callback_state(oalgpu_context *ctx, uint32_t voice, oalgpu_callback_state *out);

/* IMA4 / MS ADPCM data (FmtIMA4 / FmtMSADPCM, core/buffer_storage.h:35-43; LoadSamples, core/voice.cpp:288-484):
 * `data` = ceil(sample_len / samples_per_block) blocks of ((samples_per_block-1)/2 + 4) * channels (IMA4)
 * or ((samples_per_block-2)/2 + 7) * channels (MS) bytes.  Decoded once, on the GPU, into interleaved
 * 16-bit PCM: the handle then behaves like an OALGPU_FMT_SHORT buffer with frame_step = channels. */
enum oalgpu_adpcm_type { OALGPU_ADPCM_IMA4 = 0, OALGPU_ADPCM_MS = 1 };
int oalgpu_buffer_register_adpcm(oalgpu_context *ctx, const void *data, int adpcm_type, uint32_t channels,
    uint32_t samples_per_block, uint32_t sample_len, uint32_t loop_start, uint32_t loop_end);

/* Delayed start (Voice::mStartTime, core/voice.h:209; Voice::mix, core/voice.cpp:1023-1046): the voice --
 * initialised, not mixed yet -- starts `samples` output samples after the beginning of the next update:
 * that update mixes samples_to_do - outPos samples of it at output position outPos; updates that end
 * earlier leave it untouched (the remaining delay shrinks by samples_to_do).  A voice told to stop before
 * it started becomes Stopped.  samples < sample_rate (the reference does not schedule starts a second
 * or more ahead either). */
int oalgpu_voice_set_start_delay(oalgpu_context *ctx, uint32_t voice, uint32_t samples);

/* Applies `count` parameter blocks (voices[i] <- params[i]); the HRIR blend of getCoeffs and
 * the BiquadInterpFilter::setParams state machine run on the GPU. */
int oalgpu_voice_set_params(oalgpu_context *ctx, const uint32_t *voices,
    const oalgpu_voice_params *params, size_t count);
/* The same, split so a caller can stage parameter blocks in HBM ahead of time (one block per
 * update of a scripted scene) and apply them asynchronously on the context's stream. */
typedef struct oalgpu_param_block oalgpu_param_block;
int  oalgpu_param_block_create(oalgpu_context *ctx, const uint32_t *voices,
    const oalgpu_voice_params *params, size_t count, oalgpu_param_block **out);
int  oalgpu_param_block_apply(oalgpu_context *ctx, oalgpu_param_block *block);
/* `count` consecutive updates submitted by one call: update i applies param_blocks[i] (the array or an entry may be
 * NULL), then oalgpu_mix_update(samples_to_do, post_process) -- what a host's render loop does, in one call */
int  oalgpu_mix_update_run(oalgpu_context *ctx, oalgpu_param_block *const *param_blocks, uint32_t count,
    uint32_t samples_to_do, int post_process);
void oalgpu_param_block_destroy(oalgpu_param_block *block);
/* ProcessVoiceChanges side (alc/alu.cpp:2057-2151): Playing / Stopping / Stopped. */
int oalgpu_voice_set_state(oalgpu_context *ctx, uint32_t voice, int play_state);

/* One update: zero the dry/real and wet buses, mix every Playing|Stopping voice
 * (Voice::mix, core/voice.cpp:988-1233), and -- HRTF context, post_process != 0 -- run
 * MixDirectHrtf over the dry bus (DeviceBase::Process(HrtfPostProcess), alc/alu.cpp:289-298).
 * Asynchronous on the context's stream; oalgpu_sync() or a read-back waits. */
int oalgpu_mix_update(oalgpu_context *ctx, uint32_t samples_to_do, int post_process);
int oalgpu_sync(oalgpu_context *ctx);
/* OALGPU_CTX_RESIDENT: what the mode has done so far.  timed_*: launches that have ended -- their own durations (HIP events
 * bound to the dispatch) and the updates they covered; timed_kernel_ms / timed_updates is the voice kernel's time per update. */
typedef struct oalgpu_resident_info {
    int32_t  enabled, failed, running, door_in_device_memory;
    uint32_t launches, updates, parks, timed_launches;
    uint64_t timed_updates;
    double   timed_kernel_ms;
    uint32_t max_updates_per_launch, pad;
    /* where the time BETWEEN the kernels' work went, in microseconds summed over the updates so far (the counters are read when
     * the context is idle: oalgpu_sync first): wait_door / wait_reduction: a voice workgroup (mean over the workgroups) waiting for
     * the doorbell / for the reduction that still reads its partial set; wait_arrival / wait_post: the reduction waiting for the
     * voice workgroups / for the post-process of the update before; wait_reduced / wait_split: the post-process waiting for the
     * reduction / its FIR workgroups for the band splits */
    double   wait_door_us, wait_reduction_us, wait_arrival_us, wait_post_us, wait_reduced_us, wait_split_us;
    double   install_us, busy_us, top_us;   /* a voice workgroup (mean): installing the update's parameter block; mixing its voices up to the
                                            * arrival; from there to the next update known (the door's loads and the two waits) */
} oalgpu_resident_info;
int oalgpu_resident_stats(oalgpu_context *ctx, oalgpu_resident_info *out);
/* A launch pays for itself over a few dozen updates; a host that keeps ending it early (a synchronisation every 20 updates,
 * another entry point before every update) is better off with a launch per update.  After three launches in a row that were
 * parked before `updates` updates (default 32) the context launches per update for the next 192, then tries again; 0 = never. */
int oalgpu_resident_set_short_run(oalgpu_context *ctx, uint32_t updates);
/* the launches carry HIP events bound to their dispatch (timed_* above): ~15 us more of host time per launch, off by default */
int oalgpu_resident_set_timing(oalgpu_context *ctx, int enable);
/* a launch ends by itself after this many updates (default 4096); the next update starts a new one */
int oalgpu_resident_set_max_updates(oalgpu_context *ctx, uint32_t max_updates);
/* Run the context on a caller-owned HIP stream (hipStream_t), e.g. the stream an RCCL
 * collective is ordered on; NULL returns to a private stream. */
int oalgpu_set_stream(oalgpu_context *ctx, void *hip_stream);

/* Bus read-back (host copies).  dry: (num_dry+num_real) x 1024; wet: wet_channels x 1024;
 * hrtf_accum: (1024+128) x 2. */
int oalgpu_read_dry(oalgpu_context *ctx, float *out);
int oalgpu_read_wet(oalgpu_context *ctx, uint32_t slot, float *out);
int oalgpu_read_hrtf_accum(oalgpu_context *ctx, float *out);
/* ---- the stage behind the buses: speaker decode, dither, output PCM -----------------------------------
 * BFormatDec (core/bformatdec.cpp:27-95; DeviceBase::Process(AmbiDecPostProcess), alc/alu.cpp:282-287):
 * the post-process of a NON-HRTF context.  coeffs_hf / coeffs_lf are the constructor's `coeffs` /
 * `coeffslf`: num_out rows (one ChannelDec per output channel, index = real output line) of
 * OALGPU_MAX_AMBI_CHANNELS floats (column = dry line); coeffs_lf NULL = single band, otherwise every
 * dry line goes through a BandSplitter at xover_norm (= mXOverFreq / sample rate).  With a decoder set,
 * oalgpu_mix_update(.., post_process = 1) / oalgpu_post_process add the decoded feeds to the real output
 * lines after the effect slots ran.  num_out = 0 removes it. */
int oalgpu_set_bformat_decoder(oalgpu_context *ctx, uint32_t num_out, const float *coeffs_hf,
    const float *coeffs_lf, float xover_norm);
/* Output conversion: sample_type in DevFmtType order (core/devformat.h:56-64); dither_depth = the
 * device's DitherDepth (2^(bits-1), 0 = off), dither_seed = DitherSeed. */
enum oalgpu_output_type {
    OALGPU_OUT_I8 = 0, OALGPU_OUT_U8, OALGPU_OUT_I16, OALGPU_OUT_U16, OALGPU_OUT_I32, OALGPU_OUT_U32, OALGPU_OUT_F32
};
int oalgpu_set_output(oalgpu_context *ctx, int sample_type, float dither_depth, uint32_t dither_seed);
/* ApplyDither (alc/alu.cpp:2309-2332; in place on the output lines, like the reference) and
 * Write<T> / SampleConv<T> (alu.cpp:2335-2390) of the last update's output lines (the real output lines,
 * or the dry lines of a context without any), then ONE device-to-host copy of the interleaved PCM:
 * samples_to_do frames of frame_step samples (channels past the output lines are silent).  Bit-exact. */
int oalgpu_read_output(oalgpu_context *ctx, void *out, uint32_t samples_to_do, uint32_t frame_step);

/* Device address of the bus block [dry+real lines | wet buses | hrtf accum], its length in
 * floats, and the stream it is produced on, for zero-copy consumers: the context's main stream for
 * the serial entry points (oalgpu_mix_voices / oalgpu_post_process), its POST stream when the
 * pipelined path is active (oalgpu_mix_update on a FAST HRTF context with its own streams). */
int oalgpu_bus_device_ptr(oalgpu_context *ctx, void **ptr, size_t *nfloats, void **hip_stream);
/* Multi-GPU split of one update: mix_voices fills this rank's partial buses (no post-process);
 * after the caller has summed the bus block across ranks (one RCCL all-reduce/reduce),
 * post_process runs MixDirectHrtf on the summed buses. */
int oalgpu_mix_voices(oalgpu_context *ctx, uint32_t samples_to_do);
int oalgpu_post_process(oalgpu_context *ctx, uint32_t samples_to_do);
/* The same split on the context's own two streams (FAST HRTF contexts that were not given a
 * caller-owned stream): mix_voices_overlapped launches the voice kernel on the main stream and
 * the partial-bus reduction on the context's post stream, then returns; the caller enqueues its
 * collective on the post stream (oalgpu_post_stream), and post_process_overlapped follows it
 * there.  The next update's parameter and voice kernels run beside the collective and the
 * post-process of this one (double-buffered partial buses). */
int   oalgpu_mix_voices_overlapped(oalgpu_context *ctx, uint32_t samples_to_do);
void *oalgpu_post_stream(oalgpu_context *ctx);
int   oalgpu_post_process_overlapped(oalgpu_context *ctx, uint32_t samples_to_do, int run_post_process);
/* Whether this context's voice kernel continues the HRTF accumulator tail carried over from
 * the previous update (HrtfAccumData, core/device.h:288).  Default on.  With the buses summed
 * across ranks exactly one rank -- the one whose post-process owns the tail -- keeps it on. */
int oalgpu_set_carry_accum(oalgpu_context *ctx, int enable);

/* ---- multi-GPU inside the library: RCCL over xGMI -----------------------------------------------------
 * The reference has no multi-device path (one mixer thread per device); voices are
 * independent given their parameters, so they shard over the GPUs of a node -- one process and one
 * context per GPU -- and the only exchange is ONE sum-reduce of the bus block
 * [dry + real lines | wet buses | HrtfAccumData] to rank 0 per update.
 *   rank 0:      oalgpu_comm_unique_id(id, 128)           (ncclGetUniqueId; hand the 128 bytes to the others)
 *   every rank:  oalgpu_comm_init(ctx, id, 128, rank, world)     (ncclCommInitRank on the context's device)
 * From then on oalgpu_mix_update / oalgpu_mix_voices issue the ncclReduce themselves, on the stream that
 * produced the buses (the context's post stream in the pipelined path: the collective, the effects and the
 * post-process of update k run beside the voice kernel of update k+1), and only rank 0 -- the one rank
 * that carries the HRTF accumulator tail -- runs the effect slots and the post-process.  librccl.so is
 * resolved at run time (an instance already in the process, e.g. torch's, is reused); a host that never
 * calls these needs no RCCL.  `size` is the size of the id buffer (>= 128). */
int oalgpu_comm_unique_id(void *unique_id, size_t size);
int oalgpu_comm_init(oalgpu_context *ctx, const void *unique_id, size_t size, int rank, int world);
/* The same exchange over the library's second transport, host-staged: every rank's bus block goes through pinned
 * memory into a shared-memory ring (`name`: a POSIX shared-memory object name, "/...", the same on every rank;
 * rank 0 creates it) and rank 0's stream sums the blocks in rank order -- all in stream order, nothing synchronises
 * with the host.  For ranks RCCL cannot connect: several processes sharing ONE GPU.  Everything else of a sharded
 * update (which rank runs effects and post-process, which carries the accumulator) is the same code as with RCCL. */
int oalgpu_comm_init_host(oalgpu_context *ctx, const char *name, int rank, int world);
/* the exchange as the library sees it: this rank, the world it was given, the ranks the transport itself counts (RCCL:
 * ncclCommCount of the library's communicator) and the transport's name ("rccl", "host", "none") */
int oalgpu_comm_info(oalgpu_context *ctx, int *rank, int *world, int *transport_ranks, char *kind, size_t kind_size);
int oalgpu_comm_destroy(oalgpu_context *ctx);

/* Mixing state of one voice after the last update (the fields Voice::mix mutates). */
typedef struct oalgpu_voice_state {
    int32_t  play_state;
    int32_t  position;
    uint32_t position_frac;
    int32_t  has_buffer;
    int32_t  fading;
    float    prev_samples[OALGPU_MAX_RESAMPLER_PADDING];
    float    dry_current[OALGPU_MAX_OUTPUT_CHANNELS];
    float    hrtf_old_gain;
    uint32_t hrtf_old_delay[2];
    float    hrtf_history[OALGPU_HRTF_HISTORY_LENGTH];
    oalgpu_biquad direct_lp, direct_hp;
    float    send_current[OALGPU_MAX_SENDS][OALGPU_MAX_AMBI_CHANNELS];
    oalgpu_biquad send_lp[OALGPU_MAX_SENDS], send_hp[OALGPU_MAX_SENDS];
} oalgpu_voice_state;
int oalgpu_voice_readback(oalgpu_context *ctx, uint32_t voice, oalgpu_voice_state *out);
/* The part of it the rest of the reference looks at after an update -- GetSourceOffset's position, the play state,
 * whether the source ran out of buffer (Voice::mix then sets mCurrentBuffer = nullptr and the state Stopping,
 * core/voice.cpp:1201-1232), VoiceFlag::IsFading -- for `count` voices in one device-to-host copy. */
typedef struct oalgpu_voice_brief {
    int32_t  play_state;
    int32_t  position;
    uint32_t position_frac;
    int32_t  has_buffer;
    int32_t  fading;
    int32_t  current_buffer;      /* the handle of mCurrentBuffer (-1: none): where a streaming source's queue has got to */
    uint32_t buffers_done;        /* streaming sources: buffers played through since the voice was initialised */
} oalgpu_voice_brief;
int oalgpu_voices_readback(oalgpu_context *ctx, const uint32_t *voices, size_t count, oalgpu_voice_brief *out);

/* ---- the pipelined host boundary: an update's moved voices in, its output lines out, nothing waits ---------------------
 * What CalcPanningAndFilters (alc/alu.cpp:1512-1657) hands over for a voice whose direction moved while its filter
 * targets stayed: the HRTF direction and gain.  oalgpu_voice_move_async copies the raw 24-byte records into a ring slot --
 * device memory mapped into the host's address space where the box has a large BAR (probed once per context with a kernel
 * that reads what the host stored), pinned host memory otherwise -- and queues the kernel that installs them in front of the
 * next oalgpu_mix_update: it reads the records out of the slot and evaluates BOTH halves of HrtfStore::getCoeffs
 * (core/hrtf.cpp:192-260: the indices and blend weights, then the weighted sum of the four responses) on the device.
 * One runtime call; it returns without waiting for any of that.  HRTF contexts only. */
typedef struct oalgpu_voice_move {
    uint32_t voice;
    float hrtf_ev, hrtf_az, hrtf_dist, hrtf_spread;   /* as in oalgpu_voice_params */
    float hrtf_gain;
} oalgpu_voice_move;
int oalgpu_voice_move_async(oalgpu_context *ctx, const oalgpu_voice_move *moves, size_t count);
/* The update's output lines -- the real lines of an HRTF / decoded context (left, right, ...), else the dry lines;
 * [line][1024] floats -- two updates late without a copy in between: HRTF contexts with a post-process have the
 * post-process kernel store the lines into a pinned ring slot itself and raise the slot's sequence number (no copy launch,
 * no runtime call on the host: oalgpu_output_wait polls that word); other contexts queue a device-to-host copy behind the
 * update.  Returns a ticket; oalgpu_output_wait blocks until the lines have landed and hands them over.  The ring has four slots:
 * a ticket stays valid until four more have been handed out -- the call never fails for want of a slot, it reuses the oldest, and
 * oalgpu_output_wait on a ticket whose slot has been reused returns OALGPU_ERR_INVALID. */
int oalgpu_read_output_async(oalgpu_context *ctx, uint32_t *ticket);
int oalgpu_output_wait(oalgpu_context *ctx, uint32_t ticket, float *out, size_t out_floats);

/* What the rest of the reference has to hear about the voices, without reading them all back every update: the voices whose
 * play state (a source that runs out of buffer sets itself Stopping, core/voice.cpp:1201-1232), current buffer or count of
 * buffers played through (a streaming source, voice.cpp:1182-1218) changed since the last report -- each with its position at
 * that moment.  oalgpu_voice_events_async queues one small kernel behind the update submitted last (it writes the changes
 * into a slot of pinned host memory) and returns a ticket; oalgpu_voice_events_wait hands the report over, normally updates
 * later, when it has long landed.  Four tickets may be outstanding.  A report holds 1024 changes: beyond that
 * oalgpu_voice_events_wait fails with OALGPU_ERR_CAPACITY (count is still set) and the host reads the voices back
 * (oalgpu_voices_readback); what the report missed is not reported again.  A voice the host has just started (Playing, on a
 * buffer, not Playing in the report before) is no news and is not reported. */
typedef struct oalgpu_voice_event {
    uint32_t voice;
    int32_t  play_state;
    int32_t  has_buffer;
    int32_t  current_buffer;
    uint32_t buffers_done;
    int32_t  position;
    uint32_t position_frac;
    int32_t  fading;
} oalgpu_voice_event;
int oalgpu_voice_events_async(oalgpu_context *ctx, uint32_t *ticket);
int oalgpu_voice_events_wait(oalgpu_context *ctx, uint32_t ticket, oalgpu_voice_event *out, size_t capacity, size_t *count);

/* Timing of the last oalgpu_mix_update/mix_voices launch sequence, measured with HIP events on
 * the context's stream: total milliseconds, and the share of the voice kernel. */
int oalgpu_last_update_ms(oalgpu_context *ctx, float *total_ms, float *voice_kernel_ms);
/* Enables/disables the event timing above (off by default: it adds two event records). */
int oalgpu_set_timing(oalgpu_context *ctx, int enable);
/* ---- the EffectStates of alc/effects/ besides the reverbs -----------------------------------------
 * EffectState::deviceUpdate / update / process (core/effects/base.h:197-209) of alc/effects/{equalizer,modulator,
 * echo,dedicated}.cpp.  create = deviceUpdate; update takes the effect's EFX properties (core/effects/base.h:
 * 116-169) and what update() derives from the ambisonic layer, resolved by the caller:
 *   equalizer, modulator  target_channels[num_in] / gains[num_in] = mChans[i].mTargetChannel / mTargetGain
 *                         (MixParams::setAmbiMixParams(slot->Wet, slot->Gain, ...); OALGPU_INVALID_CHANNEL: unused)
 *   echo                  gains[2][num_out_lines] = mGains[tap].Target (ComputePanGains of the two taps)
 *   dedicated             props = NULL, gains[num_out_lines] = mTargetGains (Gain on the dialog / LFE line)
 *   compressor            props = oalgpu_compressor_props; target_channels / gains as for the equalizer
 *                         (alc/effects/compressor.cpp: envelope follower on wet channel 0, no gain ramp)
 * process: wet_in = the slot's wet bus (num_in x 1024; echo and dedicated use channel 0), out_lines = num_out_lines
 * x 1024, added to.  Like the reverb these keep the reference's operation order in both math modes: the output
 * is bit-identical to the reference's, except the modulator's sinusoid carrier (the GPU's sinf against libm). */
enum oalgpu_effect_kind {
    OALGPU_EFFECT_EQUALIZER = 0, OALGPU_EFFECT_MODULATOR, OALGPU_EFFECT_ECHO, OALGPU_EFFECT_DEDICATED, OALGPU_EFFECT_COMPRESSOR,
    OALGPU_EFFECT_CHORUS, OALGPU_EFFECT_DISTORTION, OALGPU_EFFECT_AUTOWAH, OALGPU_EFFECT_VMORPHER, OALGPU_EFFECT_FSHIFTER,
    OALGPU_EFFECT_PSHIFTER
};
enum oalgpu_modulator_waveform { OALGPU_MODULATOR_SINUSOID = 0, OALGPU_MODULATOR_SAWTOOTH, OALGPU_MODULATOR_SQUARE };
#define OALGPU_INVALID_CHANNEL 0xffffffffu
typedef struct oalgpu_equalizer_props {              /* EqualizerProps */
    float low_cutoff, low_gain, mid1_center, mid1_gain, mid1_width, mid2_center, mid2_gain, mid2_width, high_cutoff, high_gain;
} oalgpu_equalizer_props;
typedef struct oalgpu_modulator_props { float frequency, high_pass_cutoff; int32_t waveform; } oalgpu_modulator_props;
typedef struct oalgpu_echo_props { float delay, lr_delay, damping, feedback, spread; } oalgpu_echo_props;
typedef struct oalgpu_compressor_props { int32_t on_off; } oalgpu_compressor_props;
/* The rest of alc/effects/ (core/effects/base.h:93-169), same boundary:
 *   chorus / flanger    ChorusState (chorus.cpp): B- to A-Format, four LFO-modulated feedback delay lines, back to B-Format
 *   distortion          DistortionState (distortion.cpp): 4x oversampled low-pass, wave shaper, band-pass
 *   autowah             AutowahState (autowah.cpp): envelope follower on wet channel 0 drives a peaking filter per sample
 *   vocal morpher       VmorpherState (vmorpher.cpp): two 4-band formant filters blended by an LFO (phonemes A E I O U have
 *                       formants, the others are FormantFilter{} as in the reference)
 *   frequency shifter   FshifterState (fshifter.cpp): 1024-point STFT in double precision, analytic signal, phase rotation
 * update: target_channels[c] / gains[c] = mChans[c].mTargetChannel / mTargetGain as for the equalizer (the distortion's
 * gain includes props.gain, distortion.cpp:172).  The chorus, the distortion and the frequency shifter work on a
 * first-order A-Format (four lines); on a device above first order their deviceUpdate installs an up-sampler
 * (chorus.cpp:143-162) -- oalgpu_effect_set_upsampler(order_scales = AmbiScale::GetHFOrderScales(1, device order, 2D),
 * xover_norm = mXOverFreq / rate), after which update's gains are [4][num_out_lines] = ComputePanGains(target.Main,
 * AmbiScale::FirstOrderUp[c], gain) (:239-250) and process ends in BandSplitter::processHfScale + MixSamples onto every
 * line (:393-411).  NULL order_scales: first order again.  Call it before the first update.
 * Bit-identical to the reference except where libm's sinf / cosf meet the GPU's (chorus sinusoid LFO, autowah filter
 * coefficients, the morpher's sinusoid LFO: evaluated through double precision, 1e-6 of the block maximum or less).
 *   pitch shifter       PshifterState (pshifter.cpp): 1024-point STFT phase vocoder, hop 128, up to 9 wet channels (second
 *                       order); its up-sampler (a device above SECOND order) takes order_scales = GetHFOrderScales(2, ..)
 *                       and gains [9][num_out_lines] from AmbiScale::SecondOrderUp.  The transforms are not pffft's
 *                       butterfly order and atan2f / hypotf / sinf / cosf are the GPU's, so parity is a tolerance: 1e-4 of
 *                       the block maximum over runs of 20 blocks (tests/test_effects2.py; measured a few 1e-5). */
enum oalgpu_chorus_waveform { OALGPU_CHORUS_SINUSOID = 0, OALGPU_CHORUS_TRIANGLE };
enum oalgpu_vmorpher_waveform { OALGPU_VMORPHER_SINUSOID = 0, OALGPU_VMORPHER_TRIANGLE, OALGPU_VMORPHER_SAWTOOTH };
enum oalgpu_fshifter_direction { OALGPU_FSHIFTER_DOWN = 0, OALGPU_FSHIFTER_UP, OALGPU_FSHIFTER_OFF };
typedef struct oalgpu_chorus_props { int32_t waveform, phase; float rate, depth, feedback, delay; } oalgpu_chorus_props;
typedef struct oalgpu_distortion_props { float edge, gain, lowpass_cutoff, eq_center, eq_bandwidth; } oalgpu_distortion_props;
typedef struct oalgpu_autowah_props { float attack_time, release_time, resonance, peak_gain; } oalgpu_autowah_props;
typedef struct oalgpu_vmorpher_props {              /* VmorpherProps; phonemes in VMorpherPhenome order (A E I O U = 0..4) */
    float rate; int32_t phoneme_a, phoneme_b, phoneme_a_coarse_tuning, phoneme_b_coarse_tuning, waveform;
} oalgpu_vmorpher_props;
typedef struct oalgpu_fshifter_props { float frequency; int32_t left_direction, right_direction; } oalgpu_fshifter_props;
typedef struct oalgpu_pshifter_props { int32_t coarse_tune, fine_tune; } oalgpu_pshifter_props;
typedef struct oalgpu_effect oalgpu_effect;
int  oalgpu_effect_create(int device, int math_mode, int kind, uint32_t sample_rate, uint32_t num_in_channels,
    uint32_t num_out_lines, oalgpu_effect **out);
void oalgpu_effect_destroy(oalgpu_effect *effect);
int  oalgpu_effect_update(oalgpu_effect *effect, const void *props, const uint32_t *target_channels, const float *gains);
int  oalgpu_effect_process(oalgpu_effect *effect, const float *wet_in, float *out_lines, uint32_t n);
int  oalgpu_effect_set_upsampler(oalgpu_effect *effect, const float order_scales[2], float xover_norm);
/* Attach to effect slot `slot` of a context (like oalgpu_slot_set_convolution); NULL detaches. */
int  oalgpu_slot_set_effect(oalgpu_context *ctx, uint32_t slot, oalgpu_effect *effect);

/* ---- SampleConverter (core/converter.h:15-59, core/converter.cpp:175-330) ------------------------
 * The format and rate converter of the capture side and of some backends -- the second consumer of the
 * resamplers: interleaved frames of `channels` samples of src_type (oalgpu_output_type = DevFmtType order) at
 * src_rate in, dst_type at dst_rate out, with the state SampleConverter carries between calls (prep samples,
 * fractional offset).  oalgpu_converter_convert is SampleConverter::convert: it writes up to dst_frames frames to
 * `dst`, advances *src past the frames it consumed, leaves in *src_frames what it did not take, and returns the
 * frames written (or a negative oalgpu_error); oalgpu_converter_available_out is availableOut().  Host buffers;
 * results are bit-identical to the reference's (the resamplers in the reference's operation order). */
typedef struct oalgpu_converter oalgpu_converter;
int  oalgpu_converter_create(int device, int src_type, int dst_type, uint32_t channels, uint32_t src_rate,
    uint32_t dst_rate, int resampler, oalgpu_converter **out);
void oalgpu_converter_destroy(oalgpu_converter *conv);
uint32_t oalgpu_converter_available_out(const oalgpu_converter *conv, uint32_t src_frames);
int  oalgpu_converter_convert(oalgpu_converter *conv, const void **src, uint32_t *src_frames, void *dst,
    uint32_t dst_frames);

/* Name of the HIP kernel oalgpu_mix_voices launches for this context (as a profiler shows it). */
const char *oalgpu_voice_kernel_name(oalgpu_context *ctx);

/* ------------------------------------------------------------------------------------------
 * Convolution reverb: ConvolutionState (alc/effects/convolution.cpp:253-716) behind
 * EffectState::deviceUpdate / update / process (core/effects/base.h:197-209), for float impulse
 * responses of 1 .. 8 channels (mono, stereo .. 7.1, first-order B-Format: one ChannelData each,
 * all fed from the slot's input channel) at any sample rate.  The first 128 taps run as a time-domain
 * FIR, the rest as 128-tap segments in the frequency domain (LDS FFT in place of common/pffft.cpp).
 * ---------------------------------------------------------------------------------------- */
typedef struct oalgpu_convolution oalgpu_convolution;
/* deviceUpdate(device, buffer), convolution.cpp:318-471: num_out_lines = lines of the target
 * bus (EffectTarget::Main, <= 32). */
int  oalgpu_convolution_create(int device, uint32_t num_out_lines, const float *ir, uint32_t ir_len,
    oalgpu_convolution **out);
/* The general form: `ir` holds ir_frames interleaved frames of `channels` samples at ir_rate; when
 * ir_rate differs from device_rate every channel is resampled once with the reference's polyphase
 * Kaiser-sinc resampler (PPhaseResampler, common/polyphase_resampler.cpp; convolution.cpp:351-362,
 * :412-422).  ir_rate = device_rate = 0: already at the device's rate. */
int  oalgpu_convolution_create_ex(int device, uint32_t num_out_lines, const float *ir, uint32_t ir_frames,
    uint32_t channels, uint32_t ir_rate, uint32_t device_rate, oalgpu_convolution **out);
void oalgpu_convolution_destroy(oalgpu_convolution *conv);
/* PPhaseResampler::init(src_rate, dst_rate) + process(in, out) on host memory (pure host code) */
int  oalgpu_polyphase_resample(uint32_t src_rate, uint32_t dst_rate, const double *in, size_t n_in, double *out,
    size_t n_out);
/* The result of update(), convolution.cpp:474-621: mChans[0].Target (ComputePanGains of the
 * response's direction times the slot gain, computed by the caller). */
int  oalgpu_convolution_set_target_gains(oalgpu_convolution *conv, const float *gains);
/* ... for every channel of the response: gains[channel][num_out_lines] = mChans[c].Target (a stereo
 * response's two panned positions, a B-Format one's rotated and scaled rows, convolution.cpp:485-621) */
int  oalgpu_convolution_set_channel_gains(oalgpu_convolution *conv, const float *gains);
/* mMix = UpsampleMix (convolution.cpp:306-316, chosen by update() :489-513 when the device's ambisonic
 * order is above the response's): every channel goes through its BandSplitter::processScale(hf_scales[c],
 * lf_scales[c]) in front of the mix.  xover_norm = device->mXOverFreq / sample rate (:364).  NULL scales:
 * back to NormalMix. */
int  oalgpu_convolution_set_upsample(oalgpu_convolution *conv, const float *hf_scales, const float *lf_scales,
    float xover_norm);
/* process(samplesToDo, samplesIn, samplesOut), convolution.cpp:623-716, host buffers: wet_in =
 * channel 0 of the slot's wet bus (n samples), out_lines = num_out_lines x 1024, added to. */
int  oalgpu_convolution_process(oalgpu_convolution *conv, const float *wet_in, float *out_lines, uint32_t n);
/* The same on device memory, asynchronous on `hip_stream` (NULL = the default stream). */
int  oalgpu_convolution_process_device(oalgpu_convolution *conv, void *hip_stream, const float *wet_in_dev,
    float *out_lines_dev, uint32_t n);
/* Attach to effect slot `slot` of a context: oalgpu_mix_update / oalgpu_post_process then run it
 * between the bus reduction and the HRTF post-process, from the slot's wet bus (channel 0) into
 * the dry lines (alc/alu.cpp:2209-2257).  NULL detaches. */
int  oalgpu_slot_set_convolution(oalgpu_context *ctx, uint32_t slot, oalgpu_convolution *conv);

/* ------------------------------------------------------------------------------------------
 * EAX / standard reverb: ReverbState (alc/effects/reverb.cpp:567-1883) behind
 * EffectState::deviceUpdate / update / process, for a first-order (<= 4 line up-mix-free) or
 * wider target bus without ambisonic up-mixing (mUpmixOutput == false, :835-844).
 *   - update() is host work (libm parameter design, :858-1351) and runs on the CPU here too;
 *     the block it leaves behind is oalgpu_reverb_params.
 *   - process() -- B-Format -> A-Format, early reflections, the modulated feedback delay
 *     network, T60 filters, the vector all-pass, the pipeline cross-fade and the panned mix-out
 *     (:1396-1883) -- is one HIP launch; the delay lines stay resident in HBM.
 * ---------------------------------------------------------------------------------------- */
typedef struct oalgpu_reverb_props {             /* ReverbProps, core/effects/base.h:62-86 */
    float density, diffusion, gain, gain_hf, gain_lf, decay_time, decay_hf_ratio, decay_lf_ratio;
    float reflections_gain, reflections_delay, reflections_pan[3];
    float late_reverb_gain, late_reverb_delay, late_reverb_pan[3];
    float echo_time, echo_depth, modulation_time, modulation_depth, air_absorption_gain_hf;
    float hf_reference, lf_reference, room_rolloff_factor;
    int32_t decay_hf_limit;
} oalgpu_reverb_props;

typedef struct oalgpu_bq_coeffs { float b0, b1, b2, a1, a2; } oalgpu_bq_coeffs;  /* BiquadFilter::mCoeffs */
typedef struct oalgpu_reverb_pipeline {          /* ReverbPipeline fields update() computes, :502-548 */
    oalgpu_bq_coeffs filter_lp, filter_hp;       /* mFilter[*].Lp / .Hp */
    uint32_t early_delay_tap[4][2];              /* mEarlyDelayTap: [0] current, [1] target */
    float    early_delay_coeff[2];               /* mEarlyDelayCoeff */
    uint32_t late_delay_tap[4][2];               /* mLateDelayTap */
    float    mix_x, mix_y;                       /* mMixX, mMixY */
    float    early_ap_coeff;                     /* mEarly.Allpass.Coeff */
    uint32_t early_ap_offset[4];                 /* mEarly.Allpass.Offset */
    uint32_t early_offset[4];                    /* mEarly.Offset */
    float    early_coeff;                        /* mEarly.Coeff */
    float    early_gains_target[4][OALGPU_MAX_AMBI_CHANNELS];   /* mEarly.Gains[j].Target */
    uint32_t late_offset[4];                     /* mLate.Offset */
    float    late_density_gain;                  /* mLate.DensityGain */
    float    t60_mid_gain[4];                    /* mLate.T60[j].mMidGain */
    oalgpu_bq_coeffs t60_hf[4], t60_lf[4];       /* mLate.T60[j].mHFFilter / mLFFilter */
    uint32_t mod_step;                           /* mLate.Mod.Step */
    float    mod_depth;                          /* mLate.Mod.Depth */
    float    late_ap_coeff;                      /* mLate.VecAp.Coeff */
    uint32_t late_ap_offset[4];                  /* mLate.VecAp.Offset */
    float    late_gains_target[4][OALGPU_MAX_AMBI_CHANNELS];    /* mLate.Gains[j].Target */
    uint32_t fade_sample_count;                  /* mFadeSampleCount */
} oalgpu_reverb_pipeline;
typedef struct oalgpu_reverb_params {
    int32_t pipeline_state;                      /* ReverbState::PipelineState, :589-596 */
    int32_t current_pipeline;                    /* mCurrentPipeline */
    oalgpu_reverb_pipeline pipe[2];
} oalgpu_reverb_params;
enum oalgpu_reverb_pipeline_state {
    OALGPU_REVERB_DEVICE_CLEAR = 0, OALGPU_REVERB_START_FADE = 1, OALGPU_REVERB_FADING = 2,
    OALGPU_REVERB_CLEANUP = 3, OALGPU_REVERB_NORMAL = 4
};

typedef struct oalgpu_reverb oalgpu_reverb;
/* deviceUpdate, :822-852 (allocLines :728-820).  device < 0 creates a parameter-only instance
 * (update / get_params / line_lengths work, process fails with OALGPU_ERR_NO_DEVICE): the host
 * half can be exercised without a GPU. */
int  oalgpu_reverb_create(int device, uint32_t sample_rate, uint32_t num_out_lines, oalgpu_reverb **out);
void oalgpu_reverb_destroy(oalgpu_reverb *rev);
/* deviceUpdate on a device above first order (reverb.cpp:835-851): mUpmixOutput -- process() then ends in
 * MixOutAmbiUp (:658-699: A-to-B-Format rows, BandSplitter::processHfScale per row, gains that pan and
 * upsample) and update3DPanning combines its transforms with the first-order upsample matrix (:1166-1184).
 * The ambisonic layer's constants come from the caller: order_scales[0..1] = AmbiScale::GetHFOrderScales(1,
 * device order, 2D mixing), first_order_up = AmbiScale::FirstOrderUp (4 x 25, core/ambidefs.cpp),
 * xover_norm = device->mXOverFreq / frequency.  NULL order_scales: MixOutPlain again.  Takes effect with
 * the next oalgpu_reverb_update. */
int  oalgpu_reverb_set_upmix(oalgpu_reverb *rev, const float order_scales[2], const float *first_order_up,
    float xover_norm);
/* The stream update() and process() enqueue on (NULL = the default stream).  A reverb attached
 * to a context slot uses the context's effect stream. */
int  oalgpu_reverb_set_stream(oalgpu_reverb *rev, void *hip_stream);
/* OALGPU_MATH_EXACT (default of a free-standing instance): bit-identical to ReverbState::process.  OALGPU_MATH_FAST: the
 * master band-pass and the T60 damping filters (the two serial recurrences that are most of a block's time) run as
 * block scans -- the reference's output to rounding level.  An instance attached to a slot of a FAST context
 * (oalgpu_slot_set_reverb) is switched to FAST, one attached to an EXACT context to EXACT. */
int  oalgpu_reverb_set_math_mode(oalgpu_reverb *rev, int math_mode);
/* ReverbState::update, :1222-1351, for an identity first-order target map
 * (ComputePanGains with AmbiMap[i] = {1, i}, core/mixer.cpp:93-103). */
int  oalgpu_reverb_update(oalgpu_reverb *rev, const oalgpu_reverb_props *props, float slot_gain);
/* The block as the host mirror holds it (what update() wrote, and what process() has since
 * advanced: taps, coefficients, fade count, pipeline state). */
int  oalgpu_reverb_get_params(oalgpu_reverb *rev, oalgpu_reverb_params *out);
/* Installs a block computed elsewhere (a host that keeps its own ReverbState::update): only the
 * fields update() writes are taken, as after one update() call. */
int  oalgpu_reverb_set_params(oalgpu_reverb *rev, const oalgpu_reverb_params *params);
/* The 11 delay-line lengths of allocLines (in floats, 4 lines each); returns their sum. */
int  oalgpu_reverb_line_lengths(oalgpu_reverb *rev, uint32_t lengths[11]);
/* process(samplesToDo, samplesIn, samplesOut), :1813-1883, host buffers: wet_in = 4 x 1024 (the
 * slot's B-Format wet bus), out_lines = num_out_lines x 1024, added to. */
int  oalgpu_reverb_process(oalgpu_reverb *rev, const float *wet_in, float *out_lines, uint32_t n);
/* The same on device memory, asynchronous on the reverb's stream. */
int  oalgpu_reverb_process_device(oalgpu_reverb *rev, const float *wet_in_dev, float *out_lines_dev, uint32_t n);
/* Parameter-only instances: the scalar bookkeeping of one process(n) call (tap and coefficient
 * hand-over, fade countdown, pipeline state machine, :1840-1882) without any audio, so that the
 * host half can be followed through a schedule of updates on a machine without a GPU. */
int  oalgpu_reverb_skip(oalgpu_reverb *rev, uint32_t n);
/* Attach to effect slot `slot` of a context (like oalgpu_slot_set_convolution): wet bus lines
 * 0..3 in, dry lines out.  NULL detaches. */
int  oalgpu_slot_set_reverb(oalgpu_context *ctx, uint32_t slot, oalgpu_reverb *rev);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* OALGPU_H */
