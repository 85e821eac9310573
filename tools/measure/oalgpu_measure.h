/* oalgpu_measure.h -- MEASUREMENT AIDS, not part of the product (liboalgpu.so exports none of this).
 *
 * Two pieces, both built by `make measure` in openal-soft_amd/:
 *   tools/measure/liboalmeasure.so   -- oalmeasure_*: loops and clocks written against the PUBLIC C-ABI only (oalgpu.h): the
 *       pipelined host loop without a language binding's per-call cost, the event clock's floor.  bench.py uses them for
 *       config.e2e_throughput.native_loop and roofline.event_floor_ms.
 *   openal-soft_amd/liboalgpu_measure.so -- the product's sources built with -DOALGPU_MEASUREMENT: the same library plus
 *       oalgpu_debug_set_ablate / _phase_times / _wave_times and oalgpu_reverb_debug_*, which read the voice kernel's measurement variant's stamps
 *       (OALGPU_CTX_PROFILE contexts) out of the context's internals.  tools/phase_times*.py load it (OALGPU_LIB).
 */
#ifndef OALGPU_MEASURE_H
#define OALGPU_MEASURE_H

#include "../../include/oalgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* `updates` pipelined updates driven from C++ exactly as section 3c of INTEGRATION.md writes them (oalgpu_voice_move_async of
 * moves[u % move_sets] -- `count` records each --, oalgpu_mix_update, oalgpu_read_output_async, oalgpu_output_wait of the update
 * two back into `out`), so that the boundary's throughput can be stated without a language binding's per-call cost.
 * wall_s: the loop's duration; busy_s: the calling thread's time outside oalgpu_output_wait. */
int oalmeasure_pipelined_run(oalgpu_context *ctx, const oalgpu_voice_move *moves, size_t count, uint32_t move_sets,
    uint32_t updates, uint32_t samples_to_do, int post_process, float *out, size_t out_floats, double *wall_s, double *busy_s);

/* the same three submitting calls with every update's output waited for (untimed) before the next is submitted -- what they
 * cost the calling thread when no call waits behind a full queue.  submit_s: seconds per update. */
int oalmeasure_submit_cost(oalgpu_context *ctx, const oalgpu_voice_move *moves, size_t count, uint32_t move_sets,
    uint32_t updates, uint32_t samples_to_do, int post_process, float *out, size_t out_floats, double *submit_s);

/* The floor of the clock the library times its voice kernel with: an EMPTY kernel (one wavefront that returns) dispatched on
 * the context's stream with HIP events bound to the dispatch (hipExtLaunchKernel); the median of `reps` dispatches. */
int oalmeasure_event_floor_ms(oalgpu_context *ctx, uint32_t reps, float *ms);

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif
/* ---- liboalgpu_measure.so only (-DOALGPU_MEASUREMENT) ---- */
/* OALGPU_CTX_PROFILE contexts: which stages the voice kernel's measurement variant skips (1 FIR, 2 resampler, 8 direct
 * filter, 16 FIR input build); 0 = none. */
int oalgpu_debug_set_ablate(oalgpu_context *ctx, uint32_t mask);
/* OALGPU_CTX_PROFILE contexts: the [voice][8] s_memtime stamps the measurement variant recorded in the last launch
 * (out: max_voices x 8 words). */
int oalgpu_debug_phase_times(oalgpu_context *ctx, unsigned long long *out);
/* ... and the [wavefront][8] stamps behind them (kernel entry, first voice requested and parked, last voice done, partial
 * bus stored, then pass 0 in detail); out: max_voices x 8 words, *waves receives the number of wavefronts. */
int oalgpu_debug_wave_times(oalgpu_context *ctx, unsigned long long *out, uint32_t *waves);
/* the EAX reverb kernel's cycle-counter stamps (a device instance): enable for the launches from here on, then read the last
 * launch's [4 roles][8 sub-blocks][8] words (tools/reverb_phase_times.py) */
int oalgpu_reverb_debug_enable_phase_times(oalgpu_reverb *rev);
int oalgpu_reverb_debug_phase_times(oalgpu_reverb *rev, unsigned long long *out);
#if defined(__GNUC__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* OALGPU_MEASURE_H */
