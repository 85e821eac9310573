/* oalgpu_debug.h -- MEASUREMENT AIDS of liboalgpu.so, kept apart from the drop-in boundary (oalgpu.h).
 *
 * Nothing here stands in for an interface of the reference: these entry points exist so that bench.py, tools/ and the
 * profiles under profiles/ can state what the product path costs (the event clock's floor, the pipelined host loop
 * without a language binding's per-call cost, the per-phase cycle stamps of the voice kernel's measurement variant).
 * A host that integrates the library never includes this file.
 */
#ifndef OALGPU_DEBUG_H
#define OALGPU_DEBUG_H

#include "oalgpu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Measurement aid: `updates` pipelined updates driven from C++ exactly as section 3c of INTEGRATION.md writes them
 * (oalgpu_voice_move_async of moves[u % move_sets] -- `count` records each --, oalgpu_mix_update, oalgpu_read_output_async,
 * oalgpu_output_wait of the update two back into `out`), so that the boundary's throughput can be stated without a
 * language binding's per-call cost.  wall_s: the loop's duration; busy_s: the calling thread's time outside
 * oalgpu_output_wait. */
int oalgpu_debug_pipelined_run(oalgpu_context *ctx, const oalgpu_voice_move *moves, size_t count, uint32_t move_sets,
    uint32_t updates, uint32_t samples_to_do, int post_process, float *out, size_t out_floats, double *wall_s, double *busy_s);

/* Measurement aid: the same three submitting calls with every update's output waited for (untimed) before the next is
 * submitted -- what they cost the calling thread when no call waits behind a full queue.  submit_s: seconds per update. */
int oalgpu_debug_submit_cost(oalgpu_context *ctx, const oalgpu_voice_move *moves, size_t count, uint32_t move_sets,
    uint32_t updates, uint32_t samples_to_do, int post_process, float *out, size_t out_floats, double *submit_s);


/* The floor of that clock: an EMPTY kernel (one wavefront that returns) dispatched on the context's stream and timed
 * the same way as the voice kernel -- HIP events bound to the dispatch (hipExtLaunchKernel); the median of `reps`
 * dispatches.  Whatever the events include besides a kernel's own run time is in this figure too. */
int oalgpu_debug_event_floor_ms(oalgpu_context *ctx, uint32_t reps, float *ms);

/* OALGPU_CTX_PROFILE contexts: which stages the voice kernel's measurement variant skips (1 FIR, 2 resampler, 8 direct
 * filter, 16 FIR input build); 0 = none. */
int oalgpu_debug_set_ablate(oalgpu_context *ctx, uint32_t mask);
/* OALGPU_CTX_PROFILE contexts: the [voice][8] s_memtime stamps the measurement variant recorded in the last launch
 * (out: max_voices x 8 words). */
int oalgpu_debug_phase_times(oalgpu_context *ctx, unsigned long long *out);
/* ... and the [wavefront][8] stamps behind them (kernel entry, first voice requested and parked, last voice done, partial
 * bus stored, then pass 0 in detail); out: max_voices x 8 words, *waves receives the number of wavefronts. */
int oalgpu_debug_wave_times(oalgpu_context *ctx, unsigned long long *out, uint32_t *waves);

#ifdef __cplusplus
}
#endif
#endif /* OALGPU_DEBUG_H */
