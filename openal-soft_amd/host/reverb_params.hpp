// Host half of the EAX reverb: everything ReverbState does with libm before a block mixes
// (deviceUpdate's line metrics, update()'s parameter design) plus the bookkeeping process()
// performs on scalars (tap/coefficient hand-over, fade countdown, pipeline state machine), so
// that the device kernel only ever receives plain numbers.  See reverb_params.cpp.
#pragma once
#include "../../include/oalgpu.h"

namespace oalgpu {

// allocLines, alc/effects/reverb.cpp:728-820: lengths (in floats, 4 lines each) of
// main delay, then per pipeline {late input, early all-pass, early delay, late vector all-pass,
// late delay}.  Returns the total.
size_t ReverbLineLengths(float frequency, uint32_t lengths[11]);

struct ReverbHost {
    uint32_t sampleRate{48000}, numLines{4};
    // mUpmixOutput (:835-843): the panning gains pan AND upsample first-order B-Format, through
    // AmbiScale::FirstOrderUp (4 x MaxAmbiChannels, handed in by the caller)
    bool upmix{false};
    float firstOrderUp[4][OALGPU_MAX_AMBI_CHANNELS]{};
    // ReverbState::mParams, :573-587
    struct Last {
        float density{1.0f}, diffusion{1.0f}, decayTime{1.49f}, hfDecayTime{0.83f * 1.49f},
            lfDecayTime{1.0f * 1.49f}, modulationTime{0.25f}, modulationDepth{0.0f},
            hfReference{5000.0f}, lfReference{250.0f};
    } last;
    oalgpu_reverb_params params{};          // pipeline_state / current_pipeline / both pipelines
    uint32_t modIndex[2]{0u, 0u};           // mLate.Mod.Index, advanced by process()
    uint64_t offset{0};                     // ReverbState::mOffset

    ReverbHost();
    // ReverbPipeline::clear (:550-564) on the scalar fields
    void clearPipeline(int which);
    // ReverbState::update, :1222-1351; returns true for a full update (pipeline swap)
    bool update(const oalgpu_reverb_props &props, float slotGain);
    // takes the fields update() writes from `src` (as after exactly one update() there)
    bool install(const oalgpu_reverb_params &src);

    // What one process(n) call does to the scalars (:1840-1882, :1577-1585, :1757-1759, :1680).
    struct Step {
        int current;        // pipeline processed first
        int oldMode;        // 0: old pipeline idle; 1: processed; 2: processed as its final block
                            // (targets dropped to 0 first); 3: cleared, not processed
        uint64_t offset;    // mOffset for this block
        uint32_t modIndex[2];
    };
    // begin(): the state machine and what it does to the old pipeline BEFORE the block is
    // processed (targets dropped for its final block, scalars cleared); the mirror then holds
    // exactly what the kernel must see.  finish(): the hand-overs process() performs.
    Step begin(uint32_t n);
    void finish(const Step &s, uint32_t n);
};

} // namespace oalgpu
