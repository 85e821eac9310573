/* include/oalgpu_openal_hooks.hpp -- the two optional hooks of the reference-side binding (include/oalgpu_openal.hpp), as
 * alc/alu.cpp sees them: two free functions the maintainer's build defines next to its BatchMixer (oracle/ref_bridge.cpp is that
 * build here; INTEGRATION.md section 3a shows the four-line diff of alc/alu.cpp).
 *
 *   oalgpu_hook::ParamsChanged(voice)      at the point CalcVoiceParams has decided to recompute a voice (alc/alu.cpp:2012-2023):
 *                                          the batch mixer then hands over exactly those voices' parameters (BatchMixer::
 *                                          noteParamsChanged) instead of comparing every voice's.
 *   oalgpu_hook::GetCoeffs(store, voice, ...)   in the place of device->mHrtf->getCoeffs(...) at CalcPanningAndFilters' three call
 *                                          sites (alc/alu.cpp:1214-1216, :1256-1258, :1296-1298), same arguments behind the two
 *                                          new ones: when a batch mixer takes the direction (BatchMixer::noteHrtfDirection) the
 *                                          reference neither indexes nor blends the responses -- the device context does, from 24
 *                                          bytes per moved voice --; otherwise it calls store.getCoeffs(...) as before.
 */
#ifndef OALGPU_OPENAL_HOOKS_H
#define OALGPU_OPENAL_HOOKS_H

#include <span>

#include "core/hrtf.h"
#include "core/mixer/hrtfdefs.h"

struct Voice;

namespace oalgpu_hook {

void ParamsChanged(Voice *voice) noexcept;
void GetCoeffs(const HrtfStore &store, Voice *voice, float elevation, float azimuth, float distance, float spread, HrirSpan coeffs,
    std::span<unsigned, 2> delays) noexcept;

} // namespace oalgpu_hook

#endif
