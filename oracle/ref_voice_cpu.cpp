/* oracle/ref_voice_cpu.cpp -- TEST INFRASTRUCTURE (part of oracle/_ref/liboalbridge.so).
 *
 * The reference's core/voice.cpp compiled IN PLACE (an #include of the file where it lies under
 * /root/reference; nothing is copied) with one identifier renamed: Voice::mix becomes Voice::mix_cpu
 * in this translation unit, so that oracle/ref_bridge.cpp can define the real Voice::mix -- the symbol
 * alc/alu.cpp's ProcessContexts calls for every voice (alu.cpp:2201-2206) -- as the seam a maintainer
 * would patch, and route it either to the reference's own code (here) or to liboalgpu.so.
 * Being inside this translation unit also reaches the file-static HRTF mixer pointers
 * (MixHrtfSamples / MixHrtfBlendSamples, core/voice.cpp:80-81), which Voice::InitMixer alone sets. */
#define mix mix_cpu
#include "core/voice.cpp"
#undef mix

extern "C" {

void oalbridge_voice_mix_cpu(void *voice, int vstate, void *context, long long device_ns, unsigned samples_to_do)
{
    static_cast<Voice*>(voice)->mix_cpu(static_cast<Voice::State>(vstate), static_cast<ContextBase*>(context),
        std::chrono::nanoseconds{device_ns}, samples_to_do);
}

/* HrtfMixerFunc / HrtfMixerBlendFunc, core/voice.cpp:73-81 */
void oalbridge_get_hrtf_mixers(void **mixfn, void **blendfn)
{
    *mixfn = reinterpret_cast<void*>(MixHrtfSamples);
    *blendfn = reinterpret_cast<void*>(MixHrtfBlendSamples);
}
void oalbridge_set_hrtf_mixers(void *mixfn, void *blendfn)
{
    MixHrtfSamples = reinterpret_cast<HrtfMixerFunc>(mixfn);
    MixHrtfBlendSamples = reinterpret_cast<HrtfMixerBlendFunc>(blendfn);
}

} /* extern "C" */
