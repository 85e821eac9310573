// The resident launch of the HRTF hot path (OALGPU_CTX_RESIDENT): VoiceWaveKernel<..., RES = true> of voice_wave.hip in a
// translation unit of its own, because it is built with the machine-level loop-invariant code motion off
// (-mllvm -disable-machine-licm, Makefile).  The kernel's body is one update of a loop; with that pass on, whatever the body
// computes once per launch -- constants, addresses -- is hoisted in front of the update loop and kept in registers across the
// resampler and the FIR, whose peaks decide the register count: 249 VGPRs per lane instead of 217, and above 232 the
// post-stream kernels -- the reduction and the post-process, 48 registers -- no longer fit on a SIMD beside two voice
// wavefronts: kernels that wait for this one could then never start (tests/test_kernel_resources.py holds the budget).
#define OALGPU_WAVE_NO_LAUNCHER
#include "voice_wave.hip"

namespace oalgpu {
// Which layouts have a resident launch, and the launch itself (evStart / evStop: HIP events bound to the dispatch, or null).
bool WaveKernelHasResident(const DeviceLayout &L)
{
    // (the two-voices-per-wavefront kernel, or the 16-wavefront form of the voice-per-wavefront kernel: voice_wave16.hip)
    return L.hrtf && L.numSends == 0 && L.firMfma && L.irStride >= 8 && L.irStride <= 64 && L.accLines == 0 && (L.wave16 == 0 || L.wave16 == 16u);
}

hipError_t LaunchVoiceWaveResident(hipStream_t s, const DeviceLayout &L, const ResidentArgs &args, hipEvent_t evStart, hipEvent_t evStop)
{
    if(L.wave16) return LaunchVoiceWave16Resident(s, L, args, evStart, evStop);
    const NextBlock next{nullptr, nullptr, nullptr, args};
    const WaveProf none{nullptr, 0u};
    hipExtLaunchKernelGGL((VoiceWaveKernel<17, 64, 0, false, true, false, WaveArgsHrtf, 0, true>), dim3(WaveKernelGroups(L)), dim3(kWThreads), 0, s,
        evStart, evStop, 0u, WaveArgsHrtf{L}, 0u, none, next);
    return hipGetLastError();
}

// workgroups of the resident kernel one compute unit holds at once (the launch needs ALL of its workgroups on the machine)
int WaveResidentGroupsPerCu(const DeviceLayout &L)
{
    if(L.wave16) return Wave16ResidentGroupsPerCu();
    int n = 0;
    if(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, VoiceWaveKernel<17, 64, 0, false, true, false, WaveArgsHrtf, 0, true>, kWThreads, 0) != hipSuccess)
    { (void)hipGetLastError(); return 0; }
    return n;
}

} // namespace oalgpu
