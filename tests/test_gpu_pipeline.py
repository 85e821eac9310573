"""The two-stream update pipeline against the one-stream entry points, both on the GPU.

oalgpu_mix_update (voice kernel of update k+1 beside the partial-bus reduction and the post-process
of update k; cross-stream events without a system-scope fence; the 4-wave reduction that sits
beside the voice kernel) must produce, bit for bit, what oalgpu_mix_voices + oalgpu_post_process
produce with a host synchronisation after every update: same kernels, same summation order -- only
the scheduling differs.  The pipelined scene is never synchronised between its checkpoints, so a
missing dependency or a stale read shows up as a difference in the buses, the carried HRTF
accumulator or the voice states.  BASELINE configs[2] geometry (4096 voices, the bench scene)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V = 4096
UPDATES = 48
CHECK = (0, 1, 2, 7, 23, 47)
pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("variant", ["default", "apply-in-voice-kernel", "fused-reduce"])
def test_pipelined_update_equals_serial_update(synth_mhr, variant):
    """variant "apply-in-voice-kernel" (OALGPU_CTX_APPLY_IN_VOICE_KERNEL): every update is submitted one library call late and
    the parameter block that follows it is installed by the update's own wavefronts, behind the voices they mixed -- the same
    operations as ApplyParamsKernel, so still the serial scene's bits.  "fused-reduce" (OALGPU_CTX_FUSED_REDUCE): the reduction and
    the post-process of an update as one launch -- the same sums in the same order."""
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    api = oalgpu.Api(oalgpu.MATH_FAST)
    flags = {"default": 0, "apply-in-voice-kernel": oalgpu.CTX_APPLY_IN_VOICE_KERNEL, "fused-reduce": oalgpu.CTX_FUSED_REDUCE}[variant]
    papi = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=flags) if flags else api
    mhr = synth.synth_mhr_bytes()
    api._mhr = mhr
    papi._mhr = mhr

    def build(api=api):
        sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
        allv = list(range(V))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(UPDATES)]
        return sc, blocks

    piped, pblocks = build(papi)
    serial, sblocks = build()
    assert piped.voice_kernel_name() in ("VoiceBlockKernel",) or "VoiceWave" in piped.voice_kernel_name()
    got, want = {}, {}
    for k in range(UPDATES):
        piped.apply_block(pblocks[k])
        piped.mix(1024, post_process=True)              # oalgpu_mix_update: no host sync
        serial.apply_block(sblocks[k])
        serial.mix_voices(1024)
        serial.post_process(1024)
        serial.sync()
        if k in CHECK:                                  # the reads synchronise the pipelined scene
            got[k] = (piped.dry().copy(), piped.hrtf_accum().copy())
            want[k] = (serial.dry().copy(), serial.hrtf_accum().copy())
    for k in CHECK:
        assert np.array_equal(_bits(got[k][0]), _bits(want[k][0])), f"bus block differs after update {k}"
        assert np.array_equal(_bits(got[k][1]), _bits(want[k][1])), f"HRTF accumulator differs after update {k}"
        assert np.abs(want[k][0]).max() > 1e-3                       # the comparison is not of silence
    for v in range(0, V, 97):
        a, b = piped.voice_state(v), serial.voice_state(v)
        assert (a.play_state, a.position, a.position_frac) == (b.play_state, b.position, b.position_frac), v
        assert np.array_equal(_bits(a.hrtf_history), _bits(b.hrtf_history)), v
        assert np.array_equal(_bits(a.prev_samples), _bits(b.prev_samples)), v
    piped.close()
    serial.close()


@pytest.mark.parametrize("config", [2, 4, 5])
def test_parameter_block_installed_by_the_dry_line_and_send_kernels(synth_mhr, config):
    """OALGPU_CTX_APPLY_IN_VOICE_KERNEL on the other voice kernels -- dry lines in registers (config 2), stream rows with sends
    (config 4), HRTF with a send (config 5): the update's own wavefronts install the next parameter block behind the voices they
    mixed; bit for bit the scene whose blocks go through ApplyParamsKernel, one update after the other with a host synchronisation."""
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    nv, updates = 2048, 12
    mhr = synth.synth_mhr_bytes()

    def build(flags):
        api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=flags)
        api._mhr = mhr
        sc, script = bench.build_scene(oalgpu, synth, api, config, nv, 0, mhr, 0)
        allv = list(range(nv))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(updates)]
        return sc, blocks

    piped, pblocks = build(oalgpu.CTX_APPLY_IN_VOICE_KERNEL)
    serial, sblocks = build(0)
    nslots = {4: 4, 5: 1}.get(config, 0)
    for k in range(updates):
        piped.apply_block(pblocks[k])
        piped.mix(1024, post_process=True)
        serial.apply_block(sblocks[k])
        serial.mix_voices(1024)
        serial.post_process(1024)
        serial.sync()
        if k in (0, 1, 5, updates - 1):
            a, b = piped.dry().copy(), serial.dry().copy()
            assert np.array_equal(_bits(a), _bits(b)), f"config {config}: bus block differs after update {k}"
            assert np.abs(b).max() > 1e-4
            for s in range(nslots):
                assert np.array_equal(_bits(piped.wet(s)), _bits(serial.wet(s))), f"config {config}: wet bus {s} differs after update {k}"
    for v in range(0, nv, 61):
        a, b = piped.voice_state(v), serial.voice_state(v)
        assert (a.play_state, a.position, a.position_frac) == (b.play_state, b.position, b.position_frac), v
        assert np.array_equal(_bits(a.prev_samples), _bits(b.prev_samples)), v
        assert np.array_equal(_bits(a.dry_current), _bits(b.dry_current)), v
    piped.close()
    serial.close()


def test_parameter_block_installed_by_the_128_tap_kernel():
    """OALGPU_CTX_APPLY_IN_VOICE_KERNEL on a data set of more than 64 taps (VoiceWaveKernel<18, 128, ...>): the epilogue's fast
    install moves one tap pair per lane, so such responses are blended at install (ApplyRecordLean) -- with the pre-blended rows
    only the first 64 taps of a moved voice's target were replaced.  Bit for bit the scene whose blocks go through
    ApplyParamsKernel."""
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    nv, updates = 256, 6
    mhr = synth.synth_mhr_bytes(ir_size=128)

    def build(flags):
        api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=flags)
        api._mhr = mhr
        sc, script = bench.build_scene(oalgpu, synth, api, 3, nv, 0, mhr, 0)
        allv = list(range(nv))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(updates)]
        return sc, blocks

    piped, pblocks = build(oalgpu.CTX_APPLY_IN_VOICE_KERNEL)
    serial, sblocks = build(0)
    assert "128" in piped.voice_kernel_name(), piped.voice_kernel_name()
    for k in range(updates):
        piped.apply_block(pblocks[k])
        piped.mix(1024, post_process=True)
        serial.apply_block(sblocks[k])
        serial.mix_voices(1024)
        serial.post_process(1024)
        serial.sync()
        a, b = piped.hrtf_accum().copy(), serial.hrtf_accum().copy()
        assert np.array_equal(_bits(a), _bits(b)), f"HRTF accumulator differs after update {k}"
        assert np.abs(b).max() > 1e-4
    piped.close()
    serial.close()
