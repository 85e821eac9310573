"""The resident voice kernel (OALGPU_CTX_RESIDENT) against one launch per update, both on the GPU.

One launch of the HRTF voice kernel stays on its voices over many updates: every oalgpu_mix_update writes a doorbell slot
(the update's length and parameter block) and launches the update's reduction and post-process, which wait for device
counters; a workgroup starts update u + 1 when it is through with u.  The operations per voice and the summation orders are
those of the launched path, so buses, carried accumulator, output lines and every voice's state must be THE SAME BITS as
oalgpu_mix_voices + oalgpu_post_process with a host synchronisation after every update -- over 56 updates with a parameter
block each, updates of several lengths, launches that end by their own bound (every 9 updates) and by being parked (a
read-back in the middle), and outputs collected through the ring two updates late.  BASELINE configs[2] geometry (4096
voices, the bench scene); a second case runs the scene the compiled reference mixes (tests/test_gpu_baseline_configs.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V = 4096
UPDATES = 56
CHECK = (0, 1, 2, 7, 9, 10, 23, 40, 55)
SIZES = {5: 600, 17: 257, 30: 1000}          # updates shorter than a full line
pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _build(oalgpu, synth, bench, api, mhr, updates):
    sc, script = bench.build_scene(oalgpu, synth, api, 3, V, 0, mhr, 0)
    allv = list(range(V))
    moving = [v for v in allv if script.is_moving(v)]
    sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
    blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(updates)]
    return sc, blocks


FORMS = {"a voice per wavefront": 0, "two voices per wavefront": 256}      # 256: OALGPU_CTX_WAVE_PAIRS, the kernel of rounds 1-5


@pytest.mark.parametrize("form", list(FORMS))
def test_resident_updates_equal_launched_updates(synth_mhr, form):
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    mhr = synth.synth_mhr_bytes()
    rapi = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=oalgpu.CTX_RESIDENT | FORMS[form])
    sapi = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=FORMS[form])              # (the launched form of the SAME kernel)
    rapi._mhr = mhr
    sapi._mhr = mhr

    # ---- the resident scene first, alone: another context's entry points would park its kernel every time
    res, rblocks = _build(oalgpu, synth, bench, rapi, mhr, UPDATES)
    assert ("VoiceWave16Kernel<16>" if FORMS[form] == 0 else "VoiceWaveKernel<17, 64, 0, false, true>") in res.voice_kernel_name(), res.voice_kernel_name()
    res.resident_set_max_updates(9)                     # launches end by themselves every nine updates
    res.resident_set_timing(True)
    res.resident_set_short_run(0)                       # (the reads below keep the launches short: no falling back)
    got = {}
    for k in range(UPDATES):
        res.apply_block(rblocks[k])
        res.mix(SIZES.get(k, 1024), post_process=True)
        if k in CHECK:                                  # (the reads park the kernel and synchronise)
            got[k] = (res.dry().copy(), res.hrtf_accum().copy())
    info = res.resident_stats()
    assert info["enabled"] == 1 and info["failed"] == 0, info
    assert info["updates"] == UPDATES, info             # every update went through the doorbell
    assert info["launches"] >= UPDATES // 9, info
    assert info["timed_updates"] == UPDATES and info["timed_kernel_ms"] > 0.0, info
    rstate = {v: res.voice_state(v) for v in range(0, V, 97)}

    ser, sblocks = _build(oalgpu, synth, bench, sapi, mhr, UPDATES)
    want = {}
    for k in range(UPDATES):
        ser.apply_block(sblocks[k])
        ser.mix_voices(SIZES.get(k, 1024))
        ser.post_process(SIZES.get(k, 1024))
        ser.sync()
        if k in CHECK:
            want[k] = (ser.dry().copy(), ser.hrtf_accum().copy())
    for k in CHECK:
        assert np.array_equal(_bits(got[k][0]), _bits(want[k][0])), f"bus block differs after update {k}"
        assert np.array_equal(_bits(got[k][1]), _bits(want[k][1])), f"HRTF accumulator differs after update {k}"
        assert np.abs(want[k][0]).max() > 1e-3                       # the comparison is not of silence
    for v, a in rstate.items():
        b = ser.voice_state(v)
        assert (a.play_state, a.position, a.position_frac) == (b.play_state, b.position, b.position_frac), v
        assert np.array_equal(_bits(a.hrtf_history), _bits(b.hrtf_history)), v
        assert np.array_equal(_bits(a.prev_samples), _bits(b.prev_samples)), v
    res.close()
    ser.close()


@pytest.mark.parametrize("form", list(FORMS))
def test_resident_outputs_through_the_ring(synth_mhr, form):
    """every update's stereo output, collected two updates late through oalgpu_read_output_async / oalgpu_output_wait while the
    voice kernel stays resident (the post-process kernel fills the host's ring slot), against the launched path's lines"""
    import oalgpu
    from oalgpu import synth
    import bench
    mhr = synth.synth_mhr_bytes()
    updates = 24
    outs = {}
    for mode in ("resident", "launched"):
        api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=(oalgpu.CTX_RESIDENT if mode == "resident" else 0) | FORMS[form])
        api._mhr = mhr
        sc, blocks = _build(oalgpu, synth, bench, api, mhr, updates)
        got, tickets = [], []
        if mode == "resident":
            sc.resident_set_short_run(0)
            for k in range(updates):
                sc.apply_block(blocks[k])
                sc.mix(1024, post_process=True)
                tickets.append(sc.read_output_async())
                if k >= 2:
                    got.append(sc.output_wait(tickets[k - 2]).copy())
            for t in tickets[-2:]:
                got.append(sc.output_wait(t).copy())
            info = sc.resident_stats()
            assert info["failed"] == 0 and info["updates"] == updates, info
            # only the very first read (it allocates the ring) parks the kernel
            assert info["launches"] <= 3, info
        else:
            for k in range(updates):
                sc.apply_block(blocks[k])
                sc.mix(1024, post_process=True)
                got.append(sc.dry()[4:6].copy())
        outs[mode] = got
        sc.close()
    assert len(outs["resident"]) == len(outs["launched"]) == updates
    for k in range(updates):
        assert np.array_equal(_bits(outs["resident"][k]), _bits(outs["launched"][k])), k
    assert max(float(np.abs(a).max()) for a in outs["launched"]) > 1e-3


@pytest.mark.parametrize("form", list(FORMS))
def test_other_entry_points_park_the_resident_kernel(synth_mhr, form):
    """anything but apply / mix / read_output_async / output_wait tells the kernel to leave, and the next update starts a new one:
    parameters set the launched way between resident updates, a second context used in between, a parameter block applied
    without an update behind it"""
    import oalgpu
    from oalgpu import synth
    import bench
    mhr = synth.synth_mhr_bytes()
    updates = 12
    nv = 4096 if FORMS[form] == 0 else 512         # (the voice-per-wavefront kernel has a resident launch in its 16-wavefront form: a scene that fills the machine)
    outs = {}
    for mode in ("resident", "launched"):
        api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=(oalgpu.CTX_RESIDENT if mode == "resident" else 0) | FORMS[form])
        api._mhr = mhr
        sc, script = bench.build_scene(oalgpu, synth, api, 3, nv, 0, mhr, 0)
        other, _ = bench.build_scene(oalgpu, synth, api, 3, 64, 0, mhr, 0)
        allv = list(range(nv))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(updates)]
        got = []
        for k in range(updates):
            if k % 4 == 1:
                sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k + 1))   # the launched way
            else:
                sc.apply_block(blocks[k])
            if k == 6:
                sc.sync()                                # a block applied with no update behind it
                sc.apply_block(blocks[k])
            sc.mix(1024, post_process=True)
            if k % 3 == 2:
                other.mix(1024, post_process=True)       # another context of the device
                other.sync()
            if k == 8:
                # an object of the device created and destroyed while the launch runs: the destroy synchronises the device, so it
                # tells the resident kernel to leave first -- it does not sit out the kernel's 2 s watchdog
                import ctypes as C, time
                G = oalgpu.lib
                G.oalgpu_converter_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]
                G.oalgpu_converter_destroy.argtypes = [C.c_void_p]; G.oalgpu_converter_destroy.restype = None
                gh = C.c_void_p()
                assert G.oalgpu_converter_create(0, 2, 6, 1, 44100, 48000, 1, C.byref(gh)) == 0, G.oalgpu_last_error()
                t0 = time.perf_counter()
                G.oalgpu_converter_destroy(gh)
                assert time.perf_counter() - t0 < 1.0, "the destroy waited for the resident kernel's watchdog"
            if k % 5 == 4:
                got.append(sc.dry().copy())
        got.append(sc.dry().copy())
        got.append(sc.hrtf_accum().copy())
        if mode == "resident":
            info = sc.resident_stats()
            # (launches that keep being parked early send the context to the launched path for a while)
            assert info["failed"] == 0 and 3 <= info["updates"] <= updates and info["parks"] >= 3, info
        outs[mode] = got
        sc.close(); other.close()
    for a, b in zip(outs["resident"], outs["launched"]):
        assert np.array_equal(_bits(a), _bits(b))
