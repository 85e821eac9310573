"""DIRECT parity at BASELINE.json's full sizes: `bench.build_scene` (the scene bench.py times) and the
SAME scene in the compiled reference (Voice::mix for every voice, the reference's own ReverbState /
ConvolutionState on the slots, MixDirectHrtf), several updates through `oalgpu_mix_update` with the
post-process -- the pipelined two-stream path the bench line measures -- and after every update

  * every bus compared sample by sample: dry + real lines (after the effects and the post-process),
    every slot's wet bus, the carried HRTF accumulator;
  * every voice's integer state (play state, position, fraction, buffer, fade flag) compared exactly.

No shard arithmetic: the reference mixes all 4096 / 8192 voices (~60 ms per update on one core).

  config 2: 4096 voices, bsinc24 -> 5 dry lines              alc/alu.cpp:2177-2273, core/voice.cpp:934-984
  config 3: 4096 HRTF voices, on the synthetic data set AND on the reference's own Default HRTF.mhr
  config 4: 8192 voices, v % 5 sends into 4 EAX reverb slots  alc/effects/reverb.cpp:1813-1883
  config 5: 4096 HRTF voices + a 65 536-tap convolution slot  alc/effects/convolution.cpp:623-716

Tolerance (sums over thousands of voices in a different order than the reference's serial loop, FAST
math): |gpu - ref| <= 2e-5 * max|ref block| + 1e-7 per compared block, the figure of every other
multi-voice test -- widened for the HRTF accumulator and the lines fed from it to
max(2e-5, 2.5 * sqrt(voices * IrSize / 12) * 2^-23) * max|ref| (4.4e-5 at 4096 voices x 64 taps): that is
the random-walk rounding noise the REFERENCE itself carries there, because MixHrtf adds every tap
product straight into the shared fp32 accumulator (tests/test_tolerance_model.py derives and measures
it).  Integer state: exact."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REAL_MHR = os.path.join(ROOT, "tests", "golden", "default_hrtf.mhr")

UPDATES = 4
pytestmark = pytest.mark.gpu


def _oracle():
    if not ol.available("ref"):
        pytest.skip("needs the compiled reference (oracle/_ref)")
    L = ol.load("ref")
    L.L.oal_set_simd(1)
    return L


def set_reference_decoder(L, sc, synth):
    """the post-process's decoder as the REFERENCE builds it (DirectHrtfState::build on its own store, InitHrtfPanning's
    first-order layout, alc/panning.cpp:1100-1134) -- what bench.build_scene asks the product to build"""
    info = L.hrtf_raw()["info"]
    cc, hf, irsize = L.direct_hrtf_build(info.ir_size, False, synth.AMBI_POINTS_1O, synth.AMBI_MATRIX_1O, 4, 400.0,
                                         synth.AMBI_ORDER_HF_GAIN_1O)
    sc.set_direct_hrtf(cc, hf, 400.0 / info.sample_rate, irsize)


def build_reference_scene(L, synth, config, nvoices, mhr_path, conv_ir=None, num_real=None, sample_fmt="f32"):
    """bench.build_scene, restated on the reference: same buffers, voices, direct-HRTF decoder."""
    hrtf = config in (3, 5)
    nsends = {4: 4, 5: 1}.get(config, 0)
    if hrtf:
        L.hrtf_load(mhr_path)
    if num_real is None:
        num_real = 2 if hrtf else 0
    sc = ol.Scene(L, sample_rate=48000, num_dry=4 if hrtf else 5, num_real=num_real,
                  num_sends=nsends, num_slots=nsends, wet_channels=4, hrtf=hrtf)
    effects = []
    if config == 4:
        for _ in range(4):
            rev = L.make_reverb(5)
            rev.update(ol.ReverbProps.make(), 1.0)
            effects.append(rev)
    if config == 5:
        conv = L.make_convolution(4, conv_ir)
        conv.update(1.0)
        effects.append(conv)
    if hrtf:
        set_reference_decoder(L, sc, synth)
    bufs = synth.scene_buffers(config, nvoices, sample_fmt)
    handles = [sc.add_buffer(b, ol.FMT_SHORT if sample_fmt == "i16" else ol.FMT_FLOAT) for b in bufs]
    script = synth.SceneScript(config, nvoices, 0)
    for v in range(nvoices):
        sc.add_voice(handles[script.buffer_of(v, len(handles))], True, position=script.start_position(v))
    return sc, script, effects


def close_to(got, want, what, terms=(1, 1)):
    """terms = (voices, serial fp32 adds per voice and output sample in the reference)"""
    from test_tolerance_model import multi_voice_tolerance
    scale = float(np.abs(want).max())
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    bound = multi_voice_tolerance(terms[0], terms[1], scale)
    print(f"{what}: max err {err:.3e} = {err / max(scale, 1e-30):.2e} of max|ref| {scale:.3e} (bound {bound:.3e})")
    assert err <= bound, f"{what}: max err {err:.3e}, bound {bound:.3e} (max|ref| {scale:.3e})"
    return scale


def run_config(config, nvoices, mhr_path, todo=(1024, 1024, 1024, 1024), check_at=None, sample_fmt="f32", ctx_flags=0, via_blocks=False,
               expect_kernel=None):
    """check_at: the updates (0-based) after which buses and voice states are compared (None: every one).  Between
    checkpoints nothing of the product is read: its two-stream pipeline runs on unsynchronised, as in the bench.
    via_blocks: the product takes every update's parameters as a parameter block resident in HBM and runs the updates up to
    the next checkpoint in ONE oalgpu_mix_update_run call -- bench.py's timed loop, the path on which the
    OALGPU_CTX_APPLY_IN_VOICE_KERNEL / _FUSED_REDUCE / _RESIDENT contexts differ from the plain one."""
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    L = _oracle()
    api = oalgpu.Api(oalgpu.MATH_FAST, ctx_flags=ctx_flags)
    with open(mhr_path, "rb") as f:
        mhr = f.read()
    api._mhr = mhr
    hrtf = config in (3, 5)
    nslots = {4: 4, 5: 1}.get(config, 0)

    gsc, gscript = bench.build_scene(oalgpu, synth, api, config, nvoices, 0, mhr, 0, sample_fmt=sample_fmt)
    if expect_kernel is not None:
        assert expect_kernel in gsc.voice_kernel_name(), gsc.voice_kernel_name()
    conv_ir = None
    if config == 5:
        # the reference pans a mono response to the front (ConvolutionProps orientation, convolution.cpp:511-620)
        lcg = synth.Lcg(0x5EED0005)
        conv_ir = np.array([lcg.uniform(-1.0, 1.0) for _ in range(65536)], np.float32)
        conv_ir *= np.exp(-np.arange(65536) / 12000.0).astype(np.float32) * 0.05
        gsc.effects[0].set_target_gains(L.direction_coeffs([0.0, 0.0, -1.0])[:4])
    osc, oscript, oeffects = build_reference_scene(L, synth, config, nvoices, mhr_path, conv_ir, sample_fmt=sample_fmt)

    allv = list(range(nvoices))
    moving = [v for v in allv if gscript.is_moving(v)]
    sounded = False
    blocks, first_pending = [], 0
    if via_blocks:
        assert len(set(todo)) == 1, "oalgpu_mix_update_run: one length for the run"
        blocks = [gsc.param_block(allv if k == 0 else moving, bench.param_array(oalgpu, gscript, allv if k == 0 else moving, k))
                  for k in range(len(todo))]
        if ctx_flags & oalgpu.CTX_RESIDENT:
            gsc.resident_set_short_run(0)           # (the checkpoints keep the launches short: no falling back to launches per update)
    for k in range(len(todo)):
        n = todo[k]
        voices = allv if k == 0 else moving
        for v in voices:
            osc.set_params(v, oscript.fill(ol.VoiceParams(), v, k))
        # ---- the product: one pipelined update, effects and post-process included
        if not via_blocks:
            gsc.set_params_batch(voices, bench.param_array(oalgpu, gscript, voices, k))
            gsc.mix(n, post_process=True)
        elif check_at is None or k in check_at or k == len(todo) - 1:
            gsc.mix_run(blocks[first_pending:k + 1], n, post_process=True)
            first_pending = k + 1
        # ---- the reference: voice loop, then the slots' effects into the dry lines, then the post-process
        osc.mix(n, post_process=False)
        wets = [osc.wet(s) for s in range(nslots)]
        dry = osc.dry_view()
        if config == 4:
            for s in range(4):
                oeffects[s].process_n(np.ascontiguousarray(wets[s][:4]), dry[:5], n)
        if config == 5:
            oeffects[0].process(wets[0][0, :n], dry[:4])
        if hrtf:
            osc.post_process(n)
        if check_at is not None and k not in check_at:
            continue
        want_dry = osc.dry()
        got_dry = gsc.dry()
        hrtf_terms = (nvoices, 64) if hrtf else (nvoices, 1)      # RealOut L/R come out of the HRTF accumulator
        scale = close_to(got_dry[:, :n], want_dry[:, :n], f"config {config} update {k}: dry/real lines", hrtf_terms)
        sounded = sounded or scale > 0.01
        for s in range(nslots):
            close_to(gsc.wet(s)[:, :n], wets[s][:, :n], f"config {config} update {k}: wet bus of slot {s}", (nvoices, 1))
        if hrtf:
            close_to(gsc.hrtf_accum(), osc.hrtf_accum(), f"config {config} update {k}: carried HRTF accumulator", (nvoices, 64))
        # integer state of every voice, exact
        for v in range(nvoices):
            g, o = gsc.voice_state(v), osc.voice_state(v)
            assert (g.play_state, g.position, g.position_frac, g.has_buffer, g.fading) == \
                (o.play_state, o.position, o.position_frac, o.has_buffer, o.fading), (config, k, v)
    assert sounded, "the scene must actually sound"
    # float state of a sample of voices after the last update
    for v in range(0, nvoices, 97):
        g, o = gsc.voice_state(v), osc.voice_state(v)
        prev_g, prev_o = np.array(g.prev_samples[:]), np.array(o.prev_samples[:])
        assert np.abs(prev_g - prev_o).max() <= 2e-5 * max(1.0, np.abs(prev_o).max()), v
        if hrtf:
            hg, ho = np.array(g.hrtf_history[:]), np.array(o.hrtf_history[:])
            assert np.abs(hg - ho).max() <= 2e-5 * max(1.0, np.abs(ho).max()) + 1e-7, v
            assert tuple(g.hrtf_old_delay) == tuple(o.hrtf_old_delay), v
    if via_blocks and ctx_flags & oalgpu.CTX_RESIDENT:
        info = gsc.resident_stats()
        assert info["enabled"] == 1 and info["failed"] == 0 and info["updates"] == len(todo), info
    for e in oeffects:
        e.close()
    gsc.close()
    osc.close()


def test_config2_4096_voices_five_dry_lines(synth_mhr):
    run_config(2, 4096, synth_mhr)


@pytest.mark.parametrize("data_set", ["synthetic", "Default HRTF.mhr"])
def test_config3_4096_hrtf_voices(synth_mhr, data_set):
    if data_set == "synthetic":
        run_config(3, 4096, synth_mhr)
    else:
        assert os.path.exists(REAL_MHR), "tests/golden/default_hrtf.mhr is a committed fixture"
        run_config(3, 4096, REAL_MHR, todo=(1024, 1024, 1000, 1024))


def test_config4_8192_voices_four_reverb_slots(synth_mhr):
    """the default form: the rows in LDS (csrc/voice_rows.hip) -- a wavefront per voice resamples ONCE into a 4 KB slot of LDS, the
    round's filtered signals are jobs dealt to all eight wavefronts, each wavefront adds its own 128-frame slice of every row to
    the 21 lines it keeps in registers -- against the reference at full size."""
    run_config(4, 8192, synth_mhr, expect_kernel="VoiceRowsKernel")


def test_config4_stream_rows(synth_mhr):
    """OALGPU_CTX_STREAM_ROWS: a 4 KB row per mixed signal in HBM, mixed by the voice kernel's tail (csrc/voice_wave.hip; the default
    of rounds 3-5, and still the form of contexts with near-field control and sends) -- against the reference at full size."""
    import oalgpu
    run_config(4, 8192, synth_mhr, ctx_flags=oalgpu.CTX_STREAM_ROWS, expect_kernel="VoiceWaveKernel")


def test_config4_a_wavefront_per_slice(synth_mhr):
    """OALGPU_CTX_SLICE_LINES: the 21 mix lines in the registers of four wavefronts that own a 256-frame slice each
    (csrc/voice_slice.hip) -- nothing of a voice's rows leaves the CU -- against the reference at full size."""
    import oalgpu
    run_config(4, 8192, synth_mhr, ctx_flags=oalgpu.CTX_SLICE_LINES, expect_kernel="VoiceSliceKernel")


def test_config4_rows_in_lds_by_name(synth_mhr):
    """OALGPU_CTX_ROW_SLICES names the default form (include/oalgpu.h): the same kernel, the same results."""
    import oalgpu
    run_config(4, 8192, synth_mhr, ctx_flags=oalgpu.CTX_ROW_SLICES, expect_kernel="VoiceRowsKernel")


PATHS = {"stream rows": (8, "VoiceWaveKernel"), "a wavefront per slice": (128, "VoiceSliceKernel"), "rows in LDS": (0, "VoiceRowsKernel")}


@pytest.mark.parametrize("path", list(PATHS))
def test_config4_odd_update_lengths_and_a_ragged_last_workgroup(synth_mhr, path):
    """Updates that end inside a 256-frame slice, inside the gain ramp (40 < 64 frames) or inside the first slice, and a
    voice count that leaves the last workgroup partly empty."""
    run_config(4, 2039, synth_mhr, todo=(1024, 1000, 300, 40, 257, 1024), ctx_flags=PATHS[path][0], expect_kernel=PATHS[path][1])


def test_config5_4096_hrtf_voices_and_a_65536_tap_convolution(synth_mhr):
    """the default form since round 6: the voice-per-wavefront HRTF kernel with the send's signal leaving as one stream row per voice,
    mixed onto the slot's wet lines by a small kernel behind it (csrc/voice_wave16.hip: W16Sends, StreamRowsMixKernel)"""
    run_config(5, 4096, synth_mhr, expect_kernel="VoiceWave16Kernel<16, sends>")


def test_config5_wet_lines_in_the_registers_of_the_two_voice_kernel(synth_mhr):
    """OALGPU_CTX_WAVE_PAIRS: the form of rounds 4-5 (two voices per wavefront, the slot's four wet lines accumulated in registers)"""
    import oalgpu
    run_config(5, 4096, synth_mhr, ctx_flags=oalgpu.CTX_WAVE_PAIRS, expect_kernel="DeviceLayout, 4>")


# ---- SURVEY.md 8(d) "Parity check in the same run": after updates 1, 2, 8 and 50, f32 and i16 sources ----------------
# Fifty updates carry every voice's filter, fade, history and loop state (the 48 000-frame buffers wrap after 51 updates
# at 941 frames each; start positions (v * 7919) % 48000 put a wrap into the run for most voices) far past the four
# updates of the tests above: drift of any of it against core/voice.cpp:1126-1154, :1224-1232 shows up here.
SCHEDULE = (0, 1, 7, 49)


@pytest.mark.parametrize("sample_fmt", ["f32", "i16"])
def test_config2_parity_after_updates_1_2_8_50(synth_mhr, sample_fmt):
    run_config(2, 4096, synth_mhr, todo=(1024,) * 50, check_at=SCHEDULE, sample_fmt=sample_fmt)


@pytest.mark.parametrize("sample_fmt", ["f32", "i16"])
def test_config3_parity_after_updates_1_2_8_50(sample_fmt):
    assert os.path.exists(REAL_MHR), "tests/golden/default_hrtf.mhr is a committed fixture"
    run_config(3, 4096, REAL_MHR, todo=(1024,) * 50, check_at=SCHEDULE, sample_fmt=sample_fmt)


@pytest.mark.parametrize("sample_fmt", ["f32", "i16"])
def test_config4_parity_after_updates_1_2_8_50(synth_mhr, sample_fmt):
    run_config(4, 8192, synth_mhr, todo=(1024,) * 50, check_at=SCHEDULE, sample_fmt=sample_fmt, expect_kernel="VoiceRowsKernel")


def test_config4_stream_rows_after_updates_1_2_8_50(synth_mhr):
    import oalgpu
    run_config(4, 8192, synth_mhr, todo=(1024,) * 50, check_at=SCHEDULE, sample_fmt="i16", ctx_flags=oalgpu.CTX_STREAM_ROWS,
               expect_kernel="VoiceWaveKernel")


def test_config4_a_wavefront_per_slice_after_updates_1_2_8_50(synth_mhr):
    import oalgpu
    run_config(4, 8192, synth_mhr, todo=(1024,) * 50, check_at=SCHEDULE, sample_fmt="i16", ctx_flags=oalgpu.CTX_SLICE_LINES,
               expect_kernel="VoiceSliceKernel")


@pytest.mark.parametrize("sample_fmt", ["f32", "i16"])
def test_config5_parity_after_updates_1_2_8_50_on_the_default_data_set(sample_fmt):
    assert os.path.exists(REAL_MHR), "tests/golden/default_hrtf.mhr is a committed fixture"
    run_config(5, 4096, REAL_MHR, todo=(1024,) * 50, check_at=SCHEDULE, sample_fmt=sample_fmt)


@pytest.mark.parametrize("mode", ["plain", "apply_in_voice_kernel", "fused_reduce", "resident"])
def test_config3_block_driven_contexts_against_the_reference(mode):
    """bench.py's timed loop -- parameter blocks resident in HBM, runs of updates in one oalgpu_mix_update_run -- held against
    the compiled reference directly at full size, fifty updates, for the plain context and each opt-in variant of the update
    pipeline (the voice kernel installing the next block itself; the reduction fused into the post-process launch; ONE launch
    of the voice kernel that stays on the device across the run's updates)."""
    import oalgpu
    flags = {"plain": 0, "apply_in_voice_kernel": oalgpu.CTX_APPLY_IN_VOICE_KERNEL, "fused_reduce": oalgpu.CTX_FUSED_REDUCE,
             "resident": oalgpu.CTX_RESIDENT}[mode]
    run_config(3, 4096, REAL_MHR, todo=(1024,) * 50, check_at=SCHEDULE, ctx_flags=flags, via_blocks=True)


@pytest.mark.parametrize("config", [2, 5])
def test_stream_row_path_of_the_few_line_contexts(synth_mhr, config):
    """Configs 2 and 5 mix their lines in the wavefronts' registers by default; OALGPU_CTX_STREAM_ROWS puts them on the path of
    the contexts with more than 6 (4) lines -- stream rows mixed in the voice kernel's tail -- which must agree with the
    reference just the same."""
    import oalgpu
    run_config(config, 4096, synth_mhr, ctx_flags=oalgpu.CTX_STREAM_ROWS, expect_kernel="VoiceWaveKernel")
