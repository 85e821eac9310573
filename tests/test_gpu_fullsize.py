"""Parity at BASELINE.json's full sizes (4096 voices per context), through properties that do not
need the oracle to mix 4096 voices:

  * shard additivity / linearity -- the buses of the full scene equal the sum of the buses of
    two disjoint shards of it (what the multi-GPU split relies on);
  * an oracle anchor -- full scene minus the shard that holds all but the first 64 voices equals
    the oracle's mix of those 64 voices (the oracle finishes 64 voices in well under a second);
  * integer state of all 4096 voices after several updates against the closed form of
    Voice::mix's position arithmetic (core/voice.cpp:1126-1154), bit-exact;
  * determinism -- the same scene twice gives bit-identical buses (no atomics, fixed summation
    trees).

Tolerance for sums over voices: |a - b| <= 2e-5 * max|full bus| + 1e-7 (tests/test_gpu_parity.py)."""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

V = 4096
UPDATES = 4
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(synth_mhr):
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    api = oalgpu.Api(oalgpu.MATH_FAST)
    mhr = synth.synth_mhr_bytes()
    api._mhr = mhr
    return oalgpu, synth, bench, api, mhr


def run_gpu(env, config, nvoices, voice_base, updates=UPDATES, todo=1024, keep_scene=False):
    """Buses after every update (the parameter script of bench.py: every 4th voice moves)."""
    oalgpu, synth, bench, api, mhr = env
    sc, script = bench.build_scene(oalgpu, synth, api, config, nvoices, voice_base, mhr, 0)
    allv = list(range(nvoices))
    sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
    moving = [v for v in allv if script.is_moving(v)]
    hrtf = config in (3, 5)
    out = []
    for k in range(updates):
        if k > 0 and moving:
            sc.set_params_batch(moving, bench.param_array(oalgpu, script, moving, k))
        sc.mix(todo, post_process=False)
        parts = [sc.dry().ravel()]
        if hrtf:
            parts.append(sc.hrtf_accum().ravel())
        for slot in range({4: 4, 5: 1}.get(config, 0)):
            parts.append(sc.wet(slot).ravel())
        out.append(np.concatenate(parts).astype(np.float64))
    if keep_scene:
        return out, sc, script
    sc.close()
    return out


def close_to(a, b, scale, what):
    err = np.abs(a - b).max()
    assert err <= 2e-5 * scale + 1e-7, f"{what}: max err {err:.3e}, bound {2e-5 * scale + 1e-7:.3e}"


@pytest.mark.parametrize("config", [3, 2])
def test_full_scene_equals_sum_of_shards(env, config):
    full = run_gpu(env, config, V, 0)
    lo = run_gpu(env, config, V // 2, 0)
    hi = run_gpu(env, config, V // 2, V // 2)
    for k in range(UPDATES):
        scale = np.abs(full[k]).max()
        assert scale > 0.05, "the scene must actually sound"
        close_to(full[k], lo[k] + hi[k], scale, f"config {config}, update {k}")


@pytest.mark.parametrize("config", [3, 2])
def test_three_and_four_voices_per_wavefront(env, config):
    """Above 4096 voices a wavefront of the voice kernel takes three or four voices (and the second
    half of the grid takes them in reverse order, switching its issue priority half-way): 6000 and
    8192 voices against the sum of shards that run at one or two voices per wavefront."""
    for total, cut in ((6000, 2048), (8192, 4096)):
        full = run_gpu(env, config, total, 0, updates=3)
        parts = [run_gpu(env, config, min(cut, total - lo), lo, updates=3) for lo in range(0, total, cut)]
        for k in range(3):
            scale = np.abs(full[k]).max()
            assert scale > 0.05
            close_to(full[k], sum(p[k] for p in parts), scale, f"config {config}, {total} voices, update {k}")


def test_full_scene_anchored_on_the_oracle(env, synth_mhr):
    """full(4096) - rest(4032) == oracle(first 64 voices)."""
    oalgpu, synth, bench, api, mhr = env
    which = "ref" if ol.available("ref") else "port"
    L = ol.load(which)
    L.L.oal_set_simd(1)
    n0 = 64
    full = run_gpu(env, 3, V, 0)
    rest = run_gpu(env, 3, V - n0, n0)
    mhr_path = synth.write_synth_mhr(os.path.join(os.path.dirname(synth_mhr), "fullsize.mhr"))
    L.hrtf_load(mhr_path)
    osc = ol.Scene(L, num_dry=4, num_real=2, hrtf=True)
    bufs = synth.scene_buffers(3, n0)
    handles = [osc.add_buffer(b, ol.FMT_FLOAT) for b in bufs]
    script = synth.SceneScript(3, n0)
    for v in range(n0):
        osc.add_voice(handles[script.buffer_of(v, len(handles))], True, position=script.start_position(v))
        osc.set_params(v, script.fill(ol.VoiceParams(), v, 0))
    for k in range(UPDATES):
        if k > 0:
            for v in range(n0):
                if script.is_moving(v):
                    osc.set_params(v, script.fill(ol.VoiceParams(), v, k))
        osc.mix(1024, post_process=False)
        want = np.concatenate([osc.dry().ravel(), osc.hrtf_accum().ravel()]).astype(np.float64)
        close_to(full[k] - rest[k], want, np.abs(full[k]).max(), f"update {k}")
        assert np.abs(want).max() > 1e-3
    osc.close()


def test_integer_state_of_every_voice(env):
    """Position / fraction / play state of all 4096 voices after UPDATES updates of 1000 samples:
    looping static sources, so pos = (pos0 + total) wrapped into [loopStart, loopEnd) by
    ((pos - start) % (end - start)) + start after every update (voice.cpp:1126-1154)."""
    oalgpu, synth, bench, api, mhr = env
    todo = 1000
    _, sc, script = run_gpu(env, 3, V, 0, todo=todo, keep_scene=True)
    step, frames = synth.STEP_44K1, synth.BUFFER_FRAMES
    for v in range(V):
        pos, frac = script.start_position(v), 0
        for _ in range(UPDATES):
            frac += step * todo
            pos += frac >> 16
            frac &= 0xFFFF
            if pos >= frames:
                pos = pos % frames
        st = sc.voice_state(v)
        assert (st.play_state, st.position, st.position_frac, st.has_buffer, st.fading) == \
            (oalgpu.VOICE_PLAYING, pos, frac, 1, 1), v
    sc.close()


@pytest.mark.parametrize("config", [3, 4])
def test_deterministic(env, config):
    nv = V if config == 3 else 2048
    a = run_gpu(env, config, nv, 0, updates=3)
    b = run_gpu(env, config, nv, 0, updates=3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
