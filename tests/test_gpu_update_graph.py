"""A run of updates captured into ONE hipGraph (oalgpu_update_graph_*) against the same updates issued one by
one through oalgpu_param_block_apply + oalgpu_mix_update, GPU against GPU: same kernels, same arguments, same
order on each stream -- the buses, the carried HRTF accumulator and the voice states must agree bit for bit.
Covers an HRTF context (BASELINE configs[2] geometry, the bench scene), a dry-line context that ends in the
ambisonic decode, graph launches mixed with plain updates, and the refusals."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu
G = 8                      # updates per graph


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("config,voices", [(3, 4096), (2, 1024)])
def test_graph_run_equals_update_by_update(config, voices):
    import oalgpu
    from oalgpu import synth
    import bench
    assert oalgpu.device_count() > 0, "GPU tests need a HIP device"
    api = oalgpu.Api(oalgpu.MATH_FAST)
    mhr = synth.synth_mhr_bytes()
    api._mhr = mhr

    def build():
        sc, script = bench.build_scene(oalgpu, synth, api, config, voices, 0, mhr, 0)
        allv = list(range(voices))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        blocks = [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(2 * G)]
        return sc, blocks

    graphed, gblocks = build()
    plain, pblocks = build()
    hrtf = config == 3
    # graph A: updates 0..7 with their blocks; graph B: 8..15 where every other update changes nothing
    second = [b if k % 2 == 0 else None for k, b in enumerate(gblocks[G:])]
    ga = graphed.update_graph(gblocks[:G], 1024, True)
    gb = graphed.update_graph(second, 1024, True)

    def plain_run(blocks):
        for b in blocks:
            if b is not None:
                plain.apply_block(b)
            plain.mix(1024, post_process=True)

    def compare(what):
        a, b = graphed.dry().copy(), plain.dry().copy()
        assert np.abs(b).max() > 1e-3, what
        assert np.array_equal(_bits(a), _bits(b)), f"{what}: bus block differs"
        if hrtf:
            assert np.array_equal(_bits(graphed.hrtf_accum()), _bits(plain.hrtf_accum())), f"{what}: HRTF accumulator differs"
        for v in range(0, voices, 53):
            x, y = graphed.voice_state(v), plain.voice_state(v)
            assert (x.play_state, x.position, x.position_frac) == (y.play_state, y.position, y.position_frac), (what, v)
            assert np.array_equal(_bits(x.prev_samples), _bits(y.prev_samples)), (what, v)

    ga.launch(); plain_run(pblocks[:G]); compare("graph A")
    gb.launch(); plain_run([b if k % 2 == 0 else None for k, b in enumerate(pblocks[G:])]); compare("graph B")
    # graph launches between plain updates, and the same graph again (its blocks re-applied)
    graphed.apply_block(gblocks[3]); graphed.mix(1024, post_process=True)
    plain.apply_block(pblocks[3]); plain.mix(1024, post_process=True)
    ga.launch(); ga.launch(); plain_run(pblocks[:G]); plain_run(pblocks[:G])
    graphed.mix(1024, post_process=True); plain.mix(1024, post_process=True)
    compare("graphs mixed with plain updates")
    ga.close(); gb.close()
    graphed.close(); plain.close()


def test_graph_refusals():
    import oalgpu
    from oalgpu import synth
    api = oalgpu.Api(oalgpu.MATH_FAST)
    api._mhr = synth.synth_mhr_bytes()
    sc = api.make_scene(num_dry=4, num_real=2, hrtf=True, num_sends=1, num_slots=1, max_voices=8)
    with pytest.raises(RuntimeError):
        sc.update_graph([None] * 3)                 # odd count
    rev = oalgpu.Reverb(4)
    rev.update(oalgpu.ReverbProps.make(), 1.0)
    sc.set_slot_reverb(0, rev)
    with pytest.raises(RuntimeError):
        sc.update_graph([None, None])               # an effect's arguments advance on the host
    sc.set_slot_reverb(0, None)
    g = sc.update_graph([None, None])
    g.launch(); sc.sync()
    g.close()
    exact = oalgpu.Api(oalgpu.MATH_EXACT).make_scene(num_dry=3, num_real=0, hrtf=False, max_voices=8)
    with pytest.raises(RuntimeError):
        exact.update_graph([None, None])            # the workgroup-per-voice-group kernel runs on one stream
    exact.close(); rev.close(); sc.close()


def test_update_run_equals_update_by_update():
    """oalgpu_mix_update_run: `count` updates submitted by one call are the updates issued one by one"""
    import oalgpu
    from oalgpu import synth
    import bench
    api = oalgpu.Api(oalgpu.MATH_FAST)
    mhr = synth.synth_mhr_bytes()
    api._mhr = mhr
    voices = 512

    def build():
        sc, script = bench.build_scene(oalgpu, synth, api, 3, voices, 0, mhr, 0)
        allv = list(range(voices))
        moving = [v for v in allv if script.is_moving(v)]
        sc.set_params_batch(allv, bench.param_array(oalgpu, script, allv, 0))
        return sc, [sc.param_block(moving, bench.param_array(oalgpu, script, moving, k + 1)) for k in range(6)]

    a, ab = build()
    b, bb = build()
    a.mix_run([ab[0], None, ab[2], ab[3], None, ab[5]], 1024, True)
    for k, blk in enumerate(bb):
        if k not in (1, 4):
            b.apply_block(blk)
        b.mix(1024, post_process=True)
    assert np.abs(b.dry()).max() > 1e-3
    assert np.array_equal(_bits(a.dry()), _bits(b.dry()))
    assert np.array_equal(_bits(a.hrtf_accum()), _bits(b.hrtf_accum()))
    a.close(); b.close()
