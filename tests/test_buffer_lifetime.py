"""Buffer handles have a lifetime (oalgpu_buffer_release): the reference frees and reuses buffer storage and queue items
(core/buffer_storage.h:47-77, core/voice.h:84-98), so a host that deletes a buffer gives its handle up, and the library frees the
HBM copy -- and hands the handle out again -- once nothing refers to it: no voice slot initialised on it, no queue link, no
channel view.  max_buffers bounds the LIVE handles, not the registrations of a context's life."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def _scene(oalgpu, max_buffers=8):
    api = oalgpu.Api(oalgpu.MATH_FAST)
    sc = api.make_scene(num_dry=3, num_real=0, num_sends=0, num_slots=0, wet_channels=4, hrtf=False, max_voices=8, max_buffers=max_buffers)
    return sc


def _params(ol):
    return ol.make_voice_params(65536, ol.RS_POINT, dry_gains=[1.0, 0.0, 0.0])


def test_released_buffer_is_freed_when_its_last_voice_lets_go_and_the_handle_is_reused():
    import oalgpu
    import oracle_lib as ol
    rng = np.random.default_rng(7)
    a = rng.uniform(-1, 1, 6000).astype(np.float32)
    b = rng.uniform(-1, 1, 6000).astype(np.float32)
    sc = _scene(oalgpu)
    ha = sc.add_buffer(a, oalgpu.FMT_FLOAT)
    v = sc.add_voice(ha, looping=False, position=0)
    sc.set_params(v, _params(ol))
    sc.mix(1000)
    assert np.array_equal(sc.dry()[0, :1000], a[:1000])
    sc.release_buffer(ha)                                   # alDeleteBuffers while the source still plays it
    assert sc.buffer_info(ha) == (True, True, 1)            # held by the voice slot
    with pytest.raises(oalgpu.OalgpuError):
        sc.init_voice(1, ha, False)                         # ... but no longer to be named
    sc.mix(1000)
    assert np.array_equal(sc.dry()[0, :1000], a[1000:2000])  # the voice plays on
    sc.set_state(v, oalgpu.VOICE_STOPPED)                   # the slot lets go
    assert sc.buffer_info(ha)[0] is False
    hb = sc.add_buffer(b, oalgpu.FMT_FLOAT)                 # storage reallocated "at the same address": the handle comes back
    assert hb == ha
    sc.init_voice(v, hb, False)
    sc.set_params(v, _params(ol))
    sc.mix(1000)
    assert np.array_equal(sc.dry()[0, :1000], b[:1000])     # ... and plays the NEW samples
    sc.close()


def test_more_registrations_than_max_buffers_over_a_contexts_life():
    import oalgpu
    import oracle_lib as ol
    sc = _scene(oalgpu, max_buffers=4)
    v = sc.add_voice(sc.add_buffer(np.zeros(64, np.float32), oalgpu.FMT_FLOAT), looping=False)
    sc.set_params(v, _params(ol))
    for k in range(40):                                     # ten times the table
        x = np.full(2000, 0.001 * (k + 1), np.float32)
        h = sc.add_buffer(x, oalgpu.FMT_FLOAT)
        assert 0 <= h < 4
        sc.init_voice(v, h, False)                          # (the slot lets go of the buffer before)
        sc.set_params(v, _params(ol))
        sc.mix(500)
        assert np.array_equal(sc.dry()[0, :500], x[:500]), k
        sc.release_buffer(h)                                # freed when the slot is initialised on the next one
    with pytest.raises(oalgpu.OalgpuError):                 # the table bounds the live handles
        for _ in range(5):
            sc.add_buffer(np.zeros(16, np.float32), oalgpu.FMT_FLOAT)
    sc.close()


def test_queue_links_and_views_hold_their_buffers_and_unqueue_lets_them_go():
    import oalgpu
    import oracle_lib as ol
    rng = np.random.default_rng(11)
    parts = [rng.uniform(-1, 1, n).astype(np.float32) for n in (700, 900, 1500, 800)]
    whole = np.concatenate(parts)
    sc = _scene(oalgpu)
    hs = [sc.add_buffer(p, oalgpu.FMT_FLOAT) for p in parts]
    for i in range(3):
        sc.link_buffers(hs[i], hs[i + 1])
    v = sc.add_queue_voice(hs[0], looping=False)
    sc.set_params(v, _params(ol))
    got = []
    sc.mix(1000); got.append(sc.dry()[0, :1000].copy())    # through buffer 0 into buffer 1
    sc.release_buffer(hs[0])                                # the application is done with buffer 0 ...
    assert sc.buffer_info(hs[0]) == (True, True, 1)         # ... the voice slot still holds the queue's head
    with pytest.raises(oalgpu.OalgpuError):
        sc.unqueue(v, 1)                                    # nothing read back yet: the library cannot know it is processed
    cur, done = sc.queue_state(v)
    assert (cur, done) == (hs[1], 1)
    sc.unqueue(v, 1)                                        # alSourceUnqueueBuffers
    assert sc.buffer_info(hs[0])[0] is False                # freed while the source plays on
    assert sc.buffer_info(hs[1])[2] == 1                    # buffer 1: the voice's new head (buffer 0's link to it went with buffer 0)
    sc.mix(1000); got.append(sc.dry()[0, :1000].copy())
    sc.mix(1000); got.append(sc.dry()[0, :1000].copy())
    assert np.array_equal(np.concatenate(got), whole[:3000])
    # a released buffer in the middle of the queue stays while the link in front of it lives
    sc.release_buffer(hs[2])
    assert sc.buffer_info(hs[2]) == (True, True, 1)
    with pytest.raises(oalgpu.OalgpuError):
        sc.unqueue(v, 5)                                    # more than it has played through
    sc.close()


def test_channel_view_holds_the_interleaved_buffer():
    import oalgpu
    sc = _scene(oalgpu)
    st = np.arange(4000, dtype=np.float32).reshape(2000, 2) / 4000.0
    h = sc.add_buffer(st, oalgpu.FMT_FLOAT, frame_step=2)
    lib = oalgpu.lib
    view = oalgpu.check(lib.oalgpu_buffer_channel_view(sc.h, h, 1), "view")
    sc.release_buffer(h)
    assert sc.buffer_info(h) == (True, True, 1)             # the view holds the storage
    sc.release_buffer(view)
    assert sc.buffer_info(h)[0] is False and sc.buffer_info(view)[0] is False
    sc.close()
