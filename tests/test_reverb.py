"""EAX reverb (alc/effects/reverb.cpp): the C restatement of ReverbState::process against the
compiled reference, driven by the parameter block the reference's update() computed."""
import zlib

import numpy as np
import pytest

import oracle_lib as ol
from reverb_cases import CASES, wet_input, BUFFER_LINE

needs_ref = pytest.mark.skipif(not ol.available("ref"), reason="oracle/_ref not built here")


def run_schedule(ref, other, schedule, seed, nlines=4):
    """Drives `ref` (compiled reference) and `other` (anything with set_params/process_n) in
    lock-step; yields (step index, ref out, other out)."""
    x = wet_input(seed, len(schedule))
    for u, st in enumerate(schedule):
        if st["props"] is not None:
            ref.update(ol.ReverbProps.make(**st["props"]), st["slot_gain"])
            other.set_params(ref.get_params())
        o_ref = np.zeros((nlines, BUFFER_LINE), np.float32)
        o_oth = np.zeros((nlines, BUFFER_LINE), np.float32)
        o_ref[:, :7] = 0.125                     # process() ADDS into the target lines
        o_oth[:, :7] = 0.125
        ref.process_n(x[u], o_ref, st["n"])
        other.process_n(x[u], o_oth, st["n"])
        yield u, o_ref, o_oth


@needs_ref
def test_line_lengths_match():
    for rate in (44100, 48000, 96000):
        a = ol.load("ref").make_reverb(4, rate)
        b = ol.load("port").make_reverb(4, rate)
        assert a.line_lengths() == b.line_lengths()
        a.close(); b.close()


@needs_ref
@pytest.mark.parametrize("name,schedule", CASES, ids=[c[0] for c in CASES])
def test_port_matches_reference(name, schedule):
    ref = ol.load("ref").make_reverb(4)
    port = ol.load("port").make_reverb(4)
    energy = 0.0
    for u, o_ref, o_port in run_schedule(ref, port, schedule, seed=zlib.crc32(name.encode()) % 1000):
        assert np.array_equal(o_ref.view(np.uint32), o_port.view(np.uint32)), (name, u,
            np.abs(o_ref - o_port).max())
        energy += float(np.abs(o_ref[:, 7:]).sum())
    assert energy > 1.0
    ref.close(); port.close()
