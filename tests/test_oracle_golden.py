"""The C restatement (oracle/oalport.c) against the committed golden vectors generated from
the compiled reference (tests/golden/make_golden.py).  Bit-exact: sha256 of every array.
CPU only; does not need /root/reference or oracle/_ref."""
import json
import os

import numpy as np
import pytest

import golden_cases
import oracle_lib as ol

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def port_results(synth_mhr):
    if not ol.available("port"):
        pytest.skip("oracle/liboalport.so not built (run __graft_entry__.build())")
    return golden_cases.collect(ol.load("port"), synth_mhr)


def test_port_matches_reference_golden(port_results):
    with open(os.path.join(HERE, "golden", "golden.json")) as f:
        manifest = json.load(f)
    small = np.load(os.path.join(HERE, "golden", "golden_small.npz"))
    assert set(manifest["items"]) == set(port_results), "golden manifest out of date"
    bad = []
    for k, meta in manifest["items"].items():
        v = np.ascontiguousarray(port_results[k])
        assert list(v.shape) == meta["shape"] and str(v.dtype) == meta["dtype"], k
        if golden_cases.digest(v) != meta["sha256"]:
            detail = ""
            if k in small.files:
                g = small[k]
                idx = np.flatnonzero(g.view(np.uint8).ravel() != v.view(np.uint8).ravel())
                detail = f" first differing byte {idx[0]}"
            bad.append(k + detail)
    assert not bad, f"{len(bad)} golden mismatches: {bad[:8]}"
