#!/usr/bin/env python3
"""Generates tests/golden/golden.json + golden_small.npz from the COMPILED REFERENCE
(oracle/_ref/liboalref.so = kcat/openal-soft's own sources built in place by oracle/Makefile).

Run in the dev container (needs /root/reference):   python tests/golden/make_golden.py
The fixtures pin the C restatement (oracle/oalport.c) on machines without the reference.
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))

import oracle_lib as ol          # noqa: E402
import golden_cases              # noqa: E402
from oalgpu import synth         # noqa: E402


def main():
    L = ol.load("ref")
    assert L.kind == "reference"
    with tempfile.TemporaryDirectory() as td:
        mhr = synth.write_synth_mhr(os.path.join(td, "synth.mhr"))
        res = golden_cases.collect(L, mhr)
    manifest = {"generator": "tests/golden/make_golden.py", "source": "oracle/_ref/liboalref.so "
                "(kcat/openal-soft @ 2026-08-21 compiled in place, SSE variants unless .simd0)",
                "items": {}}
    small = {}
    for k, v in res.items():
        v = np.ascontiguousarray(v)
        manifest["items"][k] = {"sha256": golden_cases.digest(v), "dtype": str(v.dtype),
                                "shape": list(v.shape)}
        if v.size <= 4096:
            small[k] = v
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "golden_small.npz"), **small)
    print(f"{len(res)} items, {len(small)} stored in full")


if __name__ == "__main__":
    main()
