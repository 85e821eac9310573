#!/usr/bin/env python3
"""Generates tests/golden/golden_reverb.npz from the COMPILED REFERENCE's ReverbState
(oracle/_ref/liboalref.so; alc/effects/reverb.cpp compiled in place through oracle/ref_reverb.cpp)
for every schedule of tests/reverb_cases.py:
    params_<case>  uint8 [updates, sizeof(oal_reverb_params)]  the block after each update()
    crc_<case>     uint32 [steps]   zlib.crc32 of the target lines after each process()
    out_<case>     float32 [steps, 4, 1024]   the lines themselves, for FULL_CASES only
Run in the dev container:  python tests/golden/make_golden_reverb.py
The reverb path is restated operation for operation, so consumers compare bit for bit."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol                                             # noqa: E402
from reverb_cases import CASES, FULL_CASES, SEED, wet_input, BUFFER_LINE, out_init   # noqa: E402


def main():
    L = ol.load("ref")
    assert L.kind == "reference"
    out = {}
    for name, schedule in CASES:
        r = L.make_reverb(4)
        x = wet_input(SEED[name], len(schedule))
        blocks, crcs, lines = [], [], []
        for u, st in enumerate(schedule):
            if st["props"] is not None:
                r.update(ol.ReverbProps.make(**st["props"]), st["slot_gain"])
                blocks.append(np.frombuffer(r.get_params().as_bytes(), np.uint8).copy())
            o = out_init(4)
            r.process_n(x[u], o, st["n"])
            crcs.append(zlib.crc32(o.tobytes()))
            lines.append(o)
        out["params_" + name] = np.stack(blocks)
        out["crc_" + name] = np.asarray(crcs, np.uint32)
        if name in FULL_CASES:
            out["out_" + name] = np.stack(lines)
        r.close()
    np.savez_compressed(os.path.join(HERE, "golden_reverb.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print(os.path.getsize(os.path.join(HERE, "golden_reverb.npz")), "bytes")


if __name__ == "__main__":
    main()
