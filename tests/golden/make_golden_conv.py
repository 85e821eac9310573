#!/usr/bin/env python3
"""Generates tests/golden/golden_conv.npz from the COMPILED REFERENCE's ConvolutionState
(oracle/_ref/liboalref.so, alc/effects/convolution.cpp compiled in place): the target lines of
every tests/conv_cases.py case.  Run in the dev container:  python tests/golden/make_golden_conv.py
Floating-point fixtures (the FFT order of pffft is not restated): consumers compare with the
tolerance stated in tests/test_conv.py."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import conv_cases            # noqa: E402
import oracle_lib as ol      # noqa: E402


def main():
    L = ol.load("ref")
    assert L.kind == "reference"
    L.L.oal_set_simd(1)
    front = L.direction_coeffs([0.0, 0.0, -1.0])
    out = {"front_coeffs": front}
    for case in conv_cases.CASES:
        y = conv_cases.run_case(L.make_convolution, front, case, is_product=False)
        out[case[0]] = y[[0, 3]]                      # lines 1,2 are exactly the untouched 0.125
        assert np.all(y[[1, 2]] == 0.125)
    # the 65536-tap case: keep the last update only (line 0), 4 KB
    y = conv_cases.run_case(L.make_convolution, front, conv_cases.BIG_CASE, is_product=False)
    out[conv_cases.BIG_CASE[0]] = y[0, -1024:]
    np.savez_compressed(os.path.join(HERE, "golden_conv.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
