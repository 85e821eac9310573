"""Deterministic convolution-reverb cases shared by the oracle pin test, the golden generator and
the GPU parity test.  ``run_case(make_conv, direction_coeffs, case)`` drives any object with the
oracle_lib.Convolution interface (update/process) or the product binding (set_target_gains/process)."""
import numpy as np

CASES = [
    # (name, ir_len, [update sizes], slot gains per update)
    ("short_ir_50", 50, [1024, 1024], [0.5, 0.5]),
    ("one_segment_128", 128, [1024, 512, 1024], [0.7, 0.7, 0.2]),
    ("ragged_700", 700, [1000, 37, 128, 1024, 91, 1024], [0.5, 0.5, 0.9, 0.9, 0.1, 0.1]),
    ("long_5000", 5000, [1024, 1024, 1024, 600, 1024, 1024], [0.4] * 6),
]
BIG_CASE = ("ir_65536", 65536, [1024] * 6, [0.3] * 6)   # BASELINE configs[4] response length
NLINES = 4


def make_ir(name, ir_len):
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    t = np.arange(ir_len)
    return (rng.standard_normal(ir_len) * np.exp(-t / max(ir_len / 6.0, 8.0)) * 0.2).astype(np.float32)


def make_input(name, total):
    rng = np.random.default_rng(1000 + sum(map(ord, name)))
    return rng.uniform(-1.0, 1.0, total).astype(np.float32)


def run_case(make_conv, front_coeffs, case, is_product):
    """Returns the concatenated target lines of every update, shape (NLINES, sum(sizes))."""
    name, ir_len, sizes, gains = case
    ir = make_ir(name, ir_len)
    x = make_input(name, sum(sizes))
    conv = make_conv(NLINES, ir)
    outs = []
    pos = 0
    for n, g in zip(sizes, gains):
        if is_product:
            conv.set_target_gains(front_coeffs[:NLINES] * g)      # ComputePanGains, identity AmbiMap
        else:
            conv.update(g)
        lines = np.full((NLINES, 1024), 0.125, np.float32)          # the effect ADDS into its target
        conv.process(x[pos:pos + n], lines)
        outs.append(lines[:, :n].copy())
        assert np.all(lines[:, n:] == 0.125), "samples past samplesToDo must stay untouched"
        pos += n
    conv.close()
    return np.concatenate(outs, axis=1)


def float_reference(case):
    """Straight double-precision convolution with the gain ramp (what all implementations
    approximate); used only as a sanity anchor, the parity target is the compiled reference."""
    name, ir_len, sizes, gains = case
    ir = make_ir(name, ir_len).astype(np.float64)
    x = make_input(name, sum(sizes)).astype(np.float64)
    return np.convolve(x, ir)[:x.size]
