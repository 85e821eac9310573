"""Why the full-size HRTF parity tests (tests/test_gpu_baseline_configs.py) widen the multi-voice
tolerance with the voice count: the REFERENCE's own rounding noise.

MixHrtf adds every tap's product straight into the shared accumulator (`Accum[i+j][e] += coef * in`,
core/mixer/hrtfbase.h:38 through ApplyCoeffs, mixer_c.cpp:139-148), voice after voice: one output frame
of a scene with N voices is a serial fp32 sum of K = N * IrSize terms into a running value of order
max|bus|.  Each add rounds to the running value's ulp (variance ulp^2 / 12), so the reference carries a
random-walk error of sigma <= sqrt(K / 12) * 2^-23 * max|bus| -- 1.8e-5 * max for 4096 voices x 64 taps,
already at the 2e-5 * max every smaller scene is held to.  (The product sums per voice in registers,
then per workgroup, then over workgroups: its own noise is ~sqrt(IrSize) smaller.)  The test below
measures that noise on synthetic terms of the bench scene's statistics and checks the model the GPU
tests use:  tol = max(2e-5, 2.5 * sqrt(K / 12) * 2^-23) * max|ref| + 1e-7."""
import numpy as np


def reference_noise_bound(num_voices, terms_per_voice):
    """sigma bound of a serial fp32 accumulation of num_voices * terms_per_voice terms, relative to max|sum|."""
    return float(np.sqrt(num_voices * terms_per_voice / 12.0) * 2.0 ** -23)


def multi_voice_tolerance(num_voices, terms_per_voice, scale):
    return max(2e-5, 2.5 * reference_noise_bound(num_voices, terms_per_voice)) * scale + 1e-7


def test_serial_fp32_accumulation_noise_matches_the_model():
    rng = np.random.default_rng(7)
    voices, taps, frames = 4096, 64, 96
    # per-voice gain 10^(U(-60,-20)/20) (synth.SceneScript), source samples U(-1,1), taps decaying noise
    gains = 10.0 ** (rng.uniform(-60.0, -20.0, voices) / 20.0)
    worst = 0.0
    rel = []
    for _ in range(frames):
        terms = (gains[:, None] * rng.uniform(-1, 1, (voices, taps)) *
                 (rng.standard_normal((voices, taps)) * 0.25 * np.exp(-np.arange(taps) / 12.0))).ravel()
        terms = terms.astype(np.float32)
        serial = np.cumsum(terms, dtype=np.float32)[-1]          # the reference's order: one running fp32 sum
        exact = np.sum(terms.astype(np.float64))
        rel.append(abs(float(serial) - exact))
        worst = max(worst, abs(exact))
    rel = np.array(rel) / worst
    sigma_bound = reference_noise_bound(voices, taps)
    # the measured noise is a real fraction of the bound (the running sum is below its maximum most of the
    # time, so it stays under it), i.e. the bound is the right order of magnitude -- not a loose excuse
    assert 0.02 * sigma_bound < rel.std() < sigma_bound, (rel.std(), sigma_bound)
    assert rel.max() < 2.5 * sigma_bound
    # small scenes keep the plain 2e-5: 64 voices x 64 taps -> 2.5 sigma = 5.5e-6
    assert multi_voice_tolerance(64, 64, 1.0) == 2e-5 * 1.0 + 1e-7
    assert multi_voice_tolerance(4096, 1, 1.0) == 2e-5 * 1.0 + 1e-7          # dry-line mixes: one term per voice
    assert 4.0e-5 < multi_voice_tolerance(4096, 64, 1.0) < 5.0e-5
