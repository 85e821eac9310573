import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "openal-soft_amd"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def synth_mhr(tmp_path_factory):
    """Synthetic HRTF data set with the geometry of the reference's Default HRTF.mhr."""
    from oalgpu import synth
    p = tmp_path_factory.mktemp("hrtf") / "synth_hrtf.mhr"
    synth.write_synth_mhr(str(p))
    return str(p)


@pytest.fixture(scope="session")
def mhr_paths(synth_mhr):
    """All data sets to test with: the synthetic one, plus the reference's own Default HRTF
    when /root/reference is mounted (dev container only; it never travels to the GPU box)."""
    paths = [synth_mhr]
    real = "/root/reference/hrtf/Default HRTF.mhr"
    if os.path.exists(real):
        paths.append(real)
    return paths
