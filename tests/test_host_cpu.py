"""CPU-side checks of the product (no GPU needed): the C-ABI library loads and exports every
symbol include/oalgpu.h declares, its host-side tables / resampler preparation / biquad design
match the oracle bit-for-bit, and compute entry points fail loudly without a device."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import golden_cases
import oracle_lib as ol
import oalgpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _exported(path):
    import subprocess
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    out = subprocess.run([nm, "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted(line.split()[-1] for line in out.splitlines() if line.strip())


def _declared(path):
    hdr = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    return set(re.findall(r"\b(oalgpu_[a-z0-9_]+)\s*\(", hdr))


def test_library_exports_exactly_the_declared_symbols():
    """liboalgpu.so's dynamic symbol table IS include/oalgpu.h: every declared entry point exported, nothing else (no measurement
    aid, no C++ symbol, no kernel stub)"""
    import glob
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert [os.path.basename(h) for h in headers] == ["oalgpu.h"]                 # (the measurement header lives in tools/measure/)
    declared = _declared(headers[0])
    assert len(declared) >= 110 and not [n for n in declared if "_debug_" in n]
    exported = _exported(oalgpu.LIB_PATH)
    assert not [n for n in exported if not n.startswith("oalgpu_")], [n for n in exported if not n.startswith("oalgpu_")][:10]
    assert not [n for n in exported if "_debug_" in n]
    assert sorted(declared) == exported, (sorted(declared - set(exported)), sorted(set(exported) - declared))
    assert not [n for n in declared if not hasattr(oalgpu.lib, n)]


def test_measurement_build_is_the_product_plus_the_declared_aids():
    """`make measure`: liboalgpu_measure.so = the same exports + the oalgpu_*debug_* readers tools/measure/oalgpu_measure.h declares;
    liboalmeasure.so exports oalmeasure_* and needs nothing of liboalgpu.so beyond its public symbols"""
    measure = os.path.join(oalgpu.PKG_DIR, "liboalgpu_measure.so")
    aids = os.path.join(ROOT, "tools", "measure", "liboalmeasure.so")
    if not (os.path.exists(measure) and os.path.exists(aids)):
        pytest.skip("make measure was not run")
    mh = _declared(os.path.join(ROOT, "tools", "measure", "oalgpu_measure.h"))
    extra = sorted(set(_exported(measure)) - set(_exported(oalgpu.LIB_PATH)))
    assert extra == sorted(mh) and len(extra) == 5 and all("_debug_" in n for n in extra), extra
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", aids], check=True, capture_output=True, text=True).stdout
    wanted = {line.split()[-1] for line in out.splitlines() if "oalgpu_" in line}
    assert wanted and wanted <= set(_exported(oalgpu.LIB_PATH)), wanted
    assert [n for n in _exported(aids) if n.startswith("oalmeasure_")] == ["oalmeasure_event_floor_ms", "oalmeasure_pipelined_run", "oalmeasure_submit_cost"]


def test_tables_match_reference_golden():
    manifest = json.load(open(os.path.join(HERE, "golden", "golden.json")))["items"]
    api = oalgpu.Api()
    for which in (12, 24, 48):
        t = api.bsinc_table(which)
        assert golden_cases.digest(t["tab"]) == manifest[f"bsinc{which}.tab"]["sha256"]
        hdr = np.array(t["m"] + t["filterOffset"], np.int64)
        assert golden_cases.digest(hdr) == manifest[f"bsinc{which}.hdr"]["sha256"]
        sc = np.array([t["scaleBase"], t["scaleRange"]], np.float32)
        assert golden_cases.digest(sc) == manifest[f"bsinc{which}.scale"]["sha256"]
    assert golden_cases.digest(api.cubic_table(0)) == manifest["cubic.spline"]["sha256"]
    assert golden_cases.digest(api.cubic_table(1)) == manifest["cubic.gaussian"]["sha256"]


@pytest.mark.skipif(not ol.available("port"), reason="oracle port not built")
def test_prepare_resampler_matches_oracle():
    port = ol.load("port")
    api = oalgpu.Api()
    rng = np.random.default_rng(1)
    incs = [1, 65535, 65536, 65537, 60211, 655360] + [int(x) for x in rng.integers(1, 655360, 300)]
    for rs in range(10):
        for inc in incs:
            a, b = port.prepare_resampler(rs, inc), api.prepare_resampler(rs, inc)
            assert (a.kind, a.table, a.m, a.l, a.filter_offset) == (b.kind, b.table, b.m, b.l, b.filter_offset)
            assert np.float32(a.sf).tobytes() == np.float32(b.sf).tobytes()


@pytest.mark.skipif(not ol.available("port"), reason="oracle port not built")
def test_biquad_design_and_state_machine_match_oracle():
    port = ol.load("port")
    rng = np.random.default_rng(2)
    a, b = ol.Biquad(), oalgpu.Biquad()
    port.L.oal_biquad_reset(C.byref(a))
    oalgpu.lib.oalgpu_biquad_reset(C.byref(b))
    assert a.as_tuple() == b.as_tuple()
    for k in range(400):
        typ = int(rng.integers(0, 2))
        f0 = float(np.float32(rng.uniform(0.001, 0.6)))
        g = float(np.float32(10 ** rng.uniform(-4, 1)))
        if k % 7 == 0 and k > 0:            # repeat the previous design: the "not different" branch
            f0, g = last
        last = (f0, g)
        port.L.oal_biquad_set_params_from_slope(C.byref(a), typ, f0, g, 1.0)
        oalgpu.lib.oalgpu_biquad_set_params_from_slope(C.byref(b), typ, f0, g, 1.0)
        assert np.array(a.as_tuple()[:12], np.float32).tobytes() == np.array(b.as_tuple()[:12], np.float32).tobytes()
        assert a.counter == b.counter
        if k % 5 == 0:                      # pretend the filter ran: counter reaches 0 / stays
            a.counter = b.counter = int(rng.integers(-1, 3)) * 100


@pytest.mark.skipif(not ol.available("port"), reason="oracle port not built")
def test_splitter_init_matches_oracle():
    port = ol.load("port")
    for f0 in (400 / 48000, 0.25, 0.49, 0.6, 1e-4):
        a, b = ol.Splitter(), oalgpu.Splitter()
        port.L.oal_splitter_init(C.byref(a), f0)
        oalgpu.lib.oalgpu_splitter_init(C.byref(b), f0)
        assert np.float32(a.coeff).tobytes() == np.float32(b.coeff).tobytes()


def test_no_cpu_fallback_without_device():
    if oalgpu.device_count() > 0:
        pytest.skip("a GPU is present")
    api = oalgpu.Api()
    with pytest.raises(oalgpu.OalgpuError, match="no HIP device"):
        api.resample(oalgpu.RS_LINEAR, 60211, np.zeros(2048, np.float32), 0, 64)
    with pytest.raises(oalgpu.OalgpuError, match="no HIP device"):
        oalgpu.Scene(api, num_dry=3)


def test_context_flags_of_the_python_driver_match_the_header():
    """The ctypes driver repeats the OALGPU_CTX_* bits of include/oalgpu.h; a renumbered flag must not go unnoticed."""
    import re
    import oalgpu
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "oalgpu.h")).read()
    defs = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+OALGPU_CTX_(\w+)\s+(\d+)u", hdr)}
    assert len(defs) >= 6 and len(set(defs.values())) == len(defs)           # distinct bits
    for name, bit in defs.items():
        assert bit and bit & (bit - 1) == 0, (name, bit)                      # single bits
        assert getattr(oalgpu, "CTX_" + name) == bit, name
