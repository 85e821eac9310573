"""ctypes binding of tools/measure/liboalmeasure.so (oalgpu_measure.h): measurement loops written against the PUBLIC C-ABI of
liboalgpu.so.  Not part of the product; bench.py and tools/ import it."""
import ctypes as C
import os

import numpy as np

import oalgpu

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liboalmeasure.so")
_f32p = C.POINTER(C.c_float)
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise oalgpu.OalgpuError(f"{_PATH} is missing: `make measure` in openal-soft_amd/ (or __graft_entry__.build())")
        _lib = C.CDLL(_PATH)          # (its DT_NEEDED liboalgpu.so resolves to the library `oalgpu` already loaded: same soname)
        _lib.oalmeasure_pipelined_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                  _f32p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.oalmeasure_submit_cost.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                                _f32p, C.c_size_t, C.POINTER(C.c_double)]
        _lib.oalmeasure_event_floor_ms.argtypes = [C.c_void_p, C.c_uint32, _f32p]
    return _lib


def _flat(move_sets):
    return np.ascontiguousarray(np.concatenate([np.ascontiguousarray(m, oalgpu.MOVE_DTYPE) for m in move_sets]))


def _out(ctx):
    n = ctx.desc.num_real_channels or ctx.desc.num_dry_channels
    return np.empty((n, oalgpu.BUFFER_LINE), np.float32)


def pipelined_run(ctx, move_sets, updates, samples=oalgpu.BUFFER_LINE, post_process=True):
    """the section-3c loop of INTEGRATION.md in C++; move_sets: list of equally long MOVE_DTYPE arrays.
    Returns (wall seconds, seconds of the calling thread outside oalgpu_output_wait)."""
    flat, out = _flat(move_sets), _out(ctx)
    wall, busy = C.c_double(), C.c_double()
    oalgpu.check(lib().oalmeasure_pipelined_run(ctx.h, flat.ctypes.data_as(C.c_void_p), len(move_sets[0]), len(move_sets), updates,
                                                samples, 1 if post_process else 0, out.ctypes.data_as(_f32p), out.size,
                                                C.byref(wall), C.byref(busy)), "oalmeasure_pipelined_run")
    return wall.value, busy.value


def submit_cost(ctx, move_sets, updates, samples=oalgpu.BUFFER_LINE, post_process=True):
    """seconds per update the three submitting calls cost the calling thread when nothing is queued"""
    flat, out = _flat(move_sets), _out(ctx)
    spent = C.c_double()
    oalgpu.check(lib().oalmeasure_submit_cost(ctx.h, flat.ctypes.data_as(C.c_void_p), len(move_sets[0]), len(move_sets), updates,
                                              samples, 1 if post_process else 0, out.ctypes.data_as(_f32p), out.size,
                                              C.byref(spent)), "oalmeasure_submit_cost")
    return spent.value


def event_floor_ms(ctx, reps=200):
    """what the dispatch-bound HIP events report for an EMPTY kernel on the context's stream (median of `reps`)"""
    a = C.c_float()
    oalgpu.check(lib().oalmeasure_event_floor_ms(ctx.h, reps, C.byref(a)), "oalmeasure_event_floor_ms")
    return a.value


def use_measurement_build():
    """switch the `oalgpu` module to openal-soft_amd/liboalgpu_measure.so (the product's sources built -DOALGPU_MEASUREMENT:
    plus oalgpu_debug_* / oalgpu_reverb_debug_*); call before creating any context.  Returns the library."""
    path = os.path.join(oalgpu.PKG_DIR, "liboalgpu_measure.so")
    oalgpu.lib = oalgpu._load(path)
    L = oalgpu.lib
    L.oalgpu_debug_set_ablate.argtypes = [C.c_void_p, C.c_uint32]
    L.oalgpu_debug_phase_times.argtypes = [C.c_void_p, C.c_void_p]
    L.oalgpu_debug_wave_times.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
    L.oalgpu_reverb_debug_enable_phase_times.argtypes = [C.c_void_p]
    L.oalgpu_reverb_debug_phase_times.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    return L
