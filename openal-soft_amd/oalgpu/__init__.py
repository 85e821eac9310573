"""ctypes binding of the product C-ABI (include/oalgpu.h, openal-soft_amd/liboalgpu.so).

This is driver/test plumbing only -- the product is the shared library.  Importing the module
loads the library and fails loudly when it has not been built; every compute call fails with
OALGPU_ERR_NO_DEVICE when there is no GPU (there is no CPU fallback).
"""
import ctypes as C
import os

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(PKG_DIR, "liboalgpu.so")

BUFFER_LINE = 1024
MAX_PAD = 48
HRTF_HIST = 64
HRIR_LEN = 128
MAX_SENDS = 6
(OUT_I8, OUT_U8, OUT_I16, OUT_U16, OUT_I32, OUT_U32, OUT_F32) = range(7)      # oalgpu_output_type
MAX_OUT = 32
MAX_AMBI = 25
MATH_EXACT, MATH_FAST = 0, 1
(RS_POINT, RS_LINEAR, RS_SPLINE, RS_GAUSSIAN, RS_FAST_BSINC12, RS_BSINC12, RS_FAST_BSINC24,
 RS_BSINC24, RS_FAST_BSINC48, RS_BSINC48) = range(10)
FMT_UBYTE, FMT_SHORT, FMT_INT, FMT_FLOAT, FMT_DOUBLE, FMT_MULAW, FMT_ALAW = range(7)
FMT_DTYPES = {FMT_UBYTE: np.uint8, FMT_SHORT: np.int16, FMT_INT: np.int32, FMT_FLOAT: np.float32,
              FMT_DOUBLE: np.float64, FMT_MULAW: np.uint8, FMT_ALAW: np.uint8}
VOICE_STOPPED, VOICE_PLAYING, VOICE_STOPPING, VOICE_PENDING = range(4)

f32p = C.POINTER(C.c_float)
# oalgpu_voice_move
MOVE_DTYPE = np.dtype([("voice", np.uint32), ("hrtf_ev", np.float32), ("hrtf_az", np.float32), ("hrtf_dist", np.float32),
                       ("hrtf_spread", np.float32), ("hrtf_gain", np.float32)])
u32p = C.POINTER(C.c_uint32)


class OalgpuError(RuntimeError):
    pass


class BsincTable(C.Structure):
    _fields_ = [("scaleBase", C.c_float), ("scaleRange", C.c_float), ("m", C.c_uint32 * 16),
                ("filterOffset", C.c_uint32 * 16), ("tab", f32p), ("tablen", C.c_size_t)]


class InterpState(C.Structure):
    _fields_ = [("kind", C.c_int32), ("table", C.c_int32), ("sf", C.c_float), ("m", C.c_uint32),
                ("l", C.c_uint32), ("filter_offset", C.c_uint32)]


class Biquad(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("z1", "z2", "b0", "b1", "b2", "a1", "a2", "tb0", "tb1",
                                         "tb2", "ta1", "ta2")] + [("counter", C.c_int32)]

    def as_tuple(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


class Splitter(C.Structure):
    _fields_ = [("coeff", C.c_float), ("lp_z1", C.c_float), ("lp_z2", C.c_float),
                ("ap_z1", C.c_float)]


class HrtfInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("sample_rate", "ir_size", "num_fields", "num_elevs",
                                          "num_irs")]


class ContextDesc(C.Structure):
    _fields_ = [("device", C.c_int32), ("math_mode", C.c_int32), ("sample_rate", C.c_uint32),
                ("num_dry_channels", C.c_uint32), ("num_real_channels", C.c_uint32),
                ("num_aux_sends", C.c_uint32), ("num_slots", C.c_uint32),
                ("wet_channels", C.c_uint32), ("hrtf", C.c_int32), ("max_voices", C.c_uint32),
                ("max_buffers", C.c_uint32), ("voices_per_group", C.c_uint32), ("flags", C.c_uint32)]


# oalgpu_context_desc::flags
class VoiceEvent(C.Structure):
    _fields_ = [("voice", C.c_uint32), ("play_state", C.c_int32), ("has_buffer", C.c_int32), ("current_buffer", C.c_int32),
                ("buffers_done", C.c_uint32), ("position", C.c_int32), ("position_frac", C.c_uint32), ("fading", C.c_int32)]


CTX_FIR_VALU, CTX_PROFILE, CTX_SERIAL, CTX_STREAM_ROWS, CTX_APPLY_IN_VOICE_KERNEL, CTX_FUSED_REDUCE = 1, 2, 4, 8, 16, 32
CTX_RESIDENT = 64
CTX_SLICE_LINES = 128
CTX_WAVE_PAIRS = 256
CTX_ROW_SLICES = 512


class VoiceDesc(C.Structure):
    _fields_ = [("buffer", C.c_int32), ("looping", C.c_int32), ("position", C.c_int32),
                ("position_frac", C.c_uint32), ("frequency", C.c_uint32)]


class FilterParams(C.Structure):
    _fields_ = [("active", C.c_int32), ("gain_hf", C.c_float), ("hf_norm", C.c_float),
                ("gain_lf", C.c_float), ("lf_norm", C.c_float)]


class VoiceParams(C.Structure):
    _fields_ = [("step", C.c_uint32), ("resampler", C.c_int32), ("direct_filter", FilterParams),
                ("dry_gains", C.c_float * MAX_OUT),
                ("hrtf_ev", C.c_float), ("hrtf_az", C.c_float), ("hrtf_dist", C.c_float),
                ("hrtf_spread", C.c_float), ("hrtf_gain", C.c_float),
                ("send_slot", C.c_int32 * MAX_SENDS), ("send_filter", FilterParams * MAX_SENDS),
                ("send_gains", (C.c_float * MAX_AMBI) * MAX_SENDS)]


class VoiceState(C.Structure):
    _fields_ = [("play_state", C.c_int32), ("position", C.c_int32), ("position_frac", C.c_uint32),
                ("has_buffer", C.c_int32), ("fading", C.c_int32),
                ("prev_samples", C.c_float * MAX_PAD), ("dry_current", C.c_float * MAX_OUT),
                ("hrtf_old_gain", C.c_float), ("hrtf_old_delay", C.c_uint32 * 2),
                ("hrtf_history", C.c_float * HRTF_HIST),
                ("direct_lp", Biquad), ("direct_hp", Biquad),
                ("send_current", (C.c_float * MAX_AMBI) * MAX_SENDS),
                ("send_lp", Biquad * MAX_SENDS), ("send_hp", Biquad * MAX_SENDS)]


def _fp(a):
    return a.ctypes.data_as(f32p)


def _load(path=LIB_PATH):
    if not os.path.exists(path):
        raise OalgpuError(f"{path} is missing: build it with __graft_entry__.build() "
                          "(hipcc --offload-arch=gfx950); there is no fallback path")
    L = C.CDLL(path)
    L.oalgpu_version.restype = C.c_char_p
    L.oalgpu_last_error.restype = C.c_char_p
    L.oalgpu_bsinc_table_get.argtypes = [C.c_int, C.POINTER(BsincTable)]
    L.oalgpu_cubic_table_get.argtypes = [C.c_int, f32p]
    L.oalgpu_prepare_resampler.argtypes = [C.c_int, C.c_uint32, C.POINTER(InterpState)]
    L.oalgpu_biquad_reset.argtypes = [C.POINTER(Biquad)]
    L.oalgpu_biquad_reset.restype = None
    L.oalgpu_biquad_set_params_from_slope.argtypes = [C.POINTER(Biquad), C.c_int, C.c_float,
                                                      C.c_float, C.c_float]
    L.oalgpu_biquad_set_params_from_slope.restype = None
    L.oalgpu_splitter_init.argtypes = [C.POINTER(Splitter), C.c_float]
    L.oalgpu_splitter_init.restype = None
    L.oalgpu_resample.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, f32p, C.c_size_t,
                                  C.c_uint32, f32p, C.c_size_t]
    L.oalgpu_mix.argtypes = [C.c_int, f32p, C.c_size_t, f32p, C.c_size_t, f32p, f32p, C.c_size_t,
                             C.c_size_t]
    L.oalgpu_mix_hrtf.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_uint32, f32p, u32p, C.c_float,
                                  C.c_float, C.c_size_t]
    L.oalgpu_mix_hrtf_blend.argtypes = [C.c_int, C.c_int, f32p, f32p, C.c_uint32, f32p, u32p,
                                        C.c_float, f32p, u32p, C.c_float, C.c_size_t]
    L.oalgpu_mix_direct_hrtf.argtypes = [C.c_int, C.c_int, f32p, f32p, f32p, C.c_size_t, f32p,
                                         C.POINTER(Splitter), f32p, f32p, C.c_size_t, C.c_size_t]
    L.oalgpu_biquad_dual_process.argtypes = [C.c_int, C.POINTER(Biquad), C.POINTER(Biquad), f32p,
                                             f32p, C.c_size_t]
    L.oalgpu_context_create.argtypes = [C.POINTER(ContextDesc), C.POINTER(C.c_void_p)]
    L.oalgpu_context_destroy.argtypes = [C.c_void_p]
    L.oalgpu_context_destroy.restype = None
    L.oalgpu_hrtf_load_mhr.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.oalgpu_hrtf_info_get.argtypes = [C.c_void_p, C.POINTER(HrtfInfo)]
    L.oalgpu_hrtf_raw.argtypes = [C.c_void_p, f32p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint16),
                                  C.POINTER(C.c_uint16), f32p, C.POINTER(C.c_uint8)]
    L.oalgpu_hrtf_get_coeffs.argtypes = [C.c_void_p, f32p, C.c_size_t, f32p, u32p]
    L.oalgpu_set_direct_hrtf.argtypes = [C.c_void_p, f32p, f32p, C.c_float, C.c_uint32]
    L.oalgpu_buffer_register.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32,
                                         C.c_uint32, C.c_uint32]
    L.oalgpu_voice_init.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(VoiceDesc)]
    L.oalgpu_voice_set_params.argtypes = [C.c_void_p, u32p, C.c_void_p, C.c_size_t]
    L.oalgpu_voice_set_state.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.oalgpu_param_block_create.argtypes = [C.c_void_p, u32p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.oalgpu_param_block_apply.argtypes = [C.c_void_p, C.c_void_p]
    L.oalgpu_param_block_destroy.argtypes = [C.c_void_p]
    L.oalgpu_param_block_destroy.restype = None
    L.oalgpu_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    L.oalgpu_mix_update.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.oalgpu_mix_voices.argtypes = [C.c_void_p, C.c_uint32]
    L.oalgpu_post_process.argtypes = [C.c_void_p, C.c_uint32]
    L.oalgpu_set_carry_accum.argtypes = [C.c_void_p, C.c_int]
    L.oalgpu_sync.argtypes = [C.c_void_p]
    L.oalgpu_read_dry.argtypes = [C.c_void_p, f32p]
    L.oalgpu_read_wet.argtypes = [C.c_void_p, C.c_uint32, f32p]
    L.oalgpu_read_hrtf_accum.argtypes = [C.c_void_p, f32p]
    L.oalgpu_bus_device_ptr.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                        C.POINTER(C.c_void_p)]
    L.oalgpu_voice_readback.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(VoiceState)]
    L.oalgpu_set_timing.argtypes = [C.c_void_p, C.c_int]
    L.oalgpu_last_update_ms.argtypes = [C.c_void_p, f32p, f32p]
    return L


lib = _load()


def check(rc, what=""):
    if rc < 0:
        raise OalgpuError(f"{what} failed ({rc}): {lib.oalgpu_last_error().decode()}")
    return rc


def device_count():
    return lib.oalgpu_device_count()


class Api:
    """Per-call mirrors + table access, with the same method names as tests/oracle_lib.OracleLib
    so the parity tests can drive oracle and product through one code path."""
    kind = "oalgpu"

    def __init__(self, mode=MATH_EXACT, device=0, ctx_flags=0):
        self.mode = mode
        self.device = device
        self.ctx_flags = ctx_flags      # oalgpu_context_desc::flags of the scenes made from this Api
        self._mhr = None

    # ---- tables (host side) ----
    def bsinc_table(self, which):
        t = BsincTable()
        check(lib.oalgpu_bsinc_table_get(which, C.byref(t)), "bsinc_table_get")
        tab = np.ctypeslib.as_array(t.tab, shape=(t.tablen,)).copy()
        return dict(scaleBase=t.scaleBase, scaleRange=t.scaleRange, m=list(t.m),
                    filterOffset=list(t.filterOffset), tab=tab)

    def cubic_table(self, which):
        out = np.zeros((32, 8), np.float32)
        check(lib.oalgpu_cubic_table_get(which, _fp(out)))
        return out

    def prepare_resampler(self, resampler, increment):
        st = InterpState()
        check(lib.oalgpu_prepare_resampler(resampler, increment, C.byref(st)))
        return st

    # ---- per-call kernels (GPU) ----
    def resample(self, resampler, increment, src, frac, n):
        src = np.ascontiguousarray(src, np.float32)
        dst = np.zeros(n, np.float32)
        check(lib.oalgpu_resample(self.device, self.mode, resampler, increment, _fp(src), src.size,
                                  frac, _fp(dst), n), "oalgpu_resample")
        return dst

    def mix(self, inp, out, cur, tgt, counter, outpos):
        inp = np.ascontiguousarray(inp, np.float32)
        tgt = np.ascontiguousarray(tgt, np.float32)
        check(lib.oalgpu_mix(self.device, _fp(inp), inp.size, _fp(out), out.shape[0], _fp(cur),
                             _fp(tgt), counter, outpos), "oalgpu_mix")

    def mix_hrtf(self, inp, accum, irsize, coeffs, delay, gain, gainstep, n):
        inp = np.ascontiguousarray(inp, np.float32)
        coeffs = np.ascontiguousarray(coeffs, np.float32)
        check(lib.oalgpu_mix_hrtf(self.device, self.mode, _fp(inp), _fp(accum), irsize, _fp(coeffs),
                                  (C.c_uint32 * 2)(*delay), gain, gainstep, n), "oalgpu_mix_hrtf")

    def mix_hrtf_blend(self, inp, accum, irsize, oldc, oldd, oldgain, newc, newd, newstep, n):
        inp = np.ascontiguousarray(inp, np.float32)
        oldc = np.ascontiguousarray(oldc, np.float32)
        newc = np.ascontiguousarray(newc, np.float32)
        check(lib.oalgpu_mix_hrtf_blend(self.device, self.mode, _fp(inp), _fp(accum), irsize,
                                        _fp(oldc), (C.c_uint32 * 2)(*oldd), oldgain, _fp(newc),
                                        (C.c_uint32 * 2)(*newd), newstep, n), "oalgpu_mix_hrtf_blend")

    def mix_direct_hrtf(self, left, right, inp, accum, splitters, hfscales, chan_coeffs, irsize, n):
        nch = inp.shape[0]
        sp = (Splitter * nch)()
        for i, s in enumerate(splitters):
            sp[i] = Splitter(s.coeff, s.lp_z1, s.lp_z2, s.ap_z1)
        hf = np.ascontiguousarray(hfscales, np.float32)
        cc = np.ascontiguousarray(chan_coeffs, np.float32)
        check(lib.oalgpu_mix_direct_hrtf(self.device, self.mode, _fp(left), _fp(right), _fp(inp), nch,
                                         _fp(accum), sp, _fp(hf), _fp(cc), irsize, n),
              "oalgpu_mix_direct_hrtf")
        return list(sp)

    def biquad_dual_process(self, lp, hp, src):
        src = np.ascontiguousarray(src, np.float32)
        dst = np.zeros(src.size, np.float32)
        check(lib.oalgpu_biquad_dual_process(self.device, C.byref(lp), C.byref(hp), _fp(src),
                                             _fp(dst), src.size), "oalgpu_biquad_dual_process")
        return dst

    # ---- HRTF data set: remembered here, loaded into each context created afterwards ----
    def hrtf_load(self, path):
        with open(path, "rb") as f:
            self._mhr = f.read()
        return None

    def make_scene(self, **kw):
        return Scene(self, **kw)


class Scene:
    """Batched path: one device context, driven like tests/oracle_lib.Scene."""

    def __init__(self, api, sample_rate=48000, num_dry=3, num_real=0, num_sends=0, num_slots=0,
                 wet_channels=4, hrtf=False, max_voices=64, max_buffers=64, voices_per_group=0, flags=None):
        self.h = None
        self.api = api
        self.desc = ContextDesc(api.device, api.mode, sample_rate, num_dry, num_real, num_sends,
                                num_slots, wet_channels, 1 if hrtf else 0, max_voices, max_buffers,
                                voices_per_group, api.ctx_flags if flags is None else flags)
        h = C.c_void_p()
        check(lib.oalgpu_context_create(C.byref(self.desc), C.byref(h)), "oalgpu_context_create")
        self.h = h
        self.nvoices = 0
        if hrtf:
            if api._mhr is None:
                raise OalgpuError("HRTF scene without a loaded .mhr")
            check(lib.oalgpu_hrtf_load_mhr(self.h, api._mhr, len(api._mhr)), "oalgpu_hrtf_load_mhr")

    def close(self):
        if getattr(self, "h", None):
            lib.oalgpu_context_destroy(self.h)
            self.h = None

    def __del__(self):
        if lib is not None:             # module globals are gone when the interpreter shuts down
            self.close()

    def hrtf_info(self):
        info = HrtfInfo()
        check(lib.oalgpu_hrtf_info_get(self.h, C.byref(info)))
        return info

    def hrtf_raw(self):
        info = self.hrtf_info()
        fd = np.zeros(info.num_fields, np.float32)
        fe = np.zeros(info.num_fields, np.uint8)
        az = np.zeros(info.num_elevs, np.uint16)
        io = np.zeros(info.num_elevs, np.uint16)
        co = np.zeros((info.num_irs, HRIR_LEN, 2), np.float32)
        de = np.zeros((info.num_irs, 2), np.uint8)
        check(lib.oalgpu_hrtf_raw(self.h, _fp(fd), fe.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  az.ctypes.data_as(C.POINTER(C.c_uint16)),
                                  io.ctypes.data_as(C.POINTER(C.c_uint16)), _fp(co),
                                  de.ctypes.data_as(C.POINTER(C.c_uint8))))
        return dict(info=info, field_distance=fd, field_evcount=fe, elev_azcount=az,
                    elev_iroffset=io, coeffs=co, delays=de)

    def hrtf_get_coeffs(self, dirs):
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 4)
        n = dirs.shape[0]
        co = np.zeros((n, HRIR_LEN, 2), np.float32)
        de = np.zeros((n, 2), np.uint32)
        check(lib.oalgpu_hrtf_get_coeffs(self.h, _fp(dirs), n, _fp(co), de.ctypes.data_as(u32p)),
              "oalgpu_hrtf_get_coeffs")
        return co, de

    def add_buffer(self, data, fmt, frame_step=1, loop_start=0, loop_end=None):
        data = np.ascontiguousarray(data, FMT_DTYPES[fmt])
        n = data.size // frame_step
        if loop_end is None:
            loop_end = n
        return check(lib.oalgpu_buffer_register(self.h, data.ctypes.data_as(C.c_void_p), fmt,
                                                frame_step, n, loop_start, loop_end),
                     "oalgpu_buffer_register")

    def release_buffer(self, buffer):
        """oalgpu_buffer_release: the handle is given up; freed (and reusable) once no voice slot, queue link or view holds it"""
        lib.oalgpu_buffer_release.argtypes = [C.c_void_p, C.c_int]
        check(lib.oalgpu_buffer_release(self.h, buffer), "oalgpu_buffer_release")

    def buffer_info(self, buffer):
        """(live, release pending, references) of a buffer handle"""
        lib.oalgpu_buffer_info.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        live, pend, refs = C.c_int32(0), C.c_int32(0), C.c_uint32(0)
        check(lib.oalgpu_buffer_info(self.h, buffer, C.byref(live), C.byref(pend), C.byref(refs)), "oalgpu_buffer_info")
        return bool(live.value), bool(pend.value), refs.value

    def init_voice(self, voice, buffer, looping, position=0, frac=0, frequency=44100):
        """oalgpu_voice_init on a given voice slot (add_voice takes the next free one)"""
        d = VoiceDesc(buffer, 1 if looping else 0, position, frac, frequency)
        check(lib.oalgpu_voice_init(self.h, voice, C.byref(d)), "oalgpu_voice_init")

    def unqueue(self, voice, count):
        lib.oalgpu_voice_queue_unqueue.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        check(lib.oalgpu_voice_queue_unqueue(self.h, voice, count), "oalgpu_voice_queue_unqueue")

    def add_voice(self, buffer, looping, position=0, frac=0, frequency=44100):
        d = VoiceDesc(buffer, 1 if looping else 0, position, frac, frequency)
        v = self.nvoices
        check(lib.oalgpu_voice_init(self.h, v, C.byref(d)), "oalgpu_voice_init")
        self.nvoices += 1
        return v

    # IMA4 (adpcm_type 0) / MS ADPCM (1) data: decoded once on the GPU into 16-bit PCM
    def add_buffer_adpcm(self, data, adpcm_type, channels, samples_per_block, sample_len, loop_start=0, loop_end=None):
        data = np.ascontiguousarray(data, np.uint8)
        lib.oalgpu_buffer_register_adpcm.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_uint32] * 5
        return check(lib.oalgpu_buffer_register_adpcm(self.h, data.ctypes.data_as(C.c_void_p), adpcm_type, channels,
                                                      samples_per_block, sample_len, loop_start,
                                                      sample_len if loop_end is None else loop_end),
                     "oalgpu_buffer_register_adpcm")

    # streaming sources (same interface as tests/oracle_lib.Scene)
    def link_buffers(self, buffer, nxt):
        lib.oalgpu_buffer_queue_link.argtypes = [C.c_void_p, C.c_int, C.c_int]
        check(lib.oalgpu_buffer_queue_link(self.h, buffer, nxt), "oalgpu_buffer_queue_link")

    def add_queue_voice(self, first_buffer, looping, position=0, frac=0, frequency=44100):
        lib.oalgpu_voice_init_queue.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int32, C.c_uint32]
        v = self.nvoices
        check(lib.oalgpu_voice_init_queue(self.h, v, first_buffer, 1 if looping else 0, position, frac),
              "oalgpu_voice_init_queue")
        self.nvoices += 1
        return v

    # callback sources: `stream` (bytes-like / numpy array) is what the user function hands out, in the order asked
    def add_callback_voice(self, stream, fmt, frac=0, frequency=44100, voice=None):
        """voice: re-initialise that voice slot (a callback source that has ended starts over) instead of the next new one"""
        CB = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int32)
        raw = np.ascontiguousarray(stream).view(np.uint8).ravel().copy()
        st = {"pos": 0, "calls": 0}

        def read(_user, data, nbytes):
            st["calls"] += 1
            n = max(0, min(int(nbytes), raw.size - st["pos"]))
            if n:
                C.memmove(data, raw.ctypes.data + st["pos"], n)
            st["pos"] += n
            return n

        fn = CB(read)
        if not hasattr(self, "_callbacks"):
            self._callbacks = {}
        lib.oalgpu_voice_init_callback.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, CB, C.c_void_p]
        v = self.nvoices if voice is None else voice
        check(lib.oalgpu_voice_init_callback(self.h, v, fmt, frac, fn, None), "oalgpu_voice_init_callback")
        self._callbacks[v] = (fn, raw, st)          # keep the thunk and the stream alive
        if voice is None:
            self.nvoices += 1
        return v

    def callback_state(self, voice):
        """(mNumCallbackBlocks, mCallbackBlockOffset, CallbackStopped, calls of the user function)"""
        class CbState(C.Structure):
            _fields_ = [("position", C.c_int32), ("position_frac", C.c_uint32), ("num_blocks", C.c_uint32),
                        ("block_offset", C.c_uint32), ("stopped", C.c_int32), ("play_state", C.c_int32), ("has_buffer", C.c_int32)]
        out = CbState()
        lib.oalgpu_voice_callback_state.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(CbState)]
        check(lib.oalgpu_voice_callback_state(self.h, voice, C.byref(out)), "oalgpu_voice_callback_state")
        self._cb_mirror = out
        return out.num_blocks, out.block_offset, out.stopped, self._callbacks[voice][2]["calls"]

    def queue_state(self, voice):
        lib.oalgpu_voice_queue_state.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]
        cur, done = C.c_int32(0), C.c_uint32(0)
        check(lib.oalgpu_voice_queue_state(self.h, voice, C.byref(cur), C.byref(done)), "oalgpu_voice_queue_state")
        return cur.value, done.value

    def current_buffer(self, voice):
        return self.queue_state(voice)[0]

    # panning on the GPU: CalcDirectionCoeffs + ComputePanGains (replaces the gains of the last set_params)
    def set_ambi_map(self, index, scale):
        lib.oalgpu_context_set_ambi_map.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), f32p]
        idx = np.ascontiguousarray(index, np.uint8); sc = np.ascontiguousarray(scale, np.float32)
        check(lib.oalgpu_context_set_ambi_map(self.h, idx.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(sc)), "oalgpu_context_set_ambi_map")

    def set_slot_ambi_map(self, slot, index, scale):
        lib.oalgpu_slot_set_ambi_map.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint8), f32p]
        idx = np.ascontiguousarray(index, np.uint8); sc = np.ascontiguousarray(scale, np.float32)
        check(lib.oalgpu_slot_set_ambi_map(self.h, slot, idx.ctypes.data_as(C.POINTER(C.c_uint8)), _fp(sc)), "oalgpu_slot_set_ambi_map")

    def set_pan(self, voices, pans):
        """pans: rows of [dir x, y, z, spread, dry_gain, send_gain 0..5] (11 floats)"""
        lib.oalgpu_voice_set_pan.argtypes = [C.c_void_p, u32p, f32p, C.c_size_t]
        v = np.ascontiguousarray(voices, np.uint32)
        p = np.ascontiguousarray(pans, np.float32).reshape(len(v), 11)
        check(lib.oalgpu_voice_set_pan(self.h, v.ctypes.data_as(u32p), _fp(p), len(v)), "oalgpu_voice_set_pan")

    # near-field control (same interface as tests/oracle_lib.Scene)
    def set_nfc(self, w1, channels_per_order):
        lib.oalgpu_context_set_nfc.argtypes = [C.c_void_p, C.c_float, C.POINTER(C.c_uint32)]
        cpo = (C.c_uint32 * 5)(*(list(channels_per_order) + [0] * 5)[:5])
        check(lib.oalgpu_context_set_nfc(self.h, w1, cpo), "oalgpu_context_set_nfc")

    def set_voice_nfc(self, voice, w0):
        lib.oalgpu_voice_set_nfc.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        check(lib.oalgpu_voice_set_nfc(self.h, voice, w0), "oalgpu_voice_set_nfc")

    # B-Format sources: one voice per channel over channel views of the interleaved buffer
    # (same interface as tests/oracle_lib.Scene: `voice` = what add_ambi_voice returned)
    def add_ambi_voice(self, buffer, nch, looping, position=0, frac=0, frequency=44100):
        lib.oalgpu_buffer_channel_view.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        first = self.nvoices
        for ch in range(nch):
            view = lib.oalgpu_buffer_channel_view(self.h, buffer, ch)
            check(min(view, 0), "oalgpu_buffer_channel_view")
            self.add_voice(view, looping, position, frac, frequency)
        return first

    def set_channel_params(self, voice, channel, params):
        self.set_params(voice + channel, params)

    def set_channel_ambi_scale(self, voice, channel, xover_norm, hf_scale, lf_scale):
        lib.oalgpu_voice_set_ambi_scale.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_float]
        check(lib.oalgpu_voice_set_ambi_scale(self.h, voice + channel, xover_norm, hf_scale, lf_scale),
              "oalgpu_voice_set_ambi_scale")

    def set_params(self, voice, params):
        """params: any ctypes struct with the oalgpu_voice_params layout."""
        assert C.sizeof(params) == C.sizeof(VoiceParams)
        ids = (C.c_uint32 * 1)(voice)
        check(lib.oalgpu_voice_set_params(self.h, ids, C.byref(params), 1), "oalgpu_voice_set_params")

    def set_params_batch(self, voices, params_array):
        """voices: uint32 array; params_array: ctypes array of VoiceParams."""
        voices = np.ascontiguousarray(voices, np.uint32)
        check(lib.oalgpu_voice_set_params(self.h, voices.ctypes.data_as(u32p),
                                          C.cast(params_array, C.c_void_p), len(voices)),
              "oalgpu_voice_set_params")

    def move_async(self, moves):
        """moves: structured array of MOVE_DTYPE (voice, hrtf_ev, hrtf_az, hrtf_dist, hrtf_spread, hrtf_gain): the moved voices of
        the next update; returns without waiting (oalgpu_voice_move_async)"""
        moves = np.ascontiguousarray(moves, MOVE_DTYPE)
        lib.oalgpu_voice_move_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        check(lib.oalgpu_voice_move_async(self.h, moves.ctypes.data_as(C.c_void_p), len(moves)), "oalgpu_voice_move_async")

    def read_output_async(self):
        lib.oalgpu_read_output_async.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        t = C.c_uint32()
        check(lib.oalgpu_read_output_async(self.h, C.byref(t)), "oalgpu_read_output_async")
        return t.value

    def output_wait(self, ticket, out=None):
        n = self.desc.num_real_channels or self.desc.num_dry_channels
        if out is None:
            out = np.empty((n, BUFFER_LINE), np.float32)
        lib.oalgpu_output_wait.argtypes = [C.c_void_p, C.c_uint32, f32p, C.c_size_t]
        check(lib.oalgpu_output_wait(self.h, ticket, _fp(out), out.size), "oalgpu_output_wait")
        return out

    def voice_events_async(self):
        lib.oalgpu_voice_events_async.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        t = C.c_uint32()
        check(lib.oalgpu_voice_events_async(self.h, C.byref(t)), "oalgpu_voice_events_async")
        return t.value

    def voice_events_wait(self, ticket, capacity=1024):
        """-> list of VoiceEvent (what changed about the voices since the report before)"""
        arr = (VoiceEvent * capacity)()
        n = C.c_size_t()
        lib.oalgpu_voice_events_wait.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        check(lib.oalgpu_voice_events_wait(self.h, ticket, arr, capacity, C.byref(n)), "oalgpu_voice_events_wait")
        return [arr[i] for i in range(n.value)]

    def param_block(self, voices, params_array):
        voices = np.ascontiguousarray(voices, np.uint32)
        h = C.c_void_p()
        check(lib.oalgpu_param_block_create(self.h, voices.ctypes.data_as(u32p),
                                            C.cast(params_array, C.c_void_p), len(voices), C.byref(h)),
              "oalgpu_param_block_create")
        return h

    def apply_block(self, block):
        check(lib.oalgpu_param_block_apply(self.h, block), "oalgpu_param_block_apply")

    def mix_run(self, blocks, samples_to_do=BUFFER_LINE, post_process=False):
        """len(blocks) consecutive updates in one call (oalgpu_mix_update_run); blocks[i] may be None"""
        arr = (C.c_void_p * len(blocks))(*[b if b is not None else None for b in blocks])
        lib.oalgpu_mix_update_run.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32, C.c_int]
        check(lib.oalgpu_mix_update_run(self.h, arr, len(blocks), samples_to_do, 1 if post_process else 0), "oalgpu_mix_update_run")

    def set_stream(self, stream_ptr):
        check(lib.oalgpu_set_stream(self.h, stream_ptr), "oalgpu_set_stream")

    def mix_voices(self, samples_to_do=BUFFER_LINE):
        check(lib.oalgpu_mix_voices(self.h, samples_to_do), "oalgpu_mix_voices")

    def post_process(self, samples_to_do=BUFFER_LINE):
        check(lib.oalgpu_post_process(self.h, samples_to_do), "oalgpu_post_process")

    def mix_voices_overlapped(self, samples_to_do=BUFFER_LINE):
        check(lib.oalgpu_mix_voices_overlapped(self.h, samples_to_do), "oalgpu_mix_voices_overlapped")

    def post_stream(self):
        lib.oalgpu_post_stream.restype = C.c_void_p
        lib.oalgpu_post_stream.argtypes = [C.c_void_p]
        return lib.oalgpu_post_stream(self.h)

    def post_process_overlapped(self, samples_to_do=BUFFER_LINE, run_post_process=True):
        lib.oalgpu_post_process_overlapped.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        check(lib.oalgpu_post_process_overlapped(self.h, samples_to_do, 1 if run_post_process else 0),
              "oalgpu_post_process_overlapped")

    # multi-GPU inside the library (RCCL): see include/oalgpu.h "multi-GPU inside the library"
    def comm_init(self, unique_id, rank, world):
        lib.oalgpu_comm_init.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_int, C.c_int]
        check(lib.oalgpu_comm_init(self.h, unique_id, len(unique_id), rank, world), "oalgpu_comm_init")

    def comm_init_host(self, name, rank, world):
        """the host-staged transport (shared-memory ring `name`): ranks RCCL cannot connect, e.g. on one GPU"""
        lib.oalgpu_comm_init_host.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
        check(lib.oalgpu_comm_init_host(self.h, name.encode(), rank, world), "oalgpu_comm_init_host")

    def comm_info(self):
        """(rank, world, ranks the transport counts, transport name)"""
        lib.oalgpu_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
        r, w, t = C.c_int(0), C.c_int(0), C.c_int(0)
        kind = C.create_string_buffer(16)
        check(lib.oalgpu_comm_info(self.h, C.byref(r), C.byref(w), C.byref(t), kind, 16), "oalgpu_comm_info")
        return r.value, w.value, t.value, kind.value.decode()

    def comm_destroy(self):
        lib.oalgpu_comm_destroy.argtypes = [C.c_void_p]
        check(lib.oalgpu_comm_destroy(self.h), "oalgpu_comm_destroy")

    def set_carry_accum(self, enable):
        check(lib.oalgpu_set_carry_accum(self.h, 1 if enable else 0))

    def set_start_delay(self, voice, samples):
        lib.oalgpu_voice_set_start_delay.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        check(lib.oalgpu_voice_set_start_delay(self.h, voice, samples), "oalgpu_voice_set_start_delay")

    def set_state(self, voice, vstate):
        check(lib.oalgpu_voice_set_state(self.h, voice, vstate), "oalgpu_voice_set_state")

    def set_direct_hrtf(self, chan_coeffs, hfscales, xover_norm, irsize):
        cc = np.ascontiguousarray(chan_coeffs, np.float32)
        hf = np.ascontiguousarray(hfscales, np.float32)
        check(lib.oalgpu_set_direct_hrtf(self.h, _fp(cc), _fp(hf), xover_norm, irsize),
              "oalgpu_set_direct_hrtf")

    def set_direct_hrtf_from_store(self, points, matrix, order_hf_gain, xover_freq, ir_size=0, per_hrir_min=False):
        """DirectHrtfState::build on the context's data set (oalgpu_set_direct_hrtf_from_store)"""
        pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
        mat = np.zeros((len(pts), 16), np.float32)
        m = np.asarray(matrix, np.float32)
        mat[:, :m.shape[1]] = m
        g = np.zeros(5, np.float32)
        g[:len(order_hf_gain)] = order_hf_gain
        lib.oalgpu_set_direct_hrtf_from_store.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32,
                                                          C.POINTER(C.c_float), C.c_float, C.c_uint32, C.c_int]
        check(lib.oalgpu_set_direct_hrtf_from_store(self.h, _fp(pts), _fp(mat), len(pts), _fp(g), xover_freq, ir_size,
                                                    1 if per_hrir_min else 0), "oalgpu_set_direct_hrtf_from_store")

    # the stage behind the buses: BFormatDec, dither, output PCM (include/oalgpu.h)
    def set_bformat_decoder(self, coeffs_hf, coeffs_lf=None, xover_norm=400.0 / 48000.0):
        lib.oalgpu_set_bformat_decoder.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, C.c_float]
        if coeffs_hf is None:
            check(lib.oalgpu_set_bformat_decoder(self.h, 0, None, None, 0.0))
            return
        hf = np.ascontiguousarray(coeffs_hf, np.float32)
        lf = None if coeffs_lf is None else np.ascontiguousarray(coeffs_lf, np.float32)
        assert hf.ndim == 2 and hf.shape[1] == MAX_AMBI
        check(lib.oalgpu_set_bformat_decoder(self.h, hf.shape[0], _fp(hf), _fp(lf) if lf is not None else None, xover_norm),
              "oalgpu_set_bformat_decoder")

    def set_output(self, sample_type, dither_depth=0.0, dither_seed=22222):
        lib.oalgpu_set_output.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_uint32]
        check(lib.oalgpu_set_output(self.h, sample_type, dither_depth, dither_seed), "oalgpu_set_output")
        self._out_type = sample_type

    def read_output(self, samples_to_do=BUFFER_LINE, frame_step=2):
        lib.oalgpu_read_output.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]
        dt = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32][getattr(self, "_out_type", 6)]
        out = np.zeros(samples_to_do * frame_step, dt)
        check(lib.oalgpu_read_output(self.h, out.ctypes.data_as(C.c_void_p), samples_to_do, frame_step), "oalgpu_read_output")
        return out

    def set_slot_convolution(self, slot, conv):
        lib.oalgpu_slot_set_convolution.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        check(lib.oalgpu_slot_set_convolution(self.h, slot, conv.h if conv is not None else None))

    def set_slot_effect(self, slot, fx):
        lib.oalgpu_slot_set_effect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        check(lib.oalgpu_slot_set_effect(self.h, slot, fx.h if fx is not None else None), "oalgpu_slot_set_effect")

    def set_slot_reverb(self, slot, rev):
        lib.oalgpu_slot_set_reverb.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        check(lib.oalgpu_slot_set_reverb(self.h, slot, rev.h if rev is not None else None))

    def mix(self, samples_to_do=BUFFER_LINE, post_process=False):
        check(lib.oalgpu_mix_update(self.h, samples_to_do, 1 if post_process else 0),
              "oalgpu_mix_update")

    def sync(self):
        check(lib.oalgpu_sync(self.h), "oalgpu_sync")

    def resident_stats(self):
        """OALGPU_CTX_RESIDENT: dict of oalgpu_resident_info (launches, updates, parks, the ended launches' own time)"""
        class Info(C.Structure):
            _fields_ = [("enabled", C.c_int32), ("failed", C.c_int32), ("running", C.c_int32), ("door_in_device_memory", C.c_int32),
                        ("launches", C.c_uint32), ("updates", C.c_uint32), ("parks", C.c_uint32), ("timed_launches", C.c_uint32),
                        ("timed_updates", C.c_uint64), ("timed_kernel_ms", C.c_double),
                        ("max_updates_per_launch", C.c_uint32), ("pad", C.c_uint32),
                        ("wait_door_us", C.c_double), ("wait_reduction_us", C.c_double), ("wait_arrival_us", C.c_double),
                        ("wait_post_us", C.c_double), ("wait_reduced_us", C.c_double), ("wait_split_us", C.c_double),
                        ("install_us", C.c_double), ("busy_us", C.c_double), ("top_us", C.c_double)]
        info = Info()
        lib.oalgpu_resident_stats.argtypes = [C.c_void_p, C.POINTER(Info)]
        check(lib.oalgpu_resident_stats(self.h, C.byref(info)), "oalgpu_resident_stats")
        return {k: getattr(info, k) for k, _ in Info._fields_ if k != "pad"}

    def resident_set_short_run(self, updates):
        lib.oalgpu_resident_set_short_run.argtypes = [C.c_void_p, C.c_uint32]
        check(lib.oalgpu_resident_set_short_run(self.h, updates), "oalgpu_resident_set_short_run")

    def resident_set_timing(self, on):
        lib.oalgpu_resident_set_timing.argtypes = [C.c_void_p, C.c_int]
        check(lib.oalgpu_resident_set_timing(self.h, 1 if on else 0), "oalgpu_resident_set_timing")

    def resident_set_max_updates(self, n):
        lib.oalgpu_resident_set_max_updates.argtypes = [C.c_void_p, C.c_uint32]
        check(lib.oalgpu_resident_set_max_updates(self.h, n), "oalgpu_resident_set_max_updates")

    def dry(self):
        n = self.desc.num_dry_channels + self.desc.num_real_channels
        out = np.zeros((n, BUFFER_LINE), np.float32)
        check(lib.oalgpu_read_dry(self.h, _fp(out)), "oalgpu_read_dry")
        return out

    def wet(self, slot):
        out = np.zeros((self.desc.wet_channels, BUFFER_LINE), np.float32)
        check(lib.oalgpu_read_wet(self.h, slot, _fp(out)), "oalgpu_read_wet")
        return out

    def hrtf_accum(self):
        out = np.zeros((BUFFER_LINE + HRIR_LEN, 2), np.float32)
        check(lib.oalgpu_read_hrtf_accum(self.h, _fp(out)), "oalgpu_read_hrtf_accum")
        return out

    def voice_state(self, voice):
        st = VoiceState()
        check(lib.oalgpu_voice_readback(self.h, voice, C.byref(st)), "oalgpu_voice_readback")
        return st

    def bus_device_ptr(self):
        p, n, s = C.c_void_p(), C.c_size_t(), C.c_void_p()
        check(lib.oalgpu_bus_device_ptr(self.h, C.byref(p), C.byref(n), C.byref(s)))
        return p.value, n.value, s.value

    def voice_kernel_name(self):
        lib.oalgpu_voice_kernel_name.restype = C.c_char_p
        lib.oalgpu_voice_kernel_name.argtypes = [C.c_void_p]
        return lib.oalgpu_voice_kernel_name(self.h).decode()

    def set_timing(self, enable=True):
        check(lib.oalgpu_set_timing(self.h, 1 if enable else 0))

    def last_update_ms(self):
        a, b = C.c_float(), C.c_float()
        check(lib.oalgpu_last_update_ms(self.h, C.byref(a), C.byref(b)), "oalgpu_last_update_ms")
        return a.value, b.value


def comm_unique_id():
    """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to the other ranks)."""
    buf = C.create_string_buffer(128)
    lib.oalgpu_comm_unique_id.argtypes = [C.c_char_p, C.c_size_t]
    check(lib.oalgpu_comm_unique_id(buf, 128), "oalgpu_comm_unique_id")
    return buf.raw


class Convolution:
    """oalgpu_convolution: ConvolutionState (alc/effects/convolution.cpp); ir = [frames] (mono) or
    [frames, channels] at ir_rate (None: the device's rate)."""

    def __init__(self, num_out_lines, ir, device=0, ir_rate=None, device_rate=48000):
        lib.oalgpu_convolution_create_ex.argtypes = [C.c_int, C.c_uint32, f32p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                     C.c_uint32, C.POINTER(C.c_void_p)]
        lib.oalgpu_convolution_set_channel_gains.argtypes = [C.c_void_p, f32p]
        lib.oalgpu_convolution_set_upsample.argtypes = [C.c_void_p, f32p, f32p, C.c_float]
        lib.oalgpu_convolution_destroy.argtypes = [C.c_void_p]
        lib.oalgpu_convolution_destroy.restype = None
        lib.oalgpu_convolution_set_target_gains.argtypes = [C.c_void_p, f32p]
        lib.oalgpu_convolution_process.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32]
        lib.oalgpu_slot_set_convolution.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        self.nlines = num_out_lines
        ir = np.ascontiguousarray(ir, np.float32)
        self.channels = 1 if ir.ndim == 1 else ir.shape[1]
        h = C.c_void_p()
        check(lib.oalgpu_convolution_create_ex(device, num_out_lines, _fp(ir), ir.shape[0], self.channels,
                                               0 if ir_rate is None else ir_rate, 0 if ir_rate is None else device_rate,
                                               C.byref(h)), "oalgpu_convolution_create_ex")
        self.h = h

    def set_channel_gains(self, gains):
        g = np.ascontiguousarray(gains, np.float32).reshape(self.channels, -1)[:, :self.nlines]
        check(lib.oalgpu_convolution_set_channel_gains(self.h, _fp(np.ascontiguousarray(g))), "oalgpu_convolution_set_channel_gains")

    def set_upsample(self, hf_scales, lf_scales, xover_norm):
        if hf_scales is None:
            check(lib.oalgpu_convolution_set_upsample(self.h, None, None, 0.0))
            return
        hf = np.ascontiguousarray(hf_scales, np.float32)
        lf = np.ascontiguousarray(lf_scales, np.float32)
        check(lib.oalgpu_convolution_set_upsample(self.h, _fp(hf), _fp(lf), xover_norm), "oalgpu_convolution_set_upsample")

    def set_target_gains(self, gains):
        g = np.zeros(MAX_OUT, np.float32)
        g[:len(gains)] = gains
        check(lib.oalgpu_convolution_set_target_gains(self.h, _fp(g)))

    def process(self, wet_in, out_lines):
        wet_in = np.ascontiguousarray(wet_in, np.float32)
        assert out_lines.dtype == np.float32 and out_lines.shape == (self.nlines, BUFFER_LINE)
        check(lib.oalgpu_convolution_process(self.h, _fp(wet_in), _fp(out_lines), wet_in.size),
              "oalgpu_convolution_process")

    def close(self):
        if self.h:
            lib.oalgpu_convolution_destroy(self.h)
            self.h = None


(EFFECT_EQUALIZER, EFFECT_MODULATOR, EFFECT_ECHO, EFFECT_DEDICATED, EFFECT_COMPRESSOR, EFFECT_CHORUS, EFFECT_DISTORTION,
 EFFECT_AUTOWAH, EFFECT_VMORPHER, EFFECT_FSHIFTER, EFFECT_PSHIFTER) = range(11)
# which fields of the property struct are integers (the rest are floats), in declaration order
_EFFECT_INT_FIELDS = {EFFECT_MODULATOR: (2,), EFFECT_COMPRESSOR: (0,), EFFECT_CHORUS: (0, 1), EFFECT_VMORPHER: (1, 2, 3, 4, 5),
                      EFFECT_FSHIFTER: (1, 2), EFFECT_PSHIFTER: (0, 1)}
INVALID_CHANNEL = 0xffffffff


class Effect:
    """oalgpu_effect: the EffectStates of alc/effects/*.cpp other than the reverb and the convolution."""

    def __init__(self, kind, num_out_lines, num_in=4, sample_rate=48000, mode=MATH_FAST, device=0):
        lib.oalgpu_effect_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.oalgpu_effect_destroy.argtypes = [C.c_void_p]
        lib.oalgpu_effect_destroy.restype = None
        lib.oalgpu_effect_update.argtypes = [C.c_void_p, C.c_void_p, u32p, f32p]
        lib.oalgpu_effect_process.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32]
        lib.oalgpu_slot_set_effect.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        lib.oalgpu_effect_set_upsampler.argtypes = [C.c_void_p, f32p, C.c_float]
        self.kind, self.nlines, self.num_in = kind, num_out_lines, num_in
        h = C.c_void_p()
        check(lib.oalgpu_effect_create(device, mode, kind, sample_rate, num_in, num_out_lines, C.byref(h)), "oalgpu_effect_create")
        self.h = h

    def update(self, props, targets, gains):
        """props: the property struct's fields in order (the modulator's waveform as its third); None for dedicated"""
        g = np.ascontiguousarray(gains, np.float32)
        tp = None
        if targets is not None:
            t = np.ascontiguousarray(targets, np.uint32)
            tp = t.ctypes.data_as(u32p)
        pp = None
        if props is not None:
            raw = np.ascontiguousarray(props, np.float32).copy()
            for k in _EFFECT_INT_FIELDS.get(self.kind, ()):
                raw.view(np.int32)[k] = int(props[k])
            pp = raw.ctypes.data_as(C.c_void_p)
        check(lib.oalgpu_effect_update(self.h, pp, tp, _fp(g)), "oalgpu_effect_update")

    def set_upsampler(self, order_scales, xover_norm):
        """the A-Format effects on a device above first order; None: first order again"""
        sc = None if order_scales is None else np.ascontiguousarray(order_scales, np.float32)
        check(lib.oalgpu_effect_set_upsampler(self.h, None if sc is None else _fp(sc), xover_norm), "oalgpu_effect_set_upsampler")

    def process(self, wet_in, out_lines, n=BUFFER_LINE):
        wet_in = np.ascontiguousarray(wet_in, np.float32)
        assert wet_in.shape == (self.num_in, BUFFER_LINE) and out_lines.shape == (self.nlines, BUFFER_LINE)
        check(lib.oalgpu_effect_process(self.h, _fp(wet_in), _fp(out_lines), n), "oalgpu_effect_process")

    def close(self):
        if self.h:
            lib.oalgpu_effect_destroy(self.h)
            self.h = None


class BqCoeffs(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("b0", "b1", "b2", "a1", "a2")]


class ReverbPipelineParams(C.Structure):
    """oalgpu_reverb_pipeline."""
    _fields_ = [
        ("filter_lp", BqCoeffs), ("filter_hp", BqCoeffs),
        ("early_delay_tap", (C.c_uint32 * 2) * 4), ("early_delay_coeff", C.c_float * 2),
        ("late_delay_tap", (C.c_uint32 * 2) * 4),
        ("mix_x", C.c_float), ("mix_y", C.c_float),
        ("early_ap_coeff", C.c_float), ("early_ap_offset", C.c_uint32 * 4),
        ("early_offset", C.c_uint32 * 4), ("early_coeff", C.c_float),
        ("early_gains_target", (C.c_float * 25) * 4),
        ("late_offset", C.c_uint32 * 4), ("late_density_gain", C.c_float),
        ("t60_mid_gain", C.c_float * 4), ("t60_hf", BqCoeffs * 4), ("t60_lf", BqCoeffs * 4),
        ("mod_step", C.c_uint32), ("mod_depth", C.c_float),
        ("late_ap_coeff", C.c_float), ("late_ap_offset", C.c_uint32 * 4),
        ("late_gains_target", (C.c_float * 25) * 4),
        ("fade_sample_count", C.c_uint32),
    ]


class ReverbParams(C.Structure):
    """oalgpu_reverb_params."""
    _fields_ = [("pipeline_state", C.c_int32), ("current_pipeline", C.c_int32),
                ("pipe", ReverbPipelineParams * 2)]


class ReverbProps(C.Structure):
    """oalgpu_reverb_props (ReverbProps, core/effects/base.h:62-86); make() fills the
    AL_EAXREVERB_DEFAULT_* values (include/AL/efx.h:317-401)."""
    _fields_ = [(k, C.c_float) for k in ("density", "diffusion", "gain", "gain_hf", "gain_lf", "decay_time",
                                         "decay_hf_ratio", "decay_lf_ratio", "reflections_gain",
                                         "reflections_delay")] + [
        ("reflections_pan", C.c_float * 3), ("late_reverb_gain", C.c_float), ("late_reverb_delay", C.c_float),
        ("late_reverb_pan", C.c_float * 3)] + [(k, C.c_float) for k in (
            "echo_time", "echo_depth", "modulation_time", "modulation_depth", "air_absorption_gain_hf",
            "hf_reference", "lf_reference", "room_rolloff_factor")] + [("decay_hf_limit", C.c_int32)]

    DEFAULTS = dict(density=1.0, diffusion=1.0, gain=0.32, gain_hf=0.89, gain_lf=1.0, decay_time=1.49,
                    decay_hf_ratio=0.83, decay_lf_ratio=1.0, reflections_gain=0.05, reflections_delay=0.007,
                    reflections_pan=(0.0, 0.0, 0.0), late_reverb_gain=1.26, late_reverb_delay=0.011,
                    late_reverb_pan=(0.0, 0.0, 0.0), echo_time=0.25, echo_depth=0.0, modulation_time=0.25,
                    modulation_depth=0.0, air_absorption_gain_hf=0.994, hf_reference=5000.0,
                    lf_reference=250.0, room_rolloff_factor=0.0, decay_hf_limit=1)

    @classmethod
    def make(cls, **kw):
        d = dict(cls.DEFAULTS)
        unknown = set(kw) - set(d)
        if unknown:
            raise KeyError(sorted(unknown))
        d.update(kw)
        p = cls()
        for k, v in d.items():
            setattr(p, k, (C.c_float * 3)(*v) if isinstance(v, (tuple, list)) else v)
        return p


class Reverb:
    """oalgpu_reverb: ReverbState (alc/effects/reverb.cpp).  device=-1 makes a parameter-only
    instance (update/get_params work without a GPU; process raises)."""

    def __init__(self, num_out_lines, sample_rate=48000, device=0):
        lib.oalgpu_reverb_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
        lib.oalgpu_reverb_destroy.argtypes = [C.c_void_p]
        lib.oalgpu_reverb_destroy.restype = None
        lib.oalgpu_reverb_update.argtypes = [C.c_void_p, C.POINTER(ReverbProps), C.c_float]
        lib.oalgpu_reverb_get_params.argtypes = [C.c_void_p, C.POINTER(ReverbParams)]
        lib.oalgpu_reverb_set_params.argtypes = [C.c_void_p, C.POINTER(ReverbParams)]
        lib.oalgpu_reverb_line_lengths.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        lib.oalgpu_reverb_process.argtypes = [C.c_void_p, f32p, f32p, C.c_uint32]
        lib.oalgpu_slot_set_reverb.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        self.nlines = num_out_lines
        h = C.c_void_p()
        check(lib.oalgpu_reverb_create(device, sample_rate, num_out_lines, C.byref(h)), "oalgpu_reverb_create")
        self.h = h

    def set_math_mode(self, mode):
        """MATH_EXACT (default): bit-identical to ReverbState::process; MATH_FAST: the filter sections as block scans"""
        lib.oalgpu_reverb_set_math_mode.argtypes = [C.c_void_p, C.c_int]
        check(lib.oalgpu_reverb_set_math_mode(self.h, mode), "oalgpu_reverb_set_math_mode")

    def set_upmix(self, order_scales, first_order_up, xover_norm):
        """a device above first order: MixOutAmbiUp (see oalgpu_reverb_set_upmix); None = MixOutPlain"""
        lib.oalgpu_reverb_set_upmix.argtypes = [C.c_void_p, f32p, f32p, C.c_float]
        if order_scales is None:
            check(lib.oalgpu_reverb_set_upmix(self.h, None, None, 0.0))
            return
        sc = np.ascontiguousarray(order_scales, np.float32)
        up = np.ascontiguousarray(first_order_up, np.float32).reshape(4, 25)
        check(lib.oalgpu_reverb_set_upmix(self.h, _fp(sc), _fp(up), xover_norm), "oalgpu_reverb_set_upmix")

    def update(self, props, slot_gain=1.0):
        check(lib.oalgpu_reverb_update(self.h, C.byref(props), slot_gain), "oalgpu_reverb_update")

    def get_params(self):
        out = ReverbParams()
        check(lib.oalgpu_reverb_get_params(self.h, C.byref(out)), "oalgpu_reverb_get_params")
        return out

    def set_params(self, params):
        """`params`: a ReverbParams, or any ctypes structure of the same layout."""
        assert C.sizeof(params) == C.sizeof(ReverbParams)
        check(lib.oalgpu_reverb_set_params(self.h, C.cast(C.byref(params), C.POINTER(ReverbParams))),
              "oalgpu_reverb_set_params")

    def line_lengths(self):
        out = (C.c_uint32 * 11)()
        total = lib.oalgpu_reverb_line_lengths(self.h, out)
        return total, list(out)

    def skip(self, n):
        lib.oalgpu_reverb_skip.argtypes = [C.c_void_p, C.c_uint32]
        check(lib.oalgpu_reverb_skip(self.h, n), "oalgpu_reverb_skip")

    def process_n(self, wet_in, out_lines, n):
        wet_in = np.ascontiguousarray(wet_in, np.float32)
        assert wet_in.shape == (4, BUFFER_LINE)
        assert out_lines.dtype == np.float32 and out_lines.shape == (self.nlines, BUFFER_LINE)
        check(lib.oalgpu_reverb_process(self.h, _fp(wet_in), _fp(out_lines), n), "oalgpu_reverb_process")

    def process(self, wet_in, out_lines):
        self.process_n(wet_in, out_lines, BUFFER_LINE)

    def close(self):
        if self.h:
            lib.oalgpu_reverb_destroy(self.h)
            self.h = None
