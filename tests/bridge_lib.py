"""ctypes binding of oracle/_ref/liboalbridge.so -- TEST INFRASTRUCTURE: the reference-side binding of the
product (oracle/ref_bridge.cpp), i.e. the reference's real DeviceBase::renderSamples / ProcessContexts /
CalcVoiceParams with Voice::mix routed to the reference itself (mode CPU), to the reference on top of
GPU per-call adapters (mode ADAPTERS), or to ONE batched oalgpu_mix_update per update (mode BATCH)."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "liboalbridge.so")
MODE_CPU, MODE_ADAPTERS, MODE_BATCH = 0, 1, 2
RS_LINEAR = 1
RS_SPLINE = 2                           # enum class Resampler (core/mixer/defs.h:31-43): Point, Linear, Spline, Gaussian, ...
RS_BSINC24 = 7
GOLDEN = os.path.join(ROOT, "tests", "golden")          # holds default_hrtf.mhr, the reference's own Default HRTF.mhr
f32p = C.POINTER(C.c_float)


def available():
    return os.path.exists(PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(PATH)
        L.oalbridge_create.restype = C.c_void_p
        L.oalbridge_create.argtypes = [C.c_int, C.c_uint32, C.c_int]
        L.oalbridge_create_ex.restype = C.c_void_p
        L.oalbridge_create_ex.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_char_p, C.c_uint32]
        L.oalbridge_create_ambi2.restype = C.c_void_p
        L.oalbridge_create_ambi2.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_float]
        L.oalbridge_add_source_bformat2d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float,
                                                                                                               C.c_int, C.c_float]
        L.oalbridge_add_source_callback.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float]
        L.oalbridge_add_reverb_slot.argtypes = [C.c_void_p, C.c_void_p, C.c_float]
        L.oalbridge_add_buffer_i16.argtypes = [C.c_void_p, C.POINTER(C.c_int16), C.c_uint32, C.c_uint32, C.c_uint32]
        L.oalbridge_add_source_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float,
                                                                                                      C.c_int, C.c_float, C.c_float]
        L.oalbridge_update_source_ex.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float,
                                                                                         C.c_int, C.c_float, C.c_float]
        L.oalbridge_restart_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4 + [
            C.c_int, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float]
        L.oalbridge_batch_live_voices.argtypes = [C.c_void_p]
        L.oalbridge_source_flags.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
        L.oalbridge_adapter_calls.argtypes = [C.POINTER(C.c_uint64)]
        L.oalbridge_destroy.argtypes = [C.c_void_p]
        L.oalbridge_add_buffer.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oalbridge_add_source.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float]
        L.oalbridge_update_source.argtypes = [C.c_void_p, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float]
        L.oalbridge_stop_source.argtypes = [C.c_void_p, C.c_int]
        L.oalbridge_render.argtypes = [C.c_void_p, f32p, C.c_uint32]
        L.oalbridge_error.restype = C.c_char_p
        L.oalbridge_error.argtypes = [C.c_void_p]
        L.oalbridge_source_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
        L.oalbridge_render_lines.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_int, C.c_float, C.POINTER(C.c_uint32),
                                             C.c_void_p, C.c_uint32, C.c_uint32]
        L.oalbridge_add_buffer_interleaved.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oalbridge_add_source_stereo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float]
        L.oalbridge_add_source_queue.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_uint32, C.c_int] + [C.c_float] * 4 + [C.c_int, C.c_float, C.c_float]
        L.oalbridge_queue_buffer.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.oalbridge_source_buffer.argtypes = [C.c_void_p, C.c_int]
        L.oalbridge_set_start_delay.argtypes = [C.c_void_p, C.c_int, C.c_uint32]
        L.oalbridge_replace_buffer.argtypes = [C.c_void_p, C.c_int, f32p, C.c_uint32, C.c_int]
        L.oalbridge_batch_live_buffers.argtypes = [C.c_void_p]
        L.oalbridge_track_changes.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


class Bridge:
    def __init__(self, mode, math_mode=1, sample_rate=48000, hrtf=False, num_sends=0, mhr_dir=GOLDEN, ambi2=False, control_distance=0.0):
        """hrtf: a RenderMode::Hrtf device on the .mhr under mhr_dir (the reference's InitHrtfPanning set-up);
        num_sends: DeviceBase::NumAuxSends; ambi2: the stereo device mixing second-order 2D ambisonics (first-order B-Format
        sources are VoiceFlag::IsAmbisonic on it), with near-field control when control_distance > 0 (InitNearFieldCtrl)."""
        if ambi2:
            self.h = lib().oalbridge_create_ambi2(mode, sample_rate, math_mode, num_sends, control_distance)
        else:
            self.h = lib().oalbridge_create_ex(mode, sample_rate, math_mode, 1 if hrtf else 0, mhr_dir.encode(), num_sends)
        assert self.h, "oalbridge_create_ex failed (no .mhr under mhr_dir?)"

    def add_reverb_slot(self, props, gain=1.0):
        """props: an oracle_lib.ReverbProps (the C view of ReverbProps); returns the slot index."""
        return lib().oalbridge_add_reverb_slot(self.h, C.byref(props), gain)

    def add_buffer_i16(self, data, loop_start=0, loop_end=None):
        data = np.ascontiguousarray(data, np.int16)
        return lib().oalbridge_add_buffer_i16(self.h, data.ctypes.data_as(C.POINTER(C.c_int16)), data.size, loop_start,
                                              data.size if loop_end is None else loop_end)

    def add_source_ex(self, buffer, looping, position, gain, pos, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf):
        return lib().oalbridge_add_source_ex(self.h, buffer, 1 if looping else 0, position, gain, *pos, resampler, pitch, gain_hf,
                                             send_slot, send_gain, send_gain_hf)

    def update_source_ex(self, source, gain, pos, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf):
        lib().oalbridge_update_source_ex(self.h, source, gain, *pos, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf)

    def restart_source(self, source, buffer, looping, position, gain, pos, resampler, pitch, gain_hf, send_slot, send_gain, send_gain_hf):
        """The (Stopped) source's Voice object starts over as another source."""
        rc = lib().oalbridge_restart_source(self.h, source, buffer, 1 if looping else 0, position, gain, *pos, resampler, pitch,
                                            gain_hf, send_slot, send_gain, send_gain_hf)
        assert rc == 0, "the source's voice is not Stopped"

    def batch_live_voices(self):
        return lib().oalbridge_batch_live_voices(self.h)

    def source_flags(self, source):
        st = (C.c_int32 * 3)()
        lib().oalbridge_source_flags(self.h, source, st)
        return tuple(st)

    def error(self):
        """the first error the batch mixer reported (an update that went back to the CPU loop), '' if none"""
        lib().oalbridge_error.restype = C.c_char_p
        lib().oalbridge_error.argtypes = [C.c_void_p]
        return (lib().oalbridge_error(self.h) or b"").decode()

    def close(self):
        if self.h:
            lib().oalbridge_destroy(self.h)
            self.h = None

    # ---- the voice kinds beyond mono static sources, and buffer lifetime
    def add_buffer_interleaved(self, data, channels, loop_start=0, loop_end=None):
        data = np.ascontiguousarray(data, np.float32)
        frames = data.size // channels
        return lib().oalbridge_add_buffer_interleaved(self.h, data.ctypes.data_as(f32p), frames, channels, loop_start,
                                                      frames if loop_end is None else loop_end)

    def add_source_stereo(self, buffer, looping, position, gain, pos, resampler=RS_LINEAR, pitch=1.0, gain_hf=1.0):
        src = lib().oalbridge_add_source_stereo(self.h, buffer, 1 if looping else 0, position, gain, *pos, resampler, pitch, gain_hf)
        assert src >= 0
        return src

    def add_source_callback(self, total_frames, seed, gain, pos, resampler=RS_LINEAR, pitch=1.0, gain_hf=1.0):
        """a playing AL_SOFT_callback_buffer source whose function yields total_frames frames, then comes up short"""
        src = lib().oalbridge_add_source_callback(self.h, total_frames, seed, gain, *pos, resampler, pitch, gain_hf)
        assert src >= 0
        return src

    def add_source_bformat2d(self, buffer, looping, position, gain, pos, resampler=RS_LINEAR, pitch=1.0, gain_hf=1.0, send_slot=-1, send_gain=1.0):
        src = lib().oalbridge_add_source_bformat2d(self.h, buffer, 1 if looping else 0, position, gain, *pos, resampler, pitch, gain_hf,
                                                   send_slot, send_gain)
        assert src >= 0
        return src

    def add_source_queue(self, buffers, looping, gain, pos, resampler=RS_LINEAR, pitch=1.0, gain_hf=1.0):
        arr = (C.c_int * len(buffers))(*buffers)
        src = lib().oalbridge_add_source_queue(self.h, arr, len(buffers), 1 if looping else 0, gain, *pos, resampler, pitch, gain_hf)
        assert src >= 0
        return src

    def queue_buffer(self, last_buffer, buffer):
        lib().oalbridge_queue_buffer(self.h, last_buffer, buffer)

    def source_buffer(self, source):
        return lib().oalbridge_source_buffer(self.h, source)

    def set_start_delay(self, source, samples):
        lib().oalbridge_set_start_delay(self.h, source, samples)

    def replace_buffer(self, buffer, data, forget):
        data = np.ascontiguousarray(data, np.float32)
        assert lib().oalbridge_replace_buffer(self.h, buffer, data.ctypes.data_as(f32p), data.size, 1 if forget else 0) == 0

    def batch_live_buffers(self):
        return lib().oalbridge_batch_live_buffers(self.h)

    def set_pipelined(self, depth=2):
        """the batch mixer's pipelined mode (before the first render): outputs `depth` updates late"""
        lib().oalbridge_set_pipelined.argtypes = [C.c_void_p, C.c_uint32]
        lib().oalbridge_set_pipelined(self.h, depth)

    def drain(self, frames, max_updates=4):
        """what is outstanding in the pipelined mode: [updates][frames][2]"""
        out = np.zeros((max_updates, frames, 2), np.float32)
        lib().oalbridge_drain.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32]
        n = lib().oalbridge_drain(self.h, out.ctypes.data_as(f32p), frames, max_updates)
        if n < 0:
            raise RuntimeError("oalbridge_drain: " + lib().oalbridge_error(self.h).decode())
        return out[:n]

    def leave_pipelined(self):
        lib().oalbridge_leave_pipelined.argtypes = [C.c_void_p]
        lib().oalbridge_leave_pipelined(self.h)

    def batch_times(self):
        """seconds the batch mixer's flush spent so far: (walking the voices, submitting, collecting)"""
        t = (C.c_double * 3)()
        lib().oalbridge_batch_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        lib().oalbridge_batch_times(self.h, t)
        return tuple(t)

    def track_changes(self, on=True):
        lib().oalbridge_track_changes(self.h, 1 if on else 0)

    def hook_alu(self, on=True):
        """the binding's hooks INSIDE alc/alu.cpp (include/oalgpu_openal_hooks.hpp; the bridge library is built with
        oracle/_ref/alu_hooked.cpp): CalcVoiceParams names the voices it recomputes and CalcPanningAndFilters hands the batch mixer
        directions instead of blended responses (the device context evaluates HrtfStore::getCoeffs)"""
        lib().oalbridge_hook_alu.argtypes = [C.c_void_p, C.c_int]
        lib().oalbridge_hook_alu(self.h, 1 if on else 0)

    def hooked_directions(self):
        lib().oalbridge_hooked_directions.argtypes = [C.c_void_p]
        lib().oalbridge_hooked_directions.restype = C.c_ulonglong
        return int(lib().oalbridge_hooked_directions(self.h))

    def add_buffer(self, data, loop_start=0, loop_end=None):
        data = np.ascontiguousarray(data, np.float32)
        return lib().oalbridge_add_buffer(self.h, data.ctypes.data_as(f32p), data.size, loop_start,
                                          data.size if loop_end is None else loop_end)

    def add_source(self, buffer, looping, position, gain, pos, resampler=RS_LINEAR, pitch=1.0, gain_hf=1.0):
        return lib().oalbridge_add_source(self.h, buffer, 1 if looping else 0, position, gain, *pos, resampler, pitch, gain_hf)

    def update_source(self, source, gain, pos, resampler=RS_LINEAR, pitch=1.0, gain_hf=1.0):
        lib().oalbridge_update_source(self.h, source, gain, *pos, resampler, pitch, gain_hf)

    def stop_source(self, source):
        lib().oalbridge_stop_source(self.h, source)

    def render(self, frames):
        out = np.zeros((frames, 2), np.float32)
        rc = lib().oalbridge_render(self.h, out.ctypes.data_as(f32p), frames)
        assert rc == 0, lib().oalbridge_error(self.h).decode()
        return out

    def render_lines(self, lines, fmt, dither_depth, seed, frames, frame_step):
        """The reference's own ApplyDither + Write<T> on `lines` (<= 2 x 1024); returns (pcm, new seed)."""
        dt = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.float32][fmt]
        lines = np.ascontiguousarray(lines, np.float32)
        out = np.zeros(frames * frame_step, dt)
        sd = C.c_uint32(seed)
        rc = lib().oalbridge_render_lines(self.h, lines.ctypes.data_as(f32p), lines.shape[0], fmt, dither_depth,
                                          C.byref(sd), out.ctypes.data_as(C.c_void_p), frames, frame_step)
        assert rc == 0
        return out, sd.value

    def source_state(self, source):
        st = (C.c_int32 * 4)()
        lib().oalbridge_source_state(self.h, source, st)
        return tuple(st)


def build_config1(b, nsources=64, seed=0x5EED0001, resampler=RS_LINEAR, filtered=False):
    """BASELINE configs[0]: 64 mono f32 sources at 44.1 kHz on a 48 kHz stereo device, linear resampler,
    no effects; sources around the listener (start positions (v * 7919) % 48000)."""
    rng = np.random.default_rng(seed)
    bufs = [b.add_buffer(rng.uniform(-1, 1, 48000).astype(np.float32)) for _ in range(8)]
    srcs = []
    for v in range(nsources):
        az = rng.uniform(-np.pi, np.pi)
        dist = rng.uniform(1.0, 4.0)
        pos = (float(np.sin(az) * dist), 0.0, float(-np.cos(az) * dist))
        gain = float(10 ** (rng.uniform(-40, -12) / 20))
        srcs.append(b.add_source(bufs[v % 8], True, (v * 7919) % 48000, gain, pos, resampler=resampler,
                                 gain_hf=0.5 if (filtered and v % 4 == 1) else 1.0))
    return srcs


def move_some(b, srcs, k, seed=0x5EED0001, resampler=RS_LINEAR):
    rng = np.random.default_rng(seed + 1000 * k)
    for v in srcs[::4]:
        az = rng.uniform(-np.pi, np.pi)
        dist = rng.uniform(1.0, 4.0)
        b.update_source(v, float(10 ** (rng.uniform(-40, -12) / 20)),
                        (float(np.sin(az) * dist), 0.0, float(-np.cos(az) * dist)), resampler=resampler)


def adapter_calls():
    """Invocations of the four per-call adapters so far (process-wide): resample, mix, mix_hrtf, mix_hrtf_blend."""
    out = (C.c_uint64 * 4)()
    lib().oalbridge_adapter_calls(out)
    return tuple(out)


def _direction(rng):
    az = rng.uniform(-np.pi, np.pi)
    ev = np.arcsin(rng.uniform(-1.0, 1.0))
    d = 2.0                                        # sources 2 m away
    return (float(np.sin(az) * np.cos(ev) * d), float(np.sin(ev) * d), float(-np.cos(az) * np.cos(ev) * d))


def build_config3(b, nsources=256, seed=0x5EED0003, i16=False, slot=0):
    """BASELINE configs[2] behind the reference's own voice loop: `nsources` mono sources at 44.1 kHz on the 48 kHz HRTF
    device, bsinc24, random directions 2 m away, gains 10^(U(-60,-20)/20), a quarter of them filtered (gainHF 0.5), every
    source with send 0 into effect slot `slot` (a quarter of those through the send's own filter)."""
    rng = np.random.default_rng(seed)
    bufs = []
    for _ in range(8):
        x = rng.uniform(-1, 1, 48000).astype(np.float32)
        bufs.append(b.add_buffer_i16(np.round(x * 32767.0).astype(np.int16)) if i16 else b.add_buffer(x))
    srcs = []
    for v in range(nsources):
        gain = float(10 ** (rng.uniform(-60, -20) / 20))
        # one source in sixteen does not loop and is about to run out of buffer: it ends inside the third update
        # (Voice::mix then drops the buffer, sets Stopping and the source fades out, voice.cpp:1201-1232)
        ends = v % 16 == 5
        srcs.append(b.add_source_ex(bufs[v % 8], not ends, 48000 - 2200 - v if ends else (v * 7919) % 48000, gain, _direction(rng),
                                    RS_BSINC24, 1.0, 0.5 if v % 4 == 1 else 1.0, slot, 0.5, 0.7 if v % 4 == 2 else 1.0))
    return srcs


def move_config3(b, srcs, k, seed=0x5EED0003, slot=0):
    """every 4th source gets a new direction and gain (CalcVoiceParams runs for those: a new Hrtf.Target, MixHrtfBlend)"""
    rng = np.random.default_rng(seed + 1000 * k)
    for v in srcs[::4]:
        b.update_source_ex(v, float(10 ** (rng.uniform(-60, -20) / 20)), _direction(rng), RS_BSINC24, 1.0,
                           0.5 if v % 4 == 1 else 1.0, slot, 0.5, 0.7 if v % 4 == 2 else 1.0)
