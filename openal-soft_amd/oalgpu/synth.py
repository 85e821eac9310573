"""Synthetic inputs shared by tests/ and bench.py (no reference data is shipped).

* ``synth_mhr_bytes()`` builds an HRTF data set in the reference's on-disk ``.mhr`` v3 format
  (``MinPHR03``, core/hrtf_loader.cpp:583-721) with exactly the geometry of the reference's
  ``hrtf/Default HRTF.mhr`` -- 48 kHz, left-only (mirrored), irSize 64, one field at 1.2 m,
  13 elevations with azimuth counts {1, 180 x 11, 1} = 1982 HRIRs -- but synthetic impulse
  responses (decaying noise with a direction-dependent onset/ITD).  BASELINE config 3/5 cost
  depends only on that geometry; parity holds for any coefficient values.
* ``Lcg`` is the 32-bit LCG of SURVEY.md section 8(d) used to build the synthetic scenes.
"""
import struct

import numpy as np

DEFAULT_AZ_COUNTS = [1] + [180] * 11 + [1]


class Lcg:
    """s = s*1664525 + 1013904223 (mod 2^32); SURVEY.md 8(d)."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFF

    def next_u32(self):
        self.s = (self.s * 1664525 + 1013904223) & 0xFFFFFFFF
        return self.s

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.next_u32() / 4294967296.0)

    def array_u32(self, n):
        out = np.empty(n, np.uint64)
        s = self.s
        for i in range(n):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            out[i] = s
        self.s = s
        return out.astype(np.uint32)


def lcg_block_f32(seed, n):
    """n floats uniform in [-1, 1) from the LCG, vectorised (jump-ahead by doubling)."""
    # x_k = a^k x_0 + c (a^k - 1)/(a - 1); build a^k and the geometric sums by prefix doubling.
    a, c, mask = 1664525, 1013904223, 0xFFFFFFFF
    mult = np.empty(n, np.uint64)
    add = np.empty(n, np.uint64)
    mult[0], add[0] = a, c
    filled = 1
    while filled < n:
        take = min(filled, n - filled)
        am, aa = mult[filled - 1], add[filled - 1]   # advance-by-`filled` map
        mult[filled:filled + take] = (mult[:take] * am) & mask
        add[filled:filled + take] = (add[:take] * am + aa) & mask
        filled += take
    x = (mult * np.uint64(seed & mask) + add) & np.uint64(mask)
    return (x.astype(np.float64) / 2147483648.0 - 1.0).astype(np.float32)


def synth_mhr_bytes(seed=0x5EED1234, ir_size=64, az_counts=None, rate=48000, dist_mm=1200, fields=None, stereo=False):
    """MinPHR03 data set.  fields: [(distance in mm, azimuth counts per elevation), ...] in the file's order
    (every distance smaller than the one before, core/hrtf_loader.cpp:627-632); default: one field.
    stereo: channel type 1 (left AND right responses stored, no mirroring)."""
    if fields is None:
        fields = [(dist_mm, list(DEFAULT_AZ_COUNTS if az_counts is None else az_counts))]
    fields = [(int(d), list(a)) for d, a in fields]
    rng = np.random.default_rng(seed)
    hdr = b"MinPHR03" + struct.pack("<IBBB", rate, 1 if stereo else 0, ir_size, len(fields))
    for d, azs in fields:
        hdr += struct.pack("<HB", d, len(azs)) + bytes(azs)
    n_ir = sum(sum(a) for _, a in fields)
    ears = 2 if stereo else 1
    t = np.arange(ir_size)
    coeffs = np.zeros((n_ir, ir_size, ears), np.float64)
    delays = np.zeros((n_ir, ears), np.uint8)
    k = 0
    for fi, (d, az_counts) in enumerate(fields):
        ev_count = len(az_counts)
        near = 1.0 + 0.25 * fi                        # nearer fields: a little louder
        for e, azc in enumerate(az_counts):
            ev = -np.pi / 2 + np.pi * e / (ev_count - 1)
            for a in range(azc):
                az = 2 * np.pi * a / azc
                for ear in range(ears):
                    lateral = np.sin(az) * np.cos(ev) * (1.0 if ear == 0 else -1.0)    # +1 = source on the far side of this ear
                    onset = 4.0 + 3.0 * (1.0 + lateral)        # the far ear hears the source later
                    env = np.exp(-np.maximum(t - onset, 0) / 6.0) * (t >= np.floor(onset))
                    ir = rng.standard_normal(ir_size) * env * (0.25 + 0.2 * (1.0 - lateral)) * near
                    coeffs[k, :, ear] = np.clip(ir, -0.95, 0.95)
                    delays[k, ear] = int(round((20.0 + 18.0 * (1.0 + lateral)) * 4.0))   # quarter samples, <= 252
                k += 1
    q = np.round(coeffs * 8388608.0).astype(np.int64)
    q = np.clip(q, -8388608, 8388607) & 0xFFFFFF
    b = np.empty((n_ir, ir_size, ears, 3), np.uint8)
    b[..., 0] = q & 0xFF
    b[..., 1] = (q >> 8) & 0xFF
    b[..., 2] = (q >> 16) & 0xFF
    return hdr + b.tobytes() + delays.tobytes()


def write_synth_mhr(path, **kw):
    data = synth_mhr_bytes(**kw)
    with open(path, "wb") as f:
        f.write(data)
    return path


# ---------------------------------------------------------------------------------------------
# Synthetic multi-voice scenes of SURVEY.md section 8(d) (shared by bench.py and the tests)
# ---------------------------------------------------------------------------------------------
SRC_RATE = 44100
DEV_RATE = 48000
STEP_44K1 = 60211          # fastf2u(44100/48000 * 65536), alc/alu.cpp:1685
BUFFER_FRAMES = 48000


def scene_buffers(config_id, nvoices, fmt="f32"):
    """min(nvoices, 256) distinct mono buffers, 48000 frames, uniform [-1, 1)."""
    nbuf = min(nvoices, 256)
    out = []
    for b in range(nbuf):
        x = lcg_block_f32(0x5EED0000 + config_id * 4096 + b, BUFFER_FRAMES)
        if fmt == "i16":
            x = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16)
        out.append(x)
    return out


class SceneScript:
    """Per-voice parameters for update 0 (all voices) and for every later update (the moving
    quarter of the voices get a new random direction): config 2 (7.1 dry bus, 5 ambisonic
    lines), config 3 (HRTF), config 4 (config 2 voices with v % 5 active sends into 4 reverb
    slots, every third send low-passed) or config 5 (config 3 voices with one send into a
    convolution slot)."""

    def __init__(self, config_id, nvoices, voice_base=0, voice_map=None):
        self.config_id = config_id
        self.nvoices = nvoices
        self.voice_base = voice_base          # global index of local voice 0 (multi-GPU shards)
        self.voice_map = voice_map            # or: the global index of every local voice (shards by cost class)
        self.hrtf = config_id in (3, 5)

    def gv(self, v):
        """global index of local voice v"""
        return self.voice_map[v] if self.voice_map is not None else self.voice_base + v

    def _rng(self, gv, update):
        return Lcg(0x5EED0000 + self.config_id + 7919 * gv + 104729 * update)

    def start_position(self, v):
        return (self.gv(v) * 7919) % BUFFER_FRAMES

    def buffer_of(self, v, nbuf):
        return self.gv(v) % nbuf

    def direction(self, v, update):
        r = self._rng(self.gv(v), update if self.is_moving(v) else 0)
        az = r.uniform(-np.pi, np.pi)
        ev = float(np.arcsin(r.uniform(-1.0, 1.0)))
        gain = 10.0 ** (r.uniform(-60.0, -20.0) / 20.0)
        return ev, az, gain

    def is_moving(self, v):
        return self.gv(v) % 4 == 0

    def filter_active(self, v):
        return self.gv(v) % 4 == 1

    def fill(self, p, v, update):
        """p: a ctypes struct with the oalgpu_voice_params / oal_voice_params layout."""
        ev, az, gain = self.direction(v, update)
        p.step = STEP_44K1
        p.resampler = 7                        # Resampler::BSinc24
        p.direct_filter.active = 1 if self.filter_active(v) else 0
        p.direct_filter.gain_hf = 0.5 if self.filter_active(v) else 1.0
        p.direct_filter.hf_norm = 5000.0 / DEV_RATE
        p.direct_filter.gain_lf = 1.0
        p.direct_filter.lf_norm = 250.0 / DEV_RATE
        for i in range(6):
            p.send_slot[i] = -1
            p.send_filter[i].active = 0
            p.send_filter[i].gain_hf = 1.0
            p.send_filter[i].hf_norm = 5000.0 / DEV_RATE
            p.send_filter[i].gain_lf = 1.0
            p.send_filter[i].lf_norm = 250.0 / DEV_RATE
        nsends = {4: self.gv(v) % 5, 5: 1}.get(self.config_id, 0)
        if nsends:
            # first-order encode of the direction onto the slot's 4-line wet bus, -10 dB
            x, y, z = np.cos(az) * np.cos(ev), np.sin(az) * np.cos(ev), np.sin(ev)
            wet = [1.0, y * 1.7320508, z * 1.7320508, x * 1.7320508]
            for i in range(nsends):
                p.send_slot[i] = i
                if (self.gv(v) + i) % 3 == 0 and self.config_id == 4:
                    p.send_filter[i].active = 1
                    p.send_filter[i].gain_hf = 0.7
                for c in range(4):
                    p.send_gains[i][c] = 0.316 * gain * wet[c]
        if self.hrtf:
            p.hrtf_ev, p.hrtf_az, p.hrtf_dist, p.hrtf_spread, p.hrtf_gain = ev, az, 2.0, 0.0, gain
        else:
            # 2nd-order 2D ambisonic encode of the direction (W, Y, X, V, U), N3D-like weights
            x, y = np.cos(az) * np.cos(ev), np.sin(az) * np.cos(ev)
            coeffs = [1.0, y * 1.7320508, x * 1.7320508, 2.0 * x * y * 1.9364917,
                      (x * x - y * y) * 1.9364917]
            for c in range(5):
                p.dry_gains[c] = gain * coeffs[c]
        return p


# ---------------------------------------------------------------------------------------------
# Speaker decoders of the reference's built-in layouts (input data of BFormatDec, as InitPanning
# builds them, alc/panning.cpp:719-850): rows = real output lines, 25 columns = dry lines.
# ---------------------------------------------------------------------------------------------
def stereo_decoder():
    """StereoConfig, alc/panning.cpp:548-556: first-order 2D (W, Y, X) -> FrontLeft, FrontRight, single band."""
    m = np.zeros((2, 25), np.float32)
    m[0, :3] = [5.00000000e-1, 2.88675135e-1, 5.52305643e-2]
    m[1, :3] = [5.00000000e-1, -2.88675135e-1, 5.52305643e-2]
    return m, None


def x71_decoder():
    """X71Config, alc/panning.cpp:609-631: second-order 2D (5 dry lines) -> BackLeft, SideLeft, FrontLeft,
    FrontRight, SideRight, BackRight, dual band (HF order gains sqrt(2), sqrt(3/2), sqrt(1/2); LF 1), on the
    8 real output lines of a 7.1 device (FL FR FC LFE BL BR SL SR, core/devformat.cpp); returns (hf, lf)."""
    rows = np.array([[1.66666667e-1, 9.62250449e-2, -1.66666667e-1, -1.49071198e-1, 8.60662966e-2],
                     [1.66666667e-1, 1.92450090e-1, 0.0, 0.0, -1.72132593e-1],
                     [1.66666667e-1, 9.62250449e-2, 1.66666667e-1, 1.49071198e-1, 8.60662966e-2],
                     [1.66666667e-1, -9.62250449e-2, 1.66666667e-1, -1.49071198e-1, 8.60662966e-2],
                     [1.66666667e-1, -1.92450090e-1, 0.0, 0.0, -1.72132593e-1],
                     [1.66666667e-1, -9.62250449e-2, -1.66666667e-1, 1.49071198e-1, 8.60662966e-2]], np.float32)
    order = [0, 1, 1, 2, 2]                                   # AmbiIndex::OrderFrom2DChannel
    hf_gain = np.array([1.41421356, 1.22474487, 7.07106781e-1], np.float32)
    lines = [4, 6, 0, 1, 7, 5]
    hf = np.zeros((8, 25), np.float32)
    lf = np.zeros((8, 25), np.float32)
    for row, line in zip(rows, lines):
        hf[line, :5] = row * hf_gain[order]
        lf[line, :5] = row
    return hf, lf


# ---- InitHrtfPanning's first-order virtual-speaker layout (alc/panning.cpp:861-869, :941-950, :1021-1023): the cube's
# eight corners, the decoder rows (W, Y, Z, X in ACN order) and the per-order high-frequency gains
_D35 = np.float32(6.154797087e-01)
_D45 = np.float32(np.float32(np.pi) / np.float32(2.0)) / np.float32(2.0)       # Deg_90 / 2.0f
_D135 = np.float32(_D45 * np.float32(3.0))                                     # Deg_45 * 3.0f
AMBI_POINTS_1O = np.array([(_D35, -_D45), (_D35, -_D135), (_D35, _D45), (_D35, _D135),
                           (-_D35, -_D45), (-_D35, -_D135), (-_D35, _D45), (-_D35, _D135)], np.float32)
AMBI_MATRIX_1O = 0.125 * np.array([(1, 1, 1, 1), (1, 1, 1, -1), (1, -1, 1, 1), (1, -1, 1, -1),
                                   (1, 1, -1, 1), (1, 1, -1, -1), (1, -1, -1, 1), (1, -1, -1, -1)], np.float32)
AMBI_ORDER_HF_GAIN_1O = np.array([2.0, 1.154700538, 0.0, 0.0, 0.0], np.float32)
