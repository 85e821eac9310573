"""IMA4 and MS ADPCM ENCODERS for the tests (the reference only decodes: LoadSamples<IMA4Data> /
LoadSamples<MSADPCMData>, core/voice.cpp:288-484).  Each encoder runs the decoder's state machine to pick its
nibbles, so it returns the compressed blocks AND the int16 samples a conforming decoder must produce from
them -- the expectation the compiled reference (CPU tests) and the GPU decode kernel are both held to.

Block layouts, as the reference's decoders read them:
  IMA4   per block: for each channel [sample i16le][step index i16le]; then 4-byte words of 8 nibbles (low
         nibble first), the channels' words interleaved.  samples_per_block = 8k + 1.
  MS     per block: [predictor u8] x ch, [scale i16le] x ch, [sample 1 (newer) i16le] x ch, [sample 2 (older)
         i16le] x ch; then nibbles interleaved per channel, high nibble first.  samples_per_block even."""
import numpy as np

IMA_STEP = [7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 19, 21, 23, 25, 28, 31, 34, 37, 41, 45, 50, 55, 60, 66, 73, 80, 88, 97, 107,
            118, 130, 143, 157, 173, 190, 209, 230, 253, 279, 307, 337, 371, 408, 449, 494, 544, 598, 658, 724, 796, 876,
            963, 1060, 1166, 1282, 1411, 1552, 1707, 1878, 2066, 2272, 2499, 2749, 3024, 3327, 3660, 4026, 4428, 4871,
            5358, 5894, 6484, 7132, 7845, 8630, 9493, 10442, 11487, 12635, 13899, 15289, 16818, 18500, 20350, 22385,
            24623, 27086, 29794, 32767]
IMA_CODEWORD = [1, 3, 5, 7, 9, 11, 13, 15, -1, -3, -5, -7, -9, -11, -13, -15]
IMA_ADJUST = [-1, -1, -1, -1, 2, 4, 6, 8] * 2
MS_ADAPTION = [230, 230, 230, 230, 307, 409, 512, 614, 768, 614, 512, 409, 307, 230, 230, 230]
MS_COEFF = [(256, 0), (512, -256), (0, 0), (192, 64), (240, 0), (460, -208), (392, -232)]


def cdiv(a, b):
    """C integer division (truncates toward zero)"""
    q = abs(a) // abs(b)
    return q if (a < 0) == (b < 0) else -q


def clamp16(v):
    return max(-32768, min(32767, v))


def _put16(block, at, v):
    v &= 0xFFFF
    block[at] = v & 255
    block[at + 1] = v >> 8


def ima4_block_bytes(spb, ch):
    return ((spb - 1) // 2 + 4) * ch


def ms_block_bytes(spb, ch):
    return ((spb - 2) // 2 + 7) * ch


def encode_ima4(pcm, spb):
    """pcm: int16 [frames, channels] -> (bytes as uint8 array, decoded int16 [frames, channels])"""
    pcm = np.atleast_2d(np.asarray(pcm, np.int64).T).T
    n, ch = pcm.shape
    assert spb % 8 == 1 and spb > 1
    nblocks = (n + spb - 1) // spb
    x = np.zeros((nblocks * spb, ch), np.int64)
    x[:n] = pcm
    bb = ima4_block_bytes(spb, ch)
    out = np.zeros(nblocks * bb, np.uint8)
    dec = np.zeros_like(x)
    index = [0] * ch
    for b in range(nblocks):
        block = out[b * bb:(b + 1) * bb]
        for c in range(ch):
            sample = int(x[b * spb, c])
            _put16(block, c * 4, sample)
            _put16(block, c * 4 + 2, index[c])
            dec[b * spb, c] = sample
            idx = index[c]
            base = (ch + c) * 4
            for k in range(spb - 1):
                step = IMA_STEP[idx]
                diff = int(x[b * spb + k + 1, c]) - sample
                code = 0
                if diff < 0:
                    code, diff = 8, -diff
                if diff >= step:
                    code |= 4
                    diff -= step
                if diff >= step // 2:
                    code |= 2
                    diff -= step // 2
                if diff >= step // 4:
                    code |= 1
                sample = clamp16(sample + cdiv(IMA_CODEWORD[code] * step, 8))
                idx = max(0, min(88, idx + IMA_ADJUST[code]))
                dec[b * spb + k + 1, c] = sample
                at = base + ((k >> 1) & ~3) * ch + ((k >> 1) & 3)
                block[at] |= code << ((k & 1) * 4)
            index[c] = idx
    return out, dec[:n].astype(np.int16)


def encode_msadpcm(pcm, spb):
    pcm = np.atleast_2d(np.asarray(pcm, np.int64).T).T
    n, ch = pcm.shape
    assert spb % 2 == 0 and spb > 2
    nblocks = (n + spb - 1) // spb
    x = np.zeros((nblocks * spb, ch), np.int64)
    x[:n] = pcm
    bb = ms_block_bytes(spb, ch)
    out = np.zeros(nblocks * bb, np.uint8)
    dec = np.zeros_like(x)
    scale_keep = [16] * ch
    for b in range(nblocks):
        block = out[b * bb:(b + 1) * bb]
        for c in range(ch):
            pred = (b + 3 * c) % 7
            c0, c1 = MS_COEFF[pred]
            scale = scale_keep[c]
            h1, h0 = int(x[b * spb, c]), int(x[b * spb + 1, c])
            block[c] = pred
            _put16(block, ch + 2 * c, scale)
            _put16(block, 3 * ch + 2 * c, h0)
            _put16(block, 5 * ch + 2 * c, h1)
            dec[b * spb, c], dec[b * spb + 1, c] = h1, h0
            for k in range(spb - 2):
                p = cdiv(h0 * c0 + h1 * c1, 256)
                err = int(x[b * spb + 2 + k, c]) - p
                nib = max(-8, min(7, int(round(err / scale))))
                sample = clamp16(nib * scale + p)
                h1, h0 = h0, sample
                scale = max(cdiv(MS_ADAPTION[nib & 15] * scale, 256), 16)
                dec[b * spb + 2 + k, c] = sample
                at = c + k * ch
                block[7 * ch + (at >> 1)] |= (nib & 15) << (((at & 1) ^ 1) * 4)
            scale_keep[c] = min(scale, 32767)
    return out, dec[:n].astype(np.int16)


def test_signal(n, ch, seed):
    """a band-limited-ish int16 signal the codecs can follow (sines + a little noise)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n)[:, None]
    f = rng.uniform(0.002, 0.05, (1, ch))
    x = 9000 * np.sin(2 * np.pi * f * t) + 5000 * np.sin(2 * np.pi * 3.1 * f * t + 1.0) + rng.normal(0, 300, (n, ch))
    return np.clip(np.round(x), -32768, 32767).astype(np.int16)
