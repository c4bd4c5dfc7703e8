"""Synthetic 20 MHz frame streams for tests and bench.py (host side, numpy).

A vectorised restatement of the reference's transmit path -- frame_builder::build_frame
(src/frame_builder.cpp:53-82) over ppdu::encode (src/ppdu.cpp:65-165), symbol_mapper::map
(src/symbol_mapper.cpp:81-119) and fft::inverse (src/fft.cpp:68-96) -- so that thousands of distinct
frames can be produced quickly.  It reproduces the reference's deviations from 802.11a:
generator masks {121,91} with the newest bit at the LSB, its puncturing patterns, the single 48-entry
interleaver map, the byte-wise LSB scrambler, the 12-bit length field, and the literal preamble table.
tests/test_synth.py checks it against the oracle's build_frame.
"""
import zlib

import numpy as np

# src/rates.h:52-196: rate_field, cbps, dbps, bpsc, puncture (0: 1/2, 1: 2/3, 2: 3/4)
RATES = ((0xD, 48, 24, 1, 0), (0xE, 48, 32, 1, 1), (0xF, 48, 36, 1, 2), (0x5, 96, 48, 2, 0), (0x6, 96, 64, 2, 1), (0x7, 96, 72, 2, 2),
         (0x9, 192, 96, 4, 0), (0xA, 192, 128, 4, 1), (0xB, 192, 144, 4, 2), (0x1, 288, 192, 6, 1), (0x3, 288, 216, 6, 2))
P_REF = 0.0124          # nominal in-frame power used to set the AWGN level

_DATA_IDX = np.array([i for i in range(6, 59) if i not in (11, 25, 32, 39, 53)])
_PILOT_IDX = np.array([11, 25, 39, 53])
_PILOT_SGN = np.array([1.0, 1.0, 1.0, -1.0])
_LTS = np.array([1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 0,
                 1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1], float)


def num_symbols(rate, length):
    return -(-(16 + 8 * (length + 4) + 6) // RATES[rate][2])


def frame_samples(rate, length):
    return 320 + 80 * (num_symbols(rate, length) + 1)


def _lfsr_bits(state, n):
    out = np.zeros(n, np.uint8)
    for i in range(n):
        fb = ((state >> 6) ^ (state >> 3)) & 1
        out[i] = fb
        state = ((state << 1) & 0x7E) | fb
    return out


_POLARITY = 1.0 - 2.0 * _lfsr_bits(0x7F, 127)      # symbol_mapper.cpp:33-43 = 802.11a p_0..126
_SCRAMBLE = _lfsr_bits(93, 127)                    # ppdu.cpp:141-147, period 127


def _r12(x):
    return np.array([float("%.12g" % v) for v in np.ravel(x)]).reshape(np.shape(x))


def preamble():
    """The 320 preamble samples as the literal table of src/preamble.h:24 holds them: 802.11a 17.3.3
    training symbols printed with 12 significant digits, [0] halved by the window and [160] = -0.078."""
    s = np.zeros(64, complex)
    for k, v in zip((-24, -20, -16, -12, -8, -4, 4, 8, 12, 16, 20, 24), (1, -1, 1, -1, -1, 1, -1, -1, 1, 1, 1, 1)):
        s[k % 64] = v * (1 + 1j) * np.sqrt(13.0 / 6.0)
    sts = np.fft.ifft(s)
    lf = np.zeros(64)
    lf[np.arange(-26, 27) % 64] = _LTS
    lts = np.fft.ifft(lf)
    p = np.concatenate([np.tile(sts[:16], 10), lts[32:], lts, lts])
    p = _r12(p.real) + 1j * _r12(p.imag)
    p[0] = _r12(sts[0].real / 2) + 1j * _r12(sts[0].imag / 2)
    p[160] = -0.078
    return p


_PREAMBLE = preamble()


def _conv_encode(bits):
    """viterbi.cpp:39-62 on [N, nbits] bit arrays -> [N, 2*nbits]."""
    n = bits.shape[1]
    pad = np.concatenate([np.zeros((bits.shape[0], 6), np.uint8), bits], axis=1)

    def d(k):
        return pad[:, 6 - k:6 - k + n]
    o0 = d(0) ^ d(3) ^ d(4) ^ d(5) ^ d(6)          # mask 121 = 0b1111001
    o1 = d(0) ^ d(1) ^ d(3) ^ d(4) ^ d(6)          # mask  91 = 0b1011011
    return np.stack([o0, o1], axis=2).reshape(bits.shape[0], 2 * n)


def _puncture(coded, punct):
    if punct == 0:
        return coded
    keep = (0, 1, 3, 5) if punct == 2 else (0, 2, 3)
    grp = 6 if punct == 2 else 4
    c = coded.reshape(coded.shape[0], -1, grp)
    return c[:, :, keep].reshape(coded.shape[0], -1)


def _interleave(x):
    k = np.arange(48)
    idx = 3 * (k % 16) + k // 16                    # interleaver.h:66-75 with (48,1)
    out = np.empty_like(x).reshape(x.shape[0], -1, 48)
    out[:, :, idx] = x.reshape(x.shape[0], -1, 48)
    return out.reshape(x.shape)


def _qam_axis(bits):
    """qam.h:83-97 on [..., nb] bits -> integer constellation coordinate."""
    pt = np.zeros(bits.shape[:-1], np.int64)
    flip = np.ones(bits.shape[:-1], np.int64)
    for i in range(bits.shape[-1]):
        b = bits[..., i].astype(np.int64) * 2 - 1
        pt = b * flip + pt * 2
        flip = flip * -b
    return pt


def _modulate(bits, rate):
    bpsc = RATES[rate][3]
    nb = 1 if bpsc == 1 else bpsc // 2
    power = 1.0 if bpsc == 1 else 0.5
    nn = 1 << (nb - 1)
    sf = np.sqrt(power * nn / ((4 * nn ** 3 - nn) // 3))
    b = bits.reshape(bits.shape[0], -1, bpsc)
    re = _qam_axis(b[:, :, :nb]) * sf
    im = _qam_axis(b[:, :, nb:]) * sf if bpsc > 1 else np.zeros_like(re)
    return re + 1j * im


def _header_carriers(rate, length):
    field = ((RATES[rate][0] & 0xF) << 13) | (length & 0xFFF)
    if bin(field).count("1") & 1:
        field |= 1 << 17
    field <<= 6
    bits = np.array([(field >> (23 - i)) & 1 for i in range(24)], np.uint8)[None, :]
    return _modulate(_interleave(_conv_encode(bits)), 0)[0]


def build_frames(payloads, rate):
    """payloads: uint8[N, L] -> complex128[N, 320 + 80*(nsym+1)] (frame_builder.cpp:53-82)."""
    payloads = np.ascontiguousarray(payloads, np.uint8)
    n, length = payloads.shape
    _, cbps, dbps, bpsc, punct = RATES[rate]
    nsym = num_symbols(rate, length)
    nbits = nsym * dbps
    nbytes = nbits // 8
    data = np.zeros((n, nbytes + 1), np.uint8)
    data[:, 2:2 + length] = payloads
    for i in range(n):                               # ppdu.cpp:133-137
        crc = zlib.crc32(data[i, :2 + length].tobytes())
        data[i, 2 + length:6 + length] = np.frombuffer(np.uint32(crc).tobytes(), np.uint8)
    data[:, :nbytes] ^= _SCRAMBLE[np.arange(nbytes) % 127]
    bits = np.unpackbits(data, axis=1)[:, :nbits]
    coded = _conv_encode(bits)
    car = _modulate(_interleave(_puncture(coded, punct)), rate)          # [n, nsym*48]
    car = np.concatenate([np.broadcast_to(_header_carriers(rate, length), (n, 48)), car], axis=1).reshape(n, nsym + 1, 48)
    bins = np.zeros((n, nsym + 1, 64), complex)
    bins[:, :, _DATA_IDX] = car
    pol = _POLARITY[np.arange(nsym + 1) % 127]
    bins[:, :, _PILOT_IDX] = _PILOT_SGN[None, None, :] * pol[None, :, None]
    td = np.fft.ifft(np.roll(bins, -32, axis=2), axis=2)                 # fft.cpp:77-80,94
    sym = np.concatenate([td[:, :, 48:], td], axis=2).reshape(n, (nsym + 1) * 80)
    return np.concatenate([np.broadcast_to(_PREAMBLE, (n, 320)), sym], axis=1)


def splitmix64_bytes(seed, n_frames, length, ids=None):
    """Deterministic payload bytes: frame i = splitmix64 stream seeded with seed + i (or seed + ids[i])."""
    words = (length + 7) // 8
    idx = np.arange(n_frames, dtype=np.uint64) if ids is None else np.asarray(ids, dtype=np.uint64)
    n_frames = idx.size
    state = (np.uint64(seed) + idx)[:, None] + \
        np.uint64(0x9E3779B97F4A7C15) * np.arange(1, words + 1, dtype=np.uint64)[None, :]
    z = state
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z.view(np.uint8).reshape(n_frames, words * 8)[:, :length].copy()


def make_stream(frames, pitch, lead, snr_db, seed, cfo_hz=None, phase=True):
    """Lay frames[N, S] out at a fixed pitch (frame i starts at i*pitch + lead), add a random carrier
    phase (and optional per-frame CFO), AWGN at snr_db relative to P_REF, round to complex64.
    Returns (iq complex64[N*pitch], frame_start int64[N])."""
    n, s = frames.shape
    assert lead + s <= pitch
    rng = np.random.default_rng(seed)
    out = np.zeros((n, pitch), np.complex64)
    ph = np.exp(1j * rng.uniform(0, 2 * np.pi, n)) if phase else np.ones(n)
    sig = frames * ph[:, None]
    if cfo_hz is not None:
        f = rng.uniform(-cfo_hz, cfo_hz, n)
        sig = sig * np.exp(2j * np.pi * f[:, None] * np.arange(s)[None, :] / 20e6)
    sigma = np.sqrt(P_REF / (2.0 * 10.0 ** (snr_db / 10.0)))
    chunk = max(1, (1 << 22) // pitch)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        noise = rng.normal(0.0, sigma, (b - a, pitch, 2)).astype(np.float32)
        out[a:b] = noise[..., 0] + 1j * noise[..., 1]
        out[a:b, lead:lead + s] += sig[a:b].astype(np.complex64)
    return out.reshape(-1), np.arange(n, dtype=np.int64) * pitch + lead
