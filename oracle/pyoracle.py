"""ctypes/numpy bindings for the test oracle (oracle/liboracle.so) and, in the dev container,
for the partial build of the real reference (oracle/_ref/libfun_ofdm_ref.so).

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module; nothing under fun_ofdm_amd/ does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.environ.get("FUN_OFDM_REF", "/root/reference")

c64 = np.dtype(np.complex128)
tagged_sample = np.dtype([("sample", np.complex128), ("tag", np.int32), ("_pad", np.int32)])
tagged_vec64 = np.dtype([("samples", np.complex128, (64,)), ("tag", np.int32), ("_pad", np.int32)])
tagged_vec48 = np.dtype([("samples", np.complex128, (48,)), ("tag", np.int32), ("_pad", np.int32)])
frame_desc = np.dtype([("lts1_pos", np.int64), ("rot_start", np.int64), ("c", np.float64), ("s", np.float64),
                       ("c_prev", np.float64), ("s_prev", np.float64)])
frame_result = np.dtype([("status", np.int32), ("rate", np.int32), ("length", np.int32), ("num_symbols", np.int32)])

NONE, STS_START, STS_END, LTS_START, LTS1, LTS2, START_OF_FRAME = range(7)
ST_OK, ST_HEADER_FAIL, ST_CRC_FAIL, ST_TRUNCATED = range(4)
ST_SUPERSEDED = 5
NUM_RATES = 11
STANDARD_RATES = (0, 2, 3, 5, 6, 8, 9, 10)   # the eight 802.11a rates in the reference's enum


def build(ref=True):
    """(Re)build liboracle.so and, if the reference tree is present, oracle/_ref."""
    target = "all" if (ref and os.path.exists(os.path.join(REF_DIR, "src", "viterbi.cpp"))) else "oracle"
    subprocess.run(["make", "-s", "-C", HERE, target, "REF=" + REF_DIR], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "fo_oracle.c")):
            build(ref=False)
        L = C.CDLL(path)
        vp, sz, i32, i64, dbl = C.c_void_p, C.c_size_t, C.c_int, C.c_int64, C.c_double
        sigs = {
            "fo_rate_params_get": (i32, [i32, vp]), "fo_rate_from_field": (i32, [i32]), "fo_num_symbols": (i32, [i32, i32]),
            "fo_demod_scale": (dbl, [i32]),
            "fo_preamble_samples": (vp, []), "fo_lts_freq_domain": (vp, []), "fo_lts_time_domain_conj": (vp, []),
            "fo_polarity": (vp, []), "fo_data_subcarriers": (vp, []), "fo_pilot_subcarriers": (vp, []),
            "fo_parity": (i32, [C.c_uint]), "fo_crc32": (C.c_uint32, [vp, sz]), "fo_scramble": (None, [vp, vp, sz]),
            "fo_conv_encode": (None, [vp, vp, i32]), "fo_conv_decode": (None, [vp, vp, i32]),
            "fo_viterbi_forward": (None, [vp, i32, vp, vp, vp]), "fo_viterbi_chainback": (None, [vp, vp, i32]),
            "fo_puncture": (sz, [vp, sz, i32, vp]), "fo_depuncture": (sz, [vp, sz, i32, vp]),
            "fo_interleave": (None, [vp, sz, vp]), "fo_deinterleave": (None, [vp, sz, vp]),
            "fo_modulate": (sz, [vp, sz, i32, vp]), "fo_demodulate": (sz, [vp, sz, i32, vp]),
            "fo_encode_header": (None, [i32, i32, vp]), "fo_decode_header": (i32, [vp, vp, vp, vp]),
            "fo_encode_data": (sz, [vp, i32, i32, vp]), "fo_decode_data": (i32, [vp, i32, i32, vp, vp, vp]),
            "fo_symbol_map": (sz, [vp, sz, vp]), "fo_ifft64": (None, [vp]), "fo_fft64": (None, [vp]),
            "fo_frame_samples": (sz, [i32, i32]), "fo_build_frame": (sz, [vp, i32, i32, vp]),
            "fo_frame_detector_new": (vp, []), "fo_frame_detector_free": (None, [vp]), "fo_frame_detector_work": (None, [vp, vp, sz, vp]),
            "fo_timing_sync_new": (vp, []), "fo_timing_sync_free": (None, [vp]), "fo_timing_sync_work": (None, [vp, vp, sz, vp]),
            "fo_timing_sync_phase_acc": (dbl, [vp]),
            "fo_fft_symbols_new": (vp, []), "fo_fft_symbols_free": (None, [vp]), "fo_fft_symbols_work": (sz, [vp, vp, sz, vp]),
            "fo_channel_est_new": (vp, []), "fo_channel_est_free": (None, [vp]), "fo_channel_est_work": (sz, [vp, vp, sz, vp]),
            "fo_channel_est_state": (vp, [vp]),
            "fo_phase_tracker_new": (vp, []), "fo_phase_tracker_free": (None, [vp]), "fo_phase_tracker_work": (None, [vp, vp, sz, vp]),
            "fo_payloads_new": (vp, []), "fo_payloads_free": (None, [vp]), "fo_payloads_count": (sz, [vp]),
            "fo_payloads_len": (sz, [vp, sz]), "fo_payloads_data": (vp, [vp, sz]),
            "fo_frame_decoder_new": (vp, []), "fo_frame_decoder_free": (None, [vp]), "fo_frame_decoder_work": (None, [vp, vp, sz, vp]),
            "fo_frame_decoder_stats": (vp, [vp]),
            "fo_receiver_chain_new": (vp, []), "fo_receiver_chain_new_threaded": (vp, []), "fo_receiver_chain_free": (None, [vp]),
            "fo_receiver_chain_process_samples": (vp, [vp, vp, sz]), "fo_receiver_chain_decoder_stats": (vp, [vp]),
            "fo_chain_from_tags_f32": (None, [vp, i64, vp, sz, vp]), "fo_decode_batch_v2_f32": (None, [vp, i64, vp, vp, sz, sz, vp, sz, vp]),
            "fo_decode_batch_v2_f64": (None, [vp, i64, vp, vp, sz, sz, vp, sz, vp]),
            "fo_viterbi_forward_simd": (None, [vp, i32, vp, vp]), "fo_viterbi_simd_kind": (C.c_char_p, []), "fo_set_timed_simd_viterbi": (None, [i32]),
            "fo_pool_new": (vp, [i32]), "fo_pool_free": (None, [vp]), "fo_pool_threads": (i32, [vp]),
            "fo_pool_decode": (None, [vp, vp, vp, vp, sz, vp, sz, vp]),
            "fo_decode_alignment_f32": (None, [vp, i64, vp, vp, vp, vp, vp, vp, vp]),
            "fo_find_alignments_f32": (sz, [vp, i64, vp, sz]),
            "fo_decode_batch_f32": (None, [vp, i64, vp, vp, sz, vp, sz, vp, i32]),
        }
        for name, (res, args) in sigs.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _table(fn, dtype, n):
    addr = getattr(lib(), fn)()
    return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_byte * (np.dtype(dtype).itemsize * n))).contents).view(dtype).copy()


def rate_params(rate):
    out = np.zeros(6, np.int32)
    if lib().fo_rate_params_get(rate, _ptr(out)) != 0:
        raise ValueError("bad rate %r" % (rate,))
    return dict(rate_field=int(out[0]), cbps=int(out[1]), dbps=int(out[2]), bpsc=int(out[3]), rate=int(out[4]), punct=int(out[5]))


def num_symbols(rate, length):
    return lib().fo_num_symbols(rate, length)


def frame_samples(rate, length):
    return lib().fo_frame_samples(rate, length)


def preamble_samples():
    return _table("fo_preamble_samples", np.complex128, 320)


def lts_freq_domain():
    return _table("fo_lts_freq_domain", np.complex128, 64)


def lts_time_domain_conj():
    return _table("fo_lts_time_domain_conj", np.complex128, 64)


def polarity():
    return _table("fo_polarity", np.float64, 127)


def data_subcarriers():
    return _table("fo_data_subcarriers", np.int32, 48)


def crc32(data):
    d = np.ascontiguousarray(data, np.uint8)
    return int(lib().fo_crc32(_ptr(d), d.size))


def scramble(data):
    d = np.ascontiguousarray(data, np.uint8)
    o = np.empty_like(d)
    lib().fo_scramble(_ptr(d), _ptr(o), d.size)
    return o


def conv_encode(data, data_bits):
    d = np.ascontiguousarray(data, np.uint8)
    o = np.zeros(2 * (data_bits + 6), np.uint8)
    lib().fo_conv_encode(_ptr(d), _ptr(o), data_bits)
    return o


def conv_decode(symbols, data_bits):
    s = np.ascontiguousarray(symbols, np.uint8)
    assert s.size >= 2 * (data_bits + 6)
    o = np.zeros((data_bits + 7) // 8, np.uint8)
    lib().fo_conv_decode(_ptr(s), _ptr(o), data_bits)
    return o


def viterbi_forward(symbols, nsteps):
    s = np.ascontiguousarray(symbols, np.uint8)
    dec = np.zeros(nsteps, np.uint64)
    met = np.zeros(64, np.uint8)
    stats = np.zeros(2, np.uint64)
    lib().fo_viterbi_forward(_ptr(s), nsteps, _ptr(dec), _ptr(met), _ptr(stats))
    return dec, met, stats


def viterbi_forward_simd(symbols, nsteps):
    """The SIMD forward pass of the TIMED CPU baseline (fo_viterbi_forward_simd): must equal viterbi_forward bit for bit."""
    s = np.ascontiguousarray(symbols, np.uint8)
    dec = np.zeros(max(nsteps, 1), np.uint64)
    met = np.zeros(64, np.uint8)
    lib().fo_viterbi_forward_simd(_ptr(s), nsteps, _ptr(dec), _ptr(met))
    return dec[:nsteps], met


class Pool:
    """Pre-spawned workers with per-thread scratch: the batch decoder bench.py times as the CPU baseline (SIMD forward pass, no
    allocation per frame).  Same results as decode_batch_f32."""

    def __init__(self, threads):
        self.h = lib().fo_pool_new(int(threads))
        self.threads = int(lib().fo_pool_threads(self.h))

    def decode(self, iq, descs, ends, slot_bytes=4096):
        iq = np.ascontiguousarray(iq, np.complex64)
        descs = np.ascontiguousarray(descs, frame_desc)
        ends = np.ascontiguousarray(ends, np.int64)
        n = descs.size
        psdu = np.zeros((n, slot_bytes), np.uint8)
        res = np.zeros(n, frame_result)
        lib().fo_pool_decode(self.h, _ptr(iq), _ptr(descs), _ptr(ends), n, _ptr(psdu), slot_bytes, _ptr(res))
        return psdu, res

    def close(self):
        if getattr(self, "h", None):
            lib().fo_pool_free(self.h)
            self.h = None

    __del__ = close


def viterbi_chainback(decisions, data_bits):
    d = np.ascontiguousarray(decisions, np.uint64)
    o = np.zeros((data_bits + 7) // 8, np.uint8)
    lib().fo_viterbi_chainback(_ptr(d), _ptr(o), data_bits)
    return o


def _bytes_fn(name, data, rate, outsize):
    d = np.ascontiguousarray(data, np.uint8)
    o = np.zeros(outsize, np.uint8)
    n = getattr(lib(), name)(_ptr(d), d.size, rate, _ptr(o))
    return o[:n]


def puncture(data, rate):
    return _bytes_fn("fo_puncture", data, rate, len(data) + 8)


def depuncture(data, rate):
    return _bytes_fn("fo_depuncture", data, rate, 2 * len(data) + 8)


def interleave(data):
    d = np.ascontiguousarray(data, np.uint8)
    o = np.empty_like(d)
    lib().fo_interleave(_ptr(d), d.size, _ptr(o))
    return o


def deinterleave(data):
    d = np.ascontiguousarray(data, np.uint8)
    o = np.empty_like(d)
    lib().fo_deinterleave(_ptr(d), d.size, _ptr(o))
    return o


def modulate(bits, rate):
    b = np.ascontiguousarray(bits, np.uint8)
    o = np.zeros(b.size, np.complex128)
    n = lib().fo_modulate(_ptr(b), b.size, rate, _ptr(o))
    return o[:n]


def demodulate(carriers, rate):
    c = np.ascontiguousarray(carriers, np.complex128)
    o = np.zeros(c.size * 6, np.uint8)
    n = lib().fo_demodulate(_ptr(c), c.size, rate, _ptr(o))
    return o[:n]


def encode_header(rate, length):
    o = np.zeros(48, np.complex128)
    lib().fo_encode_header(rate, length, _ptr(o))
    return o


def decode_header(carriers48):
    c = np.ascontiguousarray(carriers48, np.complex128)
    r, l, n = C.c_int(-1), C.c_int(0), C.c_int(0)
    ok = lib().fo_decode_header(_ptr(c), C.byref(r), C.byref(l), C.byref(n))
    return (r.value, l.value, n.value) if ok else None


def encode_data(payload, rate):
    p = np.ascontiguousarray(payload, np.uint8)
    o = np.zeros(num_symbols(rate, p.size) * 48, np.complex128)
    n = lib().fo_encode_data(_ptr(p), p.size, rate, _ptr(o))
    assert n == o.size
    return o


def decode_data(carriers, rate, length, taps=False):
    c = np.ascontiguousarray(carriers, np.complex128)
    rp = rate_params(rate)
    nsym = num_symbols(rate, length)
    pay = np.zeros(max(length, 1), np.uint8)
    soft = np.zeros(2 * nsym * rp["dbps"], np.uint8)
    dec = np.zeros(nsym * rp["dbps"] // 8, np.uint8)
    ok = lib().fo_decode_data(_ptr(c), rate, length, _ptr(pay), _ptr(soft), _ptr(dec))
    pay = pay[:length] if ok else None
    return (pay, soft, dec) if taps else pay


def symbol_map(carriers):
    c = np.ascontiguousarray(carriers, np.complex128)
    o = np.zeros(c.size // 48 * 64, np.complex128)
    lib().fo_symbol_map(_ptr(c), c.size, _ptr(o))
    return o


def fft64(x):
    d = np.array(x, np.complex128)
    lib().fo_fft64(_ptr(d))
    return d


def ifft64(x):
    d = np.array(x, np.complex128)
    lib().fo_ifft64(_ptr(d))
    return d


def frame_from_coded_bits(inter, rate, length):
    """Samples of a frame from its interleaved coded bits: modulate, header, symbol_map, inverse DFT + cyclic prefix, preamble (oracle pieces,
    each pinned against the real one except the inverse DFT, fft.cpp:68-96, which is pinned by definition).  Used by the tests as well."""
    car = modulate(np.asarray(inter, np.uint8), rate)
    bins = symbol_map(np.concatenate([encode_header(rate, length), car]))
    td = np.concatenate([np.concatenate([ifft64(b)[48:], ifft64(b)]) for b in bins.reshape(-1, 64)])
    return np.concatenate([preamble_samples(), td])


def build_frame(payload, rate):
    p = np.ascontiguousarray(payload, np.uint8)
    o = np.zeros(frame_samples(rate, p.size), np.complex128)
    n = lib().fo_build_frame(_ptr(p), p.size, rate, _ptr(o))
    assert n == o.size
    return o


class _Block:
    _new = _free = None

    def __init__(self):
        self.h = getattr(lib(), self._new)()

    def __del__(self):
        if getattr(self, "h", None):
            getattr(lib(), self._free)(self.h)
            self.h = None


class FrameDetector(_Block):
    _new, _free = "fo_frame_detector_new", "fo_frame_detector_free"

    def work(self, samples):
        s = np.ascontiguousarray(samples, np.complex128)
        o = np.zeros(s.size, tagged_sample)
        lib().fo_frame_detector_work(self.h, _ptr(s), s.size, _ptr(o))
        return o


class TimingSync(_Block):
    _new, _free = "fo_timing_sync_new", "fo_timing_sync_free"

    def work(self, tagged):
        s = np.ascontiguousarray(tagged, tagged_sample)
        o = np.zeros(s.size, tagged_sample)
        lib().fo_timing_sync_work(self.h, _ptr(s), s.size, _ptr(o))
        return o

    @property
    def phase_acc(self):
        return lib().fo_timing_sync_phase_acc(self.h)


class FFTSymbols(_Block):
    _new, _free = "fo_fft_symbols_new", "fo_fft_symbols_free"

    def work(self, tagged):
        s = np.ascontiguousarray(tagged, tagged_sample)
        o = np.zeros(s.size // 64 + 4, tagged_vec64)
        n = lib().fo_fft_symbols_work(self.h, _ptr(s), s.size, _ptr(o))
        return o[:n]


class ChannelEst(_Block):
    _new, _free = "fo_channel_est_new", "fo_channel_est_free"

    def work(self, vecs):
        s = np.ascontiguousarray(vecs, tagged_vec64)
        o = np.zeros(max(s.size, 1), tagged_vec64)
        n = lib().fo_channel_est_work(self.h, _ptr(s), s.size, _ptr(o))
        return o[:n]

    @property
    def state(self):
        addr = lib().fo_channel_est_state(self.h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_double * 128)).contents).view(np.complex128).copy()


class PhaseTracker(_Block):
    _new, _free = "fo_phase_tracker_new", "fo_phase_tracker_free"

    def work(self, vecs):
        s = np.ascontiguousarray(vecs, tagged_vec64)
        o = np.zeros(s.size, tagged_vec48)
        lib().fo_phase_tracker_work(self.h, _ptr(s), s.size, _ptr(o))
        return o


def _payload_list(h):
    L = lib()
    out = []
    for i in range(L.fo_payloads_count(h)):
        n = L.fo_payloads_len(h, i)
        out.append(C.string_at(L.fo_payloads_data(h, i), n))
    return out


class FrameDecoder(_Block):
    _new, _free = "fo_frame_decoder_new", "fo_frame_decoder_free"

    def __init__(self):
        super().__init__()
        self.out = lib().fo_payloads_new()

    def work(self, vecs):
        s = np.ascontiguousarray(vecs, tagged_vec48)
        lib().fo_frame_decoder_work(self.h, _ptr(s), s.size, self.out)
        return _payload_list(self.out) if s.size else []

    @property
    def stats(self):
        addr = lib().fo_frame_decoder_stats(self.h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint64 * 4)).contents).copy()

    def __del__(self):
        if getattr(self, "out", None):
            lib().fo_payloads_free(self.out)
            self.out = None
        super().__del__()


class ReceiverChain:
    """receiver_chain.cpp:29-126 -- process_samples() returns the payloads that surface in that call."""

    def __init__(self, threaded=False):
        L = lib()
        self.h = L.fo_receiver_chain_new_threaded() if threaded else L.fo_receiver_chain_new()

    def process_samples(self, samples):
        s = np.ascontiguousarray(samples, np.complex128)
        return _payload_list(lib().fo_receiver_chain_process_samples(self.h, _ptr(s), s.size))

    def run_stream(self, samples, chunk=4096, flush_calls=8):
        """Feed a whole stream in `chunk`-sample calls plus zero chunks to flush the 5-call latency."""
        s = np.ascontiguousarray(samples, np.complex128)
        out = []
        for x in range(0, s.size, chunk):
            blk = s[x:x + chunk]
            if blk.size < chunk:
                blk = np.concatenate([blk, np.zeros(chunk - blk.size, np.complex128)])
            out += self.process_samples(blk)
        for _ in range(flush_calls):
            out += self.process_samples(np.zeros(chunk, np.complex128))
        return out

    @property
    def decoder_stats(self):
        addr = lib().fo_receiver_chain_decoder_stats(self.h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint64 * 4)).contents).copy()

    def __del__(self):
        if getattr(self, "h", None):
            lib().fo_receiver_chain_free(self.h)
            self.h = None


def decode_alignment_f32(iq, desc, end=None, taps=False):
    """iq: complex64 array (the stream); desc: frame_desc scalar/record."""
    iq = np.ascontiguousarray(iq, np.complex64)
    end = iq.size if end is None else int(end)
    d = np.array([desc], frame_desc) if not isinstance(desc, np.ndarray) else np.ascontiguousarray(desc.reshape(1), frame_desc)
    res = np.zeros(1, frame_result)
    psdu = np.zeros(4096, np.uint8)
    if taps:
        maxsym = max((end - int(d["lts1_pos"][0])) // 80 + 4, 4)
        hinv = np.zeros(64, np.complex128)
        eq = np.zeros(maxsym * 48, np.complex128)
        soft = np.zeros(maxsym * 432 + 64, np.uint8)
        fftout = np.zeros(maxsym * 64, np.complex128)
        lib().fo_decode_alignment_f32(_ptr(iq), end, _ptr(d), _ptr(psdu), _ptr(res), _ptr(hinv), _ptr(eq), _ptr(soft), _ptr(fftout))
        r = res[0]
        ns = int(r["num_symbols"])
        t = dict(hinv=hinv, eq=eq[:(1 + ns) * 48], fftout=fftout[:(3 + ns) * 64])
        if r["rate"] >= 0:
            t["soft"] = soft[:2 * ns * rate_params(int(r["rate"]))["dbps"]]
        return r, psdu[:int(r["length"])] if r["status"] == ST_OK else None, t
    lib().fo_decode_alignment_f32(_ptr(iq), end, _ptr(d), _ptr(psdu), _ptr(res), None, None, None, None)
    r = res[0]
    return r, psdu[:int(r["length"])] if r["status"] == ST_OK else None


def find_alignments_f32(iq, cap=None):
    iq = np.ascontiguousarray(iq, np.complex64)
    cap = cap or (iq.size // 400 + 16)
    out = np.zeros(cap, frame_desc)
    n = lib().fo_find_alignments_f32(_ptr(iq), iq.size, _ptr(out), cap)
    return out[:n]


def decode_batch_f32(iq, descs, ends, slot_bytes=4096, threads=1):
    iq = np.ascontiguousarray(iq, np.complex64)
    descs = np.ascontiguousarray(descs, frame_desc)
    ends = np.ascontiguousarray(ends, np.int64)
    n = descs.size
    psdu = np.zeros((n, slot_bytes), np.uint8)
    res = np.zeros(n, frame_result)
    lib().fo_decode_batch_f32(_ptr(iq), iq.size, _ptr(descs), _ptr(ends), n, _ptr(psdu), slot_bytes, _ptr(res), threads)
    return psdu, res


def chain_from_tags_f32(iq, descs):
    """fft_symbols .. frame_decoder over the tagged, rotated stream timing_sync would hand on for these alignments: payload list."""
    iq = np.ascontiguousarray(iq, np.complex64)
    descs = np.ascontiguousarray(descs, frame_desc)
    h = lib().fo_payloads_new()
    try:
        lib().fo_chain_from_tags_f32(_ptr(iq), iq.size, _ptr(descs), descs.size, h)
        return _payload_list(h)
    finally:
        lib().fo_payloads_free(h)


def decode_batch_v2_f32(iq, descs, ends=None, slot_bytes=4096, n_ctx=0):
    """The batch restatement with the partial-vector flush and frame_decoder's frame-in-progress logic (fo_decode_batch_v2_f32).
    ends: per alignment (None: the next alignment's lts1_pos, the stream's end for the last); n_ctx: the last n_ctx alignments of
    descs are context only.  -> (psdu, results) of the first descs.size - n_ctx alignments."""
    iq = np.ascontiguousarray(iq, np.complex64)
    descs = np.ascontiguousarray(descs, frame_desc)
    n = descs.size - n_ctx
    psdu = np.zeros((n, slot_bytes), np.uint8)
    res = np.zeros(n, frame_result)
    e = None if ends is None else np.ascontiguousarray(ends, np.int64)
    assert e is None or e.size == descs.size
    lib().fo_decode_batch_v2_f32(_ptr(iq), iq.size, _ptr(descs), None if e is None else _ptr(e), n, n_ctx, _ptr(psdu), slot_bytes, _ptr(res))
    return psdu, res


def decode_batch_v2_f64(iq_rotated, descs, ends=None, slot_bytes=4096, n_ctx=0):
    """fo_decode_batch_v2_f64: the same on complex128 samples timing_sync has rotated already (phasors of descs not applied)."""
    iq = np.ascontiguousarray(iq_rotated, np.complex128)
    descs = np.ascontiguousarray(descs, frame_desc)
    n = descs.size - n_ctx
    psdu = np.zeros((n, slot_bytes), np.uint8)
    res = np.zeros(n, frame_result)
    e = None if ends is None else np.ascontiguousarray(ends, np.int64)
    lib().fo_decode_batch_v2_f64(_ptr(iq), iq.size, _ptr(descs), None if e is None else _ptr(e), n, n_ctx, _ptr(psdu), slot_bytes, _ptr(res))
    return psdu, res


# ------------------------------------------------------------------------------------------------
# the real reference (dev container only)
# ------------------------------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "src", "viterbi.cpp"))


def ref():
    """ctypes handle on oracle/_ref/libfun_ofdm_ref.so (built from /root/reference in place)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libfun_ofdm_ref.so")
        if not os.path.exists(path):
            if not ref_available():
                raise RuntimeError("reference tree not present; oracle/_ref cannot be built here")
            build(ref=True)
        L = C.CDLL(path)
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
        sigs = {
            "ref_sizeof_tagged_sample": (i32, []), "ref_sizeof_tagged_vector64": (i32, []), "ref_sizeof_tagged_vector48": (i32, []),
            "ref_parity": (i32, [i32]), "ref_conv_encode": (None, [vp, vp, i32]), "ref_conv_decode": (None, [vp, vp, i32]),
            "ref_interleave": (None, [vp, sz, vp]), "ref_deinterleave": (None, [vp, sz, vp]),
            "ref_puncture": (sz, [vp, sz, i32, vp]), "ref_depuncture": (sz, [vp, sz, i32, vp]),
            "ref_modulate": (sz, [vp, sz, i32, vp]), "ref_demodulate": (sz, [vp, sz, i32, vp]),
            "ref_symbol_map": (sz, [vp, sz, vp]), "ref_rate_params": (None, [i32, vp, vp]), "ref_rate_from_field": (i32, [i32]),
            "ref_preamble_samples": (None, [vp]), "ref_lts_freq_domain": (None, [vp]), "ref_lts_time_domain_conj": (None, [vp]),
            "ref_frame_detector_new": (vp, []), "ref_frame_detector_free": (None, [vp]), "ref_frame_detector_work": (None, [vp, vp, sz, vp]),
            "ref_timing_sync_new": (vp, []), "ref_timing_sync_free": (None, [vp]), "ref_timing_sync_work": (None, [vp, vp, sz, vp]),
            "ref_channel_est_new": (vp, []), "ref_channel_est_free": (None, [vp]), "ref_channel_est_work": (sz, [vp, vp, sz, vp]),
            "ref_phase_tracker_new": (vp, []), "ref_phase_tracker_free": (None, [vp]), "ref_phase_tracker_work": (sz, [vp, vp, sz, vp]),
        }
        for name, (res, args) in sigs.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _ref = L
    return _ref


class Ref:
    """Thin numpy wrappers over the real reference functions (same call shapes as the oracle's)."""

    @staticmethod
    def conv_encode(data, data_bits):
        d = np.ascontiguousarray(data, np.uint8)
        o = np.zeros(2 * (data_bits + 6), np.uint8)
        ref().ref_conv_encode(_ptr(d), _ptr(o), data_bits)
        return o

    @staticmethod
    def conv_decode(symbols, data_bits):
        s = np.ascontiguousarray(symbols, np.uint8)
        o = np.zeros((data_bits + 7) // 8 + 8, np.uint8)
        ref().ref_conv_decode(_ptr(s), _ptr(o), data_bits)
        return o[:(data_bits + 7) // 8]

    @staticmethod
    def _bytes(name, data, rate, outsize):
        d = np.ascontiguousarray(data, np.uint8)
        o = np.zeros(outsize, np.uint8)
        n = getattr(ref(), name)(_ptr(d), d.size, rate, _ptr(o))
        return o[:n]

    @staticmethod
    def puncture(data, rate):
        return Ref._bytes("ref_puncture", data, rate, len(data) + 8)

    @staticmethod
    def depuncture(data, rate):
        return Ref._bytes("ref_depuncture", data, rate, 2 * len(data) + 8)

    @staticmethod
    def interleave(data):
        d = np.ascontiguousarray(data, np.uint8)
        o = np.empty_like(d)
        ref().ref_interleave(_ptr(d), d.size, _ptr(o))
        return o

    @staticmethod
    def deinterleave(data):
        d = np.ascontiguousarray(data, np.uint8)
        o = np.empty_like(d)
        ref().ref_deinterleave(_ptr(d), d.size, _ptr(o))
        return o

    @staticmethod
    def modulate(bits, rate):
        b = np.ascontiguousarray(bits, np.uint8)
        o = np.zeros(b.size, np.complex128)
        n = ref().ref_modulate(_ptr(b), b.size, rate, _ptr(o))
        return o[:n]

    @staticmethod
    def demodulate(carriers, rate):
        c = np.ascontiguousarray(carriers, np.complex128)
        o = np.zeros(c.size * 6, np.uint8)
        n = ref().ref_demodulate(_ptr(c), c.size, rate, _ptr(o))
        return o[:n]

    @staticmethod
    def symbol_map(carriers):
        c = np.ascontiguousarray(carriers, np.complex128)
        o = np.zeros(c.size // 48 * 64, np.complex128)
        ref().ref_symbol_map(_ptr(c), c.size, _ptr(o))
        return o

    @staticmethod
    def rate_params(rate):
        out = np.zeros(5, np.int32)
        rel = C.c_double(0)
        ref().ref_rate_params(rate, _ptr(out), C.byref(rel))
        return dict(rate_field=int(out[0]), cbps=int(out[1]), dbps=int(out[2]), bpsc=int(out[3]), rate=int(out[4]), rel_rate=rel.value)

    @staticmethod
    def table(name, n):
        o = np.zeros(n, np.complex128)
        getattr(ref(), name)(_ptr(o))
        return o

    class Block:
        def __init__(self, kind):
            self.kind = kind
            self.h = getattr(ref(), "ref_%s_new" % kind)()

        def __del__(self):
            if getattr(self, "h", None):
                getattr(ref(), "ref_%s_free" % self.kind)(self.h)
                self.h = None

        def work(self, x):
            k = self.kind
            if k == "frame_detector":
                s = np.ascontiguousarray(x, np.complex128)
                o = np.zeros(s.size, tagged_sample)
                ref().ref_frame_detector_work(self.h, _ptr(s), s.size, _ptr(o))
                return o
            if k == "timing_sync":
                s = np.ascontiguousarray(x, tagged_sample)
                o = np.zeros(s.size, tagged_sample)
                ref().ref_timing_sync_work(self.h, _ptr(s), s.size, _ptr(o))
                return o
            if k == "channel_est":
                s = np.ascontiguousarray(x, tagged_vec64)
                o = np.zeros(max(s.size, 1), tagged_vec64)
                n = ref().ref_channel_est_work(self.h, _ptr(s), s.size, _ptr(o))
                return o[:n]
            if k == "phase_tracker":
                s = np.ascontiguousarray(x, tagged_vec64)
                o = np.zeros(max(s.size, 1), tagged_vec48)
                n = ref().ref_phase_tracker_work(self.h, _ptr(s), s.size, _ptr(o))
                return o[:n]
            raise ValueError(k)
