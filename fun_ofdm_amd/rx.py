"""Host-side handle on the gfx950 receive path (ctypes over include/fun_ofdm_amd.h)."""
import ctypes as C

import numpy as np

from ._lib import lib, check

# foa_frame_desc / foa_frame_result (include/fun_ofdm_amd.h)
frame_desc_dtype = np.dtype([("lts1_pos", np.int64), ("rot_start", np.int64), ("c", np.float64), ("s", np.float64),
                             ("c_prev", np.float64), ("s_prev", np.float64)])
frame_result_dtype = np.dtype([("status", np.int32), ("rate", np.int32), ("length", np.int32), ("num_symbols", np.int32)])

ST_OK, ST_HEADER_FAIL, ST_CRC_FAIL, ST_TRUNCATED, ST_NO_SPACE, ST_SUPERSEDED = range(6)

# fun::Rate (src/rates.h:31-44)
RATE_NAMES = ("1/2 BPSK", "2/3 BPSK", "3/4 BPSK", "1/2 QPSK", "2/3 QPSK", "3/4 QPSK", "1/2 QAM16", "2/3 QAM16", "3/4 QAM16",
              "2/3 QAM64", "3/4 QAM64")
RATE_MBPS = (6, 8, 9, 12, 16, 18, 24, 32, 36, 48, 54)
STANDARD_RATES = (0, 2, 3, 5, 6, 8, 9, 10)


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


class Receiver:
    """One receiver handle = one HIP stream on one device."""

    def __init__(self, device=0):
        self._lib = lib()
        self._h = C.c_void_p()
        self._check(self._lib.foa_rx_create(C.byref(self._h), int(device)))
        self.device = int(device)
        if self.notes():
            import warnings
            warnings.warn("fun_ofdm_amd: " + self.notes())

    def _check(self, rc):
        check(rc, self._lib)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.foa_rx_destroy(self._h)
            self._h = None

    __del__ = close

    def _after_torch(self, t):
        """The library's streams know nothing of torch's (and run at other priorities): before a call that reads or writes torch tensors,
        whatever torch's CURRENT stream on the tensor's device still has queued -- the copy that made an input, the fill of an output
        tensor -- must be through.  Ordered on the DEVICE (foa_rx_after with an event recorded on that stream): the host does not wait.
        Work the caller has queued on OTHER torch streams (a side stream, a non_blocking copy stream) is not covered: record an event
        there and hand it to after() yourself."""
        import torch
        s = torch.cuda.current_stream(t.device)
        if s.query():
            return                                       # idle (the steady state of a loop that orders by other means): nothing to wait for
        ev = torch.cuda.Event()
        ev.record(s)
        self._check(self._lib.foa_rx_after(self._h, ev.cuda_event))
        self._after_keep = ev                            # (alive until the call below has queued its wait)

    def after(self, event):
        """foa_rx_after: the next call that queues device work waits (on the device) for `event`, a torch.cuda.Event that has been recorded."""
        self._check(self._lib.foa_rx_after(self._h, event.cuda_event))
        self._after_keep = event

    def record_consumed(self, event):
        """foa_rx_record_consumed: `event` (a torch.cuda.Event that has been recorded once, so that its handle exists) completes when everything
        queued so far has read its input buffers."""
        self._check(self._lib.foa_rx_record_consumed(self._h, event.cuda_event))

    def record_done(self, event):
        """foa_rx_record_done: `event` completes when everything queued so far is complete (outputs final)."""
        self._check(self._lib.foa_rx_record_done(self._h, event.cuda_event))

    def torch_waits_for_done(self, device=None):
        """Make torch's current stream wait, on the device, for everything queued on this handle so far (foa_rx_record_done): torch work queued
        afterwards on that stream sees the final PSDUs and results; the host does not wait."""
        import torch
        s = torch.cuda.current_stream(device if device is not None else self.device)
        ev = torch.cuda.Event()
        ev.record(s)                                     # (creates the event's handle; the library's record below supersedes this one)
        self.record_done(ev)
        s.wait_event(ev)

    def set_option(self, name, value):
        self._check(self._lib.foa_rx_set_option(self._h, name.encode(), int(value)))

    def notes(self):
        """Non-fatal remarks about the handle's set-up (e.g. too few hardware queues for four lanes); "" if none."""
        return (self._lib.foa_rx_notes(self._h) or b"").decode()

    def reserve(self, n_samples, n_frames):
        self._check(self._lib.foa_rx_reserve(self._h, int(n_samples), int(n_frames)))

    def sync(self):
        self._check(self._lib.foa_rx_sync(self._h))

    # ---- batch decode, host buffers -------------------------------------------------------------
    def decode_frames_host(self, iq, descs, ends, slot_bytes=4096, psdu_out=None):
        """iq: complex64[n]; descs: frame_desc_dtype[m]; ends: int64[m] -> (psdu uint8[m, slot], results[m]).
        psdu_out: optional preallocated uint8[m, slot] to receive the PSDUs."""
        iq = np.ascontiguousarray(iq, np.complex64)
        descs = np.ascontiguousarray(descs, frame_desc_dtype)
        ends = np.ascontiguousarray(ends, np.int64)
        m = descs.size
        psdu = np.zeros((m, slot_bytes), np.uint8) if psdu_out is None else psdu_out
        assert psdu.shape == (m, slot_bytes) and psdu.dtype == np.uint8 and psdu.flags.c_contiguous
        res = np.zeros(m, frame_result_dtype)
        self._check(self._lib.foa_rx_decode_frames_host(self._h, _vp(iq), iq.size, _vp(descs), _vp(ends), m, _vp(psdu), slot_bytes, _vp(res)))
        return psdu, res

    def decode_frames_f64_host(self, iq_rotated, descs, ends, slot_bytes=4096):
        """foa_rx_decode_frames_f64_host: complex128 samples that timing_sync has rotated already (the descriptors' phasors are not applied)."""
        iq = np.ascontiguousarray(iq_rotated, np.complex128)
        descs = np.ascontiguousarray(descs, frame_desc_dtype)
        ends = np.ascontiguousarray(ends, np.int64)
        m = descs.size
        psdu = np.zeros((m, slot_bytes), np.uint8)
        res = np.zeros(m, frame_result_dtype)
        self._check(self._lib.foa_rx_decode_frames_f64_host(self._h, _vp(iq), iq.size, _vp(descs), _vp(ends), m, _vp(psdu), slot_bytes, _vp(res)))
        return psdu, res

    def submit_host(self, iq, descs, ends, slot_bytes=4096, n_context=0):
        """Asynchronous decode_frames_host: returns a ticket (foa_rx_submit_host_ctx).  n_context: the last n_context alignments of
        descs / ends are context only (looked at, not decoded; no results)."""
        iq = np.ascontiguousarray(iq, np.complex64)
        descs = np.ascontiguousarray(descs, frame_desc_dtype)
        ends = np.ascontiguousarray(ends, np.int64)
        assert ends.size == descs.size and 0 <= n_context < max(descs.size, 1)
        t = C.c_uint64(0)
        m = descs.size - n_context
        self._check(self._lib.foa_rx_submit_host_ctx(self._h, _vp(iq), iq.size, _vp(descs), _vp(ends), m, n_context, slot_bytes, C.byref(t)))
        return (int(t.value), m, slot_bytes)

    def collect(self, ticket, wait=True):
        """-> (psdu, results) of a submit_host ticket, or None if wait=False and it is not complete yet."""
        t, m, slot_bytes = ticket
        psdu = np.zeros((m, slot_bytes), np.uint8)
        res = np.zeros(m, frame_result_dtype)
        rc = self._lib.foa_rx_collect(self._h, t, 1 if wait else 0, _vp(psdu), _vp(res))
        if rc < 0:
            self._check(rc)
        return (psdu, res) if rc == 1 else None

    # ---- batch decode, device buffers (torch tensors on this device) ------------------------------
    def decode_frames_dev(self, iq, descs, ends, psdu, results, n_context=0, n_lead=0, settle=True):
        """All arguments are CUDA(HIP) torch tensors already resident in HBM:
        iq complex64[n] (or float32[n,2]); descs uint8[(n_lead+m+n_context)*48] (frame_desc_dtype bytes); ends int64[n_lead+m+n_context];
        psdu uint8[m, slot]; results int32[m, 4].  n_lead: the first alignments of descs / ends were decided by an earlier piece
        (looked at only for where they sit); n_context: the last ones are context only (foa_rx_decode_frames_lead_ctx_dev).
        Asynchronous on the handle's streams, which are NOT ordered against
        torch's (and run at other priorities: a fill torch has queued for an output tensor may land after the kernels
        that write it), so the wrapper makes the call wait, on the device, for what torch's current stream has queued
        (_after_torch -> foa_rx_after).  settle=False: the
        caller orders torch's work on these tensors against the call itself (bench.py's loop does, with one event per output set)."""
        if settle:
            self._after_torch(iq)
        n = iq.numel() if iq.is_complex() else iq.numel() // 2
        m = ends.numel() - n_context - n_lead
        assert descs.numel() * descs.element_size() == ends.numel() * frame_desc_dtype.itemsize
        assert psdu.shape[0] == m and results.numel() == 4 * m
        self._check(self._lib.foa_rx_decode_frames_lead_ctx_dev(self._h, iq.data_ptr(), n, descs.data_ptr(), ends.data_ptr(), n_lead, m, n_context, psdu.data_ptr(),
                                                      psdu.shape[1], results.data_ptr()))

    def sync_dev(self, iq, descs, ends):
        """Device-side frame_detector + timing_sync: iq complex64[n] (CUDA tensor), descs uint8[cap*48], ends int64[cap]
        (CUDA tensors, filled in place).  Returns the number of alignments found."""
        n = iq.numel() if iq.is_complex() else iq.numel() // 2
        cap = ends.numel()
        assert descs.numel() * descs.element_size() >= cap * frame_desc_dtype.itemsize
        self._after_torch(iq)
        got = C.c_size_t(0)
        self._check(self._lib.foa_rx_sync_dev(self._h, iq.data_ptr(), n, descs.data_ptr(), ends.data_ptr(), cap, C.byref(got)))
        return int(got.value)

    def sync_dev_begin(self, iq, descs, ends):
        """Queue the device pre-sync (foa_rx_sync_dev_begin); sync_dev_end() waits for it and returns the count."""
        n = iq.numel() if iq.is_complex() else iq.numel() // 2
        cap = ends.numel()
        assert descs.numel() * descs.element_size() >= cap * frame_desc_dtype.itemsize
        self._after_torch(iq)
        self._check(self._lib.foa_rx_sync_dev_begin(self._h, iq.data_ptr(), n, descs.data_ptr(), ends.data_ptr(), cap))

    def sync_dev_end(self):
        got = C.c_size_t(0)
        self._check(self._lib.foa_rx_sync_dev_end(self._h, C.byref(got)))
        return int(got.value)

    def tx_frame_samples(self, length, rate):
        n = C.c_size_t(0)
        self._check(self._lib.foa_tx_build_frames_dev(self._h, None, 0, int(length), int(rate), 0, None, C.byref(n)))
        return int(n.value)

    def tx_build_frames(self, payloads, rate):
        """frame_builder::build_frame on the device: payloads uint8[n, L] (CUDA tensor) -> float64[n, S, 2] (CUDA tensor)."""
        import torch
        assert payloads.dtype == torch.uint8 and payloads.is_cuda and payloads.dim() == 2 and payloads.stride(1) == 1
        n, length = payloads.shape
        s = self.tx_frame_samples(length, rate)
        out = torch.empty((n, s, 2), dtype=torch.float64, device=payloads.device)
        got = C.c_size_t(0)
        self._after_torch(payloads)
        self._check(self._lib.foa_tx_build_frames_dev(self._h, payloads.data_ptr(), payloads.stride(0), int(length), int(rate), n, out.data_ptr(), C.byref(got)))
        self.sync()
        return out

    def tx_channel(self, frames, pitch, lead, snr_db, seed, cfo_hz=0.0):
        """Synthetic channel on the device: frames float64[n, S, 2] -> float32[n * pitch, 2] (CUDA tensors)."""
        import torch
        n, s, _ = frames.shape
        iq = torch.empty((n * pitch, 2), dtype=torch.float32, device=frames.device)
        self._after_torch(frames)
        self._check(self._lib.foa_tx_channel_dev(self._h, frames.data_ptr(), n, s, int(pitch), int(lead), float(snr_db), float(cfo_hz), int(seed), iq.data_ptr()))
        self.sync()
        return iq

    def wait_previous(self):
        """Block until the decode call before the most recent one is complete (see foa_rx_wait_previous)."""
        self._check(self._lib.foa_rx_wait_previous(self._h))

    def wait_age(self, age):
        """Block until the decode call `age` calls back is complete (see foa_rx_wait_age)."""
        self._check(self._lib.foa_rx_wait_age(self._h, int(age)))

    def forward_spacing(self, age=2):
        """foa_rx_forward_spacing_ms: (start-to-start, overlap with the pass before, own duration) in ms of the forward pass `age` calls back."""
        out = (C.c_float * 3)()
        self._check(self._lib.foa_rx_forward_spacing_ms(self._h, int(age), out))
        return float(out[0]), float(out[1]), float(out[2])

    def probe_issue(self):
        """Live issue-rate probe (foa_rx_probe_issue): {"pk_u16": {clk_per_wave_instr, ghz, wave_instr_per_s}, "vop2_u32": {...}}."""
        out = (C.c_double * 6)()
        self._check(self._lib.foa_rx_probe_issue(self._h, out))
        keys = ("clk_per_wave_instr", "ghz", "wave_instr_per_s")
        return {"pk_u16": dict(zip(keys, out[0:3])), "vop2_u32": dict(zip(keys, out[3:6]))}

    def probe_h2d(self, piece_bytes=1 << 28, in_flight=4, rounds=2):
        """foa_rx_probe_h2d: GB/s host to device the way the stream engines copy (page-locked staging, several pieces in flight)."""
        out = C.c_double(0.0)
        self._check(self._lib.foa_rx_probe_h2d(self._h, int(piece_bytes), int(in_flight), int(rounds), C.byref(out)))
        return float(out.value)

    def kernel_ms(self, previous=False, age=None):
        """HIP-event durations of the last decode (previous=True: of the one before it; age=2: of the one before that,
        which is certainly complete in a pipelined sequence of calls) in ms: header, scan, symbols, viterbi_fwd,
        viterbi_finish, total."""
        out = (C.c_float * 6)()
        if age is not None:
            self._check(self._lib.foa_rx_kernel_ms_age(self._h, int(age), out))
        else:
            self._check((self._lib.foa_rx_prev_kernel_ms if previous else self._lib.foa_rx_last_kernel_ms)(self._h, out))
        return dict(zip(("header", "scan", "symbols", "viterbi_fwd", "viterbi_finish", "total"), (float(x) for x in out)))

    def taps(self, n_frames, eq=False, soft=True, cap_symbols=None):
        """Intermediates of the last decode call (host copies): dict(hinv, eq, eq_off, soft, soft_off)."""
        cap_symbols = cap_symbols or 1400 * n_frames
        hinv = np.zeros((n_frames, 64), np.complex128)
        eq_buf = np.zeros((cap_symbols + n_frames) * 48, np.complex128) if eq else None
        eq_off = np.zeros(n_frames + 1, np.uint64)
        soft_buf = np.zeros(cap_symbols * 432, np.uint8) if soft else None
        soft_off = np.zeros(n_frames + 1, np.uint64)
        self._check(self._lib.foa_rx_get_taps(self._h, n_frames, _vp(hinv), _vp(eq_buf) if eq else None, eq_buf.size if eq else 0, _vp(eq_off),
                                    _vp(soft_buf) if soft else None, soft_buf.size if soft else 0, _vp(soft_off)))
        return dict(hinv=hinv, eq=eq_buf, eq_off=eq_off.astype(np.int64), soft=soft_buf, soft_off=soft_off.astype(np.int64))

    def decisions(self, frame, cap=40000):
        """Raw decision words of one frame of the last decode call (layout: see foa_rx_get_decisions)."""
        out = np.zeros(cap, np.uint64)
        n = C.c_size_t(0)
        self._check(self._lib.foa_rx_get_decisions(self._h, int(frame), _vp(out), cap, C.byref(n)))
        return out[:n.value]

    # ---- stage-level entry points ------------------------------------------------------------------
    def fft_forward(self, vectors):
        """fft::forward on [n,64] complex128 (src/fft.cpp:50-59)."""
        v = np.array(vectors, np.complex128).reshape(-1, 64)
        self._check(self._lib.foa_fft_forward_f64(self._h, _vp(v), v.shape[0]))
        return v

    def conv_decode(self, symbols, data_bits, n_blocks=1):
        """viterbi::conv_decode (src/viterbi.cpp:31-37) on n_blocks packed blocks."""
        s = np.ascontiguousarray(symbols, np.uint8)
        assert s.size >= n_blocks * 2 * (data_bits + 6)
        out = np.zeros((n_blocks, (data_bits + 7) // 8), np.uint8)
        self._check(self._lib.foa_conv_decode(self._h, _vp(s), _vp(out), int(data_bits), int(n_blocks)))
        return out


class Stream:
    """process_samples() entirely on the device (foa_stream_*): push samples, get the payloads of finished batches."""

    def __init__(self, receiver, batch_samples, narrow_threads=0):
        self._rx = receiver
        self._lib = receiver._lib
        self._h = C.c_void_p()
        self._check(self._lib.foa_stream_create(receiver._h, int(batch_samples), int(narrow_threads), C.byref(self._h)))

    def _check(self, rc):
        check(rc, self._lib)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.foa_stream_destroy(self._h)
            self._h = None

    __del__ = close

    def push(self, iq):
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            self._check(self._lib.foa_stream_push_f32(self._h, _vp(iq), iq.size))
        else:
            iq = iq.astype(np.complex128, copy=False)
            self._check(self._lib.foa_stream_push_f64(self._h, _vp(iq), iq.size))
        return self.take()

    def flush(self):
        self._check(self._lib.foa_stream_flush(self._h))
        return self.take(wait=True)

    def take(self, wait=False):
        """Payloads (bytes objects, stream order) of every batch that has finished (wait: of every batch submitted)."""
        out = []
        while True:
            n, nb = C.c_size_t(0), C.c_size_t(0)
            rc = self._lib.foa_stream_ready(self._h, 1 if wait else 0, C.byref(n), C.byref(nb))
            if rc < 0:
                self._check(rc)
            if rc == 0:
                return out
            buf = np.zeros(max(nb.value, 1), np.uint8)
            lens = np.zeros(max(n.value, 1), np.uint32)
            self._check(self._lib.foa_stream_take(self._h, _vp(buf), _vp(lens)))
            o = 0
            for k in range(n.value):
                out.append(buf[o:o + int(lens[k])].tobytes())
                o += int(lens[k])

    def stats(self):
        a = np.zeros(8, np.uint64)
        self._check(self._lib.foa_stream_stats(self._h, _vp(a)))
        return dict(ok=int(a[0]), header_fail=int(a[1]), crc_fail=int(a[2]), truncated=int(a[3]), no_space=int(a[4]), alignments=int(a[5]),
                    batches=int(a[6]), samples=int(a[7]))       # (truncated: FOA_ST_TRUNCATED + FOA_ST_SUPERSEDED)


class Shard(Stream):
    """The stream engine over several devices (foa_shard_*): batch k of the stream on device devices[k mod n]; payloads in stream order.
    A device may be listed more than once (two handles sharing it)."""

    def __init__(self, devices, batch_samples, narrow_threads=0):
        self._lib = lib()
        self._h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self._check(self._lib.foa_shard_create(devs, len(devices), int(batch_samples), int(narrow_threads), C.byref(self._h)))
        self.n_devices = len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.foa_shard_destroy(self._h)
            self._h = None

    __del__ = close

    def push(self, iq):
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            self._check(self._lib.foa_shard_push_f32(self._h, _vp(iq), iq.size))
        else:
            iq = iq.astype(np.complex128, copy=False)
            self._check(self._lib.foa_shard_push_f64(self._h, _vp(iq), iq.size))
        return self.take()

    def flush(self):
        self._check(self._lib.foa_shard_flush(self._h))
        return self.take(wait=True)

    def take(self, wait=False):
        out = []
        while True:
            n, nb = C.c_size_t(0), C.c_size_t(0)
            rc = self._lib.foa_shard_ready(self._h, 1 if wait else 0, C.byref(n), C.byref(nb))
            if rc < 0:
                self._check(rc)
            if rc == 0:
                return out
            buf = np.zeros(max(nb.value, 1), np.uint8)
            lens = np.zeros(max(n.value, 1), np.uint32)
            self._check(self._lib.foa_shard_take(self._h, _vp(buf), _vp(lens)))
            o = 0
            for k in range(n.value):
                out.append(buf[o:o + int(lens[k])].tobytes())
                o += int(lens[k])

    def stats(self):
        a = np.zeros(8, np.uint64)
        per = np.zeros(self.n_devices, np.uint64)
        self._check(self._lib.foa_shard_stats(self._h, _vp(a), _vp(per), self.n_devices))
        return dict(ok=int(a[0]), header_fail=int(a[1]), crc_fail=int(a[2]), truncated=int(a[3]), no_space=int(a[4]), alignments=int(a[5]),
                    batches=int(a[6]), samples=int(a[7]), per_device_alignments=[int(x) for x in per])


class Sync:
    """Streaming frame_detector + timing_sync on the host (foa_sync_*): push raw samples, get alignment
    descriptors with stream-absolute positions."""

    def __init__(self, call=4096):
        """call: the reference receiver's call size by which timing_sync.cpp:99 is decided (foa_sync_set_call); 0 = as one call."""
        self._h = C.c_void_p()
        check(lib().foa_sync_create(C.byref(self._h)))
        if call != 4096:
            check(lib().foa_sync_set_call(self._h, int(call)))

    def close(self):
        if getattr(self, "_h", None):
            lib().foa_sync_destroy(self._h)
            self._h = None

    __del__ = close

    def push(self, iq):
        """iq: complex64 or complex128 array -> frame_desc_dtype[k] completed by this chunk."""
        iq = np.ascontiguousarray(iq)
        if iq.dtype == np.complex64:
            fn = lib().foa_sync_push_f32
        else:
            iq = iq.astype(np.complex128, copy=False)
            fn = lib().foa_sync_push_f64
        out, got = [], C.c_size_t(0)
        buf = np.zeros(max(iq.size // 400, 0) + 64, frame_desc_dtype)
        check(fn(self._h, _vp(iq), iq.size, _vp(buf), buf.size, C.byref(got)))
        out.append(buf[:got.value].copy())
        while got.value == buf.size:
            check(fn(self._h, None, 0, _vp(buf), buf.size, C.byref(got)))
            out.append(buf[:got.value].copy())
        return np.concatenate(out)

    @property
    def settled(self):
        return int(lib().foa_sync_settled(self._h))


def find_alignments(iq, flush=True, call=4096):
    """One-shot sync over a whole stream (plus 160 zeros so that the tail is examined)."""
    s = Sync(call)
    d = [s.push(iq)]
    if flush:
        z = np.zeros(4096, iq.dtype if iq.dtype in (np.complex64, np.complex128) else np.complex64)
        d.append(s.push(z))
    d = np.concatenate(d)
    s.close()
    return d[d["lts1_pos"] < iq.size]


def alignment_ends(descs, n_samples):
    """Exclusive end of each alignment's samples: the next alignment's LTS1, or the stream end."""
    e = np.empty(descs.size, np.int64)
    if descs.size:
        e[:-1] = descs["lts1_pos"][1:]
        e[-1] = n_samples
    return e
