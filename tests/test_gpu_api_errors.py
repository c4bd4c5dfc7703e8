"""The C ABI's error behaviour (include/fun_ofdm_amd.h: "every call returns 0 on success and a negative FOA_E_* code", text in
foa_last_error): NULL handles and pointers, out-of-range arguments, wrong call order, capacities too small -- every entry point
refuses them with a code and a message, nothing crashes, and the handle works afterwards.  GPU only."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

E_INVALID, E_STATE = -1, -5


def test_every_entry_point_refuses_bad_arguments(po):
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    L = foa.lib()
    rx = foa.Receiver(0)
    h = rx._h
    null = C.c_void_p(0)
    dev = torch.device("cuda", 0)
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
    p = C.c_void_p(buf.data_ptr())
    host = np.zeros(1 << 16, np.uint8)
    hp = host.ctypes.data_as(C.c_void_p)
    sz, u64, f6, d6 = C.c_size_t(0), C.c_uint64(0), (C.c_float * 6)(), (C.c_double * 6)()

    def refused(rc, code=None):
        assert rc < 0 and (code is None or rc == code), rc
        assert len(L.foa_last_error()) > 0

    # NULL handle everywhere
    for rc in (L.foa_rx_reserve(null, 1, 1), L.foa_rx_set_option(null, b"pipeline", 1), L.foa_rx_sync(null), L.foa_rx_wait_age(null, 1),
               L.foa_rx_decode_frames_dev(null, p, 100, p, p, 1, p, 64, p), L.foa_rx_decode_frames_host(null, hp, 100, hp, hp, 1, hp, 64, hp),
               L.foa_rx_submit_host(null, hp, 100, hp, hp, 1, 64, C.byref(u64)), L.foa_rx_collect(null, 1, 0, hp, hp),
               L.foa_rx_sync_dev(null, p, 100, p, p, 8, C.byref(sz)), L.foa_rx_sync_dev_begin(null, p, 100, p, p, 8), L.foa_rx_sync_dev_end(null, C.byref(sz)),
               L.foa_rx_last_kernel_ms(null, f6), L.foa_rx_probe_issue(null, d6), L.foa_conv_decode(null, hp, hp, 10, 1), L.foa_fft_forward_f64(null, hp, 1),
               L.foa_decode_header_f64(null, hp, 1, hp), L.foa_tx_channel_dev(null, p, 1, 400, 512, 0, 20.0, 0.0, 1, p),
               L.foa_stream_push_f32(null, hp, 10), L.foa_stream_flush(null), L.foa_stream_ready(null, 0, C.byref(sz), C.byref(sz)),
               L.foa_sync_push_f32(null, hp, 10, hp, 4, C.byref(sz)), L.foa_sync_set_call(null, 4096)):
        refused(rc)
    assert L.foa_rx_stream(null) is None and L.foa_sync_settled(null) == 0
    L.foa_rx_destroy(null); L.foa_stream_destroy(null); L.foa_sync_destroy(null)          # no-ops
    # NULL pointers with a good handle
    for rc in (L.foa_rx_decode_frames_dev(h, null, 100, p, p, 1, p, 64, p), L.foa_rx_decode_frames_dev(h, p, 100, null, p, 1, p, 64, p),
               L.foa_rx_decode_frames_dev(h, p, 100, p, p, 1, null, 64, p), L.foa_rx_decode_frames_host(h, hp, 100, hp, null, 1, hp, 64, hp),
               L.foa_rx_sync_dev(h, null, 100, p, p, 8, C.byref(sz)), L.foa_rx_sync_dev(h, p, 100, p, p, 8, None),
               L.foa_conv_decode(h, null, hp, 10, 1), L.foa_decode_data_f64(h, hp, null, 1, hp, hp, 64), L.foa_stream_create(h, 8192, 0, None)):
        refused(rc, E_INVALID)
    # out-of-range arguments
    for name, value in ((b"no_such_option", 1), (b"tb_segment", 100), (b"tb_segment", 96 * 40), (b"tb_overlap", 50), (b"depth", 7), (b"fe_hold", 1), (b"lanes", 1),
                        (b"frontend", 5), (b"viterbi", 3), (b"sync_call", 160), (b"sync_call", -4096), (b"sync_flags", 2)):
        refused(L.foa_rx_set_option(h, name, value), E_INVALID)
    refused(L.foa_rx_set_option(h, None, 1), E_INVALID)
    refused(L.foa_rx_wait_age(h, 5), E_INVALID)
    refused(L.foa_rx_wait_age(h, -1), E_INVALID)
    for bits in (0, -3, 8 * 5000):
        refused(L.foa_conv_decode(h, hp, hp, bits, 1), E_INVALID)
    res = np.zeros(2, foa.frame_result_dtype)
    res["rate"], res["length"] = (11, 3), (10, 5000)                       # no such rate; a length beyond 4095
    off = np.array([0, 48, 96], np.uint64)
    rc = L.foa_decode_data_f64(h, np.zeros(96 * 2).ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), 2, res.ctypes.data_as(C.c_void_p), hp, 64)
    assert rc < 0 or (res["status"] != 0).all()
    refused(L.foa_tx_build_frames_dev(h, p, 64, 10, 11, 1, p, C.byref(sz)), E_INVALID)                 # rate 11
    refused(L.foa_tx_build_frames_dev(h, p, 64, 5000, 3, 1, p, C.byref(sz)), E_INVALID)               # length 5000
    refused(L.foa_stream_create(h, 100, 0, C.byref(C.c_void_p())), E_INVALID)                         # a batch shorter than a frame
    sy = C.c_void_p()
    assert L.foa_sync_create(C.byref(sy)) == 0
    refused(L.foa_sync_set_call(sy, 100), E_INVALID)
    L.foa_sync_destroy(sy)
    # device-side ordering: NULL events, more than 16 events waiting for one call
    refused(L.foa_rx_after(h, None), E_INVALID)
    refused(L.foa_rx_record_consumed(h, None), E_INVALID)
    refused(L.foa_rx_record_done(None, p), E_INVALID)
    import torch
    evs = [torch.cuda.Event() for _ in range(17)]
    for e in evs:
        e.record()
    for e in evs[:16]:
        assert L.foa_rx_after(h, e.cuda_event) == 0
    refused(L.foa_rx_after(h, evs[16].cuda_event), E_STATE)
    assert L.foa_rx_decode_frames_dev(h, p, 100, p, p, 0, p, 64, p) == 0      # (an empty call takes the registered events with it)
    assert L.foa_rx_after(h, evs[16].cuda_event) == 0
    assert L.foa_rx_record_done(h, evs[0].cuda_event) == 0 and L.foa_rx_record_consumed(h, evs[1].cuda_event) == 0
    torch.cuda.synchronize()
    # wrong order
    refused(L.foa_rx_sync_dev_end(h, C.byref(sz)), E_STATE)
    refused(L.foa_rx_collect(h, 12345, 0, hp, hp))                                                     # no such ticket
    refused(L.foa_rx_get_taps(h, 1, hp, None, 0, None, None, 0, None))                                 # nothing decoded yet / wrong count
    # nothing to do is not an error
    assert L.foa_rx_decode_frames_dev(h, p, 100, p, p, 0, p, 64, p) == 0 and L.foa_rx_sync(h) == 0
    assert L.foa_rx_sync_dev(h, p, 0, p, p, 8, C.byref(sz)) == 0 and sz.value == 0
    # ... and the handle still decodes: a PSDU slot too small is a per-frame status, not an error
    pays = synth.splitmix64_bytes(3, 4, 100)
    iq, _ = synth.make_stream(synth.build_frames(pays, 10), 2048, 200, 25.0, seed=1)
    d = foa.find_alignments(iq)
    e = foa.alignment_ends(d, iq.size)
    psdu, r = rx.decode_frames_host(iq, d, e, slot_bytes=64)
    on = [k for k in range(d.size) if r["length"][k] == 100]
    assert len(on) == 4 and all(r["status"][k] == foa.ST_NO_SPACE if hasattr(foa, "ST_NO_SPACE") else r["status"][k] == 4 for k in on)
    psdu, r = rx.decode_frames_host(iq, d, e, slot_bytes=128)
    assert [psdu[k, :100].tobytes() for k in on] == [q.tobytes() for q in pays] and all(r["status"][k] == 0 for k in on)
    rx.close()


def test_handles_are_independent_across_threads():
    """One handle per thread (the header's threading contract): four threads decode different batches at the same time on the same
    device, each through its own foa_rx; every batch comes out as it does alone."""
    import threading
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth

    def work(seed, out):
        rx = foa.Receiver(0)
        rng = np.random.default_rng(seed)
        res = []
        for k in range(12):
            pays = synth.splitmix64_bytes(seed * 100 + k, int(rng.integers(5, 150)), int(rng.integers(10, 600)))
            iq, _ = synth.make_stream(synth.build_frames(pays, int(rng.choice((0, 5, 8, 10)))), 4096 * 5, 200, 25.0, seed=seed + k)
            d = foa.find_alignments(iq)
            psdu, r = rx.decode_frames_host(iq, d, foa.alignment_ends(d, iq.size))
            ok = np.nonzero(r["status"] == 0)[0]
            res.append([psdu[i, :r["length"][i]].tobytes() for i in ok] == [p.tobytes() for p in pays])
        rx.close()
        out[seed] = res

    out = {}
    th = [threading.Thread(target=work, args=(s, out)) for s in (1, 2, 3, 4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert sorted(out) == [1, 2, 3, 4] and all(all(r) for r in out.values())
