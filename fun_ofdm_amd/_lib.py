"""Loader for csrc/libfun_ofdm_amd.so (the C ABI of include/fun_ofdm_amd.h)."""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")


class FoaError(RuntimeError):
    pass


def library_path():
    """The library (FOA_LIB: another build of it, for A/B timing on one box)."""
    return os.environ.get("FOA_LIB") or os.path.join(CSRC, "libfun_ofdm_amd.so")


def build(force=False):
    """Compile the gfx950 library in-tree with hipcc (cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-s", "-C", CSRC, "clean"], check=True)
    subprocess.run(["make", "-s", "-j", "6", "-C", CSRC], check=True)
    return library_path()


_lib = None

_SIGS = {
    "foa_version": (C.c_int, []),
    "foa_last_error": (C.c_char_p, []),
    "foa_device_count": (C.c_int, []),
    "foa_rx_notes": (C.c_char_p, [C.c_void_p]),
    "foa_rx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "foa_rx_destroy": (None, [C.c_void_p]),
    "foa_rx_reserve": (C.c_int, [C.c_void_p, C.c_size_t, C.c_size_t]),
    "foa_rx_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "foa_rx_decode_frames_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_rx_decode_frames_ctx_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_rx_decode_frames_lead_ctx_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                    C.c_void_p]),
    "foa_rx_decode_frames_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_rx_decode_frames_f64_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_rx_sync": (C.c_int, [C.c_void_p]),
    "foa_rx_stream": (C.c_void_p, [C.c_void_p]),
    "foa_rx_submit_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64)]),
    "foa_rx_submit_host_ctx": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_uint64)]),
    "foa_rx_collect": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]),
    "foa_rx_after": (C.c_int, [C.c_void_p, C.c_void_p]),
    "foa_rx_record_consumed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "foa_rx_record_done": (C.c_int, [C.c_void_p, C.c_void_p]),
    "foa_rx_decode_frames_dev_after": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_rx_sync_dev_begin_after": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_rx_wait_age": (C.c_int, [C.c_void_p, C.c_int]),
    "foa_rx_wait_previous": (C.c_int, [C.c_void_p]),
    "foa_rx_kernel_ms_age": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "foa_rx_probe_issue": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "foa_rx_probe_h2d": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "foa_rx_forward_spacing_ms": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "foa_rx_prev_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "foa_rx_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "foa_rx_get_taps": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_rx_get_decisions": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "foa_tx_build_frames_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]),
    "foa_tx_channel_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.c_double,
                                     C.c_uint64, C.c_void_p]),
    "foa_rx_sync_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "foa_rx_sync_dev_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_rx_sync_dev_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "foa_stream_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "foa_stream_destroy": (None, [C.c_void_p]),
    "foa_stream_push_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_stream_push_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_stream_push_f64_owned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "foa_stream_flush": (C.c_int, [C.c_void_p]),
    "foa_stream_ready": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "foa_stream_take": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "foa_stream_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "foa_shard_create": (C.c_int, [C.c_void_p, C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "foa_shard_destroy": (None, [C.c_void_p]),
    "foa_shard_devices": (C.c_int, [C.c_void_p]),
    "foa_shard_push_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_shard_push_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_shard_push_f64_owned": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "foa_shard_flush": (C.c_int, [C.c_void_p]),
    "foa_shard_ready": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "foa_shard_take": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "foa_shard_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "foa_sync_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "foa_sync_destroy": (None, [C.c_void_p]),
    "foa_sync_push_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "foa_sync_push_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "foa_sync_settled": (C.c_int64, [C.c_void_p]),
    "foa_sync_set_call": (C.c_int, [C.c_void_p, C.c_int64]),
    "foa_fft_forward_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_conv_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_size_t]),
    "foa_channel_estimate_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "foa_equalize_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_phase_track_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_decode_header_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "foa_decode_data_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]),
}

EXPORTS = tuple(_SIGS)


def lib():
    """The loaded library.  There is deliberately no fallback: a missing or unloadable HIP library
    is an error."""
    global _lib
    if _lib is None:
        path = library_path()
        if not os.path.exists(path):
            raise FoaError("%s not found: build it with fun_ofdm_amd.build() / `make -C fun_ofdm_amd/csrc` "
                           "(this package has no CPU implementation)" % path)
        # PyTorch-ROCm wheels bundle their own libamdhip64; two HIP runtimes in one process do not
        # coexist (the second sees no GPU).  Importing torch first makes this library bind to the
        # runtime torch already loaded, so device buffers and streams can be shared with it.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(rc, L=None):
    if rc != 0:
        raise FoaError("fun_ofdm_amd error %d: %s" % (rc, (L or lib()).foa_last_error().decode()))
