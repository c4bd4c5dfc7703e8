"""fun_ofdm_amd -- MI355X-native 802.11a-like OFDM receive hot path behind fun_ofdm's
receiver_chain::process_samples() / block<I,O>::work() surface.

The arithmetic lives in hand-written HIP kernels for gfx950 (csrc/), reached through the C ABI of
include/fun_ofdm_amd.h.  This Python layer is host plumbing (ctypes + torch device buffers); it has
no CPU implementation of the path and raises if the HIP library is missing.
"""
from ._lib import lib, library_path, FoaError, build  # noqa: F401
from .rx import (Receiver, Sync, Stream, Shard, find_alignments, alignment_ends, frame_desc_dtype, frame_result_dtype, ST_OK, ST_HEADER_FAIL, ST_CRC_FAIL,  # noqa: F401
                 ST_TRUNCATED, ST_NO_SPACE, ST_SUPERSEDED, RATE_NAMES, RATE_MBPS, STANDARD_RATES)
