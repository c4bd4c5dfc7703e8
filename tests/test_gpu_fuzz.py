"""Short runs of the random differential checkers of tests/manual/ (the long runs are recorded in profiles/r03_soak.txt and profiles/r04_soak.txt): every
random input must come out of the HIP path exactly as it comes out of the oracle.  GPU only."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual"))


def test_random_streams_presync_device_host_and_oracle_agree(po):
    """Device pre-sync (blocking and in two halves), host restatement and the oracle's blocks in 4096-sample calls: the same alignments on
    random streams (frames at 3 .. 30 dB, gaps from none, ends anywhere, a NaN sample in a fifth of them), the frames the reference drops at
    its call boundaries (timing_sync.cpp:99) included."""
    import stress_sync
    tot, bad, host_bad = stress_sync.run(100, 260)
    assert bad == 0 and host_bad == 0 and tot > 800


def test_random_streams_stream_engine_equals_oracle_chain(po):
    """process_samples on the device (random push and batch sizes) against the oracle's receiver_chain: equal ordered payload lists."""
    import stress_stream
    tot, bad = stress_stream.run(200, 300)
    assert bad == 0 and tot > 300


def test_random_soft_byte_blocks_through_the_viterbi_path(po):
    """Seven soft-byte distributions, 1 .. 32 900 data bits, 1 .. 9 blocks per call, random chain-back segmentation: the oracle's decoder,
    and the compiled reference's when oracle/_ref is there."""
    import stress_viterbi
    nblocks, bad, _ = stress_viterbi.run(40000, 40150)
    assert bad == 0 and nblocks > 400


def test_perturbed_alignment_descriptors_decode_like_the_oracle(po):
    """Shifted positions, phasors off the unit circle, ends cut anywhere, positions at random -- also with frame amplitudes over twelve decades."""
    import stress_decode
    tot, passed, bad = stress_decode.run(50000, 50150)
    assert bad == 0 and tot > 1500 and passed > 100
    tot, passed, bad = stress_decode.run(51000, 51060, 6.0)
    assert bad == 0 and tot > 300


def test_colliding_frames_and_false_alarms_inside_frames_equal_the_block_chain(po):
    """fft_symbols.cpp:41-50 / channel_est.cpp:77-81 / frame_decoder.cpp:52-68: a preamble inside a frame -- the second frame weaker, equal or
    stronger, its SIGNAL valid or garbled, its LTS1 anywhere in the first frame or late in its last symbol, bare preambles, pile-ups.  The
    ordered payload list of the batch path (host and device pre-sync), of the stream engine and of fun_amd::receiver_chain::process_samples
    in its three modes against the oracle's BLOCK-LEVEL chain, which models the partial-vector flush."""
    import stress_collide
    tot, n_al, bad = stress_collide.run_gpu(1000, 1060, cpp=True)
    assert not bad and n_al > 100 and tot > 15, (tot, n_al, bad)
    tot, n_al, bad = stress_collide.run_gpu(2000, 2200)
    assert not bad and n_al > 350, (tot, n_al, bad)


def test_placed_tags_partial_vector_flush_and_frames_in_progress_equal_the_blocks(po):
    """fft_symbols.cpp:41-50 (the partly filled vector pushed when LTS1 arrives mid-symbol), channel_est.cpp:77-81, frame_decoder.cpp:52-88 (a frame
    fills on with whatever vectors follow its SIGNAL; a valid SIGNAL abandons it) with tags PLACED where they decide a frame's fate (late in its
    last symbol, a symbol earlier, anywhere, on noise, in pile-ups down to one sample apart): the device's batch path handed those descriptors against the oracle's BLOCKS fed with
    the same tag stream (ordered payload list) and against the per-alignment restatement (status, fields, PSDUs of every alignment).  A third
    of the delivered payloads of these cases exist only because of those rules."""
    import stress_tags
    tot, n_al, bad, hits = stress_tags.run_gpu(0, 400)
    assert bad == 0 and n_al > 2000 and tot > 180 and hits > 60, (tot, n_al, bad, hits)
    # ... and as the pre-rotated complex<double> stream the fused stage block hands over (foa_rx_decode_frames_f64_host)
    tot, n_al, bad = stress_tags.run_gpu_f64(400, 520)
    assert bad == 0 and n_al > 600 and tot > 40, (tot, n_al, bad)
