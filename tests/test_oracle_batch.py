"""The oracle's batch restatement (fo_decode_batch_v2_f32: what fft_symbols .. frame_decoder do, per alignment) against the oracle's BLOCKS
fed with the same tag stream (fo_chain_from_tags_f32) -- and its own generalisations: explicit ends, unlinked ends, context alignments.
CPU only; the device's batch path is compared with the same functions in tests/test_gpu_fuzz.py."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "manual"))


def test_restatement_equals_the_blocks_on_placed_tags(po):
    """fft_symbols.cpp:41-50 / channel_est.cpp:77-81 / frame_decoder.cpp:52-88 with tags placed where they decide a frame's fate: the ordered
    payload list of the blocks = the restatement's; decoding every alignment on its own and fixing up the cut ones (fo_decode_batch_f32) = the
    restatement, status by status."""
    import stress_tags
    tot, n_al, bad, hits = stress_tags.run_cpu(0, 250)
    assert bad == 0 and n_al > 1500 and tot > 140 and hits > 45, (tot, n_al, bad, hits)


def test_unlinked_ends_make_alignments_independent(po):
    """An end that is not the next alignment's LTS1 is where the stream ends for that alignment: it decodes exactly as it does alone, and
    never SUPERSEDED."""
    import stress_tags
    rng = np.random.default_rng(5)
    seen_trunc = 0
    for seed in range(40):
        s, d = stress_tags.make_case(seed)
        nxt = np.append(d["lts1_pos"][1:], s.size).astype(np.int64)
        ends = nxt - rng.integers(1, 40, d.size)                       # a little short of the next LTS1: unlinked
        psdu, res = po.decode_batch_v2_f32(s, d, ends)
        for j in range(d.size):
            r1, p1 = po.decode_alignment_f32(s, d[j], end=int(ends[j]))
            assert tuple(res[j]) == tuple(r1), (seed, j, res[j], r1)
            if r1["status"] == 0:
                assert np.array_equal(psdu[j, :r1["length"]], p1[:r1["length"]])
        assert not np.any(res["status"] == po.ST_SUPERSEDED)
        seen_trunc += int(np.count_nonzero(res["status"] == po.ST_TRUNCATED))
    assert seen_trunc > 10


def test_context_alignments_serve_the_frames_before_them(po):
    """A stream decoded in two pieces: the first k alignments with the rest as context give the results of the one-piece decode; without
    context a frame that fills on into the missing alignments is TRUNCATED (the samples handed over end), never anything else."""
    import stress_tags
    differ = 0
    for seed in range(120):
        s, d = stress_tags.make_case(seed)
        if d.size < 3:
            continue
        ends = np.append(d["lts1_pos"][1:], s.size).astype(np.int64)
        full_p, full_r = po.decode_batch_v2_f32(s, d, ends)
        for k in range(1, d.size):
            p1, r1 = po.decode_batch_v2_f32(s, d, ends, n_ctx=d.size - k)
            assert np.array_equal(r1.view(np.int32), full_r[:k].view(np.int32)) and np.array_equal(p1, full_p[:k]), (seed, k)
            p0, r0 = po.decode_batch_v2_f32(s, d[:k], ends[:k])            # no context: the last alignment's end is nobody's LTS1 in this call
            same = r0["status"] == full_r["status"][:k]
            assert np.all(same | (r0["status"] == po.ST_TRUNCATED)), (seed, k, r0["status"], full_r["status"][:k])
            differ += int(np.count_nonzero(~same))
    assert differ > 20          # the cases really depend on what follows
