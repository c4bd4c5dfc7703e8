"""BASELINE.json configs: 1 (the loopback frame) and 5 (continuous mixed-rate stream with CFO, at a size the oracle walks) against the oracle's
receiver_chain, and 2 and 3 at their FULL sizes as property tests (the oracle cannot decode 10 000 frames in seconds, so the size-independent
property is asserted on every frame and the oracle on a sample): every frame whose CRC passed carries exactly the payload that was sent,
(nearly) every frame passes at 25 dB, and on the first 256 alignments status, header fields, PSDUs -- the CRC failures included -- equal
the oracle's.  Workloads are built on the device (foa_tx_build_frames_dev + foa_tx_channel_dev).  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(po, rate, length, n_frames, seed, sample=256):
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    dev = torch.device("cuda", 0)
    rx = foa.Receiver(0)
    try:
        pays = synth.splitmix64_bytes(0x0FD0 + rate, n_frames, length)
        frames = rx.tx_build_frames(torch.from_numpy(pays).to(dev), rate)
        s = frames.shape[1]
        pitch = -(-(s + 576) // 4096) * 4096                    # a multiple of 4096, lead 176
        d_iq = rx.tx_channel(frames, pitch, 176, 25.0, seed=seed)
        del frames
        cap = d_iq.shape[0] // 512 + 64
        d_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        d_end = torch.zeros(cap, dtype=torch.int64, device=dev)
        m = rx.sync_dev(d_iq, d_desc, d_end)
        d_psdu = torch.zeros((m, length), dtype=torch.uint8, device=dev)
        d_res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
        for _ in range(2):                                       # twice: pipelined calls on rotating work sets give the same answer
            rx.decode_frames_dev(d_iq, d_desc[:m * 48], d_end[:m], d_psdu, d_res)
        rx.sync()
        res = d_res.cpu().numpy()
        psdu = d_psdu.cpu().numpy()
        descs = d_desc.cpu().numpy()[:m * 48].view(foa.frame_desc_dtype)
        ends = d_end[:m].cpu().numpy()
        on = np.nonzero((descs["lts1_pos"] - 360) % pitch == 0)[0]              # alignments that sit on a frame (lead 176 + 184)
        which = (descs["lts1_pos"][on] - 360) // pitch
        ok = res[on, 0] == foa.ST_OK
        assert np.unique(which).size == on.size
        assert np.array_equal(psdu[on][ok], pays[which][ok]), "a CRC-passing frame does not carry its payload"
        assert np.all(res[on, 1] == rate) or np.count_nonzero(res[on, 1] != rate) <= 2
        assert np.all(res[on][ok, 2] == length)
        # the oracle on the first `sample` alignments: same statuses (CRC failures included), same fields, same PSDUs
        k = min(sample, m)
        h_iq = d_iq[:int(ends[k - 1])].cpu().numpy().reshape(-1).view(np.complex64)
        e = ends[:k].copy()
        opsdu, ores = po.decode_batch_f32(h_iq, descs[:k], e, slot_bytes=length, threads=8)
        assert np.array_equal(ores.view(np.int32).reshape(-1, 4), res[:k]), "status / header fields differ from the oracle's on the sample"
        okk = res[:k, 0] == foa.ST_OK
        assert np.array_equal(opsdu[okk], psdu[:k][okk])
        return int(on.size), int(np.count_nonzero(ok)), int(np.count_nonzero(res[:k, 0] == foa.ST_CRC_FAIL)), int(np.count_nonzero(ores["status"] == foa.ST_CRC_FAIL))
    finally:
        rx.close()


def test_config2_ten_thousand_frames(po):
    """BASELINE configs[1]: 10 000 frames x 1024-byte PSDU, 64-QAM r=3/4 (54 Mbps), AWGN 25 dB."""
    found, ok, gpu_fail, cpu_fail = _run(po, 10, 1024, 10000, seed=7919)
    assert found >= 9990 and ok >= found - 10 and gpu_fail == cpu_fail, (found, ok, gpu_fail, cpu_fail)


@pytest.mark.parametrize("rate", [0, 2, 3, 5, 6, 8, 9, 10])
def test_config3_thousand_frames_per_rate(po, rate):
    """BASELINE configs[2]: the eight 802.11a rates, 4092-byte payloads (PSDU incl. CRC = 4096 bytes), 1 000 frames per rate, 25 dB.  (9 Mbps long
    frames fail their CRC now and then in the reference too, SURVEY 8d: the failures must be the same ones.)"""
    found, ok, gpu_fail, cpu_fail = _run(po, rate, 4092, 1000, seed=300 + rate, sample=96 if rate < 5 else 256)
    assert found >= 995 and gpu_fail == cpu_fail, (rate, found, ok, gpu_fail, cpu_fail)
    assert ok >= (found - 8 if rate != 2 else int(0.7 * found)), (rate, found, ok)


def test_config1_loopback_one_bpsk_frame(po):
    """BASELINE configs[0] (examples/test_sim.cpp's loopback shape): ONE frame, BPSK rate 1/2, a 1500-byte text payload, no noise,
    fed in 4096-sample calls with zeros behind it -- the case the reference's own CPU receiver_chain runs.  The oracle's receiver_chain returns
    exactly that payload; the batch path (host pre-sync + foa_rx_decode_frames_host), the stream engine (process_samples on the device) in
    4096-sample pushes and the same through a 4096-sample batch must return the same list."""
    import fun_ofdm_amd as foa
    text = (b"This is the payload of the single frame of the loopback test: 1500 bytes of text, as examples/test_sim.cpp sends them. " * 14)[:1500]
    pay = np.frombuffer(text, np.uint8)
    fr = po.build_frame(pay, 0)
    iq = np.concatenate([np.zeros(700, complex), fr, np.zeros(6 * 4096 + 300, complex)]).astype(np.complex64)
    want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert want == [text]
    rx = foa.Receiver(0)
    try:
        descs = foa.find_alignments(iq)
        assert descs.size == 1
        psdu, res = rx.decode_frames_host(iq, descs, foa.alignment_ends(descs, iq.size))
        assert res["status"][0] == foa.ST_OK and res["rate"][0] == 0 and res["length"][0] == 1500 and psdu[0, :1500].tobytes() == text
        opsdu, ores = po.decode_batch_f32(iq, descs, foa.alignment_ends(descs, iq.size))
        assert np.array_equal(res.view(np.int32), ores.view(np.int32)) and np.array_equal(psdu, opsdu)
        for batch in (4096, 1 << 16):
            st = foa.Stream(rx, batch)
            try:
                got = []
                for a in range(0, iq.size, 4096):
                    got += st.push(iq[a:a + 4096])
                got += st.flush()
            finally:
                st.close()
            assert got == want, batch
    finally:
        rx.close()


def test_config5_continuous_mixed_rate_stream_with_cfo(po):
    """BASELINE configs[4] at a size the oracle's chain walks in seconds: frames cycling the eight standard rates back to back (no gap at all),
    1024-byte payloads, a carrier frequency offset per frame within +-4 kHz, 25 dB; device pre-sync + decode (the bench's config-5 leg does
    this on 45 M samples) and the stream engine against the oracle's receiver_chain in 4096-sample calls: the same ordered payload list."""
    import torch
    import fun_ofdm_amd as foa
    rng = np.random.default_rng(505)
    parts, pays = [np.zeros(500, complex)], []
    for i in range(48):
        rate = foa.STANDARD_RATES[i % 8]
        pay = rng.integers(0, 256, 1024, dtype=np.uint8)
        f = po.build_frame(pay, rate)
        f = f * np.exp(2j * np.pi * rng.uniform(-4000, 4000) * np.arange(f.size) / 20e6 + 1j * rng.uniform(0, 6.28))
        parts.append(f)
        pays.append(pay.tobytes())
    parts.append(np.zeros(900, complex))
    s = np.concatenate(parts)
    iq = (s + (rng.normal(size=s.size) + 1j * rng.normal(size=s.size)) * np.sqrt(0.0124 / 2 / 10 ** 2.5)).astype(np.complex64)
    want = po.ReceiverChain().run_stream(iq.astype(np.complex128))
    assert len(want) >= 40 and all(w in pays for w in want)
    dev = torch.device("cuda", 0)
    rx = foa.Receiver(0)
    try:
        t_iq = torch.from_numpy(iq.view(np.float32).reshape(-1, 2)).to(dev)
        cap = iq.size // 300 + 64
        d_desc = torch.zeros(cap * 48, dtype=torch.uint8, device=dev)
        d_end = torch.zeros(cap, dtype=torch.int64, device=dev)
        m = rx.sync_dev(t_iq, d_desc, d_end)
        d_psdu = torch.zeros((m, 1024), dtype=torch.uint8, device=dev)
        d_res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
        rx.decode_frames_dev(t_iq, d_desc[:m * 48], d_end[:m], d_psdu, d_res)
        rx.sync()
        res, psdu = d_res.cpu().numpy(), d_psdu.cpu().numpy()
        got = [psdu[i, :res[i, 2]].tobytes() for i in range(m) if res[i, 0] == foa.ST_OK]
        assert got == want
        st = foa.Stream(rx, 1 << 16)
        try:
            got2 = []
            for a in range(0, iq.size, 4096):
                got2 += st.push(iq[a:a + 4096])
            got2 += st.flush()
        finally:
            st.close()
        assert got2 == want
    finally:
        rx.close()
