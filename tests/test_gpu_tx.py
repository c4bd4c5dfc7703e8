"""Transmit side on the device: foa_tx_build_frames_dev against the oracle's build_frame, and a loop-back
that never leaves HBM (device TX -> device channel -> device pre-sync -> decode).  GPU only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rx():
    import fun_ofdm_amd as foa
    r = foa.Receiver(0)
    yield r
    r.close()


@pytest.mark.parametrize("rate", list(range(11)))
def test_tx_frames_match_oracle(rx, po, rate):
    """frame_builder::build_frame: every sample within 1e-12 of the oracle's (the integer stages are exact or the
    constellation points would be off by O(1); the IFFT factorisations differ in the last bits); preamble identical."""
    import torch
    rng = np.random.default_rng(60 + rate)
    for length in (1, 37, 100, 1024 if rate == 10 else 333, 4095 if rate in (0, 10) else 700):
        pays = rng.integers(0, 256, (3, length), dtype=np.uint8)
        dev = torch.from_numpy(pays).to("cuda:0")
        got = rx.tx_build_frames(dev, rate).cpu().numpy()
        got = got[..., 0] + 1j * got[..., 1]
        for i in range(pays.shape[0]):
            want = po.build_frame(pays[i], rate)
            assert got[i].size == want.size == rx.tx_frame_samples(length, rate), (rate, length)
            assert np.array_equal(got[i][:320], want[:320]), (rate, length)
            err = np.abs(got[i] - want).max() / np.abs(want).max()
            assert err < 1e-12, (rate, length, i, err)


def test_tx_payload_pitch_and_zero_length(rx, po):
    import torch
    buf = torch.zeros((4, 64), dtype=torch.uint8, device="cuda:0")
    buf[:, :10] = torch.arange(40, dtype=torch.uint8, device="cuda:0").reshape(4, 10)
    view = buf[:, :10]                                           # pitch 64, length 10
    got = rx.tx_build_frames(view, 5).cpu().numpy()
    for i in range(4):
        want = po.build_frame(np.arange(10 * i, 10 * i + 10, dtype=np.uint8), 5)
        g = got[i, :, 0] + 1j * got[i, :, 1]
        assert np.abs(g - want).max() < 1e-12
    empty = torch.zeros((2, 0), dtype=torch.uint8, device="cuda:0")
    got = rx.tx_build_frames(empty, 0).cpu().numpy()
    want = po.build_frame(np.zeros(0, np.uint8), 0)
    assert np.abs(got[0, :, 0] + 1j * got[0, :, 1] - want).max() < 1e-12


@pytest.mark.parametrize("rate,length,snr,cfo,min_ok", [(10, 1024, 25.0, 0.0, 300), (6, 300, 20.0, 3000.0, 240), (0, 60, 20.0, 4000.0, 250)])
def test_loopback_stays_on_device(rx, rate, length, snr, cfo, min_ok):
    """payloads -> device TX -> device channel -> device frame_detector/timing_sync -> decode: at 25 dB every transmitted
    frame comes back bit-exact (at lower SNR the reference's detector misses some; those that are found and pass the
    CRC must carry their payload); nothing but payloads in and PSDUs out crosses PCIe."""
    import torch
    import fun_ofdm_amd as foa
    from fun_ofdm_amd import synth
    n = 300
    pays = synth.splitmix64_bytes(0x70 + rate, n, length)
    dev = torch.device("cuda", 0)
    frames = rx.tx_build_frames(torch.from_numpy(pays).to(dev), rate)
    s = frames.shape[1]
    pitch = -(-(s + 600) // 4096) * 4096
    lead = 176
    iq = rx.tx_channel(frames, pitch, lead, snr, seed=1234 + rate, cfo_hz=cfo)
    cap = n * pitch // 256 + 64                                  # low SNR: the detector also fires on noise
    descs = torch.zeros(cap * foa.frame_desc_dtype.itemsize, dtype=torch.uint8, device=dev)
    ends = torch.zeros(cap, dtype=torch.int64, device=dev)
    m = rx.sync_dev(iq, descs, ends)
    assert 0 < m <= cap
    psdu = torch.zeros((m, length), dtype=torch.uint8, device=dev)
    res = torch.zeros((m, 4), dtype=torch.int32, device=dev)
    rx.decode_frames_dev(iq, descs[:m * foa.frame_desc_dtype.itemsize], ends[:m], psdu, res)
    rx.sync()
    d = descs.cpu().numpy()[:m * foa.frame_desc_dtype.itemsize].view(foa.frame_desc_dtype)
    res, psdu = res.cpu().numpy(), psdu.cpu().numpy()
    real = np.nonzero(((d["lts1_pos"] - (lead + 184)) % pitch == 0) & (res[:, 0] == 0))[0]
    which = (d["lts1_pos"][real] - (lead + 184)) // pitch
    assert real.size >= min_ok and np.unique(which).size == real.size
    assert (res[real, 1] == rate).all() and (res[real, 2] == length).all()
    assert np.array_equal(psdu[real], pays[which])
