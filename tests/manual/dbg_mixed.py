import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import fun_ofdm_amd as foa
from oracle import pyoracle as po
sys.path.insert(0, 'tests')
from test_gpu_parity import _make_stream, _ends
rx = foa.Receiver(0, xcheck=True); rx.set_option("viterbi", 1)
rng = np.random.default_rng(23)
specs = [(r, int(rng.integers(1, 400))) for r in range(11)] * 2 + [(10, 1024), (0, 37), (2, 1500), (9, 4095), (8, 1)]
iq, pays = _make_stream(po, rng, specs, snr_db=19.0, cfo_hz=3000.0)
descs = po.find_alignments_f32(iq); ends = _ends(descs, iq.size)
psdu, res = rx.decode_frames_host(iq, descs, ends)
opsdu, ores = po.decode_batch_f32(iq, descs, ends, threads=4)
for f in range(descs.size):
    rp = po.rate_params(int(ores[f]['rate'])) if ores[f]['rate'] >= 0 else None
    T = ores[f]['num_symbols'] * rp['dbps'] if rp else 0
    bad = tuple(res[f]) != tuple(ores[f])
    print(f, 'T', T, 'T%60', T % 60, 'T%6', T % 6, tuple(res[f]), tuple(ores[f]), 'BAD' if bad else '')

def slot_to_ref(dec_slot):
    """slot-order words -> reference decision_t words"""
    out = np.zeros_like(dec_slot)
    for t in range(dec_slot.size):
        w = int(dec_slot[t]); r = 0
        sh = (t + 1) % 6
        for p in range(64):
            if (w >> p) & 1:
                lab = ((p << sh) | (p >> (6 - sh))) & 63
                r |= 1 << lab
        out[t] = r
    return out
t = rx.taps(descs.size)
for f in (0, 1, 2, 3, 23, 24):
    soft = t["soft"][t["soft_off"][f]:t["soft_off"][f + 1]]
    n = soft.size // 2
    want, _, _ = po.viterbi_forward(soft, n)
    got = slot_to_ref(rx.decisions(f))
    bad = np.nonzero(got != want)[0]
    print('frame', f, 'steps', n, 'mismatching steps', bad.size, bad[:10], (bad % 60)[:10] if bad.size else '')
