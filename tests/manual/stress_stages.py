"""Random carriers through the stage-level entry points foa_decode_header_f64 and foa_decode_data_f64 (ppdu::decode_header /
decode_data: demap, de-interleave, de-puncture, Viterbi, descramble, CRC) against the oracle: valid SIGNAL / data symbols at any
noise level and amplitude (constellation scaled by 1e-3 .. 1e3: hard-limited demapping), garbage carriers, huge and tiny values.
Usage (GPU box, from the repo root): python3 tests/manual/stress_stages.py [first seed] [last seed]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import fun_ofdm_amd as foa
from oracle import pyoracle as po


def run(lo, hi):
    rx = foa.Receiver(0)
    L, h = rx._lib, rx._h
    bad = n_hdr = n_data = n_ok = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        # ---- headers
        k = int(rng.integers(1, 40))
        car = np.zeros((k, 48), np.complex128)
        for i in range(k):
            kind = int(rng.integers(0, 4))
            if kind < 2:
                c = po.encode_header(int(rng.integers(0, 11)), int(rng.integers(0, 4096)))
                c = c * 10 ** rng.uniform(-3, 3) + (rng.normal(size=48) + 1j * rng.normal(size=48)) * rng.uniform(0, 0.8) * (kind == 1)
            elif kind == 2:
                c = (rng.normal(size=48) + 1j * rng.normal(size=48)) * 10 ** rng.uniform(-6, 6)
            else:
                c = rng.choice((-1.0, 1.0, 0.0, 1e12, -1e12, 1e-300), 48) + 1j * rng.choice((-1.0, 1.0, 0.0), 48)
            car[i] = c
        res = np.zeros(k, foa.frame_result_dtype)
        foa._lib.check(L.foa_decode_header_f64(h, car.ctypes.data_as(C.c_void_p), k, res.ctypes.data_as(C.c_void_p)), L)
        for i in range(k):
            n_hdr += 1
            want = po.decode_header(car[i])
            got = (int(res["rate"][i]), int(res["length"][i]), int(res["num_symbols"][i])) if res["status"][i] == 0 else None
            if got != want:
                bad += 1
                print("FAIL seed", seed, "header", i, got, want)
        # ---- data
        nf = int(rng.integers(1, 8))
        rates = [int(rng.integers(0, 11)) for _ in range(nf)]
        lens = [int(rng.choice((0, 1, int(rng.integers(2, 200)), int(rng.integers(200, 1500))))) for _ in range(nf)]
        cars, offs = [], [0]
        for r, ln in zip(rates, lens):
            pay = rng.integers(0, 256, ln, dtype=np.uint8)
            c = po.encode_data(pay, r)
            kind = int(rng.integers(0, 4))
            if kind == 1:
                c = c + (rng.normal(size=c.size) + 1j * rng.normal(size=c.size)) * rng.uniform(0, 0.5)
            elif kind == 2:
                c = c * 10 ** rng.uniform(-3, 3)
            elif kind == 3:
                c = (rng.normal(size=c.size) + 1j * rng.normal(size=c.size)) * 10 ** rng.uniform(-3, 6)
            cars.append(c)
            offs.append(offs[-1] + c.size)
        allc = np.concatenate(cars) if cars else np.zeros(0, np.complex128)
        off = np.array(offs, np.uint64)
        res = np.zeros(nf, foa.frame_result_dtype)
        res["rate"], res["length"] = rates, lens
        slot = 1504
        psdu = np.zeros((nf, slot), np.uint8)
        foa._lib.check(L.foa_decode_data_f64(h, allc.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p), nf, res.ctypes.data_as(C.c_void_p),
                                             psdu.ctypes.data_as(C.c_void_p), slot), L)
        for i in range(nf):
            n_data += 1
            want = po.decode_data(cars[i], rates[i], lens[i])
            ok = res["status"][i] == 0
            n_ok += int(ok)
            if (want is None) != (not ok) or (ok and not np.array_equal(psdu[i, :lens[i]], want)):
                bad += 1
                print("FAIL seed", seed, "frame", i, "rate", rates[i], "length", lens[i], "status", res["status"][i], "oracle", None if want is None else "ok")
    rx.close()
    return n_hdr, n_data, n_ok, bad


if __name__ == "__main__":
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    hi = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    n_hdr, n_data, n_ok, bad = run(lo, hi)
    print("seeds %d..%d done: %d SIGNAL symbols, %d data frames (%d pass their CRC); results that differ from the oracle: %d" % (lo, hi - 1, n_hdr, n_data, n_ok, bad))
