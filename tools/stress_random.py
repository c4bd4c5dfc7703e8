import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import fun_ofdm_amd as foa
from oracle import pyoracle as po
import test_gpu_parity as T
rx = foa.Receiver(0)
bad = 0
for seed in range(200, 260):
    try:
        T.test_random_batches_vs_oracle.__wrapped__(rx, po, seed) if hasattr(T.test_random_batches_vs_oracle, '__wrapped__') else T.test_random_batches_vs_oracle(rx, po, seed)
    except AssertionError as e:
        bad += 1; print('FAIL', seed, str(e)[:200])
print('done, failures:', bad)
