import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np
import fun_ofdm_amd as foa
from oracle import pyoracle as po
import test_gpu_parity as T
rx = T._RxPair()          # product + cross-check receivers, as the test fixture
bad = 0
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 260
for seed in range(lo, hi):
    try:
        T.test_random_batches_vs_oracle.__wrapped__(rx, po, seed) if hasattr(T.test_random_batches_vs_oracle, '__wrapped__') else T.test_random_batches_vs_oracle(rx, po, seed)
    except AssertionError as e:
        bad += 1; print('FAIL', seed, str(e)[:200])
print('seeds %d..%d done, failures: %d' % (lo, hi - 1, bad))
