"""Bitwise comparison of the GPU's 12x12 solver (dump written by tools/ubench/eig12_low4 on the GPU box) with the CPU restatement
oracle.eig12_low4 (test infrastructure; development aid).   python tests/sweeps/check_eig12_low4.py gpurun_out/eig12_low4.bin"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as orc
raw = open(sys.argv[1], 'rb').read()
n = int(np.frombuffer(raw[:4], np.int32)[0])
d = np.frombuffer(raw[4:], np.float64)
A, ev, w = d[:n * 144].reshape(n, 12, 12), d[n * 144:n * 192].reshape(n, 4, 12), d[n * 192:n * 196].reshape(n, 4)
bad = 0; worst = 0.0
for i in range(n):
    w4, v4 = orc.eig12_low4(A[i])
    if not (np.array_equal(v4.view(np.uint64), ev[i].view(np.uint64)) and np.array_equal(w4.view(np.uint64), w[i].view(np.uint64))):
        bad += 1; worst = max(worst, np.abs(v4 - ev[i]).max())
print(f'{n} matrices: {n - bad} bit-identical to the CPU restatement, {bad} differ (max |dv| {worst:.3e})')
