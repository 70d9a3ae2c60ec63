#!/usr/bin/env python3
"""K0 (this repo's initialiser) against the reference's own initialiser restated (EPnP + OpenCV's RANSAC loop,
oracle/epnp.inc), compared AFTER the LM on config-2 batches — the table in DESIGN.md §5.  CPU only (oracle on both sides;
the HIP kernel is bit-identical to the K0 oracle on masks: tests/test_gpu_parity.py).

    python tests/sweeps/k0_vs_epnp.py [--seeds 8] [--B 1024]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=8)
    ap.add_argument('--B', type=int, default=1024)
    a = ap.parse_args()
    from oracle import oracle as orc
    from test_oracle_epnp import k0_vs_epnp_statistics
    S = [k0_vs_epnp_statistics(orc, 1234 + 7919 * i, B=a.B, num_threads=0) for i in range(a.seeds)]
    cat = lambda k: np.concatenate([s[k] for s in S])
    ok, same, iou, d, maha = cat('ok'), cat('same'), cat('iou'), cat('d'), cat('maha')
    n = len(ok)
    print(f'objects {n}   valid K0 {cat("valid0").sum()}   valid EPnP {cat("valid1").sum()}')
    print(f'identical inlier sets {same.mean():.4f}   mask IoU mean {iou.mean():.5f}  p1 {np.quantile(iou, .01):.4f}  min {iou.min():.4f}')
    for name, sel in (('identical sets', same & ok), ('different sets', ~same & ok), ('all', ok)):
        q = np.quantile(d[sel], [.5, .9, .99, 1.0]); m = np.nanquantile(maha[sel], [.5, .9, .99, 1.0])
        print(f'{name:15s} n={sel.sum():6d}  max|dpose| p50/p90/p99/max = ' + ' '.join(f'{v:.2e}' for v in q) +
              '   in posterior sigmas = ' + ' '.join(f'{v:.2e}' for v in m))
    for tol in (1e-4, 1e-3, 1e-2, 1e-1):
        print(f'fraction of objects with max|dpose| <= {tol:g}: {np.mean(d[ok] <= tol):.4f}')


if __name__ == '__main__':
    main()
