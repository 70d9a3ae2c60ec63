#!/usr/bin/env python3
"""The oracle's LM with the trust-region step solved two ways — Cholesky of the Jacobi-scaled normal equations (what the HIP
kernel does) and Householder QR of the augmented matrix [J S; D] (what Ceres' DENSE_QR does, pnp_uncert_cpu.cpp:270-274) — over
a sweep of config-2 batches: iteration counts, exit reasons, trust-region radii and fp64 poses must coincide.  CPU only.

    python tests/sweeps/lm_qr_vs_chol_sweep.py [--seeds 200] [--B 1024]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=200)
    ap.add_argument('--B', type=int, default=1024)
    a = ap.parse_args()
    from oracle import oracle as orc
    from monorun_amd import synthetic as syn
    n = it_diff = why_diff = 0
    worst_pose = worst_tr = 0.0
    hist = {}
    for i in range(a.seeds):
        # vary the conditions like tests/sweeps/gpu_parity_sweep.py: outlier share, noise, layout
        b = syn.make_batch(B=a.B, seed=1000 + i, outlier_frac=(0.05, 0.15, 0.3)[i % 3], noise_3d=(0.01, 0.03, 0.08)[(i // 3) % 3])
        x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(v) for v in syn.pnp_boundary(b, planar=False)]
        orc.set_lm_options(qr=False)
        r0 = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_pose64=True, num_threads=0)
        orc.set_lm_options(qr=True)
        r1 = orc.u2d_pnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, return_diag=True, return_pose64=True, num_threads=0)
        orc.set_lm_options()
        n += a.B
        it_diff += int((r0[6][:, 0] != r1[6][:, 0]).sum()); why_diff += int((r0[6][:, 2] != r1[6][:, 2]).sum())
        worst_pose = max(worst_pose, float(np.abs(r0[7] - r1[7]).max()))
        worst_tr = max(worst_tr, float(np.abs(r0[4] / np.maximum(r1[4], 1e-30) - 1).max()))
        for v, c in zip(*np.unique(r0[6][:, 0].astype(int), return_counts=True)):
            hist[int(v)] = hist.get(int(v), 0) + int(c)
        assert np.array_equal(r0[5], r1[5]) and np.array_equal(r0[0], r1[0])
    print(f'objects {n}: iteration-count differences {it_diff}, exit-reason differences {why_diff}, '
          f'worst |pose_chol - pose_qr| (fp64) {worst_pose:.3e}, worst relative radius difference (float32 outputs) {worst_tr:.3e}')
    print('LM iteration histogram:', {k: hist[k] for k in sorted(hist)})


if __name__ == '__main__':
    main()
