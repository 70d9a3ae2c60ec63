"""What the version-dependent decisions of the restated initialiser / LM move (oracle/epnp.inc header): the CPU
restatement run with each switch either way on config-2 objects.  Test infrastructure (imports the oracle); CPU only.
    python tests/sweeps/epnp_version_choices.py [--objects 8192] [--threads 16]  > profiles/r04_epnp_version_choices.txt"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from monorun_amd import synthetic as syn
from oracle import oracle as orc


def run(n, threads, seed0=1234):
    out = []
    for k in range(0, n, 1024):
        b = syn.make_batch(B=min(1024, n - k), seed=seed0 + 7919 * (k // 1024))
        x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
        out.append(orc.u2d_pnp_epnp(x2d, istd, x3d, K, ur, vr, 0.5, 0.6, thr, True, num_threads=threads, return_init=True))
    return [np.concatenate([o[i] for o in out]) for i in range(len(out[0]))]


def report(name, a, b):
    ok = a[0] & b[0]
    d = np.concatenate([np.angle(np.exp(1j * (a[1] - b[1]))), a[2] - b[2]], 1)
    ad = np.abs(d).max(1)[ok]
    print(f'{name}: valid {int(a[0].sum())} / {int(b[0].sum())} of {len(a[0])}; different valid flags {int((a[0] != b[0]).sum())}; objects with a different inlier mask '
          f'{int((a[5] != b[5]).any(1).sum())}; post-LM pose difference max {ad.max():.3e}, p99 {np.quantile(ad, 0.99):.3e}, median {np.median(ad):.3e}; '
          f'objects beyond 1e-4: {int((ad > 1e-4).sum())}, beyond 1e-6: {int((ad > 1e-6).sum())}; bit-identical float32 poses: {int((np.abs(d).max(1) == 0).sum())}')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--objects', type=int, default=8192); ap.add_argument('--threads', type=int, default=min(16, orc.max_threads()))
    a = ap.parse_args()
    print(f'{a.objects} config-2 objects (8 seeds x 1024), the reference flow restated on the CPU (EPnP / RANSAC initialiser + LM + covariance)')
    base = run(a.objects, a.threads)
    # (i) the re-fit's normalised image points: float64 (default since round 4) vs float32 (round 3)
    orc.set_epnp_refit_f64(False)
    f32 = run(a.objects, a.threads)
    orc.set_epnp_refit_f64(True)
    report('(i)   re-fit normalised in float64 (default) vs float32', base, f32)
    dd = np.abs(base[6] - f32[6]); dd[:, 0] = np.abs(np.angle(np.exp(1j * (base[6][:, 0] - f32[6][:, 0]))))
    good = np.isfinite(dd).all(1) & base[0] & f32[0]
    print(f'      initial pose [yaw0, t] of {int(good.sum())} objects, float64 vs float32 re-fit: max {dd[good].max():.3e}, p99 {np.quantile(dd[good].max(1), 0.99):.3e}, '
          f'median {np.median(dd[good].max(1)):.3e}')
    # (iii) Ceres' gradient test before the first step
    orc.set_lm_iter0_gradient_test(False)
    g0 = run(a.objects, a.threads)
    orc.set_lm_iter0_gradient_test(True)
    report("(iii) Ceres' gradient test at iteration 0 on (default) vs off", base, g0)
    # (iv) the eigen-solver of M^T M: the specification (four smallest eigenvectors) vs round 3's complete QL decomposition vs cyclic Jacobi
    for mode, nm in ((2, 'Householder + implicit QL (round 3)'), (1, 'cyclic Jacobi')):
        orc.set_epnp_eig_mode(mode)
        alt = run(a.objects, a.threads)
        orc.set_epnp_eig_mode(0)
        report(f'(iv)  M^T M solver: low4 (specification) vs {nm}', base, alt)


if __name__ == '__main__':
    main()
