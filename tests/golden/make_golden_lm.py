#!/usr/bin/env python3
"""Golden fixture G7 — LM trajectories (per-pass cost / candidate cost / model cost change / relative decrease / radius /
step norm / outcome) of 64 objects, produced by the oracle's Ceres-1.14 restatement in its DENSE_QR-literal mode
(oracle/pnp_oracle.c: Householder QR of [J S; D], model cost change from the model residuals).

Unlike G1-G6 this fixture does NOT come from code of the reference (Ceres is absent and cannot be built: oracle/Makefile,
DESIGN.md §6); it FREEZES the restatement as committed data so that (a) any later change of the oracle or of the HIP kernel's
LM shows up as a diff against data, not against a moving target, and (b) the kernel is compared pass by pass (truncated runs
with max_num_iterations = k) with committed numbers.  The objects are picked to cover the LM's control flow: ordinary
convergence by every tolerance, rejected steps, invalid steps, the five-invalid-steps failure, max-iteration exits, an
evaluation failure, z / u / v clamps.

    python tests/golden/make_golden_lm.py        # rewrites tests/golden/g7_lm_trajectories.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from monorun_amd import synthetic as syn  # noqa: E402
from oracle import oracle as orc  # noqa: E402

HW = 14                      # 196 correspondences per object keep the fixture small
P = HW * HW
N_OBJ = 64
MAXP = 64                    # trace rows kept per object


def solve(o, **kw):
    clips = np.array([0.5, o['ur'][0], o['ur'][1], o['vr'][0], o['vr'][1]], np.float64)
    return orc.pnp_uncert_opt(o['x2d'].astype(np.float64), o['x3d'].astype(np.float64), o['w'].astype(np.float64), o['K'].astype(np.float64),
                              o['init'], clips, **kw)


def candidates():
    """a stream of (tag, object) with inputs exactly as the kernel will see them (float32 arrays, fp64 initial pose)"""
    rng = np.random.default_rng(20260928)
    for rep in range(40):
        b = syn.make_batch(B=48, hw=HW, seed=500 + rep, outlier_frac=(0.0, 0.15, 0.4)[rep % 3])
        x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
        gt = np.concatenate([b['gt_yaw'][:, None], b['gt_t']], 1)
        for i in range(48):
            kind = rng.integers(0, 8)
            o = dict(x2d=x2d[i].copy(), w=istd[i].copy(), x3d=x3d[i].copy(), K=K[0].copy(), ur=ur[0].copy(), vr=vr[0].copy())
            if kind <= 2:      # mild perturbation of the ground truth
                o['init'] = gt[i] + rng.normal(0, 1, 4) * np.array([0.05, 0.2, 0.05, 0.8]); tag = 'mild'
            elif kind == 3:    # far off: rejected steps, many iterations
                o['init'] = gt[i] + rng.normal(0, 1, 4) * np.array([1.5, 3.0, 1.0, 12.0]); tag = 'far'
            elif kind == 4:    # behind / very close to the camera: z clamp active, flat directions
                o['init'] = np.array([gt[i, 0] + rng.normal(0, 0.5), gt[i, 1], gt[i, 2], rng.uniform(-3.0, 1.0)]); tag = 'zclamp'
            elif kind == 5:    # far to the side: u / v clamps active
                o['init'] = gt[i] + np.array([rng.normal(0, 0.3), rng.choice([-1, 1]) * rng.uniform(30, 80), rng.normal(0, 3), 0.0]); tag = 'uvclamp'
            elif kind == 6:    # huge weights on a few points: ill-conditioned, tiny steps
                o['w'][rng.integers(0, P, 5)] *= np.float32(1e4)
                o['init'] = gt[i] + rng.normal(0, 1, 4) * np.array([0.3, 1.0, 0.3, 3.0]); tag = 'illcond'
            else:              # tight clip window: most projections clamped
                o['ur'] = np.array([500.0, 520.0], np.float32); o['vr'] = np.array([150.0, 160.0], np.float32)
                o['init'] = gt[i] + rng.normal(0, 1, 4) * np.array([0.3, 1.0, 0.3, 3.0]); tag = 'window'
            yield tag, o
    # hand-made degenerate objects
    b = syn.make_batch(B=4, hw=HW, seed=9)
    x2d, istd, x3d, K, ur, vr, thr = [np.ascontiguousarray(a) for a in syn.pnp_boundary(b, planar=False)]
    gt = np.concatenate([b['gt_yaw'][:, None], b['gt_t']], 1)
    base = lambda i: dict(x2d=x2d[i].copy(), w=istd[i].copy(), x3d=x3d[i].copy(), K=K[0].copy(), ur=ur[0].copy(), vr=vr[0].copy(), init=gt[i] + 0.1)
    o = base(0); o['w'][:] = 0; yield 'zero_weights', o                     # J = 0: gradient tolerance at iteration 0
    o = base(1); o['x3d'][7, 0] = np.nan; yield 'nan_point', o               # evaluation failure
    o = base(2); o['w'][:] = np.float32(1e18); yield 'huge_weights', o       # r^2 ~ 1e40: H ~ 1e80 -> steps of non-finite quality
    o = base(3); o['x3d'][:] = o['x3d'][0]; o['x2d'][:] = o['x2d'][0]; yield 'one_point', o     # rank-deficient: 196 copies of one correspondence


def main():
    picked, seen = [], {}
    want = {'why1': 3, 'why2': 8, 'why3': 14, 'why4': 6, 'why5': 2, 'why6': 3, 'why7': 1, 'rejected': 10, 'invalid': 4, 'long': 6}
    pool, n_pool, diverge = [], 0, {}
    for tag, o in candidates():
        r = solve(o, qr=True, trace=True)
        rc = solve(o, qr=False)
        n_pool += 1
        if rc['iters'] != r['iters'] or rc['why'] != r['why'] or (r['val'] and np.abs(rc['pose'] - r['pose']).max() > 1e-7 * max(1.0, np.abs(r['pose']).max())):
            # normal equations and QR part ways only where J S is numerically rank-deficient (cond^2 > 1/eps): initial poses
            # behind the camera, where the z clamp removes a column's information.  Counted, reported, kept out of the fixture.
            diverge[tag] = diverge.get(tag, 0) + 1
            continue
        t = r['trace']
        feats = {f'why{r["why"]}'}
        if len(t) and (t[:, 7] == 0).any():
            feats.add('rejected')
        if len(t) and (t[:, 7] == -1).any():
            feats.add('invalid')
        if r['iters'] >= 12:
            feats.add('long')
        pool.append((tag, o, r, feats))
    # the hand-made degenerate objects first, then a greedy cover of the wanted features, then ordinary objects
    for tag, o, r, feats in pool:
        if tag in ('zero_weights', 'nan_point', 'huge_weights', 'one_point'):
            picked.append((tag, o, r))
            for f in feats:
                seen[f] = seen.get(f, 0) + 1
    for tag, o, r, feats in pool:
        if any(o is p[1] for p in picked):
            continue
        need = [f for f in feats if seen.get(f, 0) < want.get(f, 0)]
        if need and len(picked) < N_OBJ - 8:
            picked.append((tag, o, r))
            for f in feats:
                seen[f] = seen.get(f, 0) + 1
    for tag, o, r, feats in pool:
        if len(picked) >= N_OBJ:
            break
        if not any(o is p[1] for p in picked) and tag in ('mild', 'zero_weights', 'nan_point', 'huge_weights', 'one_point'):
            picked.append((tag, o, r))
    picked = picked[:N_OBJ]
    assert len(picked) == N_OBJ
    n = N_OBJ
    out = dict(x2d=np.stack([p[1]['x2d'] for p in picked]), w=np.stack([p[1]['w'] for p in picked]), x3d=np.stack([p[1]['x3d'] for p in picked]),
               K=np.stack([p[1]['K'] for p in picked]), ur=np.stack([p[1]['ur'] for p in picked]), vr=np.stack([p[1]['vr'] for p in picked]),
               init=np.stack([p[1]['init'] for p in picked]), tag=np.array([p[0] for p in picked]))
    trace = np.full((n, MAXP, len(orc.TRACE_FIELDS)), np.nan)
    for i, (tag, o, r) in enumerate(picked):
        t = r['trace']
        assert len(t) <= MAXP
        trace[i, :len(t)] = t
        # the Cholesky mode must walk the same path (what the kernel implements)
        rc = solve(o, qr=False, trace=True)
        assert rc['iters'] == r['iters'] and rc['why'] == r['why'], (i, tag, rc['iters'], r['iters'], rc['why'], r['why'])
    out.update(trace=trace, n_pass=np.array([len(p[2]['trace']) for p in picked], np.int32), iters=np.array([p[2]['iters'] for p in picked], np.int32),
               why=np.array([p[2]['why'] for p in picked], np.int32), val=np.array([p[2]['val'] for p in picked], np.int32),
               pose=np.stack([p[2]['pose'] for p in picked]), radius=np.array([p[2]['tr'] for p in picked]),
               final_cost=np.array([p[2]['final_cost'] for p in picked]), trace_fields=np.array(orc.TRACE_FIELDS),
               pool_size=np.int32(n_pool), pool_divergent_qr_vs_cholesky=np.array(sorted(diverge.items()), dtype=object).astype(str))
    path = os.path.join(ROOT, 'tests', 'golden', 'g7_lm_trajectories.npz')
    np.savez_compressed(path, **out)
    why, cnt = np.unique(out['why'], return_counts=True)
    print('wrote', path, os.path.getsize(path), 'bytes;  exit reasons', dict(zip(why.tolist(), cnt.tolist())),
          ' objects with rejected steps', int(((trace[:, :, 7] == 0).any(1)).sum()), ' with invalid steps', int(((trace[:, :, 7] == -1).any(1)).sum()),
          ' iterations max', int(out['iters'].max()), ' candidate pool', n_pool, ' QR/Cholesky divergent (excluded)', diverge, ' tags', dict(zip(*[a.tolist() for a in np.unique(out['tag'], return_counts=True)])))


def main_b():
    """Golden fixture G7b — the candidates G7 leaves OUT: starts where the Jacobi-scaled J is numerically rank-deficient (initial
    pose at / behind the camera: every point z-clamped, a column without information; or > 10 m off), so that the LM's two step
    solvers part ways — Ceres' DENSE_QR on [J S; D] versus the Cholesky factorisation of the normal equations that the HIP kernel
    uses.  For these inputs the kernel can only be pinned to its OWN step solver: the fixture freezes the oracle's CHOLESKY-mode
    trajectories (same columns as G7) and records, per object, what the QR mode returns (iterations, exit reason, pose), so that the
    gap is a committed number (profiles/r03_g7b_qr_vs_cholesky.txt is printed from it)."""
    picked = []
    n_pool = 0
    for tag, o in candidates():
        rq = solve(o, qr=True)
        rc = solve(o, qr=False, trace=True)
        n_pool += 1
        if rc['iters'] != rq['iters'] or rc['why'] != rq['why'] or (rq['val'] and np.abs(rc['pose'] - rq['pose']).max() > 1e-7 * max(1.0, np.abs(rq['pose']).max())):
            picked.append((tag, o, rc, rq))
    n = len(picked)
    out = dict(x2d=np.stack([p[1]['x2d'] for p in picked]), w=np.stack([p[1]['w'] for p in picked]), x3d=np.stack([p[1]['x3d'] for p in picked]),
               K=np.stack([p[1]['K'] for p in picked]), ur=np.stack([p[1]['ur'] for p in picked]), vr=np.stack([p[1]['vr'] for p in picked]),
               init=np.stack([p[1]['init'] for p in picked]), tag=np.array([p[0] for p in picked]))
    trace = np.full((n, MAXP, len(orc.TRACE_FIELDS)), np.nan)
    for i, (tag, o, rc, rq) in enumerate(picked):
        assert len(rc['trace']) <= MAXP
        trace[i, :len(rc['trace'])] = rc['trace']
    out.update(trace=trace, n_pass=np.array([len(p[2]['trace']) for p in picked], np.int32), iters=np.array([p[2]['iters'] for p in picked], np.int32),
               why=np.array([p[2]['why'] for p in picked], np.int32), val=np.array([p[2]['val'] for p in picked], np.int32),
               pose=np.stack([p[2]['pose'] for p in picked]), radius=np.array([p[2]['tr'] for p in picked]),
               final_cost=np.array([p[2]['final_cost'] for p in picked]), trace_fields=np.array(orc.TRACE_FIELDS),
               qr_iters=np.array([p[3]['iters'] for p in picked], np.int32), qr_why=np.array([p[3]['why'] for p in picked], np.int32),
               qr_val=np.array([p[3]['val'] for p in picked], np.int32), qr_pose=np.stack([p[3]['pose'] for p in picked]),
               qr_final_cost=np.array([p[3]['final_cost'] for p in picked]), pool_size=np.int32(n_pool))
    path = os.path.join(ROOT, 'tests', 'golden', 'g7b_lm_rank_deficient_starts.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', n, 'objects of a pool of', n_pool)
    report(out)


def report(z):
    """per-object gap between the two step solvers on the G7b objects (the text committed as profiles/r03_g7b_qr_vs_cholesky.txt)"""
    print('# G7b: LM starts where Cholesky (normal equations; the HIP kernel) and DENSE_QR (Ceres) part ways — oracle, fp64')
    print('# obj tag        iters(chol/qr) exit(chol/qr) valid(chol/qr)  max|pose_chol - pose_qr|   cost_chol        cost_qr')
    gaps = []
    for i in range(len(z['iters'])):
        gap = float(np.abs(z['pose'][i] - z['qr_pose'][i]).max())
        gaps.append(gap)
        print(f"{i:4d} {str(z['tag'][i]):10s} {int(z['iters'][i]):3d}/{int(z['qr_iters'][i]):<3d}        {int(z['why'][i])}/{int(z['qr_why'][i])}           {int(z['val'][i])}/{int(z['qr_val'][i])}        "
              f"{gap:12.4e}          {float(z['final_cost'][i]):12.6e} {float(z['qr_final_cost'][i]):12.6e}")
    gaps = np.array(gaps)
    same_exit = (z['why'] == z['qr_why']) & (z['iters'] == z['qr_iters'])
    lower = z['final_cost'] <= z['qr_final_cost'] * (1 + 1e-9)
    print(f'# {len(gaps)} objects; identical iteration count and exit reason: {int(same_exit.sum())}; pose gap median {np.median(gaps):.3e}, p90 {np.percentile(gaps, 90):.3e}, '
          f'max {gaps.max():.3e}; objects within 1e-4: {int((gaps <= 1e-4).sum())}; final cost of the Cholesky mode <= the QR mode\'s in {int(lower.sum())} objects')


if __name__ == '__main__':
    if '--b' in sys.argv:
        main_b()
    elif '--report-b' in sys.argv:
        report(dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g7b_lm_rank_deficient_starts.npz'), allow_pickle=True)))
    else:
        main()
