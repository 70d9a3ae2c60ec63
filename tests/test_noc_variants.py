"""N4: pnp_noc_uncert / pnp_noc_cov_uncert (ext.h:15-43) — exported by the reference, never
called by it, implemented for ABI completeness.  No reference oracle exists (Ceres absent): the CPU
restatement is pinned by finite differences of its own robust cost and by scipy; the GPU entry points are
compared with it through the reference's exact C signatures."""
import ctypes

import numpy as np
import pytest
from scipy.optimize import minimize

from monorun_amd import synthetic as syn


def _problem(full_cov, seed=0, n=200, outliers=10):
    c = syn.cube_config1(n_points=n, seed=2 + seed)
    dims = np.array([3.89, 1.53, 1.62])
    noc = c['pts3d'] / dims
    rng = np.random.default_rng(seed)
    p2 = c['pts2d'] + rng.normal(0, 1.0, c['pts2d'].shape)
    p2[:outliers] += 40.0                                   # gross outliers: the Huber branch is active
    w = rng.uniform(0.3, 0.8, (n, 2))
    if full_cov:
        w = np.stack([w[:, 0], rng.uniform(-0.1, 0.1, n), w[:, 1]], 1)
    logdim = np.log(dims) + np.array([0.05, -0.03, 0.02])
    lw = np.array([5.0, 4.0, 6.0])
    init = np.concatenate([logdim, c['gt_pose'] + np.array([0.1, 0.3, 0.1, 1.0])])
    return dict(p2=p2, noc=noc, w=w, logdim=logdim, lw=lw, K=c['K'], init=init, clips=c['clips'], delta=2.0,
                gt=np.concatenate([np.log(dims), c['gt_pose']]))


@pytest.mark.parametrize('full_cov', [False, True])
def test_oracle_gradient_is_derivative_of_robust_cost(orc, full_cov):
    q = _problem(full_cov)
    f = lambda x: orc.noc_cost_grad(q['p2'], q['noc'], q['w'], q['logdim'], q['lw'], q['K'], x, q['clips'], q['delta'], full_cov)
    for x in (q['init'], q['gt'] + 0.01):
        ok, c0, g, H = f(x)
        assert ok and np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > -1e-9)
        fd = np.array([(f(x + 1e-6 * np.eye(7)[k])[1] - f(x - 1e-6 * np.eye(7)[k])[1]) / 2e-6 for k in range(7)])
        assert np.abs(g - fd).max() <= 1e-5 * max(1.0, np.abs(g).max())


@pytest.mark.parametrize('full_cov', [False, True])
def test_oracle_lm_reaches_the_robust_minimum(orc, full_cov):
    q = _problem(full_cov)
    r = orc.pnp_noc(q['p2'], q['noc'], q['w'], q['logdim'], q['lw'], q['K'], q['init'], q['clips'], q['delta'], full_cov)
    assert r['val'] == 1 and r['final_cost'] < r['initial_cost'] and 1 <= r['iters'] <= 50
    f = lambda x: orc.noc_cost_grad(q['p2'], q['noc'], q['w'], q['logdim'], q['lw'], q['K'], x, q['clips'], q['delta'], full_cov)[1]
    m = minimize(f, r['dimpose'], method='BFGS', options=dict(gtol=1e-10))
    assert 0 <= (r['final_cost'] - m.fun) / m.fun < 1e-5            # function_tolerance 1e-6 on the discarded step
    assert np.abs(r['dimpose'] - q['gt']).max() < 0.6               # sane estimate despite 5 % gross outliers
    # without outliers and with a huge delta the Huber loss is inactive: plain least squares, GT recovered
    q2 = _problem(full_cov, outliers=0)
    r2 = orc.pnp_noc(q2['p2'], q2['noc'], q2['w'], q2['logdim'], q2['lw'], q2['K'], q2['init'], q2['clips'], 1e9, full_cov)
    assert r2['val'] == 1 and np.abs(r2['dimpose'][3:] - q2['gt'][3:]).max() < 0.6    # depth trades against the (5 % off) log-dims prior


def _call(lib, name, q):
    dp = ctypes.POINTER(ctypes.c_double)
    arrs = [np.ascontiguousarray(q[k], np.float64) for k in ('p2', 'noc', 'w', 'logdim', 'lw', 'K', 'init')]
    clips = np.ascontiguousarray(q['clips'], np.float64)
    val = np.zeros(1, np.int32)
    out = np.zeros(7)
    getattr(lib, name)(*[a.ctypes.data_as(dp) for a in arrs], val.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                       out.ctypes.data_as(dp), q['p2'].shape[0], clips.ctypes.data_as(dp), ctypes.c_double(q['delta']))
    return int(val[0]), out


@pytest.mark.gpu
@pytest.mark.parametrize('full_cov', [False, True])
def test_gpu_entry_points_match_oracle(orc, full_cov):
    from monorun_amd import _lib
    lib = _lib.load()
    name = 'pnp_noc_cov_uncert' if full_cov else 'pnp_noc_uncert'
    for seed in range(3):
        q = _problem(full_cov, seed=seed, n=150 + 90 * seed)
        val, out = _call(lib, name, q)
        r = orc.pnp_noc(q['p2'], q['noc'], q['w'], q['logdim'], q['lw'], q['K'], q['init'], q['clips'], q['delta'], full_cov)
        assert val == r['val'] == 1
        assert np.abs(out - r['dimpose']).max() <= 1e-6, (seed, np.abs(out - r['dimpose']).max())
    # NaN input: evaluation failure -> not usable, dimpose = init (memcpy at pnp_uncert_cpu.cpp:309)
    q = _problem(full_cov)
    q['noc'] = q['noc'].copy(); q['noc'][3, 1] = np.nan
    val, out = _call(lib, name, q)
    assert val == 0 and np.array_equal(out, q['init'])


@pytest.mark.gpu
@pytest.mark.parametrize('full_cov', [False, True])
def test_batched_entry_point_matches_oracle_object_by_object(orc, full_cov):
    """mr_pnp_noc_batched: B objects in one launch (one workgroup each, device fp64 buffers) against the oracle's solve of
    every object — dimpose within 1e-6, identical LM iteration counts; shared K / clips and per-object K / clips."""
    import torch
    from monorun_amd import _lib
    lib = _lib.load()
    dev = torch.device('cuda:0')
    B, n = 24, 200
    qs = [_problem(full_cov, seed=100 + i, n=n) for i in range(B)]
    cat = lambda k: torch.from_numpy(np.stack([np.asarray(q[k], np.float64) for q in qs])).to(dev).contiguous()
    p2, noc, w, logdim, lw, init = [cat(k) for k in ('p2', 'noc', 'w', 'logdim', 'lw', 'init')]
    for per_object in (False, True):
        K = cat('K').reshape(B, 9) if per_object else torch.from_numpy(np.asarray(qs[0]['K'], np.float64).reshape(1, 9)).to(dev)
        clips = cat('clips') if per_object else torch.from_numpy(np.asarray(qs[0]['clips'], np.float64).reshape(1, 5)).to(dev)
        out = torch.zeros(B, 7, dtype=torch.float64, device=dev); val = torch.zeros(B, dtype=torch.int32, device=dev)
        diag = torch.zeros(B, 2, dtype=torch.float64, device=dev)
        _lib.check(lib.mr_pnp_noc_batched(int(full_cov), p2.data_ptr(), noc.data_ptr(), w.data_ptr(), logdim.data_ptr(), lw.data_ptr(), K.data_ptr(),
                                          K.shape[0], init.data_ptr(), clips.data_ptr(), clips.shape[0], float(qs[0]['delta']), B, n,
                                          out.data_ptr(), val.data_ptr(), diag.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        for i, q in enumerate(qs):
            r = orc.pnp_noc(q['p2'], q['noc'], q['w'], q['logdim'], q['lw'], q['K'], q['init'], q['clips'], q['delta'], full_cov)
            assert int(val[i]) == r['val'] == 1 and np.abs(out[i].cpu().numpy() - r['dimpose']).max() <= 1e-6, i
            assert int(diag[i, 0]) == r['iters'], (i, int(diag[i, 0]), r['iters'])
