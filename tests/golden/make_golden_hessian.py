#!/usr/bin/env python3
"""G8: the reference's exact Hessian (monorun/ops/least_squares/hessian.py:5-64, `forward_exact_hessian=True`).

Runs only in the authoring container (needs /root/reference, read-only).  The fixture is data: seeded inputs and the
matrices the reference's own code returned for them.

    python tests/golden/make_golden_hessian.py         # rewrites tests/golden/g8_exact_hessian.npz

exact_hessian no longer runs on torch >= 2 as it stands: forward_proj writes in place into an output of Tensor.split
under autograd (jacobian.py:27-29), which current autograd refuses ("output of a function that returns multiple views").
The arithmetic is unaffected by whether split hands out views or copies, so this script gives Tensor.split copy semantics
WHILE the reference function runs (a shim in this process only; nothing of the reference is modified or copied) and
checks the shim on the non-autograd path: approx_hessian with and without it is bit-identical.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/monorun'
OUT = os.path.dirname(os.path.abspath(__file__))


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


class split_returns_copies:
    def __enter__(self):
        self.orig = torch.Tensor.split
        orig = self.orig
        torch.Tensor.split = lambda t, *a, **k: tuple(x.clone() for x in orig(t, *a, **k))

    def __exit__(self, *exc):
        torch.Tensor.split = self.orig


def main():
    pk = types.ModuleType('refls')
    pk.__path__ = []
    sys.modules['refls'] = pk
    _load('refls.jacobian', 'ops/least_squares/jacobian.py')
    hes = _load('refls.hessian', 'ops/least_squares/hessian.py')

    rng = np.random.default_rng(808)
    B, P = 10, 96
    K1 = np.array([[707.0912, 0.0, 601.8873], [0.0, 707.0912, 183.1104], [0.0, 0.0, 1.0]])
    dims = np.array([3.89, 1.53, 1.62])
    yaw = rng.uniform(-np.pi, np.pi, (B, 1))
    t = np.stack([rng.uniform(-8, 8, B), rng.uniform(1, 2, B), rng.uniform(6, 40, B)], 1)
    x3d = (rng.uniform(-0.5, 0.5, (B, P, 3)) + np.array([0.0, -0.5, 0.0])) * dims
    c, s = np.cos(yaw[:, 0]), np.sin(yaw[:, 0])
    R = np.zeros((B, 3, 3)); R[:, 0, 0] = c; R[:, 0, 2] = s; R[:, 1, 1] = 1; R[:, 2, 0] = -s; R[:, 2, 2] = c
    Xc = np.einsum('bij,bpj->bpi', R, x3d) + t[:, None]
    uvz = np.einsum('ij,bpj->bpi', K1, Xc)
    x2d = uvz[..., :2] / uvz[..., 2:] + rng.normal(0, 1.5, (B, P, 2))            # residuals of a few pixels: the second-order term matters
    istd = np.exp(-rng.normal(np.log(2.0), 0.5, (B, P, 2))) / 10.0
    mask = rng.random((B, P)) > 0.25
    # the pose the Hessian is evaluated at: near the generating pose, and for some objects such that clips occur
    yaw_e = yaw + rng.normal(0, 0.05, (B, 1))
    t_e = t + rng.normal(0, 0.2, (B, 3))
    t_e[1, 2] = 1.2                      # object 1: close to the camera -> some points behind z_min (both rows masked)
    t_e[2] = [11.5, 1.5, 10.0]           # object 2: at the right border -> u beyond u_max for part of the points (u row masked only)
    t_e[3] = [0.0, 6.2, 10.0]            # object 3: at the lower border -> v beyond v_max for part of the points (v row masked only)
    t_e[9] = [30.0, 1.5, 10.0]                 # object 9: every u row clipped -> tx unobservable, h exactly singular (inverse raises in the reference)
    mask[4] = True                       # object 4: every point an inlier
    mask[5, 6:] = False                  # object 5: six inliers only
    K = np.repeat(K1[None], B, 0)
    K[6, 0, 1] = 3.0                     # object 6: skewed camera
    K[7, 2] = [1e-4, -2e-4, 1.001]       # object 7: third row not (0,0,1): the analytic expressions are differentiated as they stand
    u_range = np.repeat(np.array([[-200.0, 1442.0]]), B, 0)
    v_range = np.repeat(np.array([[-200.0, 575.0]]), B, 0)
    u_range[8] = [550.0, 700.0]          # object 8: tight ranges -> many clipped rows
    v_range[8] = [150.0, 220.0]
    z_min = 0.5

    def T(a, dt):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dt)

    out = dict(x2d=x2d, istd=istd, x3d=x3d, K=K, u_range=u_range, v_range=v_range, yaw=yaw_e, t=t_e, mask=mask, z_min=np.float64(z_min))
    for name, dt in (('f64', torch.float64), ('f32', torch.float32)):
        args = [T(x2d, dt), T(istd, dt), T(x3d, dt), T(K, dt), T(u_range, dt), T(v_range, dt), z_min, T(yaw_e, dt), T(t_e, dt),
                torch.from_numpy(mask)]
        ha0 = hes.approx_hessian(*[a.clone() if torch.is_tensor(a) else a for a in args])
        with split_returns_copies():
            ha1 = hes.approx_hessian(*[a.clone() if torch.is_tensor(a) else a for a in args])
            he = hes.exact_hessian(*[a.clone() if torch.is_tensor(a) else a for a in args])
            # broadcast forms the pipeline uses: one camera, one range pair
            he_b = hes.exact_hessian(args[0][:4].clone(), args[1][:4].clone(), args[2][:4].clone(), args[3][:1].clone(), args[4][:1].clone(),
                                     args[5][:1].clone(), z_min, args[7][:4].clone(), args[8][:4].clone(), args[9][:4].clone())
            he_nomask = hes.exact_hessian(*[a.clone() if torch.is_tensor(a) else a for a in args[:9]], None)
        torch.set_grad_enabled(True)
        assert torch.equal(ha0, ha1), 'the split shim changed the arithmetic'
        assert torch.equal(he_b, he[:4])
        out['h_exact_' + name] = he.detach().numpy()
        out['h_approx_' + name] = ha0.detach().numpy()
        out['h_exact_nomask_' + name] = he_nomask.detach().numpy()
        if name == 'f64':
            out['cov_exact_f64'] = torch.inverse(he.detach()[:9]).numpy()          # object 9 is singular
            d = (he - ha0).abs().amax(dim=(1, 2)) / ha0.abs().amax(dim=(1, 2))
            print('relative size of the second-order term per object:', np.round(d.numpy(), 4))
            print('asymmetry:', float((he - he.transpose(1, 2)).abs().max()))
    np.savez_compressed(os.path.join(OUT, 'g8_exact_hessian.npz'), **out)
    print('wrote g8_exact_hessian.npz', {k: v.shape for k, v in out.items() if hasattr(v, 'shape')})


if __name__ == '__main__':
    main()
