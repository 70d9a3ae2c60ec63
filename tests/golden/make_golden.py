#!/usr/bin/env python3
"""Generate the committed golden fixtures by IMPORTING the reference's importable pieces.

Runs only in the authoring container (needs /root/reference, read-only).  The fixtures are data:
seeded inputs + the outputs the reference's own code produced for them.  No reference source is
copied; nothing here runs on the GPU box.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is imported from /root/reference/monorun (by file path, under a synthetic package; the real
package __init__ cannot be imported — it pulls the unbuilt cffi `_ext`, cv2, mmdet):
  ops/least_squares/jacobian.py, hessian.py ................ G1, G2  (R2, R7 of SURVEY.md §8a)
  core/bbox_3d/{coord,dim,proj_error}_coder/*.py ........... G3      (R10, R11, R13) via an mmcv.utils stub
  models/.../dense_decoders/fcn_noc_decoder.py::slice_pred . G3      (R9)  via mmcv/mmdet stubs
  models/.../optimizers/uncert_prop_pnp_optimizer.py ....... G4      (R8)  via mmdet stubs + a recording PnP
  core/bbox_3d/iou_calculators/rotate_iou_kernel.py ........ G5      (N1)  rotated IoU device functions via a numba stub
  core/evaluation/kitti_utils/{eval,rotate_iou}.py ........ G6      (N2)  KITTI evaluator as plain Python under a numba stub
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference/monorun'
OUT = os.path.dirname(os.path.abspath(__file__))


# ------------------------------------------------------------------ stubbed import machinery ---
def _pkg(name):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
    return m


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


class _Registry:
    def __init__(self, name):
        self.name, self.d = name, {}

    def register_module(self, *a, **k):
        def deco(cls):
            self.d[cls.__name__] = cls
            return cls
        return deco


def _build_from_cfg(cfg, reg, default_args=None):
    cfg = dict(cfg)
    cls = reg.d[cfg.pop('type')]
    if default_args:
        cfg.update(default_args)
    return cls(**cfg)


def _identity_deco(*a, **k):
    def deco(f):
        return f
    return deco


def install_stubs():
    mmcv = _pkg('mmcv')
    mu = _pkg('mmcv.utils'); mu.Registry = _Registry; mu.build_from_cfg = _build_from_cfg
    mc = _pkg('mmcv.cnn'); mc.ConvModule = mc.build_upsample_layer = mc.build_plugin_layer = object
    mo = _pkg('mmcv.ops'); mo.Conv2d = object
    moc = _pkg('mmcv.ops.carafe'); moc.CARAFEPack = object
    mmcv.utils, mmcv.cnn, mmcv.ops = mu, mc, mo
    md = _pkg('mmdet'); mdc = _pkg('mmdet.core')
    mdc.auto_fp16 = _identity_deco; mdc.force_fp32 = _identity_deco; mdc.multi_apply = None
    mdm = _pkg('mmdet.models'); mdb = _pkg('mmdet.models.builder')
    mdb.HEADS = _Registry('heads'); mdb.build_loss = lambda cfg: None
    md.core, md.models = mdc, mdm
    return mdb.HEADS


def load_reference():
    heads = install_stubs()
    for p in ('monorun', 'monorun.ops', 'monorun.ops.least_squares', 'monorun.core', 'monorun.core.bbox_3d',
              'monorun.core.bbox_3d.coord_coder', 'monorun.core.bbox_3d.dim_coder', 'monorun.core.bbox_3d.proj_error_coder',
              'monorun.models', 'monorun.models.roi_heads', 'monorun.models.roi_heads.bbox_3d_heads',
              'monorun.models.roi_heads.bbox_3d_heads.optimizers', 'monorun.models.roi_heads.bbox_3d_heads.dense_decoders'):
        _pkg(p)
    jac = _load('monorun.ops.least_squares.jacobian', 'ops/least_squares/jacobian.py')
    hes = _load('monorun.ops.least_squares.hessian', 'ops/least_squares/hessian.py')
    bld = _load('monorun.core.bbox_3d.builder', 'core/bbox_3d/builder.py')
    noc = _load('monorun.core.bbox_3d.coord_coder.noc_coder', 'core/bbox_3d/coord_coder/noc_coder.py')
    dim = _load('monorun.core.bbox_3d.dim_coder.multiclass_norm_dim_coder', 'core/bbox_3d/dim_coder/multiclass_norm_dim_coder.py')
    prj = _load('monorun.core.bbox_3d.proj_error_coder.distance_invar_proj_error_coder',
                'core/bbox_3d/proj_error_coder/distance_invar_proj_error_coder.py')
    core = sys.modules['monorun.core']
    core.build_rotation_coder = lambda cfg: None
    core.bbox3d_overlaps_aligned_torch = None
    core.build_coord_coder = bld.build_coord_coder
    core.masked_dense_target = None
    # recording PnP behind the reference's own build_pnp call site
    rec = {}

    class RecorderPnP(torch.nn.Module):
        def __init__(self, **kw):
            super().__init__()
            rec['cfg'] = kw

        def forward(self, c2d, istd, c3d, cam, u_range, v_range, thr=None):
            rec['args'] = (c2d, istd, c3d, cam, u_range, v_range, thr)
            bn = c2d.size(0)
            g = torch.Generator().manual_seed(7)
            a = torch.randn(bn, 4, 4, generator=g, dtype=torch.float64).to(c2d.dtype)
            cov = a @ a.transpose(1, 2) + torch.eye(4, dtype=c2d.dtype)
            rec['cov'] = cov
            return (torch.ones(bn, dtype=torch.bool), torch.zeros(bn, 1), torch.zeros(bn, 3), cov, None)

    sys.modules['monorun.ops'].build_pnp = lambda cfg: RecorderPnP(**{k: v for k, v in cfg.items() if k != 'type'})
    opt = _load('monorun.models.roi_heads.bbox_3d_heads.optimizers.uncert_prop_pnp_optimizer',
                'models/roi_heads/bbox_3d_heads/optimizers/uncert_prop_pnp_optimizer.py')
    dec = _load('monorun.models.roi_heads.bbox_3d_heads.dense_decoders.fcn_noc_decoder',
                'models/roi_heads/bbox_3d_heads/dense_decoders/fcn_noc_decoder.py')
    return dict(jac=jac, hes=hes, noc=noc, dim=dim, prj=prj, opt=opt, dec=dec, rec=rec)


# ------------------------------------------------------------------------------- G1 / G2 -------
def make_g1_g2(ref):
    rng = np.random.default_rng(20240601)
    B, P = 7, 784
    W, Himg = 1242.0, 375.0
    K0 = np.array([[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]])
    K = np.repeat(K0[None], B, 0)
    K[5, 0, 1] = 3.5          # object 5: skewed K (full-K path of forward_proj)
    K[5, 1, 0] = -1.25
    yaw = np.array([0.3, -1.2, 2.0, 0.7, -2.6, 1.1, 0.5])
    t = np.array([[1.0, 1.5, 15.0], [0.4, 1.4, 1.6], [-6.8, 1.6, 6.0], [3.0, 1.7, 30.0], [-5.0, 1.2, 22.0], [2.0, 1.5, 12.0],
                  [-14.0, 1.6, 6.0]])   # object 6: every u beyond the border -> singular J^T J
    dims = np.array([3.89, 1.53, 1.62])
    X = np.empty((B, P, 3))
    X[..., 0] = rng.uniform(-0.5, 0.5, (B, P)) * dims[0]
    X[..., 1] = rng.uniform(-1.0, 0.0, (B, P)) * dims[1]
    X[..., 2] = rng.uniform(-0.5, 0.5, (B, P)) * dims[2]
    # project with the GT pose, add pixel noise -> observations
    c, s = np.cos(yaw), np.sin(yaw)
    Xc = c[:, None] * X[..., 0] + s[:, None] * X[..., 2] + t[:, None, 0]
    Yc = X[..., 1] + t[:, None, 1]
    Zc = -s[:, None] * X[..., 0] + c[:, None] * X[..., 2] + t[:, None, 2]
    Zs = np.maximum(Zc, 0.5)
    x2d = np.stack([K[:, None, 0, 0] * Xc / Zs + K[:, None, 0, 2], K[:, None, 1, 1] * Yc / Zs + K[:, None, 1, 2]], -1)
    x2d = np.clip(x2d, -400, 2000) + rng.normal(0, 2.0, (B, P, 2))
    istd = np.exp(-rng.normal(np.log(2.0), 0.5, (B, P, 2))) / 10.0
    u_range = np.repeat(np.array([[-200.0, W + 200]]), B, 0)
    v_range = np.repeat(np.array([[-200.0, Himg + 200]]), B, 0)
    inlier = np.ones((B, P), bool)
    inlier[3] = rng.uniform(size=P) > 0.4                      # object 3: partial inlier mask
    # evaluate slightly away from GT so that residuals are non-trivial
    yaw_e = yaw + rng.normal(0, 0.05, B)
    t_e = t + rng.normal(0, 0.1, (B, 3))
    z_min = 0.5
    out = dict(x2d=x2d, istd=istd, x3d=X, K=K, u_range=u_range, v_range=v_range, inlier=inlier,
               yaw=yaw_e, t=t_e, z_min=np.float64(z_min))
    for tag, dt in (('f64', torch.float64), ('f32', torch.float32)):
        tt = lambda a: torch.tensor(a, dtype=dt)
        args = (tt(x2d), tt(istd), tt(X), tt(K), tt(u_range), tt(v_range), z_min, tt(yaw_e)[:, None], tt(t_e), torch.tensor(inlier))
        jt, jy, err = ref['jac'].get_jacobian_and_error(*args)
        h = ref['hes'].approx_hessian(*args)
        cov = torch.inverse(h[:6])          # object 6 is singular by construction
        # also the un-masked variant (inlier_mask=None)
        h_nomask = ref['hes'].approx_hessian(*args[:-1], None)
        if tag == 'f64':
            out['jac_t_f64'] = jt.numpy(); out['jac_yaw_f64'] = jy.numpy(); out['err_f64'] = err.numpy()
        out['h_' + tag] = h.numpy(); out['cov_' + tag] = cov.numpy(); out['h_nomask_' + tag] = h_nomask.numpy()
    # broadcast (1,3,3)/(1,2) forms on objects 0..4 (un-skewed K)
    tt = lambda a: torch.tensor(a, dtype=torch.float64)
    hb = ref['hes'].approx_hessian(tt(x2d[:5]), tt(istd[:5]), tt(X[:5]), tt(K[:1]), tt(u_range[:1]), tt(v_range[:1]), z_min,
                                   tt(yaw_e[:5])[:, None], tt(t_e[:5]), torch.tensor(inlier[:5]))
    out['h_bcast_f64'] = hb.numpy()
    # sanity: which special cases are actually exercised
    out['n_zclip'] = np.array([(Zc[b] < z_min).sum() for b in range(B)])
    np.savez_compressed(os.path.join(OUT, 'g1_g2_jacobian_hessian.npz'), **out)
    uproj = K[:, None, 0, 0] * Xc / Zs + K[:, None, 0, 2]
    out['n_uclip'] = np.array([((uproj[b] < -200) | (uproj[b] > W + 200)).sum() for b in range(B)])
    np.savez_compressed(os.path.join(OUT, 'g1_g2_jacobian_hessian.npz'), **out)
    print('G1/G2: z-clipped', out['n_zclip'], 'u-clipped', out['n_uclip'], '| cond(h)', np.linalg.cond(out['h_f64'][:6]).round(0))


# ------------------------------------------------------------------------------- G3 ------------
def make_g3(ref):
    rng = np.random.default_rng(7)
    B, C, h, w = 8, 3, 28, 28
    all_pred = rng.normal(0, 1, (B, 2 * C * 5, h, w)).astype(np.float32)
    labels = np.array([0, 1, 2, 0, 2, 1, 0, 0])
    flip = np.array([False, True, False, True, True, False, False, True])
    dim = rng.normal(0, 1, (B, 3)).astype(np.float32)
    dim_var = (rng.uniform(0.01, 0.2, (B, 3)) ** 2).astype(np.float32)
    ap = torch.tensor(all_pred)
    # flip branch: literal restatement of fcn_noc_decoder.py:225-235 (inside forward(), not separable)
    v = ap.view(B, 2, ap.size(1) // 2, h, w)
    inds = torch.arange(0, B, dtype=torch.long)
    sel = v[inds, inds.new_tensor(flip)]
    fake = types.SimpleNamespace(class_agnostic=False, num_classes=C, noc_channels=3, uncert_channels=2)
    noc_pred, noc_var, proj_logstd = ref['dec'].FCNNOCDecoder.slice_pred(fake, sel, torch.tensor(labels))
    assert noc_var is None
    fake_ag = types.SimpleNamespace(class_agnostic=True, num_classes=1, noc_channels=3, uncert_channels=2)
    ag_pred = torch.tensor(all_pred[:, :10])
    v_ag = ag_pred.view(B, 2, 5, h, w)[inds, inds.new_tensor(flip)]
    noc_ag, _, logstd_ag = ref['dec'].FCNNOCDecoder.slice_pred(fake_ag, v_ag, torch.tensor(labels))
    dim_coder = ref['dim'].MultiClassNormDimCoder()
    noc_coder = ref['noc'].NOCCoder()
    prj_coder = ref['prj'].DistanceInvarProjErrorCoder(ref_length=1.6, ref_focal_y=722, target_std=0.15)
    assert abs(prj_coder.scaling_denomitor - 173.28) < 1e-9
    dims, dims_var = dim_coder.decode(torch.tensor(dim), torch.tensor(dim_var), torch.tensor(labels))
    c3d, c3d_var = noc_coder.decode(noc_pred, None, dims, dims_var, False)
    logstd_px = prj_coder.decode_logstd(proj_logstd, c3d_var, None)
    c3d_nv, c3d_var_nv = noc_coder.decode(noc_pred, None, dims, None, False)
    assert c3d_var_nv is None
    logstd_px_nv = prj_coder.decode_logstd(proj_logstd, None, None)
    cov = torch.tensor(rng.normal(0, 1, (B, 4, 4)).astype(np.float32))
    tvec = torch.tensor(rng.uniform(5, 50, (B, 3)).astype(np.float32))
    cov_corr = prj_coder.cov_correction(cov, torch.norm(tvec, p=2, dim=1))
    np.savez_compressed(
        os.path.join(OUT, 'g3_decode_chain.npz'),
        all_pred=all_pred, labels=labels, flip=flip, dim=dim, dim_var=dim_var,
        noc_pred=noc_pred.numpy(), proj_logstd=proj_logstd.numpy(),
        noc_agnostic=noc_ag.numpy(), logstd_agnostic=logstd_ag.numpy(),
        dims=dims.numpy(), dims_var=dims_var.numpy(), c3d=c3d.numpy(), c3d_var=c3d_var.numpy(),
        logstd_px=logstd_px.numpy(), logstd_px_novar=logstd_px_nv.numpy(),
        cov_in=cov.numpy(), tvec=tvec.numpy(), cov_corr=cov_corr.numpy())
    print('G3: decode chain', c3d.shape, c3d_var.shape, logstd_px.shape)


# ------------------------------------------------------------------------------- G4 ------------
def make_g4(ref):
    rng = np.random.default_rng(11)
    B, h, w = 5, 28, 28
    x1 = rng.uniform(0, 900, B); y1 = rng.uniform(0, 250, B)
    bw = rng.uniform(20, 300, B); bh = rng.uniform(20, 120, B)
    px = (np.arange(w) + 0.5); py = (np.arange(h) + 0.5)
    c2d = np.empty((B, 2, h, w), np.float32)
    c2d[:, 0] = ((x1 - 0.5)[:, None] + px[None] * (bw / w)[:, None])[:, None, :]
    c2d[:, 1] = ((y1 - 0.5)[:, None] + py[None] * (bh / h)[:, None])[:, :, None]
    logstd = rng.normal(np.log(2.0), 0.5, (B, 2, h, w)).astype(np.float32)
    c3d = rng.normal(0, 1, (B, 3, h, w)).astype(np.float32)
    Kc = np.array([[[707.0912, 0, 601.8873], [0, 707.0912, 183.1104], [0, 0, 1]]], np.float32)
    img_shapes = np.array([[375.0, 1242.0]], np.float32)
    head = ref['opt'].UncertPropPnPOptimizer(
        pnp=dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False),
        rotation_coder=dict(type='Vec2DRotationCoder'), allowed_border=200, epnp_ransac_thres_ratio=0.2)
    with torch.no_grad():
        head.cov_calib_logscale.copy_(torch.tensor([0.3, -0.2, 0.1, 0.5]))
        ret = head(torch.tensor(c2d), torch.tensor(logstd), torch.tensor(c3d), torch.tensor(Kc), torch.tensor(img_shapes))
    a = ref['rec']['args']
    np.savez_compressed(
        os.path.join(OUT, 'g4_pose_head_prep.npz'),
        coords_2d=c2d, coords_2d_logstd=logstd, coords_3d=c3d, cam=Kc, img_shapes=img_shapes,
        cov_calib_logscale=np.array([0.3, -0.2, 0.1, 0.5], np.float32),
        pnp_coords_2d=a[0].numpy(), pnp_istd=a[1].numpy(), pnp_coords_3d=a[2].numpy(),
        pnp_strides=np.array([a[0].stride(), a[1].stride(), a[2].stride()]),
        u_range=a[4].numpy(), v_range=a[5].numpy(), ransac_thr=a[6].numpy(),
        pose_cov=ref['rec']['cov'].numpy(), pose_cov_calib=ret[4].detach().numpy(),
        pnp_cfg=np.array(sorted(ref['rec']['cfg'].items()), dtype=object).astype(str))
    print('G4: PnP-boundary strides seen by the reference:', a[0].stride(), a[1].stride(), a[2].stride())


# ------------------------------------------------------------------------------- G5 ------------
def make_g5():
    """Rotated-rectangle IoU from the reference's own numba-CUDA device functions
    (core/bbox_3d/iou_calculators/rotate_iou_kernel.py:11-255), executed as plain Python under a numba stub
    (cuda.local.array -> numpy, jit decorators -> identity).  Pins the IoU restatement that the N1 NMS uses."""
    nb = _pkg('numba')
    cu = _pkg('numba.cuda')

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = _jit; cu.jit = _jit; nb.cuda = cu
    nb.float32 = np.float32; nb.int32 = np.int32
    cu.local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype))
    cu.shared = cu.local
    _pkg('monorun.core.bbox_3d.iou_calculators')
    rk = _load('monorun.core.bbox_3d.iou_calculators.rotate_iou_kernel', 'core/bbox_3d/iou_calculators/rotate_iou_kernel.py')
    rng = np.random.default_rng(5)
    n = 400
    a = np.stack([rng.uniform(-6, 6, n), rng.uniform(-6, 6, n), rng.uniform(1.0, 5.0, n), rng.uniform(0.5, 2.5, n), rng.uniform(-np.pi, np.pi, n)], 1).astype(np.float32)
    b = a.copy()
    b[:, :2] += rng.normal(0, 1.5, (n, 2)).astype(np.float32)
    b[:, 2:4] *= rng.uniform(0.7, 1.4, (n, 2)).astype(np.float32)
    b[:, 4] += rng.normal(0, 0.8, n).astype(np.float32)
    b[:20] = a[:20]                                   # identical boxes
    b[20:40, 4] = a[20:40, 4]; b[20:40, :2] = a[20:40, :2]      # concentric, same angle, different size
    b[40:60, :2] += 30                                # disjoint
    iou = np.array([rk.devRotateIoUEval(a[i], b[i], -1) for i in range(n)], np.float64)
    np.savez_compressed(os.path.join(OUT, 'g5_rotate_iou.npz'), boxes_a_xywhr=a, boxes_b_xywhr=b, iou=iou)
    print('G5: rotated IoU pairs', n, 'mean', iou.mean().round(4), 'zeros', (iou == 0).sum(), 'ones', (np.abs(iou - 1) < 1e-6).sum())


# ------------------------------------------------------------------------------- G6 ------------
def make_g6():
    """KITTI evaluator (N2): the reference's own core/evaluation/kitti_utils/eval.py executed as plain Python under a
    numba stub on a synthetic label/detection set; its numba-CUDA rotated-IoU launch (rotate_iou.py:340-378) is
    replaced by a loop over pairs that calls the file's own device function devRotateIoUEval with the kernel's
    argument order (query box first, rotate_iou.py:335-337)."""
    nb = _pkg('numba')
    cu = _pkg('numba.cuda')

    def _jit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.jit = _jit; cu.jit = _jit; nb.cuda = cu; nb.prange = range
    nb.float32 = np.float32; nb.int32 = np.int32
    cu.local = types.SimpleNamespace(array=lambda shape, dtype: np.zeros(shape, dtype))
    cu.shared = cu.local
    _pkg('monorun.core.evaluation'); _pkg('monorun.core.evaluation.kitti_utils')
    riou = _load('monorun.core.evaluation.kitti_utils.rotate_iou', 'core/evaluation/kitti_utils/rotate_iou.py')
    ev = _load('monorun.core.evaluation.kitti_utils.eval', 'core/evaluation/kitti_utils/eval.py')

    def rotate_iou_eval(boxes, query_boxes, criterion=-1, device_id=0):
        boxes = boxes.astype(np.float32); query_boxes = query_boxes.astype(np.float32)
        out = np.zeros((boxes.shape[0], query_boxes.shape[0]), np.float32)
        for n in range(boxes.shape[0]):
            for k in range(query_boxes.shape[0]):
                out[n, k] = riou.devRotateIoUEval(query_boxes[k], boxes[n], criterion)
        return out.astype(boxes.dtype)
    riou.rotate_iou_gpu_eval = rotate_iou_eval

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
    from monorun_amd import synthetic as syn
    gts, dts = syn.make_kitti_annos(n_img=60, seed=7)
    out = {}
    out.update(syn.pack_kitti_annos(gts, 'gt_')); out.update(syn.pack_kitti_annos(dts, 'dt_'))
    classes = ['Car', 'Pedestrian', 'Cyclist']
    for crit in ('R40', 'R11'):
        text, d = ev.kitti_eval(gts, dts, classes, eval_types=['bbox', 'bev', '3d'], criteria=crit)
        out['text_' + crit] = np.array(text)
        out['dict_keys_' + crit] = np.array(sorted(d.keys()))
        out['dict_vals_' + crit] = np.array([d[k] for k in sorted(d.keys())], np.float64)
    # kitti_eval_coco_style (eval.py:772-842) cannot run: np.linspace(*float_array) (eval.py:634) raises on numpy >= 1.18 and
    # do_eval receives a bool where it iterates eval_types (eval.py:636-638); it is never called by the pipeline.
    mo = np.stack([np.array([[0.7, 0.5, 0.5]] * 3), np.array([[0.7, 0.5, 0.5], [0.5, 0.25, 0.25], [0.5, 0.25, 0.25]])], 0)
    for metric in (0, 1, 2):
        ret = ev.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        for k in ('precision', 'recall', 'orientation'):
            out[f'm{metric}_{k}'] = ret[k]
        ov = ev.calculate_iou_partly(dts, gts, metric, num_parts=60)[0]
        out[f'm{metric}_overlaps'] = np.concatenate([o.reshape(-1) for o in ov]).astype(np.float64)
    np.savez_compressed(os.path.join(OUT, 'g6_kitti_eval.npz'), **out)
    print('G6: KITTI eval\n' + str(out['text_R40']))


if __name__ == '__main__':
    torch.manual_seed(0)
    ref = load_reference()
    make_g1_g2(ref)
    make_g3(ref)
    make_g4(ref)
    make_g5()
    make_g6()
    for f in sorted(os.listdir(OUT)):
        if f.endswith('.npz'):
            print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, 'KiB')
