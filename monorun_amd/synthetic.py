"""Seeded synthetic 2D-3D correspondence batches (configs 1, 2 and 5).

The reference ships no test data and no dataset is reachable, so bench.py and the parity tests
use these generators.  They only produce *inputs* at the PnP boundary
(``PnPUncert.forward`` of /root/reference/monorun/ops/least_squares/pnp_uncert.py:125-142) and at
the NOC-head boundary; nothing here computes a pose.

Object frame (KITTI): x along length, y down with the origin at the bottom-centre
(y in [-h, 0], consistent with the NOC mean (-0.1,-0.5,0) of noc_coder.py:9), z along width.
"""
import numpy as np

# demo/calib.csv of the reference (fx = fy = 707.0912, cx = 601.8873, cy = 183.1104)
KITTI_K = np.array([[707.0912, 0.0, 601.8873], [0.0, 707.0912, 183.1104], [0.0, 0.0, 1.0]], np.float64)
IMG_W, IMG_H = 1242, 375
# multiclass_norm_dim_coder.py:8-15 — (l, h, w) mean / std per class (car, pedestrian, cyclist)
DIM_MEANS = np.array([(3.89, 1.53, 1.62), (0.82, 1.78, 0.63), (1.77, 1.72, 0.57)])
DIM_STDS = np.array([(0.44, 0.14, 0.11), (0.25, 0.13, 0.12), (0.15, 0.10, 0.14)])
NOC_MEANS = (-0.1, -0.5, 0.0)            # noc_coder.py:12-13
NOC_STDS = (0.35, 0.23, 0.34)


def cube_config1(n_points=64, seed=0):
    """Config 1: 64 noise-free correspondences on a car-sized box, GT yaw=0.3, t=(1,1.5,15)."""
    rng = np.random.default_rng(seed)
    l, h, w = DIM_MEANS[0]
    face = rng.integers(0, 6, n_points)
    a, b = rng.uniform(-0.5, 0.5, n_points), rng.uniform(-0.5, 0.5, n_points)
    X = np.empty((n_points, 3))
    ax = face // 2
    sgn = np.where(face % 2 == 0, -0.5, 0.5)
    # axis ax pinned to a face, the other two uniform
    X[:, 0] = np.where(ax == 0, sgn, a) * l
    X[:, 1] = (np.where(ax == 1, sgn, np.where(ax == 0, a, b)) - 0.5) * h
    X[:, 2] = np.where(ax == 2, sgn, b) * w
    yaw, t = 0.3, np.array([1.0, 1.5, 15.0])
    K = KITTI_K
    c, s = np.cos(yaw), np.sin(yaw)
    Xc = c * X[:, 0] + s * X[:, 2] + t[0]
    Yc = X[:, 1] + t[1]
    Zc = -s * X[:, 0] + c * X[:, 2] + t[2]
    x2d = np.stack([K[0, 0] * Xc / Zc + K[0, 2], K[1, 1] * Yc / Zc + K[1, 2]], 1)
    return dict(pts2d=x2d, pts3d=X, wgt2d=np.ones((n_points, 2)), K=K.copy(),
                clips=np.array([0.5, -200.0, IMG_W + 200.0, -200.0, IMG_H + 200.0]),
                gt_pose=np.array([yaw, *t]), init_pose=np.array([yaw, *t]) + np.array([0.2, 0.5, 0.2, 2.0]))


def _box_corners(dims):
    l, h, w = dims[:, 0], dims[:, 1], dims[:, 2]
    sx = np.array([1, 1, 1, 1, -1, -1, -1, -1]) * 0.5
    sy = np.array([0, 0, -1, -1, 0, 0, -1, -1]) * 1.0
    sz = np.array([1, -1, 1, -1, 1, -1, 1, -1]) * 0.5
    return np.stack([l[:, None] * sx, h[:, None] * sy, w[:, None] * sz], -1)      # (n, 8, 3)


def _to_cam(X, yaw, t):
    c, s = np.cos(yaw)[:, None], np.sin(yaw)[:, None]
    return np.stack([c * X[..., 0] + s * X[..., 2] + t[:, None, 0], X[..., 1] + t[:, None, 1],
                     -s * X[..., 0] + c * X[..., 2] + t[:, None, 2]], -1)


def make_batch(B=1024, hw=28, seed=1234, outlier_frac=0.15, noise_3d=0.03, outlier_noise_3d=0.3,
               K=KITTI_K, img_wh=(IMG_W, IMG_H)):
    """Config 2 (hw=28, seed 1234) / config 5 (hw=56, seed 4321) generator.

    Returns float32 NCHW maps as the NOC head side produces them plus everything needed to form
    the PnP-boundary tensors:
      coords_2d (B,2,hw,hw)  RoI bin-centre grid (R12 analytic form)
      coords_3d (B,3,hw,hw)  back-projected object coordinates (+ noise, + gross outliers)
      logstd    (B,2,hw,hw)  pixel log-std  (istd = exp(-logstd)/10, uncert_prop_pnp_optimizer.py:73)
      rois (B,4) xyxy, labels (B,), dims (B,3), gt_yaw (B,), gt_t (B,3), outlier (B,hw,hw) bool
    """
    rng = np.random.default_rng(seed)
    W, H = img_wh
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    labels = np.empty(0, np.int64); dims = np.empty((0, 3)); yaw = np.empty(0); t = np.empty((0, 3)); rois = np.empty((0, 4))
    while labels.shape[0] < B:
        n = 2 * B
        lab = rng.integers(0, 3, n)
        dm = np.maximum(rng.normal(DIM_MEANS[lab], DIM_STDS[lab]), 0.3)
        yw = rng.uniform(-np.pi, np.pi, n)
        tt = np.stack([rng.uniform(-15, 15, n), rng.uniform(1, 2, n), rng.uniform(5, 60, n)], 1)
        cam = _to_cam(_box_corners(dm), yw, tt)
        ok = (cam[..., 2] > 1.0).all(1)
        u = fx * cam[..., 0] / np.maximum(cam[..., 2], 1e-3) + cx
        v = fy * cam[..., 1] / np.maximum(cam[..., 2], 1e-3) + cy
        box = np.stack([u.min(1), v.min(1), u.max(1), v.max(1)], 1)
        ok &= (box[:, 0] >= 0) & (box[:, 1] >= 0) & (box[:, 2] <= W - 1) & (box[:, 3] <= H - 1)
        ok &= (box[:, 3] - box[:, 1] >= 8.0)
        labels = np.concatenate([labels, lab[ok]]); dims = np.concatenate([dims, dm[ok]])
        yaw = np.concatenate([yaw, yw[ok]]); t = np.concatenate([t, tt[ok]]); rois = np.concatenate([rois, box[ok]])
    labels, dims, yaw, t, rois = labels[:B], dims[:B], yaw[:B], t[:B], rois[:B]

    # R12 grid: u(px) = x1 - 0.5 + (px + 0.5) * (x2 - x1) / w
    pc = np.arange(hw) + 0.5
    gu = (rois[:, 0] - 0.5)[:, None] + pc[None] * ((rois[:, 2] - rois[:, 0]) / hw)[:, None]      # (B, hw)
    gv = (rois[:, 1] - 0.5)[:, None] + pc[None] * ((rois[:, 3] - rois[:, 1]) / hw)[:, None]
    U = np.broadcast_to(gu[:, None, :], (B, hw, hw))
    V = np.broadcast_to(gv[:, :, None], (B, hw, hw))
    # rays in the object frame
    d_cam = np.stack([(U - cx) / fx, (V - cy) / fy, np.ones_like(U)], -1).reshape(B, -1, 3)
    c, s = np.cos(yaw)[:, None], np.sin(yaw)[:, None]
    def rot_t(vv):      # R_y^T v
        return np.stack([c * vv[..., 0] - s * vv[..., 2], vv[..., 1], s * vv[..., 0] + c * vv[..., 2]], -1)
    d = rot_t(d_cam)
    o = rot_t(-t[:, None, :])
    lo = np.stack([-dims[:, 0] / 2, -dims[:, 1], -dims[:, 2] / 2], 1)[:, None, :]
    hi = np.stack([dims[:, 0] / 2, np.zeros(B), dims[:, 2] / 2], 1)[:, None, :]
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = 1.0 / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
    tn, tf = np.minimum(t0, t1).max(-1), np.maximum(t0, t1).min(-1)
    hit = (tn <= tf) & (tn > 0)
    ctr = (lo + hi) / 2
    tc = ((ctr - o) * d).sum(-1) / (d * d).sum(-1)                       # closest approach to the box centre
    tpar = np.where(hit, tn, tc)
    X = np.clip(o + tpar[..., None] * d, lo, hi)
    outlier = ~hit
    outlier |= rng.uniform(size=outlier.shape) < outlier_frac
    X = X + rng.normal(0, noise_3d, X.shape) + outlier[..., None] * rng.normal(0, outlier_noise_3d, X.shape)
    logstd = rng.normal(np.log(2.0), 0.5, (B, hw * hw, 2)) + outlier[..., None] * np.log(10.0)

    def nchw(a):
        return np.ascontiguousarray(a.reshape(B, hw, hw, -1).transpose(0, 3, 1, 2), np.float32)
    coords_2d = np.empty((B, 2, hw, hw), np.float32)
    coords_2d[:, 0] = U
    coords_2d[:, 1] = V
    return dict(coords_2d=coords_2d, coords_3d=nchw(X), logstd=nchw(logstd), rois=rois.astype(np.float32),
                labels=labels, dims=dims.astype(np.float32), gt_yaw=yaw, gt_t=t, outlier=outlier.reshape(B, hw, hw),
                K=K[None].astype(np.float32), img_shape=np.array([[H, W]], np.float32))


def pnp_boundary(batch, allowed_border=200, ransac_ratio=0.2, std_scale=10.0, planar=True):
    """NCHW maps -> the tensors PnPUncert.forward receives (uncert_prop_pnp_optimizer.py:73-88).
    planar=True keeps the reference's strided views (strides (C*P, 1, P)); False makes (B,P,C)
    contiguous copies."""
    c2d, c3d, ls = batch['coords_2d'], batch['coords_3d'], batch['logstd']
    B, _, h, w = c2d.shape
    istd = (np.exp(-ls) / np.float32(std_scale)).astype(np.float32)
    H, W = batch['img_shape'][0]
    u_range = np.array([[-allowed_border, W + allowed_border]], np.float32)
    v_range = np.array([[-allowed_border, H + allowed_border]], np.float32)
    thr = (np.float32(ransac_ratio) * (c2d[:, 1, -1, 0] - c2d[:, 1, 0, 0])).astype(np.float32)
    def view(a):
        v = a.reshape(B, a.shape[1], h * w).transpose(0, 2, 1)
        return v if planar else np.ascontiguousarray(v)
    return view(c2d), view(istd), view(c3d), batch['K'], u_range, v_range, thr


# ---------------------------------------------------------------------------------------------------
# KITTI-style annotations for the evaluator (N2).  No dataset is reachable offline, so the evaluator is
# exercised on synthetic label / detection sets in the exact dict format the reference's evaluator reads
# (kitti3d_dataset.py:230-305: name, truncated, occluded, alpha, bbox, dimensions [l,h,w], location
# [x, y(bottom), z] in the camera frame, rotation_y, score).
_KITTI_DIMS = {  # mean l, h, w
    'Car': (3.89, 1.53, 1.62), 'Van': (5.08, 2.21, 1.90), 'Pedestrian': (0.84, 1.76, 0.66),
    'Person_sitting': (0.80, 1.27, 0.59), 'Cyclist': (1.76, 1.74, 0.60),
}


def _kitti_bbox(loc, dims, ry, K=KITTI_K, img_wh=(1242, 375)):
    l, h, w = dims
    xc = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    yc = np.array([0, 0, 0, 0, -h, -h, -h, -h], np.float64)
    zc = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    c, s = np.cos(ry), np.sin(ry)
    X = np.stack([c * xc + s * zc + loc[0], yc + loc[1], -s * xc + c * zc + loc[2]])
    X[2] = np.maximum(X[2], 0.1)
    uv = (K @ X)[:2] / X[2]
    x1, y1, x2, y2 = uv[0].min(), uv[1].min(), uv[0].max(), uv[1].max()
    full = max(x2 - x1, 1e-6) * max(y2 - y1, 1e-6)
    x1c, y1c, x2c, y2c = np.clip(x1, 0, img_wh[0] - 1), np.clip(y1, 0, img_wh[1] - 1), np.clip(x2, 0, img_wh[0] - 1), np.clip(y2, 0, img_wh[1] - 1)
    vis = max(x2c - x1c, 0) * max(y2c - y1c, 0)
    return np.array([x1c, y1c, x2c, y2c]), float(1.0 - vis / full)


def make_kitti_annos(n_img=60, seed=7, max_gt=7, dtype=np.float32):
    """Synthetic (gt_annos, dt_annos) lists: jittered detections of most labels, class confusions, false positives,
    DontCare regions with detections inside them, empty images, all three difficulty regimes."""
    rng = np.random.default_rng(seed)
    names = ['Car', 'Car', 'Car', 'Pedestrian', 'Pedestrian', 'Cyclist', 'Cyclist', 'Van', 'Person_sitting']
    gts, dts = [], []
    for _ in range(n_img):
        n_obj = int(rng.integers(0, max_gt + 1))
        g = dict(name=[], truncated=[], occluded=[], alpha=[], bbox=[], dimensions=[], location=[], rotation_y=[])
        d = dict(name=[], alpha=[], bbox=[], dimensions=[], location=[], rotation_y=[], score=[])
        for _o in range(n_obj):
            nm = names[int(rng.integers(len(names)))]
            dims = np.array(_KITTI_DIMS[nm]) * rng.uniform(0.85, 1.15, 3)
            z = rng.uniform(5, 45); x = rng.uniform(-0.55, 0.55) * z; y = rng.uniform(1.4, 1.9)
            ry = rng.uniform(-np.pi, np.pi)
            loc = np.array([x, y, z])
            bbox, trunc = _kitti_bbox(loc, dims, ry)
            if bbox[2] - bbox[0] < 4 or bbox[3] - bbox[1] < 4:
                continue
            occ = int(rng.choice([0, 0, 0, 1, 1, 2, 3]))
            g['name'].append(nm); g['truncated'].append(min(trunc + rng.uniform(0, 0.05), 1.0)); g['occluded'].append(occ)
            g['alpha'].append(ry - np.arctan2(x, z)); g['bbox'].append(bbox); g['dimensions'].append(dims)
            g['location'].append(loc); g['rotation_y'].append(ry)
            if rng.uniform() < 0.85:                     # detected
                q = 1.0 - rng.beta(1.2, 3.5)               # quality (mostly good, a tail of poor localisations)
                sig = 0.05 + 0.5 * (1 - q)
                dloc = loc + rng.normal(0, sig, 3) * np.array([1, 0.3, 1.5])
                ddim = dims * rng.uniform(1 - 0.1 * (1 - q) - 0.01, 1 + 0.1 * (1 - q) + 0.01, 3)
                dry = ry + rng.normal(0, 0.02 + 0.25 * (1 - q))
                dbox = bbox + rng.normal(0, 0.5 + 4 * (1 - q), 4)
                dn = nm if rng.uniform() < 0.93 else ['Car', 'Pedestrian', 'Cyclist'][int(rng.integers(3))]
                if dn in ('Van', 'Person_sitting'):
                    dn = 'Car' if dn == 'Van' else 'Pedestrian'
                d['name'].append(dn); d['alpha'].append(dry - np.arctan2(dloc[0], dloc[2] + 0.27)); d['bbox'].append(dbox)
                d['dimensions'].append(ddim); d['location'].append(dloc); d['rotation_y'].append(dry)
                d['score'].append(np.clip(0.15 + 0.8 * q + rng.normal(0, 0.05), 0.01, 0.999))
        n_dc = int(rng.integers(0, 3))
        for _o in range(n_dc):
            x1 = rng.uniform(0, 1100); y1 = rng.uniform(120, 250); w = rng.uniform(30, 140); h = rng.uniform(20, 90)
            g['name'].append('DontCare'); g['truncated'].append(-1); g['occluded'].append(-1); g['alpha'].append(-10)
            g['bbox'].append(np.array([x1, y1, x1 + w, y1 + h])); g['dimensions'].append(np.full(3, -1.0))
            g['location'].append(np.full(3, -1000.0)); g['rotation_y'].append(-10)
            if rng.uniform() < 0.7:                      # a detection inside the DontCare region
                nm = ['Car', 'Pedestrian', 'Cyclist'][int(rng.integers(3))]
                dims = np.array(_KITTI_DIMS[nm]); z = rng.uniform(40, 70)
                bx = np.array([x1 + 0.2 * w, y1 + 0.2 * h, x1 + 0.8 * w, y1 + 0.8 * h])
                xx = (0.5 * (bx[0] + bx[2]) - KITTI_K[0, 2]) * z / KITTI_K[0, 0]
                ry = rng.uniform(-np.pi, np.pi)
                d['name'].append(nm); d['alpha'].append(ry - np.arctan2(xx, z + 0.27)); d['bbox'].append(bx)
                d['dimensions'].append(dims); d['location'].append(np.array([xx, 1.6, z])); d['rotation_y'].append(ry)
                d['score'].append(rng.uniform(0.05, 0.6))
        for _o in range(int(rng.integers(0, 3))):          # false positives
            nm = ['Car', 'Pedestrian', 'Cyclist'][int(rng.integers(3))]
            dims = np.array(_KITTI_DIMS[nm]) * rng.uniform(0.9, 1.1, 3)
            z = rng.uniform(5, 60); x = rng.uniform(-0.5, 0.5) * z; ry = rng.uniform(-np.pi, np.pi)
            loc = np.array([x, rng.uniform(1.4, 1.9), z])
            bbox, _t = _kitti_bbox(loc, dims, ry)
            d['name'].append(nm); d['alpha'].append(ry - np.arctan2(x, z + 0.27)); d['bbox'].append(bbox)
            d['dimensions'].append(dims); d['location'].append(loc); d['rotation_y'].append(ry)
            d['score'].append(rng.uniform(0.02, 0.7))

        def arr(v, shape):
            return np.asarray(v, dtype).reshape(shape)
        ng, nd = len(g['name']), len(d['name'])
        gts.append(dict(name=np.array(g['name'], dtype='<U16'), truncated=arr(g['truncated'], (ng,)), occluded=arr(g['occluded'], (ng,)),
                        alpha=arr(g['alpha'], (ng,)), bbox=arr(g['bbox'], (ng, 4)), dimensions=arr(g['dimensions'], (ng, 3)),
                        location=arr(g['location'], (ng, 3)), rotation_y=arr(g['rotation_y'], (ng,)), score=np.zeros(ng, dtype)))
        order = np.argsort(-np.asarray(d['score'], np.float64), kind='stable') if nd else np.zeros(0, np.int64)
        dts.append(dict(name=np.array(d['name'], dtype='<U16')[order], truncated=np.full(nd, -1, np.int8), occluded=np.full(nd, -1, np.int8),
                        alpha=arr(d['alpha'], (nd,))[order], bbox=arr(d['bbox'], (nd, 4))[order], dimensions=arr(d['dimensions'], (nd, 3))[order],
                        location=arr(d['location'], (nd, 3))[order], rotation_y=arr(d['rotation_y'], (nd,))[order], score=arr(d['score'], (nd,))[order]))
    return gts, dts


_ANNO_FLOAT_KEYS = ('truncated', 'occluded', 'alpha', 'bbox', 'dimensions', 'location', 'rotation_y', 'score')


def pack_kitti_annos(annos, prefix):
    """list of annotation dicts -> flat arrays (for .npz fixtures)."""
    out = {prefix + 'count': np.array([len(a['name']) for a in annos], np.int64),
           prefix + 'name': np.concatenate([np.asarray(a['name'], dtype='<U16') for a in annos]) if annos else np.zeros(0, '<U16')}
    for k in _ANNO_FLOAT_KEYS:
        out[prefix + k] = np.concatenate([np.asarray(a[k]) for a in annos], 0)
    return out


def unpack_kitti_annos(z, prefix):
    cnt = z[prefix + 'count']
    off = np.concatenate([[0], np.cumsum(cnt)])
    annos = []
    for i in range(len(cnt)):
        s = slice(off[i], off[i + 1])
        a = dict(name=z[prefix + 'name'][s].copy())
        for k in _ANNO_FLOAT_KEYS:
            a[k] = z[prefix + k][s].copy()
        annos.append(a)
    return annos


# ---------------------------------------------------------------------------------------------------
# Synthetic raw NOC-head outputs (the inverse of the decode chain) — lets the whole post-head tail
# (decode -> PnP -> 3-D boxes -> KITTI files -> evaluator) run end to end without the CNN or a dataset.
def encode_head_outputs(batch, num_classes=3, seed=0):
    """make_batch() maps -> (all_pred (B, 2*C*5, h, w) float32, dim (B,3) normalised dimensions).
    Branch 0 (no flip) of the object's class carries noc = (X / dims - mu_noc) / sigma_noc and the pixel log-std;
    every other channel is noise, so a wrong channel pick shows up as a wrong pose."""
    rng = np.random.default_rng(seed)
    c3d, ls, labels, dims = batch['coords_3d'], batch['logstd'], batch['labels'], batch['dims']
    B, _, h, w = c3d.shape
    C = num_classes
    all_pred = rng.normal(0, 1, (B, 2 * C * 5, h, w)).astype(np.float32)
    noc = (c3d / dims[:, :, None, None] - np.asarray(NOC_MEANS, np.float32)[None, :, None, None]) / np.asarray(NOC_STDS, np.float32)[None, :, None, None]
    for b in range(B):
        c = int(labels[b])
        all_pred[b, 3 * c:3 * c + 3] = noc[b]
        all_pred[b, 3 * C + 2 * c:3 * C + 2 * c + 2] = ls[b]
    dim = ((dims - DIM_MEANS[labels]) / DIM_STDS[labels]).astype(np.float32)
    return all_pred, dim
