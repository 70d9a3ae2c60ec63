#!/usr/bin/env python3
"""End-to-end harness for BASELINE configs 3/4: the whole post-NOC-head tail on the GPU, evaluated with the KITTI protocol.

    python tools/kitti_val.py --synthetic 200                     # self-contained: synthetic labels + head outputs
    python tools/kitti_val.py --labels <label_2 dir> --calib <calib dir> --ids val.txt --dumps <dir of <id>.npz>

Per image the harness needs what the detector hands to the pose stage (monorun_roi_head.py:509-534): a dump `<id>.npz`
with all_pred (n, 2*C*5, 28, 28), labels (n,), dim (n,3), dim_var (n,3, optional), rois (n,4|5), scores (n,), bboxes (n,4).
It runs decode + PnP (one fused launch; `--initialiser epnp`: K2, the reference's EPnP / RANSAC initialiser restated on the GPU, the LM),
packs the 3-D boxes per class, writes KITTI result files and evaluates them against the label files.  With torch.distributed
(`--gpus N`, one rank per GPU) the OBJECTS of every batch of images are split into contiguous shards over the ranks and the packed
per-object rows are exchanged with one RCCL all-gather per batch (parallel.RcclAllGather) before rank 0 evaluates (config 4).
No multi-GPU box was ever reachable: the world > 1 branch is exercised by the gloo test tests/test_distributed_gloo.py only."""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monorun_amd import evaluation as ev, synthetic as syn                      # noqa: E402
from monorun_amd.consumers import get_bbox_3d_result                             # noqa: E402
from monorun_amd.pose_head import UncertPropPnPOptimizer, pose_from_head         # noqa: E402

CLASSES = ('Car', 'Pedestrian', 'Cyclist')


def write_synthetic_split(root, n_img, seed=0, objs_per_img=6, outlier_frac=0.1, noise_3d=0.03):
    """Synthetic 'dataset': KITTI label + calib files and head-output dumps for n_img images."""
    os.makedirs(os.path.join(root, 'label_2')); os.makedirs(os.path.join(root, 'calib')); os.makedirs(os.path.join(root, 'dumps'))
    batch = syn.make_batch(B=n_img * objs_per_img, seed=seed, outlier_frac=outlier_frac, noise_3d=noise_3d)
    all_pred, dim = syn.encode_head_outputs(batch, seed=seed)
    rng = np.random.default_rng(seed)
    P2 = np.concatenate([syn.KITTI_K, np.array([[44.86], [0.2164], [0.0027]])], 1)          # KITTI-like P2 with a camera offset
    K, t_cam = ev.cam_t_vec_from_calib(P2)
    ids = []
    for i in range(n_img):
        s = slice(i * objs_per_img, (i + 1) * objs_per_img)
        iid = f'{i:06d}'; ids.append(iid)
        with open(os.path.join(root, 'calib', iid + '.txt'), 'w') as f:
            for cam in range(4):
                f.write(f'P{cam}: ' + ' '.join(f'{v:.12e}' for v in P2.reshape(-1)) + '\n')
        with open(os.path.join(root, 'label_2', iid + '.txt'), 'w') as f:
            for j in range(s.start, s.stop):
                l, h, w = batch['dims'][j]; x, y, z = batch['gt_t'][j] - t_cam; ry = batch['gt_yaw'][j]
                x1, y1, x2, y2 = batch['rois'][j]
                f.write(f"{CLASSES[batch['labels'][j]]} 0.00 0 {ry - np.arctan2(x, z):.4f} {x1:.2f} {y1:.2f} {x2:.2f} {y2:.2f} "
                        f"{h:.4f} {w:.4f} {l:.4f} {x:.4f} {y:.4f} {z:.4f} {ry:.4f}\n")
        np.savez(os.path.join(root, 'dumps', iid + '.npz'), all_pred=all_pred[s], labels=batch['labels'][s], dim=dim[s],
                 rois=batch['rois'][s], scores=rng.uniform(0.5, 1.0, objs_per_img).astype(np.float32),
                 bboxes=batch['rois'][s])
    with open(os.path.join(root, 'val.txt'), 'w') as f:
        f.write('\n'.join(ids) + '\n')
    return dict(labels=os.path.join(root, 'label_2'), calib=os.path.join(root, 'calib'), ids=os.path.join(root, 'val.txt'),
                dumps=os.path.join(root, 'dumps'))


def load_batch(dumps_dir, ids, infos, img_shape):
    """The objects of a batch of images as ONE list (what the detector hands to the pose stage, monorun_roi_head.py:509-534), with
    per-object camera / image shape / flip flag so that images of different cameras share a launch.  numpy arrays on the host."""
    cols = dict(all_pred=[], labels=[], dim=[], dim_var=[], rois=[], scores=[], scores_2d=[], bboxes=[], K=[], hw=[], flip=[], img=[])
    with_var = []
    for k, (iid, info) in enumerate(zip(ids, infos)):
        d = np.load(os.path.join(dumps_dir, iid + '.npz'))
        n = len(d['labels'])
        # dumps written by monorun_amd.integration.PoseStageDump also carry the image's flip flag, its own camera and the reference's
        # final scores (score head x class score): used when present
        K = np.asarray(d['cam_intrinsic'], np.float32).reshape(3, 3) if 'cam_intrinsic' in d else np.asarray(info['cam_intrinsic'], np.float32)
        hw = np.asarray(d['img_shape'], np.float32).reshape(-1)[:2] if 'img_shape' in d else np.asarray(img_shape, np.float32)
        r = np.asarray(d['rois'], np.float32).reshape(n, -1)
        cols['all_pred'].append(np.asarray(d['all_pred'])); cols['labels'].append(np.asarray(d['labels'], np.int64)); cols['dim'].append(np.asarray(d['dim'], np.float32))
        with_var.append('dim_var' in d)
        if with_var[-1]:
            cols['dim_var'].append(np.asarray(d['dim_var'], np.float32))
        cols['rois'].append(r[:, 1:5] if r.shape[1] == 5 else r)
        cols['scores'].append(np.asarray(d['scores_ref'] if 'scores_ref' in d else d['scores'], np.float32).reshape(n))
        cols['scores_2d'].append(np.asarray(d['scores'], np.float32).reshape(n)); cols['bboxes'].append(np.asarray(d['bboxes'], np.float32).reshape(n, 4))
        cols['K'].append(np.broadcast_to(K, (n, 3, 3))); cols['hw'].append(np.broadcast_to(hw, (n, 2)))
        cols['flip'].append(np.full(n, bool(d['flip']) if 'flip' in d else False)); cols['img'].append(np.full(n, k, np.int64))
    if any(with_var) and not all(with_var):
        # one launch decodes every object of the batch with OR without the dimension variance: a mixed batch would silently change
        # dimensions_var for the images that carry it
        raise ValueError('load_batch: the dumps of one batch disagree on dim_var (' + ', '.join(i for i, w in zip(ids, with_var) if not w)[:200] +
                         ' lack it): dump them with the same head configuration, or use --images-per-batch 1')
    out = {k: (np.concatenate(v) if v else None) for k, v in cols.items() if k != 'dim_var'}
    out['dim_var'] = np.concatenate(cols['dim_var']) if cols['dim_var'] else None
    return out


def pose_objects(head, ob, lo, hi, dev):
    """The product's pose stage on objects [lo, hi) of a batch: pose_from_head (one fused launch; K2 + EPnP/RANSAC + LM when the head was
    built with initialiser='epnp').  Returns dict(pose (n,4), cov (n,4,4), valid (n,) u8, dims (n,3)) of device tensors."""
    t = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(device=dev, dtype=dt)
    res = pose_from_head(head, t(ob['all_pred']), t(ob['labels'], torch.int64), t(ob['flip'], torch.bool), t(ob['dim']),
                         t(ob['dim_var']) if ob['dim_var'] is not None else None, t(ob['rois']), t(ob['K']), t(ob['hw']))
    return dict(pose=torch.cat([res['yaw_pred'], res['t_vec_pred']], 1), cov=res['pose_cov_calib'], valid=res['ret_val'].to(torch.uint8), dims=res['dimensions_pred'])


def run(a, pose_fn=None, backend='nccl', dev=None, evaluate_fn=None):
    """The harness proper.  pose_fn(objects, lo, hi) -> dict(pose, cov, valid, dims) replaces the product's pose stage (the CPU test
    of the world > 1 branch passes a stand-in: no GPU there — and, for the same reason, evaluate_fn in place of the HIP evaluator
    monorun_amd.evaluation.evaluate); backend 'nccl' = RCCL (one rank per GPU), 'gloo' = CPU ranks.
    With world > 1 the OBJECTS of every batch of images are split into contiguous shards, one per rank, each rank
    solves its shard, and ONE all-gather of the packed per-object rows (pose, covariance, validity, decoded dimensions: 100 bytes
    per object) per batch gives every rank the whole batch — BASELINE config 4's "proposals sharded across the GPUs with an RCCL
    all-gather of poses".  Returns (ap_dict, text) on rank 0, (None, None) elsewhere."""
    from monorun_amd.parallel import PackedResults, agreed_rccl_all_gather, sharded_pnp
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if dev is None:
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if backend == 'nccl' else torch.device('cpu')
    if dev.type == 'cuda':
        torch.cuda.set_device(dev)
    import torch.distributed as dist
    own_group = False
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend, rank=rank, world_size=world, **(dict(device_id=dev) if backend == 'nccl' else {}))
        own_group = True
    if a.synthetic:
        paths = write_synthetic_split(tempfile.mkdtemp(prefix='mr_kitti_'), a.synthetic) if rank == 0 else None
        if world > 1:
            box = [paths]; dist.broadcast_object_list(box, src=0); paths = box[0]
        a.labels, a.calib, a.ids, a.dumps = paths['labels'], paths['calib'], paths['ids'], paths['dumps']
    ids = [l.strip() for l in open(a.ids) if l.strip()]
    if pose_fn is None:
        head = UncertPropPnPOptimizer(pnp=dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False,
                                               initialiser=a.initialiser)).to(dev)
        pose_fn = lambda ob, lo, hi: pose_objects(head, ob, lo, hi, dev)
    # private communicator, collectives on a side stream — on every rank or on none (any set-up problem: torch.distributed's all-gather)
    exchange, why = agreed_rccl_all_gather(dev) if (world > 1 and backend == 'nccl') else (None, None)
    if why and rank == 0:
        print(f'kitti_val: private RCCL communicator unavailable ({why}); using torch.distributed all_gather_into_tensor', file=sys.stderr)
    infos = []
    for iid in ids:
        calib = ev.open_calib_file(os.path.join(a.calib, iid + '.txt'), 2)
        label = ev.open_label_file(os.path.join(a.labels, iid + '.txt')) if a.labels else None
        infos.append(ev.parse_ann_info(label, calib, CLASSES))
    results, n_coll, t_read = {}, 0, 0.0
    t0 = time.perf_counter()
    for i0 in range(0, len(ids), a.images_per_batch):
        i1 = min(i0 + a.images_per_batch, len(ids))
        tr0 = time.perf_counter()
        ob = load_batch(a.dumps, ids[i0:i1], infos[i0:i1], tuple(a.img_shape))      # every rank reads the batch's dumps (page-cached after the first): not pose-stage time
        t_read += time.perf_counter() - tr0
        n = 0 if ob['labels'] is None else len(ob['labels'])

        def solve(lo, hi, packed):
            r = pose_fn(ob, lo, hi)
            m = hi - lo
            packed.pose[:m] = r['pose']; packed.cov[:m] = r['cov']; packed.valid[:m] = r['valid']; packed.extra[:m] = r['dims']
        if n == 0:
            g = None
        elif world > 1:
            g = sharded_pnp(solve, n, dev, extra_f32=3, exchange=exchange); n_coll += 1
        else:
            pk = PackedResults(n, dev, extra_f32=3)
            solve(0, n, pk)
            g = dict(pose=pk.pose, cov=pk.cov, valid=pk.valid.bool(), extra=pk.extra)
        if rank == 0:
            for k in range(i1 - i0):
                sel = np.where(ob['img'] == k)[0] if n else np.zeros(0, np.int64)
                if len(sel) == 0:
                    results[i0 + k] = dict(bbox_results=[np.zeros((0, 5), np.float32) for _ in CLASSES], bbox_3d_results=[np.zeros((0, 8), np.float32) for _ in CLASSES])
                    continue
                st = torch.from_numpy(sel).to(g['pose'].device)
                lab = torch.from_numpy(ob['labels'][sel]).to(g['pose'].device)
                scores = torch.from_numpy(ob['scores'][sel]).to(g['pose'].device) * g['valid'][st].float()          # failed solves drop to score 0
                b3 = get_bbox_3d_result(g['extra'][st], g['pose'][st, :1], g['pose'][st, 1:], scores, lab, len(CLASSES), to_np=True)
                b2 = np.concatenate([ob['bboxes'][sel], ob['scores_2d'][sel, None]], 1)
                results[i0 + k] = dict(bbox_results=[b2[ob['labels'][sel] == c] for c in range(len(CLASSES))], bbox_3d_results=b3)
    if dev.type == 'cuda':
        torch.cuda.synchronize()
    t_pose = time.perf_counter() - t0 - t_read
    ap_dict = text = None
    if rank == 0:
        out = a.out or tempfile.mkdtemp(prefix='mr_kitti_results_')
        t0 = time.perf_counter()
        ap_dict, text, _ = (evaluate_fn or ev.evaluate)([results[i] for i in range(len(ids))], infos, CLASSES, filenames=[iid + '.png' for iid in ids], result_dir=out)
        print(text)
        n_obj = sum(len(r['bbox_3d_results'][c]) for r in results.values() for c in range(len(CLASSES)))
        print(f'{len(ids)} images, {n_obj} objects, initialiser {a.initialiser!r}: pose stage {t_pose:.2f} s ({world} rank(s), objects sharded, {n_coll} packed '
              f'all-gather(s) of 100-byte rows{" over a private RCCL communicator" if exchange is not None else ""}; reading the dumps took {t_read:.2f} s more), '
              f'formatting + result files + evaluation {time.perf_counter() - t0:.2f} s; result files in {out}/data')
    if exchange is not None:
        exchange.close()
    if own_group:
        dist.barrier()
        dist.destroy_process_group()
    return ap_dict, text


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, default=0, help='generate this many synthetic images instead of reading a dataset')
    ap.add_argument('--labels'); ap.add_argument('--calib'); ap.add_argument('--ids'); ap.add_argument('--dumps')
    ap.add_argument('--out', default=None, help='directory for the KITTI result files')
    ap.add_argument('--img-shape', type=int, nargs=2, default=(375, 1242))
    ap.add_argument('--gpus', type=int, default=1, help='shard the objects over this many GPUs (the script starts its own ranks)')
    ap.add_argument('--initialiser', choices=('k0', 'epnp'), default='epnp',
                    help="'epnp' (default): the reference's cv2.solvePnPRansac(EPNP) restated on the GPU; 'k0': the one-launch kernel's own initialiser (fast mode)")
    ap.add_argument('--images-per-batch', type=int, default=16, help='images whose objects share one launch (and, sharded, one all-gather)')
    return ap.parse_args(argv)


def main():
    a = parse_args()
    from monorun_amd import launch
    if a.gpus > 1 and not launch.in_distributed_job():
        return launch.spawn_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    run(a)
    return 0


if __name__ == '__main__':
    sys.exit(main())
