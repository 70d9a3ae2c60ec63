#!/usr/bin/env python3
"""End-to-end harness for BASELINE configs 3/4: the whole post-NOC-head tail on the GPU, evaluated with the KITTI protocol.

    python tools/kitti_val.py --synthetic 200                     # self-contained: synthetic labels + head outputs
    python tools/kitti_val.py --labels <label_2 dir> --calib <calib dir> --ids val.txt --dumps <dir of <id>.npz>

Per image the harness needs what the detector hands to the pose stage (monorun_roi_head.py:509-534): a dump `<id>.npz`
with all_pred (n, 2*C*5, 28, 28), labels (n,), dim (n,3), dim_var (n,3, optional), rois (n,4|5), scores (n,), bboxes (n,4).
It runs decode + PnP (one fused launch), packs the 3-D boxes per class, writes KITTI result files and evaluates them
against the label files.  With torch.distributed (torchrun, one rank per GPU) images are sharded round-robin over the
ranks and the per-image results are gathered on rank 0 before the evaluation (config 4)."""
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


def run_image(head, dump, K, img_shape, dev):
    t = lambda a, dt=torch.float32: torch.from_numpy(np.asarray(a)).to(device=dev, dtype=dt)
    n = len(dump['labels'])
    labels = t(dump['labels'], torch.int64)
    if n == 0:
        return dict(bbox_results=[np.zeros((0, 5), np.float32) for _ in CLASSES], bbox_3d_results=[np.zeros((0, 8), np.float32) for _ in CLASSES])
    # dumps written by monorun_amd.integration.PoseStageDump also carry the image's flip flag, its own camera and the reference's
    # final scores (score head x class score): use them when present
    flip = bool(dump['flip']) if 'flip' in dump else False
    if 'cam_intrinsic' in dump:
        K = t(np.asarray(dump['cam_intrinsic']).reshape(1, 3, 3))
    if 'img_shape' in dump:
        img_shape = tuple(float(v) for v in np.asarray(dump['img_shape']).reshape(-1)[:2])
    res = pose_from_head(head, t(dump['all_pred']), labels, flip, t(dump['dim']), t(dump['dim_var']) if 'dim_var' in dump else None,
                         t(dump['rois']), K, img_shape)
    scores = t(dump['scores_ref'] if 'scores_ref' in dump else dump['scores']) * res['ret_val'].float()           # failed solves drop to score 0
    b3 = get_bbox_3d_result(res['dimensions_pred'], res['yaw_pred'], res['t_vec_pred'], scores, labels, len(CLASSES), to_np=True)
    b2 = np.concatenate([np.asarray(dump['bboxes'], np.float32).reshape(n, 4), np.asarray(dump['scores'], np.float32).reshape(n, 1)], 1)
    lab = np.asarray(dump['labels'])
    return dict(bbox_results=[b2[lab == c] for c in range(len(CLASSES))], bbox_3d_results=b3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--synthetic', type=int, default=0, help='generate this many synthetic images instead of reading a dataset')
    ap.add_argument('--labels'); ap.add_argument('--calib'); ap.add_argument('--ids'); ap.add_argument('--dumps')
    ap.add_argument('--out', default=None, help='directory for the KITTI result files')
    ap.add_argument('--img-shape', type=int, nargs=2, default=(375, 1242))
    ap.add_argument('--gpus', type=int, default=1, help='shard the images over this many GPUs (the script starts its own ranks)')
    a = ap.parse_args()
    from monorun_amd import launch
    if a.gpus > 1 and not launch.in_distributed_job():
        return launch.spawn_ranks(a.gpus, os.path.abspath(__file__), sys.argv[1:])
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    tmp = None
    if a.synthetic:
        tmp = tempfile.mkdtemp(prefix='mr_kitti_')
        paths = write_synthetic_split(tmp, a.synthetic) if rank == 0 else None
        if world > 1:
            box = [paths]; dist.broadcast_object_list(box, src=0); paths = box[0]
        a.labels, a.calib, a.ids, a.dumps = paths['labels'], paths['calib'], paths['ids'], paths['dumps']
    ids = [l.strip() for l in open(a.ids) if l.strip()]
    head = UncertPropPnPOptimizer().to(dev)
    infos, results = [], {}
    t0 = time.perf_counter()
    for i, iid in enumerate(ids):
        calib = ev.open_calib_file(os.path.join(a.calib, iid + '.txt'), 2)
        label = ev.open_label_file(os.path.join(a.labels, iid + '.txt')) if a.labels else None
        infos.append(ev.parse_ann_info(label, calib, CLASSES))
        if i % world == rank:
            K = torch.from_numpy(infos[-1]['cam_intrinsic'])[None].to(dev)
            results[i] = run_image(head, np.load(os.path.join(a.dumps, iid + '.npz')), K, tuple(a.img_shape), dev)
    torch.cuda.synchronize()
    t_pose = time.perf_counter() - t0
    if world > 1:
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(results, gathered, dst=0)
        if rank == 0:
            results = {k: v for part in gathered for k, v in part.items()}
    if rank == 0:
        out = a.out or tempfile.mkdtemp(prefix='mr_kitti_results_')
        t0 = time.perf_counter()
        ap_dict, text, _ = ev.evaluate([results[i] for i in range(len(ids))], infos, CLASSES, filenames=[iid + '.png' for iid in ids], result_dir=out)
        print(text)
        n_obj = sum(len(r['bbox_3d_results'][c]) for r in results.values() for c in range(len(CLASSES)))
        print(f'{len(ids)} images, {n_obj} objects: pose stage {t_pose:.2f} s ({world} rank(s), incl. file reads), '
              f'formatting + result files + evaluation {time.perf_counter() - t0:.2f} s; result files in {out}/data')
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == '__main__':
    sys.exit(main())
