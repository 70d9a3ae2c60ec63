"""Consumers of the PnP outputs on the inference tail, forward only:

  * ``score_head_inputs``  — the feature vector MLPScoreHead builds from the pose
    (/root/reference/monorun/models/roi_heads/bbox_3d_heads/score_heads/mlp_score_head.py:101-103)
  * ``get_bbox_3d_result`` / ``xywhr2xyxyr`` / ``multiclass_3d_result_nms`` — same contracts as
    monorun/models/roi_heads/monorun_roi_head.py:606-677, with the rotated-BEV NMS done by this library's
    HIP kernel (``mr_nms_bev_batched``) instead of ``mmdet3d.ops.iou3d.nms_gpu`` (third-party CUDA op).
"""
import numpy as np
import torch

from . import _lib


def score_head_inputs(yaw, t_vec, pose_cov, dimensions):
    """[yaw, t_vec, tril(pose_cov) in torch.tril_indices(4, 4) order, dimensions] -> (n, 17)."""
    r, c = torch.tril_indices(4, 4, device=pose_cov.device)
    return torch.cat([yaw, t_vec, pose_cov[:, r, c], dimensions], dim=1)


def get_bbox_3d_result(dimensions, yaw, t_vec, scores, labels, num_classes, to_np=False):
    """Per-class list of (n_c, 8) rows [l, h, w, x, y, z, ry, score] (monorun_roi_head.py:606-617)."""
    rows = torch.cat((dimensions, t_vec, yaw, scores.unsqueeze(1)), dim=1)
    if to_np:
        rows, labels = rows.cpu().numpy(), labels.cpu().numpy()
    return [rows[labels == c] for c in range(num_classes)]


def xywhr2xyxyr(boxes_xywhr):
    """[cx, cy, w, h, r] -> [x1, y1, x2, y2, r] (monorun_roi_head.py:657-677)."""
    centre, half = boxes_xywhr[:, 0:2], boxes_xywhr[:, 2:4] / 2
    return torch.cat((centre - half, centre + half, boxes_xywhr[:, 4:5]), dim=1)


def nms_bev(boxes_xyxyr_list, scores_list, thr):
    """Batched rotated-BEV NMS: one group per list entry, ONE launch.  Returns a list of int64 tensors of
    kept indices (local to each group, descending score)."""
    lib = _lib.load()
    dev = boxes_xyxyr_list[0].device
    if dev.type != 'cuda':
        raise RuntimeError('monorun_amd.consumers.nms_bev runs on an MI355X only (no CPU fallback)')
    sizes = [int(b.shape[0]) for b in boxes_xyxyr_list]
    total, groups = sum(sizes), len(sizes)
    offsets = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=dev)
    f32 = dict(device=dev, dtype=torch.float32)
    boxes = torch.cat([b.detach().to(**f32).reshape(-1, 5) for b in boxes_xyxyr_list]).contiguous() if total else torch.zeros(0, 5, **f32)
    scores = torch.cat([s.detach().to(**f32).reshape(-1) for s in scores_list]).contiguous() if total else torch.zeros(0, **f32)
    keep = torch.empty(max(total, 1), dtype=torch.int64, device=dev)
    num = torch.zeros(groups, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.mr_nms_bev_batched(boxes.data_ptr(), scores.data_ptr(), offsets.data_ptr(), groups, max(sizes) if sizes else 0,
                                          float(thr), keep.data_ptr(), num.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    num_h = num.cpu().tolist()
    off_h = offsets.cpu().tolist()
    return [keep[off_h[g]:off_h[g] + num_h[g]] for g in range(groups)]


def multiclass_3d_result_nms(bbox_3d_result, nms_thr=0.25, to_np=True):
    """Rotated-BEV NMS per class over a list of (n, 8) [l, h, w, x, y, z, ry, score] tensors; returns (kept rows per class,
    kept indices per class), as numpy when to_np — the contract of monorun_roi_head.py:619-655, including its n <= 1 branch
    that returns zeros(n) as indices."""
    big = [i for i, b in enumerate(bbox_3d_result) if b.size(0) > 1]
    kept = {}
    if big:
        ks = nms_bev([xywhr2xyxyr(bbox_3d_result[i][:, [3, 5, 0, 2, 6]]) for i in big], [bbox_3d_result[i][:, 7] for i in big], nms_thr)
        kept = dict(zip(big, ks))
    out, inds = [], []
    for i, b in enumerate(bbox_3d_result):
        n = b.size(0)
        if i in kept:
            k = kept[i]
            out.append(b[k].cpu().numpy() if to_np else b[k])
            inds.append(k.cpu().numpy() if to_np else k)
        else:
            out.append(b.cpu().numpy() if to_np else b)
            inds.append(np.zeros(n, dtype=np.int64) if to_np else b.new_zeros((n, ), dtype=torch.int64))
    return out, inds
