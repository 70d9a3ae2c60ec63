"""KITTI object evaluation on the GPU + the KITTI wire format of the results (N3).

Host-side mirror of the reference's `monorun.core.evaluation` (`kitti_eval`, `kitti_eval_coco_style`; /root/reference/
monorun/core/evaluation/kitti_utils/eval.py) with the same entry points, argument meaning, result text and result dict.
The reference runs numba-jitted CPU loops plus a numba-CUDA rotated-IoU kernel (rotate_iou.py) — neither exists on a
ROCm box.  Here the three loop nests are HIP kernels behind the C ABI (`mr_kitti_overlaps`, `mr_kitti_match`,
include/monorun_pnp.h); the host keeps what is string handling, sorting and formatting:

    annotations -> flat arrays + prefix offsets (once)           host   numpy
    ignore codes per (class, difficulty)      clean_data          host   numpy, vectorised (string compares)
    per-image overlap blocks                  eval.py:84-158      GPU    kitti_overlap_kernel
    greedy matching, pass 1 (true positives)  eval.py:161-279     GPU    kitti_match_kernel, all classes x difficulties x overlaps
    41 recall-sampled score thresholds        eval.py:8-25        host   sort + scan
    matching, pass 2 at every threshold       eval.py:291-338     GPU    kitti_match_kernel + ordered reduction
    precision / recall / AOS curves, AP, text eval.py:536-770     host   numpy

There is no CPU fallback: without the HIP library or a GPU these functions raise.
"""
import io
import os
import shutil

import numpy as np
import torch

from . import _lib

__all__ = ['kitti_eval', 'kitti_eval_coco_style', 'eval_class', 'calculate_iou_partly', 'get_mAP', 'get_thresholds',
           'do_eval', 'format_results', 'format_gt_anno', 'write_result_files', 'cam_t_vec_from_calib',
           'open_calib_file', 'open_label_file', 'parse_ann_info', 'evaluate']

_CLASS_NAMES = ('car', 'pedestrian', 'cyclist')                     # eval.py:29
_MIN_HEIGHT = (40, 25, 25)
_MAX_OCCLUSION = (0, 1, 2)
_MAX_TRUNCATION = (0.15, 0.3, 0.5)
_N_SAMPLE_PTS = 41
_CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting'}
_NAME_TO_CLASS = {v: k for k, v in _CLASS_TO_NAME.items()}


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError('monorun_amd.evaluation needs a HIP device (there is no CPU fallback)')
    return torch.device('cuda', torch.cuda.current_device())


def _cat(annos, key, width=None):
    parts = [np.asarray(a[key]) for a in annos]
    if width is not None:
        parts = [p.reshape(-1, width) for p in parts]
    return np.concatenate(parts, 0) if parts else np.zeros((0,) if width is None else (0, width))


def _to_dev(a, dev, dtype=None):
    """Host array -> device tensor; an empty array becomes one zero element so that its pointer is never NULL."""
    a = np.ascontiguousarray(a if dtype is None else np.asarray(a).astype(dtype))
    if a.size == 0:
        a = np.zeros((1,) + a.shape[1:] if a.ndim > 1 and 0 not in a.shape[1:] else (1,), a.dtype)
    return torch.from_numpy(a).to(dev)


class _Side:
    """One list of annotation dicts as flat arrays."""

    def __init__(self, annos):
        self.count = np.array([len(a['name']) for a in annos], np.int64)
        self.off = np.concatenate([[0], np.cumsum(self.count)]).astype(np.int64)
        self.name = np.concatenate([np.asarray(a['name'], dtype=str) for a in annos]) if len(annos) else np.zeros(0, str)
        self.bbox = _cat(annos, 'bbox', 4)
        self.loc = _cat(annos, 'location', 3)
        self.dims = _cat(annos, 'dimensions', 3)
        self.ry = _cat(annos, 'rotation_y')
        self.alpha = _cat(annos, 'alpha')
        n = int(self.off[-1])
        self.score = _cat(annos, 'score') if all('score' in a for a in annos) else np.zeros(n, np.float32)
        self.truncated = _cat(annos, 'truncated') if all('truncated' in a for a in annos) else np.zeros(n)
        self.occluded = _cat(annos, 'occluded') if all('occluded' in a for a in annos) else np.zeros(n)
        rows = np.zeros((n, 12), np.float64)
        rows[:, 0:4] = self.bbox; rows[:, 4:7] = self.loc; rows[:, 7:10] = self.dims; rows[:, 10] = self.ry; rows[:, 11] = self.score
        self.rows = rows
        # dtype numba saw for the concatenated 3-D boxes (eval.py:383-398) and for gt_datas / dt_datas (eval.py:436-442)
        self.box3d_dtype = np.result_type(self.loc.dtype, self.dims.dtype, self.ry.dtype)
        self.datas_dtype = np.result_type(self.bbox.dtype, self.alpha.dtype)


class _Session:
    """Annotations resident on the device for one evaluation (shared by the three metrics)."""

    def __init__(self, gt_annos, dt_annos):
        assert len(gt_annos) == len(dt_annos)
        self.dev = _device()
        self.lib = _lib.load()
        self.n_img = len(gt_annos)
        self.gt, self.dt = _Side(gt_annos), _Side(dt_annos)
        self.dt_datas_dtype = np.result_type(self.dt.datas_dtype, self.dt.score.dtype)
        t = lambda a, dt=None: _to_dev(a, self.dev, dt)
        self.d_dt_off, self.d_gt_off = t(self.dt.off), t(self.gt.off)
        self.d_dt_rows, self.d_gt_rows = t(self.dt.rows), t(self.gt.rows)
        self.d_dt_alpha, self.d_gt_alpha = t(self.dt.alpha, np.float64), t(self.gt.alpha, np.float64)
        dc = self.gt.name == 'DontCare'                                    # eval.py:64-65 (case-sensitive)
        self.dc_count = np.array([int(dc[self.gt.off[i]:self.gt.off[i + 1]].sum()) for i in range(self.n_img)], np.int64)
        self.d_dc_off = t(np.concatenate([[0], np.cumsum(self.dc_count)]).astype(np.int64))
        self.d_dc_box = t(self.gt.bbox[dc].reshape(-1, 4), np.float64)
        self.max_det = int(self.dt.count.max()) if self.n_img else 0
        self._ov = {}

    def overlaps(self, metric):
        """Flat per-image overlap blocks (rows = detections, columns = labels) for one metric, cached."""
        if metric in self._ov:
            return self._ov[metric]
        ov_off = np.concatenate([[0], np.cumsum(self.dt.count * self.gt.count)]).astype(np.int64)
        total = int(ov_off[-1])
        d_off = _to_dev(ov_off, self.dev)
        ov = torch.zeros(max(total, 1), dtype=torch.float64, device=self.dev)
        if metric == 0:
            both32 = self.dt.bbox.dtype == np.float32 and self.gt.bbox.dtype == np.float32
            arith32, out32 = int(both32), int(self.dt.bbox.dtype == np.float32)
        else:
            arith32, out32 = int(self.dt.box3d_dtype == np.float32 and self.gt.box3d_dtype == np.float32), 1
        st = torch.cuda.current_stream().cuda_stream
        _lib.check(self.lib.mr_kitti_overlaps(metric, arith32, out32, self.n_img, self.d_dt_off.data_ptr(), self.d_gt_off.data_ptr(),
                                              d_off.data_ptr(), total, self.d_dt_rows.data_ptr(), self.d_gt_rows.data_ptr(), ov.data_ptr(), st))
        self._ov[metric] = (ov, d_off, ov_off)
        return self._ov[metric]

    def ignore_codes(self, classes, difficulties):
        """clean_data (eval.py:28-80) for every (class, difficulty): int8 [n_cd, total] codes 0 count / 1 ignore / -1 other."""
        g, d = self.gt, self.dt
        gname, dname = np.char.lower(g.name), np.char.lower(d.name)
        gh = g.bbox[:, 3] - g.bbox[:, 1] if len(g.bbox) else np.zeros(0)
        dh = np.abs(d.bbox[:, 3] - d.bbox[:, 1]) if len(d.bbox) else np.zeros(0)
        ig = np.zeros((len(classes) * len(difficulties), len(gname)), np.int8)
        idt = np.zeros((len(classes) * len(difficulties), len(dname)), np.int8)
        for m, cls in enumerate(classes):
            want = _CLASS_NAMES[cls]
            kind = np.where(gname == want, 1, -1)
            if want == 'pedestrian':
                kind = np.where(gname == 'person_sitting', 0, kind)
            elif want == 'car':
                kind = np.where(gname == 'van', 0, kind)
            for l, diff in enumerate(difficulties):
                # scalar comparisons against Python numbers: float64, as under the numpy the reference was written for
                hard = (g.occluded.astype(np.float64) > _MAX_OCCLUSION[diff]) | (g.truncated.astype(np.float64) > _MAX_TRUNCATION[diff]) \
                    | (gh.astype(np.float64) <= _MIN_HEIGHT[diff])
                ig[m * len(difficulties) + l] = np.where((kind == 1) & ~hard, 0, np.where((kind == 0) | (hard & (kind == 1)), 1, -1))
                idt[m * len(difficulties) + l] = np.where(dh.astype(np.float64) < _MIN_HEIGHT[diff], 1, np.where(dname == want, 0, -1))
        return ig, idt


def get_thresholds(scores, num_gt, num_sample_pts=_N_SAMPLE_PTS):
    """eval.py:8-25 — scores (descending) at which the recall first reaches each of the sampling points."""
    s = np.sort(np.asarray(scores, np.float64))[::-1]
    n = len(s)
    out, cur = [], 0.0
    for i in range(n):
        left = (i + 1) / num_gt
        right = (i + 2) / num_gt if i < n - 1 else left
        if i < n - 1 and (right - cur) < (cur - left):
            continue
        out.append(s[i])
        cur += 1 / (num_sample_pts - 1.0)
    return out


def calculate_iou_partly(gt_annos, dt_annos, metric, num_parts=50, _session=None):
    """eval.py:341-416.  Returns (overlaps, parted_overlaps, total_gt_num, total_dt_num) where overlaps[i] is the
    (len(gt_annos[i]), len(dt_annos[i])) float64 block of image i (first argument = rows, as in the reference, which
    calls this with the detections first).  Only the per-image blocks are computed, so `parted_overlaps` is the same list
    (one part per image) and `num_parts` is accepted for compatibility."""
    s = _session if _session is not None else _Session(dt_annos, gt_annos)       # rows of the kernel = its "dt" side
    ov, _, ov_off = s.overlaps(metric)
    host = ov.cpu().numpy()
    blocks = [host[ov_off[i]:ov_off[i + 1]].reshape(int(s.dt.count[i]), int(s.gt.count[i])) for i in range(s.n_img)]
    return blocks, blocks, s.dt.count.copy(), s.gt.count.copy()


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False, num_parts=200,
               _session=None):
    """eval.py:450-567 -> dict(recall, precision, orientation), arrays (class, difficulty, overlap, 41).
    min_overlaps: [num_overlap, metric, class]."""
    s = _session if _session is not None else _Session(gt_annos, dt_annos)
    lib, dev = s.lib, s.dev
    n_cls, n_diff, n_ov = len(current_classes), len(difficultys), len(min_overlaps)
    shape = (n_cls, n_diff, n_ov, _N_SAMPLE_PTS)
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    if s.n_img == 0:
        return dict(recall=recall, precision=precision, orientation=aos)
    ov, d_ov_off, _ = s.overlaps(metric)
    ig, idt = s.ignore_codes(current_classes, difficultys)
    n_valid = (ig == 0).sum(1)
    combos = [(m, l, k) for m in range(n_cls) for l in range(n_diff) for k in range(n_ov)]
    combo_cd = np.array([m * n_diff + l for m, l, k in combos], np.int32)
    combo_mo = np.array([min_overlaps[k, metric, m] for m, l, k in combos], np.float64)
    t = lambda a: _to_dev(a, dev)
    d_ig, d_idt, d_cd, d_mo = t(ig), t(idt), t(combo_cd), t(combo_mo)
    total_gt, total_dt = int(s.gt.off[-1]), int(s.dt.off[-1])
    nc = len(combos)
    alpha32 = int(s.gt.datas_dtype == np.float32 and s.dt_datas_dtype == np.float32)
    dtdata32 = int(s.dt_datas_dtype == np.float32)
    st = torch.cuda.current_stream().cuda_stream

    def launch(second, thr, nthr, match, pr, ws, ws_bytes):
        _lib.check(lib.mr_kitti_match(
            second, metric, int(bool(compute_aos)), alpha32, dtdata32, s.n_img, s.max_det,
            s.d_dt_off.data_ptr(), s.d_gt_off.data_ptr(), d_ov_off.data_ptr(), s.d_dc_off.data_ptr(), total_dt, total_gt,
            ov.data_ptr(), s.d_dt_rows.data_ptr(), s.d_dt_alpha.data_ptr(), s.d_gt_alpha.data_ptr(), s.d_dc_box.data_ptr(),
            d_ig.data_ptr(), d_idt.data_ptr(), nc, d_cd.data_ptr(), d_mo.data_ptr(),
            thr, nthr, match, pr, ws, ws_bytes, st))

    # pass 1: the scores of the true positives at threshold 0 (eval.py:499-516)
    match = torch.full((nc, max(total_gt, 1)), float('nan'), dtype=torch.float64, device=dev)
    launch(0, None, None, match.data_ptr(), None, None, 0)
    match_h = match.cpu().numpy()
    thr_h = np.zeros((nc, _N_SAMPLE_PTS), np.float64)
    nthr_h = np.zeros(nc, np.int32)
    for c in range(nc):
        row = match_h[c]
        th = get_thresholds(row[~np.isnan(row)], n_valid[combo_cd[c]])
        nthr_h[c] = len(th)
        thr_h[c, :len(th)] = th
    # pass 2: tp / fp / fn / similarity at every threshold (eval.py:517-540)
    d_thr, d_nthr = t(thr_h), t(nthr_h)
    pr = torch.zeros((nc, _N_SAMPLE_PTS, 4), dtype=torch.float64, device=dev)
    ws_bytes = int(lib.mr_kitti_match_workspace_bytes(s.n_img, nc))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    launch(1, d_thr.data_ptr(), d_nthr.data_ptr(), None, pr.data_ptr(), ws.data_ptr(), ws_bytes)
    pr_h = pr.cpu().numpy()
    with np.errstate(divide='ignore', invalid='ignore'):
        for c, (m, l, k) in enumerate(combos):
            n = int(nthr_h[c])
            tp, fp, fn, sim = (pr_h[c, :n, j] for j in range(4))
            recall[m, l, k, :n] = tp / (tp + fn)
            precision[m, l, k, :n] = tp / (tp + fp)
            if compute_aos:
                aos[m, l, k, :n] = sim / (tp + fp)
            for arr in (precision, recall) + ((aos,) if compute_aos else ()):
                for i in range(n):                                   # envelope from the right, NaN-propagating like np.max
                    arr[m, l, k, i] = np.max(arr[m, l, k, i:])
    return dict(recall=recall, precision=precision, orientation=aos)


def get_mAP(prec, criteria='R11'):
    """eval.py:570-580: 11-point (samples 0, 4, ..., 40) or 40-point (samples 1..40) interpolated average precision."""
    assert criteria in ['R11', 'R40']
    picks = range(0, prec.shape[-1], 4) if criteria == 'R11' else range(1, prec.shape[-1])
    total = 0
    for i in picks:
        total = total + prec[..., i]
    return total / (11 if criteria == 'R11' else 40) * 100


def do_eval(gt_annos, dt_annos, current_classes, min_overlaps, eval_types=('bbox', 'bev', '3d'), criteria='R11'):
    """eval.py:592-625 -> (mAP_bbox, mAP_bev, mAP_3d, mAP_aos), each [class, difficulty, overlap] (or None)."""
    s = _Session(gt_annos, dt_annos)
    diffs = [0, 1, 2]
    ret = eval_class(gt_annos, dt_annos, current_classes, diffs, 0, min_overlaps, compute_aos=('aos' in eval_types), _session=s)
    mAP_bbox = get_mAP(ret['precision'], criteria)
    mAP_aos = get_mAP(ret['orientation'], criteria) if 'aos' in eval_types else None
    mAP_bev = mAP_3d = None
    if 'bev' in eval_types:
        mAP_bev = get_mAP(eval_class(gt_annos, dt_annos, current_classes, diffs, 1, min_overlaps, _session=s)['precision'], criteria)
    if '3d' in eval_types:
        mAP_3d = get_mAP(eval_class(gt_annos, dt_annos, current_classes, diffs, 2, min_overlaps, _session=s)['precision'], criteria)
    return mAP_bbox, mAP_bev, mAP_3d, mAP_aos


def _class_ids(current_classes):
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    return [_NAME_TO_CLASS[c] if isinstance(c, str) else c for c in current_classes]


def _alpha_valid(dt_annos):
    """eval.py:690-697: AOS is evaluated when the first non-empty detection set carries a real alpha."""
    for anno in dt_annos:
        if anno['alpha'].shape[0] != 0:
            return bool(anno['alpha'][0] != -10)
    return False


def kitti_eval(gt_annos, dt_annos, current_classes, eval_types=['bbox', 'bev', '3d'], criteria='R11'):
    """KITTI evaluation (eval.py:647-769): returns (result text, dict of 'KITTI/<Class>_<3D|BEV|2D>_<difficulty>_<strict|loose>')."""
    assert 'bbox' in eval_types, 'must evaluate bbox at least'
    strict = np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3)
    loose = np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])
    current_classes = _class_ids(current_classes)
    min_overlaps = np.stack([strict, loose], 0)[:, :, current_classes]          # [2, metric, class]
    compute_aos = _alpha_valid(dt_annos)
    types = list(eval_types) + (['aos'] if compute_aos else [])             # (the reference appends to the caller's list)
    mAPbbox, mAPbev, mAP3d, mAPaos = do_eval(gt_annos, dt_annos, current_classes, min_overlaps, types, criteria=criteria)

    out = io.StringIO()
    ret_dict = {}
    difficulty = ['easy', 'moderate', 'hard']
    for j, cls in enumerate(current_classes):
        cname = _CLASS_TO_NAME[cls]
        for i in range(min_overlaps.shape[0]):
            out.write('{} AP@{:.2f}, {:.2f}, {:.2f}:\n'.format(cname, *min_overlaps[i, :, j]))
            for label, arr in (('bbox', mAPbbox), ('bev ', mAPbev), ('3d  ', mAP3d)):
                if arr is not None:
                    out.write(label + ' AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*arr[j, :, i]))
            if compute_aos:
                out.write('aos  AP:{:.2f}, {:.2f}, {:.2f}\n'.format(*mAPaos[j, :, i]))
            for idx in range(3):
                postfix = f'{difficulty[idx]}_' + ('strict' if i == 0 else 'loose')
                for tag, arr in (('3D', mAP3d), ('BEV', mAPbev), ('2D', mAPbbox)):
                    if arr is not None:
                        ret_dict[f'KITTI/{cname}_{tag}_{postfix}'] = arr[j, idx, i]
    if len(current_classes) > 1:
        out.write('\nOverall AP@{}, {}, {}:\n'.format(*difficulty))
        means = {}
        for label, tag, arr in (('bbox', '2D', mAPbbox), ('bev ', 'BEV', mAPbev), ('3d  ', '3D', mAP3d)):
            if arr is not None:
                means[tag] = arr.mean(axis=0)
                out.write(label + ' AP:{:.4f}, {:.4f}, {:.4f}\n'.format(*means[tag][:, 0]))
        if compute_aos:
            out.write('aos  AP:{:.2f}, {:.2f}, {:.2f}\n'.format(*mAPaos.mean(axis=0)[:, 0]))
        for idx in range(3):
            for tag in ('3D', 'BEV', '2D'):
                if tag in means:
                    ret_dict[f'KITTI/Overall_{tag}_{difficulty[idx]}'] = means[tag][idx, 0]
    return out.getvalue(), ret_dict


def kitti_eval_coco_style(gt_annos, dt_annos, current_classes, criteria='R11'):
    """COCO-style evaluation (eval.py:772-842): AP averaged over 10 overlap thresholds per class.  The reference's
    version no longer runs (np.linspace with a float count, eval.py:634; a bool passed where eval types are iterated,
    eval.py:636-638); this follows its evident intent: thresholds linspace(lo, hi, 10), bbox/bev/3d (+ aos)."""
    ranges = {0: (0.5, 0.95, 10), 1: (0.25, 0.7, 10), 2: (0.25, 0.7, 10), 3: (0.5, 0.95, 10), 4: (0.25, 0.7, 10)}
    current_classes = _class_ids(current_classes)
    min_overlaps = np.zeros((10, 3, len(current_classes)))
    for j, cls in enumerate(current_classes):
        lo, hi, num = ranges[cls]
        min_overlaps[:, :, j] = np.linspace(lo, hi, int(num))[:, None]
    compute_aos = _alpha_valid(dt_annos)
    types = ['bbox', 'bev', '3d'] + (['aos'] if compute_aos else [])
    mAPs = [None if m is None else m.mean(-1) for m in do_eval(gt_annos, dt_annos, current_classes, min_overlaps, types, criteria=criteria)]
    mAPbbox, mAPbev, mAP3d, mAPaos = mAPs
    out = io.StringIO()
    for j, cls in enumerate(current_classes):
        lo, hi, num = ranges[cls]
        print(f'{_CLASS_TO_NAME[cls]} ' + 'coco AP@{:.2f}:{:.2f}:{:.2f}:'.format(lo, (hi - lo) / (num - 1), hi), file=out)
        print(f'bbox AP:{mAPbbox[j, 0]:.2f}, {mAPbbox[j, 1]:.2f}, {mAPbbox[j, 2]:.2f}', file=out)
        print(f'bev  AP:{mAPbev[j, 0]:.2f}, {mAPbev[j, 1]:.2f}, {mAPbev[j, 2]:.2f}', file=out)
        print(f'3d   AP:{mAP3d[j, 0]:.2f}, {mAP3d[j, 1]:.2f}, {mAP3d[j, 2]:.2f}', file=out)
        if compute_aos:
            print(f'aos  AP:{mAPaos[j, 0]:.2f}, {mAPaos[j, 1]:.2f}, {mAPaos[j, 2]:.2f}', file=out)
    return out.getvalue()


# ------------------------------------------------------------------------------------------------
# N3: the KITTI wire format of the pipeline's results (monorun/datasets/kitti3d_dataset.py)

def cam_t_vec_from_calib(calib):
    """kitti3d_dataset.py:117-123: P2 = K [I | t]  ->  (K (3,3), t = K^-1 P2[:, 3]) — the camera offset the 3-D
    results are shifted by before they are written."""
    calib = np.asarray(calib, np.float64)
    K = calib[:, :3]
    return K, np.linalg.solve(K, calib[:, 3])                     # K is upper triangular; same solution as solve_triangular


def format_results(results, gt_ann_infos, classes):
    """kitti3d_dataset.py:230-270: per-image dict(bbox_results=[(n_c,5)] per class, bbox_3d_results=[(n_c,8)] per class
    with rows [l,h,w, x,y,z, ry, score]) -> KITTI detection annotation dicts, sorted by descending 3-D score, locations
    shifted by the image's cam_t_vec, alpha = ry - atan2(x, z + 0.27)."""
    det_annos = []
    for result, info in zip(results, gt_ann_infos):
        b2, b3 = result['bbox_results'], result['bbox_3d_results']
        name = np.array([classes[i] for i, per_class in enumerate(b2) for _ in per_class])
        n = name.shape[0]
        all2 = np.concatenate(b2, axis=0)
        all3 = np.concatenate(b3, axis=0).copy()
        all3[:, 3:6] -= info['cam_t_vec']
        order = all3[:, 7].argsort()[::-1]
        all2, all3, name = all2[order], all3[order], name[order]
        loc, ry = all3[:, 3:6], all3[:, 6]
        det_annos.append(dict(
            name=name, truncated=np.full(n, -1, dtype=np.int8), occluded=np.full(n, -1, dtype=np.int8),
            alpha=ry - np.arctan2(loc[:, 0], loc[:, 2] + 0.27), bbox=all2[:, :4], dimensions=all3[:, :3], location=loc,
            rotation_y=ry, score=all3[:, 7]))
    return det_annos


def format_gt_anno(ann_info, classes):
    """kitti3d_dataset.py:272-305: label dict of one image (objects first, then the DontCare regions)."""
    n_obj, n_dc = len(ann_info['bboxes']), len(ann_info['bboxes_ignore'])
    n = n_obj + n_dc
    f32 = np.float32
    b3 = np.asarray(ann_info['bboxes_3d_eval']).reshape(n_obj, -1)
    return dict(
        name=[classes[label] for label in ann_info['labels']] + ['DontCare'] * n_dc,
        truncated=np.array(list(ann_info['truncation']) + [-1] * n_dc, dtype=f32),
        occluded=np.array(list(ann_info['occlusion']) + [-1] * n_dc, dtype=f32),
        alpha=np.array(list(ann_info['alpha']) + [-10] * n_dc, dtype=f32),
        bbox=np.concatenate((np.asarray(ann_info['bboxes']).reshape(n_obj, 4), np.asarray(ann_info['bboxes_ignore']).reshape(n_dc, 4)), axis=0),
        dimensions=np.concatenate((b3[:, :3], np.full((n_dc, 3), -1, dtype=f32)), axis=0),
        location=np.concatenate((b3[:, 3:6], np.full((n_dc, 3), -1000, dtype=f32)), axis=0),
        rotation_y=np.concatenate((b3[:, 6], np.full(n_dc, -10, dtype=f32)), axis=0),
        score=np.zeros(n, dtype=f32),
        index=np.concatenate((np.arange(n_obj, dtype=np.int32), np.full(n_dc, -1, dtype=np.int32)), axis=0),
        group_ids=np.arange(n, dtype=np.int32))


def write_result_files(results, filenames, result_dir):
    """kitti3d_dataset.py:307-325: one '<stem>.txt' per image, KITTI column order
    (name truncated occluded alpha bbox[4] h w l x y z ry score); an existing directory is replaced."""
    if os.path.exists(result_dir):
        shutil.rmtree(result_dir)
    os.mkdir(result_dir)
    for result, filename in zip(results, filenames):
        stem, _ = os.path.splitext(filename)
        cols = np.concatenate(
            (result['name'].reshape(-1, 1), result['truncated'].reshape(-1, 1), result['occluded'].reshape(-1, 1),
             result['alpha'].reshape(-1, 1), result['bbox'], result['dimensions'][:, [1, 2, 0]], result['location'],
             result['rotation_y'].reshape(-1, 1), result['score'].reshape(-1, 1)), axis=1)
        np.savetxt(os.path.join(result_dir, stem + '.txt'), cols, delimiter=' ', fmt='%s')


# ------------------------------------------------------------------------------------------------
# KITTI files -> annotation infos, and the dataset-level `evaluate` (monorun/datasets/kitti3d_dataset.py)

def open_calib_file(calib_file, cam=2):
    """kitti3d_dataset.py:40-47: row `cam` (P0..P3) of a KITTI calib file as a (3,4) float32 projection matrix."""
    assert 0 <= cam <= 3
    with open(calib_file) as f:
        row = f.readlines()[cam]
    return np.array([float(v) for v in row.strip().split(' ')[1:]], dtype=np.float32).reshape((3, 4))


def open_label_file(path):
    """kitti3d_dataset.py:49-56: list of label rows [name, truncation, occlusion(int), alpha, x1, y1, x2, y2, h, w, l, x, y, z, ry]."""
    rows = []
    with open(path) as f:
        for line in f:
            parts = line.strip().split(' ')
            if parts == ['']:
                continue
            rows.append([p if i == 0 else int(float(p)) if i == 2 else float(p) for i, p in enumerate(parts)])
    return rows


def parse_ann_info(label, calib, classes=('Car', 'Pedestrian', 'Cyclist')):
    """kitti3d_dataset.py:116-178 (`_parse_ann_info`): the per-image annotation info.  Only objects of `classes` are kept
    (plus DontCare boxes as `bboxes_ignore`); 3-D boxes are [l,h,w, x,y,z, ry]: `bboxes_3d` in camera space (shifted by
    cam_t_vec), `bboxes_3d_eval` in the label file's reference space.  label=None (test mode): calibration only."""
    K, cam_t_vec = cam_t_vec_from_calib(calib)
    ann = dict(cam_intrinsic=K.astype(np.float32), cam_t_vec=cam_t_vec.astype(np.float32))
    if label is None:
        return ann
    ids, boxes, labels, ignore, trunc, occ, alpha, b3 = [], [], [], [], [], [], [], []
    for object_id, inst in enumerate(label):
        if inst[0] in classes:
            ids.append(object_id); labels.append(classes.index(inst[0]))
            trunc.append(inst[1]); occ.append(inst[2]); alpha.append(inst[3]); boxes.append(inst[4:8]); b3.append(inst[8:15])
        elif inst[0].lower() == 'dontcare':
            ignore.append(inst[4:8])
    boxes = np.array(boxes, dtype=np.float32).reshape(-1, 4)
    b3 = np.array(b3, dtype=np.float32).reshape(-1, 7)
    b3[:, [0, 1, 2]] = b3[:, [2, 0, 1]]                              # hwl -> lhw
    b3_eval = b3.copy()
    b3[:, 3:6] += ann['cam_t_vec']
    ann.update(object_ids=np.array(ids, dtype=np.int64), bboxes=boxes, labels=np.array(labels, dtype=np.int64),
               bboxes_ignore=np.array(ignore, dtype=np.float32).reshape(-1, 4), truncation=trunc, occlusion=occ, alpha=alpha,
               bboxes_3d=b3, bboxes_3d_eval=b3_eval)
    return ann


def evaluate(results, gt_ann_infos, classes=('Car', 'Pedestrian', 'Cyclist'), metric=('bbox', 'bev', '3d'), filenames=None,
             result_dir=None, summary_file=None, print_summary=False, use_r40=True):
    """`KITTI3DDataset.evaluate` (kitti3d_dataset.py:195-228): pipeline results -> KITTI detection annotations (+ result
    files under result_dir/data) -> kitti_eval against the labels.  Returns (ap_dict, result text, detection annos);
    infos without labels (test mode) give an empty dict."""
    det_annos = format_results(results, gt_ann_infos, classes)
    if result_dir is not None:
        if not os.path.exists(result_dir):
            os.mkdir(result_dir)
        write_result_files(det_annos, filenames, os.path.join(result_dir, 'data'))
    if not all('bboxes' in info for info in gt_ann_infos):
        return dict(), '', det_annos
    gt_annos = [format_gt_anno(info, classes) for info in gt_ann_infos]
    text, ap = kitti_eval(gt_annos, det_annos, list(classes), eval_types=list(metric), criteria='R40' if use_r40 else 'R11')
    if print_summary:
        print('\n' + text)
    if summary_file is not None:
        with open(summary_file, 'w') as f:
            f.write(text)
    return ap, text, det_annos
