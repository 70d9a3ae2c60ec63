"""CPU oracle of the KITTI object evaluator — TEST INFRASTRUCTURE ONLY.

Restates, in plain Python/numpy loops, what monorun/core/evaluation/kitti_utils/eval.py computes (the
reference runs it under numba, with a numba-CUDA rotated-IoU kernel, rotate_iou.py).  Only tests/ and
__graft_entry__.smoke() may import this file; the product (monorun_amd/evaluation.py + the HIP kernels) never does.

Pinned by golden fixture G6 (tests/golden/g6_kitti_eval.npz): the reference's own eval.py executed as plain Python on
a synthetic label/detection set — result text, AP dict, precision/recall/orientation curves and per-image overlaps.
Rotated intersections are computed here by fp64 Sutherland-Hodgman clipping and then rounded to float32 (the
reference clips in float32 with a different vertex-sorting algorithm): overlaps agree to ~1e-6, not bit for bit.
"""
import math

import numpy as np

CLASS_NAMES = ('car', 'pedestrian', 'cyclist')                      # eval.py:29
MIN_HEIGHT = (40, 25, 25)                                           # eval.py:30
MAX_OCCLUSION = (0, 1, 2)                                           # eval.py:31
MAX_TRUNCATION = (0.15, 0.3, 0.5)                                   # eval.py:32
N_SAMPLE_PTS = 41                                                   # eval.py:481
NO_DET = -10000000                                                  # eval.py:186


# ------------------------------------------------------------------------------- overlaps -----
def _corners(box):
    """rotate_iou.py:204-227: corners of [cx, cy, dx, dy, angle] (clockwise order, rotated clockwise)."""
    cx, cy, dx, dy, ang = (float(v) for v in box)
    c, s = math.cos(ang), math.sin(ang)
    loc = ((-dx / 2, -dy / 2), (-dx / 2, dy / 2), (dx / 2, dy / 2), (dx / 2, -dy / 2))
    return [(c * x + s * y + cx, -s * x + c * y + cy) for x, y in loc]


def _area2(poly):
    return sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1] for i in range(len(poly)))


def _clip(subject, clipper):
    out = subject
    n = len(clipper)
    for i in range(n):
        a, b = clipper[i], clipper[(i + 1) % n]
        inp, out = out, []
        if not inp:
            break

        def side(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for k in range(len(inp)):
            p, q = inp[k], inp[(k + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    return out


def rotated_intersection(box_a, box_b):
    """Area of the intersection of two rotated rectangles [cx, cy, dx, dy, angle] (rotate_iou.py:230-253)."""
    pa, pb = _corners(box_a), _corners(box_b)
    if _area2(pa) < 0:
        pa = pa[::-1]
    if _area2(pb) < 0:
        pb = pb[::-1]
    poly = _clip(pa, pb)
    return abs(_area2(poly)) / 2 if len(poly) >= 3 else 0.0


def rotated_overlap(query, box, criterion):
    """devRotateIoUEval(query, box, criterion) (rotate_iou.py:256-281), inputs float32, result rounded to float32."""
    q, b = np.asarray(query, np.float32), np.asarray(box, np.float32)
    a1, a2 = float(q[2] * q[3]), float(b[2] * b[3])
    inter = rotated_intersection(q, b)
    if criterion == -1:
        v = inter / (a1 + a2 - inter)
    elif criterion == 0:
        v = inter / a1
    elif criterion == 1:
        v = inter / a2
    else:
        v = inter
    return float(np.float32(v))


def image_overlap(box, query, criterion=-1):
    """One entry of image_box_overlap (eval.py:84-112), in the dtype of the inputs."""
    t = np.result_type(box.dtype, query.dtype).type
    q_area = t((query[2] - query[0]) * (query[3] - query[1]))
    iw = t(min(box[2], query[2]) - max(box[0], query[0]))
    if not iw > 0:
        return t(0)
    ih = t(min(box[3], query[3]) - max(box[1], query[1]))
    if not ih > 0:
        return t(0)
    b_area = t((box[2] - box[0]) * (box[3] - box[1]))
    ua = {-1: t(t(b_area + q_area) - t(iw * ih)), 0: b_area, 1: q_area}.get(criterion, t(1.0))
    return t(t(iw * ih) / ua)


def overlaps_one_image(dt, gt, metric):
    """(n_dt, n_gt) float64 overlaps of one image, the block eval_class reads as overlaps[i]
    (calculate_iou_partly is called with detections first, eval.py:477; metric 0 bbox, 1 bev, 2 3d)."""
    nd, ng = len(dt['name']), len(gt['name'])
    ov = np.zeros((nd, ng), np.float64)
    for j in range(nd):
        for i in range(ng):
            if metric == 0:
                res = image_overlap(dt['bbox'][j], gt['bbox'][i])                       # eval.py:364
                ov[j, i] = np.asarray(res, dt['bbox'].dtype)                            # zeros(dtype=boxes.dtype), eval.py:87
                continue
            bj = np.array([dt['location'][j, 0], dt['location'][j, 2], dt['dimensions'][j, 0], dt['dimensions'][j, 2], dt['rotation_y'][j]])
            gi = np.array([gt['location'][i, 0], gt['location'][i, 2], gt['dimensions'][i, 0], gt['dimensions'][i, 2], gt['rotation_y'][i]])
            if metric == 1:
                ov[j, i] = rotated_overlap(gi, bj, -1)                                   # eval.py:366-382
            else:
                # d3_box_overlap (eval.py:121-158): BEV intersection area x height overlap, camera frame (y = bottom, down)
                t = np.result_type(dt['location'].dtype, gt['location'].dtype, np.float32).type
                rinc = np.float32(rotated_overlap(gi, bj, 2))
                val = np.float32(0.0)
                if rinc > 0:
                    yb, hb = t(dt['location'][j, 1]), t(dt['dimensions'][j, 1])
                    yq, hq = t(gt['location'][i, 1]), t(gt['dimensions'][i, 1])
                    iw = t(min(yb, yq) - max(t(yb - hb), t(yq - hq)))
                    if iw > 0:
                        a1 = t(t(t(dt['dimensions'][j, 0]) * hb) * t(dt['dimensions'][j, 2]))
                        a2 = t(t(t(gt['dimensions'][i, 0]) * hq) * t(gt['dimensions'][i, 2]))
                        inc = t(iw * t(rinc))
                        val = np.float32(t(inc / t(t(a1 + a2) - inc)))
                ov[j, i] = val
    return ov


# ------------------------------------------------------------------------------- filtering -----
def clean(gt, dt, cls, difficulty):
    """clean_data (eval.py:28-80): which labels / detections count (0), are ignored (1) or belong to another class (-1)."""
    want = CLASS_NAMES[cls]
    ign_gt, ign_dt, dc, n_valid = [], [], [], 0
    for i in range(len(gt['name'])):
        nm = str(gt['name'][i]).lower()
        h = gt['bbox'][i, 3] - gt['bbox'][i, 1]
        if nm == want:
            kind = 1
        elif (want == 'pedestrian' and nm == 'person_sitting') or (want == 'car' and nm == 'van'):
            kind = 0
        else:
            kind = -1
        # Python-number thresholds: float64 comparisons (numpy < 2 semantics, the reference's era)
        hard = float(gt['occluded'][i]) > MAX_OCCLUSION[difficulty] or float(gt['truncated'][i]) > MAX_TRUNCATION[difficulty] or float(h) <= MIN_HEIGHT[difficulty]
        if kind == 1 and not hard:
            ign_gt.append(0)
            n_valid += 1
        elif kind == 0 or (hard and kind == 1):
            ign_gt.append(1)
        else:
            ign_gt.append(-1)
        if str(gt['name'][i]) == 'DontCare':
            dc.append(gt['bbox'][i])
    for j in range(len(dt['name'])):
        h = abs(dt['bbox'][j, 3] - dt['bbox'][j, 1])
        if float(h) < MIN_HEIGHT[difficulty]:
            ign_dt.append(1)
        elif str(dt['name'][j]).lower() == want:
            ign_dt.append(0)
        else:
            ign_dt.append(-1)
    dc = np.stack(dc, 0).astype(np.float64) if dc else np.zeros((0, 4))
    return n_valid, np.array(ign_gt, np.int64), np.array(ign_dt, np.int64), dc


# ------------------------------------------------------------------------------- matching ------
def match_image(ov, gt_alpha, dt_alpha, dt_score, dt_bbox, ign_gt, ign_dt, dc, metric, min_overlap, thresh=0.0,
                second_pass=False, aos=False):
    """compute_statistics_jit (eval.py:161-279) for one image -> (tp, fp, fn, similarity, matched scores)."""
    nd, ng = len(ign_dt), len(ign_gt)
    taken = [False] * nd
    below = [second_pass and dt_score[j] < thresh for j in range(nd)]
    tp = fp = fn = 0
    sim = 0
    scores, deltas = [], []
    for i in range(ng):
        if ign_gt[i] == -1:
            continue
        pick, best_score, best_ov, via_ignored = -1, NO_DET, 0, False
        for j in range(nd):
            if ign_dt[j] == -1 or taken[j] or below[j]:
                continue
            o = ov[j, i]
            if not o > min_overlap:
                continue
            if not second_pass:
                if dt_score[j] > best_score:
                    pick, best_score = j, dt_score[j]
            elif ign_dt[j] == 0 and (o > best_ov or via_ignored):
                pick, best_score, best_ov, via_ignored = j, 1, o, False
            elif ign_dt[j] == 1 and best_score == NO_DET:
                pick, best_score, via_ignored = j, 1, True
        if best_score == NO_DET:
            if ign_gt[i] == 0:
                fn += 1
        elif ign_gt[i] == 1 or ign_dt[pick] == 1:
            taken[pick] = True
        else:
            tp += 1
            scores.append(dt_score[pick])
            if aos:
                deltas.append(gt_alpha[i] - dt_alpha[pick])
            taken[pick] = True
    if second_pass:
        live = [not (taken[j] or ign_dt[j] == -1 or ign_dt[j] == 1 or below[j]) for j in range(nd)]
        fp = sum(live)
        stuff = 0
        if metric == 0:
            for k in range(dc.shape[0]):
                for j in range(nd):
                    if taken[j] or ign_dt[j] in (-1, 1) or below[j]:
                        continue
                    # image_box_overlap(dt_bboxes, dc_bboxes, 0): float64 arithmetic stored in the detections' dtype
                    if np.asarray(image_overlap(np.asarray(dt_bbox[j], np.float64), dc[k], 0), dt_bbox.dtype) > min_overlap:
                        taken[j] = True
                        stuff += 1
        fp -= stuff
        if aos:
            if tp > 0 or fp > 0:
                sim = 0.0
                for d in deltas:                                   # numba's np.sum is a sequential loop
                    sim += (1.0 + math.cos(d)) / 2.0
            else:
                sim = -1
    return tp, fp, fn, sim, np.array(scores, np.float64)


def sample_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """get_thresholds (eval.py:8-25): the scores at which recall crosses the 41 sampling points."""
    s = np.sort(np.asarray(scores, np.float64))[::-1]
    cur, out = 0.0, []
    for i, sc in enumerate(s):
        left = (i + 1) / num_gt
        right = (i + 2) / num_gt if i < len(s) - 1 else left
        if (right - cur) < (cur - left) and i < len(s) - 1:
            continue
        out.append(sc)
        cur += 1 / (num_sample_pts - 1.0)
    return np.array(out, np.float64)


# ------------------------------------------------------------------------------- curves --------
def eval_class(gt_annos, dt_annos, classes, difficulties, metric, min_overlaps, compute_aos=False):
    """eval_class (eval.py:450-567) -> dict(recall, precision, orientation), each (class, difficulty, overlap, 41)."""
    n = len(gt_annos)
    ovs = [overlaps_one_image(dt_annos[i], gt_annos[i], metric) for i in range(n)]
    shape = (len(classes), len(difficulties), len(min_overlaps), N_SAMPLE_PTS)
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(classes):
        for l, diff in enumerate(difficulties):
            prep = [clean(gt_annos[i], dt_annos[i], cls, diff) for i in range(n)]
            n_valid = sum(p[0] for p in prep)
            for k, mo in enumerate(min_overlaps[:, metric, m]):
                def run(i, thr, second):
                    g, d = gt_annos[i], dt_annos[i]
                    t_dt = np.result_type(d['bbox'].dtype, d['alpha'].dtype, d['score'].dtype)     # dtype of dt_datas (eval.py:439-442)
                    return match_image(ovs[i], g['alpha'], d['alpha'], d['score'], d['bbox'].astype(t_dt), prep[i][1], prep[i][2], prep[i][3],
                                       metric, mo, thr, second, compute_aos and second)
                pool = np.concatenate([run(i, 0.0, False)[4] for i in range(n)]) if n else np.zeros(0)
                thr = sample_thresholds(pool, n_valid)
                pr = np.zeros((len(thr), 4))
                for i in range(n):
                    for t, th in enumerate(thr):
                        tp, fp, fn, sim, _ = run(i, th, True)
                        pr[t, 0] += tp; pr[t, 1] += fp; pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                with np.errstate(divide='ignore', invalid='ignore'):
                    for t in range(len(thr)):
                        recall[m, l, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 2])
                        precision[m, l, k, t] = pr[t, 0] / (pr[t, 0] + pr[t, 1])
                        if compute_aos:
                            aos[m, l, k, t] = pr[t, 3] / (pr[t, 0] + pr[t, 1])
                for t in range(len(thr)):                            # monotone envelope from the right (eval.py:547-554)
                    precision[m, l, k, t] = np.max(precision[m, l, k, t:])
                    recall[m, l, k, t] = np.max(recall[m, l, k, t:])
                    if compute_aos:
                        aos[m, l, k, t] = np.max(aos[m, l, k, t:])
    return dict(recall=recall, precision=precision, orientation=aos)


def get_map(prec, criteria='R11'):
    """get_mAP (eval.py:570-580): 11-point (every 4th sample from 0) or 40-point (samples 1..40) interpolated AP."""
    idx = range(0, prec.shape[-1], 4) if criteria == 'R11' else range(1, prec.shape[-1])
    tot = 0
    for i in idx:
        tot = tot + prec[..., i]
    return tot / (11 if criteria == 'R11' else 40) * 100


KITTI_MIN_OVERLAPS = np.stack([                                     # eval.py:666-673, [strict|loose, metric, class]
    np.array([[0.7, 0.5, 0.5, 0.7, 0.5]] * 3),
    np.array([[0.7, 0.5, 0.5, 0.7, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25], [0.5, 0.25, 0.25, 0.5, 0.25]])], 0)
CLASS_TO_NAME = {0: 'Car', 1: 'Pedestrian', 2: 'Cyclist', 3: 'Van', 4: 'Person_sitting'}


def kitti_ap(gt_annos, dt_annos, classes, criteria='R40'):
    """The numbers kitti_eval reports (eval.py:647-769): dict metric -> (class, difficulty, strict|loose) AP arrays."""
    ids = [c if isinstance(c, int) else {v: k for k, v in CLASS_TO_NAME.items()}[c] for c in classes]
    mo = KITTI_MIN_OVERLAPS[:, :, ids]
    aos = any(len(a['alpha']) for a in dt_annos) and next(a for a in dt_annos if len(a['alpha']))['alpha'][0] != -10
    out = {}
    r = eval_class(gt_annos, dt_annos, ids, [0, 1, 2], 0, mo, compute_aos=aos)
    out['bbox'] = get_map(r['precision'], criteria)
    if aos:
        out['aos'] = get_map(r['orientation'], criteria)
    out['bev'] = get_map(eval_class(gt_annos, dt_annos, ids, [0, 1, 2], 1, mo)['precision'], criteria)
    out['3d'] = get_map(eval_class(gt_annos, dt_annos, ids, [0, 1, 2], 2, mo)['precision'], criteria)
    return out
