"""Dump hook for the reference's test path: what ``MonoRUnRoIHead.simple_test`` hands to its pose stage
(/root/reference/monorun/models/roi_heads/monorun_roi_head.py:442-534), one ``<image id>.npz`` per image, in the format
``tools/kitti_val.py --dumps`` reads — so BASELINE configs 3 / 4 (KITTI val end to end) can be run through this repo's
post-head tail on a box that has the dataset and checkpoints but no CUDA-only dependencies.

    from monorun_amd.integration import PoseStageDump
    model = build_detector(cfg.model, ...); load_checkpoint(model, ckpt)         # the reference's own tools/test.py set-up
    with PoseStageDump(model.roi_head, 'dumps/'):                                 # wraps four call sites, restores them on exit
        single_gpu_test(model, data_loader)
    # then:  python tools/kitti_val.py --labels <label_2> --calib <calib> --ids val.txt --dumps dumps/

What is recorded (all float32 / int64 numpy arrays, n = detections of the image):
    all_pred (n, 2*C*5, h, w)  output of ``noc_head.conv_final``           (fcn_noc_decoder.py:224, before flip_correction)
    labels (n,), flip (bool)   ``det_labels`` / ``img_metas[0]['flip']``   (monorun_roi_head.py:463-470, :509)
    dim (n,3), dim_var (n,3)   ``reg_results['dim_pred' | 'dim_var']``     (:492, :504-507; dim_var only if the head predicts it)
    rois (n,4)                 ``bbox_3d_rois[:, 1:]`` (test-scale xyxy)    (:473-478)
    bboxes (n,4), scores (n,)  ``det_bboxes`` split                         (:463)
    cam_intrinsic (3,3), img_shape, filename
    scores_ref (n,)            the reference's own final scores (``_score_forward`` x class score, :535-547) when its pose stage
                               ran — lets the harness rank detections exactly as the reference did
The hook only reads; the wrapped functions return what they returned before.
"""
import os

import numpy as np


def _np(t):
    return t.detach().float().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)


class PoseStageDump:
    def __init__(self, roi_head, out_dir, image_id=None):
        """roi_head: a built MonoRUnRoIHead (anything with .noc_head.conv_final, .bbox_head.get_bboxes, ._reg_forward,
        ._score_forward and .simple_test works).  image_id: optional callable img_meta -> file stem (default: the stem of
        img_meta['filename'] / ['ori_filename'])."""
        self.head, self.out_dir = roi_head, out_dir
        self.image_id = image_id or (lambda m: os.path.splitext(os.path.basename(m.get('ori_filename') or m['filename']))[0])
        self._orig, self._hook, self._cur = [], None, {}
        self.written = []

    # -- context manager -------------------------------------------------------------------------------------
    def __enter__(self):
        os.makedirs(self.out_dir, exist_ok=True)
        h = self.head
        self._hook = h.noc_head.conv_final.register_forward_hook(lambda mod, inp, out: self._cur.__setitem__('all_pred', _np(out)))
        self._wrap(h.bbox_head, 'get_bboxes', self._after_get_bboxes)
        self._wrap(h, '_reg_forward', self._after_reg_forward)
        self._wrap(h, '_score_forward', self._after_score_forward)
        self._wrap(h, 'simple_test', self._after_simple_test, before=self._before_simple_test)
        return self

    def __exit__(self, *exc):
        for obj, name, fn, was_instance_attr in reversed(self._orig):
            if was_instance_attr:
                setattr(obj, name, fn)
            else:
                delattr(obj, name)                                      # the class's own method shows through again
        self._orig.clear()
        if self._hook is not None:
            self._hook.remove()
            self._hook = None
        return False

    def _wrap(self, obj, name, after, before=None):
        orig = getattr(obj, name)
        self._orig.append((obj, name, orig, name in getattr(obj, '__dict__', {})))

        def wrapped(*a, **kw):
            if before is not None:
                before(*a, **kw)
            out = orig(*a, **kw)
            after(out, *a, **kw)
            return out
        setattr(obj, name, wrapped)

    # -- the four call sites ---------------------------------------------------------------------------------
    def _before_simple_test(self, x, proposal_list, img_metas, proposals=None, coord_2d=None, cam_intrinsic=None, rescale=False):
        self._cur = dict(meta=img_metas[0], rescale=bool(rescale))
        if cam_intrinsic is not None:
            self._cur['cam_intrinsic'] = _np(cam_intrinsic[0][0]).reshape(3, 3)

    def _after_get_bboxes(self, out, *a, **kw):
        det_bboxes, det_labels = out
        self._cur['det_bboxes'], self._cur['labels'] = _np(det_bboxes), _np(det_labels).astype(np.int64)

    def _after_reg_forward(self, out, x, rois, labels=None, *a, **kw):
        self._cur['rois'] = _np(rois)[:, 1:5]
        self._cur['dim'] = _np(out['dim_pred'])
        if out.get('dim_var') is not None:
            self._cur['dim_var'] = _np(out['dim_var'])

    def _after_score_forward(self, out, *a, **kw):
        self._cur['score_head'] = _np(out['scores']).reshape(-1)

    def _after_simple_test(self, out, *a, **kw):
        c = self._cur
        if 'all_pred' not in c or 'rois' not in c or len(c.get('labels', ())) == 0:
            return                                                      # no detections: nothing reaches the pose stage
        m = c['meta']
        db = c['det_bboxes']
        rec = dict(all_pred=c['all_pred'].astype(np.float32), labels=c['labels'], flip=np.bool_(bool(m.get('flip', False))),
                   dim=c['dim'].astype(np.float32), rois=c['rois'].astype(np.float32), bboxes=db[:, :4].astype(np.float32),
                   scores=db[:, 4].astype(np.float32), img_shape=np.asarray(m['img_shape'][:2], np.float32),
                   filename=np.str_(m.get('ori_filename') or m.get('filename', '')))
        for k in ('dim_var', 'cam_intrinsic'):
            if k in c:
                rec[k] = c[k].astype(np.float32)
        if 'score_head' in c:
            cfg = getattr(self.head, 'test_cfg', None)
            s = c['score_head']
            if getattr(getattr(self.head, 'score_head', None), 'pre_sigmoid', False):
                s = 1.0 / (1.0 + np.exp(-s))
            rec['scores_ref'] = (db[:, 4] * s if getattr(cfg, 'mult_2d_score', False) else s).astype(np.float32)
        path = os.path.join(self.out_dir, self.image_id(m) + '.npz')
        np.savez(path, **rec)
        self.written.append(path)
