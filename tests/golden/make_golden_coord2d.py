#!/usr/bin/env python3
"""G9: the image's coordinate map as the reference's data pipeline produces it — by IMPORTING the reference.

    python tests/golden/make_golden_coord2d.py            # rewrites tests/golden/g9_coord2d.npz

Runs only in the authoring container (needs /root/reference, read-only).  What is imported (by file path, under mmcv / mmdet stubs
— the real packages are not in this image): /root/reference/monorun/datasets/pipelines/loading.py, whose
`LoadAnnotations3D._gen_coord_2d` (:67-78) is CALLED for every case.  What is restated here, because it lives in mmcv (third
party, absent): `mmcv.impad(dense, shape=pad_shape, padding_mode='edge')` as Pad3D applies it to the dense fields
(transforms.py:55-74; pad_shape = the image shape rounded up to size_divisor = 32, padding on the bottom and on the right,
cv2.BORDER_REPLICATE = numpy's mode='edge') and `mmcv.imflip(dense, 'horizontal')` of RandomFlip3D (transforms.py:36-52; numpy's
flip of the column axis) — both are pure index operations, no arithmetic.  The fixture is data: shapes in, arrays (small cases) or
border strips + SHA-256 of the whole array (KITTI-sized cases) out; layout (2, Hp, Wp) = ImageToTensor of the (Hp, Wp, 2) field."""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/monorun'
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_loading():
    class _Reg:
        def register_module(self, *a, **k):
            return lambda cls: cls
    mmcv = types.ModuleType('mmcv'); sys.modules['mmcv'] = mmcv
    md = types.ModuleType('mmdet'); md.__path__ = []; sys.modules['mmdet'] = md
    mdd = types.ModuleType('mmdet.datasets'); mdd.PIPELINES = _Reg(); sys.modules['mmdet.datasets'] = mdd
    spec = importlib.util.spec_from_file_location('ref_loading', os.path.join(REF, 'datasets/pipelines/loading.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def impad_edge_to_divisor(dense, divisor=32):
    h, w = dense.shape[:2]
    hp, wp = -(-h // divisor) * divisor, -(-w // divisor) * divisor
    return np.pad(dense, ((0, hp - h), (0, wp - w), (0, 0)), mode='edge')


def main():
    ref = load_reference_loading()
    out = {}
    cases = [(37, 61, False), (37, 61, True), (64, 96, False), (33, 1, False), (375, 1242, False), (375, 1242, True), (370, 1224, False)]
    out['cases'] = np.array([[h, w, int(f)] for h, w, f in cases], np.int64)
    for n, (h, w, flip) in enumerate(cases):
        results = dict(img_shape=(h, w, 3), ori_shape=(h, w, 3))
        results = ref.LoadAnnotations3D._gen_coord_2d(results)               # the reference's own function
        dense = results['coord_2d']
        assert dense.shape == (h, w, 2) and dense.dtype == np.float32 and results['dense_fields'] == ['coord_2d']
        if flip:
            dense = np.flip(dense, axis=1)                                    # mmcv.imflip(dense, 'horizontal')
        padded = impad_edge_to_divisor(dense)                                 # mmcv.impad(dense, shape=pad_shape, padding_mode='edge')
        chw = np.ascontiguousarray(np.moveaxis(padded, -1, 0))                # ImageToTensor: (2, Hp, Wp)
        out[f'shape_{n}'] = np.array(chw.shape, np.int64)
        out[f'sha256_{n}'] = np.frombuffer(hashlib.sha256(chw.tobytes()).digest(), np.uint8)
        if h * w <= 8192:
            out[f'map_{n}'] = chw
        else:                                                                 # the last 40 rows and columns (the padding and what it replicates), and the first 4
            out[f'bottom_{n}'] = chw[:, -40:, :].copy(); out[f'right_{n}'] = chw[:, :, -40:].copy()
            out[f'top_{n}'] = chw[:, :4, :].copy(); out[f'left_{n}'] = chw[:, :, :4].copy()
    np.savez_compressed(os.path.join(OUT, 'g9_coord2d.npz'), **out)
    print('wrote g9_coord2d.npz:', {k: v.shape for k, v in out.items() if k.startswith('shape') is False and not k.startswith('sha')}.__len__(), 'arrays,',
          os.path.getsize(os.path.join(OUT, 'g9_coord2d.npz')), 'bytes')


if __name__ == '__main__':
    main()
