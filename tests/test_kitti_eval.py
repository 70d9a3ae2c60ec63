"""N2/N3: KITTI evaluator and result formatting.

CPU: the oracle (oracle/kitti_eval.py) against golden G6 = the reference's own eval.py run as plain Python on a synthetic
label/detection set (tests/golden/make_golden.py::make_g6); host-side formatting.  GPU: monorun_amd.evaluation (HIP
kernels behind mr_kitti_overlaps / mr_kitti_match) against G6 and against the oracle on further seeded sets.
"""
import os

import numpy as np
import pytest

from monorun_amd import synthetic as syn
from oracle import kitti_eval as ke

GOLD = os.path.join(os.path.dirname(__file__), 'golden', 'g6_kitti_eval.npz')
CLASSES = ['Car', 'Pedestrian', 'Cyclist']


@pytest.fixture(scope='module')
def g6():
    z = np.load(GOLD)
    return z, syn.unpack_kitti_annos(z, 'gt_'), syn.unpack_kitti_annos(z, 'dt_')


# ------------------------------------------------------------------------------------- CPU ------
def test_fixture_is_reproducible(g6):
    """The committed annotations are what the seeded generator produces (the fixture carries them anyway)."""
    z, gts, dts = g6
    g2, d2 = syn.make_kitti_annos(n_img=len(gts), seed=7)
    for a, b in zip(gts + dts, g2 + d2):
        assert list(a['name']) == list(b['name'])
        np.testing.assert_array_equal(a['bbox'], b['bbox'])


def test_oracle_overlaps_match_reference(g6):
    z, gts, dts = g6
    for metric, tol in ((0, 0.0), (1, 2e-4), (2, 2e-4)):
        ov = np.concatenate([ke.overlaps_one_image(dts[i], gts[i], metric).reshape(-1) for i in range(len(gts))])
        ref = z[f'm{metric}_overlaps']
        assert ov.shape == ref.shape
        # metric 0 is bit-exact; the rotated ones differ by the float32 noise of the reference's own clipping
        assert np.abs(ov - ref).max() <= tol, (metric, np.abs(ov - ref).max())
        assert ((ov > 0) == (ref > 0)).mean() > 0.995


def test_oracle_curves_match_reference(g6):
    z, gts, dts = g6
    mo = ke.KITTI_MIN_OVERLAPS[:, :, [0, 1, 2]]
    for metric in (0, 1, 2):
        r = ke.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        for k in ('precision', 'recall', 'orientation'):
            np.testing.assert_allclose(r[k], z[f'm{metric}_{k}'], rtol=0, atol=1e-12, equal_nan=True, err_msg=f'{metric} {k}')


def test_oracle_ap_matches_reference_dict(g6):
    z, gts, dts = g6
    for crit in ('R40', 'R11'):
        ap = ke.kitti_ap(gts, dts, CLASSES, crit)
        ref = dict(zip(z['dict_keys_' + crit].tolist(), z['dict_vals_' + crit].tolist()))
        for j, cname in enumerate(CLASSES):
            for di, dname in enumerate(('easy', 'moderate', 'hard')):
                for si, sname in enumerate(('strict', 'loose')):
                    for tag, key in (('3D', '3d'), ('BEV', 'bev'), ('2D', 'bbox')):
                        assert abs(ap[key][j, di, si] - ref[f'KITTI/{cname}_{tag}_{dname}_{sname}']) < 1e-9


def test_oracle_thresholds_properties():
    rng = np.random.default_rng(0)
    sc = rng.uniform(0, 1, 500)
    th = ke.sample_thresholds(sc, 600)
    assert len(th) <= 41 and np.all(np.diff(th) <= 0) and th[0] == sc.max()
    assert len(ke.sample_thresholds(np.zeros(0), 10)) == 0
    # every label detected, 40 labels: one threshold per label + the recall-0 point is skipped by construction
    assert len(ke.sample_thresholds(np.linspace(1, 0.1, 40), 40)) == 40


def test_perfect_detections_have_unit_precision():
    """Detections identical to the labels: every sampled precision is 1 and the last sampled recall is 1 (with few
    labels fewer than 41 recall points exist, so the AP itself is n/40 — KITTI's sampling, not an error)."""
    gts, _ = syn.make_kitti_annos(n_img=12, seed=3)
    dts = []
    for g in gts:
        keep = np.array([n in ('Car', 'Pedestrian', 'Cyclist') for n in g['name']], bool)
        d = {k: (v[keep].copy() if isinstance(v, np.ndarray) else v) for k, v in g.items()}
        d['score'] = np.linspace(0.9, 0.5, keep.sum()).astype(np.float32)
        dts.append(d)
    mo = ke.KITTI_MIN_OVERLAPS[:, :, [0, 1, 2]]
    for metric in (0, 1, 2):
        r = ke.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        p = r['precision']
        assert np.all((p == 0) | (p == 1) | np.isnan(p)) and (p == 1).sum() > 20
        assert np.nanmax(r['recall']) == 1.0
        if metric == 0:
            o = r['orientation']
            assert np.all((o == 0) | (np.abs(o - 1) < 1e-12) | np.isnan(o))


def test_result_formatting_roundtrip(tmp_path):
    """N3: format_results / format_gt_anno / write_result_files (kitti3d_dataset.py:230-325) — pure host code."""
    from monorun_amd import evaluation as ev
    rng = np.random.default_rng(1)
    calib = np.array([[707.0, 0, 600.0, 45.0], [0, 707.0, 180.0, -0.3], [0, 0, 1, 0.005]])
    K, t = ev.cam_t_vec_from_calib(calib)
    np.testing.assert_allclose(K @ t, calib[:, 3], atol=1e-12)
    per_class2 = [rng.uniform(0, 300, (n, 5)).astype(np.float32) for n in (2, 0, 1)]
    per_class3 = [np.concatenate([rng.uniform(1, 4, (n, 3)), rng.uniform(-5, 40, (n, 3)), rng.uniform(-3, 3, (n, 1)), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32) for n in (2, 0, 1)]
    res = ev.format_results([dict(bbox_results=per_class2, bbox_3d_results=per_class3)], [dict(cam_t_vec=t.astype(np.float32))], CLASSES)[0]
    assert list(res['name'][np.argsort(-res['score'], kind='stable')]) == list(res['name'])       # descending score
    assert np.all(np.diff(res['score']) <= 0) and sorted(res['name']) == ['Car', 'Car', 'Cyclist']
    all3 = np.concatenate(per_class3, 0)
    i = int(np.argmax(all3[:, 7]))
    np.testing.assert_allclose(res['location'][0], all3[i, 3:6] - t.astype(np.float32), rtol=1e-6)
    np.testing.assert_allclose(res['alpha'][0], res['rotation_y'][0] - np.arctan2(res['location'][0, 0], res['location'][0, 2] + 0.27), rtol=1e-6)
    assert per_class3[0][0, 3] == all3[0, 3]                                                      # inputs not modified in place
    gt = ev.format_gt_anno(dict(bboxes=np.zeros((2, 4), np.float32), bboxes_ignore=np.ones((1, 4), np.float32), labels=[0, 2],
                                truncation=[0.0, 0.2], occlusion=[0, 1], alpha=[0.1, -0.2],
                                bboxes_3d_eval=np.arange(14, dtype=np.float32).reshape(2, 7)), CLASSES)
    assert gt['name'] == ['Car', 'Cyclist', 'DontCare'] and gt['alpha'][-1] == -10 and gt['location'][-1, 0] == -1000
    assert gt['bbox'].shape == (3, 4) and gt['index'].tolist() == [0, 1, -1]
    out = tmp_path / 'data'
    ev.write_result_files([res], ['000007.png'], str(out))
    rows = (out / '000007.txt').read_text().strip().split('\n')
    assert len(rows) == 3 and len(rows[0].split(' ')) == 16 and rows[0].split(' ')[0] == res['name'][0]
    np.testing.assert_allclose([float(x) for x in rows[0].split(' ')[8:11]], res['dimensions'][0][[1, 2, 0]], rtol=1e-6)   # h w l


# ------------------------------------------------------------------------------------- GPU ------
@pytest.mark.gpu
def test_gpu_overlaps_vs_oracle_and_reference(g6):
    from monorun_amd import evaluation as ev
    z, gts, dts = g6
    for metric in (0, 1, 2):
        blocks, _, n_rows, n_cols = ev.calculate_iou_partly(dts, gts, metric)
        assert [b.shape for b in blocks] == [(len(d['name']), len(g['name'])) for d, g in zip(dts, gts)]
        got = np.concatenate([b.reshape(-1) for b in blocks])
        orc = np.concatenate([ke.overlaps_one_image(dts[i], gts[i], metric).reshape(-1) for i in range(len(gts))])
        ref = z[f'm{metric}_overlaps']
        if metric == 0:
            np.testing.assert_array_equal(got, ref)                    # bit-exact (float32 arithmetic, same operation order)
        else:
            assert np.abs(got - orc).max() <= 1e-6                     # fp64 clipping on both sides, rounded to float32
            assert np.abs(got - ref).max() <= 2e-4                     # the reference's own float32 clipping noise


@pytest.mark.gpu
def test_gpu_curves_and_text_match_reference(g6):
    from monorun_amd import evaluation as ev
    z, gts, dts = g6
    mo = ke.KITTI_MIN_OVERLAPS[:, :, [0, 1, 2]]
    for metric in (0, 1, 2):
        r = ev.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        for k in ('precision', 'recall', 'orientation'):
            np.testing.assert_allclose(r[k], z[f'm{metric}_{k}'], rtol=0, atol=1e-12, equal_nan=True, err_msg=f'{metric} {k}')
    for crit in ('R40', 'R11'):
        types = ['bbox', 'bev', '3d']
        text, d = ev.kitti_eval(gts, dts, CLASSES, eval_types=types, criteria=crit)
        assert types == ['bbox', 'bev', '3d']                          # the caller's list is not mutated
        assert text == str(z['text_' + crit])
        ref = dict(zip(z['dict_keys_' + crit].tolist(), z['dict_vals_' + crit].tolist()))
        assert sorted(d) == sorted(ref)
        for k in ref:
            assert abs(d[k] - ref[k]) < 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize('seed,dtype,n_img', [(11, np.float32, 40), (12, np.float64, 25), (13, np.float32, 3)])
def test_gpu_eval_vs_oracle_seeded(seed, dtype, n_img):
    from monorun_amd import evaluation as ev
    gts, dts = syn.make_kitti_annos(n_img=n_img, seed=seed, dtype=dtype)
    mo = ke.KITTI_MIN_OVERLAPS[:, :, [0, 1, 2]]
    for metric in (0, 1, 2):
        a = ev.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        b = ke.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], metric, mo, compute_aos=(metric == 0))
        for k in ('precision', 'recall', 'orientation'):
            np.testing.assert_allclose(a[k], b[k], rtol=0, atol=1e-12, equal_nan=True, err_msg=f'{metric} {k}')


@pytest.mark.gpu
def test_gpu_eval_edge_cases():
    from monorun_amd import evaluation as ev
    gts, dts = syn.make_kitti_annos(n_img=10, seed=21)
    empty = {k: v[:0].copy() for k, v in dts[0].items()}
    # no detections at all: every precision is 0/0 -> NaN in the reference (eval.py:541-543), AP NaN; nothing crashes
    text, d = ev.kitti_eval(gts, [dict(empty) for _ in gts], ['Car'], eval_types=['bbox', 'bev', '3d'], criteria='R40')
    assert 'Car AP@0.70' in text and all(np.isnan(v) or v == 0 for v in d.values())
    # images without labels / without detections mixed in
    dts2 = [dict(empty) if i % 3 == 0 else a for i, a in enumerate(dts)]
    gts2 = [{k: v[:0].copy() for k, v in g.items()} if i % 4 == 1 else g for i, g in enumerate(gts)]
    mo = ke.KITTI_MIN_OVERLAPS[:, :, [0, 1, 2]]
    a = ev.eval_class(gts2, dts2, [0, 1, 2], [0, 1, 2], 2, mo)
    b = ke.eval_class(gts2, dts2, [0, 1, 2], [0, 1, 2], 2, mo)
    np.testing.assert_allclose(a['precision'], b['precision'], rtol=0, atol=1e-12, equal_nan=True)
    # single class given as a bare string, coco-style summary
    t1, d1 = ev.kitti_eval(gts, dts, 'Car', eval_types=['bbox', 'bev', '3d'], criteria='R40')
    assert 'Overall' not in t1 and 'KITTI/Car_3D_moderate_strict' in d1
    t2, d2 = ev.kitti_eval(gts, dts, CLASSES, eval_types=['bbox'], criteria='R11')            # 2-D only: no bev / 3d lines or keys
    assert 'bev ' not in t2 and '3d  ' not in t2 and 'aos' in t2 and not any('_3D_' in k or '_BEV_' in k for k in d2)
    assert abs(d2['KITTI/Car_2D_moderate_strict'] - ke.get_map(ke.eval_class(gts, dts, [0, 1, 2], [0, 1, 2], 0, ke.KITTI_MIN_OVERLAPS[:, :, [0, 1, 2]])['precision'], 'R11')[0, 1, 0]) < 1e-9
    coco = ev.kitti_eval_coco_style(gts, dts, CLASSES, criteria='R40')
    assert coco.count('coco AP@') == 3 and 'Car coco AP@0.50:0.05:0.95:' in coco
    mo10 = np.zeros((10, 3, 1)); mo10[:, :, 0] = np.linspace(0.5, 0.95, 10)[:, None]
    want = ke.get_map(ke.eval_class(gts, dts, [0], [0, 1, 2], 2, mo10)['precision'], 'R40').mean(-1)
    line = [l for l in coco.split('\n') if l.startswith('3d ')][0]
    assert line == '3d   AP:{:.2f}, {:.2f}, {:.2f}'.format(*want[0])


def test_kitti_file_io_roundtrip(tmp_path):
    """open_label_file / open_calib_file / parse_ann_info (kitti3d_dataset.py:40-56,116-178) on files written in the KITTI
    format, and result files read back as labels."""
    from monorun_amd import evaluation as ev
    P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]])
    (tmp_path / 'c.txt').write_text(''.join(f'P{i}: ' + ' '.join(f'{v:.12e}' for v in (P2 + i).reshape(-1)) + '\n' for i in range(4)))
    calib = ev.open_calib_file(str(tmp_path / 'c.txt'), 2)
    np.testing.assert_allclose(calib, P2 + 2, rtol=1e-6)
    (tmp_path / 'l.txt').write_text(
        'Car 0.00 0 -1.58 587.01 173.33 614.12 200.12 1.65 1.67 3.64 -0.65 1.71 46.70 -1.59\n'
        'Van 0.10 1 1.00 10.0 20.0 30.0 40.0 2.0 1.9 5.0 1.0 1.5 20.0 0.5\n'
        'DontCare -1 -1 -10 503.89 169.71 590.61 190.13 -1 -1 -1 -1000 -1000 -1000 -10\n'
        'Cyclist 0.25 2 0.30 100.0 120.0 150.0 200.0 1.70 0.60 1.80 -5.0 1.6 12.0 0.1\n')
    label = ev.open_label_file(str(tmp_path / 'l.txt'))
    assert len(label) == 4 and label[0][0] == 'Car' and isinstance(label[0][2], int) and label[3][2] == 2
    ann = ev.parse_ann_info(label, calib)
    assert ann['labels'].tolist() == [0, 2] and ann['object_ids'].tolist() == [0, 3]        # the Van is dropped, DontCare kept apart
    assert ann['bboxes_ignore'].shape == (1, 4) and ann['bboxes'].shape == (2, 4)
    np.testing.assert_allclose(ann['bboxes_3d_eval'][0], [3.64, 1.65, 1.67, -0.65, 1.71, 46.70, -1.59], rtol=1e-6)   # lhw xyz ry
    np.testing.assert_allclose(ann['bboxes_3d'][0, 3:6] - ann['bboxes_3d_eval'][0, 3:6], ann['cam_t_vec'], rtol=1e-5)
    np.testing.assert_allclose(ann['cam_intrinsic'] @ ann['cam_t_vec'], calib[:, 3], rtol=1e-4, atol=1e-4)
    assert set(ev.parse_ann_info(None, calib)) == {'cam_intrinsic', 'cam_t_vec'}               # test mode
    gt = ev.format_gt_anno(ann, ('Car', 'Pedestrian', 'Cyclist'))
    assert gt['name'] == ['Car', 'Cyclist', 'DontCare'] and gt['occluded'].tolist() == [0, 2, -1]
    # result files are label files with a score column
    det = dict(name=np.array(['Car']), truncated=np.array([-1], np.int8), occluded=np.array([-1], np.int8), alpha=np.array([0.5], np.float32),
               bbox=np.array([[1, 2, 3, 4]], np.float32), dimensions=np.array([[3.9, 1.5, 1.6]], np.float32),
               location=np.array([[1.0, 1.6, 20.0]], np.float32), rotation_y=np.array([0.2], np.float32), score=np.array([0.9], np.float32))
    ev.write_result_files([det], ['000001.png'], str(tmp_path / 'data'))
    back = ev.open_label_file(str(tmp_path / 'data' / '000001.txt'))
    np.testing.assert_allclose(back[0][8:11], [1.5, 1.6, 3.9], rtol=1e-6)                    # h w l
    assert back[0][0] == 'Car' and abs(back[0][15] - 0.9) < 1e-6


@pytest.mark.gpu
def test_gpu_end_to_end_harness_synthetic():
    """tools/kitti_val.py --synthetic: label/calib files + synthetic raw head outputs -> fused decode+PnP -> 3-D boxes ->
    KITTI result files -> evaluator.  Small 3-D noise, so the recovered boxes must score a high AP."""
    import subprocess, sys, re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'kitti_val.py'), '--synthetic', '40'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout
    block = text[text.index('Car AP@0.70, 0.50, 0.50'):]
    ap3d = [float(v) for v in re.search(r'3d   AP:([\d.]+), ([\d.]+), ([\d.]+)', block).groups()]
    bev = [float(v) for v in re.search(r'bev  AP:([\d.]+), ([\d.]+), ([\d.]+)', block).groups()]
    assert min(ap3d[1:]) > 80 and min(bev[1:]) > 80, text
    assert '40 images, 240 objects' in text
    # the reference's initialiser restated on the GPU through the same harness (K2 -> EPnP / RANSAC -> LM per batch of images)
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'kitti_val.py'), '--synthetic', '40', '--initialiser', 'epnp'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    block = out.stdout[out.stdout.index('Car AP@0.70, 0.50, 0.50'):]
    ap3d = [float(v) for v in re.search(r'3d   AP:([\d.]+), ([\d.]+), ([\d.]+)', block).groups()]
    assert min(ap3d[1:]) > 80 and "initialiser 'epnp'" in out.stdout
