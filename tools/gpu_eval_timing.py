"""Wall time of the KITTI evaluator at validation-split size (3 769 images) on synthetic annotations (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from monorun_amd import synthetic as syn, evaluation as ev
n = int(os.environ.get('QN', 3769))
t = time.time(); gts, dts = syn.make_kitti_annos(n_img=n, seed=99); print(f'{n} images, {sum(len(a["name"]) for a in gts)} labels, {sum(len(a["name"]) for a in dts)} detections (generated in {time.time()-t:.1f} s)')
for it in range(2):
    torch.cuda.synchronize(); t = time.time()
    text, d = ev.kitti_eval(gts, dts, ['Car', 'Pedestrian', 'Cyclist'], eval_types=['bbox', 'bev', '3d'], criteria='R40')
    torch.cuda.synchronize(); print(f'kitti_eval (3 classes x 3 difficulties x 2 overlaps x bbox/bev/3d + aos): {time.time()-t:.2f} s')
print(text)
