"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/monorun_pnp.h declares; the Python surface mirrors monorun/ops (names, constructor, registry,
empty batch, loud failure without a GPU).  No compute calls — there is no GPU here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, 'include', 'monorun_pnp.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(?:int|void|const char \*)\s*\*?\s*([a-z_0-9]+)\s*\(', src)))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for n in ('mr_pnp_uncert_batched', 'pnp_uncert', 'mr_noc_decode_batched', 'mr_pnp_version',
              'mr_pnp_error_string', 'mr_pnp_last_hip_error', 'mr_pnp_device_count'):
        assert n in names, names


def test_library_loads_and_exports_every_declared_symbol():
    from monorun_amd import _lib
    assert os.path.exists(_lib.SO), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.SO)
    for n in _declared_functions():
        assert hasattr(lib, n), f'{n} declared in include/monorun_pnp.h but not exported'
    h = _lib.load()
    assert h.mr_pnp_version() == 100
    assert h.mr_pnp_error_string(0) == b'ok' and b'argument' in h.mr_pnp_error_string(-1)
    for n in _lib.EXPORTED_SYMBOLS:
        assert hasattr(h, n)


def test_reference_signature_of_legacy_symbol():
    """ext.h:1-13 — 5 double* in, int*, 3 double*, int, double*."""
    from monorun_amd import _lib
    at = _lib.load().pnp_uncert.argtypes
    dp = ctypes.POINTER(ctypes.c_double)
    assert list(at) == [dp, dp, dp, dp, dp, ctypes.POINTER(ctypes.c_int), dp, dp, dp, ctypes.c_int, dp]


def test_ops_surface_mirrors_reference():
    import monorun_amd.ops as ops
    assert set(ops.__all__) >= {'u2d_pnp_cpu', 'build_pnp', 'PnPUncert', 'pnp_uncert'}
    from monorun_amd.ops import build_pnp, PnPUncert, PNP
    m = build_pnp(dict(type='PnPUncert', z_min=0.5, epnp_istd_thres=0.6, inlier_opt_only=True, forward_exact_hessian=False))
    assert isinstance(m, PnPUncert) and isinstance(m, torch.nn.Module)
    assert (m.z_min, m.epnp_istd_thres, m.inlier_opt_only, m.coord_istd_normalize, m.forward_exact_hessian, m.use_6dof, m.eps) == \
           (0.5, 0.6, True, False, False, False, 1e-6)
    assert len(list(m.state_dict())) == 0                       # no parameters or buffers, like the reference
    assert m.initialiser == 'epnp'                              # the reference's config dict builds the REFERENCE's flow
    d = PnPUncert()
    assert (d.z_min, d.epnp_istd_thres, d.inlier_opt_only, d.initialiser) == (0.5, 0.6, True, 'epnp')
    assert build_pnp(dict(type='PnPUncert', initialiser='k0')).initialiser == 'k0'       # the explicit fast mode
    import inspect
    from monorun_amd.ops import pnp_uncert, u2d_pnp_cpu
    from monorun_amd.ops.least_squares.pnp_uncert import DEFAULT_INITIALISER
    assert DEFAULT_INITIALISER == 'epnp'
    assert inspect.signature(pnp_uncert).parameters['initialiser'].default is None and inspect.signature(u2d_pnp_cpu).parameters['initialiser'].default is None
    with pytest.raises(TypeError):                               # the reference's latent default-dict bug is preserved:
        build_pnp(dict(type='PnPUncert', backward_exact_hessian=True))   # pnp_uncert.py:93-99 does not accept it
    with pytest.raises(KeyError):
        build_pnp(dict(type='NoSuchPnP'))


def test_empty_batch_returns_typed_empties_without_a_gpu():
    from monorun_amd.ops import u2d_pnp_cpu
    e = u2d_pnp_cpu(np.zeros((0, 784, 2), np.float32), np.zeros((0, 784, 2), np.float32), np.zeros((0, 784, 3), np.float32),
                    np.eye(3, dtype=np.float32)[None], np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32))
    assert [a.shape for a in e] == [(0,), (0, 1), (0, 3), (0, 4, 4), (0, 1), (0, 784)]
    assert e[0].dtype == bool and e[5].dtype == bool and e[1].dtype == np.float32


@pytest.mark.skipif(torch.cuda.is_available(), reason='only meaningful without a GPU')
def test_no_cpu_fallback_fails_loudly():
    from monorun_amd.ops import pnp_uncert, u2d_pnp_cpu
    x = torch.zeros(2, 16, 2)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        pnp_uncert(x, x + 1, torch.zeros(2, 16, 3), torch.eye(3)[None], torch.zeros(1, 2), torch.zeros(1, 2))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        u2d_pnp_cpu(np.zeros((2, 16, 2), np.float32), np.ones((2, 16, 2), np.float32), np.zeros((2, 16, 3), np.float32),
                    np.eye(3, dtype=np.float32)[None], np.zeros((1, 2), np.float32), np.zeros((1, 2), np.float32))


def test_product_never_imports_the_oracle():
    """③: only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, 'monorun_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dp, f)).read()
                assert 'oracle' not in txt.replace('the CPU oracle', '').replace('against the CPU oracle', ''), os.path.join(dp, f)
    # tools/ holds product-side utilities (profiling, timing, the KITTI harness): none may import the checker either; the
    # sweeps that DO compare against it live under tests/sweeps/
    import re
    for f in os.listdir(os.path.join(ROOT, 'tools')):
        if f.endswith(('.py', '.sh')):
            txt = open(os.path.join(ROOT, 'tools', f)).read()
            assert not re.search(r'^\s*(from|import)\s+oracle', txt, re.M), f


def test_synthetic_generator_shapes_and_determinism():
    from monorun_amd import synthetic as syn
    a, b = syn.make_batch(B=16, seed=1234), syn.make_batch(B=16, seed=1234)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert a['coords_2d'].shape == (16, 2, 28, 28) and a['coords_3d'].shape == (16, 3, 28, 28) and a['logstd'].shape == (16, 2, 28, 28)
    x2d, istd, x3d, K, ur, vr, thr = syn.pnp_boundary(a, planar=True)
    assert x2d.strides == (1568 * 4, 4, 784 * 4) and x3d.strides == (2352 * 4, 4, 784 * 4)
    assert np.array_equal(ur, [[-200, 1442]]) and np.array_equal(vr, [[-200, 575]])
    np.testing.assert_allclose(thr, 0.2 * 27 / 28 * (a['rois'][:, 3] - a['rois'][:, 1]), rtol=1e-4)


def test_capi_argument_validation_without_a_gpu():
    """Bad arguments are rejected before any HIP call, so this runs on CPU: error codes of include/monorun_pnp.h."""
    from monorun_amd import _lib
    lib = _lib.load()
    st = (ctypes.c_int64 * 3)(1568, 1, 784)
    buf = (ctypes.c_float * 64)()
    p = ctypes.addressof(buf)

    def call(B=4, P=784, x2d=p, cam_batch=1, range_batch=1, cov=p, dtype=0, flags=0):
        return lib.mr_pnp_uncert_batched(x2d, st, p, st, p, st, dtype, p, cam_batch, p, p, range_batch, None, None, B, P,
                                         0.5, 0.6, 1, flags, p, p, cov, p, p, None, None)
    assert call(B=0) == 0                                     # empty batch: success, nothing launched
    assert call(B=-1) == -1 and call(P=3) == -1 and call(P=10 ** 6) == -1
    assert call(x2d=None) == -1 and call(cov=None) == -1
    assert call(cov=None, flags=_lib.MR_COV_NONE, B=0) == 0
    assert call(cam_batch=3) == -1 and call(range_batch=2) == -1
    assert call(P=8192) == -2                                 # tile does not fit the 160 KB LDS
    assert call(dtype=7) == -2
    assert lib.mr_pnp_error_string(-2).startswith(b'unsupported') and lib.mr_pnp_error_string(-99) == b'unknown error'
    # the reference's initialiser: workspace sizing is host arithmetic, bad arguments are refused before any HIP call
    assert lib.mr_epnp_workspace_bytes(0, 784) == 0 and lib.mr_epnp_workspace_bytes(8, 3) == 0
    w1, w2 = lib.mr_epnp_workspace_bytes(1024, 784), lib.mr_epnp_workspace_bytes(2048, 784)
    assert 10e6 < w1 < 20e6 and w1 % 256 == 0 and 1.9 * w1 < w2 < 2.1 * w1

    def ecall(B=4, P=784, x2d=p, cam_batch=1, max_iters=30, out=p, work=None, nbytes=0):
        return lib.mr_epnp_ransac_batched(x2d, st, p, st, p, st, 0, p, cam_batch, p, B, P, 0.6, 0, max_iters, out, p, p, None, None, work, nbytes, None)
    assert ecall(B=0) == 0
    assert ecall(B=-1) == -1 and ecall(P=3) == -1 and ecall(max_iters=0) == -1 and ecall(max_iters=31) == -1
    assert ecall(x2d=None) == -1 and ecall(out=None) == -1 and ecall(cam_batch=2) == -1
    # NMS / decode entry points validate too
    assert lib.mr_nms_bev_batched(None, None, None, 0, 0, 0.1, None, None, None) == 0
    assert lib.mr_nms_bev_batched(p, p, p, 1, 4096, 0.1, p, p, None) == -2
    assert lib.mr_nms_bev_batched(p, p, None, 1, 8, 0.1, p, p, None) == -1


def test_img_shape_and_flip_forms_the_pipeline_passes():
    """mmdet hands img_meta['img_shape'] = (H, W, 3) and a per-image bool flip (Python or numpy bool, or a 0-dim tensor):
    the host glue takes the first two entries per image like the reference (uncert_prop_pnp_optimizer.py:75-80) and turns
    every scalar flip form into B flags (one flag per object otherwise; a wrong count is refused, not read out of bounds)."""
    from monorun_amd import pose_head as ph
    cpu = torch.device('cpu')
    for shp in ((375, 1242), (375, 1242, 3), [[375, 1242, 3]], np.array([[375., 1242.]]), torch.tensor([[375., 1242., 3.]])):
        ur, vr = ph._clip_ranges(shp, 200, cpu)
        assert ur.tolist() == [[-200.0, 1442.0]] and vr.tolist() == [[-200.0, 575.0]]
    ur, vr = ph._clip_ranges(((375, 1242, 3), (370, 1224, 3)), 200, cpu)
    assert ur.tolist() == [[-200.0, 1442.0], [-200.0, 1424.0]] and vr.tolist() == [[-200.0, 575.0], [-200.0, 570.0]]
    with pytest.raises(AssertionError):
        ph._clip_ranges((375, 1242, 3, 1), 200, cpu)
    for f in (True, np.bool_(True), torch.tensor(True), np.array(True), np.array([True])):
        fl = ph._flip_flags(f, 5, cpu)
        assert fl.dtype == torch.uint8 and fl.tolist() == [1] * 5
    assert ph._flip_flags([True, False, True], 3, cpu).tolist() == [1, 0, 1]
    assert ph._flip_flags(torch.tensor([False]), 1, cpu).tolist() == [0]
    with pytest.raises(AssertionError):
        ph._flip_flags([True, False], 3, cpu)


def test_pose_stage_dump_hook_on_a_stand_in_roi_head(tmp_path):
    """monorun_amd.integration.PoseStageDump wraps the four call sites of MonoRUnRoIHead.simple_test
    (monorun_roi_head.py:442-547) and writes what tools/kitti_val.py reads.  mmdet is absent, so the hook is exercised on a
    stand-in object with the same attribute structure and call order; the wrapped functions must behave as before."""
    import types
    from monorun_amd.integration import PoseStageDump
    n, C = 5, 3
    conv = torch.nn.Conv2d(4, 2 * C * 5, 1)

    class Head:
        def __init__(self):
            self.noc_head = types.SimpleNamespace(conv_final=conv)
            self.bbox_head = types.SimpleNamespace(get_bboxes=lambda *a, **k: (torch.cat([torch.rand(n, 4) * 100, torch.rand(n, 1)], 1), torch.arange(n) % C))
            self.score_head = types.SimpleNamespace(pre_sigmoid=True)
            self.test_cfg = types.SimpleNamespace(mult_2d_score=True)

        def _reg_forward(self, x, rois, labels):
            return dict(dim_pred=torch.randn(len(rois), 3), dim_var=torch.rand(len(rois), 3))

        def _score_forward(self, *a):
            return dict(scores=torch.randn(n, 1))

        def simple_test(self, x, proposal_list, img_metas, proposals=None, coord_2d=None, cam_intrinsic=None, rescale=False):
            det_bboxes, det_labels = self.bbox_head.get_bboxes()
            rois = torch.cat([torch.zeros(n, 1), det_bboxes[:, :4]], 1)
            self._reg_forward(x, rois, det_labels)
            self.noc_head.conv_final(torch.randn(n, 4, 28, 28))
            self._score_forward()
            return 'results'
    h = Head()
    orig = h.simple_test
    meta = dict(filename='/data/kitti/training/image_2/000123.png', img_shape=(375, 1242, 3), flip=False)
    with PoseStageDump(h, str(tmp_path)) as d:
        assert h.simple_test(None, None, [meta], cam_intrinsic=[[torch.eye(3)]]) == 'results'
    assert h.simple_test.__func__ is orig.__func__ and d.written == [str(tmp_path / '000123.npz')]       # restored on exit
    z = np.load(d.written[0])
    assert z['all_pred'].shape == (n, 30, 28, 28) and z['labels'].dtype == np.int64 and z['rois'].shape == (n, 4)
    assert z['dim'].shape == (n, 3) and z['dim_var'].shape == (n, 3) and z['bboxes'].shape == (n, 4) and z['scores'].shape == (n,)
    assert z['scores_ref'].shape == (n,) and np.all((z['scores_ref'] >= 0) & (z['scores_ref'] <= 1)) and z['img_shape'].tolist() == [375.0, 1242.0]
    assert np.array_equal(z['rois'], z['bboxes']) and z['cam_intrinsic'].shape == (3, 3) and not bool(z['flip'])


def test_integration_md_cffi_block_is_the_generated_prototype_text_and_every_symbol_is_exported():
    """INTEGRATION.md §3 shows the text a maintainer feeds to cffi's `ffi.cdef`.  cffi is absent here, so instead of executing it
    the test checks (a) the block is exactly what tools/gen_cffi_cdef.py derives from include/monorun_pnp.h, (b) it contains only
    what cffi's declaration parser accepts (prototypes over plain C / stdint types and plain-integer #defines: no comments, no
    `extern "C"`, no parenthesised or shifted macro values), (c) every prototype names a symbol the built library exports, with
    the argument count the header declares."""
    import ctypes
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import gen_cffi_cdef as gen
    md = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    block = md.split('<!-- cffi-cdef:begin -->')[1].split('<!-- cffi-cdef:end -->')[0]
    text = block.split('```c\n')[1].split('```')[0]
    assert text == gen.cdef_text(), 'INTEGRATION.md cffi block is stale: regenerate with tools/gen_cffi_cdef.py'
    allowed = re.compile(r'^(#define MR_\w+ (0x[0-9A-Fa-f]+|\d+)|[A-Za-z_][\w \*]*\([\w \*,]*\);)$')
    protos = []
    for line in text.strip().split('\n'):
        assert allowed.match(line), line
        assert '/*' not in line and '//' not in line and 'extern' not in line and '<<' not in line and '(-' not in line
        if not line.startswith('#define'):
            m = re.match(r'^(.*?)(\w+)\((.*)\);$', line)
            name, args = m.group(2), m.group(3)
            for tok in re.findall(r'[A-Za-z_]\w*', m.group(1) + ' ' + re.sub(r'\b\w+(?=\s*(,|$))', '', args)):
                assert tok in ('const', 'void', 'int', 'float', 'double', 'char', 'uint8_t', 'int8_t', 'int32_t', 'int64_t', 'size_t'), (name, tok)
            protos.append((name, 0 if args.strip() == 'void' else args.count(',') + 1))
    from monorun_amd import _lib
    lib = ctypes.CDLL(_lib.SO)
    hdr = {n: len(a) for n, _, a in gen.header_prototypes()}
    assert set(n for n, _ in protos) == set(hdr) == set(_lib.EXPORTED_SYMBOLS)
    for name, nargs in protos:
        assert hasattr(lib, name) and hdr[name] == nargs, name
