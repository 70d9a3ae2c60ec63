python -m pytest tests/test_distributed_gloo.py tests/test_gpu_parity.py::test_bench_line_describes_the_regime_it_measured tests/test_kitti_eval.py -x -q -m gpu 2>&1 | tail -15
