python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python __graft_entry__.py smoke 2>&1 | tail -2
bash tools/r05_gpu6.sh
