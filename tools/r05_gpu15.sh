python -m pytest tests -x -q -m gpu 2>&1 | tail -4
bash tools/profile_r05.sh r05
