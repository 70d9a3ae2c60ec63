python -m pytest tests -x -q -m gpu 2>&1 | tail -15
