python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | cut -c1-600
