timeout -k 10 1100 python -m pytest tests -q -m gpu 2>&1 | tail -40
