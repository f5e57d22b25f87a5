python3 scripts/quick.py 1000 1024 2>&1 | grep -v amdgpu
timeout -k 10 1000 python -m pytest tests -q -m gpu 2>&1 | tail -4
