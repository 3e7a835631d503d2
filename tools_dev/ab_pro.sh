set -x
python -m pytest tests/test_gpu_groups.py -x -q -m gpu 2>&1 | tail -15
python tools_dev/ab.py "ops.use_prologue(False)" "ops.use_prologue(True)" 2>&1 | tail -4
