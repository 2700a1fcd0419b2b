python -m pytest tests/test_train_gpu.py -m gpu -x -q -s 2>&1 | tail -12
