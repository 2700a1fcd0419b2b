python -m pytest tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -15
