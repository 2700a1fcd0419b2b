python -m pytest tests/test_model_gpu.py -m gpu -x -q -k full_size -s 2>&1 | grep -v amdgpu.ids | tail -30
