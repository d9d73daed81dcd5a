#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh or topk or tri or eig" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_pca.py tests/test_gpu_fullsize.py -x -q -m gpu -k "annular or annulus or 4d or c4 or c3" 2>&1 | tail -3
timeout 300 python tools/time_c4.py 2>&1 | grep -v amdgpu | tail -3
timeout 300 python tools/time_c3.py 2>&1 | grep -v amdgpu | tail -1
