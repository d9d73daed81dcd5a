#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh or topk or tri or eig" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_pca.py -x -q -m gpu -k "more_than or svd" 2>&1 | tail -5
timeout 300 python tools/time_topk.py 2>&1 | grep -v amdgpu
