#!/bin/bash
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh or topk or tri or eig" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c5" 2>&1 | tail -3
timeout 600 python tools/run_c5.py 2>&1 | grep -v amdgpu.ids | tail -5
