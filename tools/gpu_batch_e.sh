#!/bin/bash
timeout 300 python tests/helpers/rccl_world1.py 2>&1 | grep -v amdgpu.ids | tail -15
timeout 600 python -m pytest tests/test_gpu_pca.py -x -q -m gpu -k "rccl" 2>&1 | tail -5
