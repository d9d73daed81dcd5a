#!/bin/bash
timeout 300 python tools/time_rot.py 512 400 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/time_rot.py 256 2000 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/time_rot.py 1024 100 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rot or shear" 2>&1 | tail -2
