#!/bin/bash
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python tools/time_configs.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/time_topk.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/prof_stage.py pca 400 512 3 2>&1 | grep -v amdgpu.ids | tail -1
timeout 300 python tools/time_c4.py 2>&1 | grep -v amdgpu | tail -2
timeout 300 python tools/time_msdi.py 2>&1 | grep -v amdgpu | tail -3
