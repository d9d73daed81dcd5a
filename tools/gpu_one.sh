#!/bin/bash
timeout 300 python tools/time_rot.py 1024 200 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/time_rot.py 1024 200 rot_4096_w1=0 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests -x -q -m gpu -k "1024 or c5 or quadrant or rot" 2>&1 | tail -3
timeout 600 python tools/run_c5.py 2>&1 | grep -v amdgpu.ids | tail -5
