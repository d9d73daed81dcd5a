#!/bin/bash
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "eigh" 2>&1 | tail -15 > $O/pytest_eigh.txt
timeout 600 python -m pytest tests/test_gpu_procs.py -q -x -s 2>&1 | tail -15 > $O/pytest_procs.txt
timeout 120 python tools/time_configs.py 2>&1 | grep -v amdgpu.ids > $O/configs.txt
VIPMI_OPTS=rot_1024_q=1 timeout 120 python tools/time_configs.py 2>&1 | grep -v amdgpu.ids > $O/configs_q.txt
cat $O/pytest_eigh.txt $O/pytest_procs.txt $O/configs.txt $O/configs_q.txt
