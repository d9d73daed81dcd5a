#!/bin/bash
O=gpurun_out/r5e; mkdir -p $O
timeout 300 python tools/r5_lat.py 2>&1 | grep -v amdgpu.ids > $O/lat.txt
timeout 300 python tools/r5_lat.py pipe 2>&1 | grep -v amdgpu.ids >> $O/lat.txt
timeout 900 python -X faulthandler -m pytest tests/test_gpu_pca.py -q -x -s -k "float64" 2>&1 | grep -v amdgpu.ids | head -80 > $O/pytest_f64.txt
timeout 900 python -X faulthandler -m pytest tests/test_gpu_pca.py -q -x -s -k "6144" 2>&1 | grep -v amdgpu.ids | head -80 > $O/pytest_6144.txt
cat $O/lat.txt $O/pytest_f64.txt $O/pytest_6144.txt
