#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
timeout 300 python tools/r5_f64dbg.py 2>&1 | grep -v amdgpu.ids > $O/f64dbg.txt
timeout 300 python tools/pca_many_trace.py 2>&1 | grep -v amdgpu.ids > $O/many.txt
timeout 900 python -m pytest tests/test_gpu_pca.py -q -x 2>&1 | tail -5 > $O/pytest_pca.txt
cat $O/f64dbg.txt $O/many.txt $O/pytest_pca.txt
