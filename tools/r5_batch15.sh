#!/bin/bash
O=gpurun_out/r5o; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "annular_eigh or eigh_topk" 2>&1 | tail -5 > $O/pytest1.txt
timeout 900 python -m pytest tests/test_gpu_pca.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -q -x -k "annul" 2>&1 | tail -5 > $O/pytest2.txt
for o in "" "ann_gather=0" "" "ann_gather=0"; do VIPMI_OPTS=$o timeout 120 python tools/time_configs.py c3 2>&1 | grep -v amdgpu.ids | sed "s/^/[$o] /" >> $O/c3.txt; done
cat $O/pytest1.txt $O/pytest2.txt $O/c3.txt
