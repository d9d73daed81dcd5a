#!/bin/bash
O=gpurun_out/r5b; mkdir -p $O
for args in "256 1600" "256 1600 rot_1024_q=1" "256 400" "256 400 rot_1024_q=1" "256 1600" "256 1600 rot_1024_q=1"; do
  timeout 120 python tools/time_rot.py $args 2>&1 | grep -v amdgpu.ids >> $O/rot.txt
done
VIPMI_OPTS=rot_1024_q=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "rot" 2>&1 | tail -3 > $O/pytest_q.txt
for w in rot512 rot256 gram idle; do timeout 60 python tools/power_probe.py $w 2>&1 | grep -v amdgpu.ids >> $O/power.txt; done
cat $O/rot.txt $O/pytest_q.txt $O/power.txt
