#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
timeout 300 python tools/r5_f64dbg.py 2>&1 | grep -v amdgpu.ids > $O/f64dbg.txt
timeout 300 python tools/h2d_overlap.py 2>&1 | grep -v amdgpu.ids > $O/h2d.txt
timeout 60 python tools/power_probe.py rot1024 2>&1 | grep -v amdgpu.ids > $O/power1024.txt
cat $O/f64dbg.txt $O/h2d.txt $O/power1024.txt
