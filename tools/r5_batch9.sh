#!/bin/bash
# A/B: complex multiply as two asm statements (libvipmi_split.so) against the product library; numpy-in legs alone
O=gpurun_out/r5i; mkdir -p $O
for rep in 1 2; do
for args in "1024 100" "512 400" "256 1600"; do
  timeout 120 python tools/time_rot.py $args 2>&1 | grep -v amdgpu.ids | sed 's/^/base  /' >> $O/ab.txt
  VIPMI_LIB_PATH=$PWD/vip_amd/libvipmi_split.so timeout 120 python tools/time_rot.py $args 2>&1 | grep -v amdgpu.ids | sed 's/^/split /' >> $O/ab.txt
done; done
timeout 300 python - > $O/numpy_legs.txt 2>&1 <<'P'
import numpy as np, torch, bench, json
from vip_amd import backend as B
from vip_amd.psfsub import pca
for rep in range(2):
    r = bench.numpy_in_legs(400, 512, 20, np.linspace(0, 90, 400), 0, pca, B, torch)
    print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in r.items() if k not in ("note", "pipelined")}, r["pipelined"]["ms_per_cube"])
P
cat $O/ab.txt; grep -v amdgpu $O/numpy_legs.txt | tail -4; uptime
