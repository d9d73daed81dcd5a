#!/bin/bash
set -x
cp vip_amd/libvipmi.so /tmp/libvipmi.keep
cp vip_amd/csrc/eigh_tri.o /tmp/eigh_tri.keep
( cd vip_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVIPMI_TRI_PROFILE -c eigh_tri.hip -o eigh_tri.o && make )
timeout 300 python tools/tri_profile.py 2>&1 | tail -12
cp /tmp/eigh_tri.keep vip_amd/csrc/eigh_tri.o; cp /tmp/libvipmi.keep vip_amd/libvipmi.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh or topk or tri or eig" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_pca.py -x -q -m gpu -k "annular or annulus" 2>&1 | tail -5
timeout 300 python tools/tri_profile.py 2>&1 | tail -12
timeout 300 python tools/time_c3.py 2>&1 | tail -6
