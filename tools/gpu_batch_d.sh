#!/bin/bash
cp vip_amd/libvipmi.so /tmp/libvipmi.keep
cp vip_amd/csrc/eigh_tri.o /tmp/eigh_tri.keep
( cd vip_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVIPMI_TRI_PROFILE -c eigh_tri.hip -o eigh_tri.o && make ) > /dev/null 2>&1
timeout 300 python tools/tri_profile.py 2>&1 | grep -v amdgpu.ids | grep "reg=1\|update+corner [1-9]"
cp /tmp/eigh_tri.keep vip_amd/csrc/eigh_tri.o; cp /tmp/libvipmi.keep vip_amd/libvipmi.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "eigh or topk or tri or eig" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_pca.py tests/test_gpu_fullsize.py -x -q -m gpu -k "annular or annulus or c3 or 4d or c4" 2>&1 | tail -4
timeout 300 python tools/tri_profile.py 2>&1 | grep "reg=1"
timeout 300 python tools/time_c3.py 2>&1 | grep -v amdgpu | tail -1
