#!/bin/bash
cp vip_amd/libvipmi.so /tmp/libvipmi.keep
cp vip_amd/csrc/eigh_tri.o /tmp/eigh_tri.keep
for D in "" "-DVIPMI_EXP_NOTAIL" "-DVIPMI_EXP_NOGROUPS" "-DVIPMI_EXP_NOTAIL -DVIPMI_EXP_NOGROUPS"; do
( cd vip_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DVIPMI_TRI_PROFILE $D -c eigh_tri.hip -o eigh_tri.o && make ) > /dev/null 2>&1
echo "variant: $D"
timeout 300 python tools/tri_profile.py 2>&1 | grep -v amdgpu.ids | grep "reg=1\|wave0-phase [1-9]" | head -2
done
cp /tmp/eigh_tri.keep vip_amd/csrc/eigh_tri.o; cp /tmp/libvipmi.keep vip_amd/libvipmi.so
