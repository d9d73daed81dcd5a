#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gram" 2>&1 | tail -3
for a in "400 512" "400 512 gram_wpw=8" "200 256" "200 256 gram_wpw=8" "100 512" "100 512 gram_wpw=8" "1000 512" "1000 512 gram_wpw=8"; do timeout 300 python tools/time_gram.py $a 2>&1 | grep -v amdgpu; done
timeout 300 python tools/time_gram_batched.py 2>&1 | grep -v amdgpu | tail -4
timeout 300 python tools/time_c3.py 2>&1 | grep -v amdgpu | tail -1
timeout 300 python tools/time_c4.py 2>&1 | grep -v amdgpu | tail -2
