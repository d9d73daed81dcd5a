#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gram" 2>&1 | tail -3
for a in "400 512" "200 256" "2000 1024" "100 512" "50 128"; do timeout 300 python tools/time_gram.py $a 2>&1 | grep -v amdgpu; done
timeout 300 python tools/time_gram_batched.py 2>&1 | grep -v amdgpu | tail -4
