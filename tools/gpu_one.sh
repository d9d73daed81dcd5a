#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gram" 2>&1 | tail -3
for a in "400 512" "400 512 gram_pack=0" "200 256" "200 256 gram_pack=0" "100 512" "100 512 gram_pack=0" "1000 512" "1000 512 gram_pack=0" "2000 1024" "2000 1024 gram_pack=0"; do timeout 300 python tools/time_gram.py $a 2>&1 | grep -v amdgpu; done
timeout 300 python tools/time_gram_batched.py 2>&1 | grep -v amdgpu | tail -4
