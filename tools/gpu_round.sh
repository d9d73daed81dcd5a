#!/bin/bash
# Run ON THE GPU BOX: the whole -m gpu suite, the default bench line, the per-config times and the profile round.
R=${1:-r02}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q --timeout=600 --durations=12 2>&1 | tail -24 > gpurun_out/${R}_pytest_gpu.log
timeout 900 python bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err
timeout 900 python tools/time_configs.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${R}_configs.txt
timeout 300 python tools/time_topk.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/${R}_configs.txt
timeout 300 python tools/prof_stage.py pca 400 512 3 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/${R}_configs.txt
timeout 2700 bash tools/profile_round.sh $R > gpurun_out/${R}_profile_round.log 2>&1
