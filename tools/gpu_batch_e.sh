#!/bin/bash
timeout 600 python tools/time_topk_w.py 2>&1 | grep -v amdgpu.ids
