#!/bin/bash
timeout 300 python tools/time_small_calls.py 2>&1 | grep -v amdgpu.ids
