#!/bin/bash
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | tail -1
